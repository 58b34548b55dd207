"""`FeatureLearningBlock` wiring of the reference's classification model (models/cls_model.py:10-145)
for the shipped path (`ds_which: token`, `fl_which: n2p`), built from the drop-in modules of this
package.  It reproduces the CALL PROTOCOL only -- submodule names (hence state_dict keys), the order
EdgeConv x2 -> N2P -> [DownSampleToken -> N2P -> gather_by_idx(xyz)] x2, the res-link max-pool heads --
so that a reference checkpoint of the block loads and the sampler can be exercised mid-network.
The MLP head, STN and the segmentation decoder are out of scope."""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .attention import Neighbor2PointAttention, Point2PointAttention
from .downsample import DownSampleGlobal, DownSampleLocal, DownSampleToken
from .embedding import EdgeConv


class FeatureLearningBlock(nn.Module):
    def __init__(self, config_feature_learning_block, fps=False):
        super().__init__()
        cfg = config_feature_learning_block
        self.fps = fps  # reference cls_model.py:100,117-131: FPS pre-selection of 2M points before each sampler
        self.M_list = cfg.downsample.M
        sampler = {"token": DownSampleToken, "global": DownSampleGlobal, "local": DownSampleLocal}.get(cfg.downsample.ds_which)
        if sampler is None:
            raise NotImplementedError
        fl_which = getattr(cfg.attention, "fl_which", "n2p")
        if fl_which not in ("n2p", "p2p"):
            raise ValueError("Only n2p and p2p are valid for fl_which")
        layer_cls = Neighbor2PointAttention if fl_which == "n2p" else Point2PointAttention
        self.res_link_enable = cfg.res_link.enable
        self.embedding_list = nn.ModuleList([EdgeConv(cfg.embedding, l) for l in range(len(cfg.embedding.K))])
        self.downsample_list = nn.ModuleList([sampler(cfg.downsample, l) for l in range(len(cfg.downsample.M))])
        self.feature_learning_layer_list = nn.ModuleList(
            [layer_cls(cfg.attention, l) for l in range(len(cfg.attention.K))])
        outs = cfg.attention.ff_conv2_channels_out
        if self.res_link_enable:
            self.conv_list = nn.ModuleList([nn.Conv1d(c, 1024, kernel_size=1, bias=False) for c in outs])
        else:
            self.conv = nn.Conv1d(outs[-1], 1024, kernel_size=1, bias=False)
        self.M_list = cfg.downsample.M

    def _sample(self, i, x, x_xyz, noise):
        layer = self.downsample_list[i]
        if isinstance(layer, DownSampleToken):
            return layer(x, x_xyz, noise=noise)[0]
        return layer(x, x_xyz)[0]

    def forward(self, x, noise_list=None):
        """x (B,3,N) xyz.  noise_list: optional per-sampler-layer Exp(1) tensors (parity tests)."""
        x_list = []
        x_xyz = x.clone()
        for embedding in self.embedding_list:
            x = embedding(x)
            x_list.append(x)
        x = torch.cat(x_list, dim=1)
        x = self.feature_learning_layer_list[0](x)
        if self.res_link_enable:
            res_link_list = [self.conv_list[0](x).max(dim=-1)[0]]
            for i in range(len(self.downsample_list)):
                noise = None if noise_list is None else noise_list[i]
                if self.fps:
                    x_idx = ops.farthest_point_sample(torch.permute(x_xyz, (0, 2, 1)), self.M_list[i] * 2)
                    x = torch.gather(x, 2, x_idx.unsqueeze(1).expand(-1, x.shape[1], -1))
                    x_xyz_down = torch.gather(x_xyz, 2, x_idx.unsqueeze(1).expand(-1, 3, -1))
                    (x, idx_select) = self._sample(i, x, x_xyz_down, noise)
                    idx_select = torch.gather(x_idx.unsqueeze(1), 2, idx_select)
                else:
                    (x, idx_select) = self._sample(i, x, x_xyz, noise)
                x = self.feature_learning_layer_list[i + 1](x)
                x_xyz = ops.gather_by_idx(x_xyz, idx_select)
                res_link_list.append(self.conv_list[i + 1](x).max(dim=-1)[0])
            self.res_link_list = res_link_list
            return torch.cat(res_link_list, dim=1), res_link_list
        for i in range(len(self.downsample_list)):
            noise = None if noise_list is None else noise_list[i]
            x = self._sample(i, x, None, noise)[0]
            x = self.feature_learning_layer_list[i + 1](x)
        return self.conv(x).max(dim=-1)[0]


class SegFeatureLearningBlock(nn.Module):
    """The segmentation block (reference models/seg_model.py:7-133, seg.yaml): EdgeConv x2 -> N2P ->
    [sampler -> N2P -> gather xyz] x2 -> [UpSampleInterpolation -> N2P] x2, returning per-point features
    (B,128,N).  Wiring only; same submodule names and state_dict keys as the reference block."""

    def __init__(self, config_feature_learning_block):
        super().__init__()
        cfg = config_feature_learning_block
        sampler = {"token": DownSampleToken, "global": DownSampleGlobal, "local": DownSampleLocal}.get(cfg.downsample.ds_which)
        if sampler is None:
            raise ValueError("Only global_carve and local_insert are valid for ds_which!")
        if cfg.upsample.us_which != "interpolation":
            if cfg.upsample.us_which in ("crossA", "selfA"):
                raise NotImplementedError("us_which crossA / selfA are not built (shipped seg.yaml: interpolation)")
            raise ValueError("Only crossA and selfA are valid for us_which!")
        from .upsample import UpSampleInterpolation
        self.embedding_list = nn.ModuleList([EdgeConv(cfg.embedding, l) for l in range(len(cfg.embedding.K))])
        self.downsample_list = nn.ModuleList([sampler(cfg.downsample, l) for l in range(len(cfg.downsample.M))])
        self.feature_learning_layer_list = nn.ModuleList(
            [Neighbor2PointAttention(cfg.attention, l) for l in range(len(cfg.attention.K))])
        self.upsample_list = nn.ModuleList([UpSampleInterpolation(cfg.upsample, l) for l in range(len(cfg.upsample.q_in))])

    def forward(self, x, noise_list=None):
        x_xyz = x[:, :3, :]
        x_list = []
        for embedding in self.embedding_list:
            x = embedding(x)
            x_list.append(x)
        x = torch.cat(x_list, dim=1)
        x = self.feature_learning_layer_list[0](x)
        x_list = [x]
        points_drop_list, idx_select_list, idx_drop_list = [], [], []
        x_xyz_list = [x_xyz]
        for i in range(len(self.downsample_list)):
            layer = self.downsample_list[i]
            if isinstance(layer, DownSampleToken):
                noise = None if noise_list is None else noise_list[i]
                (x, idx_select), (points_drop, idx_drop) = layer(x, x_xyz, noise=noise)
            else:
                (x, idx_select), (points_drop, idx_drop) = layer(x, x_xyz)
            x = self.feature_learning_layer_list[i + 1](x)
            x_xyz = ops.gather_by_idx(x_xyz, idx_select)
            x_list.append(x)
            x_xyz_list.append(x_xyz)
            points_drop_list.append(points_drop)
            idx_select_list.append(idx_select)
            idx_drop_list.append(idx_drop)
        split = int((len(self.feature_learning_layer_list) - 1) / 2)
        x = ((x_list.pop(), idx_select_list.pop(), x_xyz_list.pop()), (points_drop_list.pop(), idx_drop_list.pop()))
        for j in range(len(self.upsample_list)):
            x_tmp = x_list.pop()
            x_xyz_tmp = x_xyz_list[-1 - j]
            x = self.upsample_list[j](x_tmp, x, x_xyz_tmp)
            x = self.feature_learning_layer_list[j + 1 + split](x)
            if j < len(self.upsample_list) - 1:
                x = ((x, idx_select_list.pop(), x_xyz_list[-1 - j]), (points_drop_list.pop(), idx_drop_list.pop()))
        return x


def seg_block_config(M=(1024, 512)):
    """`config.feature_learning_block` of the shipped segmentation preset (seg.yaml)."""
    from .attention import attention_config
    from .config import sampler_config, to_attr
    from .embedding import embedding_config
    from .upsample import upsample_config
    return to_attr(dict(embedding=embedding_config("seg"), downsample=sampler_config("seg", M=list(M)),
                        attention=attention_config("seg"), upsample=upsample_config("seg")))


def block_config(preset: str = "cls", M=(1024, 512)):
    """`config.feature_learning_block` of the shipped classification preset."""
    from .attention import attention_config
    from .config import sampler_config, to_attr
    from .embedding import embedding_config
    att = attention_config(preset)
    att["fl_which"] = "n2p"
    return to_attr(dict(res_link=dict(enable=True), embedding=embedding_config(preset),
                        downsample=sampler_config(preset, M=list(M)), attention=att))

"""Test / bench harness around the drop-in layers: the encoder (and, for segmentation, decoder) of the reference's
feature-learning blocks, so that the sampler can be exercised mid-network and a reference checkpoint of a block
loads (BASELINE.json configs[1] and configs[2]).

Only the CONTRACT is taken from the reference (models/cls_model.py:10-145, models/seg_model.py:7-133): the
submodule attribute names -- they are the state_dict keys -- and what each layer is fed.  The control flow is this
package's own: the encoder produces a list of resolution levels (`_Level`), the classification block pools each
level, the segmentation block walks the list back up.  The MLP heads, STN and trainers stay out of scope
(SURVEY.md section 2)."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import linear, ops
from .attention import Neighbor2PointAttention, Point2PointAttention
from .downsample import DownSampleGlobal, DownSampleLocal, DownSampleToken
from .embedding import EdgeConv

_SAMPLERS = {"token": DownSampleToken, "global": DownSampleGlobal, "local": DownSampleLocal}


@dataclass
class _Level:
    """One resolution of a cloud batch on its way through the encoder."""
    feat: torch.Tensor                      # (B, C, n) features after this level's attention layer
    xyz: torch.Tensor                       # (B, 3, n) coordinates of the same points
    picked: Optional[torch.Tensor] = None   # (B, 1, n) indices of these points in the previous level
    dropped: Tuple = (None, None)           # what the sampler set aside: (features, indices) or (None, None)


class _Encoder(nn.Module):
    """EdgeConv embeddings, one attention layer per resolution, one sampler between two resolutions."""

    def _build_encoder(self, cfg, attention_cls):
        sampler_cls = _SAMPLERS.get(cfg.downsample.ds_which)
        if sampler_cls is None:
            raise NotImplementedError(f"ds_which={cfg.downsample.ds_which!r}: token, global and local are built")
        self.embedding_list = nn.ModuleList(EdgeConv(cfg.embedding, i) for i in range(len(cfg.embedding.K)))
        self.downsample_list = nn.ModuleList(sampler_cls(cfg.downsample, i) for i in range(len(cfg.downsample.M)))
        self.feature_learning_layer_list = nn.ModuleList(
            attention_cls(cfg.attention, i) for i in range(len(cfg.attention.K)))

    def _run_sampler(self, i: int, feat, xyz, noise, forced_idx=None):
        layer = self.downsample_list[i]
        if isinstance(layer, DownSampleToken):
            # the selection noise and (parity tests) the indices to gather are explicit inputs
            return layer(feat, xyz, noise=noise, forced_idx=forced_idx)
        return layer(feat, xyz, forced_idx=forced_idx)

    def _encode(self, xyz: torch.Tensor, noise_list: Optional[Sequence], pre_select=None,
                forced_idx_list: Optional[Sequence] = None, on_level=None) -> List[_Level]:
        """on_level(i, feat) -> feat: called with every level's features before anything else consumes them (the
        classification trunk pools a level there, so that the head and the sampler share one backward node)."""
        stacked = []
        feat = xyz
        for edge_conv in self.embedding_list:      # each EdgeConv feeds the next; all of them are concatenated
            feat = edge_conv(feat)
            stacked.append(feat)
        tap = on_level if on_level is not None else (lambda i, f: f)
        level = _Level(tap(0, self.feature_learning_layer_list[0](torch.cat(stacked, dim=1))), xyz[:, :3, :])
        levels = [level]
        for i in range(len(self.downsample_list)):
            noise = noise_list[i] if noise_list is not None else None
            feat_in, xyz_in, remap = level.feat, level.xyz, None
            if pre_select is not None:             # farthest-point pre-selection in front of the sampler
                feat_in, xyz_in, remap = pre_select(i, level)
            forced = forced_idx_list[i] if forced_idx_list is not None else None
            (feat, picked), dropped = self._run_sampler(i, feat_in, xyz_in, noise, forced)
            if remap is not None:
                picked = torch.gather(remap.unsqueeze(1), 2, picked)
            level = _Level(tap(i + 1, self.feature_learning_layer_list[i + 1](feat)), ops.gather_by_idx(level.xyz, picked),
                           picked, dropped)
            levels.append(level)
        return levels


FUSED_HEADS = True  # False: the stock Conv1d + max (A/B runs)
SPLIT_HEADS = os.environ.get("SAMBLE_SPLIT_HEADS", "1") != "0"  # False: the pooled heads as consumers of their own (a zero fill + a dense add per level in the backward)


def _pooled_head(head: nn.Conv1d, feat: torch.Tensor, args_out: Optional[list] = None, forced_arg=None) -> torch.Tensor:
    """`conv(x).max(dim=-1)[0]` of the classification trunk (models/cls_model.py:113, 136, 144) as one HIP pass: the
    (B, 1024, N) tensor is never written, the backward touches the arg-max columns only (csrc/linear.hip).
    args_out: the arg-max points (B, O) are appended to it (introspection: the point a gradient is routed to)."""
    if FUSED_HEADS and head.bias is None and linear.linear_max_supported(feat, head.weight):
        y, arg = linear._LinearMax.apply(feat, head.weight, forced_arg)
        if args_out is not None:
            args_out.append(arg)
        return y
    if forced_arg is not None:
        raise NotImplementedError("forced_arg needs the fused pooled head")
    y, arg = head(feat).max(dim=-1)
    if args_out is not None:
        args_out.append(arg)
    return y


class FeatureLearningBlock(_Encoder):
    """Classification trunk: every level is projected to 1024 channels and max-pooled; the pooled vectors are
    concatenated (`res_link.enable`), or only the last level is pooled.  Returns (B, 1024 * levels) and the list
    of pooled vectors, like the reference block."""

    def __init__(self, config_feature_learning_block, fps=False):
        super().__init__()
        cfg = config_feature_learning_block
        which = getattr(cfg.attention, "fl_which", "n2p")
        if which not in ("n2p", "p2p"):
            raise ValueError("Only n2p and p2p are valid for fl_which")
        self.fps = fps
        self.M_list = cfg.downsample.M
        self.res_link_enable = cfg.res_link.enable
        self._build_encoder(cfg, Neighbor2PointAttention if which == "n2p" else Point2PointAttention)
        widths = cfg.attention.ff_conv2_channels_out
        if self.res_link_enable:
            self.conv_list = nn.ModuleList(nn.Conv1d(c, 1024, kernel_size=1, bias=False) for c in widths)
        else:
            self.conv = nn.Conv1d(widths[-1], 1024, kernel_size=1, bias=False)

    def _fps_subset(self, i: int, level: _Level):
        """2 M_i farthest points of the level (reference cls_model.py:117-131, only with fps=True)."""
        keep = ops.farthest_point_sample(level.xyz.permute(0, 2, 1), self.M_list[i] * 2)
        take = keep.unsqueeze(1)
        return (torch.gather(level.feat, 2, take.expand(-1, level.feat.shape[1], -1)),
                torch.gather(level.xyz, 2, take.expand(-1, 3, -1)), keep)

    def forward(self, x, noise_list=None, forced_idx_list=None, forced_head_args=None):
        """x (B,3,N) coordinates.  noise_list: optional per-sampler Exp(1) tensors.  forced_idx_list: parity-test hook,
        per sampler the indices to gather instead of its own selection (the samplers' `forced_idx`): everything
        behind a sampler is then compared on the reference's own point set, whatever a near-tie did to the selection.
        forced_head_args: the same for the pooled heads' arg-max points (`linear._LinearMax`)."""
        from .attention import deferred_batch_counts
        n_heads = len(self.conv_list) if self.res_link_enable else 1
        self.head_args = [None] * n_heads   # per pooled head: the arg-max point of every (cloud, output)
        fa = forced_head_args
        fused = self.res_link_enable and FUSED_HEADS and SPLIT_HEADS
        pooled_early = []

        def pool_level(i, feat):
            """the level's pooled head where the level is produced: head and onward path as ONE autograd node"""
            head = self.conv_list[i]
            if not (head.bias is None and linear.linear_max_supported(feat, head.weight)):
                pooled_early.append(None)
                return feat
            y, arg, onward = linear._LinearMaxSplit.apply(feat, head.weight, fa[i] if fa is not None else None)
            pooled_early.append(y)
            self.head_args[i] = arg
            return onward

        def pool_late(i, head, feat):
            got = []
            y = _pooled_head(head, feat, got, fa[i] if fa is not None else None)
            self.head_args[i] = got[-1]
            return y

        with deferred_batch_counts():
            levels = self._encode(x, noise_list, self._fps_subset if (self.fps and self.res_link_enable) else None,
                                  forced_idx_list, pool_level if fused else None)
        if not self.res_link_enable:
            return pool_late(0, self.conv, levels[-1].feat)
        early = pooled_early if fused else [None] * len(levels)
        pooled = [y if y is not None else pool_late(i, head, level.feat)
                  for i, (y, head, level) in enumerate(zip(early, self.conv_list, levels))]
        self.res_link_list = pooled
        self.level_feats = [level.feat.detach() for level in levels]
        return torch.cat(pooled, dim=1), pooled


class SegFeatureLearningBlock(_Encoder):
    """Segmentation trunk: the encoder's levels, then one interpolation upsampling + attention layer per level
    on the way back up; returns per-point features (B, 128, N)."""

    def __init__(self, config_feature_learning_block):
        super().__init__()
        cfg = config_feature_learning_block
        if cfg.upsample.us_which != "interpolation":
            if cfg.upsample.us_which in ("crossA", "selfA"):
                raise NotImplementedError("us_which crossA / selfA are not built (shipped seg.yaml: interpolation)")
            raise ValueError("Only crossA and selfA are valid for us_which!")
        from .upsample import UpSampleInterpolation
        self._build_encoder(cfg, Neighbor2PointAttention)
        self.upsample_list = nn.ModuleList(UpSampleInterpolation(cfg.upsample, i) for i in range(len(cfg.upsample.q_in)))

    def forward(self, x, noise_list=None, forced_idx_list=None):
        from .attention import deferred_batch_counts
        with deferred_batch_counts():
            return self._forward(x, noise_list, forced_idx_list)

    def _forward(self, x, noise_list, forced_idx_list):
        levels = self._encode(x, noise_list, forced_idx_list=forced_idx_list)
        first_decoder_layer = 1 + (len(self.feature_learning_layer_list) - 1) // 2
        coarse = levels[-1]
        for j, upsample in enumerate(self.upsample_list):
            fine = levels[-2 - j]
            # the reference's upsampling layers take the coarse side as ((features, indices, xyz), (dropped features,
            # dropped indices)) -- kept, so that the layer stays a drop-in
            merged = upsample(fine.feat, ((coarse.feat, coarse.picked, coarse.xyz), coarse.dropped), fine.xyz)
            coarse = _Level(self.feature_learning_layer_list[first_decoder_layer + j](merged), fine.xyz, fine.picked,
                            fine.dropped)
        return coarse.feat


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

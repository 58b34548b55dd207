#!/usr/bin/env python3
"""Golden fixture for the classification FeatureLearningBlock call protocol (BASELINE.json
configs[1] at a small size): the UNMODIFIED reference block (models/cls_model.py:10-145, cls.yaml)
on CPU with deterministic parameters, forward AND backward under a fixed upstream gradient (the gradients of a
spread of parameters are stored: first EdgeConv, both samplers, an attention layer behind each sampler, the last
projection).  Run from the repo root:
    python tests/golden/make_golden_block.py"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("SAMBLE_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch

from samble_amd import synth
from oracle import torch_oracle as O
from tests.golden.make_golden_layers import block_config

from models import cls_model as ref_cls  # noqa: E402


from tests.util import fill_parameters  # noqa: E402  (same deterministic fill as the GPU test)


GRAD_KEYS_CLS = ("embedding_list.0.conv1.0.weight", "embedding_list.1.conv2.0.weight",
                 "downsample_list.0.bin_tokens", "downsample_list.0.q_conv.weight", "downsample_list.0.v_conv.weight",
                 "downsample_list.1.bin_tokens", "downsample_list.1.k_conv.weight",
                 "feature_learning_layer_list.0.ff.0.weight", "feature_learning_layer_list.1.q_conv.weight",
                 "feature_learning_layer_list.1.bn1.weight", "feature_learning_layer_list.2.v_conv.weight",
                 "conv_list.0.weight", "conv_list.2.weight")
GRAD_KEYS_SEG = ("embedding_list.0.conv1.0.weight", "downsample_list.0.bin_tokens", "downsample_list.0.k_conv.weight",
                 "downsample_list.1.bin_tokens", "downsample_list.1.q_conv.weight",
                 "feature_learning_layer_list.1.q_conv.weight", "feature_learning_layer_list.2.ff.2.weight",
                 "feature_learning_layer_list.3.k_conv.weight", "feature_learning_layer_list.4.bn2.bias",
                 "feature_learning_layer_list.4.ff.2.weight", "upsample_list.0.conv.0.weight",
                 "upsample_list.1.res_conv.0.weight", "upsample_list.1.res_conv.1.weight")
ROW_STRIDE = 8   # wide matrices are stored every 8th output row (the fixture stays small)


def thin(a: np.ndarray) -> np.ndarray:
    return a[::ROW_STRIDE] if (a.ndim >= 2 and a.shape[0] >= 512) else a


def grads_of(blk, keys):
    params = dict(blk.named_parameters())
    return {"grad/" + k: thin(params[k].grad.detach().numpy()) for k in keys}


def main():
    torch.set_num_threads(8)
    B, N, M, seed = 2, 256, [128, 64], 9100
    cfg = block_config("cls")
    cfg.downsample.M = list(M)
    blk = ref_cls.FeatureLearningBlock(cfg)
    fill_parameters(blk, seed)
    blk.train()
    xyz = torch.from_numpy(synth.xyz_clouds(B, N, seed + 500))
    nb = cfg.downsample.bin.num_bins[0]
    torch.manual_seed(seed)
    feat, res = blk(xyz)
    feat.backward(torch.from_numpy(synth.normal(tuple(feat.shape), seed + 900)))   # the test's upstream gradient
    torch.manual_seed(seed)
    noise0 = O.draw_noise(B * nb, N)
    noise1 = O.draw_noise(B * nb, M[0])
    out = dict(meta=np.array([B, N, M[0], M[1], nb, seed], dtype=np.int64), feat=feat.detach().numpy(),
               **grads_of(blk, GRAD_KEYS_CLS),
               noise0=noise0.numpy(), noise1=noise1.numpy(),
               idx0=blk.downsample_list[0].idx.numpy(), idx1=blk.downsample_list[1].idx.numpy(),
               score0=blk.downsample_list[0].attention_point_score.detach().numpy(),
               names=np.array([n for n, _ in blk.named_parameters()]), torch_version=np.array(torch.__version__))
    path = os.path.join(HERE, "block_cls_small.npz")
    np.savez_compressed(path, **out)
    print("block_cls_small: ok,", os.path.getsize(path) // 1024, "KiB; params", sum(p.numel() for p in blk.parameters()))


def seg_main():
    """The segmentation block (models/seg_model.py:7-133, seg.yaml: down 256 -> 128 -> 64, interpolation up)."""
    from models import seg_model as ref_seg
    torch.set_num_threads(8)
    B, N, M, seed = 2, 256, [128, 64], 9300
    cfg = block_config("seg")
    cfg.downsample.M = list(M)
    blk = ref_seg.FeatureLearningBlock(cfg)
    fill_parameters(blk, seed)
    blk.train()
    xyz = torch.from_numpy(synth.xyz_clouds(B, N, seed + 500))
    nb = cfg.downsample.bin.num_bins[0]
    torch.manual_seed(seed)
    feat = blk(xyz)
    feat.backward(torch.from_numpy(synth.normal(tuple(feat.shape), seed + 900)))
    torch.manual_seed(seed)
    noise0 = O.draw_noise(B * nb, N)
    noise1 = O.draw_noise(B * nb, M[0])
    out = dict(meta=np.array([B, N, M[0], M[1], nb, seed], dtype=np.int64), feat=feat.detach().numpy(),
               **grads_of(blk, GRAD_KEYS_SEG),
               noise0=noise0.numpy(), noise1=noise1.numpy(),
               idx0=blk.downsample_list[0].idx.numpy(), idx1=blk.downsample_list[1].idx.numpy(),
               names=np.array([n for n, _ in blk.named_parameters()]), torch_version=np.array(torch.__version__))
    path = os.path.join(HERE, "block_seg_small.npz")
    np.savez_compressed(path, **out)
    print("block_seg_small: ok,", os.path.getsize(path) // 1024, "KiB; params", sum(p.numel() for p in blk.parameters()))


if __name__ == "__main__":
    main()
    seg_main()

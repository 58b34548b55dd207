#!/usr/bin/env python3
"""Golden fixtures for Point2PointAttention (reference models/attention.py:253-355) as the reference configures
it -- 4 heads of 32 channels -- for asm dot / l2 / l2+, from the UNMODIFIED reference on CPU.

    python tests/golden/make_golden_p2p.py

Parameters are filled deterministically by position (tests/util.fill_parameters does the same for the build's
module); train mode; one forward + backward.  Only data is written."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.environ.get("SAMBLE_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch

from make_golden_layers import block_config  # noqa: E402  (same yaml loader)
from models import attention as ref_att  # noqa: E402  (the reference)
from samble_amd import synth
from util import fill_parameters  # noqa: E402


def run(asm, seed, C=128, heads=4, name=None):
    B, N = 2, 320
    cfg = block_config("cls").attention
    cfg.asm[0] = asm
    cfg.num_heads[0] = heads
    for key in ("q_in", "q_out", "k_in", "k_out", "v_in", "v_out", "ff_conv1_channels_in", "ff_conv2_channels_out"):
        cfg[key][0] = C
    cfg.ff_conv1_channels_out[0] = cfg.ff_conv2_channels_in[0] = 4 * C
    mod = ref_att.Point2PointAttention(cfg, 0)
    fill_parameters(mod, seed)
    mod.train()
    x = torch.from_numpy(synth.features(B, C, N, seed + 10) * 0.5).requires_grad_(True)
    y = mod(x)
    g = torch.from_numpy(synth.normal((B, C, N), seed + 20))
    y.backward(g)
    out = dict(meta=np.array([B, C, N, cfg.num_heads[0], seed], dtype=np.int64), asm=np.array(asm),
               torch_version=np.array(torch.__version__), y=y.detach().numpy(), dx=x.grad.numpy())
    for pname, p in mod.named_parameters():
        out["grad__" + pname] = p.grad.numpy()
    name = name or "layer_p2p_" + {"dot": "dot", "l2": "l2", "l2+": "l2plus"}[asm]
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: ok, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    for i, asm in enumerate(("dot", "l2", "l2+")):
        run(asm, 9100 + 50 * i)
    # a shape outside the multi-head kernels (64 channels, 8 heads of 8): samble_amd/attention.py runs it in torch
    run("l2", 9400, C=64, heads=8, name="layer_p2p_c64_heads8_l2")

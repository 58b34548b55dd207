#!/usr/bin/env python3
"""Golden fixture for the sampler's residual link (reference models/downsample.py:75-83, 292-298:
`res.enable = True`, `res.ff = True`) from the UNMODIFIED reference on CPU in the build container.

    python tests/golden/make_golden_res.py

The reference module is built from the reference's own yaml with res.enable / res.ff switched on, every
parameter is filled deterministically by position (tests/util.fill_parameters does the same for the build's
module), train mode (BatchNorm uses batch statistics and updates its running buffers), one forward +
backward with the selection noise replayed by seed.  Only data is written (inputs by seed, outputs,
gradients by parameter name, BatchNorm buffers after the step)."""
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

from make_golden import reference_config  # noqa: E402  (same yaml loader)
from models import downsample as ref_ds  # noqa: E402  (the reference)
from oracle import torch_oracle as O
from samble_amd import synth
from util import fill_parameters  # noqa: E402

CASES = [dict(name="res_ff_topk", sample_mode="topk", ff=True), dict(name="res_noff_random", sample_mode="random", ff=False)]


def run(case, seed):
    B, C, N, M = 2, 128, 256, 128
    cfg = reference_config("cls")
    cfg.res.enable[0] = True
    cfg.res.ff[0] = case["ff"]
    cfg.bin.sample_mode[0] = case["sample_mode"]
    mod = ref_ds.DownSampleToken(cfg, 0)
    mod.M = M
    fill_parameters(mod, seed)
    mod.train()
    nb = mod.num_bins
    x = torch.from_numpy(synth.features(B, C, N, seed + 10)).requires_grad_(True)
    torch.manual_seed(seed + 7)
    (x_ds, idx), _ = mod(x)
    torch.manual_seed(seed + 7)
    noise = O.draw_noise(B * nb, N)
    g = torch.from_numpy(synth.normal((B, C, M), seed + 99))
    x_ds.backward(g)
    out = dict(meta=np.array([B, C, N, M, nb, seed], dtype=np.int64), sample_mode=np.array(case["sample_mode"]),
               ff=np.array(case["ff"]), torch_version=np.array(torch.__version__),
               noise=noise.numpy(), idx=idx.numpy(), x_ds=x_ds.detach().numpy(), dx=x.grad.numpy(),
               score=mod.attention_point_score.detach().numpy())
    for name, p in mod.named_parameters():
        out["grad__" + name] = p.grad.numpy()
    for name, b in mod.named_buffers():
        out["buf__" + name] = b.detach().numpy()
    path = os.path.join(HERE, "layer_" + case["name"] + ".npz")
    np.savez_compressed(path, **out)
    print(f"{case['name']}: ok, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    torch.set_num_threads(8)
    for i, case in enumerate(CASES):
        run(case, 7300 + 100 * i)

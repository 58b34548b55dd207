#!/usr/bin/env python3
"""Per-fixture, per-cloud identity of the module's sampled indices with the reference's (both matrix modes):
the measured table that tests/test_gpu_module.py pins.  Run on the GPU box."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from samble_amd import ops
from tests.util import Golden, golden_names

out = {}
for mode in ("tri", "f32"):
    ops.MATRIX_MODE = mode
    for name in golden_names():
        g = Golden(name)
        mod = g.module("cuda:0")
        rows = []
        for call in range(g.calls):
            noise = None if g.sample_mode == "topk" else g.t("noise", call).to("cuda:0")
            (x_ds, idx), _ = mod(g.x(call).to("cuda:0"), noise=noise)
            same = (idx.cpu()[:, 0] == g.t("idx", call)[:, 0]).all(1)
            cdiff = (mod.k_point_to_choose.cpu() != g.t("counts", call)).any(1)
            rows.append({"same": same.int().tolist(), "counts_differ": cdiff.int().tolist()})
        out[f"{mode}/{name}"] = rows
    # the slim full-size fixtures the reference wrote (tests/golden/make_golden_headline.py): the headline configuration
    # B=32, N=2048 -> 1024 and the stress geometry B=2, N=8192 -> 4096
    import numpy as np
    from tests.test_gpu_module import SLIM_FIXTURES, headline_module_and_step
    for name in SLIM_FIXTURES:
        d, mod, idx, _, _ = headline_module_and_step("cuda:0", name)
        same = (idx.cpu()[:, 0] == torch.from_numpy(d["idx"].astype(np.int64))).all(1)
        cdiff = (mod.k_point_to_choose.cpu() != torch.from_numpy(d["counts"])).any(1)
        out[f"{mode}/{name}"] = [{"same": same.int().tolist(), "counts_differ": cdiff.int().tolist()}]
print(json.dumps(out, indent=1))

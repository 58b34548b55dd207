"""Timing-only ablations of knn_tri (wrong outputs for codes >= 100)."""
import sys, time
import torch
sys.path.insert(0, ".")
from samble_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
B, C, N, K = 32, 128, 2048, 32
x = torch.randn(B, C, N, generator=torch.Generator().manual_seed(0)).to(dev)
for code, name in ((48, "budget 48"), (48, "budget 48"), (200, "adaptive keep 0"), (202, "adaptive keep 2"), (204, "adaptive keep 4"), (206, "adaptive keep 6"), (210, "adaptive keep 10")):
    lib.samble_knn_tri_config(1, code)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ops.stage_knn(x, x, K)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%-44s %.1f us (incl. ~25 us split + rownorm)" % (name, dt * 1e6))
lib.samble_knn_tri_config(1, 205)

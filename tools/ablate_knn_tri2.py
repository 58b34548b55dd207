"""Slope of knn_tri over the number of key tiles (timing-only ablations for codes >= 100)."""
import sys, time
import torch
sys.path.insert(0, ".")
from samble_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
B, C, Nq, K = 32, 128, 2048, 32
a = torch.randn(B, C, Nq, generator=torch.Generator().manual_seed(0)).to(dev)
for code in (103, 3):
    for Nk in (512, 1024, 2048, 4096):
        x = torch.randn(B, C, Nk, generator=torch.Generator().manual_seed(Nk)).to(dev)
        lib.samble_knn_tri_config(1, code)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                ops.stage_knn(a, x, K)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("code %3d Nk %5d  %.1f us" % (code, Nk, dt * 1e6))
lib.samble_knn_tri_config(1, 3)

"""Dev check of the split-bf16 kNN kernel against the fp32-MFMA one (GPU)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from samble_amd import ops, _lib
from tests.util import set_agreement
dev = torch.device("cuda:0")
lib = _lib.load()
B, C, N, K = 32, 128, 2048, 32
x = torch.randn(B, C, N, generator=torch.Generator().manual_seed(0)).to(dev)
lib.samble_knn_tri_config(0, 0)
ref = ops.stage_knn(x, x, K)
res = {}
for steps in (3, 24):
    lib.samble_knn_tri_config(1, steps)
    got = ops.stage_knn(x, x, K)
    torch.cuda.synchronize()
    print("steps", steps, "agreement with fp32 kernel", set_agreement(got[:4].cpu(), ref[:4].cpu()),
          "exact rows", (got == ref).all(-1).float().mean().item())
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ops.stage_knn(x, x, K)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("   tri knn (split + rownorm + kernel) %.1f us" % (dt * 1e6))
lib.samble_knn_tri_config(0, 0)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        ops.stage_knn(x, x, K)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("fp32 knn %.1f us" % (dt * 1e6))
# ragged sizes and cross sets
lib.samble_knn_tri_config(1, 205)
for (Nq, Nk) in ((1000, 1000), (333, 1500), (96, 64)):
    a = torch.randn(2, C, Nq, generator=torch.Generator().manual_seed(Nq)).to(dev)
    bb = a if Nq == Nk else torch.randn(2, C, Nk, generator=torch.Generator().manual_seed(Nk)).to(dev)
    got = ops.stage_knn(a, bb, K)
    d = ((a.double().permute(0, 2, 1)[:, :, None, :] - bb.double().permute(0, 2, 1)[:, None, :, :]) ** 2).sum(-1)
    want = d.topk(K, dim=-1, largest=False)[1]
    print((Nq, Nk), "agreement with fp64", set_agreement(got.cpu(), want.cpu()))

"""Timing-only ablations of attn_stats_tri (wrong outputs): where does the time go?"""
import sys, time
import torch
sys.path.insert(0, ".")
from samble_amd import ops, _lib
dev = torch.device("cuda:0")
B, N, nt, D = 32, 2048, 6, 128
qkv = torch.randn(B, N + nt, 3 * D, generator=torch.Generator().manual_seed(0)).to(dev)
q, k = qkv[:, :N, :D], qkv[:, :, D:2 * D]
ops.MATRIX_MODE = "tri"
imgs = (ops.stage_tri_split(q)[0], ops.stage_tri_split(k)[0])
lib = _lib.load()
for mode, name in ((0, "full"), (1, "no map stores"), (2, "no MFMA"), (3, "neither"), (4, "same phase order"), (5, "same order, no stores")):
    lib.samble_debug_ablate(1, mode)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            ops.stage_attn_stats(q, k, N, nt, images=imgs)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%-24s %.1f us" % (name, dt * 1e6))
lib.samble_debug_ablate(1, 0)

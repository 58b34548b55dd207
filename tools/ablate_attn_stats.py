#!/usr/bin/env python3
"""Timing ablation of attn_stats_kernel on the metric shape (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from samble_amd import _lib, ops, synth
B, N, nt, D = 32, 2048, 6, 128
qkv = torch.from_numpy(synth.normal((B, N + nt, 3 * D), 1) * 0.5).cuda()
q, k, v = qkv[:, :N, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / it
lib = _lib.load()
flops = 2 * N * (N + nt) * D * B
for mode, name in ((0, "real kernel"), (1, "map stores skipped"), (2, "tile staging skipped"),
                   (3, "stores hit the same lines each tile"), (4, "stores into a compact 128 KB scratch")):
    lib.samble_debug_ablate(1, mode)
    ms = t(lambda: ops.stage_attn_stats(q, k, N, nt))
    print(f"{name:28s} {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s")
lib.samble_debug_ablate(1, 0)

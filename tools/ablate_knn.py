#!/usr/bin/env python3
"""Timing ablation of the fused kNN kernel (GPU box): full kernel vs the same kernel with the top-K
selection compiled out vs the round-1 two-kernel path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from samble_amd import _lib, ops, synth
x = torch.from_numpy(synth.features(32, 128, 2048, 1)).cuda()
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / it
lib = _lib.load()
for mode, name in ((0, "stream (default, drains keep 6 entries)"), (200, "stream, drains to empty"), (206, "stream keep 6"),
                   (99, "stream, no insertions (filter + ring only; wrong results)"), (206, "keep 6"),
                   (3, "fused (previous generation)"), (2, "fused, selection ablated"), (1, "two-kernel (key matrix via HBM)")):
    lib.samble_knn_force_unfused(mode)
    print(f"{name:35s} {t(lambda: ops.stage_knn(x, x, 32)):.3f} ms")
lib.samble_knn_force_unfused(0)

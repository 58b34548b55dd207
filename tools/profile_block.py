#!/usr/bin/env python3
"""Per-kernel table of steady-state FeatureLearningBlock steps (torch.profiler, device activities only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from samble_amd import synth
from samble_amd.blocks import FeatureLearningBlock, block_config
B, N = 32, 2048
blk = FeatureLearningBlock(block_config("cls")).cuda().train()
xyz = torch.from_numpy(synth.xyz_clouds(B, N, 77)).cuda()
def step():
    for p in blk.parameters():
        p.grad = None
    out, _ = blk(xyz)
    out.sum().backward()
for _ in range(4):
    step()
torch.cuda.synchronize()
STEPS = 4
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print(f"device time per step: {tot / STEPS / 1e3:.2f} ms")
for e in rows[:45]:
    print(f"{e.key[:78]:78s} {e.count // STEPS:4d}/step {e.device_time_total / STEPS:9.1f} us/step")

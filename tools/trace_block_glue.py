#!/usr/bin/env python3
"""Which Python lines launch the stock (non-samble) kernels of a block step: torch.profiler with stacks, one step.
    python tools/trace_block_glue.py [cls|seg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from samble_amd import synth
from samble_amd.blocks import FeatureLearningBlock, SegFeatureLearningBlock, block_config, seg_block_config
seg = (sys.argv[1] if len(sys.argv) > 1 else "seg") == "seg"
dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = (SegFeatureLearningBlock(seg_block_config()) if seg else FeatureLearningBlock(block_config("cls"))).to(dev).train()
xyz = torch.from_numpy(synth.xyz_clouds(32, 2048, 77)).to(dev)
opt = torch.optim.SGD(blk.parameters(), lr=1e-3)
def step():
    opt.zero_grad(set_to_none=True)
    out = blk(xyz)
    out = out[0] if isinstance(out, (tuple, list)) else out
    out.float().square().mean().backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
# every aten op that launched a stock kernel, grouped by (op, the innermost frame of this package on its stack)
from collections import defaultdict
groups = defaultdict(lambda: [0.0, 0])
bwd = defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    dt = getattr(ev, "self_device_time_total", 0)
    if dt <= 0 or not ev.name.startswith("aten::"):
        continue
    frames = [f for f in (ev.stack or []) if "samble_amd/" in f or "trace_block_glue" in f]
    where = frames[0].split("samble_amd/")[-1][:90] if frames else "(autograd engine thread: no Python frame)"
    g = groups[(ev.name, where)]
    g[0] += dt
    g[1] += 1
tot = sum(v[0] for v in groups.values())
print(f"stock kernels launched by aten ops: {tot:.0f} us in {sum(v[1] for v in groups.values())} launches per step")
for (name, where), (dt, n) in sorted(groups.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{dt:8.1f} us  x{n:<3d} {name:28s} {where}")
# backward nodes of the autograd graph that own stock kernels (the engine thread has no Python stack)
for ev in prof.key_averages():
    dt = getattr(ev, "device_time_total", 0)
    if dt > 0 and ("Backward" in ev.key or "AccumulateGrad" in ev.key) and "evaluate_function" not in ev.key:
        print(f"   node {ev.key[:50]:50s} x{ev.count:<3d} {dt:8.1f} us (kernels of any kind under it)")

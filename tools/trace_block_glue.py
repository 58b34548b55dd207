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
rows = []
for ev in prof.key_averages(group_by_stack_n=6):
    dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
    if dt <= 0: continue
    stack = [s for s in ev.stack if "samble_amd" in s or "bench" in s]
    rows.append((dt, ev.count, ev.key, stack[:2]))
rows.sort(key=lambda r: -r[0])
for dt, n, key, stack in rows[:45]:
    print(f"{dt:9.1f} us  x{n:<3d} {key[:48]:48s} {' | '.join(s.split('/')[-1][:70] for s in stack)}")

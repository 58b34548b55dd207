#!/usr/bin/env python3
"""Stock aten ops of one block step grouped by (op, input shapes, forward | backward thread): which tensors the glue
launches work on.    python tools/trace_block_shapes.py [cls|seg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import threading
import torch
from collections import defaultdict
from torch.profiler import profile, ProfilerActivity
from samble_amd import synth
from samble_amd.blocks import FeatureLearningBlock, SegFeatureLearningBlock, block_config, seg_block_config
seg = (sys.argv[1] if len(sys.argv) > 1 else "cls") == "seg"
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
main_tid = threading.get_native_id()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
groups = defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    dt = getattr(ev, "self_device_time_total", 0)
    if dt <= 0:
        continue
    where = "fwd" if ev.thread == main_tid or ev.thread == threading.get_ident() else "bwd"
    shapes = str([s for s in (ev.input_shapes or []) if s])[:110]
    g = groups[(ev.name, where, shapes)]
    g[0] += dt; g[1] += 1
tot = sum(v[0] for v in groups.values()); n = sum(v[1] for v in groups.values())
print(f"stock aten launches: {n} per step, {tot:.0f} us")
for (name, where, shapes), (dt, c) in sorted(groups.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{dt:8.1f} us x{c:<3d} {where} {name:26s} {shapes}")

#!/usr/bin/env python3
"""Timing of BASELINE.json configs[1] geometry on the GPU box: the whole FeatureLearningBlock
(EdgeConv x2 -> N2P -> sampler 2048->1024 -> N2P -> sampler 1024->512 -> N2P), forward + backward,
B=32, with a per-module breakdown from torch events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from samble_amd import synth
from samble_amd.blocks import FeatureLearningBlock, block_config

B, N = 32, 2048
dev = "cuda:0"
blk = FeatureLearningBlock(block_config("cls")).to(dev).train()
xyz = torch.from_numpy(synth.xyz_clouds(B, N, 77)).to(dev)

def step():
    for p in blk.parameters():
        p.grad = None
    out, _ = blk(xyz)
    out.sum().backward()

for _ in range(3):
    step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    step()
b.record(); b.synchronize()
print(f"block fwd+bwd: {a.elapsed_time(b) / 5:.2f} ms / step  ({B * 5 / a.elapsed_time(b) * 1e3:.0f} clouds/s)")

# forward-only breakdown per top-level module
times = {}
def hook_pair(name, mod):
    ev = {}
    def pre(m, inp):
        ev["s"] = torch.cuda.Event(enable_timing=True); ev["s"].record()
    def post(m, inp, out):
        e = torch.cuda.Event(enable_timing=True); e.record(); times.setdefault(name, []).append((ev["s"], e))
    mod.register_forward_pre_hook(pre); mod.register_forward_hook(post)
for i, m in enumerate(blk.embedding_list): hook_pair(f"edgeconv{i}", m)
for i, m in enumerate(blk.feature_learning_layer_list): hook_pair(f"n2p{i}", m)
for i, m in enumerate(blk.downsample_list): hook_pair(f"sampler{i}", m)
with torch.no_grad():
    blk(xyz); times.clear(); blk(xyz)
torch.cuda.synchronize()
for k, v in times.items():
    print(f"  forward {k:10s} {sum(s.elapsed_time(e) for s, e in v):7.3f} ms")

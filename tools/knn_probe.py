#!/usr/bin/env python3
"""kNN alone at the metric shape (B=32, C=128, N=2048, K=32), for rocprofv3 passes: tools/pmc_deep.sh with
PMC_CMD="python3 tools/knn_probe.py"."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from samble_amd import ops, synth

B, C, N, K = 32, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 32
x = torch.from_numpy(synth.features(B, C, N, 2001)).cuda()
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    idx = ops.stage_knn(x, x, K)
torch.cuda.synchronize()
print(idx.sum().item())

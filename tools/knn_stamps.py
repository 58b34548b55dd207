#!/usr/bin/env python3
"""Phase cycles of knn_duo_kernel from a -DSAMBLE_KNN_STAMP scratch build (tools/knn_stamp_run.sh): the distance
output carries the stamps of every wave.  usage: knn_stamps.py <lib.so> [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from samble_amd import _lib, synth
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
B, C, K = 32, 128, 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
x = torch.from_numpy(synth.features(B, C, N, 2001)).cuda()
idx = torch.empty((B, N, K), dtype=torch.int32, device="cuda")
dist = torch.zeros((B, N, K), dtype=torch.float32, device="cuda")
nbytes = _lib.query("samble_knn_workspace_bytes", B, C, N, N, K, 0)
ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
for _ in range(3):
    # raw call: the dist post-processing kernel would overwrite the stamps, so read the workspace copy instead
    _lib.call("samble_knn_f32", x.data_ptr(), C * N, N, x.data_ptr(), C * N, N, B, C, K, 0, idx.data_ptr(), dist.data_ptr(),
              ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
# keys buffer inside the workspace: [knorm B*N][qnorm B*N][scale B][keys B*N*K]
keys = ws.view(torch.float32)[2 * B * N + B: 2 * B * N + B + B * N * K].view(B, N // 32, 32, K)[:, :, 0, :12]
names = ["seed", "products", "select", "barrier", "loop_end", "total", "prunes", "max ring", "mean ring", "seed:dma", "seed:wait", "seed:barrier"]
m = keys.reshape(-1, 12).double()
m = m[m[:, 5] > 0]  # rows that carry stamps (one per wave)
for i, n in enumerate(names):
    col = m[:, i]
    print(f"{n:12s} mean {col.mean().item():12.0f}  min {col.min().item():12.0f}  max {col.max().item():12.0f}")

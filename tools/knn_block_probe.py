#!/usr/bin/env python3
"""The kNN calls of one FeatureLearningBlock forward (real intermediate features, not the synthetic Gaussian ones), each
replayed on a -DSAMBLE_KNN_STAMP scratch build: time, prunes and ring occupancy per call.
usage: knn_block_probe.py <stamped lib.so>   (tools/knn_stamp_run.sh builds tools/scratch/lib_knn_stamps.so)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from samble_amd import ops, synth
from samble_amd.blocks import FeatureLearningBlock, block_config

dev = torch.device("cuda:0")
torch.manual_seed(1000)
blk = FeatureLearningBlock(block_config("cls")).to(dev).train()
xyz = torch.from_numpy(synth.xyz_clouds(32, 2048, 77)).to(dev)
calls = []
orig = ops.stage_knn
def spy(a, b, K, *args, **kw):
    if a.shape[1] in (64, 128):
        calls.append((a.detach().clone(), K))
    return orig(a, b, K, *args, **kw)
ops.stage_knn = spy
blk(xyz)
ops.stage_knn = orig
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
lib.samble_knn_workspace_bytes.restype = ctypes.c_size_t
lib.samble_knn_workspace_bytes.argtypes = [ctypes.c_int] * 5 + [ctypes.c_int]
lib.samble_knn_f32.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_long,
                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                               ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
for x, K in calls:
    B, C, N = x.shape
    x = x.contiguous()
    idx = torch.empty((B, N, K), dtype=torch.int32, device=dev)
    dist = torch.zeros((B, N, K), dtype=torch.float32, device=dev)
    nbytes = lib.samble_knn_workspace_bytes(B, C, N, N, K, 0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: lib.samble_knn_f32(x.data_ptr(), C * N, N, x.data_ptr(), C * N, N, B, C, K, 0, idx.data_ptr(), dist.data_ptr(),
                                     ws.data_ptr(), nbytes, st)
    for _ in range(2): assert run() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    keys = ws.view(torch.float32)[2 * B * N + B: 2 * B * N + B + B * N * K].view(B, N // 32, 32, K)[:, :, 0, :12]
    m = keys.reshape(-1, 12).double()
    m = m[m[:, 5] > 0]
    xc = x - x.mean(dim=2, keepdim=True)
    print(f"C={C} N={N} K={K}: {e0.elapsed_time(e1) / 5 * 1e3:7.1f} us (prep + kernel)  prunes/wave {m[:, 6].mean():5.1f}  max ring {m[:, 7].max():3.0f} "
          f"mean ring {m[:, 8].mean():5.1f}  |x-mean| rms {xc.pow(2).sum(1).sqrt().mean():8.3f}  distinct rows {torch.unique(x[0].T, dim=0).shape[0]}")

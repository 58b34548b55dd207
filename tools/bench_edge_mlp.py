#!/usr/bin/env python3
"""edge_mlp_fwd / edge_mlp_bwd alone (B=32, N=2048, K=32, 64 channels) on every library given: real kNN lists of a synthetic
cloud and index-local lists (neighbours = the next 32 points), so that the gather's share shows.
    python tools/bench_edge_mlp.py samble_amd/libsamble_hip.so tools/scratch/edge/lib_*.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from samble_amd import ops, synth
B, N, K, C = 32, 2048, 32, 64
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.features(B, 64, N, 5)).to(dev)
nn_real = ops.stage_knn(x, x, K).contiguous()
nn_local = ((torch.arange(N, device=dev)[:, None] + torch.arange(K, device=dev)[None, :]) % N).int().expand(B, -1, -1).contiguous()
g = torch.Generator(device=dev).manual_seed(1)
ap = torch.randn((B, N, C), device=dev, generator=g); bp = torch.randn((B, N, C), device=dev, generator=g)
W2 = torch.randn((C, C), device=dev, generator=g) * 0.1
kext = torch.randint(0, K, (B, N, C), device=dev, generator=g).to(torch.uint8)
sdv = torch.randn((B, N, C), device=dev, generator=g); c0c1 = torch.randn((2, C), device=dev, generator=g) * 0.01
du = torch.empty((B, N, K, C), device=dev); dusum = torch.empty((B, N, C), device=dev)
vp = ctypes.c_void_p
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    nparts = lib.samble_edge_partial_count()
    dwp = torch.empty((nparts, C, C), device=dev); part = torch.empty((nparts, 2, C), dtype=torch.float64, device=dev)
    ymax = torch.empty((B, N, C), device=dev); ymin = torch.empty_like(ymax)
    kmax = torch.empty((B, N, C), dtype=torch.uint8, device=dev); kmin = torch.empty_like(kmax)
    fb = lib.samble_edge_mlp_bwd_f32; fb.argtypes = [vp] * 7 + [ctypes.c_int] * 4 + [vp] * 4
    ff = lib.samble_edge_mlp_fwd_f32; ff.argtypes = [vp] * 4 + [ctypes.c_int] * 4 + [vp] * 6
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for nn in (nn_real, nn_local):
        def bwd():
            assert fb(ap.data_ptr(), bp.data_ptr(), nn.data_ptr(), W2.data_ptr(), kext.data_ptr(), sdv.data_ptr(), c0c1.data_ptr(),
                      B, N, K, C, du.data_ptr(), dusum.data_ptr(), dwp.data_ptr(), st) == 0
        def fwd():
            assert ff(ap.data_ptr(), bp.data_ptr(), nn.data_ptr(), W2.data_ptr(), B, N, K, C, ymax.data_ptr(), ymin.data_ptr(),
                      kmax.data_ptr(), kmin.data_ptr(), part.data_ptr(), st) == 0
        for fn in (fwd, bwd):
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{os.path.basename(path):28s} kNN lists: fwd {out[0]:6.1f} bwd {out[1]:6.1f} us | local lists: fwd {out[2]:6.1f} bwd {out[3]:6.1f} us")
    if hasattr(lib, "samble_scratch_edge_stamps"):  # a -DSAMBLE_STAMPS build: cycles between the marks of edge_mlp_bwd_tri
        buf = (ctypes.c_ulonglong * 160)()
        lib.samble_scratch_edge_stamps.argtypes = [vp]
        assert lib.samble_scratch_edge_stamps(buf) == 0
        v = list(buf)
        names = ["gather + y (both)", "dy", "dW2", "dh", "du tile", "du stores", "dusum"]
        for wv in range(8):
            for it in range(2):
                st_ = v[(wv * 2 + it) * 10:(wv * 2 + it) * 10 + 8]
                print(f"  wave {wv} point {2 + it}: " + "  ".join(f"{names[i]} {st_[i + 1] - st_[i]}" for i in range(7)) + f"  | total {st_[7] - st_[0]}")

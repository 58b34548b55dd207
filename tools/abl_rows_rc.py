"""Timing-only ablations of attn_rows_rc_tri: each lib_<mask>.so is the library with SAMBLE_RC_ABL=<mask>."""
import ctypes, glob, os, sys, torch
B, N, nt, M = 32, 2048, 6, 1024
NK = N + nt
dev = torch.device("cuda:0")
tiles = lambda r: (r + 31) // 32
g = torch.Generator(device=dev).manual_seed(1)
def img(rows):
    return (torch.randn(B * tiles(rows) * 24576 // 2, device=dev, generator=g) * 0.3).to(torch.bfloat16).view(torch.uint8)
q, k, v = img(N), img(NK), img(NK)
lse = torch.full((B, N), 8.0, device=dev)
idx = torch.stack([torch.randperm(N, device=dev)[:M] for _ in range(B)]).contiguous()
out = torch.empty((B, 128, M), device=dev)
ld = 32 * tiles(NK)
pmap = torch.empty((B, M, ld), device=dev)
for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scratch", "abl", "lib_*.so")), key=lambda s: int(s.split("_")[-1][:-3])):
    lib = ctypes.CDLL(f)
    fn = lib.samble_attn_rows_fwd_recompute_tri_f32
    fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    def run(pm):
        rc = fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), lse.data_ptr(), idx.data_ptr(), B, N, nt, M, 128, out.data_ptr(),
                pmap.data_ptr() if pm else None, ld, st)
        assert rc == 0, rc
    res = []
    for pm in (True, False):
        for _ in range(3): run(pm)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(pm)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{os.path.basename(f):12s} pmap {res[0]:7.1f} us   no pmap {res[1]:7.1f} us")

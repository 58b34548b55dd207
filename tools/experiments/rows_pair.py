"""The rows-pair experiment (tools/experiments/attn_rows_pair.hip) against the library's one-wave kernel on the metric's
shape: bitwise comparison of x_ds and the P map, event timing of both, and the s_memtime marks of one pair of waves.
Libraries: tools/scratch/pair/lib_<ablation mask>.so (tools/experiments/run_rows_pair.sh).
The bitwise comparison held at commit 1f6d637; the library's kernel has since moved its logit products to two fp16 planes
(a different K image): "DIFFERENT" is expected now, the timings and stamps remain meaningful."""
import ctypes, glob, os, torch
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
ld = 32 * tiles(NK)
names = ["top", "first reads", "unpack, DMA issue"] + [f"slots {6 * i}-{6 * i + 5}" for i in range(8)] + ["x out + waits", "barrier"]

def timed(call):
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3

for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scratch", "pair", "lib_*.so")),
                key=lambda s: int(s.split("_")[-1][:-3])):
    lib = ctypes.CDLL(f)
    one = lib.samble_attn_rows_fwd_recompute_tri_f32
    one.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    pair = lib.samble_experiment_attn_rows_pair_tri
    pair.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    mask = int(os.path.basename(f).split("_")[-1][:-3])
    for pm in (False, True):
        outs = [torch.empty((B, 128, M), device=dev) for _ in range(2)]
        maps = [torch.zeros((B, M, ld), device=dev) if pm else None for _ in range(2)]
        mp = lambda i: maps[i].data_ptr() if pm else None
        run_one = lambda: one(q.data_ptr(), k.data_ptr(), v.data_ptr(), lse.data_ptr(), idx.data_ptr(), B, N, nt, M, 128,
                              outs[0].data_ptr(), mp(0), ld, st)
        run_pair = lambda: pair(q.data_ptr(), k.data_ptr(), v.data_ptr(), lse.data_ptr(), idx.data_ptr(), B, N, nt, M,
                                128 ** -0.5, outs[1].data_ptr(), mp(1), ld, st)
        assert run_one() == 0 and run_pair() == 0
        torch.cuda.synchronize()
        same = torch.equal(outs[0], outs[1]) and (not pm or torch.equal(maps[0], maps[1]))
        print(f"mask {mask:2d} {'P map' if pm else 'no map'}: one wave {timed(run_one):6.1f} us, pair {timed(run_pair):6.1f} us, "
              + ("bitwise equal" if same else "DIFFERENT" + (" (expected: ablation)" if mask else "")))
        buf = (ctypes.c_ulonglong * 32)()
        lib.samble_scratch_pr_stamps.argtypes = [ctypes.c_void_p]
        assert lib.samble_scratch_pr_stamps(buf) == 0
        s = list(buf)
        for c in range(2):
            w = s[16 * c:16 * c + 13]
            print(f"    wave {'AB'[c]}: " + " ".join(f"{names[i]}:{w[i] - w[i - 1]}" for i in range(1, 13)) + f" | total {w[12] - w[0]}")

"""Timing-only ablations of bwd_dq_tri (wrong results for modes != 0), kernel time from the library's HIP events."""
import sys, time
import torch
sys.path.insert(0, ".")
from samble_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
B, N, nt, D, M = 32, 2048, 6, 128, 1024
qkv = torch.randn(B, N + nt, 3 * D, generator=torch.Generator().manual_seed(0)).to(dev)
q, k, v = qkv[:, :N, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
ops.MATRIX_MODE = "tri"
imgs = ops.stage_tri_split_qkv(qkv, N, for_backward=True)
smap, lse, _ = ops.stage_attn_stats(q, k, N, nt, images=imgs[:2])
idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(bb))[:M] for bb in range(B)]).to(dev)
x_ds = ops.stage_attn_rows(smap, lse, v, idx, N, nt, v_image=imgs[2])
g = torch.randn(B, D, M, generator=torch.Generator().manual_seed(1)).to(dev)
dq = torch.empty(B, N, D, device=dev); dk = torch.empty(B, N + nt, D, device=dev); dv = torch.empty(B, N + nt, D, device=dev)
for mode, name in ((0, "full"), (21, "dq: tiles staged once"), (22, "dq: no matrix products"), (23, "dq: neither")):
    lib.samble_debug_ablate(1, mode)
    for kid, kname in ((6, "dq"), (3, "dkdv")):
        lib.samble_debug_time_kernel(kid)
        for rep in range(2):
            for _ in range(5):
                ops.stage_attn_rows_bwd(q, k, v, smap, lse, x_ds, idx, g, N, nt, dq, dk, dv, images=imgs[3:])
            torch.cuda.synchronize()
            ms = lib.samble_debug_kernel_ms()
        print("%-28s %-5s %.1f us" % (name, kname, ms * 1e3))
lib.samble_debug_ablate(1, 0)

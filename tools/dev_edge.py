"""Edge shapes through the split-bf16 kernels: no tokens, tiny clouds, B not a multiple of 8."""
import math, sys
import torch
sys.path.insert(0, ".")
from samble_amd import ops
dev = torch.device("cuda:0")
ops.MATRIX_MODE = "tri"
for (B, N, nt, M) in ((1, 64, 0, 32), (3, 257, 0, 100), (5, 96, 2, 96), (2, 33, 6, 1), (9, 2048, 6, 1024)):
    D = 128
    g0 = torch.Generator().manual_seed(N + nt)
    qkv = torch.randn(B, N + nt, 3 * D, generator=g0).to(dev)
    q, k, v = qkv[:, :N, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b))[:M] for b in range(B)]).to(dev)
    g = torch.randn(B, D, M, generator=g0).to(dev)
    qd, kd, vd = (t.double().detach().requires_grad_(True) for t in (q, k, v))
    s = (qd @ kd.transpose(1, 2)) / math.sqrt(D)
    o = torch.softmax(s, -1) @ vd
    rows = torch.gather(o, 1, idx[..., None].expand(-1, -1, D))
    rows.permute(0, 2, 1).backward(g.double())
    smap, lse, tok = ops.stage_attn_stats(q, k, N, nt)
    x_ds = ops.stage_attn_rows(smap, lse, v, idx, N, nt)
    dq = torch.full((B, N, D), float("nan"), device=dev)
    dk = torch.full((B, N + nt, D), float("nan"), device=dev)
    dv = torch.full((B, N + nt, D), float("nan"), device=dev)
    ops.stage_attn_rows_bwd(q, k, v, smap, lse, x_ds, idx, g, N, nt, dq, dk, dv)
    errs = [float((x_ds.double() - rows.detach().permute(0, 2, 1)).abs().max())]
    for got, ref in ((dq, qd.grad), (dk, kd.grad), (dv, vd.grad)):
        errs.append(float((got.double() - ref).abs().max() / (ref.abs().max() + 1e-30)))
    print((B, N, nt, M), "x_ds abs err %.2e  dq/dk/dv rel err %.2e %.2e %.2e" % tuple(errs),
          "finite", bool(torch.isfinite(dq).all() and torch.isfinite(dk).all() and torch.isfinite(dv).all()))

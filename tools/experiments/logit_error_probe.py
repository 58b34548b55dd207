"""GPU: error of the logits S = Q K^T / sqrt(D) against fp64 for the fp32-MFMA kernel and the default kernel (two fp16 planes
under power-of-two scales, three products) at several operand scales, one key row 50x the others."""
import os, sys, math, torch
sys.path.insert(0, os.getcwd())
from samble_amd import ops as o_, synth
B, N, nt = 2, 1024, 6
g = torch.Generator().manual_seed(1)
for scale_q, scale_k in ((1, 1), (1e-3, 1e-3), (1e3, 1e-2), (30, 30)):
    q = torch.randn(B, N, 128, generator=g) * scale_q
    k = torch.randn(B, N + nt, 128, generator=g) * scale_k
    k[:, 5] *= 50  # one large key row in the first tile
    s = (q.double() @ k.double().transpose(1, 2)) / math.sqrt(128)
    res = {}
    for mode in ("f32", "tri"):
        o_.MATRIX_MODE = mode
        smap, lse, _ = o_.stage_attn_stats(q.cuda(), k.cuda(), N, nt)
        d = smap[:, :, :N + nt].cpu().double() - s
        res[mode] = (d.pow(2).mean().sqrt().item() / s.pow(2).mean().sqrt().item(), d.abs().max().item() / s.abs().max().item())
    print(f"|q|~{scale_q:g} |k|~{scale_k:g}: rms err / rms(S): fp32-MFMA {res['f32'][0]:.2e}, fp16x3 {res['tri'][0]:.2e};  max err / max|S|: {res['f32'][1]:.2e}, {res['tri'][1]:.2e}")

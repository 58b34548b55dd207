import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from samble_amd import ops, synth
from samble_amd.attention import _attention_from_projection
B, C, N, K, H = 2, 128, 256, 32, 4
qkv = torch.from_numpy(synth.normal((B, N, 3 * C), 31) * 0.5).cuda()
g = torch.from_numpy(synth.normal((B, C, N), 32)).cuda()
nn_idx = torch.stack([torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b * N + i))[:K]
                                   for i in range(N)]) for b in range(B)]).int().cuda()
for diff in (False, True):
    part = qkv.detach().clone().requires_grad_(True)
    out = _attention_from_projection(part, nn_idx, H, diff)
    ref = torch.autograd.grad(out, part, g)[0]
    got = ops.stage_n2p_attn_bwd(qkv, nn_idx, g, H, diff)
    for name, sl in (("dQ", slice(0, C)), ("dK", slice(C, 2 * C)), ("dV", slice(2 * C, 3 * C))):
        r, o = ref[..., sl], got[..., sl]
        print(f"diff={diff} {name}: max|err| {(o - r).abs().max().item():.3e} of max {r.abs().max().item():.3e}; "
              f"ratio of norms {o.norm().item() / r.norm().item():.4f}; per-head err",
              [round((o[..., 32*h:32*h+32] - r[..., 32*h:32*h+32]).abs().max().item(), 4) for h in range(4)])

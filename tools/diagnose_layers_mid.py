#!/usr/bin/env python3
"""Layer-by-layer check at the mid-size block's shapes (B=4; N = 1024, 512, 256): each drop-in layer against a float64 /
stock-torch evaluation of the same expression on the device.  Written to find what tests/golden/block_cls_mid.npz flagged."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from samble_amd import synth, attention as A, embedding as E, ops, blocks as BK, linear as L

dev = "cuda:0"
rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))

def bn_check(B, C, N):
    gen = torch.Generator().manual_seed(B * 1000 + N)
    x = (torch.randn(B, C, N, generator=gen) * 0.7 + torch.randn(1, C, 1, generator=gen) * 2.0).to(dev)
    g = (torch.randn(B, C, N, generator=gen) + 0.5 * torch.randn(1, C, 1, generator=gen)).to(dev)
    gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
    beta = torch.randn(C, generator=gen).to(dev)
    y, m, v, _ = ops.stage_bn_train(x, gamma, beta, None, None, 0.1, 1e-5)
    dx, dg, db = ops.stage_bn_train_bwd(x, g, gamma, m, v)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = torch.nn.functional.batch_norm(xd, None, None, gd, bd, True, 0.1, 1e-5)
    yr.backward(g.double())
    print(f"BN ({B},{C},{N}): y {rel(y, yr.detach()):.1e} dx {rel(dx, xd.grad):.1e} dgamma {rel(dg, gd.grad):.1e} dbeta {rel(db, bd.grad):.1e}", flush=True)

def n2p_check(B, N):
    torch.manual_seed(5)
    cfg = A.attention_config("cls")
    one = A.Neighbor2PointAttention(cfg, 0).to(dev).train()
    with torch.no_grad():
        for p in one.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    ref = copy.deepcopy(one)
    x = torch.from_numpy(synth.features(B, 128, N, 3100 + N)).to(dev)
    g = torch.from_numpy(synth.normal((B, 128, N), 3200 + N)).to(dev)
    outs = []
    for mod, stock in ((one, False), (ref, True)):
        old = (A.FUSED_LAYER, A.OWN_BATCHNORM, A.FUSED_FFN)
        if stock:
            A.FUSED_LAYER, A.OWN_BATCHNORM, A.FUSED_FFN = False, False, False
            mod.hip_attention = False
        try:
            xin = x.clone().requires_grad_(True)
            y = mod(xin)
            y.backward(g)
        finally:
            A.FUSED_LAYER, A.OWN_BATCHNORM, A.FUSED_FFN = old
        outs.append((y.detach(), xin.grad, {n: p.grad for n, p in mod.named_parameters()}))
    (y1, dx1, g1), (y2, dx2, g2) = outs
    print(f"N2P ({B},128,{N}) fused vs stock torch: y {rel(y1, y2):.1e} dx {rel(dx1, dx2):.1e} " +
          " ".join(f"{n}={rel(g1[n], g2[n]):.0e}" for n in g2), flush=True)

def edge_check(B, N, layer):
    torch.manual_seed(11)
    fused = E.EdgeConv(E.embedding_config("cls"), layer).to(dev).train()
    with torch.no_grad():
        for p in fused.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    stock = copy.deepcopy(fused)
    stock.fused = False
    C = 3 if layer == 0 else 64
    x = (torch.from_numpy(synth.xyz_clouds(B, N, 50)) if layer == 0 else torch.from_numpy(synth.features(B, 64, N, 51))).to(dev)
    g = torch.from_numpy(synth.normal((B, 64, N), 60)).to(dev)
    res = []
    for m in (fused, stock):
        xin = x.clone().requires_grad_(True)
        y = m(xin)
        y.backward(g)
        res.append((y.detach(), xin.grad, {n: p.grad for n, p in m.named_parameters()}))
    (y1, dx1, g1), (y2, dx2, g2) = res
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    print(f"EdgeConv layer {layer} ({B},{C},{N}) fused vs stock: y {l2(y1, y2):.1e} dx {l2(dx1, dx2):.1e} " +
          " ".join(f"{n}={l2(g1[n], g2[n]):.0e}" for n in g2), flush=True)

def head_check(B, N):
    torch.manual_seed(3)
    head = torch.nn.Conv1d(128, 1024, 1, bias=False).to(dev)
    x = torch.from_numpy(synth.features(B, 128, N, 70 + N)).to(dev)
    g = torch.from_numpy(synth.normal((B, 1024), 71)).to(dev)
    res = []
    for fusedh in (True, False):
        old = BK.FUSED_HEADS
        BK.FUSED_HEADS = fusedh
        try:
            xin = x.clone().requires_grad_(True)
            head.zero_grad()
            y = BK._pooled_head(head, xin)
            y.backward(g)
        finally:
            BK.FUSED_HEADS = old
        res.append((y.detach(), xin.grad.clone(), head.weight.grad.clone()))
    print(f"pooled head ({B},128,{N}): y {rel(res[0][0], res[1][0]):.1e} dx {rel(res[0][1], res[1][1]):.1e} dW {rel(res[0][2], res[1][2]):.1e}", flush=True)

def sampler_check(B, N, M):
    from oracle import torch_oracle as O
    from samble_amd import sampler_config
    from samble_amd.downsample import DownSampleToken
    nb, C, seed = 6, 128, 4242
    mod = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0)
    wq, wk, wv, tok = synth.sampler_weights(C, nb, seed)
    with torch.no_grad():
        mod.q_conv.weight.copy_(torch.from_numpy(wq)); mod.k_conv.weight.copy_(torch.from_numpy(wk))
        mod.v_conv.weight.copy_(torch.from_numpy(wv)); mod.bin_tokens.copy_(torch.from_numpy(tok))
    mod = mod.to(dev)
    x_np, noise_np, g_np = synth.features(B, C, N, 77), synth.exp1((B * nb, N), 78), synth.normal((B, C, M), 79)
    spec = O.SamplerSpec(M=M, K=32, C=C, num_bins=nb)
    st = O.SamplerState(*(torch.from_numpy(a.copy()) for a in (wq, wk, wv, tok)))
    ref = O.sampler_grads(spec, st, torch.from_numpy(x_np), torch.from_numpy(g_np), torch.from_numpy(noise_np))
    x = torch.from_numpy(x_np).to(dev).requires_grad_(True)
    (x_ds, idx), _ = mod(x, noise=torch.from_numpy(noise_np).to(dev), forced_idx=ref["idx"].to(dev))
    x_ds.backward(torch.from_numpy(g_np).to(dev))
    print(f"sampler ({B},128,{N}->{M}) vs CPU oracle through its indices: x_ds {rel(x_ds.detach().cpu(), ref['x_ds']):.1e} dx {rel(x.grad.cpu(), ref['dx']):.1e} "
          f"dwq {rel(mod.q_conv.weight.grad.cpu(), ref['dwq']):.1e} dwk {rel(mod.k_conv.weight.grad.cpu(), ref['dwk']):.1e} "
          f"dwv {rel(mod.v_conv.weight.grad.cpu(), ref['dwv']):.1e} dtok {rel(mod.bin_tokens.grad.cpu(), ref['dtokens']):.1e}", flush=True)

for N in (1024, 512, 256):
    bn_check(4, 128, N)
for N in (1024, 512, 256):
    n2p_check(4, N)
edge_check(4, 1024, 0)
edge_check(4, 1024, 1)
for N in (1024, 512, 256):
    head_check(4, N)
sampler_check(4, 1024, 512)
sampler_check(4, 512, 256)

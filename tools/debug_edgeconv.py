#!/usr/bin/env python3
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from samble_amd import synth
from samble_amd.embedding import EdgeConv, embedding_config
def w(shape, seed, scale):
    return torch.from_numpy((synth.normal(shape, seed).astype(np.float64) * scale).astype(np.float32))
for layer, B, N in ((0, 1, 2048), (0, 2, 256), (0, 1, 1024), (1, 1, 2048)):
    cfg = embedding_config("cls"); cin = cfg.conv1_in[layer] // 2; seed = 9100 + 10 * layer + N
    mod = EdgeConv(cfg, layer)
    with torch.no_grad():
        mod.conv1[0].weight.copy_(w(tuple(mod.conv1[0].weight.shape), seed + 1, 0.3 if cin == 3 else 0.1))
        mod.conv2[0].weight.copy_(w((64, 64, 1, 1), seed + 2, 0.12))
    mod = mod.cuda().train(); ref = copy.deepcopy(mod); ref.fused = False
    x_np = synth.xyz_clouds(B, N, seed) if cin == 3 else synth.features(B, cin, N, seed)
    x = torch.from_numpy(x_np).cuda().requires_grad_(True); xr = torch.from_numpy(x_np).cuda().requires_grad_(True)
    g = torch.from_numpy(synth.normal((B, 64, N), seed + 20)).cuda()
    y = mod(x); yr = ref(xr); y.backward(g); yr.backward(g)
    print(f"layer {layer} B {B} N {N}: y err {(y-yr).abs().max().item():.2e} of {yr.abs().max().item():.2e}")
    def rel(a, b): return f"{(a-b).abs().max().item():.3e} of {b.abs().max().item():.3e}"
    print("   dx", rel(x.grad, xr.grad))
    for (n1, p1), (_, p2) in zip(mod.named_parameters(), ref.named_parameters()):
        print("  ", n1, rel(p1.grad, p2.grad))
    # the reference twice: its own run-to-run / fp noise level
    xr2 = torch.from_numpy(x_np).cuda().double().requires_grad_(True)
    ref64 = copy.deepcopy(ref).double(); ref64.fused = False
    y64 = ref64(xr2); y64.backward(g.double())
    print("   dx stock32 vs stock64", rel(xr.grad.double(), xr2.grad), " fused vs stock64", rel(x.grad.double(), xr2.grad))

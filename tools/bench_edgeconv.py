#!/usr/bin/env python3
"""EdgeConv layer timing (B=32, N=2048, shipped cls shapes), fused HIP body vs the stock torch composition."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from samble_amd import synth
from samble_amd.embedding import EdgeConv, embedding_config
B, N = 32, 2048
def t(fn, it=5):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / it
for layer in (0, 1):
    cfg = embedding_config("cls"); cin = cfg.conv1_in[layer] // 2
    mod = EdgeConv(cfg, layer).cuda().train(); ref = copy.deepcopy(mod); ref.fused = False
    x_np = synth.xyz_clouds(B, N, 5) if cin == 3 else synth.features(B, cin, N, 5)
    g = torch.from_numpy(synth.normal((B, 64, N), 6)).cuda()
    def run(m):
        x = torch.from_numpy(x_np).cuda().requires_grad_(True)
        def f():
            for p in m.parameters(): p.grad = None
            x.grad = None
            m(x).backward(g)
        return f
    def fwd(m):
        x = torch.from_numpy(x_np).cuda()
        def f():
            with torch.no_grad(): m(x)
        return f
    print(f"layer {layer} (C_in {cin}): fused fwd {t(fwd(mod)):.3f} ms, fwd+bwd {t(run(mod)):.3f} ms | stock fwd {t(fwd(ref)):.3f} ms, fwd+bwd {t(run(ref)):.3f} ms")

#!/usr/bin/env python3
"""Stage-by-stage comparison of the GPU path with a golden fixture (run on the GPU box):
    python tools/diagnose_fixture.py cls_random_cfg1
Prints where, if anywhere, the GPU's integers first leave the reference's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from tests.util import Golden, set_agreement

name = sys.argv[1] if len(sys.argv) > 1 else "cls_random_cfg1"
g = Golden(name)
dev = "cuda:0"
mod = g.module(dev)
for call in range(g.calls):
    x = g.x(call).to(dev)
    noise = None if g.sample_mode == "topk" else g.t("noise", call).to(dev)
    (x_ds, idx), _ = mod(x, noise=noise)
    print(f"== {name} call {call}: B={g.B} N={g.N} M={g.M} nb={g.nb}")
    knn_g = np.sort(mod.knn_idx.cpu().numpy(), -1)
    knn_r = g.t("knn_sorted", call).numpy().astype(np.int64)
    rows_bad = (knn_g != knn_r).any(-1).sum()
    print(f"knn: set agreement {set_agreement(mod.knn_idx.cpu(), g.t('knn_sorted', call).long()):.6f}; rows differing {rows_bad} of {g.B*g.N}")
    indeg_bad = (mod.knn_indegree.cpu() != g.t("indeg", call)).sum().item()
    print(f"indeg: entries differing {indeg_bad}")
    s, sr = mod.attention_point_score.cpu(), g.t("score", call)
    rel = ((s - sr).abs() / sr.abs().clamp_min(1e-30))
    print(f"score: max rel err {rel.max().item():.3e}; median {rel.median().item():.3e}; >1e-5: {(rel>1e-5).sum().item()}")
    z, zr = mod.normalized_score.cpu(), g.t("z", call).reshape(g.B, g.N)
    print(f"z: max abs err {(z-zr).abs().max().item():.3e}")
    up, upr = mod.bin_boundaries[0].cpu(), g.t("upper", call)
    print(f"upper gpu {up.flatten().tolist()}\nupper ref {upr.flatten().tolist()}")
    bits = mod._member_bits.cpu().long()
    bin_id = torch.log2(bits.float()).round().to(torch.int8)
    bad_bins = (bin_id != g.t("bin_id", call)).sum().item()
    print(f"bin ids differing: {bad_bins}; cap diff {(mod.max_num_points.cpu().long()-g.t('cap',call)).abs().sum().item()}")
    print(f"w_pre max abs err {(mod.bin_weights_beforerelu.cpu()-g.t('w_pre',call)).abs().max().item():.3e}")
    cd = (mod.k_point_to_choose.cpu() != g.t("counts", call)).any(1).sum().item()
    print(f"counts: clouds differing {cd}")
    a, b = idx.cpu()[:, 0], g.t("idx", call)[:, 0]
    print(f"idx: set agreement {set_agreement(a, b):.6f}; positions differing {(a!=b).sum().item()} of {a.numel()}; clouds identical {(a==b).all(1).sum().item()} of {g.B}")
    # stage-wise: inject the reference's score -> does the select path reproduce idx exactly?
    from samble_amd import ops
    sc = g.t("score", call).reshape(g.B, g.N).to(dev)
    zz = ops.stage_zscore(sc)
    zbad = (zz.cpu() != zr).sum().item()
    print(f"zscore(ref score): entries not bit-equal to reference z: {zbad} of {zr.numel()}")

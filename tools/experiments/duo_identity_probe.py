#!/usr/bin/env python3
"""CPU experiment (no GPU, no library): would a two-fp16-plane operand scheme for the ATTENTION keep the sampled indices?

The oracle's sampler forward is run three ways on the same inputs and noise:
  ref   as it is (MKL sgemm logits)
  f32*  the logits formed in float64 from the fp32 operands and rounded once to fp32 -- another, equally valid fp32
        evaluation (what the split-bf16 kernels are, to within 2^-23 |q||k|): the flip rate of "a different summation
        order" alone
  duo   the same, with q and k first rounded to two fp16 planes under a per-cloud power-of-two scale (22 significant bits,
        DESIGN.md section 3c) -- what a three-product scheme would compute
and the sampled index sets are compared with `ref` per cloud.  usage: duo_identity_probe.py [B] [N] [M] [seeds]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import torch_oracle as O
from samble_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
M = int(sys.argv[3]) if len(sys.argv) > 3 else N // 2
SEEDS = int(sys.argv[4]) if len(sys.argv) > 4 else 3
torch.set_num_threads(8)


def duo_round(t, cloud_dims):
    """two fp16 planes under a per-cloud power-of-two scale that puts max|t| below 2^13"""
    amax = t.abs().amax(dim=cloud_dims, keepdim=True).clamp_min(1e-30)
    s = torch.exp2(12 - torch.floor(torch.log2(amax)))
    ts = t * s
    h = ts.half().float()
    l = (ts - h).half().float()
    return (h + l) / s


mode = "ref"
orig_logits = O.attention_logits


def logits(q, k, asm="dot"):
    if mode == "ref" or asm != "dot":
        return orig_logits(q, k, asm)
    if mode == "duo":
        q, k = duo_round(q, (1, 2, 3)), duo_round(k, (1, 2, 3))
    return (q.double() @ k.double()).float() / math.sqrt(q.shape[-1])


O.attention_logits = logits
tot = {"f32*": [0, 0, 0], "duo": [0, 0, 0]}
for seed in range(SEEDS):
    for sample_mode, idx_mode in (("random", "sparse_col_sqr"), ("topk", "sparse_col_sqr"), ("uniform", "sparse_row_std")):
        spec = O.SamplerSpec(M=M, K=32, C=128, num_bins=6, idx_mode=idx_mode, sample_mode=sample_mode)
        w = synth.sampler_weights(128, 6, 7000 + seed)
        x = torch.from_numpy(synth.features(B, 128, N, 7100 + seed))
        noise = O.draw_noise(B * 6, N, torch.Generator().manual_seed(seed))
        out = {}
        for mode in ("ref", "f32*", "duo"):
            st = O.SamplerState(*(torch.from_numpy(a.copy()) for a in w))
            _, idx = O.sampler_forward(spec, st, x, noise=noise.clone())
            out[mode] = (idx.reshape(B, -1), st.trace["score"].reshape(B, -1), st.trace["counts"].reshape(B, -1))
        for m in ("f32*", "duo"):
            same = sum(int(torch.equal(out[m][0][b].sort().values, out["ref"][0][b].sort().values)) for b in range(B))
            diff = sum(len(set(out[m][0][b].tolist()) ^ set(out["ref"][0][b].tolist())) // 2 for b in range(B))
            cnt = sum(int(not torch.equal(out[m][2][b], out["ref"][2][b])) for b in range(B))
            err = float((out[m][1] - out["ref"][1]).abs().max() / out["ref"][1].abs().max())
            tot[m][0] += same; tot[m][1] += diff; tot[m][2] += cnt
            print(f"seed {seed} {sample_mode:8s} {idx_mode:15s} {m:5s}: clouds identical {same}/{B}, indices differing {diff:4d}, "
                  f"clouds with other bin counts {cnt}, score max rel err {err:.2e}")
n = SEEDS * 3 * B
for m in tot:
    print(f"TOTAL {m:5s}: identical clouds {tot[m][0]}/{n}, differing indices {tot[m][1]}, clouds with other counts {tot[m][2]}")

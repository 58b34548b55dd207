#!/usr/bin/env python3
"""The HEADLINE configuration pinned against the unmodified reference (BASELINE.json metric: ModelNet40 cls layer 0,
B=32, C=128, N=2048 -> M=1024, 6 bins, K=32, sparse_col_sqr, dynamic boundaries from a fresh state, random T=0.1):
the reference's `DownSampleToken` (models/downsample.py:15-378 with utils/ops.py:174-236, 385-619) runs forward + backward
on CPU in the build container; what a parity test needs is kept SLIM -- the sampled indices as int16, the per-bin counts,
capacities and weights, the boundaries, the bin of every point as int8, per-cloud float64 sums of x_ds / dx, the four
parameter gradients -- plus the Exp(1) selection noise torch.multinomial consumed, (192, 2048) float32: the CPU generator's
stream is not bit-reproducible across hosts (vector width of its log), so the seed alone would not pin it.

Run from the repo root:   python tests/golden/make_golden_headline.py      (~1 minute; writes headline_cls_B32_N2048.npz)
Only data is written: no reference source or bytecode leaves /root/reference."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import make_golden as G  # the reference import, config loader and the oracle == reference assertions
from make_golden import O, same, synth

NAME = "headline_cls_B32_N2048"
CASE = dict(name=NAME, cfg="cls", B=32, N=2048, M=1024, calls=1, big=False)
SEED = 1000 * 77
# BASELINE.json configs[4]'s geometry (N = 8192 -> 4096: multi-tile selection chain, 257 key tiles per row) on two clouds --
# the reference materialises ~10 (B, N, N + 6) float32 tensors, 537 MB each at B = 2
STRESS = ("stress_cls_B2_N8192", dict(name="stress_cls_B2_N8192", cfg="cls", B=2, N=8192, M=4096, calls=1, big=False), 1000 * 78)


def main(name=NAME, case=CASE, seed=SEED):
    global NAME, CASE, SEED
    NAME, CASE, SEED = name, case, seed
    torch.set_num_threads(8)
    mod, spec, (wq, wk, wv, tok) = G.build_reference(CASE, SEED)
    B, N, M, nb, C = CASE["B"], CASE["N"], CASE["M"], spec.num_bins, spec.C
    st = O.SamplerState(*(torch.from_numpy(a.copy()) for a in (wq, wk, wv, tok)))
    x = torch.from_numpy(synth.features(B, C, N, SEED + 10))
    nseed = SEED + 7
    xr = x.clone().requires_grad_(True)
    torch.manual_seed(nseed)
    (x_ds_r, idx_r), _ = mod(xr)
    torch.manual_seed(nseed)
    noise = O.draw_noise(B * nb, N)
    x_ds_o, idx_o = O.sampler_forward(spec, st, x, noise)
    tr = st.trace
    same(idx_o, idx_r, "idx")
    same(x_ds_o, x_ds_r, "x_ds")
    same(tr["score"], mod.attention_point_score, "score")
    same(tr["counts"], mod.k_point_to_choose, "counts")
    same(tr["w_pre"], mod.bin_weights_beforerelu, "bin weights")
    same(tr["upper"], mod.bin_boundaries[0], "upper")
    same(tr["lower"], mod.bin_boundaries[1], "lower")
    member = tr["member"]
    assert bool((member.sum(-1) == 1).all())
    g_np = synth.normal((B, C, M), SEED + 99)
    x_ds_r.backward(torch.from_numpy(g_np))
    xd, dx = x_ds_r.detach().double(), xr.grad.double()
    assert int(idx_r.max()) < 32768
    out = dict(
        meta=np.array([B, C, N, M, nb, spec.K, 1, SEED], dtype=np.int64), noise_seed=np.array(nseed, dtype=np.int64),
        noise=noise.numpy(),
        torch_version=np.array(torch.__version__),
        idx=idx_r[:, 0].numpy().astype(np.int16), counts=tr["counts"].numpy(), cap=tr["cap"].numpy(),
        w_pre=tr["w_pre"].detach().numpy(), upper=tr["upper"].numpy(), lower=tr["lower"].numpy(),
        quantiles=tr["quantiles"].numpy(), bin_id=member.squeeze(1).float().argmax(-1).to(torch.int8).numpy(),
        score_cloud_sums=np.stack([tr["score"].double().sum((1, 2)).numpy(), tr["score"].double().square().sum((1, 2)).numpy()], 1),
        score_first=tr["score"][:2].numpy(),
        x_ds_cloud_sums=np.stack([xd.sum((1, 2)).numpy(), xd.square().sum((1, 2)).numpy()], 1),
        dx_cloud_sums=np.stack([dx.sum((1, 2)).numpy(), dx.square().sum((1, 2)).numpy()], 1),
        dwq=mod.q_conv.weight.grad.numpy(), dwk=mod.k_conv.weight.grad.numpy(), dwv=mod.v_conv.weight.grad.numpy(),
        dtokens=mod.bin_tokens.grad.numpy(),
    )
    path = os.path.join(HERE, NAME + ".npz")
    np.savez_compressed(path, **out)
    print(f"{NAME}: ok, {os.path.getsize(path)/1024:.0f} KiB; counts[0] = {out['counts'][0].tolist()}")


if __name__ == "__main__":
    which = set(sys.argv[1:])
    if not which or "headline" in which:
        main()
    if not which or "stress" in which:
        main(*STRESS)

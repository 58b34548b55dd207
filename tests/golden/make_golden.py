#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED reference
(stevenczwu/SAMBLE, mounted read-only at /root/reference) on CPU in the build container.

Run from the repo root:   python tests/golden/make_golden.py

For every case it
  1. builds the reference `DownSampleToken` from the reference's own yaml configs,
  2. loads deterministic weights / inputs from samble_amd.synth,
  3. runs forward (+ backward) with `torch.manual_seed(s)` right before the call, so
     the only generator consumer is torch.multinomial (reference utils/ops.py:595),
  4. re-draws the same Exp(1) noise and asserts the identity
        multinomial(p, M) == topk(p / noise, M)          
  5. runs oracle/torch_oracle.py on the same inputs and asserts it is BIT-IDENTICAL to
     the reference for every recorded tensor,
  6. writes inputs' seeds + noise + outputs to tests/golden/<case>.npz.

Only data is written: no reference source or bytecode leaves /root/reference.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("SAMBLE_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch
import yaml

from samble_amd import synth
from oracle import torch_oracle as O

from models import downsample as ref_ds  # noqa: E402  (the reference)
from utils import ops as ref_ops  # noqa: E402


class _Attr(dict):
    __getattr__ = dict.__getitem__


def _attr(d):
    if isinstance(d, dict):
        return _Attr({k: _attr(v) for k, v in d.items()})
    if isinstance(d, list):
        return [_attr(v) for v in d]
    return d


def _merge(a, b):
    for k, v in b.items():
        if k in a and isinstance(a[k], dict) and isinstance(v, dict):
            _merge(a[k], v)
        else:
            a[k] = v
    return a


def reference_config(which: str):
    base = yaml.safe_load(open(os.path.join(REF, "configs/default.yaml")))
    usr = yaml.safe_load(open(os.path.join(REF, f"configs/{which}.yaml")))
    return _attr(_merge(base, usr)).feature_learning_block.downsample


CASES = [
    # name, yaml, B, N, M, overrides, calls, store_big
    dict(name="cls_random_dyn", cfg="cls", B=2, N=256, M=128, calls=2, big=True),
    dict(name="seg_random_dyn", cfg="seg", B=2, N=256, M=128, calls=2, big=False),
    dict(name="cls_topk_dyn", cfg="cls", B=2, N=256, M=128, calls=1, big=True,
         sample_mode="topk"),
    dict(name="cls_uniform_dyn", cfg="cls", B=2, N=256, M=128, calls=1, big=False,
         sample_mode="uniform"),
    dict(name="cls_random_static", cfg="cls", B=2, N=256, M=128, calls=1, big=False,
         static=[0.5, 0.03, -0.23, -0.43, -0.63]),
    dict(name="cls_random_B3_N512", cfg="cls", B=3, N=512, M=200, calls=2, big=False),
    dict(name="cls_random_cfg1", cfg="cls", B=8, N=1024, M=512, calls=1, big=False),
    dict(name="cls_colsum_topk", cfg="cls", B=2, N=256, M=128, calls=1, big=False,
         sample_mode="topk", idx_mode="col_sum"),
    dict(name="cls_l2_random", cfg="cls", B=2, N=256, M=128, calls=1, big=True, asm="l2"),
    dict(name="cls_onetoken_relumean", cfg="cls", B=2, N=256, M=128, calls=1, big=False,
         token_mode="one_token", relu_mean_order="relu_mean"),
    dict(name="cls_mode1_random", cfg="cls", B=2, N=256, M=128, calls=1, big=False, boltzmann_T="mode_1"),
    dict(name="seg_mode2_rowstd", cfg="seg", B=2, N=256, M=128, calls=1, big=False, boltzmann_T="mode_2",
         idx_mode="sparse_row_std"),
    # round 3: l2 scoring with the dense score modes (models/downsample.py:154-189 + 315-320)
    dict(name="cls_l2_colsum_topk", cfg="cls", B=2, N=256, M=128, calls=1, big=False, asm="l2", sample_mode="topk",
         idx_mode="col_sum"),
    dict(name="cls_l2_rowstd_random", cfg="cls", B=2, N=256, M=128, calls=1, big=False, asm="l2", idx_mode="row_std"),
    # round 4: a 64-channel layer (q_in = q_out = ... = 64)
    dict(name="cls_c64_random", cfg="cls", B=2, N=256, M=128, calls=2, big=True, C=64),
    # ... and a 256-channel one (no 256-channel attention kernels: DownSampleToken._forward_wide)
    dict(name="cls_c256_random", cfg="cls", B=2, N=256, M=128, calls=1, big=True, C=256),
    # round 5: the narrow layer with l2 logits: -|q - k|^2 / sqrt(C) is quadratic in (q, k), the 1 / sqrt(C) cannot ride
    # on W_q alone the way it does for dot logits
    dict(name="cls_c64_l2_random", cfg="cls", B=2, N=256, M=128, calls=1, big=True, C=64, asm="l2"),
]


def build_reference(case, seed):
    cfg = reference_config(case["cfg"])
    layer = 0
    if "sample_mode" in case:
        cfg.bin.sample_mode[layer] = case["sample_mode"]
    if "idx_mode" in case:
        cfg.idx_mode[layer] = case["idx_mode"]
    if "asm" in case:
        cfg.asm[layer] = case["asm"]
    if "token_mode" in case:
        cfg.bin.token_mode[layer] = case["token_mode"]
    if "relu_mean_order" in case:
        cfg.bin.relu_mean_order[layer] = case["relu_mean_order"]
    if "boltzmann_T" in case:
        cfg.bin.boltzmann_T[layer] = case["boltzmann_T"]
    if "C" in case:  # a narrower layer (the reference constructor takes any q_in / q_out ...: models/downsample.py:33-56)
        for key in ("q_in", "q_out", "k_in", "k_out", "v_in", "v_out"):
            cfg[key][layer] = case["C"]
    if "static" in case:
        cfg.bin.dynamic_boundaries_enable = False
        cfg.bin.bin_boundaries[layer] = list(case["static"])
    mod = ref_ds.DownSampleToken(cfg, layer)
    mod.M = case["M"]
    nb = mod.num_bins
    C = cfg.q_in[layer]
    wq, wk, wv, tok = synth.sampler_weights(C, nb if cfg.bin.token_mode[layer] == "multi_token" else 1, seed)
    with torch.no_grad():
        mod.q_conv.weight.copy_(torch.from_numpy(wq))
        mod.k_conv.weight.copy_(torch.from_numpy(wk))
        mod.v_conv.weight.copy_(torch.from_numpy(wv))
        mod.bin_tokens.copy_(torch.from_numpy(tok))
    spec = O.SamplerSpec(
        M=case["M"], K=cfg.K, C=C, num_bins=nb, asm=cfg.asm[layer], idx_mode=cfg.idx_mode[layer],
        sample_mode=cfg.bin.sample_mode[layer], boltzmann_T=cfg.bin.boltzmann_T[layer],
        relu_mean_order=cfg.bin.relu_mean_order[layer], token_mode=cfg.bin.token_mode[layer],
        dynamic_boundaries=bool(cfg.bin.dynamic_boundaries_enable),
        momentum=cfg.bin.momentum_update_factor[layer],
        static_boundaries=list(case["static"]) if "static" in case else None,
    )
    return mod, spec, (wq, wk, wv, tok)


def same(a, b, what):
    a = a.detach() if isinstance(a, torch.Tensor) else a
    b = b.detach() if isinstance(b, torch.Tensor) else b
    if not torch.equal(a, b):
        raise AssertionError(f"oracle != reference for {what}: max|d|={(a.float()-b.float()).abs().max()}")


def run_case(case, case_id):
    seed = 1000 * case_id
    mod, spec, (wq, wk, wv, tok) = build_reference(case, seed)
    B, N, M, nb = case["B"], case["N"], case["M"], spec.num_bins
    C = spec.C
    st = O.SamplerState(*(torch.from_numpy(a.copy()) for a in (wq, wk, wv, tok)))
    out = dict(
        meta=np.array([B, C, N, M, nb, spec.K, case["calls"], seed], dtype=np.int64),
        torch_version=np.array(torch.__version__),
        sample_mode=np.array(spec.sample_mode), idx_mode=np.array(spec.idx_mode), asm=np.array(spec.asm),
        boltzmann_T=np.array(spec.boltzmann_T if isinstance(spec.boltzmann_T, str) else float(spec.boltzmann_T)),
        token_mode=np.array(spec.token_mode), relu_mean_order=np.array(spec.relu_mean_order),
        momentum=np.array(spec.momentum),
        dynamic=np.array(spec.dynamic_boundaries),
        static=np.array(case.get("static", []), dtype=np.float32),
    )
    for call in range(case["calls"]):
        x_np = synth.features(B, C, N, seed + 10 + call)
        x = torch.from_numpy(x_np)
        last = call == case["calls"] - 1
        nseed = seed + 7 + call
        # --- reference
        xr = x.clone().requires_grad_(last)
        torch.manual_seed(nseed)
        (x_ds_r, idx_r), (d0, d1) = mod(xr)
        assert d0 is None and d1 is None
        # --- the noise torch.multinomial consumed
        torch.manual_seed(nseed)
        noise = O.draw_noise(B * nb, N)
        # --- oracle, noise injected
        x_ds_o, idx_o = O.sampler_forward(spec, st, x, noise if spec.sample_mode != "topk" else None)
        tr = st.trace
        same(idx_o, idx_r, "idx")
        same(x_ds_o, x_ds_r, "x_ds")
        same(tr["score"], mod.attention_point_score, "score")
        same(tr["member"], mod.bin_points_mask, "bin mask")
        same(tr["counts"], mod.k_point_to_choose, "counts")
        same(tr["w_pre"], mod.bin_weights_beforerelu, "bin weights")
        same(tr["tok_logits"], mod.attention_bins_beforesoftmax, "token logits")
        same(tr["upper"], mod.bin_boundaries[0], "upper")
        same(tr["lower"], mod.bin_boundaries[1], "lower")
        _, ref_knn = ref_ops.knn(x.permute(0, 2, 1), x.permute(0, 2, 1), spec.K)
        same(tr["knn_idx"], ref_knn, "knn idx")
        member = tr["member"]
        assert bool((member.sum(-1) == 1).all()), "bins are one-hot"
        bin_id = member.squeeze(1).float().argmax(-1).to(torch.int8)
        p = f"c{call}_"
        out[p + "noise"] = noise.numpy()
        out[p + "idx"] = idx_r.numpy()
        out[p + "score"] = tr["score"].numpy()
        out[p + "z"] = tr["z"].numpy()
        out[p + "tok_logits"] = tr["tok_logits"].detach().numpy()
        out[p + "lse"] = tr["lse"].detach().numpy()
        out[p + "knn_sorted"] = np.sort(ref_knn.numpy(), axis=-1).astype(np.int16)
        out[p + "indeg"] = tr["indeg"].numpy().astype(np.int32)
        out[p + "upper"] = tr["upper"].numpy()
        out[p + "lower"] = tr["lower"].numpy()
        if tr["quantiles"] is not None:
            out[p + "quantiles"] = tr["quantiles"].numpy()
        out[p + "bin_id"] = bin_id.numpy()
        out[p + "cap"] = tr["cap"].numpy()
        out[p + "w_pre"] = tr["w_pre"].detach().numpy()
        out[p + "counts"] = tr["counts"].numpy()
        xd = x_ds_r.detach().numpy()
        out[p + "x_ds_sum"] = np.array([xd.astype(np.float64).sum(), (xd.astype(np.float64) ** 2).sum()])
        if case["big"] or (call == 0 and xd.nbytes <= 200_000):
            out[p + "x_ds"] = xd
        if last:
            g_np = synth.normal((B, C, M), seed + 99)
            x_ds_r.backward(torch.from_numpy(g_np))
            # oracle grads from a fresh copy of the pre-call boundary state
            out[p + "dx_sum"] = np.array([xr.grad.double().sum().item(), (xr.grad.double() ** 2).sum().item()])
            # per cloud as well (dx is separable per cloud): lets a test compare the clouds whose sampled indices
            # agree even where the full dx is not stored
            out[p + "dx_cloud_sums"] = np.stack([xr.grad.double().sum((1, 2)).numpy(),
                                                 (xr.grad.double() ** 2).sum((1, 2)).numpy()], axis=1)
            grads = dict(dx=xr.grad, dwq=mod.q_conv.weight.grad, dwk=mod.k_conv.weight.grad,
                         dwv=mod.v_conv.weight.grad, dtokens=mod.bin_tokens.grad)
            leaves = [t.detach().clone().requires_grad_(True) for t in (x, st.wq, st.wk, st.wv, st.tokens)]
            q, k, v = O.project_qkv(leaves[0], leaves[4], leaves[1], leaves[2], leaves[3])
            A, _, _ = O.attention_map(q, k, N, spec.asm)
            O.gather_attend(A, v, idx_r).backward(torch.from_numpy(g_np))
            for name, leaf in zip(("dx", "dwq", "dwk", "dwv", "dtokens"), leaves):
                same(leaf.grad, grads[name], name)
            for name, gten in grads.items():
                if name == "dx" and not case["big"]:
                    continue
                out[p + name] = gten.numpy()
    # multinomial identity on a plain tensor as well (independent of the module)
    torch.manual_seed(123)
    pr = torch.rand(4, 64) + 1e-3
    torch.manual_seed(5)
    a = torch.multinomial(pr, 40)
    torch.manual_seed(5)
    nz = O.draw_noise(4, 64)
    assert torch.equal(a, torch.topk(pr / nz, 40, dim=1)[1]), "multinomial identity broke"
    path = os.path.join(HERE, case["name"] + ".npz")
    np.savez_compressed(path, **out)
    print(f"{case['name']}: ok, {os.path.getsize(path)/1024:.0f} KiB")


def main():
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    for i, case in enumerate(CASES, start=1):
        if only and case["name"] not in only:
            continue
        run_case(case, i)


if __name__ == "__main__":
    main()

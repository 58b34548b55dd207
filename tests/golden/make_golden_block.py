#!/usr/bin/env python3
"""Golden fixture for the classification FeatureLearningBlock call protocol (BASELINE.json
configs[1] at a small size): the UNMODIFIED reference block (models/cls_model.py:10-145, cls.yaml)
on CPU with deterministic parameters, forward AND backward under a fixed upstream gradient (the gradients of a
spread of parameters are stored: first EdgeConv, both samplers, an attention layer behind each sampler, the last
projection).  Run from the repo root:
    python tests/golden/make_golden_block.py"""
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

from samble_amd import synth
from oracle import torch_oracle as O
from tests.golden.make_golden_layers import block_config

from models import cls_model as ref_cls  # noqa: E402


from tests.util import fill_parameters  # noqa: E402  (same deterministic fill as the GPU test)


GRAD_KEYS_CLS = ("embedding_list.0.conv1.0.weight", "embedding_list.1.conv2.0.weight",
                 "downsample_list.0.bin_tokens", "downsample_list.0.q_conv.weight", "downsample_list.0.v_conv.weight",
                 "downsample_list.1.bin_tokens", "downsample_list.1.k_conv.weight",
                 "feature_learning_layer_list.0.ff.0.weight", "feature_learning_layer_list.1.q_conv.weight",
                 "feature_learning_layer_list.1.bn1.weight", "feature_learning_layer_list.2.v_conv.weight",
                 "conv_list.0.weight", "conv_list.2.weight")
GRAD_KEYS_SEG = ("embedding_list.0.conv1.0.weight", "downsample_list.0.bin_tokens", "downsample_list.0.k_conv.weight",
                 "downsample_list.1.bin_tokens", "downsample_list.1.q_conv.weight",
                 "feature_learning_layer_list.1.q_conv.weight", "feature_learning_layer_list.2.ff.2.weight",
                 "feature_learning_layer_list.3.k_conv.weight", "feature_learning_layer_list.4.bn2.bias",
                 # (layers 2 and 3's bn2.bias are left out: the BatchNorm of the upsampling layer behind them removes
                 # a constant, their true gradient is zero and what the reference returns is rounding noise)
                 "feature_learning_layer_list.4.ff.2.weight", "upsample_list.0.conv.0.weight",
                 "upsample_list.1.res_conv.0.weight", "upsample_list.1.res_conv.1.weight")
ROW_STRIDE = 8   # wide matrices are stored every 8th output row (the fixture stays small)


def thin(a: np.ndarray) -> np.ndarray:
    return a[::ROW_STRIDE] if (a.ndim >= 2 and a.shape[0] >= 512) else a


def grads_of(blk, keys):
    params = dict(blk.named_parameters())
    return {"grad/" + k: thin(params[k].grad.detach().numpy()) for k in keys}


def reference_block(kind, seed, B, N, M, threads=8, mkldnn=True, exact_cdist=False, forced_idx=None):
    """The unmodified reference block, forward + backward, under one of two equally valid fp32 evaluations of the same
    code (8 threads with oneDNN, or 1 thread without: other summation orders inside the same ATen ops).
    exact_cdist: torch.cdist in its compute_mode "donot_use_mm_for_euclid_dist" for the duration of the call -- the
    reference source is untouched, ATen evaluates sqrt(sum((a - b)^2)) instead of sqrt(|a|^2 + |b|^2 - 2 a.b)."""
    from models import seg_model as ref_seg
    torch.set_num_threads(threads)
    torch.backends.mkldnn.enabled = mkldnn
    aten_cdist = torch.cdist
    if exact_cdist:
        torch.cdist = lambda a, b, *args, **kw: aten_cdist(a, b, compute_mode="donot_use_mm_for_euclid_dist")
    # forced_idx (one (B,1,M) tensor per sampler, in call order): the sampler's index generator hands back THESE indices
    # -- the name `models.downsample` imported is rebound for the duration of the call, the reference source is untouched --
    # so that a second evaluation can be compared on the first one's point sets
    from models import downsample as ref_ds_mod
    own_generator = ref_ds_mod.generating_downsampled_index
    if forced_idx is not None:
        queue = [t.clone() for t in forced_idx]
        ref_ds_mod.generating_downsampled_index = lambda *a, **k: queue.pop(0)
    try:
        return _reference_block(kind, seed, B, N, M, ref_seg)
    finally:
        ref_ds_mod.generating_downsampled_index = own_generator
        torch.cdist = aten_cdist
        torch.set_num_threads(8)
        torch.backends.mkldnn.enabled = True


def _reference_block(kind, seed, B, N, M, ref_seg):
    cfg = block_config(kind)
    cfg.downsample.M = list(M)
    blk = (ref_cls if kind == "cls" else ref_seg).FeatureLearningBlock(cfg)
    fill_parameters(blk, seed)
    blk.train()
    xyz = torch.from_numpy(synth.xyz_clouds(B, N, seed + 500))
    # the pooled heads' arg-max points (`conv(x).max(dim=-1)`, cls_model.py:113,136,144: autograd routes the gradient to
    # that one point): recorded by forward hooks, so that a test can tell a near-tie resolved the other way from an error
    blk.head_args = []
    hooks = [m.register_forward_hook(lambda mod, inp, outp: blk.head_args.append(outp.detach().max(dim=-1)[1]))
             for m in getattr(blk, "conv_list", [])]
    torch.manual_seed(seed)
    out = blk(xyz)
    for h in hooks:
        h.remove()
    feat = out[0] if kind == "cls" else out
    feat.backward(torch.from_numpy(synth.normal(tuple(feat.shape), seed + 900)))   # the test's upstream gradient
    return cfg, blk, feat


def self_noise(blk, other, keys):
    """max |g - g'| / max |g| per stored gradient between the two evaluations: what "the reference's gradient" means to
    no better than this.  Where a hidden unit sits within rounding of a LeakyReLU kink, or two edges tie for a pooled
    maximum, the gradient is a coin toss between ANY two fp32 evaluations (seed 9100: 2e-3 on
    feature_learning_layer_list.0.ff.0.weight and 3e-4 upstream of it; some seeds flip a sampled index): the fixture's
    seed is one where the reference agrees with itself to 5e-5 everywhere and samples the same indices."""
    a, b = dict(blk.named_parameters()), dict(other.named_parameters())
    return np.array([float((a[k].grad - b[k].grad).abs().max() / a[k].grad.abs().max()) for k in keys], dtype=np.float64)


def make(kind, seed, keys, B=2, N=256, M=(128, 64), name=None, pick=True):
    """pick=True (the small fixtures): the seed must be one where the reference agrees with itself (asserted).
    pick=False (round 6, `block_cls_mid`): the seed is fixed BEFORE looking and nothing is rejected -- the reference's
    self-noise per gradient and whether its two evaluations sample the same points are recorded, and the test's
    tolerances are written in terms of them."""
    M = list(M)
    name = name or f"block_{kind}_small"
    cfg, blk, feat = reference_block(kind, seed, B, N, M)
    _, twin, feat2 = reference_block(kind, seed, B, N, M, threads=1, mkldnn=False)
    twin_same_idx = [bool(torch.equal(a.idx, b.idx)) for a, b in zip(blk.downsample_list, twin.downsample_list)]
    twin_clouds_same = [int((a.idx[:, 0] == b.idx[:, 0]).all(1).sum()) for a, b in zip(blk.downsample_list, twin.downsample_list)]
    if not pick and not all(twin_same_idx):
        # the reference's second evaluation samples other points: its gradients are those of another computation.  The
        # noise floor that means something for a comparison THROUGH the first evaluation's indices is the second
        # evaluation on those same indices
        _, twin, feat2 = reference_block(kind, seed, B, N, M, threads=1, mkldnn=False,
                                         forced_idx=[l.idx for l in blk.downsample_list])
    noise_floor = self_noise(blk, twin, keys)
    if pick:
        assert all(twin_same_idx), "the reference's own two evaluations sample different points: choose another seed"
        assert noise_floor.max() < 5e-5, ("choose another seed", dict(zip(keys, noise_floor)))
    nb = cfg.downsample.bin.num_bins[0]
    torch.manual_seed(seed)
    noise0 = O.draw_noise(B * nb, N)
    noise1 = O.draw_noise(B * nb, M[0])
    out = dict(meta=np.array([B, N, M[0], M[1], nb, seed], dtype=np.int64), feat=feat.detach().numpy(),
               **grads_of(blk, keys), grad_keys=np.array(list(keys)), grad_self_noise=noise_floor,
               feat_self_noise=np.array(float((feat - feat2).abs().max())),
               noise0=noise0.numpy(), noise1=noise1.numpy(),
               idx0=blk.downsample_list[0].idx.numpy(), idx1=blk.downsample_list[1].idx.numpy(),
               score0=blk.downsample_list[0].attention_point_score.detach().numpy(),
               names=np.array([n for n, _ in blk.named_parameters()]), torch_version=np.array(torch.__version__),
               twin_same_idx=np.array(twin_same_idx), twin_clouds_same=np.array(twin_clouds_same),
               seed_picked=np.array(bool(pick)),
               **{f"head_arg{i}": a.numpy().astype(np.int16) for i, a in enumerate(getattr(blk, "head_args", []))})
    if kind == "seg":
        # The interpolation layers weigh neighbours by 1 / (d + 1e-8) (models/upsample.py:205-213) and every coarse point
        # IS a fine point: d = 0 there.  ATen's default cdist (|a|^2 + |b|^2 - 2 a.b) returns ~1e-3 of rounding noise
        # instead, so the reference blends ~1 % of the two other neighbours in at those points, an amount no other
        # evaluation of the same expression reproduces.  Second set of outputs with cdist in its exact mode (d = 0 at
        # coinciding points, as the HIP search computes it): what the reference computes when that noise is out.
        _, blk_x, feat_x = reference_block(kind, seed, B, N, M, exact_cdist=True,
                                           forced_idx=None if pick else [l.idx for l in blk.downsample_list])
        out.update({k.replace("grad/", "exact/grad/"): v for k, v in grads_of(blk_x, keys).items()})
        out.update({"exact/feat": feat_x.detach().numpy(), "exact/idx0": blk_x.downsample_list[0].idx.numpy(),
                    "exact/idx1": blk_x.downsample_list[1].idx.numpy()})
        print("   exact-cdist run: indices equal to the default run:",
              [bool(torch.equal(a.idx, b.idx)) for a, b in zip(blk.downsample_list, blk_x.downsample_list)],
              "feat max|diff| %.3e" % float((feat - feat_x).abs().max()))
    if not pick:   # the mid-size fixture keeps the score of two clouds only (the full tensor is of no use to its test)
        out["score0"] = out["score0"][:2]
        if kind == "seg":   # per-point features of ONE cloud in full, of every cloud as float64 (sum, sum of squares)
            for key in ("feat", "exact/feat"):
                full = torch.from_numpy(out[key]).double()
                out[key + "_cloud_sums"] = torch.stack([full.sum((1, 2)), full.square().sum((1, 2))], 1).numpy()
                out[key] = out[key][:1]
        if kind == "cls":
            # ATen's default cdist (|a|^2 + |b|^2 - 2 a.b, ~1e-3 of rounding noise on 64- / 128-channel features) decides
            # near-ties of the K-th neighbour by that noise; at 4 x 1024 rows per search a few neighbour sets differ from
            # the exact distance order, which the HIP search reproduces.  Second run of the same unmodified reference
            # block with cdist in its exact mode, on the first run's sampled indices: the strict comparison
            _, blk_x, feat_x = reference_block(kind, seed, B, N, M, exact_cdist=True, forced_idx=[l.idx for l in blk.downsample_list])
            out.update({k.replace("grad/", "exact/grad/"): v for k, v in grads_of(blk_x, keys).items()})
            out["exact/feat"] = feat_x.detach().numpy()
            out.update({f"exact/head_arg{i}": a.numpy().astype(np.int16) for i, a in enumerate(blk_x.head_args)})
            # ... and the exact-cdist run left to sample for ITSELF: what the reference selects when its neighbour sets are
            # those of the exact distance order (the default run's selection follows its cdist's rounding noise)
            _, blk_o, _ = reference_block(kind, seed, B, N, M, exact_cdist=True)
            out["exact_own/idx0"] = blk_o.downsample_list[0].idx.numpy().astype(np.int16)
            out["exact_own/idx1"] = blk_o.downsample_list[1].idx.numpy().astype(np.int16)
            print("   exact-cdist run sampling for itself: clouds identical to the default run per layer:",
                  [int((a.idx[:, 0] == b.idx[:, 0]).all(1).sum()) for a, b in zip(blk.downsample_list, blk_o.downsample_list)])
            print("   exact-cdist run on the same indices: feat max|diff| to the default run %.3e; gradient difference per key:"
                  % float((feat - feat_x).abs().max()),
                  {k: "%.1e" % v for k, v in zip(keys, self_noise(blk, blk_x, keys))})
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: ok,", os.path.getsize(path) // 1024, "KiB; params", sum(p.numel() for p in blk.parameters()),
          "; gradient self-noise max %.1e" % noise_floor.max(), "; twin samples the same points:", twin_same_idx,
          "clouds identical per layer:", twin_clouds_same, "of", B)


if __name__ == "__main__":
    which = set(sys.argv[1:])
    if not which or "small" in which:
        make("cls", 9114, GRAD_KEYS_CLS)   # (9100, the seed of rounds 2-4, sits on a LeakyReLU kink: see self_noise)
        make("seg", 9300, GRAD_KEYS_SEG)
    if not which or "mid" in which:
        # round 6 (verdict r5: "block fixtures are tiny and seed-picked"): four clouds of 1024 points through
        # 1024 -> 512 -> 256, the seed written down before the first run, nothing rejected
        make("cls", 9500, GRAD_KEYS_CLS, B=4, N=1024, M=(512, 256), name="block_cls_mid", pick=False)
    if "full" in which:
        # BASELINE configs[1]'s own geometry (2048 -> 1024 -> 512) at a quarter of its batch: eight clouds, unpicked seed
        make("cls", 9700, GRAD_KEYS_CLS, B=8, N=2048, M=(1024, 512), name="block_cls_full", pick=False)
    if "segfull" in which:
        # BASELINE configs[2]'s own geometry (2048 -> 1024 -> 512 -> 1024 -> 2048) on eight clouds, unpicked seed
        make("seg", 9800, GRAD_KEYS_SEG, B=8, N=2048, M=(1024, 512), name="block_seg_full", pick=False)
    if not which or "segmid" in which:
        make("seg", 9600, GRAD_KEYS_SEG, B=4, N=1024, M=(512, 256), name="block_seg_mid", pick=False)

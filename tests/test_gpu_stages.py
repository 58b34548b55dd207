"""Stage-level parity of every HIP kernel (through the C ABI) against the CPU oracle / fp64 torch.

Integer outputs are compared bit-exactly given the oracle's stage inputs; floating point within the
tolerance written next to each assert."""
import math

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from samble_amd import synth
from tests.util import Golden, golden_names, set_agreement

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ops():
    from samble_amd import ops as o
    return o


# ---------------------------------------------------------------------------------------------
# kNN
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,C,N,K", [(2, 128, 256, 32), (3, 128, 1000, 32), (2, 64, 512, 16), (2, 128, 2048, 32)])
def test_knn_feature_space(B, C, N, K):
    x = torch.from_numpy(synth.features(B, C, N, 5 + N))
    idx = ops().stage_knn(x.to(DEV), x.to(DEV), K).cpu()
    pts = x.permute(0, 2, 1)
    ref_d, ref_i = O.knn(pts, pts, K)
    assert idx.dtype == torch.int32 and idx.shape == (B, N, K)
    assert bool((idx[:, :, 0] == torch.arange(N)).all()), "nearest neighbour of a point is itself"
    agree = set_agreement(idx, ref_i)
    # fp32 rounding differs between the MFMA Gram and ATen's cdist: near-ties may flip
    assert agree >= 0.9995, agree


def test_knn_fused_and_two_kernel_paths_agree():
    """A/B of the fused Gram+top-K kernel against the round-1 path that writes the key matrix."""
    B, C, N, K = 2, 128, 1024, 32
    x = torch.from_numpy(synth.features(B, C, N, 77)).to(DEV)
    fused_i, fused_d = ops().stage_knn(x, x, K, want_dist=True)
    plain_i, plain_d = ops().stage_knn(x, x, K, want_dist=True, variant=ops().KNN_TWO_KERNEL)
    assert set_agreement(fused_i.cpu(), plain_i.cpu()) >= 0.9998
    pts = x.cpu().permute(0, 2, 1)
    ref_d, ref_i = O.knn(pts, pts, K)
    assert set_agreement(fused_i.cpu(), ref_i) >= 0.9995
    torch.testing.assert_close(fused_d.cpu()[:, :, 1:], -ref_d[:, :, 1:], rtol=1e-3, atol=1e-3)
    # nearest first: distances ascending along K
    assert bool((fused_d[:, :, 1:] >= fused_d[:, :, :-1] - 1e-6).all())


@pytest.mark.parametrize("variant", ["default", "fp32_mfma", "two_kernel"])
def test_knn_offset_and_anisotropic_clouds(variant):
    """A cloud far from the origin (per-cloud offset of 50 sigma) with channel scales over two decades: the
    reference centres on the cloud mean before cdist (utils/ops.py:23-25); a Gram-form kernel that does not
    loses the neighbour ordering to cancellation.  Truth = exact fp64 distances."""
    B, C, N, K = 2, 128, 1024, 32
    x = torch.from_numpy(synth.features(B, C, N, 4711)).double()
    scale = torch.from_numpy(10.0 ** (2.0 * synth.uniform((1, C, 1), 4712).astype(np.float64) - 1.0))  # 0.1 .. 10
    off = torch.from_numpy(synth.normal((B, C, 1), 4713)).double() * 50.0
    xs = ((x + off) * scale).float()
    pts = xs.double().permute(0, 2, 1)
    d = ((pts[:, :, None, :] - pts[:, None, :, :]) ** 2).sum(-1)
    want = d.topk(K, dim=-1, largest=False)[1]
    o_ = ops()
    v = {"default": 0, "fp32_mfma": o_.KNN_FP32_MFMA, "two_kernel": o_.KNN_TWO_KERNEL}[variant]
    got, dist = o_.stage_knn(xs.to(DEV), xs.to(DEV), K, want_dist=True, variant=v)
    assert bool((got[:, :, 0].cpu() == torch.arange(N)).all()), "nearest neighbour of a point is itself"
    assert set_agreement(got.cpu(), want) >= 0.9995
    # the oracle (ATen cdist on the centred, scaled points) under the same test, for calibration of the bar
    _, ref_i = O.knn(xs.permute(0, 2, 1), xs.permute(0, 2, 1), K)
    assert set_agreement(ref_i, want) >= 0.999
    # reference-normalised distances of the winners
    ref_d, _ = O.knn(xs.permute(0, 2, 1), xs.permute(0, 2, 1), K)
    torch.testing.assert_close(dist.cpu()[:, :, 1:], -ref_d[:, :, 1:], rtol=2e-3, atol=2e-3)


def test_knn_degenerate_clouds():
    """All points identical / two clusters of duplicates: every distance ties.  The seed bound of the split-bf16
    kernel is useless here (everything passes): the result must still be a valid neighbour list (distinct
    indices, ties by ascending index) and the kernel must not overflow its candidate rings."""
    B, C, N, K = 2, 128, 512, 32
    x = torch.zeros(B, C, N)
    x[1, :, N // 2:] = 1.0  # cloud 1: two clusters of 256 duplicates
    got = ops().stage_knn(x.to(DEV), x.to(DEV), K).cpu()
    assert int(got.min()) >= 0 and int(got.max()) < N
    assert all(len(set(r.tolist())) == K for r in got.reshape(-1, K))
    assert torch.equal(got[0], torch.arange(K, dtype=torch.int32).expand(N, K)), "all ties: ascending index"
    lo, hi = got[1, : N // 2], got[1, N // 2:]
    assert torch.equal(lo, torch.arange(K, dtype=torch.int32).expand(N // 2, K))
    assert torch.equal(hi, (N // 2 + torch.arange(K, dtype=torch.int32)).expand(N // 2, K))


def test_knn_duplicate_burst_in_one_tile():
    """62 copies of one point: 30 scattered over the first eight 32-key tiles, 32 filling the ninth.  For a query among
    them every copy ties at distance 0, so its candidate ring holds 30 entries when one tile delivers 32 more: the
    ring's overflow path has to prune a row that holds FEWER than K entries (round 3: that case left the half-lane
    counters stale and an output slot unwritten).  Ties go by ascending index."""
    B, C, N, K = 2, 128, 512, 32
    x = torch.from_numpy(synth.features(B, C, N, 4242)).clone()
    g = torch.Generator().manual_seed(7)
    scattered = torch.randperm(256, generator=g)[:30].sort()[0]
    dup = torch.cat([scattered, torch.arange(256, 288)])
    x[:, :, dup] = x[:, :, dup[:1]]
    got = ops().stage_knn(x.to(DEV), x.to(DEV), K).cpu()
    assert int(got.min()) >= 0 and int(got.max()) < N
    assert all(len(set(r.tolist())) == K for r in got.reshape(-1, K))
    want = dup[:K].to(torch.int32)
    for b in range(B):
        assert torch.equal(got[b, dup], want.expand(len(dup), K)), "copies of a point: the K lowest indices, ascending"
    # the other rows: which of the tied copies enter a list is arbitrary in the fp64 reference, the distances are not
    p = x.double().permute(0, 2, 1)
    d = ((p[:, :, None, :] - p[:, None, :, :]) ** 2).sum(-1)
    ref_d = d.topk(K, dim=-1, largest=False)[0]
    got_d = torch.gather(d, 2, got.long()).sort(-1)[0]
    assert float(((got_d - ref_d).abs() > 1e-3).double().mean()) <= 5e-4


def test_knn_ragged_sizes_fused():
    B, C, K = 2, 128, 32
    for Nq, Nk in ((200, 330), (129, 97), (1000, 1000)):
        a = torch.from_numpy(synth.features(B, C, Nq, Nq)).to(DEV)
        bset = torch.from_numpy(synth.features(B, C, Nk, Nk + 1)).to(DEV)
        idx = ops().stage_knn(a, bset, K).cpu()
        assert int(idx.min()) >= 0 and int(idx.max()) < Nk
        _, ref_i = O.knn(a.cpu().permute(0, 2, 1), bset.cpu().permute(0, 2, 1), K)
        assert set_agreement(idx, ref_i) >= 0.999, (Nq, Nk)
        assert all(len(set(r.tolist())) == K for r in idx.reshape(-1, K)[::17])


@pytest.mark.parametrize("C,Nq,Nk,K", [(3, 2048, 2048, 32), (3, 512, 512, 32), (3, 300, 64, 32), (6, 1000, 700, 16),
                                        (32, 640, 640, 32), (128, 512, 512, 32), (128, 96, 64, 32), (64, 256, 1024, 16)])
def test_knn_narrow_sets_and_short_key_sets_on_the_matrix_core_kernel(C, Nq, Nk, K):
    """C < 64 (xyz: C = 3) runs on the fp16 matrix-core kernel with zero channels behind the real ones
    (csrc/knn.hip knn_duo_channels), and key sets of fewer than 64 K points take 2 or 4 seed candidates per tile
    (csrc/knn_duo.hip FINE): neighbour sets against the oracle, distances against the reference-normalised ones, no
    duplicate and no out-of-range index, rows in ascending distance."""
    B = 2
    gen = (lambda n, s: synth.xyz_clouds(B, n, s)) if C == 3 else (lambda n, s: synth.features(B, C, n, s))
    a = torch.from_numpy(gen(Nq, 400 + Nq + C))
    b = a if Nq == Nk else torch.from_numpy(gen(Nk, 900 + Nk + C))
    idx, dist = ops().stage_knn(a.to(DEV), b.to(DEV), K, want_dist=True)
    idx, dist = idx.cpu(), dist.cpu()
    ref_d, ref_i = O.knn(a.permute(0, 2, 1), b.permute(0, 2, 1), K)
    assert int(idx.min()) >= 0 and int(idx.max()) < Nk
    assert all(len(set(r.tolist())) == K for r in idx.reshape(-1, K)[::7])
    assert set_agreement(idx, ref_i) >= 0.999, set_agreement(idx, ref_i)
    # squared distances: the kernel forms |a|^2 + |b|^2 - 2 a.b from 22-bit operands, so the error is absolute in d^2
    # (a square root would blow it up next to d = 0, the self match)
    d2, ref2 = dist.double() ** 2, ref_d.double() ** 2
    top = float(ref2.max())
    assert bool((d2[:, :, 1:] >= d2[:, :, :-1] - 1e-5 * top).all())
    torch.testing.assert_close(d2, ref2, rtol=1e-3, atol=1e-5 * top)


def test_knn_xyz_cross_set_with_distance():
    B, Nq, Nk, K = 2, 700, 300, 3
    a = torch.from_numpy(synth.xyz_clouds(B, Nq, 11))
    b = torch.from_numpy(synth.xyz_clouds(B, Nk, 12))
    idx, dist = ops().stage_knn(a.to(DEV), b.to(DEV), K, want_dist=True)
    ref_d, ref_i = O.knn(a.permute(0, 2, 1), b.permute(0, 2, 1), K)
    assert set_agreement(idx.cpu(), ref_i) >= 0.999
    # positive distances of the reference-normalised points; C=3 path is exact (a-b)^2
    torch.testing.assert_close(dist.cpu(), -ref_d, rtol=2e-4, atol=2e-5)


def test_knn_reference_named_wrapper():
    B, C, N, K = 2, 128, 256, 32
    x = torch.from_numpy(synth.features(B, C, N, 3)).to(DEV)
    d, i = ops().knn(x.permute(0, 2, 1), x.permute(0, 2, 1), K)
    rd, ri = O.knn(x.cpu().permute(0, 2, 1), x.cpu().permute(0, 2, 1), K)
    assert i.dtype == torch.int64 and d.shape == (B, N, K)
    assert set_agreement(i.cpu(), ri) >= 0.9995
    # self distance is exactly 0 here but ~1e-2 in ATen's mm path: compare from the 2nd neighbour on
    torch.testing.assert_close(d.cpu()[:, :, 1:], rd[:, :, 1:], rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("C,Nq,Nk,K", [(128, 300, 300, 24), (128, 300, 300, 5), (3, 500, 200, 2), (64, 256, 256, 50),
                                       (128, 200, 200, 100), (3, 100, 6, 5), (128, 40, 40, 40)])
def test_knn_any_list_length(C, Nq, Nk, K):
    """The reference's knn takes any k (utils/ops.py:17-44); the kernels keep lists of 1, 3, 8, 16, 20, 32, 40 or 64.
    Any other k is the head of the next size's list (nearest first); k > 64, or a key set shorter than the next size,
    runs the reference's expression on the device.  Rounds 1-5 refused such k."""
    B = 2
    mk = synth.xyz_clouds if C == 3 else (lambda b, n, sd: synth.features(b, C, n, sd))
    a, b = torch.from_numpy(mk(B, Nq, 21)), torch.from_numpy(mk(B, Nk, 22 if Nk != Nq else 21))
    idx, dist = ops().stage_knn(a.to(DEV), b.to(DEV), K, want_dist=True)
    ref_d, ref_i = O.knn(a.permute(0, 2, 1), b.permute(0, 2, 1), K)
    assert idx.shape == (B, Nq, K) and idx.dtype == torch.int32 and dist.shape == (B, Nq, K)
    assert set_agreement(idx.cpu(), ref_i) >= 0.999
    top = float(ref_d.abs().max())
    assert bool((dist[:, :, 1:] >= dist[:, :, :-1] - 1e-5 * top).all()), "nearest first"
    torch.testing.assert_close(dist.cpu()[:, :, 1:], -ref_d[:, :, 1:], rtol=2e-3, atol=2e-3 * top)
    with pytest.raises(ValueError):
        ops().stage_knn(a.to(DEV), b.to(DEV), Nk + 1)


# ---------------------------------------------------------------------------------------------
# QKV projection
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,N,nt", [(2, 256, 6), (3, 1000, 4), (2, 2048, 6), (1, 100, 1)])
def test_projection_forward_backward(B, N, nt):
    C = 128
    x = torch.from_numpy(synth.features(B, C, N, 40 + N))
    tok = torch.from_numpy(synth.normal((C, nt), 41)) * 0.1
    w = torch.from_numpy(synth.normal((3 * C, C), 42)) * 0.09
    g = torch.from_numpy(synth.normal((B, N + nt, 3 * C), 43))
    xd, td, wd = (t.double().requires_grad_(True) for t in (x, tok, w))
    xt = torch.cat((xd, td.unsqueeze(0).expand(B, -1, -1)), dim=2)
    ref = torch.matmul(xt.transpose(1, 2), wd.t())
    ref.backward(g.double())
    got = ops().stage_proj_fwd(x.to(DEV), tok.to(DEV), w.to(DEV))
    torch.testing.assert_close(got.cpu().double(), ref.detach(), rtol=2e-5, atol=2e-5)
    dx, dw, dtok = ops().stage_proj_bwd(g.to(DEV), x.to(DEV), tok.to(DEV), w.to(DEV), True, True)
    for a, r, name in ((dx, xd.grad, "dx"), (dw, wd.grad, "dW"), (dtok, td.grad, "dtokens")):
        err = (a.cpu().double() - r).abs().max().item()
        assert err <= 3e-5 * r.abs().max().item() + 1e-6, (name, err)
    # run-to-run identical (fixed-order reductions)
    dx2, dw2, dtok2 = ops().stage_proj_bwd(g.to(DEV), x.to(DEV), tok.to(DEV), w.to(DEV), True, True)
    assert torch.equal(dw, dw2) and torch.equal(dx, dx2) and torch.equal(dtok, dtok2)


# ---------------------------------------------------------------------------------------------
# attention forward / backward
# ---------------------------------------------------------------------------------------------
def _qkv(B, N, nt, seed, D=128):
    q = torch.from_numpy(synth.normal((B, N, D), seed))
    k = torch.from_numpy(synth.normal((B, N + nt, D), seed + 1))
    v = torch.from_numpy(synth.normal((B, N + nt, D), seed + 2))
    return q, k, v


@pytest.mark.parametrize("B,N,nt", [(2, 256, 6), (1, 1000, 4), (2, 1024, 6), (1, 96, 1)])
def test_attn_fwd(B, N, nt):
    q, k, v = _qkv(B, N, nt, 100 + N)
    O_, lse, tok = ops().stage_attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), N, nt)
    s = (q.double() @ k.double().transpose(1, 2)) / math.sqrt(128)
    ref_o = torch.softmax(s, -1) @ v.double()
    # fp32 MFMA chains of length 128 / N: a few 1e-6 relative
    torch.testing.assert_close(O_.cpu().double(), ref_o, rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(lse.cpu().double(), torch.logsumexp(s, -1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(tok.cpu().double(), s[:, :, N:], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("B,N,nt", [(2, 256, 6), (1, 1000, 4)])
def test_dense_map_statistics(B, N, nt):
    """idx_mode col_sum / row_std: column sums and row std of the point-to-point block A[:, :N]."""
    q, k, v = _qkv(B, N, nt, 700 + N)
    q, k = q * 0.4, k * 0.4
    s = (q.double() @ k.double().transpose(1, 2)) / math.sqrt(128)
    A = torch.softmax(s, -1)[:, :, :N]
    O_, lse, tok, rstd = ops().stage_attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), N, nt, want_row_std=True)
    torch.testing.assert_close(rstd.cpu().double(), A.std(dim=-1), rtol=2e-4, atol=1e-9)
    col = ops().stage_attn_colsum(q.to(DEV), k.to(DEV), lse)
    torch.testing.assert_close(col.cpu().double(), A.sum(dim=-2), rtol=2e-5, atol=1e-9)
    score, z = ops().stage_stat_score(col)
    assert torch.equal(score, col)
    zr = (A.sum(-2) - A.sum(-2).mean(-1, keepdim=True)) / A.sum(-2).std(-1, unbiased=False, keepdim=True)
    torch.testing.assert_close(z.cpu().double(), zr, rtol=1e-3, atol=1e-4)


def test_attn_fwd_strided_views():
    """The module hands q/k/v as column slices of one (B, N+nt, 3D) projection."""
    B, N, nt, D = 2, 256, 6, 128
    qkv = torch.from_numpy(synth.normal((B, N + nt, 3 * D), 9)).to(DEV)
    q, k, v = qkv[:, :N, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    O_, lse, tok = ops().stage_attn_fwd(q, k, v, N, nt)
    s = (q.double() @ k.double().transpose(1, 2)) / math.sqrt(D)
    torch.testing.assert_close(O_.double(), torch.softmax(s, -1) @ v.double(), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("B,N,nt,M", [(2, 256, 6, 128), (1, 1000, 4, 333), (2, 1024, 6, 512)])
def test_attn_bwd(B, N, nt, M):
    D = 128
    q, k, v = _qkv(B, N, nt, 300 + N)
    g = torch.from_numpy(synth.normal((B, D, M), 7))
    idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b))[:M] for b in range(B)])
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    s = (qd @ kd.transpose(1, 2)) / math.sqrt(D)
    o = torch.softmax(s, -1) @ vd
    rows = torch.gather(o, 1, idx[..., None].expand(-1, -1, D))  # (B,M,D)
    rows.permute(0, 2, 1).backward(g.double())

    o_ = ops()
    qg, kg, vg = q.to(DEV), k.to(DEV), v.to(DEV)
    O_, lse, _ = o_.stage_attn_fwd(qg, kg, vg, N, nt)
    dq = torch.full((B, N, D), float("nan"), device=DEV)
    dk = torch.full((B, N + nt, D), float("nan"), device=DEV)
    dv = torch.full((B, N + nt, D), float("nan"), device=DEV)
    for split in (0, 1):  # fused 5-product backward, then the two-kernel (7-product) path
        dq.fill_(float("nan")); dk.fill_(float("nan")); dv.fill_(float("nan"))
        o_.stage_attn_bwd(qg, kg, vg, O_, lse, idx.to(DEV), g.to(DEV), N, nt, dq, dk, dv, variant=split)
        for got, ref, name in ((dq, qd.grad, "dq"), (dk, kd.grad, "dk"), (dv, vd.grad, "dv")):
            assert torch.isfinite(got).all(), (name, split)
            scale = ref.abs().max().item()
            err = (got.cpu().double() - ref).abs().max().item()
            assert err <= 3e-5 * scale + 1e-7, (name, split, err, scale)
    # the fused path is run-to-run identical (slabs are summed in a fixed order)
    a = dq.clone()
    o_.stage_attn_bwd(qg, kg, vg, O_, lse, idx.to(DEV), g.to(DEV), N, nt, dq, dk, dv)
    b2 = dq.clone()
    o_.stage_attn_bwd(qg, kg, vg, O_, lse, idx.to(DEV), g.to(DEV), N, nt, dq, dk, dv)
    assert torch.equal(b2, dq)


def test_split_products_are_fp32_equivalent():
    """The bf16 x 6 logits are as close to fp64 as the fp32-MFMA logits (same inputs, both kernels)."""
    B, N, nt = 2, 1024, 6
    q, k, _ = _qkv(B, N, nt, 4242)
    s = (q.double() @ k.double().transpose(1, 2)) / math.sqrt(128)
    o_ = ops()
    err = {}
    old = o_.MATRIX_MODE
    try:
        for mode in ("f32", "tri"):
            o_.MATRIX_MODE = mode
            smap, lse, _ = o_.stage_attn_stats(q.to(DEV), k.to(DEV), N, nt)
            d = smap[:, :, :N + nt].cpu().double() - s
            err[mode] = (d.pow(2).mean().sqrt().item(), d.abs().max().item())
    finally:
        o_.MATRIX_MODE = old
    assert err["tri"][0] <= 1.25 * err["f32"][0] and err["tri"][1] <= 2.0 * err["f32"][1], err


@pytest.mark.parametrize("sq,sk", [(1.0, 1.0), (1e-3, 1e-3), (1e3, 1e-2), (3e4, 3e-5)])
def test_logit_products_are_blind_to_operand_scale(sq, sk):
    """The logit products run on two fp16 planes under per-tile (K) and per-row (Q) power-of-two scales
    (csrc/tri_dev.h): whatever the operands' magnitude, a key row 50x the others in a tile, a tile of zeros, a zero query
    row -- the logits stay as close to fp64 as the fp32-MFMA kernel's, and nothing overflows fp16."""
    B, N, nt = 2, 512, 6
    g = torch.Generator().manual_seed(77)
    q = torch.randn(B, N, 128, generator=g) * sq
    k = torch.randn(B, N + nt, 128, generator=g) * sk
    k[:, 5] *= 50.0
    k[:, 64:96] = 0.0   # a whole key tile of zeros
    q[:, 3] = 0.0
    s = (q.double() @ k.double().transpose(1, 2)) / math.sqrt(128)
    o_ = ops()
    err = {}
    old = o_.MATRIX_MODE
    try:
        for mode in ("f32", "tri"):
            o_.MATRIX_MODE = mode
            smap, lse, _ = o_.stage_attn_stats(q.to(DEV), k.to(DEV), N, nt)
            got = smap[:, :, :N + nt].cpu().double()
            assert bool(torch.isfinite(got).all()) and bool(torch.isfinite(lse).all())
            d = got - s
            err[mode] = (d.pow(2).mean().sqrt().item(), d.abs().max().item())
            assert bool((got[:, :, 64:96] == 0).all()) and bool((got[:, 3] == 0).all())
    finally:
        o_.MATRIX_MODE = old
    assert err["tri"][0] <= 1.25 * err["f32"][0] and err["tri"][1] <= 2.0 * err["f32"][1], err


# ---------------------------------------------------------------------------------------------
# two-pass forward with the logit map in HBM (attn_stats / attn_rows / sparse_score_map / rows_bwd)
# ---------------------------------------------------------------------------------------------
@pytest.fixture(params=["tri", "f32"])
def matrix_mode(request):
    """Both matrix-instruction families behind the same stage functions (ops.MATRIX_MODE)."""
    o_ = ops()
    old = o_.MATRIX_MODE
    o_.MATRIX_MODE = request.param
    yield request.param
    o_.MATRIX_MODE = old


@pytest.mark.parametrize("B,N,nt,M", [(2, 256, 6, 128), (1, 1000, 4, 333), (2, 1024, 6, 512), (1, 96, 1, 50),
                                      (1, 1025, 6, 700)])
def test_two_pass_forward(B, N, nt, M, matrix_mode):
    D = 128
    q, k, v = _qkv(B, N, nt, 900 + N)
    idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b))[:M] for b in range(B)])
    s = (q.double() @ k.double().transpose(1, 2)) / math.sqrt(D)
    o_ = ops()
    smap, lse, tok = o_.stage_attn_stats(q.to(DEV), k.to(DEV), N, nt)
    ld = smap.shape[2]
    assert ld % 32 == 0 and ld >= N + nt
    # a logit is one fp32-accumulated MFMA chain of length 128: 1e-6 relative to |q||k|
    torch.testing.assert_close(smap[:, :, :N + nt].cpu().double(), s, rtol=1e-5, atol=2e-5)
    assert torch.isneginf(smap[:, :, N + nt:]).all()
    torch.testing.assert_close(lse.cpu().double(), torch.logsumexp(s, -1), rtol=1e-5, atol=1e-5)
    assert torch.equal(tok, smap[:, :, N:N + nt])
    x_ds = o_.stage_attn_rows(smap, lse, v.to(DEV), idx.to(DEV), N, nt)
    ref = torch.gather(torch.softmax(s, -1) @ v.double(), 1, idx[..., None].expand(-1, -1, D)).permute(0, 2, 1)
    torch.testing.assert_close(x_ds.cpu().double(), ref, rtol=2e-4, atol=2e-5)
    # the single-pass kernel computes the same logits with the same MFMA chain
    O1, lse1, tok1 = o_.stage_attn_fwd(q.to(DEV), k.to(DEV), v.to(DEV), N, nt)
    if matrix_mode == "f32":
        assert torch.equal(tok1, tok)
    else:  # split-bf16 products: other rounding, same accuracy
        torch.testing.assert_close(tok1, tok, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lse1, lse, rtol=0, atol=4e-6)
    torch.testing.assert_close(o_.stage_gather_rows(O1, idx.to(DEV)), x_ds, rtol=1e-5, atol=4e-6)


@pytest.mark.parametrize("B,N,nt,M", [(2, 256, 6, 128), (1, 1000, 4, 333)])
def test_two_pass_l2_forward_backward(B, N, nt, M, matrix_mode):
    """asm 'l2' (reference downsample.py:154-175): S = -|q - k|^2 / sqrt(D) through the same kernels."""
    D = 128
    q, k, v = _qkv(B, N, nt, 1300 + N)
    q, k = q * 0.3, k * 0.3
    g = torch.from_numpy(synth.normal((B, D, M), 17))
    idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b))[:M] for b in range(B)])
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    s = -((qd ** 2).sum(-1, keepdim=True) - 2 * qd @ kd.transpose(1, 2) + (kd ** 2).sum(-1).unsqueeze(1)) / math.sqrt(D)
    o = torch.softmax(s, -1) @ vd
    rows = torch.gather(o, 1, idx[..., None].expand(-1, -1, D))
    rows.permute(0, 2, 1).backward(g.double())
    o_ = ops()
    qg, kg, vg = q.to(DEV), k.to(DEV), v.to(DEV)
    smap, lse, tok = o_.stage_attn_stats(qg, kg, N, nt, "l2")
    torch.testing.assert_close(smap[:, :, :N + nt].cpu().double(), s.detach(), rtol=1e-5, atol=3e-5)
    torch.testing.assert_close(lse.cpu().double(), torch.logsumexp(s.detach(), -1), rtol=1e-5, atol=2e-5)
    assert torch.equal(tok, smap[:, :, N:N + nt])
    x_ds = o_.stage_attn_rows(smap, lse, vg, idx.to(DEV), N, nt)
    torch.testing.assert_close(x_ds.cpu().double(), rows.detach().permute(0, 2, 1), rtol=2e-4, atol=2e-5)
    dq = torch.full((B, N, D), float("nan"), device=DEV)
    dk = torch.full((B, N + nt, D), float("nan"), device=DEV)
    dv = torch.full((B, N + nt, D), float("nan"), device=DEV)
    o_.stage_attn_rows_bwd(qg, kg, vg, smap, lse, x_ds, idx.to(DEV), g.to(DEV), N, nt, dq, dk, dv, "l2")
    for got, ref, name in ((dq, qd.grad, "dq"), (dk, kd.grad, "dk"), (dv, vd.grad, "dv")):
        assert torch.isfinite(got).all(), name
        scale = ref.abs().max().item()
        err = (got.cpu().double() - ref).abs().max().item()
        assert err <= 5e-5 * scale + 1e-7, (name, err, scale)


def test_two_pass_strided_views(matrix_mode):
    B, N, nt, D, M = 2, 256, 6, 128, 100
    qkv = torch.from_numpy(synth.normal((B, N + nt, 3 * D), 19)).to(DEV)
    q, k, v = qkv[:, :N, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b))[:M] for b in range(B)]).to(DEV)
    smap, lse, tok = ops().stage_attn_stats(q, k, N, nt)
    x_ds = ops().stage_attn_rows(smap, lse, v, idx, N, nt)
    s = (q.double() @ k.double().transpose(1, 2)) / math.sqrt(D)
    ref = torch.gather(torch.softmax(s, -1) @ v.double(), 1, idx[..., None].expand(-1, -1, D)).permute(0, 2, 1)
    torch.testing.assert_close(x_ds.double(), ref, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("mode", ["sparse_col_sqr", "sparse_col_sum", "sparse_col_avg", "sparse_row_sum",
                                  "sparse_row_std"])
def test_sparse_score_from_map(mode):
    B, N, nt, K = 2, 512, 6, 32
    q, k, v = _qkv(B, N, nt, 55)
    q, k = q * 0.3, k * 0.3
    nn = torch.stack([torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b * N + i))[:K]
                                   for i in range(N)]) for b in range(B)]).int()
    s = (q.double() @ k.double().transpose(1, 2)) / math.sqrt(128)
    A = torch.softmax(s, -1)[:, :, :N]
    mask = torch.zeros(B, N, N, dtype=torch.float64).scatter_(2, nn.long(), 1.0)
    sparse = A * mask
    num = mask.sum(-2) + 1e-8
    ref = {"sparse_col_sum": sparse.sum(-2), "sparse_col_avg": sparse.sum(-2) / num,
           "sparse_col_sqr": sparse.sum(-2) / num / num, "sparse_row_sum": sparse.sum(-1),
           "sparse_row_std": torch.std(sparse.masked_select(mask != 0).view(B, N, K), dim=-1)}[mode]
    ref[torch.isnan(ref)] = 0
    smap, lse, _ = ops().stage_attn_stats(q.to(DEV), k.to(DEV), N, nt)
    score, z, indeg = ops().stage_sparse_score_map(smap, lse, nn.to(DEV), mode)
    assert torch.equal(indeg.cpu().long(), mask.sum(-2).long())
    torch.testing.assert_close(score.cpu().double(), ref, rtol=2e-5, atol=1e-9)
    again = ops().stage_sparse_score_map(smap, lse, nn.to(DEV), mode)[0]
    assert torch.equal(score, again)  # integer accumulation: order independent


@pytest.mark.parametrize("B,N,nt,M", [(2, 256, 6, 128), (1, 1000, 4, 333), (2, 1024, 6, 512), (1, 1025, 6, 77)])
def test_attn_rows_bwd(B, N, nt, M, matrix_mode):
    D = 128
    q, k, v = _qkv(B, N, nt, 300 + N)
    g = torch.from_numpy(synth.normal((B, D, M), 7))
    idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b))[:M] for b in range(B)])
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    s = (qd @ kd.transpose(1, 2)) / math.sqrt(D)
    o = torch.softmax(s, -1) @ vd
    rows = torch.gather(o, 1, idx[..., None].expand(-1, -1, D))
    rows.permute(0, 2, 1).backward(g.double())
    o_ = ops()
    qg, kg, vg = q.to(DEV), k.to(DEV), v.to(DEV)
    smap, lse, _ = o_.stage_attn_stats(qg, kg, N, nt)
    x_ds = o_.stage_attn_rows(smap, lse, vg, idx.to(DEV), N, nt)
    dq = torch.full((B, N, D), float("nan"), device=DEV)
    dk = torch.full((B, N + nt, D), float("nan"), device=DEV)
    dv = torch.full((B, N + nt, D), float("nan"), device=DEV)
    o_.stage_attn_rows_bwd(qg, kg, vg, smap, lse, x_ds, idx.to(DEV), g.to(DEV), N, nt, dq, dk, dv)
    for got, ref, name in ((dq, qd.grad, "dq"), (dk, kd.grad, "dk"), (dv, vd.grad, "dv")):
        assert torch.isfinite(got).all(), name
        scale = ref.abs().max().item()
        err = (got.cpu().double() - ref).abs().max().item()
        assert err <= 3e-5 * scale + 1e-7, (name, err, scale)
    a = (dq.clone(), dk.clone(), dv.clone())
    o_.stage_attn_rows_bwd(qg, kg, vg, smap, lse, x_ds, idx.to(DEV), g.to(DEV), N, nt, dq, dk, dv)
    assert torch.equal(a[0], dq) and torch.equal(a[1], dk) and torch.equal(a[2], dv)  # deterministic


# ---------------------------------------------------------------------------------------------
# sparse score
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["sparse_col_sqr", "sparse_col_sum", "sparse_col_avg", "sparse_row_sum",
                                  "sparse_row_std"])
def test_sparse_score_modes(mode):
    B, N, nt, K = 2, 512, 6, 32
    q, k, v = _qkv(B, N, nt, 55)
    q, k = q * 0.3, k * 0.3
    nn = torch.stack([torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b * N + i))[:K]
                                   for i in range(N)]) for b in range(B)]).int()
    s = (q.double() @ k.double().transpose(1, 2)) / math.sqrt(128)
    A = torch.softmax(s, -1)[:, :, :N]
    mask = torch.zeros(B, N, N, dtype=torch.float64).scatter_(2, nn.long(), 1.0)
    sparse = A * mask
    num = mask.sum(-2) + 1e-8
    ref = {"sparse_col_sum": sparse.sum(-2), "sparse_col_avg": sparse.sum(-2) / num,
           "sparse_col_sqr": sparse.sum(-2) / num / num, "sparse_row_sum": sparse.sum(-1),
           "sparse_row_std": torch.std(sparse.masked_select(mask != 0).view(B, N, K), dim=-1)}[mode]
    ref[torch.isnan(ref)] = 0
    lse = torch.logsumexp(s, -1).float()
    score, z, indeg = ops().stage_sparse_score(q.to(DEV), k.to(DEV), lse.to(DEV), nn.to(DEV), mode)
    assert torch.equal(indeg.cpu().long(), mask.sum(-2).long())
    torch.testing.assert_close(score.cpu().double(), ref, rtol=2e-5, atol=1e-9)
    zr = (ref - ref.mean(-1, keepdim=True)) / ref.std(-1, unbiased=False, keepdim=True)
    torch.testing.assert_close(z.cpu().double(), zr, rtol=1e-3, atol=1e-4)


def test_sparse_score_is_run_to_run_identical():
    B, N, nt, K = 2, 1024, 6, 32
    q, k, _ = _qkv(B, N, nt, 5)
    nn = torch.randint(0, N, (B, N, K), generator=torch.Generator().manual_seed(1)).int().to(DEV)
    s = (q @ k.transpose(1, 2)) / math.sqrt(128)
    lse = torch.logsumexp(s, -1).to(DEV)
    a = ops().stage_sparse_score(q.to(DEV), k.to(DEV), lse, nn, "sparse_col_sqr")[0]
    for _ in range(3):
        b = ops().stage_sparse_score(q.to(DEV), k.to(DEV), lse, nn, "sparse_col_sqr")[0]
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------
# bins / counts / selection, driven by the golden fixtures (integer-exact given oracle inputs)
# ---------------------------------------------------------------------------------------------
def test_zscore_matches_oracle_within_ulps():
    g = Golden("cls_random_dyn")
    z = ops().stage_zscore(g.t("score").reshape(g.B, g.N).to(DEV)).cpu()
    ref = g.t("z").reshape(g.B, g.N)
    # torch's mean is an ISA-dependent cascade sum; ours is the correctly rounded mean: <= 2 ulp of z scale
    torch.testing.assert_close(z, ref, rtol=0, atol=4e-7 * float(ref.abs().max()))


@pytest.mark.parametrize("n,nb", [(512, 6), (65536, 6), (65536, 4), (100003, 8), (131072, 2)])
def test_batch_quantiles_exact(n, nb):
    z = torch.from_numpy(synth.normal((n,), 900 + nb))
    z[::97] = z[5]  # ties
    got = ops().stage_batch_quantiles(z.to(DEV), nb).cpu()
    assert torch.equal(got, O.batch_quantiles(z, nb))


@pytest.mark.parametrize("kind", ["constant", "two_values", "outliers", "wide", "tiny_spread"])
def test_batch_quantiles_exact_on_awkward_distributions(kind):
    """Inputs that overflow or defeat the linear histogram of the fast path must still be exact
    (the kernel falls back to its radix select)."""
    n, nb = 65536, 6
    z = torch.from_numpy(synth.normal((n,), 77))
    if kind == "constant":
        z = torch.full((n,), 0.25)
    elif kind == "two_values":
        z = torch.where(z > 0, torch.tensor(1.5), torch.tensor(-0.5))
    elif kind == "outliers":
        z[:100] = 1e6
        z[100:200] = -1e6
    elif kind == "wide":
        z = z * 100.0
    elif kind == "tiny_spread":
        z = 1.0 + z * 1e-6
    got = ops().stage_batch_quantiles(z.to(DEV), nb).cpu()
    assert torch.equal(got, O.batch_quantiles(z, nb))


@pytest.mark.parametrize("name", golden_names())
def test_select_stages_exact_on_golden(name):
    g = Golden(name)
    o_ = ops()
    for call in range(g.calls):
        z = g.t("z", call).reshape(g.B, g.N).to(DEV)
        score = g.t("score", call).reshape(g.B, g.N).to(DEV)
        tok = g.t("tok_logits", call).reshape(g.B, g.N, -1).to(DEV)
        if g.dynamic:
            q = o_.stage_batch_quantiles(z, g.nb).cpu()
            assert torch.equal(q, g.t("quantiles", call)), "batch quantiles"
        upper, lower = g.t("upper", call).to(DEV), g.t("lower", call).to(DEV)
        member, cap, w_pre, w = o_.stage_bin_assign(z, tok, upper, lower, g.relu_mean_order == "relu_mean")
        bits = member.cpu().long()
        assert bool((bits > 0).all()) and bool(((bits & (bits - 1)) == 0).all()), "one bin per point"
        bin_id = torch.log2(bits.float()).round().to(torch.int8)
        assert torch.equal(bin_id, g.t("bin_id", call)), "bin ids"
        assert torch.equal(cap.cpu().long(), g.t("cap", call)), "bin populations"
        torch.testing.assert_close(w_pre.cpu(), g.t("w_pre", call), rtol=2e-6, atol=1e-7)
        # counts: exact given the reference's own weights
        w_ref = torch.relu(g.t("w_pre", call)).to(DEV)
        counts = o_.stage_alloc_counts(w_ref, cap, g.M)
        assert torch.equal(counts.cpu(), g.t("counts", call)), "counts"
        noise = None if g.sample_mode == "topk" else g.t("noise", call).to(DEV)
        idx = o_.stage_bin_select(score, z, member, counts, g.M, g.sample_mode, g.boltzmann_T, noise)
        assert torch.equal(idx.cpu(), g.t("idx", call).reshape(g.B, g.M)), "sampled indices"


def test_alloc_counts_exact_random_trials():
    rng = np.random.default_rng(3)
    for trial in range(40):
        B, nb, N, M = 32, (6 if trial % 2 == 0 else 4), 2048, 1024
        cuts = np.sort(rng.integers(0, N + 1, size=(B, nb - 1)), axis=1)
        cap = np.diff(np.concatenate([np.zeros((B, 1), int), cuts, np.full((B, 1), N)], 1), axis=1)
        w = np.maximum(rng.standard_normal((B, nb)).astype(np.float32) * (0.3 if trial % 3 else 2.0), 0)
        if trial % 5 == 0:
            w[:, rng.integers(0, nb)] = 0
        ref = O.allocate_counts(torch.from_numpy(w.copy()), torch.from_numpy(cap.astype(np.int64)), M)
        got = ops().stage_alloc_counts(torch.from_numpy(w).to(DEV), torch.from_numpy(cap.astype(np.int32)).to(DEV), M)
        assert torch.equal(got.cpu(), ref), trial


@pytest.mark.parametrize("mode", ["topk", "uniform"])
def test_bin_select_exact_vs_oracle_full_size(mode):
    """topk / uniform keys involve no transcendental: exact at the metric size."""
    B, N, nb, M = 8, 2048, 6, 1024
    # exact ties have no defined order in torch.sort / topk: build scores that are distinct by
    # construction (a permutation of N levels, spaced far above one fp32 ulp)
    levels = np.stack([np.random.default_rng(b).permutation(N) for b in range(B)]).astype(np.float64)
    score = torch.from_numpy(((levels + 1.0) * 3.7e-7).astype(np.float32)).reshape(B, 1, N)
    assert all(len(np.unique((score[b, 0] + 1e-8).numpy())) == N for b in range(B))
    z = O.zscore(score)
    state = O.blend_boundaries(None, O.batch_quantiles(z.reshape(B, 1, N, 1), nb), nb, 0.99)
    member = O.bin_membership(z, state)
    cap = member.squeeze(1).sum(1)
    w = torch.rand(B, nb, generator=torch.Generator().manual_seed(2))
    counts = O.allocate_counts(w, cap, M)
    noise = torch.from_numpy(synth.exp1((B * nb, N), 22))
    ref = O.select_indices(score, member, counts, M, mode, 0.1, noise)
    bits = (member.squeeze(1).long() * (1 << torch.arange(nb))).sum(-1).to(torch.uint8)
    got = ops().stage_bin_select(score.reshape(B, N).to(DEV), z.reshape(B, N).to(DEV), bits.to(DEV),
                                 counts.to(DEV), M, mode, 0.1, noise.to(DEV))
    assert torch.equal(got.cpu(), ref.reshape(B, M))


def test_bin_select_random_mode_full_size_vs_oracle():
    """`random` (Boltzmann) selection at the metric size from the oracle's stage inputs.  Its key
    exp(tanh(z)/T) / sum / q goes through tanh and exp: MKL's closed-source vector math library on the CPU (this
    torch build routes contiguous float exp / tanh to vsExp / vsTanh: tools/libm_probe shows that a bit-exact
    restatement of Sleef u10 -- the open routine torch also ships -- differs from torch.exp in 9.5 % of arguments),
    ocml on the GPU, each within 1 ulp of the true value but not of each other, so two keys closer than a few ulp
    may trade places.  Pinned here: every position where the two index tensors differ is such a near-tie (the two keys
    involved agree to 1e-6 relative), there are at most a handful of them, and the per-bin SETS differ only by
    such near-ties at the cut."""
    B, N, nb, M = 8, 2048, 6, 1024
    score = torch.from_numpy(np.abs(synth.normal((B, 1, N), 31)) * 1e-4 + 1e-6)
    z = O.zscore(score)
    state = O.blend_boundaries(None, O.batch_quantiles(z.reshape(B, 1, N, 1), nb), nb, 0.99)
    member = O.bin_membership(z, state)
    cap = member.squeeze(1).sum(1)
    w = torch.rand(B, nb, generator=torch.Generator().manual_seed(5))
    counts = O.allocate_counts(w, cap, M)
    noise = torch.from_numpy(synth.exp1((B * nb, N), 32))
    ref = O.select_indices(score, member, counts, M, "random", 0.1, noise).reshape(B, M)
    bits = (member.squeeze(1).long() * (1 << torch.arange(nb))).sum(-1).to(torch.uint8)
    got = ops().stage_bin_select(score.reshape(B, N).to(DEV), z.reshape(B, N).to(DEV), bits.to(DEV),
                                 counts.to(DEV), M, "random", 0.1, noise.to(DEV)).cpu()
    diff = got != ref
    assert int(diff.sum()) <= 8, int(diff.sum())
    if bool(diff.any()):
        # the keys of the oracle (fp32, as the reference forms them) for the points involved
        zz = O.zscore(score).reshape(B, N)
        p = torch.exp(torch.tanh(zz) / 0.1)
        bin_of = member.squeeze(1).float().argmax(-1)
        for b, pos in diff.nonzero().tolist():
            i, j = int(got[b, pos]), int(ref[b, pos])
            assert int(bin_of[b, i]) == int(bin_of[b, j]), "a swap never crosses bins"
            t = int(bin_of[b, i])
            ki = float(p[b, i] / noise[b * nb + t, i])
            kj = float(p[b, j] / noise[b * nb + t, j])
            assert abs(ki - kj) <= 1e-6 * max(ki, kj), (b, pos, ki, kj)


def test_select_chain_with_a_constant_score_cloud():
    """One cloud of the batch has a constant score: its z-score is 0/0 (or x/0), every comparison with the
    boundaries is false, all its bins are empty.  The reference itself fails there (its float -> int count allocation
    yields INT_MIN entries and torch.stack raises); ours must stay in bounds -- bin 0 of that cloud is handed all M
    picks and the select kernel serves them from the whole cloud -- and the healthy clouds of the batch must come out
    exactly as the oracle's (static boundaries, so the poisoned batch quantiles of the reference do not enter)."""
    B, N, nb, M = 3, 2048, 6, 1024
    score = torch.from_numpy(np.abs(synth.normal((B, 1, N), 77)) * 1e-4 + 1e-6)
    score[1] = 0.25  # exactly representable: mean exact, (x - mean) / std = 0 / 0
    state = O.static_boundary_state([1.2, 0.4, -0.1, -0.5, -0.9], nb)
    o_ = ops()
    z = o_.stage_zscore(score.reshape(B, N).to(DEV))
    assert bool(torch.isnan(z[1]).all()) and bool(torch.isfinite(z[[0, 2]]).all())
    tok = torch.from_numpy(synth.normal((B, N, nb), 78)).to(DEV)
    member, cap, w_pre, w = o_.stage_bin_assign(z, tok, state[0].to(DEV), state[1].to(DEV), False)
    assert int(cap[1].sum()) == 0 and int(member[1].sum()) == 0
    counts = o_.stage_alloc_counts(w, cap, M)
    assert counts[1].tolist() == [M, 0, 0, 0, 0, 0]
    idx = o_.stage_bin_select(score.reshape(B, N).to(DEV), z, member, counts, M, "topk", 0.1).cpu()
    assert int(idx.min()) >= 0 and int(idx.max()) < N
    assert all(len(set(r.tolist())) == M for r in idx)
    # healthy clouds: the oracle on the same stage inputs
    zc = z.cpu().reshape(B, 1, N)
    for b in (0, 2):
        mem = O.bin_membership(zc[b:b + 1], state)
        capb = mem.squeeze(1).sum(1)
        assert torch.equal(cap[b:b + 1].cpu().long(), capb)
        cnt = O.allocate_counts(w[b:b + 1].cpu(), capb, M)
        assert torch.equal(counts[b:b + 1].cpu(), cnt)
        ref = O.select_indices(score[b:b + 1], mem, cnt, M, "topk", 0.1).reshape(1, M)
        assert torch.equal(idx[b:b + 1], ref)


def test_bin_select_ties_break_by_ascending_index():
    """torch.sort leaves the order of exactly equal keys unspecified; ours is defined."""
    B, N, nb, M = 1, 256, 2, 64
    score = torch.full((B, N), 1e-4)
    z = torch.zeros(B, N)
    member = torch.ones(B, N, dtype=torch.uint8)  # everyone in bin 0
    counts = torch.tensor([[M, 0]], dtype=torch.int32)
    got = ops().stage_bin_select(score.to(DEV), z.to(DEV), member.to(DEV), counts.to(DEV), M, "topk", 0.1)
    assert torch.equal(got.cpu(), torch.arange(M).reshape(1, M))


def test_gather_rows_and_points():
    B, N, D, M = 2, 300, 128, 77
    O_ = torch.from_numpy(synth.normal((B, N, D), 1)).to(DEV)
    idx = torch.stack([torch.randperm(N)[:M] for _ in range(B)]).to(DEV)
    got = ops().stage_gather_rows(O_, idx)
    ref = torch.gather(O_, 1, idx[..., None].expand(-1, -1, D)).permute(0, 2, 1)
    assert torch.equal(got, ref)
    xyz = torch.from_numpy(synth.xyz_clouds(B, N, 4)).to(DEV)
    assert torch.equal(ops().gather_by_idx(xyz, idx.unsqueeze(1)), O.gather_points(xyz.cpu(), idx.cpu().unsqueeze(1)).to(DEV))


def test_abi_rejects_bad_arguments():
    from samble_amd import _lib
    pts = torch.from_numpy(synth.normal((1, 4, 8), 77)).to(DEV)
    with pytest.raises(ValueError):
        ops().stage_knn(pts, pts, 9)  # more neighbours than keys
    # (a list length the kernels are not built for is the head of the next one's list since round 6)
    assert torch.equal(ops().stage_knn(pts, pts, 5), ops().stage_knn(pts, pts, 8)[:, :, :5])
    with pytest.raises(ValueError):
        ops().stage_bin_select(torch.zeros(1, 8, device=DEV), torch.zeros(1, 8, device=DEV),
                               torch.zeros(1, 8, dtype=torch.uint8, device=DEV),
                               torch.zeros(1, 2, dtype=torch.int32, device=DEV), 4, "bogus", 0.1)
    with pytest.raises(_lib.SambleError):
        ops().stage_knn(torch.zeros(1, 4, 8), torch.zeros(1, 4, 8), 3)  # CPU tensors: no fallback


def test_blend_boundaries_kernel_is_bitwise_the_reference_expression():
    """utils/ops.py:201-233: first call stores the quantiles, later calls blend in place; the fused kernel
    must round exactly like the reference's separate fp32 multiply / multiply / add."""
    o_ = ops()
    nb, mu = 6, 0.99
    q1 = torch.from_numpy(synth.normal((nb - 1,), 41)).to(DEV)
    st = o_.blend_boundaries(None, q1, nb, mu)
    assert st[0].shape == (1, 1, 1, nb) and torch.isposinf(st[0][0, 0, 0, 0]) and torch.isneginf(st[1][0, 0, 0, -1])
    assert torch.equal(st[0][0, 0, 0, 1:], q1) and torch.equal(st[1][0, 0, 0, :-1], q1)
    ref = st[0][0, 0, 0, 1:].cpu().clone()
    up_ptr = st[0].data_ptr()
    for k in range(5):
        qk = torch.from_numpy(synth.normal((nb - 1,), 42 + k)).to(DEV)
        st = o_.blend_boundaries(st, qk, nb, mu)
        ref = ref * mu + (1 - mu) * qk.cpu()  # the reference's expression, fp32 on the CPU
        assert st[0].data_ptr() == up_ptr, "blended in place"
        assert torch.equal(st[0][0, 0, 0, 1:].cpu(), ref) and torch.equal(st[1][0, 0, 0, :-1].cpu(), ref)


# ---------------------------------------------------------------------------------------------
# split-bf16 operand images (csrc/tri_dev.h) and the kernels that only exist in that mode
# ---------------------------------------------------------------------------------------------
def _bf16_planes_to_f64(words: np.ndarray) -> np.ndarray:
    """uint16 bf16 bit patterns -> float64 values."""
    return (words.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def test_operand_images_hold_every_significand_bit():
    """Decode the two image layouts on the host exactly as tri_dev.h documents them: the three bf16 planes
    of every element sum to the fp32 value (all 24 bits), rows past the end are zeros."""
    B, R, D = 2, 77, 128
    x = torch.from_numpy(synth.normal((B, R, D), 31) * np.exp(synth.normal((B, R, 1), 32) * 3)).float().to(DEV)
    rm, tr = ops().stage_tri_split(x, want_rm=True, want_tr=True)
    ntiles = (R + 31) // 32
    xpad = np.zeros((B, ntiles * 32, D), np.float64)
    xpad[:, :R] = x.cpu().numpy().astype(np.float64)
    # RM: tile -> [chunk c = 3 * (channel / 8) + piece][row r][8 bf16]
    a = rm.cpu().numpy().view(np.uint16).reshape(B, ntiles, 16, 3, 32, 8)
    val = _bf16_planes_to_f64(a).sum(3)                      # (B, tile, group, row, e)
    got = val.transpose(0, 1, 3, 2, 4).reshape(B, ntiles * 32, D)
    assert np.array_equal(got, xpad)
    # TR: tile -> [chunk c = 3 * (2 s + h) + piece][channel d][8 bf16], element e = tile row 16 s + 8 (e >> 2) + 4 h + (e & 3)
    t = tr.cpu().numpy().view(np.uint16).reshape(B, ntiles, 4, 3, 128, 8)
    tv = _bf16_planes_to_f64(t).sum(3)                       # (B, tile, cg, d, e)
    rows = np.empty((4, 8), np.int64)
    for cg in range(4):
        for e in range(8):
            rows[cg, e] = 16 * (cg >> 1) + 8 * (e >> 2) + 4 * (cg & 1) + (e & 3)
    got_t = np.zeros_like(xpad).reshape(B, ntiles, 32, D)
    for cg in range(4):
        for e in range(8):
            got_t[:, :, rows[cg, e], :] = tv[:, :, cg, :, e]
    assert np.array_equal(got_t.reshape(B, ntiles * 32, D), xpad)


def test_qkv_split_matches_the_single_operand_splits():
    B, N, nt, D = 2, 200, 6, 128
    qkv = torch.from_numpy(synth.normal((B, N + nt, 3 * D), 77)).to(DEV)
    o_ = ops()
    q_img, k_img, v_tr, k_tr, v_rm = o_.stage_tri_split_qkv(qkv, N, for_backward=True)
    assert torch.equal(q_img, o_.stage_tri_split(qkv[:, :N, :D])[0])
    k_rm_ref, k_tr_ref = o_.stage_tri_split(qkv[:, :, D:2 * D], want_rm=True, want_tr=True)
    k_rm_ref = o_.stage_k_logit_form(k_rm_ref, qkv[:, :, D:2 * D])  # the K row image is handed over in its logit form
    v_rm_ref, v_tr_ref = o_.stage_tri_split(qkv[:, :, 2 * D:], want_rm=True, want_tr=True)
    v_rm_ref = o_.stage_k_logit_form(v_rm_ref, qkv[:, :, 2 * D:])  # (the backward's V row image: the same form)
    assert torch.equal(k_img, k_rm_ref) and torch.equal(k_tr, k_tr_ref)
    assert torch.equal(v_rm, v_rm_ref) and torch.equal(v_tr, v_tr_ref)


@pytest.mark.parametrize("Nq,Nk,K", [(512, 512, 32), (333, 1500, 16), (2048, 2048, 32)])
def test_knn_split_bf16_kernel_agrees_with_the_fp32_kernel(Nq, Nk, K):
    """Same selection, different rounding of the Gram entries: the neighbour SETS agree except at
    near-ties, and both agree with fp64."""
    B, C = 2, 128
    a = torch.from_numpy(synth.normal((B, C, Nq), 5)).to(DEV)
    bb = a if Nq == Nk else torch.from_numpy(synth.normal((B, C, Nk), 6)).to(DEV)
    o_ = ops()
    ref = o_.stage_knn(a, bb, K, variant=o_.KNN_FP32_MFMA)
    got, dist = o_.stage_knn(a, bb, K, want_dist=True, variant=0)
    assert set_agreement(got.cpu(), ref.cpu()) >= 0.9995
    d = ((a.double().permute(0, 2, 1)[:, :, None, :] - bb.double().permute(0, 2, 1)[:, None, :, :]) ** 2).sum(-1)
    want = d.topk(K, dim=-1, largest=False)[1]
    assert set_agreement(got.cpu(), want.cpu()) >= 0.9995
    assert bool((dist[:, :, 1:] >= dist[:, :, :-1]).all()), "nearest first"


def test_split_bf16_backward_variants_agree():
    """dS-map backward (4 products per tile, default) vs the fused dP/dV/dK kernel (5 products, no dS map):
    same gradients up to summation order."""
    B, N, nt, M, D = 2, 1000, 6, 333, 128
    q, k, v = _qkv(B, N, nt, 4321)
    g = torch.from_numpy(synth.normal((B, D, M), 9)).to(DEV)
    idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b))[:M] for b in range(B)]).to(DEV)
    o_ = ops()
    old = o_.MATRIX_MODE
    out = {}
    try:
        o_.MATRIX_MODE = "tri"
        qg, kg, vg = q.to(DEV), k.to(DEV), v.to(DEV)
        smap, lse, _ = o_.stage_attn_stats(qg, kg, N, nt)
        x_ds = o_.stage_attn_rows(smap, lse, vg, idx, N, nt)
        for use_map in (1, 0):
            dq = torch.full((B, N, D), float("nan"), device=DEV)
            dk = torch.full((B, N + nt, D), float("nan"), device=DEV)
            dv = torch.full((B, N + nt, D), float("nan"), device=DEV)
            o_.stage_attn_rows_bwd(qg, kg, vg, smap, lse, x_ds, idx, g, N, nt, dq, dk, dv, variant=1 - use_map)
            out[use_map] = (dq, dk, dv)
    finally:
        o_.MATRIX_MODE = old
    for a, b2, name in zip(out[1], out[0], ("dq", "dk", "dv")):
        assert torch.isfinite(a).all() and torch.isfinite(b2).all(), name
        scale = float(a.abs().max())
        assert float((a - b2).abs().max()) <= 2e-5 * scale + 1e-7, name


@pytest.mark.parametrize("B,N,nt,M", [(1, 64, 0, 32), (3, 257, 0, 100), (5, 96, 2, 96), (2, 33, 6, 1), (9, 512, 6, 200)])
def test_split_bf16_edge_shapes(B, N, nt, M, matrix_mode):
    """No token keys, clouds smaller than a workgroup, a single sampled row, batch sizes that are not a
    multiple of the 8 XCDs: forward + backward against fp64 autograd."""
    D = 128
    q, k, v = _qkv(B, N, nt, 7000 + N + nt)
    g = torch.from_numpy(synth.normal((B, D, M), 70)).to(DEV)
    idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b))[:M] for b in range(B)])
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    s = (qd @ kd.transpose(1, 2)) / math.sqrt(D)
    rows = torch.gather(torch.softmax(s, -1) @ vd, 1, idx[..., None].expand(-1, -1, D))
    rows.permute(0, 2, 1).backward(g.double().cpu())
    o_ = ops()
    qg, kg, vg = q.to(DEV), k.to(DEV), v.to(DEV)
    smap, lse, _ = o_.stage_attn_stats(qg, kg, N, nt)
    x_ds = o_.stage_attn_rows(smap, lse, vg, idx.to(DEV), N, nt)
    torch.testing.assert_close(x_ds.cpu().double(), rows.detach().permute(0, 2, 1), rtol=2e-4, atol=2e-5)
    dq = torch.full((B, N, D), float("nan"), device=DEV)
    dk = torch.full((B, N + nt, D), float("nan"), device=DEV)
    dv = torch.full((B, N + nt, D), float("nan"), device=DEV)
    o_.stage_attn_rows_bwd(qg, kg, vg, smap, lse, x_ds, idx.to(DEV), g, N, nt, dq, dk, dv)
    for got, ref, name in ((dq, qd.grad, "dq"), (dk, kd.grad, "dk"), (dv, vd.grad, "dv")):
        assert torch.isfinite(got).all(), name
        scale = ref.abs().max().item()
        assert (got.cpu().double() - ref).abs().max().item() <= 3e-5 * scale + 1e-7, name


@pytest.mark.parametrize("B,N,nb,nt", [(2, 256, 6, 6), (32, 2048, 6, 6), (3, 1000, 4, 4), (5, 512, 6, 1), (16, 8192, 6, 6)])
def test_fused_select_chain_equals_the_stage_kernels(B, N, nb, nt):
    """csrc/chain.hip (score + z + batch quantiles in one launch, boundary update + bins + counts in another) against
    the stand-alone stage kernels it replaces: every output BITWISE equal, over a first call (boundaries initialised
    from the quantiles) and a second one (momentum blend in place), and with static boundaries."""
    o_ = ops()
    assert o_.chain_supported(B, N, nb)
    K, M = 32, N // 2
    q, k, v = _qkv(B, N, nt, 77 + N)
    q, k = q * 0.3, k * 0.3
    nn = torch.stack([torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b * N + i))[:K]
                                   for i in range(N)]) for b in range(B)]).int().to(DEV) if N <= 1024 else \
        o_.stage_knn(q.permute(0, 2, 1).contiguous().to(DEV), q.permute(0, 2, 1).contiguous().to(DEV), K)
    smap, lse, tok = o_.stage_attn_stats(q.to(DEV), k.to(DEV), N, nt)
    state_a = state_b = state_c = None
    for call in range(2):
        # stage kernels
        score, z, indeg = o_.stage_sparse_score_map(smap, lse, nn, "sparse_col_sqr")
        quant = o_.stage_batch_quantiles(z, nb)
        state_a = o_.blend_boundaries(state_a, quant, nb, 0.99)
        member, cap, w_pre, w = o_.stage_bin_assign(z, tok, state_a[0], state_a[1], False)
        counts = o_.stage_alloc_counts(w, cap, M)
        # fused chain
        score2, z2, indeg2, quant2, cws = o_.stage_score_quantiles(smap, lse, nn, "sparse_col_sqr", nb, True)
        state_b, member2, cap2, w_pre2, w2, counts2 = o_.stage_bin_plan(z2, tok, quant2, state_b, nb, 0.99, False, M, cws)
        torch.cuda.synchronize()
        for a, b2, what in ((score, score2, "score"), (z, z2, "z"), (indeg, indeg2, "indeg"), (quant, quant2, "quantiles"),
                            (state_a[0], state_b[0], "upper"), (state_a[1], state_b[1], "lower"),
                            (member, member2, "member"), (cap, cap2, "cap"), (w_pre, w_pre2, "w_pre"), (w, w2, "w"),
                            (counts, counts2, "counts")):
            assert torch.equal(a, b2), (what, call)
        assert bool((counts2.sum(1) == M).all())
        # the same chain as ONE launch (a single rank has nothing to exchange between the quantiles and the bin plan)
        (score5, z5, indeg5, quant5, state_c, member5, cap5, w_pre5, w5, counts5, _) = o_.stage_select_chain(
            lse, tok, nn, "sparse_col_sqr", nb, True, state_c, 0.99, False, M, smap=smap)
        for a, b2, what in ((score, score5, "score"), (z, z5, "z"), (indeg, indeg5, "indeg"), (quant, quant5, "quantiles"),
                            (state_a[0], state_c[0], "upper"), (state_a[1], state_c[1], "lower"),
                            (member, member5, "member"), (cap, cap5, "cap"), (w_pre, w_pre5, "w_pre"), (w, w5, "w"),
                            (counts, counts5, "counts")):
            assert torch.equal(a, b2), ("one launch", what, call)
        lse = lse + 0.01 * (call + 1)  # other scores for the second call
    # static boundaries: no quantiles, state untouched
    up = state_a[0].clone()
    score3, z3, _, quant3, cws = o_.stage_score_quantiles(smap, lse, nn, "sparse_col_sqr", nb, False)
    assert quant3 is None
    st, member3, cap3, _, w3, counts3 = o_.stage_bin_plan(z3, tok, None, [state_a[0], state_a[1]], nb, 0.99, True, M, cws)
    member4, cap4, _, w4 = o_.stage_bin_assign(z3, tok, state_a[0], state_a[1], True)
    assert torch.equal(st[0], up) and torch.equal(member3, member4) and torch.equal(cap3, cap4) and torch.equal(w3, w4)
    assert torch.equal(counts3, o_.stage_alloc_counts(w4, cap4, M))
    (_, z6, _, quant6, st6, member6, cap6, _, w6, counts6, _) = o_.stage_select_chain(
        lse, tok, nn, "sparse_col_sqr", nb, False, [state_a[0], state_a[1]], 0.99, True, M, smap=smap)
    assert quant6 is None and torch.equal(st6[0], up) and torch.equal(z6, z3) and torch.equal(member6, member4)
    assert torch.equal(cap6, cap4) and torch.equal(w6, w4) and torch.equal(counts6, counts3)


def test_select_chain_gives_up_cleanly_and_the_layer_falls_back():
    """The grid barrier of the fused chain is bounded, and its give-up is clean (csrc/chain.hip grid_barrier / chain_bail):
    with the give-up injected (the entries' spin_budget argument = 0xFFFFFFFF) the barrier gives up at once -- the kernel ends
    (no trap: the context lives on), leaves valid placeholder integers and raises the workspace's status word; the layer
    sees the word at its next call, raises SAMBLE_E_TIMEOUT once, and from then on runs the stand-alone stage kernels,
    whose results equal those of a layer that never used the chain."""
    from samble_amd import _lib, sampler_config
    from samble_amd.downsample import DownSampleToken
    B, C, N, M, nb = 4, 128, 512, 256, 6
    x = torch.from_numpy(synth.features(B, C, N, 611)).to(DEV)
    noise = torch.from_numpy(synth.exp1((B * nb, N), 612)).to(DEV)
    torch.manual_seed(9)
    mod = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0).to(DEV)
    assert ops().chain_supported(B, N, nb)
    try:
        ops().CHAIN_SPIN_BUDGET = 0xFFFFFFFF
        (x_ds, idx), _ = mod(x, noise=noise)               # the chain gives up inside this call
        torch.cuda.synchronize()                           # ... and the process is still alive
    finally:
        ops().CHAIN_SPIN_BUDGET = 0
    i = idx[:, 0].cpu()
    assert int(i.min()) >= 0 and int(i.max()) < N and all(len(set(r.tolist())) == M for r in i)   # placeholders, but valid
    assert bool((mod.k_point_to_choose.cpu()[:, 0] == M).all()) and torch.isfinite(x_ds).all()
    assert mod.bin_boundaries is not None and bool(torch.isnan(mod.bin_boundaries[0][0, 0, 0, 1:]).all())  # never written
    with pytest.raises(_lib.SambleError, match="SAMBLE_E_TIMEOUT"):
        mod(x, noise=noise)
    assert mod._chain_watch.tripped and mod._chain_watch.reported
    assert mod.bin_boundaries is None                      # the give-up of a FIRST call leaves no state: first call again
    (x_ds2, idx2), _ = mod(x, noise=noise)                 # stage kernels now; no second raise
    ref = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0).to(DEV)
    ref.load_state_dict(mod.state_dict())
    ref._chain_watch.observed = ref._chain_watch.reported = True   # a layer that never takes the chain
    (x_ds3, idx3), _ = ref(x, noise=noise)
    assert torch.equal(idx2, idx3) and torch.equal(x_ds2, x_ds3)
    good = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0).to(DEV)   # and the chain itself, with its real budget
    good.load_state_dict(mod.state_dict())
    (x_ds4, idx4), _ = good(x, noise=noise)
    assert torch.equal(idx4, idx3) and torch.equal(x_ds4, x_ds3) and not good._chain_watch.timed_out(sync=True)
    # a give-up in a LATER call leaves the boundaries of the last good call, bit for bit; and the word is reported once even
    # when the sampler core's own look at the mailbox (before its chain launch) is the call site that finds it
    before = [t.clone() for t in good.bin_boundaries]
    try:
        ops().CHAIN_SPIN_BUDGET = 0xFFFFFFFF
        good(x, noise=noise)
        torch.cuda.synchronize()
    finally:
        ops().CHAIN_SPIN_BUDGET = 0
    assert torch.equal(good.bin_boundaries[0], before[0]) and torch.equal(good.bin_boundaries[1], before[1])
    assert good._chain_usable(B, N, nb) is False and good._chain_watch.observed and not good._chain_watch.reported
    with pytest.raises(_lib.SambleError, match="SAMBLE_E_TIMEOUT"):
        good(x, noise=noise)
    (x_ds5, idx5), _ = good(x, noise=noise)
    assert torch.equal(good.bin_boundaries[0][0, 0, 0, 0:1].cpu(), torch.tensor([float("inf")]))
    assert bool(torch.isfinite(good.bin_boundaries[0][0, 0, 0, 1:]).all())
    # a copy of the module (EMA / SWA) does not share or clone the pinned mailbox
    import copy
    import pickle
    twin = copy.deepcopy(mod._chain_watch)
    assert twin.flag is None and twin.observed and twin.reported
    assert pickle.loads(pickle.dumps(mod._chain_watch)).flag is None
    donor = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0).to(DEV)
    donor.load_state_dict(mod.state_dict())
    with torch.no_grad():
        donor(x, noise=noise)                              # its mailbox is pinned now
    assert donor._chain_watch.flag is not None and donor._chain_watch.flag.is_pinned()
    donor.attention_bins_beforesoftmax = None              # (plain tensor attributes deep-copy; nothing else is in the way)
    fresh = copy.deepcopy(donor)
    assert fresh._chain_watch.flag is None
    with torch.no_grad():
        (x_ds6, idx6), _ = fresh(x, noise=noise)           # pins its own mailbox and runs the chain
    torch.cuda.synchronize()
    assert fresh._chain_watch.flag is not None and fresh._chain_watch.flag.is_pinned()
    assert fresh._chain_watch.flag.data_ptr() != donor._chain_watch.flag.data_ptr()


def test_give_up_on_one_rank_does_not_reach_the_others_quantiles():
    """Two-launch form (the multi-rank path): score_quantiles writes (nb,) = quantiles + a validity count of 1; a give-up
    writes zeros.  bin_plan divides the all-reduced sums by the all-reduced count: with one of two "ranks" dead the
    healthy rank's boundaries are exactly its own quantiles (sum / 1), with both alive the mean (sum / 2) as the
    reference's `all_reduce; / world_size` (utils/ops.py:191-199)."""
    o_ = ops()
    B, N, nb, M, K = 4, 512, 6, 256, 32
    gen = torch.Generator().manual_seed(4)
    lse = torch.randn(B, N, generator=gen).to(DEV)
    tok = torch.randn(B, N, nb, generator=gen).to(DEV)
    nn = torch.stack([torch.stack([torch.randperm(N, generator=gen)[:K] for _ in range(N)]) for _ in range(B)]).int().to(DEV)
    smap = torch.randn(B, N, N + nb, generator=gen).to(DEV)
    score, z, indeg, q_ok, cws = o_.stage_score_quantiles(smap, lse, nn, "sparse_col_sqr", nb, True, counted=True)
    assert q_ok.shape == (nb,) and float(q_ok[-1]) == 1.0
    try:
        o_.CHAIN_SPIN_BUDGET = 0xFFFFFFFF
        _, _, _, q_dead, cws_dead = o_.stage_score_quantiles(smap, lse + 1.0, nn, "sparse_col_sqr", nb, True, counted=True)
        torch.cuda.synchronize()
    finally:
        o_.CHAIN_SPIN_BUDGET = 0
    assert torch.equal(q_dead.cpu(), torch.zeros(nb))
    # "all-reduce" of a healthy and a dead rank, seen from the healthy one
    summed = q_ok + q_dead
    st, *_ = o_.stage_bin_plan(z, tok, summed, None, nb, 0.99, False, M, cws, counted=True)
    assert torch.equal(st[0][0, 0, 0, 1:], q_ok[:-1]) and torch.equal(st[1][0, 0, 0, :-1], q_ok[:-1])
    # two healthy ranks with different quantiles: the reference's expression, bit for bit (also for a divisor of 3)
    for world in (2, 3):
        other = q_ok.clone()
        other[:-1] += 0.125
        summed = q_ok.clone()
        for _ in range(world - 1):
            summed = summed + other
        _, _, _, _, cws2 = o_.stage_score_quantiles(smap, lse, nn, "sparse_col_sqr", nb, True, counted=True)
        st2, *_ = o_.stage_bin_plan(z, tok, summed, None, nb, 0.99, False, M, cws2, counted=True)
        # (true division, as ATen's CPU kernel -- the oracle's -- does it; ATen's CUDA kernel multiplies by 1 / world: the
        # same bits for the power-of-two world sizes a node has)
        assert torch.equal(st2[0][0, 0, 0, 1:].cpu(), summed[:-1].cpu() / world)


@pytest.mark.parametrize("B,N,nt,M,K", [(2, 256, 6, 128, 32), (3, 1000, 4, 333, 16), (32, 2048, 6, 1024, 32),
                                         (1, 77, 1, 40, 16), (4, 4096, 6, 2048, 32), (1, 8500, 6, 700, 32),
                                         (2, 33, 8, 7, 16), (5, 513, 3, 100, 32), (1, 2047, 6, 1023, 32),
                                         (7, 96, 0, 96, 16)])
def test_map_free_forward_equals_the_map_pipeline(B, N, nt, M, K):
    """The forward that never builds the N x (N+nt) logit map (attn_stats_nl_tri + attn_rows_rc_tri, csrc/attn_tri.hip)
    against the two-pass map kernels it replaces: lse, token logits, the K neighbour logits of every row, every score
    mode's score / z / in-degree, x_ds, the P rows of the sampled points and all three gradients BITWISE equal."""
    o_ = ops()
    old = o_.MATRIX_MODE
    o_.MATRIX_MODE = "tri"
    try:
        q, k, v = _qkv(B, N, nt, 4000 + N)
        qkv = torch.cat((torch.cat((q, torch.zeros(B, nt, 128)), 1) * 0.3, k * 0.3, v), dim=2).to(DEV).contiguous()
        qd, kd, vd = qkv[:, :N, :128], qkv[:, :, 128:256], qkv[:, :, 256:]
        g = torch.Generator().manual_seed(N)
        nn = torch.stack([torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(N)]) for _ in range(B)]) \
            .int().to(DEV) if N <= 1024 else o_.stage_knn(qd.permute(0, 2, 1).contiguous(), qd.permute(0, 2, 1).contiguous(), K)
        idx = torch.stack([torch.randperm(N, generator=g)[:M] for _ in range(B)]).to(DEV)
        imgs = o_.stage_tri_split_qkv(qkv, N, for_backward=True)
        smap, lse, tok = o_.stage_attn_stats(qd, kd, N, nt, images=imgs[:2])
        nn_sorted, masks = o_.stage_nn_prepare(nn)
        assert torch.equal(nn_sorted, nn.sort(dim=-1).values)
        bits = torch.zeros((B, N, 32 * ((N + 31) // 32)), dtype=torch.bool, device=DEV)
        bits.scatter_(2, nn.long(), True)
        words = (bits.view(B, N, -1, 32).long() << torch.arange(32, device=DEV)).sum(-1)   # (B, N, T) as uint32 values
        assert torch.equal(masks.long() & 0xFFFFFFFF, words.permute(0, 2, 1))
        nl, lse2, tok2, _ = o_.stage_attn_stats_nl(imgs[0], imgs[1], masks, B, N, nt, K)
        assert torch.equal(lse2, lse) and torch.equal(tok2, tok)
        assert torch.equal(nl, torch.gather(smap, 2, nn_sorted.long()))
        for mode in ("sparse_col_sum", "sparse_col_avg", "sparse_col_sqr", "sparse_row_sum", "sparse_row_std"):
            a = o_.stage_sparse_score_map(smap, lse, nn, mode)
            b2 = o_.stage_sparse_score_map(nl, lse, nn_sorted, mode, compact=True)
            for x1, x2, what in zip(a, b2, ("score", "z", "indeg")):
                assert torch.equal(x1, x2), (mode, what)
            if N > 8192:  # the pass's LDS accumulators hold N <= 8192 columns: longer clouds keep the compact route
                with pytest.raises(Exception):
                    o_.stage_attn_stats_nl(imgs[0], imgs[1], masks, B, N, nt, K, want_nl=False, score=(nn_sorted, mode, None))
                continue
            # ... and with the statistics accumulated by the pass itself (no logit array at all)
            none, lse3, tok3, sws = o_.stage_attn_stats_nl(imgs[0], imgs[1], masks, B, N, nt, K, want_nl=False,
                                                           score=(nn_sorted, mode, None))
            assert none is None and torch.equal(lse3, lse) and torch.equal(tok3, tok)
            c3 = o_.stage_sparse_score_map(None, lse, nn_sorted, mode, ws=sws)
            for x1, x3, what in zip(a, c3, ("score", "z", "indeg")):
                assert torch.equal(x1, x3), (mode, what, "fused")
        if o_.chain_supported(B, N, 6) and N <= 8192:
            ref = o_.stage_score_quantiles(smap, lse, nn, "sparse_col_sqr", 6, True)
            _, _, _, sws = o_.stage_attn_stats_nl(imgs[0], imgs[1], masks, B, N, nt, K, want_nl=False,
                                                  score=(nn_sorted, "sparse_col_sqr", 6))
            got = o_.stage_score_quantiles(None, lse, nn_sorted, "sparse_col_sqr", 6, True, ws=sws)
            for x1, x2, what in zip(ref[:4], got[:4], ("score", "z", "indeg", "quantiles")):
                assert torch.equal(x1, x2), ("chain", what)
        x_ds = o_.stage_attn_rows(smap, lse, vd, idx, N, nt, v_image=imgs[2])
        x_ds2, pmap = o_.stage_attn_rows_recompute(imgs[0], imgs[1], imgs[2], lse, idx, N, nt, True)
        x_ds3, none = o_.stage_attn_rows_recompute(imgs[0], imgs[1], imgs[2], lse, idx, N, nt, False)
        assert none is None and torch.equal(x_ds2, x_ds) and torch.equal(x_ds3, x_ds)
        rows = torch.gather(smap, 1, idx[:, :, None].expand(-1, -1, smap.shape[2]))
        p_ref = torch.exp(rows[:, :, :N + nt] - torch.gather(lse, 1, idx)[:, :, None])
        torch.testing.assert_close(pmap[:, :, :N + nt], p_ref, rtol=2e-6, atol=1e-12)
        assert bool((pmap[:, :, N + nt:] == 0).all())
        gr = torch.randn(B, 128, M, generator=g).to(DEV)
        grads = []
        for m, variant in ((smap, 0), (pmap, o_.ROWS_BWD_PMAP)):
            dqkv = torch.full_like(qkv, float("nan"))
            o_.stage_attn_rows_bwd(qd, kd, vd, m, lse, x_ds, idx, gr, N, nt, dqkv[:, :N, :128], dqkv[:, :, 128:256],
                                   dqkv[:, :, 256:], images=(imgs[3], imgs[4]), variant=variant)
            dqkv[:, N:, :128] = 0
            grads.append(dqkv)
        torch.cuda.synchronize()
        assert not bool(torch.isnan(grads[1]).any())
        assert torch.equal(grads[0], grads[1])
    finally:
        o_.MATRIX_MODE = old


def _k_image_live_equal(a, b2):
    """Row images in their logit form (K, and the backward's V): the two fp16 planes of every (group, row) chunk and the
    tile's scale word; the rest of the third piece slots is dead space (the projection leaves it unwritten)."""
    va, vb = a.view(-1, 16, 3, 32, 16), b2.view(-1, 16, 3, 32, 16)
    return bool(torch.equal(va[:, :, :2], vb[:, :, :2]) and torch.equal(va[:, 0, 2, 0, :4], vb[:, 0, 2, 0, :4]))


@pytest.mark.parametrize("B,N,nt", [(2, 256, 6), (3, 1000, 4), (1, 77, 1), (32, 2048, 6), (2, 96, 0), (1, 20, 8)])
def test_projection_writes_the_operand_images_itself(B, N, nt):
    """samble_proj_fwd_split_tri_f32 (images of the full point tiles from the projection kernel's accumulators + a split
    launch over the tiles with token rows / a ragged end) against projection + tri_split_qkv: the same bytes in qkv and
    in all five images."""
    o_ = ops()
    old = o_.MATRIX_MODE
    o_.MATRIX_MODE = "tri"
    try:
        x = torch.from_numpy(synth.normal((B, 128, N), 900 + N)).to(DEV)
        tokens = torch.from_numpy(synth.normal((128, max(nt, 1)), 901))[:, :nt].contiguous().to(DEV)
        w = (torch.from_numpy(synth.normal((384, 128), 902)) * 0.1).to(DEV)
        qkv = o_.stage_proj_fwd(x, tokens, w)
        imgs = o_.stage_tri_split_qkv(qkv, N, for_backward=True)
        for want in ("fwd+bwd", "fwd"):
            qkv2, imgs2 = o_.stage_proj_fwd(x, tokens, w, images=want)
            torch.cuda.synchronize()
            assert torch.equal(qkv2, qkv)
            assert len(imgs2) == (6 if want == "fwd+bwd" else 3)
            if want == "fwd+bwd":  # ... + the transposed image of W for the projection's own backward
                # (two fp16 planes under per-tile scales, the form csrc/linear.hip's lin_dx reads -- the projection's input
                # gradient runs on that kernel: round 5; a three-plane build of linear.hip keeps the three-plane image)
                from samble_amd import _lib as lib_, linear as L_
                if lib_.query("samble_linear_two_plane_build"):
                    w_tr = L_.weight_images(w, want_rm=False)[1]
                else:
                    _, w_tr = o_.stage_tri_split(w.unsqueeze(0), want_rm=False, want_tr=True)
                assert torch.equal(imgs2[5], w_tr)
            for j, (a, b2) in enumerate(zip(imgs, imgs2)):
                assert (_k_image_live_equal(a, b2) if j in (1, 4) else torch.equal(a, b2)), (want, "image", j, int((a != b2).sum()))
            # SAMBLE_PROJ_ROWS_Q_ONLY: the same images, the Q columns, the token rows and a ragged last tile's rows; the K / V
            # columns of the full point tiles are not written at all (the poison planted below survives there)
            import samble_amd.ops as _o
            empty = torch.empty
            try:
                _o.torch.empty = lambda *a_, **k_: empty(*a_, **k_).fill_(-7.0) if k_.get("dtype") == torch.float32 else empty(*a_, **k_)
                qkv3, imgs3 = o_.stage_proj_fwd(x, tokens, w, images=want, q_only=True)
            finally:
                _o.torch.empty = empty
            torch.cuda.synchronize()
            nfull = (N // 32) * 32
            assert torch.equal(qkv3[:, :, :128][:, :N], qkv[:, :N, :128]) and torch.equal(qkv3[:, N:], qkv[:, N:])
            assert torch.equal(qkv3[:, nfull:N], qkv[:, nfull:N])
            if nfull > 1:  # (row N-1 may be written in full by the waves past the end: they recompute that row)
                assert bool((qkv3[:, :nfull - 1, 128:] == -7.0).all())
            for j, (a, b2) in enumerate(zip(imgs, imgs3)):
                assert (_k_image_live_equal(a, b2) if j in (1, 4) else torch.equal(a, b2)), (want, "q_only image", j, int((a != b2).sum()))
    finally:
        o_.MATRIX_MODE = old


def test_projection_reads_three_weight_tensors_where_they_are():
    """(Wq, Wk, Wv) as three tensors of their own (the reference's q_conv / k_conv / v_conv weights): the same bytes in qkv,
    in every image and in the backward's three results as with the concatenated (3C, C) block."""
    o_ = ops()
    old = o_.MATRIX_MODE
    o_.MATRIX_MODE = "tri"
    try:
        B, N, nt = 3, 1000, 6
        x = torch.from_numpy(synth.normal((B, 128, N), 1900)).to(DEV)
        tokens = torch.from_numpy(synth.normal((128, nt), 1901)).to(DEV)
        w = (torch.from_numpy(synth.normal((384, 128), 1902)) * 0.1).to(DEV)
        # three allocations in another order than [q, k, v], with padding between them
        w3 = tuple(t.clone() for t in (w[256:], w[:128], w[128:256]))
        w3 = (w3[1], w3[2], w3[0])
        for want in ("fwd+bwd", "fwd", ""):
            a = o_.stage_proj_fwd(x, tokens, w, images=want)
            b2 = o_.stage_proj_fwd(x, tokens, w3, images=want)
            if not want:
                assert torch.equal(a, b2)
                continue
            assert torch.equal(a[0], b2[0])
            for j, (p, q2) in enumerate(zip(a[1], b2[1])):  # (the K / V row images keep unwritten bytes in their dead plane)
                assert (_k_image_live_equal(p, q2) if j in (1, 4) else torch.equal(p, q2)), (want, j)
        qkv, imgs = o_.stage_proj_fwd(x, tokens, w3, images="fwd+bwd")
        dqkv = torch.from_numpy(synth.normal(tuple(qkv.shape), 1903)).to(DEV)
        ref = o_.stage_proj_bwd(dqkv, x, tokens, w, True, True, w_tr=imgs[5])
        for kw in ({"w_tr": imgs[5]}, {}):  # ({}: no image at hand -> concatenated on the host side)
            got = o_.stage_proj_bwd(dqkv, x, tokens, w3, True, True, **kw)
            assert all(torch.equal(p, q) for p, q in zip(ref, got))
        with pytest.raises(ValueError):
            o_.stage_proj_fwd(x, tokens, (w3[0], w3[1]))
        lib = o_._lib
        ws = torch.empty(lib.query("samble_proj_bwd_tri_workspace_bytes", B, N), dtype=torch.uint8, device=DEV)
        with pytest.raises(lib.SambleError):  # three tensors without the transposed image
            lib.call("samble_proj_bwd_tri_f32", dqkv.data_ptr(), dqkv.stride(0), dqkv.stride(1), x.data_ptr(), 128 * N, B, 128,
                     N, tokens.data_ptr(), nt, w3[0].data_ptr(), w3[1].data_ptr(), w3[2].data_ptr(), None, None, 128 * N, None,
                     None, None, ws.data_ptr(), ws.numel(), None)
        # dx_residual (round 5): the kernel's epilogue adds a tensor of dx's layout -- bitwise the separate torch add
        res = torch.from_numpy(synth.normal((B, 128, N), 1904)).to(DEV)
        plus = o_.stage_proj_bwd(dqkv, x, tokens, w, True, True, dx_residual=res)
        assert torch.equal(plus[0], res + ref[0]) and torch.equal(plus[1], ref[1]) and torch.equal(plus[2], ref[2])
    finally:
        o_.MATRIX_MODE = old


def test_nn_prepare_clears_the_score_workspace_on_request():
    """samble_nn_prepare's clear range (aligned: in the kernel; odd: a memset) and the statistics pass told that the
    workspace is clean: the same workspace bytes and outputs as the pass that zeroes it itself."""
    o_ = ops()
    B, N, nt, K = 3, 1000, 6, 32
    q, k, _ = (t.to(DEV) for t in _qkv(B, N, nt, 2100))
    g = torch.Generator().manual_seed(21)
    nn_idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:K] for _ in range(N)]) for _ in range(B)]).int().to(DEV)
    qimg = o_.stage_tri_split(q)[0]
    kimg = o_.stage_k_logit_form(o_.stage_tri_split(k)[0], k)
    ref_sorted, ref_masks = o_.stage_nn_prepare(nn_idx)
    ref = o_.stage_attn_stats_nl(qimg, kimg, ref_masks, B, N, nt, K, want_nl=False, score=(ref_sorted, "sparse_col_sqr", 6))
    for odd in (0, 5):
        ws = o_.score_workspace(B, N, 6, DEV)
        ws.fill_(0xA5)
        target = ws[odd:] if odd else ws
        got_sorted, got_masks = o_.stage_nn_prepare(nn_idx, clear=target)
        torch.cuda.synchronize()
        assert torch.equal(got_sorted, ref_sorted) and torch.equal(got_masks, ref_masks)
        assert bool((target == 0).all()) and (not odd or bool((ws[:odd] == 0xA5).all()))
    ws = o_.score_workspace(B, N, 6, DEV).fill_(0xA5)
    s2, m2 = o_.stage_nn_prepare(nn_idx, clear=ws)
    got = o_.stage_attn_stats_nl(qimg, kimg, m2, B, N, nt, K, want_nl=False, score=(s2, "sparse_col_sqr", 6), cleared_ws=ws)
    assert got[3] is ws
    assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3])
    with pytest.raises(ValueError):
        o_.stage_attn_stats_nl(qimg, kimg, m2, B, N, nt, K, want_nl=False, score=(s2, "sparse_col_sqr", 6), cleared_ws=ws[16:])
    # K = 16 lists as well
    nn16 = nn_idx[:, :, :16].contiguous()
    ref16 = o_.stage_nn_prepare(nn16)
    ws16 = o_.score_workspace(B, N, None, DEV).fill_(0x5A)
    got16 = o_.stage_nn_prepare(nn16, clear=ws16)
    assert torch.equal(got16[0], ref16[0]) and torch.equal(got16[1], ref16[1]) and bool((ws16 == 0).all())


@pytest.mark.parametrize("mode", ["uniform", "random"])
def test_bin_select_draws_its_own_noise_like_the_injected_one(mode):
    """The seeded select (Exp(1) drawn inside the kernel from a Philox state) returns, bit for bit, the indices of the
    select on the tensor samble_exp1_noise_f32 writes for the same state -- so the oracle comparison of the
    injected-noise entry covers it -- and that tensor is an Exp(1) sample: positive, finite, the right moments, a
    Kolmogorov-Smirnov distance from 1 - exp(-x) as small as a sample of its size gives, other states other numbers."""
    from scipy import stats
    o_ = ops()
    B, N, nb, M = 8, 2048, 6, 1024
    score = torch.from_numpy(np.abs(synth.normal((B, 1, N), 41)) * 1e-4 + 1e-6)
    z = O.zscore(score)
    state = O.blend_boundaries(None, O.batch_quantiles(z.reshape(B, 1, N, 1), nb), nb, 0.99)
    member = O.bin_membership(z, state)
    counts = O.allocate_counts(torch.rand(B, nb, generator=torch.Generator().manual_seed(6)), member.squeeze(1).sum(1), M)
    bits = (member.squeeze(1).long() * (1 << torch.arange(nb))).sum(-1).to(torch.uint8)
    args = (score.reshape(B, N).to(DEV), z.reshape(B, N).to(DEV), bits.to(DEV), counts.to(DEV), M, mode, 0.1)
    seed, offset = 0x1234_5678_9ABC_DEF1, 4 * 123_456_789_012
    noise = o_.stage_exp1_noise(seed, offset, B * nb, N, DEV)
    got = o_.stage_bin_select(*args, philox=(seed, offset))
    assert torch.equal(got, o_.stage_bin_select(*args, noise=noise))
    ref = O.select_indices(score, member, counts, M, mode, 0.1, noise.cpu()).reshape(B, M)
    assert int((got.cpu() != ref).sum()) <= (8 if mode == "random" else 0)   # (random: exp / tanh near-ties, see above)
    v = noise.double().cpu().numpy().ravel()
    assert np.isfinite(v).all() and (v > 0).all()
    assert abs(v.mean() - 1.0) < 0.02 and abs(v.var() - 1.0) < 0.05
    assert stats.kstest(v, "expon").statistic < 0.01
    assert abs(np.corrcoef(v[:-1], v[1:])[0, 1]) < 0.02
    other = o_.stage_exp1_noise(seed, offset + 4, B * nb, N, DEV)
    assert not bool((other == noise).any()) or float((other == noise).float().mean()) < 1e-3
    assert torch.equal(noise, o_.stage_exp1_noise(seed, offset, B * nb, N, DEV))
    # the module-level default: torch's device generator keys the draw, and moves on
    torch.manual_seed(77)
    a = o_.stage_bin_select(*args)
    b2 = o_.stage_bin_select(*args)
    torch.manual_seed(77)
    assert torch.equal(a, o_.stage_bin_select(*args)) and not torch.equal(a, b2)
    with pytest.raises(o_._lib.SambleError):
        o_.stage_bin_select(*args, philox=(seed, 3))


def test_rows_backward_clears_the_token_rows_of_dq_on_request():
    """SAMBLE_BWD_DQ_TOKEN_ROWS: dQ handed over as the first N rows of the projection's (B, N + nt, .) gradient block --
    the nt rows behind them come back as zeros (they were NaN), the N rows and dK / dV as without the flag."""
    o_ = ops()
    old = o_.MATRIX_MODE
    o_.MATRIX_MODE = "tri"
    try:
        B, N, nt, M, D = 2, 1000, 6, 333, 128
        q, k, v = (t.to(DEV) for t in _qkv(B, N, nt, 2300))
        gr = torch.from_numpy(synth.normal((B, D, M), 8)).to(DEV)
        idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b))[:M] for b in range(B)]).to(DEV)
        smap, lse, _ = o_.stage_attn_stats(q, k, N, nt)
        x_ds = o_.stage_attn_rows(smap, lse, v, idx, N, nt)
        ref = [torch.full((B, n, D), float("nan"), device=DEV) for n in (N, N + nt, N + nt)]
        o_.stage_attn_rows_bwd(q, k, v, smap, lse, x_ds, idx, gr, N, nt, *ref)
        block = torch.full((B, N + nt, 3 * D), float("nan"), device=DEV)
        o_.stage_attn_rows_bwd(q, k, v, smap, lse, x_ds, idx, gr, N, nt, block[:, :N, :D], block[:, :, D:2 * D],
                               block[:, :, 2 * D:], dq_token_rows=True)
        assert torch.equal(block[:, :N, :D], ref[0]) and torch.equal(block[:, :, D:2 * D], ref[1])
        assert torch.equal(block[:, :, 2 * D:], ref[2])
        assert bool((block[:, N:, :D] == 0).all())
        o_.MATRIX_MODE = "f32"
        with pytest.raises(ValueError):
            o_.stage_attn_rows_bwd(q, k, v, smap, lse, x_ds, idx, gr, N, nt, block[:, :N, :D], block[:, :, D:2 * D],
                                   block[:, :, 2 * D:], dq_token_rows=True)
    finally:
        o_.MATRIX_MODE = old


def test_small_ops_against_reference_vectors():
    """norm_range, sort_chunk, l2_global, fps (reference utils/ops.py:148-171, 239-259, 115-122, 646-692) against vectors
    from the unmodified reference (tests/golden/make_golden_ops.py)."""
    import os
    from tests.util import GOLDEN_DIR
    d = np.load(os.path.join(GOLDEN_DIR, "layer_ops_small.npz"))
    B, H, N, D, nb, seed = [int(v) for v in d["meta"]]
    o_ = ops()
    score = torch.from_numpy(synth.normal((B, H, N), seed)) * 0.7 + 0.1
    score[0, 0, 17] = score[0, 0, 400]
    sg = score.to(DEV)
    for mode in ("minmax", "sigmoid", "tanh", "z-score"):
        got = o_.norm_range(sg, dim=-1, n_min=0.25, n_max=2.0, mode=mode)
        torch.testing.assert_close(got.cpu(), torch.from_numpy(d["norm_" + mode.replace("-", "")]), rtol=2e-6, atol=2e-6)
    with pytest.raises(ValueError):
        o_.norm_range(sg, mode="bogus")
    for tag, desc in (("asc", False), ("desc", True)):
        xs, ids = o_.sort_chunk(sg, nb, dim=-1, descending=desc)
        assert [t.shape[-1] for t in xs] == d[f"chunk_sizes_{tag}"].tolist() and len(ids) == nb
        # values are exact (a sort moves values, it does not compute); indices agree except inside the exact tie
        assert torch.equal(torch.cat(xs, dim=-1).cpu(), torch.from_numpy(d[f"sorted_{tag}"]))
        order = torch.cat(ids, dim=-1).cpu()
        ref = torch.from_numpy(d[f"order_{tag}"])
        assert order.dtype == torch.int64 and int((order != ref).sum()) <= 2
        assert torch.equal(torch.gather(score, -1, order), torch.from_numpy(d[f"sorted_{tag}"]))
    q = torch.from_numpy(synth.normal((B, H, 40, D), seed + 1)).to(DEV)
    k = torch.from_numpy(synth.normal((B, H, D, 40), seed + 2)).to(DEV)
    torch.testing.assert_close(o_.l2_global(q, k).cpu(), torch.from_numpy(d["l2_global"]), rtol=1e-5, atol=1e-5)
    Bf, Nf, Cf, npnt, sd = [int(v) for v in d["fps_meta"]]
    xyz = torch.from_numpy(synth.xyz_clouds(Bf, Nf, sd + 3)).to(DEV)
    x = torch.from_numpy(synth.normal((Bf, Cf, Nf), sd + 4)).to(DEV)
    idx = o_.farthest_point_sample(xyz.permute(0, 2, 1), npnt, torch.from_numpy(d["fps_start"]).to(DEV))
    assert torch.equal(idx.cpu(), torch.from_numpy(d["fps_idx"])[:, 0])
    assert torch.equal(o_.index_points_for_fps(x.permute(0, 2, 1), idx).permute(0, 2, 1).cpu(), torch.from_numpy(d["fps_x"]))
    (xf, idf), rest = o_.fps(x, xyz, npnt)      # (its own torch.randint start: shapes and the gather relation)
    assert rest == (None, None) and idf.shape == (Bf, 1, npnt) and xf.shape == (Bf, Cf, npnt)
    assert torch.equal(xf, torch.gather(x, 2, idf.expand(-1, Cf, -1)))


def test_k_logit_form_is_idempotent():
    """samble_tri_k_logit_form twice = once (include/samble.h: converted tiles carry a tag), on full, ragged and
    token-bearing row counts."""
    o_ = ops()
    for (B, R) in ((2, 256), (1, 70), (3, 262)):
        rows = torch.from_numpy(synth.normal((B, R, 128), 900 + R)).to(DEV)
        img = o_.stage_tri_split(rows)[0]
        raw = img.clone()
        o_.stage_k_logit_form(img, rows)
        once = img.clone()
        assert not torch.equal(once, raw)
        o_.stage_k_logit_form(img, rows)
        assert torch.equal(img, once)

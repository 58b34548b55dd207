"""End-to-end parity of the drop-in DownSampleToken on the GPU against the golden fixtures that the
reference produced (tests/golden/*.npz) and against the CPU oracle, plus size-independent
properties at the metric size (B=32, N=2048 -> 1024)."""
import os

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from samble_amd import synth
from tests.util import Golden, golden_names, set_agreement

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

@pytest.fixture(params=["tri", "f32"])
def matrix_mode(request):
    """Both matrix-instruction families (split-bf16 default, fp32 MFMA) behind the same module."""
    from samble_amd import ops
    old = ops.MATRIX_MODE
    ops.MATRIX_MODE = request.param
    yield request.param
    ops.MATRIX_MODE = old


def _pinned_identity(mode, name, call):
    """Per-cloud identity (1 = sampled indices equal the reference's) measured on MI355X and committed:
    tests/expected_identity.json, written by tools/fixture_identity.py."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "expected_identity.json")
    if not os.path.exists(path):
        return None
    table = json.load(open(path))
    rows = table.get(f"{mode}/{name}")
    return rows[call]["same"] if rows else None


@pytest.mark.parametrize("name", golden_names())
def test_module_against_reference_fixture(name, matrix_mode):
    g = Golden(name)
    mod = g.module(DEV)
    for call in range(g.calls):
        last = call == g.calls - 1
        x = g.x(call).to(DEV).requires_grad_(last)
        noise = None if g.sample_mode == "topk" else g.t("noise", call).to(DEV)
        (x_ds, idx), (d0, d1) = mod(x, noise=noise)
        assert d0 is None and d1 is None
        assert idx.shape == (g.B, 1, g.M) and idx.dtype == torch.int64 and x_ds.shape == (g.B, g.C, g.M)
        # fp tolerance on scores (fp32 MFMA vs MKL summation order)
        score_ref = g.t("score", call)
        torch.testing.assert_close(mod.attention_point_score.cpu(), score_ref, rtol=3e-5, atol=1e-9)
        torch.testing.assert_close(mod.attention_bins_beforesoftmax.cpu(), g.t("tok_logits", call), rtol=1e-4,
                                   atol=2e-5)
        if g.idx_mode.startswith("sparse"):
            assert set_agreement(mod.knn_idx.cpu(), g.t("knn_sorted", call).long()) >= 0.9995
        torch.testing.assert_close(mod.bin_boundaries[0].cpu(), g.t("upper", call), rtol=1e-4, atol=1e-5)
        # sampled indices, end to end.  Selection is a discontinuous function of fp32 values and the
        # reference's count allocation is ill-conditioned by construction: when all bins but one
        # saturate, the float water-filling (ops.py:403-424) lands on an exact integer and `.int()`
        # truncates it up or down on a 1e-9 difference of a bin weight (measured: |dw| <= 7e-9 flips a
        # cloud).  MKL/Sleef vs MFMA/ocml arithmetic differ by more than that.  Given the reference's own stage
        # inputs every integer is exact (test_select_stages_exact_on_golden); END TO END the measured
        # per-cloud identities are pinned in tests/expected_identity.json (tools/fixture_identity.py): a cloud
        # that is identical there must stay identical, and a cloud that is not must be explained by one of the
        # two measured mechanisms below.
        same = (idx.cpu()[:, 0] == g.t("idx", call)[:, 0]).all(1)
        pinned = _pinned_identity(matrix_mode, name, call)
        assert pinned is not None, (f"{matrix_mode}/{name} has no row in tests/expected_identity.json: run "
                                    "tools/fixture_identity.py on the GPU box and commit its output")
        lost = [b for b in range(g.B) if pinned[b] and not bool(same[b])]
        assert not lost, f"clouds {lost} were identical to the reference when the table was pinned"
        counts_same = (mod.k_point_to_choose.cpu() == g.t("counts", call)).all(1)
        for b in range(g.B):
            if bool(same[b]):
                continue
            got_b, ref_b = idx.cpu()[b, 0], g.t("idx", call)[b, 0]
            if not bool(counts_same[b]):
                # (1) one pick moved between two bins: the truncation flip of the water-filling.  Evidence per cloud:
                # the integer stage is exact on both sides -- the reference's water-filling (oracle) run on OUR bin
                # weights gives OUR counts, run on the fixture's weights it gives the fixture's -- and the weights
                # themselves agree to 2e-6
                dc = (mod.k_point_to_choose.cpu()[b].long() - g.t("counts", call)[b].long())
                assert int(dc.abs().sum()) == 2 and int(dc.sum()) == 0, dc.tolist()
                torch.testing.assert_close(mod.bin_weights_beforerelu.cpu()[b], g.t("w_pre", call)[b], rtol=2e-6,
                                           atol=1e-7)
                cap = g.t("cap", call).long()
                assert torch.equal(mod.max_num_points.cpu().long(), cap), "bin populations"
                w_ours = torch.relu(mod.bin_weights_beforerelu.cpu()) if g.relu_mean_order == "mean_relu" else \
                    mod.bin_weights_beforerelu.cpu()
                w_ref = torch.relu(g.t("w_pre", call)) if g.relu_mean_order == "mean_relu" else g.t("w_pre", call)
                if g.relu_mean_order == "mean_relu":
                    assert torch.equal(O.allocate_counts(w_ours.clone(), cap, g.M)[b], mod.k_point_to_choose.cpu()[b])
                    assert torch.equal(O.allocate_counts(w_ref.clone(), cap, g.M)[b], g.t("counts", call)[b].int())
                assert len(set(got_b.tolist()) ^ set(ref_b.tolist())) <= 2
            else:
                # (2) same counts: a near-tie of two selection keys (or of two scores at a bin boundary) resolved
                # the other way -- at most a couple of points change place or membership
                assert len(set(got_b.tolist()) ^ set(ref_b.tolist())) <= 4, (name, b)
        assert int(same.sum()) >= g.B - max(1, g.B // 4), "sampled indices differ from the reference"
        assert set_agreement(idx.cpu()[:, 0], g.t("idx", call)[:, 0]) >= 0.99
        if g.has("x_ds", call):
            torch.testing.assert_close(x_ds.detach().cpu()[same], g.t("x_ds", call)[same], rtol=1e-4, atol=2e-5)
        if not last:
            continue
        # gradients.  dx is separable per cloud: always compared on the clouds whose indices are the reference's; the
        # parameter gradients sum over the batch, so they are compared when every cloud is identical AND, on every
        # fixture, in a second pass that gathers the reference's own indices (forced_idx: tests/golden/make_golden.py
        # forms the reference gradients the same way, through its idx)
        up = g.upstream().to(DEV)
        x_ds.backward(up)

        def check_params(tag):
            for got, key in ((mod.q_conv.weight.grad, "dwq"), (mod.k_conv.weight.grad, "dwk"),
                             (mod.v_conv.weight.grad, "dwv"), (mod.bin_tokens.grad, "dtokens")):
                if not g.has(key, call):
                    continue
                ref = g.t(key, call)
                err = (got.cpu() - ref).abs().max().item()
                assert err <= 2e-4 * ref.abs().max().item() + 1e-6, (tag, key, err)

        def check_dx(grad, rows, tag):
            gd = grad.cpu().double()
            if g.has("dx", call) and bool(rows.any()):
                ref = g.t("dx", call)
                err = (grad.cpu()[rows] - ref[rows]).abs().max().item()
                assert err <= 2e-4 * ref.abs().max().item() + 1e-6, (tag, "dx", err)
            if g.has("dx_cloud_sums", call) and bool(rows.any()):
                # every fixture: per-cloud (sum, sum of squares) of dx in float64 -- the big fixture stores no full dx
                ref = g.t("dx_cloud_sums", call)[rows]
                got = torch.stack([gd.sum((1, 2)), gd.square().sum((1, 2))], dim=1)[rows]
                n = gd[0].numel()
                assert bool(((got[:, 1] - ref[:, 1]).abs() <= 1e-3 * ref[:, 1]).all()), (tag, "dx sum of squares")
                assert bool(((got[:, 0] - ref[:, 0]).abs() <= 2e-4 * (n * ref[:, 1]).sqrt() + 1e-6).all()), (tag, "dx sum")

        check_dx(x.grad, same, "own idx")
        if bool(same.all()):
            check_params("own idx")
        mod.zero_grad()
        x2 = g.x(call).to(DEV).requires_grad_(True)
        (x_ds2, idx2), _ = mod(x2, noise=noise, forced_idx=g.t("idx", call).to(DEV))
        assert torch.equal(idx2.cpu(), g.t("idx", call))
        if g.has("x_ds", call):
            torch.testing.assert_close(x_ds2.detach().cpu(), g.t("x_ds", call), rtol=1e-4, atol=2e-5)
        x_ds2.backward(up)
        check_dx(x2.grad, torch.ones(g.B, dtype=torch.bool), "reference idx")
        check_params("reference idx")


def test_state_dict_keys_and_forward_contract():
    g = Golden("cls_random_dyn")
    mod = g.module(DEV)
    assert sorted(mod.state_dict().keys()) == ["bin_tokens", "k_conv.weight", "q_conv.weight", "v_conv.weight"]
    assert mod.state_dict()["bin_tokens"].shape == (1, 128, 6)
    assert mod.state_dict()["q_conv.weight"].shape == (128, 128, 1)
    assert mod.bin_boundaries is None
    (x_ds, idx), _ = mod(g.x().to(DEV))
    up, lo = mod.bin_boundaries
    assert up.shape == (1, 1, 1, 6) and torch.isinf(up[0, 0, 0, 0]) and torch.isinf(lo[0, 0, 0, -1])
    assert mod.bin_points_mask.shape == (g.B, 1, g.N, 6) and mod.bin_points_mask.dtype == torch.bool
    mod.output_variable_calculatio()
    assert len(mod.idx_chunks) == 6 and len(mod.idx_chunks[0]) == g.B
    assert mod.output_variables("idx", "bin_prob")[0] is mod.idx
    # a15 against the reference's fixture: chunk (t, b) = the points of cloud b in bin t, ascending
    # (models/downsample.py:346-362: torch.where on the mask), bin_prob = the weights before relu
    bin_id = g.t("bin_id").long()
    for t in range(6):
        for b in range(g.B):
            want = torch.where(bin_id[b] == t)[0].reshape(1, -1)
            assert torch.equal(mod.idx_chunks[t][b].cpu(), want), (t, b)
    assert mod.bin_prob is mod.bin_weights_beforerelu
    torch.testing.assert_close(mod.bin_prob.detach().cpu().reshape(g.B, 6), g.t("w_pre").reshape(g.B, 6), rtol=1e-4,
                               atol=1e-6)


def test_metric_size_properties_and_determinism():
    """B=32, N=2048 -> M=1024 (BASELINE.json configs[1] layer 0): the oracle takes seconds here, so
    check properties: indices unique and in range, per-bin counts honoured, bin order, run-to-run
    identical, and agreement with the oracle's sampled sets."""
    from samble_amd import sampler_config
    from samble_amd.downsample import DownSampleToken
    B, C, N, M, nb = 32, 128, 2048, 1024, 6
    torch.manual_seed(0)
    mod = DownSampleToken(sampler_config("cls"), 0).to(DEV)
    x = torch.from_numpy(synth.features(B, C, N, 2001)).to(DEV)
    noise = torch.from_numpy(synth.exp1((B * nb, N), 2002)).to(DEV)
    (x_ds, idx), _ = mod(x, noise=noise)
    i = idx[:, 0].cpu()
    assert int(i.min()) >= 0 and int(i.max()) < N
    assert all(len(set(r.tolist())) == M for r in i), "sampling is without replacement"
    counts = mod.k_point_to_choose.cpu()
    assert bool((counts.sum(1) == M).all()) and bool((counts <= mod.max_num_points.cpu()).all())
    bits = mod._member_bits.cpu().long()
    bin_of = torch.log2(bits.float()).long()
    for b in range(0, B, 7):
        picked_bins = bin_of[b][i[b]]
        assert bool((picked_bins[1:] >= picked_bins[:-1]).all()), "bins are emitted in ascending order"
        assert torch.equal(torch.bincount(picked_bins, minlength=nb), counts[b].long())
    # second module, same weights and inputs, fresh boundary state -> identical outputs
    mod2 = DownSampleToken(sampler_config("cls"), 0).to(DEV)
    mod2.load_state_dict(mod.state_dict())
    (x_ds2, idx2), _ = mod2(x, noise=noise)
    assert torch.equal(idx2, idx) and torch.equal(x_ds2, x_ds)
    # the oracle on ALL 32 clouds (dense N x N on the CPU, a few seconds) under the GPU's boundaries: the end-to-end index
    # match rate at the headline configuration, asserted and printed ("stage-wise bit-exactness plus a
    # reported end-to-end index match rate"; reference utils/ops.py:467-619)
    spec = O.SamplerSpec(M=M, K=32, C=C, num_bins=nb, dynamic_boundaries=False)
    st = O.SamplerState(mod.q_conv.weight.detach().cpu(), mod.k_conv.weight.detach().cpu(),
                        mod.v_conv.weight.detach().cpu(), mod.bin_tokens.detach().cpu(),
                        [t.cpu().clone() for t in mod.bin_boundaries])
    x_ref, idx_ref = O.sampler_forward(spec, st, x.cpu(), noise.cpu())
    agree = set_agreement(idx[:, 0].cpu(), idx_ref[:, 0])
    same_rows = (idx[:, 0].cpu() == idx_ref[:, 0]).all(1)
    pos = float((idx[:, 0].cpu() == idx_ref[:, 0]).float().mean())
    print(f"\nmetric size B={B} N={N}->{M}: clouds with the oracle's exact index tensor {int(same_rows.sum())} of {B}; "
          f"positions identical {pos:.5f}; sampled-set agreement {agree:.5f}")
    assert agree >= 0.995, agree
    # measured: 28-29 of 32 on every box of rounds 3-5 (the three that differ: one truncation flip of the float
    # water-filling or one near-tie of two selection keys each, asserted below); floor = measured - 2
    assert int(same_rows.sum()) >= 26, int(same_rows.sum())
    for b in range(B):   # a cloud that differs does so by a handful of points (near-ties of keys / one count flip)
        assert len(set(idx[b, 0].tolist()) ^ set(idx_ref[b, 0].tolist())) <= 8, b
    torch.testing.assert_close(x_ds.cpu()[same_rows], x_ref[same_rows], rtol=1e-4, atol=2e-5)


def test_stress_size_properties():
    """BASELINE.json configs[4] at full size: B=16 clouds, N=8192 -> 4096 (the 16-cloud batch changes the XCD
    placement and allocates the whole multi-GB map the way the bench's stress workload does).
    Size-independent properties + agreement of the two-pass path with the single-pass flash path."""
    import samble_amd.downsample as D
    from samble_amd import sampler_config
    B, C, N, M, nb = 16, 128, 8192, 4096, 6
    cfg = sampler_config("cls", M=[M, M // 2])
    mod = D.DownSampleToken(cfg, 0).to(DEV)
    x = torch.from_numpy(synth.features(B, C, N, 5001)).to(DEV).requires_grad_(True)
    noise = torch.from_numpy(synth.exp1((B * nb, N), 5002)).to(DEV)
    g = torch.from_numpy(synth.normal((B, C, M), 5003)).to(DEV)
    (x_ds, idx), _ = mod(x, noise=noise)
    x_ds.backward(g)
    i = idx[:, 0].cpu()
    assert x_ds.shape == (B, C, M) and int(i.min()) >= 0 and int(i.max()) < N
    assert all(len(set(r.tolist())) == M for r in i)
    counts = mod.k_point_to_choose.cpu()
    assert bool((counts.sum(1) == M).all()) and bool((counts <= mod.max_num_points.cpu()).all())
    assert torch.isfinite(x_ds).all() and torch.isfinite(x.grad).all()
    grads2 = [p.grad.clone() for p in mod.parameters()]
    dx2 = x.grad.clone()
    # same weights / inputs / boundaries through the single-pass kernels
    mod1 = D.DownSampleToken(cfg, 0).to(DEV)
    mod1.load_state_dict(mod.state_dict())
    x1 = x.detach().clone().requires_grad_(True)
    D.TWO_PASS = False
    try:
        (x_ds1, idx1), _ = mod1(x1, noise=noise)
        x_ds1.backward(g)
    finally:
        D.TWO_PASS = True
    # the two paths round the logits differently (split-bf16 products vs one fp32 MFMA chain): a sampled
    # index may flip where two scores tie to the last bits, and a cloud's per-bin counts may move by one pick
    # where the float water-filling lands on an integer (reference utils/ops.py:403-424) -- which shifts every
    # later position of that cloud.  So: sets nearly identical everywhere; positions compared on the clouds whose
    # counts agree (most of them)
    assert set_agreement(idx1[:, 0].cpu(), idx[:, 0].cpu()) >= 0.999
    same_counts = (mod1.k_point_to_choose == mod.k_point_to_choose).all(1)
    assert int(same_counts.sum()) >= (3 * B) // 4, int(same_counts.sum())
    same = idx1[:, 0] == idx[:, 0]                     # (B, M) positions holding the same point
    assert float(same[same_counts].float().mean()) >= 0.99
    keep = same[:, None, :].expand_as(x_ds)
    torch.testing.assert_close(x_ds1[keep], x_ds[keep], rtol=1e-4, atol=2e-5)
    # dx is separable per cloud: compared on every cloud whose index tensor agrees; the parameter gradients (sums over
    # the batch) through a second backward of the single-pass path on the two-pass path's indices
    rows = same.all(1)
    assert int(rows.sum()) >= B // 2
    scale = float(dx2.abs().max())
    assert float((x1.grad[rows] - dx2[rows]).abs().max()) <= 2e-4 * scale
    mod1.zero_grad()
    x3 = x.detach().clone().requires_grad_(True)
    D.TWO_PASS = False
    try:
        (x_ds3, idx3), _ = mod1(x3, noise=noise, forced_idx=idx)
        x_ds3.backward(g)
    finally:
        D.TWO_PASS = True
    assert torch.equal(idx3, idx)
    torch.testing.assert_close(x_ds3, x_ds, rtol=1e-4, atol=2e-5)
    assert float((x3.grad - dx2).abs().max()) <= 2e-4 * scale
    for a, b2 in zip(grads2, [p.grad for p in mod1.parameters()]):
        assert float((a - b2).abs().max()) <= 2e-4 * float(a.abs().max()) + 1e-7


def test_seg_preset_and_second_layer_shape():
    from samble_amd import sampler_config
    from samble_amd.downsample import DownSampleToken
    mod = DownSampleToken(sampler_config("seg"), 1).to(DEV)  # layer 1: 1024 -> 512, 4 bins
    x = torch.from_numpy(synth.features(4, 128, 1024, 31)).to(DEV).requires_grad_(True)
    (x_ds, idx), _ = mod(x)
    assert x_ds.shape == (4, 128, 512) and idx.shape == (4, 1, 512)
    x_ds.sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0


def test_token_logits_stay_differentiable():
    g = Golden("cls_random_dyn")
    mod = g.module(DEV)
    x = g.x().to(DEV).requires_grad_(True)
    mod(x, noise=g.t("noise").to(DEV))
    loss = (mod.attention_bins_beforesoftmax ** 2).sum()
    loss.backward()
    # reference value of d loss / d tokens from autograd on the oracle
    st = g.oracle_state()
    xr = g.x().requires_grad_(True)
    tok = st.tokens.clone().requires_grad_(True)
    q, k, v = O.project_qkv(xr, tok, st.wq, st.wk, st.wv)
    _, _, tl = O.attention_map(q, k, g.N)
    (tl ** 2).sum().backward()
    ref = tok.grad
    err = (mod.bin_tokens.grad.cpu() - ref).abs().max().item()
    assert err <= 1e-3 * ref.abs().max().item(), err


@pytest.mark.parametrize("name", ["layer_res_ff_topk", "layer_res_noff_random"])
def test_res_block_against_reference_fixture(name, matrix_mode):
    """res.enable (and res.ff): the residual link of reference models/downsample.py:75-83, 292-298 -- a gather of
    CHANNEL 0 of x at the sampled indices broadcast-added to x_ds, BatchNorm, optional FFN + BatchNorm --
    against a fixture from the unmodified reference (tests/golden/make_golden_res.py): output, every parameter
    gradient, dx and the BatchNorm buffers after one training step."""
    from samble_amd import sampler_config
    from samble_amd.downsample import DownSampleToken
    from tests.util import fill_parameters, layer_fixture
    d = layer_fixture(name)
    B, C, N, M, nb, seed = [int(v) for v in d["meta"]]
    cfg = sampler_config("cls", M=[M, M // 2])
    cfg.res.enable = [True, True]
    cfg.res.ff = [bool(d["ff"]), bool(d["ff"])]
    cfg.bin.sample_mode = [str(d["sample_mode"])] * 2
    mod = DownSampleToken(cfg, 0)
    fill_parameters(mod, seed)
    mod = mod.to(DEV).train()
    ref_names = sorted(k[len("grad__"):] for k in d.files if k.startswith("grad__"))
    assert sorted(n for n, _ in mod.named_parameters()) == ref_names, "state_dict keys of the residual link"
    x = torch.from_numpy(synth.features(B, C, N, seed + 10)).to(DEV).requires_grad_(True)
    (x_ds, idx), _ = mod(x, noise=torch.from_numpy(d["noise"]).to(DEV))
    assert torch.equal(idx.cpu(), torch.from_numpy(d["idx"])), "sampled indices"
    torch.testing.assert_close(x_ds.detach().cpu(), torch.from_numpy(d["x_ds"]), rtol=2e-4, atol=2e-5)
    x_ds.backward(torch.from_numpy(synth.normal((B, C, M), seed + 99)).to(DEV))
    for pname, p in mod.named_parameters():
        ref = torch.from_numpy(d["grad__" + pname])
        err = float((p.grad.cpu() - ref).abs().max())
        assert err <= 3e-4 * float(ref.abs().max()) + 1e-6, (pname, err)
    ref = torch.from_numpy(d["dx"])
    assert float((x.grad.cpu() - ref).abs().max()) <= 3e-4 * float(ref.abs().max()) + 1e-6
    for bname, b in mod.named_buffers():
        refb = torch.from_numpy(d["buf__" + bname])
        if refb.dtype.is_floating_point:
            torch.testing.assert_close(b.cpu(), refb, rtol=2e-4, atol=1e-6)
        else:
            assert torch.equal(b.cpu(), refb)


def _map_free_fixtures():
    """the fixtures the map-free forward serves: a sparse_* score mode with dot-product logits"""
    out = []
    for n in golden_names():
        g = Golden(n)
        # (a wider layer runs torch's matmuls and convolutions, DownSampleToken._forward_wide: the switch does not reach
        # it, and those libraries do not promise run-to-run identical bits)
        if g.idx_mode.startswith("sparse") and g.asm == "dot" and g.C <= 128:
            out.append(n)
    return out


@pytest.mark.parametrize("name", _map_free_fixtures())
def test_map_free_module_equals_map_module(name):
    """DownSampleToken with and without the logit map (downsample.MAP_FREE): same indices, same x_ds, same
    boundaries and same input / parameter gradients, bit for bit, on the reference fixtures' inputs."""
    import samble_amd.downsample as D
    from samble_amd import ops
    g = Golden(name)
    outs = []
    old_mode, old_free = ops.MATRIX_MODE, D.MAP_FREE
    try:
        ops.MATRIX_MODE = "tri"
        for free in (False, True):
            D.MAP_FREE = free
            mod = g.module(DEV)
            res = []
            for call in range(g.calls):
                x = g.x(call).to(DEV).requires_grad_(True)
                noise = None if g.sample_mode == "topk" else g.t("noise", call).to(DEV)
                (x_ds, idx), _ = mod(x, noise=noise)
                w = torch.linspace(-1, 1, x_ds.numel(), device=DEV).view_as(x_ds)
                mod.zero_grad()
                ((x_ds * w).sum() + mod.attention_bins_beforesoftmax.square().sum()).backward()
                res.append([x_ds.detach(), idx, mod.attention_point_score, mod.bin_boundaries[0].clone(), x.grad] +
                           [p.grad.clone() for p in mod.parameters() if p.grad is not None])
            outs.append(res)
    finally:
        ops.MATRIX_MODE, D.MAP_FREE = old_mode, old_free
    for call, (a, b) in enumerate(zip(*outs)):
        assert len(a) == len(b) and len(a) > 6
        for j, (t1, t2) in enumerate(zip(a, b)):
            assert torch.equal(t1, t2), (call, j)


@pytest.mark.parametrize("name", ["cls_random_dyn", "seg_random_dyn", "cls_random_static", "cls_random_cfg1"])
def test_one_launch_chain_module_equals_two_launch_module(name):
    """downsample.FUSED_CHAIN: the integer tail as one launch (single rank) against the two launches a process group
    needs -- same indices, outputs, boundaries and gradients, bit for bit, over the fixtures' calls."""
    import samble_amd.downsample as D
    g = Golden(name)
    outs = []
    old = D.FUSED_CHAIN
    try:
        for fused in (False, True):
            D.FUSED_CHAIN = fused
            mod = g.module(DEV)
            res = []
            for call in range(g.calls):
                x = g.x(call).to(DEV).requires_grad_(True)
                (x_ds, idx), _ = mod(x, noise=g.t("noise", call).to(DEV))
                x_ds.square().sum().backward()
                res.append([x_ds.detach(), idx, mod.attention_point_score, mod.normalized_score, mod.bin_boundaries[0].clone(),
                            mod.bin_boundaries[1].clone(), mod.k_point_to_choose, mod._member_bits, x.grad,
                            mod.bin_tokens.grad.clone()])
                mod.zero_grad()
            outs.append(res)
    finally:
        D.FUSED_CHAIN = old
    for call, (a, b) in enumerate(zip(*outs)):
        for j, (t1, t2) in enumerate(zip(a, b)):
            assert torch.equal(t1, t2), (call, j)


def test_map_free_long_cloud_takes_the_compact_score_route():
    """N = 8500 > 8192: pass 1 cannot hold the score accumulators in LDS, so the module keeps the K neighbour logits per
    row and runs the separate score pass on them -- still without the N x (N+nt) map, same outputs as with the map."""
    import samble_amd.downsample as D
    from samble_amd import ops, sampler_config
    from samble_amd.downsample import DownSampleToken
    B, C, N, M, nb = 1, 128, 8500, 1000, 6
    x = torch.from_numpy(synth.features(B, C, N, 31)).to(DEV)
    noise = torch.from_numpy(synth.exp1((B * nb, N), 32)).to(DEV)
    outs = []
    old_mode, old_free = ops.MATRIX_MODE, D.MAP_FREE
    try:
        ops.MATRIX_MODE = "tri"
        for free in (False, True):
            D.MAP_FREE = free
            torch.manual_seed(3)
            mod = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0).to(DEV)
            xin = x.clone().requires_grad_(True)
            (x_ds, idx), _ = mod(xin, noise=noise)
            x_ds.sum().backward()
            outs.append((x_ds.detach(), idx, mod.attention_point_score, xin.grad, mod.q_conv.weight.grad))
    finally:
        ops.MATRIX_MODE, D.MAP_FREE = old_mode, old_free
    for j, (t1, t2) in enumerate(zip(*outs)):
        assert torch.equal(t1, t2), j


ODD_SHAPES = [(1, 32, 1, 6, "random"), (1, 33, 32, 4, "topk"), (2, 40, 39, 2, "uniform"), (3, 63, 17, 8, "random"),
              (1, 64, 64, 6, "random"), (5, 65, 7, 4, "topk"), (2, 100, 99, 6, "random"), (7, 129, 64, 8, "random"),
              (1, 1000, 999, 4, "topk"), (2, 1023, 512, 6, "random"), (2, 1025, 3, 8, "random"), (1, 2047, 1024, 2, "uniform"),
              (3, 2049, 1000, 6, "random"), (129, 96, 40, 6, "random"), (200, 64, 20, 4, "topk"), (1, 4097, 4000, 6, "random")]


@pytest.mark.parametrize("B,N,M,nb,mode", ODD_SHAPES)
def test_odd_shapes_against_the_oracle(B, N, M, nb, mode):
    """Ragged and extreme shapes of one layer call (N = K exactly, N not a multiple of any tile, M = 1, M = N, M = N - 1,
    more clouds than the fused chain takes, 2 to 8 bins, the three draws): forward + backward on the GPU, the oracle on
    the CPU, each forming its OWN boundaries from its own z-scores as a first call does (utils/ops.py:174-236).  Indices
    unique and in range, everything finite, the sampled sets the oracle's (a handful of near-tie clouds among hundreds),
    the features of every identically sampled cloud to 2e-6."""
    from samble_amd import sampler_config
    from samble_amd.downsample import DownSampleToken
    C = 128
    torch.manual_seed(B * 1000 + N)
    cfg = sampler_config("cls", M=[M, max(M // 2, 1)], bin__num_bins=[nb, nb], bin__sample_mode=[mode, mode])
    mod = DownSampleToken(cfg, 0).to(DEV)
    x = torch.from_numpy(synth.features(B, C, N, 7 + N)).to(DEV).requires_grad_(True)
    noise = torch.from_numpy(synth.exp1((B * nb, N), 8 + N)).to(DEV)
    (x_ds, idx), _ = mod(x, noise=noise)
    x_ds.backward(torch.ones_like(x_ds))
    i = idx[:, 0].cpu()
    assert idx.shape == (B, 1, M) and int(i.min()) >= 0 and int(i.max()) < N
    assert all(len(set(r.tolist())) == M for r in i), "sampling is without replacement"
    assert bool(torch.isfinite(x_ds).all()) and bool(torch.isfinite(x.grad).all())
    assert bool((mod.k_point_to_choose.sum(1) == M).all())
    spec = O.SamplerSpec(M=M, K=32, C=C, num_bins=nb, dynamic_boundaries=True, sample_mode=mode)
    st = O.SamplerState(mod.q_conv.weight.detach().cpu(), mod.k_conv.weight.detach().cpu(), mod.v_conv.weight.detach().cpu(),
                        mod.bin_tokens.detach().cpu(), None)
    x_ref, idx_ref = O.sampler_forward(spec, st, x.detach().cpu(), noise.cpu())
    assert set_agreement(i, idx_ref[:, 0]) >= 0.99
    same = (i == idx_ref[:, 0]).all(1)
    assert int(same.sum()) >= max(1, int(0.85 * B)) or set_agreement(i, idx_ref[:, 0]) == 1.0, int(same.sum())
    if bool(same.any()):
        torch.testing.assert_close(x_ds.detach().cpu()[same], x_ref[same], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(mod.bin_boundaries[0].cpu().flatten()[1:], st.boundaries[0].flatten()[1:], rtol=1e-4, atol=1e-6)


def test_sizes_no_layer_can_take_are_refused_on_the_host():
    """M > N, M < 1, N < K, a bin count outside 2..8: a ValueError / NotImplementedError naming the layer and the sizes,
    before any launch (round 6: M > N reached the kernels and ended in a GPU memory fault; the reference fails inside
    topk with "selected index k out of range").  The samplers that return the dropped points also need M < N, as the
    reference does ("cannot reshape tensor of 0 elements")."""
    from samble_amd import ops, sampler_config
    from samble_amd.downsample import DownSampleGlobal, DownSampleLocal, DownSampleToken
    x = torch.from_numpy(synth.features(2, 128, 256, 3)).to(DEV)
    for M in (300, 257, 0):
        with pytest.raises(ValueError, match="need 1 <= M"):
            DownSampleToken(sampler_config("cls", M=[M, 1]), 0).to(DEV)(x)
        for cls_, kw in ((DownSampleGlobal, dict(idx_mode=["col_sum", "col_sum"])), (DownSampleLocal, dict(idx_mode=["local_std", "local_std"]))):
            with pytest.raises(ValueError, match="need 1 <= M"):
                cls_(sampler_config("cls", M=[M, 1], **kw), 0).to(DEV)(x)
    for cls_, kw in ((DownSampleGlobal, dict(idx_mode=["col_sum", "col_sum"])), (DownSampleLocal, dict(idx_mode=["local_std", "local_std"]))):
        with pytest.raises(ValueError, match="M < N"):
            cls_(sampler_config("cls", M=[256, 1], **kw), 0).to(DEV)(x)
    (y, idx), _ = DownSampleToken(sampler_config("cls", M=[256, 1]), 0).to(DEV)(x)       # M = N: every point, as the reference
    assert sorted(idx[0, 0].tolist()) == list(range(256))
    with pytest.raises(ValueError, match="fewer than the K"):
        DownSampleToken(sampler_config("cls", M=[8, 4]), 0).to(DEV)(x[:, :, :20].contiguous())
    with pytest.raises(ValueError, match="forced_idx"):
        DownSampleToken(sampler_config("cls", M=[8, 4]), 0).to(DEV)(x, forced_idx=torch.full((2, 8), 256, device=DEV))
    for nb in (1, 9, 16):
        with pytest.raises(NotImplementedError, match="num_bins"):
            DownSampleToken(sampler_config("cls", bin__num_bins=[nb, nb]), 0)
    # the C entries refuse the same sizes by themselves (a caller that is not this module)
    z = torch.zeros((2, 256), device=DEV)
    with pytest.raises(ops._lib.SambleError, match="1 <= M <= N"):
        ops.stage_bin_select(z, z, torch.ones((2, 256), dtype=torch.uint8, device=DEV),
                             torch.full((2, 1), 300, dtype=torch.int32, device=DEV), 300, "top_raw", None)


def test_layer_under_activation_checkpointing():
    """torch.utils.checkpoint (non-reentrant) around the layer: its unpack hooks allow ONE read of ctx.saved_tensors per
    backward -- two of the layer's nodes read it piecewise and raised CheckpointError (found in round 6 by running the
    block the way a memory-bound trainer would).  With static boundaries and an injected draw nothing in the layer is
    stateful, so the recomputed backward must equal the plain one bit for bit."""
    from torch.utils.checkpoint import checkpoint
    from samble_amd import sampler_config
    from samble_amd.downsample import DownSampleToken
    B, C, N, M, nb = 3, 128, 300, 100, 6
    torch.manual_seed(2)
    mod = DownSampleToken(sampler_config("cls", M=[M, M // 2], bin__dynamic_boundaries_enable=False), 0).to(DEV)
    x = torch.from_numpy(synth.features(B, C, N, 61)).to(DEV)
    noise = torch.from_numpy(synth.exp1((B * nb, N), 62)).to(DEV)
    g = torch.from_numpy(synth.normal((B, C, M), 63)).to(DEV)
    res = []
    for wrapped in (False, True):
        mod.zero_grad()
        xin = x.clone().requires_grad_(True)
        fn = lambda t: mod(t, noise=noise)[0][0]   # noqa: E731
        y = checkpoint(fn, xin, use_reentrant=False) if wrapped else fn(xin)
        y.backward(g)
        res.append([y.detach().clone(), xin.grad.clone()] + [p.grad.clone() for p in mod.parameters()])
    for j, (a, b) in enumerate(zip(*res)):
        assert torch.equal(a, b), j


def test_cloud_past_the_lds_score_accumulators():
    """N = 13000 > 12800: a cloud's N column accumulators (12 bytes each) no longer fit a workgroup's LDS, the fused select
    chain does not take the shape (samble_select_chain_supported) and the score pass adds its integer terms to the cloud's
    global accumulators directly (csrc/score.hip DIRECT) -- rounds 1-5 refused such clouds ("N too large for LDS"; the
    reference takes any N).  Both forward pipelines bit-identical to each other; neighbour sets, score, and sampled sets
    against the oracle under the GPU's boundaries; N = 16384 (bin_select's limit) runs."""
    import samble_amd.downsample as D
    from samble_amd import ops, sampler_config
    from samble_amd.downsample import DownSampleToken
    B, C, N, M, nb = 1, 128, 13000, 3000, 6
    assert not ops.chain_supported(B, N, nb) and ops.chain_supported(B, 12800, nb)
    x = torch.from_numpy(synth.features(B, C, N, 51)).to(DEV)
    noise = torch.from_numpy(synth.exp1((B * nb, N), 52)).to(DEV)
    outs = []
    old_mode, old_free = ops.MATRIX_MODE, D.MAP_FREE
    try:
        ops.MATRIX_MODE = "tri"
        for free in (False, True):
            D.MAP_FREE = free
            torch.manual_seed(3)
            mod = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0).to(DEV)
            xin = x.clone().requires_grad_(True)
            (x_ds, idx), _ = mod(xin, noise=noise)
            x_ds.sum().backward()
            outs.append((x_ds.detach(), idx, mod.attention_point_score, xin.grad, mod.q_conv.weight.grad))
    finally:
        ops.MATRIX_MODE, D.MAP_FREE = old_mode, old_free
    for j, (t1, t2) in enumerate(zip(*outs)):
        assert torch.equal(t1, t2), j
    spec = O.SamplerSpec(M=M, K=32, C=C, num_bins=nb, dynamic_boundaries=False)
    st = O.SamplerState(mod.q_conv.weight.detach().cpu(), mod.k_conv.weight.detach().cpu(), mod.v_conv.weight.detach().cpu(),
                        mod.bin_tokens.detach().cpu(), [t.cpu().clone() for t in mod.bin_boundaries])
    _, idx_ref = O.sampler_forward(spec, st, x.cpu(), noise.cpu())
    assert set_agreement(mod.knn_idx.cpu(), st.trace["knn_idx"]) >= 0.9999
    off = ~torch.isclose(mod.attention_point_score[:, 0].cpu(), st.trace["score"].reshape(B, N), rtol=1e-4, atol=1e-9)
    assert int(off.sum()) <= 8, int(off.sum())     # (a near-tie at a row's K-th neighbour moves one entry between two columns)
    assert set_agreement(idx[:, 0].cpu(), idx_ref[:, 0]) >= 0.995
    # the longest cloud the selection kernels take (bin_select keeps a cloud's keys in LDS): runs, unique, finite
    B2, N2, M2 = 1, 16384, 5000
    mod2 = DownSampleToken(sampler_config("cls", M=[M2, M2 // 2]), 0).to(DEV)
    x2 = torch.from_numpy(synth.features(B2, C, N2, 53)).to(DEV).requires_grad_(True)
    (y2, i2), _ = mod2(x2, noise=torch.from_numpy(synth.exp1((B2 * nb, N2), 54)).to(DEV))
    y2.sum().backward()
    assert len(set(i2[0, 0].tolist())) == M2 and int(i2.max()) < N2 and bool(torch.isfinite(y2).all())
    assert bool(torch.isfinite(x2.grad).all())
    with pytest.raises(ops._lib.SambleError, match="N <= 16384"):
        mod3 = DownSampleToken(sampler_config("cls", M=[100, 50]), 0).to(DEV)
        mod3(torch.from_numpy(synth.features(1, C, 16500, 55)).to(DEV))


def test_map_free_without_the_fused_chain():
    """B = 130 clouds: more workgroups than the fused select chain takes (one per cloud, B <= 128), so the map-free
    forward hands its in-pass score statistics to the stand-alone finalize / quantile / bin kernels -- same outputs as
    the logit-map pipeline."""
    import samble_amd.downsample as D
    from samble_amd import ops, sampler_config
    from samble_amd.downsample import DownSampleToken
    B, C, N, M, nb = 130, 128, 96, 40, 6
    assert not ops.chain_supported(B, N, nb)
    x = torch.from_numpy(synth.features(B, C, N, 41)).to(DEV)
    noise = torch.from_numpy(synth.exp1((B * nb, N), 42)).to(DEV)
    outs = []
    old_mode, old_free = ops.MATRIX_MODE, D.MAP_FREE
    try:
        ops.MATRIX_MODE = "tri"
        for free in (False, True):
            D.MAP_FREE = free
            torch.manual_seed(5)
            mod = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0).to(DEV)
            xin = x.clone().requires_grad_(True)
            (x_ds, idx), _ = mod(xin, noise=noise)
            x_ds.sum().backward()
            outs.append((x_ds.detach(), idx, mod.attention_point_score, mod.bin_boundaries[0].clone(), xin.grad,
                         mod.k_conv.weight.grad))
    finally:
        ops.MATRIX_MODE, D.MAP_FREE = old_mode, old_free
    for j, (t1, t2) in enumerate(zip(*outs)):
        assert torch.equal(t1, t2), j


@pytest.mark.parametrize("k", [-40, -12, 20, 60])
def test_backward_is_exactly_linear_in_a_power_of_two(k):
    """Every operand of the backward's matrix products carries power-of-two scales (tile scales of dO / Q / V, the cloud's
    largest |dS|, csrc/tri_dev.h) or is split into planes that scale exactly: an upstream gradient x 2^k must give every
    gradient x 2^k, bit for bit -- from 2^-40 to 2^60 (nothing underflows into fp16's denormals, nothing overflows) --
    and a zero gradient gives zeros."""
    g = Golden("cls_random_dyn")
    noise = g.t("noise").to(DEV)
    up = g.upstream().to(DEV)
    grads = []
    for scale in (1.0, 2.0 ** k, 0.0):
        mod = g.module(DEV)
        x = g.x().to(DEV).requires_grad_(True)
        (x_ds, _), _ = mod(x, noise=noise)
        x_ds.backward(up * scale)
        grads.append([x.grad.clone()] + [p.grad.clone() for p in (mod.q_conv.weight, mod.k_conv.weight, mod.v_conv.weight,
                                                                   mod.bin_tokens)])
    for a, b2, z in zip(*grads):
        assert bool(torch.isfinite(b2).all())
        assert torch.equal(a * 2.0 ** k, b2), float((a * 2.0 ** k - b2).abs().max())
        assert bool((z == 0).all())


def test_step_is_graph_capturable_and_replays_bit_identically():
    """forward + backward of the metric layer (B=32, N=2048 -> 1024, injected noise) captured as ONE
    hipGraph (torch.cuda.graph), replayed twice: idx, x_ds, dx and the four parameter gradients are bit-identical between
    the replays AND to the eager run of the same step from the same boundary state.  (Round 3's segfault in
    hipStreamEndCapture does not reproduce on this tree: tools/experiments/graph_capture_probe.py runs six configurations,
    with and without the fused chain and its pinned-memory mailbox.)"""
    from samble_amd import sampler_config
    from samble_amd.downsample import DownSampleToken
    B, C, N, M, nb = 32, 128, 2048, 1024, 6
    torch.manual_seed(3)
    mod = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0).to(DEV)
    x = torch.from_numpy(synth.features(B, C, N, 7101)).to(DEV).requires_grad_(True)
    noise = torch.from_numpy(synth.exp1((B * nb, N), 7102)).to(DEV)
    g = torch.from_numpy(synth.normal((B, C, M), 7103)).to(DEV)
    params = [mod.q_conv.weight, mod.k_conv.weight, mod.v_conv.weight, mod.bin_tokens]

    def step():
        for p in params + [x]:
            if p.grad is not None:
                p.grad.zero_()
        (x_ds, idx), _ = mod(x, noise=noise)
        x_ds.backward(g)
        return x_ds, idx

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()                                      # allocator pools, first-call boundary state, grads allocated
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    state = [t.clone() for t in mod.bin_boundaries]

    def snapshot(out):
        return [out[0].detach().clone(), out[1].clone(), x.grad.clone()] + [p.grad.clone() for p in params] + \
               [t.clone() for t in mod.bin_boundaries]

    eager = snapshot(step())
    torch.cuda.synchronize()
    for t, t0 in zip(mod.bin_boundaries, state):
        t.copy_(t0)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    replays = []
    for _ in range(2):
        for t, t0 in zip(mod.bin_boundaries, state):   # (the captured step blends into these very tensors)
            t.copy_(t0)
        graph.replay()
        torch.cuda.synchronize()
        replays.append(snapshot(out))
    names = ["x_ds", "idx", "dx", "dWq", "dWk", "dWv", "dtokens", "upper", "lower"]
    for n, a, b2, e in zip(names, replays[0], replays[1], eager):
        assert torch.equal(a, b2), f"{n}: the two replays differ"
        assert torch.equal(a, e), f"{n}: replay differs from the eager step"
    assert not mod._chain_watch.timed_out(sync=True)


SLIM_FIXTURES = ("headline_cls_B32_N2048", "stress_cls_B2_N8192")


def _headline_fixture(name="headline_cls_B32_N2048"):
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    B, C, N, M, nb, K, _, seed = [int(v) for v in d["meta"]]
    noise = torch.from_numpy(d["noise"])   # the Exp(1) draws the reference's torch.multinomial consumed (utils/ops.py:595)
    return d, (B, C, N, M, nb, K, seed), noise


def headline_module_and_step(dev=DEV, name="headline_cls_B32_N2048"):
    """The layer on a slim full-size fixture's inputs: (fixture, module after forward + backward, idx, x_ds, x)."""
    from samble_amd import sampler_config
    from samble_amd.downsample import DownSampleToken
    d, (B, C, N, M, nb, K, seed), noise = _headline_fixture(name)
    mod = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0)
    wq, wk, wv, tok = synth.sampler_weights(C, nb, seed)
    with torch.no_grad():
        mod.q_conv.weight.copy_(torch.from_numpy(wq))
        mod.k_conv.weight.copy_(torch.from_numpy(wk))
        mod.v_conv.weight.copy_(torch.from_numpy(wv))
        mod.bin_tokens.copy_(torch.from_numpy(tok))
    mod = mod.to(dev)
    x = torch.from_numpy(synth.features(B, C, N, seed + 10)).to(dev).requires_grad_(True)
    (x_ds, idx), _ = mod(x, noise=noise.to(dev))
    x_ds.backward(torch.from_numpy(synth.normal((B, C, M), seed + 99)).to(dev))
    return d, mod, idx, x_ds, x


@pytest.mark.parametrize("fixture_name", SLIM_FIXTURES)
def test_headline_configuration_against_the_reference_fixture(matrix_mode, fixture_name):
    """(`stress_cls_B2_N8192`: the same test on BASELINE configs[4]'s geometry, N = 8192 -> 4096 on two clouds.)
    BASELINE.json's metric configuration -- B=32, C=128, N=2048 -> 1024, 6 bins, K=32, sparse_col_sqr, DYNAMIC boundaries
    from a fresh state, random T=0.1 -- against a fixture the UNMODIFIED reference wrote in the build container
    (tests/golden/make_golden_headline.py; reference utils/ops.py:385-432, 467-619): the whole-batch quantile boundaries,
    bin populations, bin weights and counts, and the sampled indices per cloud.  The per-cloud identity is pinned in
    tests/expected_identity.json (tools/fixture_identity.py): a cloud identical there stays identical; one that is not is
    explained by one of the two measured mechanisms (a truncation flip of the float water-filling, a near-tie of two keys)."""
    d, mod, idx, x_ds, x = headline_module_and_step(name=fixture_name)
    B, M = idx.shape[0], idx.shape[2]
    torch.testing.assert_close(mod.bin_boundaries[0].cpu(), torch.from_numpy(d["upper"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(mod.bin_boundaries[1].cpu(), torch.from_numpy(d["lower"]), rtol=1e-4, atol=1e-5)
    score = mod.attention_point_score.cpu()
    # scores: fp tolerance -- except where a neighbour SET differs: the reference's cdist takes ATen's mm path, whose
    # rounding noise decides near-ties of the K-th neighbour (kNN parity is a set-match rate); one
    # flipped neighbour moves the in-degree of two columns, i.e. their scores by a few per cent
    want = torch.from_numpy(d["score_first"])
    off = ((score[:2] - want).abs() > 3e-5 * want.abs() + 1e-9)
    assert float(off.float().mean()) <= 5e-3 and float(((score[:2] - want).abs() / want.abs().clamp_min(1e-12)).max()) <= 0.2, \
        (int(off.sum()), float(((score[:2] - want).abs() / want.abs().clamp_min(1e-12)).max()))
    torch.testing.assert_close(score.double().sum((1, 2)), torch.from_numpy(d["score_cloud_sums"])[:, 0], rtol=2e-4, atol=0)
    cap = torch.from_numpy(d["cap"]).long()
    cap_same = (mod.max_num_points.cpu().long() == cap).all(1)
    assert int(cap_same.sum()) >= max(B - 2, 0), "bin populations (a point whose z sits on a boundary may change bin)"
    torch.testing.assert_close(mod.bin_weights_beforerelu.cpu()[cap_same], torch.from_numpy(d["w_pre"])[cap_same], rtol=2e-5, atol=1e-6)
    got, ref = idx.cpu()[:, 0], torch.from_numpy(d["idx"].astype(np.int64))
    same = (got == ref).all(1)
    counts_same = (mod.k_point_to_choose.cpu() == torch.from_numpy(d["counts"])).all(1)
    print(f"\nheadline fixture ({matrix_mode}): clouds with the reference's exact index tensor {int(same.sum())} of {B}: "
          f"{same.int().tolist()}; counts identical {int(counts_same.sum())} of {B}; set agreement {set_agreement(got, ref):.5f}")
    pinned = _pinned_identity(matrix_mode, fixture_name, 0)
    assert pinned is not None, "no row for the headline fixture in tests/expected_identity.json (tools/fixture_identity.py)"
    lost = [b for b in range(B) if pinned[b] and not bool(same[b])]
    assert not lost, f"clouds {lost} carried the reference's exact indices when the table was pinned"
    w_ours, w_ref = torch.relu(mod.bin_weights_beforerelu.cpu()), torch.relu(torch.from_numpy(d["w_pre"]))
    counts_ours, counts_ref = mod.k_point_to_choose.cpu(), torch.from_numpy(d["counts"])
    for b in range(B):
        if bool(same[b]):
            continue
        differ = len(set(got[b].tolist()) ^ set(ref[b].tolist()))
        if not bool(counts_same[b]):
            # picks moved between bins: the float water-filling (utils/ops.py:403-424) lands on an integer and `.int()`
            # truncates it up or down on a 1e-9 difference of a bin weight (once or twice per cloud; the remainder then goes
            # to one bin).  Evidence per cloud: the integer stage is exact on both sides -- the reference's own allocation
            # (oracle) run on OUR weights gives OUR counts, run on the fixture's weights the fixture's -- and the weights
            # agree to 2e-6; or ONE point sits on a bin boundary and changed bin (capacities differ by one point)
            # (one flip moves one pick; a bin that saturates on one side and not on the other -- its share lands within 1e-6
            # of its capacity -- re-routes a dozen: same discontinuity, same proof)
            dc = counts_ours[b].long() - counts_ref[b].long()
            assert int(dc.sum()) == 0, (b, dc.tolist())
            if bool(cap_same[b]):
                torch.testing.assert_close(mod.bin_weights_beforerelu.cpu()[b], torch.from_numpy(d["w_pre"])[b], rtol=2e-5, atol=1e-6)
                assert torch.equal(O.allocate_counts(w_ours.clone(), cap, M)[b], counts_ours[b])
                assert torch.equal(O.allocate_counts(w_ref.clone(), cap, M)[b], counts_ref[b].int())
            else:   # one point on a bin boundary changed bin; the allocation on OUR populations and weights gives OUR counts
                cap_ours = mod.max_num_points.cpu().long()
                assert int((cap_ours[b] - cap[b]).abs().sum()) <= 2, (b, cap_ours[b].tolist(), cap[b].tolist())
                assert torch.equal(O.allocate_counts(w_ours.clone(), cap_ours, M)[b], counts_ours[b])
            assert differ <= int(dc.abs().sum()) + 2, (b, dc.tolist(), differ)
        else:                          # a near-tie of two selection keys or of a score and a boundary
            assert differ <= 4, (b, differ)
    assert set_agreement(got, ref) >= 0.999
    # values on the clouds that carry the reference's indices: per-cloud float64 sums of x_ds and dx
    xd, dx = x_ds.detach().cpu().double(), x.grad.cpu().double()
    for t, key in ((xd, "x_ds_cloud_sums"), (dx, "dx_cloud_sums")):
        sums = torch.stack([t.sum((1, 2)), t.square().sum((1, 2))], dim=1)
        want = torch.from_numpy(d[key])
        torch.testing.assert_close(sums[same][:, 1], want[same][:, 1], rtol=2e-4, atol=0)
        scale = want[:, 1].sqrt() * (t[0].numel() ** 0.5)
        assert bool(((sums[same][:, 0] - want[same][:, 0]).abs() <= 2e-5 * scale[same]).all()), key
    if bool(same.all()):
        for p, key in ((mod.q_conv.weight, "dwq"), (mod.k_conv.weight, "dwk"), (mod.v_conv.weight, "dwv"), (mod.bin_tokens, "dtokens")):
            ref_g = torch.from_numpy(d[key])
            assert float((p.grad.cpu() - ref_g).abs().max()) <= 2e-4 * float(ref_g.abs().max()) + 1e-6, key

"""csrc/linear.hip: the 1x1 convolutions around the neighbour / sampler kernels (reference models/attention.py:187-192:
the attention layers' feed-forward part; models/cls_model.py:113,136,144: Conv1d + max over the points) against the same
expressions in float64 torch, forward and backward, at ragged and full sizes."""
import numpy as np
import pytest
import torch

from samble_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _w(shape, seed, scale):
    return torch.from_numpy((synth.normal(shape, seed).astype(np.float64) * scale).astype(np.float32))


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize("B,N,O", [(2, 300, 512), (1, 33, 256), (3, 1024, 512), (2, 77, 1024), (32, 2048, 512),
                                   (32, 1024, 512), (32, 500, 256)])   # (the last two: the launch shapes of the coarse levels)
def test_linear_stages_against_float64(B, N, O):
    from samble_amd import linear as L
    x = torch.from_numpy(synth.features(B, 128, N, 50 + N)).to(DEV)
    W = _w((O, 128), 51 + N, 0.09).to(DEV)
    rm, tr = L.weight_images(W)
    ref = torch.einsum("oc,bcn->bno", W.double(), x.double())
    out = L.stage_linear_fwd(x, rm, O)
    assert out.shape == (B, N, O) and _rel(out, ref) <= 2e-6
    lk = L.stage_linear_fwd(x, rm, O, L.LIN_LEAKY)
    assert _rel(lk, torch.nn.functional.leaky_relu(ref, 0.2)) <= 2e-6
    assert torch.equal(lk, torch.where(out > 0, out, 0.2 * out))                     # the epilogue is elementwise on the same sums
    mk = L.stage_linear_fwd(x, rm, O, L.LIN_LEAKY_MASK, ref=lk)
    assert torch.equal(mk, torch.where(lk > 0, out, 0.2 * out))
    assert torch.equal(L.stage_linear_fwd(x, rm, O), out), "run-to-run identical"
    # the same pair with the activation's sign as one bit per value (what the layers use: 1/32 of the mask pass's reads)
    lkb, bits = L.stage_linear_fwd(x, rm, O, L.LIN_LEAKY_BITS)
    assert torch.equal(lkb, lk) and bits.numel() * 8 == B * N * O
    assert torch.equal(L.stage_linear_fwd(x, rm, O, L.LIN_LEAKY_MASK_BITS, bits=bits), mk)
    # dx: point-major rows back to the channel-major side
    g = torch.from_numpy(synth.normal((B, N, O), 52 + N)).to(DEV)
    dx = L.stage_linear_dx(g, tr, O)
    assert dx.shape == (B, 128, N) and _rel(dx, torch.einsum("oc,bno->bcn", W.double(), g.double())) <= 2e-6
    # ... with a residual on the kernel's epilogue: bitwise the separate add, also in place (ragged N: the clamped lanes)
    r = torch.from_numpy(synth.normal((B, 128, N), 53 + N)).to(DEV)
    assert torch.equal(L.stage_linear_dx(g, tr, O, residual=r), r + dx)
    r2 = r.clone()
    assert L.stage_linear_dx(g, tr, O, residual=r2, out=r2) is r2 and torch.equal(r2, r + dx)
    # dW: contraction over clouds and points, deterministic
    if O % 256 == 0:
        dW = L.stage_linear_dw(g, x, O)
        assert _rel(dW, torch.einsum("bno,bcn->oc", g.double(), x.double())) <= 3e-6
        assert torch.equal(L.stage_linear_dw(g, x, O), dW)
    # max over the points
    y, arg = L.stage_linear_amax(x, rm, O)
    ref_y, ref_arg = ref.max(dim=1)
    assert _rel(y, ref_y) <= 2e-6
    same = arg.long().cpu() == ref_arg.cpu()
    # a near-tie of two points may resolve the other way in fp32: the point picked must then hold the maximum to rounding
    picked = torch.gather(ref, 1, arg.long().unsqueeze(1)).squeeze(1)
    assert float(same.float().mean()) >= 0.999 and float(((ref_y - picked) / ref_y.abs().clamp_min(1e-3)).max()) <= 1e-5
    assert torch.equal(y, torch.gather(out, 1, arg.long().unsqueeze(1)).squeeze(1)), "the maximum is one of the kernel's own sums"
    assert bool((out <= y.unsqueeze(1)).all())


@pytest.mark.parametrize("B,N", [(2, 300), (4, 2048), (1, 40)])
def test_ffn_against_the_torch_modules(B, N):
    """ffn = Conv1d(128->512) -> LeakyReLU(0.2) -> Conv1d(512->128) with the Conv1d weights (models/attention.py:187-192):
    output, input gradient and both weight gradients against the stock modules in float64."""
    from samble_amd import linear as L
    ff = torch.nn.Sequential(torch.nn.Conv1d(128, 512, 1, bias=False), torch.nn.LeakyReLU(0.2), torch.nn.Conv1d(512, 128, 1, bias=False))
    with torch.no_grad():
        ff[0].weight.copy_(_w((512, 128, 1), 61, 0.09))
        ff[2].weight.copy_(_w((128, 512, 1), 62, 0.045))
    x = torch.from_numpy(synth.features(B, 128, N, 63 + N))
    g = torch.from_numpy(synth.normal((B, 128, N), 64 + N))
    ref = ff.double()
    xd = x.double().requires_grad_(True)
    yr = ref(xd)
    yr.backward(g.double())
    w1 = ff[0].weight.detach().float().to(DEV).requires_grad_(True)
    w2 = ff[2].weight.detach().float().to(DEV).requires_grad_(True)
    xg = x.to(DEV).requires_grad_(True)
    assert L.ffn_supported(xg, w1, w2)
    y = L.ffn(xg, w1, w2)
    y.backward(g.to(DEV))
    assert _rel(y, yr) <= 3e-6
    assert _rel(xg.grad, xd.grad) <= 3e-6
    assert _rel(w1.grad, ref[0].weight.grad) <= 5e-6 and _rel(w2.grad, ref[2].weight.grad) <= 5e-6
    # deterministic
    xg2 = x.to(DEV).requires_grad_(True)
    w1b, w2b = w1.detach().clone().requires_grad_(True), w2.detach().clone().requires_grad_(True)
    y2 = L.ffn(xg2, w1b, w2b)
    y2.backward(g.to(DEV))
    assert torch.equal(y2, y) and torch.equal(xg2.grad, xg.grad) and torch.equal(w1b.grad, w1.grad) and torch.equal(w2b.grad, w2.grad)


@pytest.mark.parametrize("B,N,O", [(2, 300, 1024), (32, 2048, 1024), (3, 64, 128)])
def test_linear_max_against_conv_max(B, N, O):
    """linear_max = Conv1d(128->O)(x).max(dim=-1)[0] (models/cls_model.py:113,136): values, dx and dW against float64
    torch (whose max(dim) backward also sends the gradient to the ONE index it returned); a point that is the arg-max of
    several outputs collects all of them."""
    from samble_amd import linear as L
    conv = torch.nn.Conv1d(128, O, 1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(_w((O, 128, 1), 71, 0.09))
    x = torch.from_numpy(synth.features(B, 128, N, 72 + N))
    g = torch.from_numpy(synth.normal((B, O), 73 + N))
    xd = x.double().requires_grad_(True)
    cd = conv.double()
    yr, ar = cd(xd).max(dim=-1)
    yr.backward(g.double())
    w = conv.weight.detach().float().to(DEV).requires_grad_(True)
    xg = x.to(DEV).requires_grad_(True)
    assert L.linear_max_supported(xg, w)
    y, arg = L._LinearMax.apply(xg, w)
    y.backward(g.to(DEV))
    assert _rel(y, yr) <= 2e-6
    same = (arg.long().cpu() == ar)
    assert float(same.float().mean()) >= 0.999
    if bool(same.all()):
        assert _rel(xg.grad, xd.grad) <= 3e-6 and _rel(w.grad, cd.weight.grad) <= 3e-6
    else:  # a near-tie picked another point for a few outputs: the float64 gradient through OUR indices
        x2 = x.double().requires_grad_(True)
        cd.zero_grad()
        torch.gather(cd(x2), 2, arg.long().cpu().unsqueeze(-1)).squeeze(-1).backward(g.double())
        assert _rel(xg.grad, x2.grad) <= 3e-6 and _rel(w.grad, cd.weight.grad) <= 3e-6
    # columns that are nobody's arg-max get exactly zero
    hit = torch.zeros((B, N), dtype=torch.bool)
    hit.scatter_(1, arg.long().cpu(), True)
    assert bool((xg.grad.cpu().abs().sum(1)[~hit] == 0).all())
    # deterministic
    xg2 = x.to(DEV).requires_grad_(True)
    wb = w.detach().clone().requires_grad_(True)
    y2, arg2 = L._LinearMax.apply(xg2, wb)
    y2.backward(g.to(DEV))
    assert torch.equal(y2, y) and torch.equal(arg2, arg) and torch.equal(xg2.grad, xg.grad) and torch.equal(wb.grad, w.grad)


def test_transposed_weight_entries_equal_the_transposing_copies():
    """samble_linear_weight_images_t_f32 / samble_linear_dw_t_tri_f32 (round 5): the second FFN convolution's weight is
    (128, H); its W^T images and its gradient in (128, H) layout come straight from / go straight to that layout -- bit
    for bit what the copies `.t().contiguous()` around the plain entries produced."""
    from samble_amd import linear as L
    H, B, N = 512, 3, 700
    W2 = _w((128, H), 91, 0.045).to(DEV)
    rm_t, tr_t = L.weight_images(W2, transposed=True)
    rm, tr = L.weight_images(W2.t().contiguous())
    assert torch.equal(rm_t, rm) and torch.equal(tr_t, tr)
    g = torch.from_numpy(synth.normal((B, N, H), 92)).to(DEV)
    x = torch.from_numpy(synth.features(B, 128, N, 93)).to(DEV)
    assert torch.equal(L.stage_linear_dw(g, x, H, transposed=True), L.stage_linear_dw(g, x, H).t().contiguous())
    # both weights of a feed-forward layer in one launch: the four images of the two single calls
    W1 = _w((H, 128), 94, 0.09).to(DEV)
    r1, t1 = L.weight_images(W1)
    for got, want in zip(L.ffn_weight_images(W1, W2), (r1, t1, rm_t, tr_t)):
        assert torch.equal(got, want)


def test_linear_max_propagates_nan_like_torch():
    """A NaN activation must not be masked by the pooled head: conv(x).max(dim=-1) (models/cls_model.py:113) returns NaN
    for every output of a cloud that has a NaN point, with the FIRST NaN point as the argument -- the fused pass does
    the same; the other clouds are untouched.  A wider head than the kernels take reports unsupported (stock path)."""
    from samble_amd import linear as L
    B, N, O = 3, 300, 256
    w = _w((O, 128, 1), 81, 0.09).to(DEV)
    x = torch.from_numpy(synth.features(B, 128, N, 82)).to(DEV)
    y0, a0 = L._LinearMax.apply(x, w)
    x_bad = x.clone()
    x_bad[1, 5, 170] = float("nan")     # poisons every output channel of cloud 1 at point 170 ...
    x_bad[1, 9, 233] = float("nan")     # ... and at point 233, in another tile and wave half
    y, arg = L._LinearMax.apply(x_bad, w)
    yr, ar = torch.nn.functional.conv1d(x_bad, w).max(dim=-1)
    assert bool(torch.isnan(yr[1]).all()) and bool((ar[1] == 170).all())
    assert bool(torch.isnan(y[1]).all()) and bool((arg[1].long() == 170).all())
    assert torch.equal(y[0], y0[0]) and torch.equal(y[2], y0[2]) and torch.equal(arg[0], a0[0]) and torch.equal(arg[2], a0[2])
    assert not L.linear_max_supported(x, torch.zeros((8192, 128, 1), device=DEV))
    assert not L.ffn_supported(x, torch.zeros((8192, 128, 1), device=DEV), torch.zeros((128, 8192, 1), device=DEV))


@pytest.mark.parametrize("B,C,N,O", [(2, 64, 300, 128), (3, 3, 2048, 128), (1, 6, 77, 256), (2, 100, 513, 128)])
def test_linear_rows_with_fewer_input_channels(B, C, N, O):
    """linear_rows: x (B, C <= 128, N) -> (B, N, O) = (W x)^T, the per-point projections of EdgeConv's conv1
    (models/embedding.py:20-28; C = 3 or 64 there), against float64 torch: output, dx and dW."""
    from samble_amd import linear as L
    x = torch.from_numpy(synth.normal((B, C, N), 80 + N)).to(DEV).requires_grad_(True)
    W = _w((O, C), 81 + N, 0.2).to(DEV).requires_grad_(True)
    g = torch.from_numpy(synth.normal((B, N, O), 82 + N)).to(DEV)
    assert L.linear_supported(x, W)
    y = L.linear_rows(x, W)
    y.backward(g)
    xd, Wd = x.detach().double().requires_grad_(True), W.detach().double().requires_grad_(True)
    yr = torch.einsum("oc,bcn->bno", Wd, xd)
    yr.backward(g.double())
    assert y.shape == (B, N, O) and _rel(y.detach(), yr.detach()) <= 2e-6
    assert x.grad.shape == (B, C, N) and _rel(x.grad, xd.grad) <= 3e-6
    assert W.grad.shape == (O, C) and _rel(W.grad, Wd.grad) <= 3e-6


@pytest.mark.parametrize("B,N,k", [(2, 300, 1), (3, 77, 2), (32, 2048, 2), (4, 1024, 1), (1, 513, 2)])
def test_channel_major_pointwise_conv_against_float64(B, N, k):
    """The interpolation layers' two 1x1 convolutions (reference models/upsample.py:142-150: `conv` on the coarse set,
    `res_conv(torch.cat((pcd_up, interpolated), dim=1))`), channel-major in and out, the concatenation never formed."""
    from samble_amd import linear as L
    xs = [torch.from_numpy(synth.features(B, 128, N, 300 + N + i)).to(DEV).requires_grad_(True) for i in range(k)]
    w = _w((128, 128 * k, 1), 310 + N, 0.09).to(DEV).requires_grad_(True)
    assert L.pointwise_cm_supported(w, *xs)
    y = L.pointwise_cm(w, *xs)
    cat = torch.cat([x.detach().double() for x in xs], dim=1).requires_grad_(True)
    wd = w.detach().double().requires_grad_(True)
    ref = torch.nn.functional.conv1d(cat, wd)
    assert y.shape == (B, 128, N) and y.is_contiguous() and _rel(y, ref) <= 2e-6
    g = torch.from_numpy(synth.normal((B, 128, N), 320 + N)).to(DEV)
    y.backward(g)
    ref.backward(g.double())
    for i, x in enumerate(xs):
        assert _rel(x.grad, cat.grad[:, 128 * i:128 * i + 128]) <= 2e-6
    assert w.grad.shape == w.shape and _rel(w.grad, wd.grad) <= 3e-6
    # the stages behind it: the second half accumulates onto the first bit for bit like a separate add; run-to-run identical
    W = w.detach().reshape(128, -1)
    rm0, _ = L.weight_images(W[:, :128].contiguous(), want_tr=False)
    a = L.stage_linear_fwd_cm(xs[0].detach(), rm0, 128)
    assert torch.equal(a, L.stage_linear_fwd(xs[0].detach(), rm0, 128).transpose(1, 2)), "the point-major kernel's sums, stored the other way"
    if k == 2:
        rm1, _ = L.weight_images(W[:, 128:].contiguous(), want_tr=False)
        b = L.stage_linear_fwd_cm(xs[1].detach(), rm1, 128)
        assert torch.equal(L.stage_linear_fwd_cm(xs[1].detach(), rm1, 128, out=a.clone(), accumulate=True), b + a)
        assert torch.equal(L.pointwise_cm(w.detach(), *[x.detach() for x in xs]), b + a)
    dw = L.stage_linear_dw_cm(g, xs[0].detach())
    assert torch.equal(dw, L.stage_linear_dw_cm(g, xs[0].detach()))
    assert torch.equal(dw, L.stage_linear_dw(g.transpose(1, 2).contiguous(), xs[0].detach(), 128)), "the same sums as from point-major g"


@pytest.mark.parametrize("B,N,H", [(32, 2048, 512), (2, 300, 512), (3, 77, 256), (1, 33, 1024), (5, 1024, 512)])
def test_chain_equals_the_two_kernels(B, N, H):
    """samble_linear_chain_f32 (round 5): a feed-forward layer's two convolutions in one sweep -- the intermediate, its sign
    words and the output are BIT FOR BIT those of lin_fwd (leaky / mask from sign words) followed by lin_dx, with and without
    the residual, forward form and backward form (reference models/attention.py:187-192 `ff` and its input gradient)."""
    from samble_amd import linear as L
    x = torch.from_numpy(synth.features(B, 128, N, 700 + N)).to(DEV)
    W1 = _w((H, 128), 701 + N, 0.09).to(DEV)
    W2 = _w((128, H), 702 + N, 0.045).to(DEV)
    w1_rm, w1_tr, w2t_rm, w2t_tr = L.ffn_weight_images(W1, W2)
    r = torch.from_numpy(synth.normal((B, 128, N), 703 + N)).to(DEV)
    # forward: y = W2 leaky(W1 x) [+ r]
    hr, bits = L.stage_linear_fwd(x, w1_rm, H, L.LIN_LEAKY_BITS)
    for res in (None, r):
        want = L.stage_linear_dx(hr, w2t_tr, H, residual=res)
        out, mid, b2 = L.stage_linear_chain(x, w1_rm, w2t_tr, H, L.LIN_LEAKY_BITS, residual=res)
        assert torch.equal(mid, hr) and torch.equal(b2, bits) and torch.equal(out, want)
    out, mid, _ = L.stage_linear_chain(x, w1_rm, w2t_tr, H, L.LIN_LEAKY_BITS, want_mid=False)
    assert mid is None and torch.equal(out, L.stage_linear_dx(hr, w2t_tr, H))
    ref = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(torch.nn.functional.conv1d(x.double(), W1.double().unsqueeze(-1)), 0.2),
                                     W2.double().unsqueeze(-1))
    assert _rel(out, ref) <= 3e-6
    # backward: dx = W1^T (mask (W2^T dy)) [+ r]
    dy = torch.from_numpy(synth.normal((B, 128, N), 704 + N)).to(DEV)
    dh = L.stage_linear_fwd(dy, w2t_rm, H, L.LIN_LEAKY_MASK_BITS, bits=bits)
    for res in (None, r):
        want = L.stage_linear_dx(dh, w1_tr, H, residual=res)
        dx, mid, _ = L.stage_linear_chain(dy, w2t_rm, w1_tr, H, L.LIN_LEAKY_MASK_BITS, bits=bits, residual=res)
        assert torch.equal(mid, dh) and torch.equal(dx, want)


@pytest.mark.parametrize("B,N,H", [(32, 2048, 4096), (6, 1024, 2048), (9, 512, 4096), (3, 300, 1024)])
def test_chain_without_the_intermediate_waits_for_its_weight_tiles(B, N, H):
    """with mid == NULL no stores are issued, and the hand-counted `s_waitcnt vmcnt` of lin_chain must
    not count them -- or the barrier releases the waves onto an LDS slot whose weight tile has not landed.  Long hidden
    layers (128 tiles: 128 chances per wave), all three workgroup sizes (8 / 4 / 2 waves), many repetitions: the output must
    be bit for bit the one computed WITH the intermediate, every time."""
    from samble_amd import linear as L
    x = torch.from_numpy(synth.features(B, 128, N, 900 + N)).to(DEV)
    W1 = _w((H, 128), 901 + N, 0.09).to(DEV)
    W2 = _w((128, H), 902 + N, 0.02).to(DEV)
    w1_rm, w1_tr, w2t_rm, w2t_tr = L.ffn_weight_images(W1, W2)
    want, mid, bits = L.stage_linear_chain(x, w1_rm, w2t_tr, H, L.LIN_LEAKY_BITS)
    assert torch.equal(want, L.stage_linear_dx(mid, w2t_tr, H))
    for _ in range(8):
        out, none, b2 = L.stage_linear_chain(x, w1_rm, w2t_tr, H, L.LIN_LEAKY_BITS, want_mid=False)
        assert none is None and torch.equal(b2, bits) and torch.equal(out, want)
    dy = torch.from_numpy(synth.normal((B, 128, N), 903 + N)).to(DEV)
    want_dx, _, _ = L.stage_linear_chain(dy, w2t_rm, w1_tr, H, L.LIN_LEAKY_MASK_BITS, bits=bits)
    for _ in range(4):
        dx, _, _ = L.stage_linear_chain(dy, w2t_rm, w1_tr, H, L.LIN_LEAKY_MASK_BITS, bits=bits, want_mid=False)
        assert torch.equal(dx, want_dx)


@pytest.mark.parametrize("B,N,O", [(32, 257, 2048), (16, 2048 + 5, 1024), (40, 256 + 32, 4096), (7, 512 + 17, 2048)])
def test_channel_major_conv_tail_chunk_with_idle_waves(B, N, O):
    """in a tail chunk with N % 256 <= 224 some waves of the channel-major lin_fwd lie wholly past N;
    they used to issue no stores, so their `vmcnt(54)` did not wait for their share of the next weight tile's DMA -- which
    the OTHER waves then read.  Many output tiles, repeated: bit for bit the point-major kernel's sums, and float64."""
    from samble_amd import linear as L
    x = torch.from_numpy(synth.features(B, 128, N, 950 + N)).to(DEV)
    W = _w((O, 128), 951 + N, 0.09).to(DEV)
    rm, _ = L.weight_images(W, want_tr=False)
    want = L.stage_linear_fwd(x, rm, O).transpose(1, 2).contiguous()
    ref = torch.einsum("oc,bcn->bon", W.double(), x.double())
    assert _rel(want, ref) <= 2e-6
    for _ in range(8):
        got = L.stage_linear_fwd_cm(x, rm, O)
        assert torch.equal(got, want)


@pytest.mark.parametrize("B,N,O", [(3, 300, 1024), (2, 2048, 4096), (32, 2048, 1024)])
def test_pooled_head_backward_groups_are_ordered_whatever_their_size(B, N, O):
    """amax_sort (round 6: atomic slots + a sort per group instead of a quadratic count): the gradient of
    `conv(x).max(dim=-1)` (models/cls_model.py:113) against float64 on ordinary clouds AND on degenerate ones -- a constant
    cloud sends EVERY output's gradient to point 0 (one group of O members), a cloud with two distinct points makes two big
    groups -- run to run identical."""
    from samble_amd import linear as L
    W = _w((O, 128), 400 + O, 0.09).to(DEV)
    x = torch.from_numpy(synth.features(B, 128, N, 401 + N)).to(DEV)
    x[0] = x[0, :, :1]                                   # cloud 0: all points equal -> every arg-max is point 0
    x[1, :, 1::2] = x[1, :, :1]                          # cloud 1: two distinct points alternating
    x[1, :, 0::2] = x[1, :, 1:2].clone() * 0 + x[1, :, 2:3]
    gy = torch.from_numpy(synth.normal((B, O), 402)).to(DEV)
    rm, _ = L.weight_images(W, want_tr=False)
    y, arg = L.stage_linear_amax(x, rm, O)
    assert bool((arg[0] == 0).all()) and int(arg[1].max()) <= 1
    dx, dW = L.stage_amax_bwd(x, arg, gy, W)
    dx2, dW2 = L.stage_amax_bwd(x, arg, gy, W)
    assert torch.equal(dx, dx2) and torch.equal(dW, dW2)
    # float64 from the kernel's own arg-max points
    dxr = torch.zeros(B, 128, N, dtype=torch.float64, device=DEV)
    contrib = gy.double().unsqueeze(-1) * W.double().unsqueeze(0)                     # (B, O, 128)
    dxr.permute(0, 2, 1).scatter_add_(1, arg.long().unsqueeze(-1).expand(-1, -1, 128), contrib)
    cols = torch.gather(x.double().permute(0, 2, 1), 1, arg.long().unsqueeze(-1).expand(-1, -1, 128))   # (B, O, 128)
    dWr = (gy.double().unsqueeze(-1) * cols).sum(0)
    assert _rel(dx, dxr) <= 1e-5 and _rel(dW, dWr) <= 1e-5   # (fp32 sums of up to O terms in a fixed order)

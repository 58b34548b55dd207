"""1x1 convolutions over 128 input channels on the HIP kernels of csrc/linear.hip, behind two autograd nodes:

* `ffn(x, w1, w2)`       = Conv1d(128->H) -> LeakyReLU(0.2) -> Conv1d(H->128), the feed-forward part of the reference's
                           attention layers (models/attention.py:187-192, `self.ff`), H a multiple of 256;
* `linear_max(x, w)`     = Conv1d(128->O)(x).max(dim=-1)[0], the pooled heads of the classification trunk
                           (models/cls_model.py:113, 136, 144), O a multiple of 32 -- the (B, O, N) tensor is never built.

x is channel-major (B, 128, N) as the modules hold it; weights are the Conv1d weights (out, in, 1).  fp32 in and out;
products on the bf16 matrix cores with split fp32 operands (fp32-equivalent, csrc/tri_dev.h).  No CPU path.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .ops import _f32c, _need_gpu, _p, _stream

LIN_PLAIN, LIN_LEAKY, LIN_LEAKY_MASK, LIN_LEAKY_BITS, LIN_LEAKY_MASK_BITS = 0, 1, 2, 4, 5   # include/samble.h SAMBLE_LIN_*


def weight_images(W: torch.Tensor, want_rm: bool = True, want_tr: bool = True, transposed: bool = False):
    """W (O, C <= 128) -> (row image | None, transposed image | None) as uint8 tensors (include/samble.h: operand images;
    a narrower W is padded with zero columns).  transposed: W is given as (128, O) and the images are those of W^T (O, 128)
    -- no transposing copy in front of the launch."""
    _need_gpu(W)
    W = _f32c(W)
    if transposed and not _lib.query("samble_linear_two_plane_build"):
        # a three-bf16-plane A/B build of csrc/linear.hip has no transposed-weight entry: the copy it saves, then the plain one
        W, transposed = W.t().contiguous(), False
    if transposed:
        C, O = W.shape
        assert C == 128
    else:
        if W.shape[1] < 128:
            W = torch.nn.functional.pad(W, (0, 128 - W.shape[1]))
        O, C = W.shape
    with torch.cuda.device(W.device):
        nbytes = _lib.query("samble_linear_image_bytes", O)
        rm = torch.empty(nbytes, dtype=torch.uint8, device=W.device) if want_rm else None
        tr = torch.empty(nbytes, dtype=torch.uint8, device=W.device) if want_tr else None
        _lib.call("samble_linear_weight_images_t_f32" if transposed else "samble_linear_weight_images_f32", W.data_ptr(), O,
                  C, _p(rm), _p(tr), _stream())
    return rm, tr


def ffn_weight_images(W1: torch.Tensor, W2: torch.Tensor):
    """The four images a feed-forward layer needs, in one launch: W1 (H, 128) -> (row, transposed); W2 (128, H) -> the (row,
    transposed) images of W2^T, read from W2 as the module holds it."""
    if not _lib.query("samble_linear_two_plane_build"):
        return (*weight_images(W1), *weight_images(W2, transposed=True))
    _need_gpu(W1, W2)
    W1, W2 = _f32c(W1), _f32c(W2)
    H = W1.shape[0]
    assert W1.shape == (H, 128) and W2.shape == (128, H)
    with torch.cuda.device(W1.device):
        nbytes = _lib.query("samble_linear_image_bytes", H)
        imgs = [torch.empty(nbytes, dtype=torch.uint8, device=W1.device) for _ in range(4)]
        _lib.call("samble_linear_weight_images_pair_f32", W1.data_ptr(), H, imgs[0].data_ptr(), imgs[1].data_ptr(), W2.data_ptr(),
                  H, imgs[2].data_ptr(), imgs[3].data_ptr(), _stream())
    return tuple(imgs)


def stage_linear_fwd(x: torch.Tensor, w_rm: torch.Tensor, O: int, epilogue: int = LIN_PLAIN, ref=None, bits=None):
    """x (B,128,N) -> (B,N,O) point-major rows: epilogue(W x).  LIN_LEAKY_BITS returns (out, sign words): one bit per value
    of the activation's sign; LIN_LEAKY_MASK_BITS takes them as `bits` where LIN_LEAKY_MASK takes the activation as `ref`
    (bit-identical results, 1/32 of the bytes)."""
    _need_gpu(x, w_rm, ref, bits)
    x = _f32c(x)
    B, C, N = x.shape
    with torch.cuda.device(x.device):
        out = torch.empty((B, N, O), dtype=torch.float32, device=x.device)
        if ref is not None and (ref.shape != out.shape or not ref.is_contiguous() or ref.dtype != torch.float32):
            raise ValueError("ref must be a contiguous fp32 (B, N, O) tensor")
        side = ref
        if epilogue == LIN_LEAKY_BITS:
            bits = torch.empty(_lib.query("samble_linear_sign_bytes", B, N, O), dtype=torch.uint8, device=x.device)
        if epilogue in (LIN_LEAKY_BITS, LIN_LEAKY_MASK_BITS):
            if bits is None or bits.dtype != torch.uint8 or bits.numel() != _lib.query("samble_linear_sign_bytes", B, N, O):
                raise ValueError("bits must be the sign words LIN_LEAKY_BITS returned for this shape")
            side = bits
        _lib.call("samble_linear_fwd_tri_f32", x.data_ptr(), C * N, B, C, N, w_rm.data_ptr(), O, int(epilogue), _p(side),
                  out.data_ptr(), out.stride(0), out.stride(1), _stream())
    return (out, bits) if epilogue == LIN_LEAKY_BITS else out


def chain_supported(x: torch.Tensor, H: int) -> bool:
    """The one-sweep kernel (8, 4 or 2 waves per workgroup by launch size) needs 128 channels and the two-plane build."""
    return CHAIN and x.shape[1] == 128 and bool(_lib.query("samble_linear_two_plane_build"))


CHAIN = __import__("os").environ.get("SAMBLE_LIN_CHAIN", "1") != "0"   # "0": lin_fwd + lin_dx as two launches (A/B runs)


def stage_linear_chain(x: torch.Tensor, wa_rm: torch.Tensor, wb_tr: torch.Tensor, H: int, epilogue: int, bits=None,
                       residual=None, want_mid: bool = True):
    """x (B,128,N) -> (out (B,128,N) = Wb-contraction of mid [+ residual], mid (B,N,H) = epilogue(Wa x) | None, sign words):
    lin_fwd and lin_dx in one sweep (csrc/linear.hip lin_chain); epilogue LIN_LEAKY_BITS writes the sign words,
    LIN_LEAKY_MASK_BITS reads `bits`."""
    _need_gpu(x, wa_rm, wb_tr, bits, residual)
    x = _f32c(x)
    B, C, N = x.shape
    assert C == 128
    with torch.cuda.device(x.device):
        nb = _lib.query("samble_linear_sign_bytes", B, N, H)
        if epilogue == LIN_LEAKY_BITS:
            bits = torch.empty(nb, dtype=torch.uint8, device=x.device)
        if bits is None or bits.dtype != torch.uint8 or bits.numel() != nb:
            raise ValueError("bits must be the sign words of this shape")
        if residual is not None:
            assert residual.shape == x.shape and residual.is_contiguous() and residual.dtype == torch.float32
        mid = torch.empty((B, N, H), dtype=torch.float32, device=x.device) if want_mid else None
        out = torch.empty((B, 128, N), dtype=torch.float32, device=x.device)
        _lib.call("samble_linear_chain_f32", x.data_ptr(), C * N, B, N, wa_rm.data_ptr(), wb_tr.data_ptr(), H, int(epilogue),
                  _p(mid), N * H, H, bits.data_ptr(), out.data_ptr(), 128 * N, _p(residual), _stream())
    return out, mid, bits


def stage_linear_amax(x: torch.Tensor, w_rm: torch.Tensor, O: int):
    """x (B,128,N) -> (y (B,O) = max over the points of W x, arg (B,O) int32 = the first point that reaches it)."""
    _need_gpu(x, w_rm)
    x = _f32c(x)
    B, C, N = x.shape
    with torch.cuda.device(x.device):
        y = torch.empty((B, O), dtype=torch.float32, device=x.device)
        arg = torch.empty((B, O), dtype=torch.int32, device=x.device)
        nbytes = _lib.query("samble_linear_amax_workspace_bytes", B, N, O)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        _lib.call("samble_linear_amax_fwd_tri_f32", x.data_ptr(), C * N, B, C, N, w_rm.data_ptr(), O, y.data_ptr(),
                  arg.data_ptr(), ws.data_ptr(), nbytes, _stream())
    return y, arg


def stage_linear_dx(g: torch.Tensor, w_tr: torch.Tensor, O: int, C: int = 128, residual=None, out=None) -> torch.Tensor:
    """g (B,N,O) point-major -> (B,C,N): W^T g [+ residual (B,C,N), added by the kernel's epilogue; out: the tensor to
    write, which may be `residual` itself (in-place accumulation)]."""
    _need_gpu(g, w_tr, residual, out)
    g = _f32c(g)
    B, N, Og = g.shape
    assert Og == O
    if residual is not None:
        assert residual.shape == (B, C, N) and residual.is_contiguous() and residual.dtype == torch.float32
    with torch.cuda.device(g.device):
        dx = out if out is not None else torch.empty((B, C, N), dtype=torch.float32, device=g.device)
        assert dx.shape == (B, C, N) and dx.is_contiguous() and dx.dtype == torch.float32
        _lib.call("samble_linear_dx_tri_f32", g.data_ptr(), g.stride(0), g.stride(1), w_tr.data_ptr(), O, B, C, N,
                  dx.data_ptr(), C * N, _p(residual), _stream())
    return dx


def stage_linear_dw(g: torch.Tensor, x: torch.Tensor, O: int, transposed: bool = False) -> torch.Tensor:
    """g (B,N,O), x (B,C,N) -> dW (O,C) = sum over clouds and points of g^T x^T (deterministic); transposed (C = 128):
    the same sum as (128, O)."""
    _need_gpu(g, x)
    g, x = _f32c(g), _f32c(x)
    B, N, Og = g.shape
    C = x.shape[1]
    assert Og == O and x.shape == (B, C, N) and (C == 128 or not transposed)
    with torch.cuda.device(g.device):
        dW = torch.empty((128, O) if transposed else (O, 128), dtype=torch.float32, device=g.device)
        nbytes = _lib.query("samble_linear_dw_workspace_bytes", B, N, O)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
        _lib.call("samble_linear_dw_t_tri_f32" if transposed else "samble_linear_dw_tri_f32", g.data_ptr(), g.stride(0),
                  g.stride(1), x.data_ptr(), C * N, B, C, N, O, dW.data_ptr(), ws.data_ptr(), nbytes, _stream())
    return dW if (C == 128 or transposed) else dW[:, :C]


def stage_linear_fwd_cm(x: torch.Tensor, w_rm: torch.Tensor, O: int, out=None, accumulate: bool = False) -> torch.Tensor:
    """x (B,C<=128,N) channel-major -> (B,O,N) channel-major: W x [added to `out` when accumulate]."""
    _need_gpu(x, w_rm, out)
    x = _f32c(x)
    B, C, N = x.shape
    with torch.cuda.device(x.device):
        if out is None:
            assert not accumulate
            out = torch.empty((B, O, N), dtype=torch.float32, device=x.device)
        assert out.shape == (B, O, N) and out.is_contiguous() and out.dtype == torch.float32
        _lib.call("samble_linear_fwd_cm_f32", x.data_ptr(), C * N, B, C, N, w_rm.data_ptr(), O, int(accumulate),
                  out.data_ptr(), O * N, _stream())
    return out


def stage_linear_dw_cm(g: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """g (B,O,N), x (B,C<=128,N), both channel-major -> dW (O,C) = sum over clouds and points of g x^T (deterministic)."""
    _need_gpu(g, x)
    g, x = _f32c(g), _f32c(x)
    B, O, N = g.shape
    C = x.shape[1]
    assert x.shape == (B, C, N)
    with torch.cuda.device(g.device):
        dW = torch.empty((O, 128), dtype=torch.float32, device=g.device)
        nbytes = _lib.query("samble_linear_dw_workspace_bytes", B, N, O)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
        _lib.call("samble_linear_dw_cm_f32", g.data_ptr(), O * N, x.data_ptr(), C * N, B, C, N, O, dW.data_ptr(),
                  ws.data_ptr(), nbytes, _stream())
    return dW if C == 128 else dW[:, :C]


def stage_amax_bwd(x: torch.Tensor, arg: torch.Tensor, gy: torch.Tensor, W: torch.Tensor, into: Optional[torch.Tensor] = None):
    """Backward of stage_linear_amax: -> (dx (B,128,N), zero outside the arg-max columns; dW (O,128)).
    into: a contiguous float32 (B,128,N) tensor holding ANOTHER gradient of x -- the arg-max columns are added to it in
    place and it is returned as dx (no zero fill, no separate add)."""
    _need_gpu(x, arg, gy, W, into)
    x, gy, W = _f32c(x), _f32c(gy), _f32c(W)
    B, C, N = x.shape
    O = W.shape[0]
    with torch.cuda.device(x.device):
        if into is not None:
            assert into.shape == x.shape and into.dtype == torch.float32 and into.is_contiguous()
        dx = into if into is not None else torch.zeros_like(x)
        dW = torch.empty((O, 128), dtype=torch.float32, device=x.device)
        nbytes = _lib.query("samble_amax_bwd_workspace_bytes", B, N, O)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        _lib.call("samble_amax_bwd_f32", x.data_ptr(), C * N, B, C, N, arg.data_ptr(), gy.data_ptr(), W.data_ptr(), O,
                  dx.data_ptr(), C * N, dW.data_ptr(), ws.data_ptr(), nbytes, _stream())
    return dx, dW


class _FFN(torch.autograd.Function):
    """Conv1d(128->H, no bias) -> LeakyReLU(0.2) -> Conv1d(H->128, no bias) on (B,128,N); the hidden activation lives
    point-major (B,N,H) and is the only tensor kept for the backward besides the input."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, w1, w2):
        H = w1.shape[0]
        W1 = w1.reshape(H, 128)
        w1_rm, w1_tr, w2t_rm, w2t_tr = ffn_weight_images(W1, w2.reshape(128, H))   # (W2^T's images, read from W2 as it is)
        if chain_supported(x, H):                                     # both products in one sweep
            y, hr, bits = stage_linear_chain(x, w1_rm, w2t_tr, H, LIN_LEAKY_BITS)
        else:
            hr, bits = stage_linear_fwd(x, w1_rm, H, LIN_LEAKY_BITS)  # leaky(W1 x), (B,N,H), and its signs as bits
            y = stage_linear_dx(hr, w2t_tr, H)                        # y[c][n] = sum_j W2[c][j] hr[n][j]
        ctx.save_for_backward(x, hr, w1_tr, w2t_rm, bits)
        ctx.H = H
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, hr, w1_tr, w2t_rm, bits = ctx.saved_tensors
        H = ctx.H
        dy = _f32c(dy)
        if chain_supported(dy, H) and ctx.needs_input_grad[0]:
            dx, dh, _ = stage_linear_chain(dy, w2t_rm, w1_tr, H, LIN_LEAKY_MASK_BITS, bits=bits)
        else:
            dh = stage_linear_fwd(dy, w2t_rm, H, LIN_LEAKY_MASK_BITS, bits=bits)  # (W2^T dy) * leaky'(h): sign(hr) = sign(h)
            dx = stage_linear_dx(dh, w1_tr, H) if ctx.needs_input_grad[0] else None
        dw1 = stage_linear_dw(dh, x, H).reshape(H, 128, 1) if ctx.needs_input_grad[1] else None
        dw2 = stage_linear_dw(hr, dy, H, transposed=True).reshape(128, H, 1) if ctx.needs_input_grad[2] else None
        return dx, dw1, dw2


class _Linear(torch.autograd.Function):
    """out[b][n][o] = sum_c W[o][c] x[b][c][n]: x channel-major (B, C <= 128, N) -> point-major (B, N, O), O a multiple of
    128 (the per-point projections of EdgeConv, models/embedding.py:20-28)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, W):
        O, C = W.shape
        w_rm, w_tr = weight_images(W)
        ctx.save_for_backward(x, w_tr)
        ctx.dims = (O, C)
        return stage_linear_fwd(x, w_rm, O)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x, w_tr = ctx.saved_tensors
        O, C = ctx.dims
        g = _f32c(g)
        dx = stage_linear_dx(g, w_tr, O, C) if ctx.needs_input_grad[0] else None
        dW = stage_linear_dw(g, x, O) if ctx.needs_input_grad[1] else None
        return dx, dW


def linear_supported(x: torch.Tensor, W: torch.Tensor) -> bool:
    return (x.is_cuda and x.dim() == 3 and x.dtype == torch.float32 and W.dim() == 2 and W.shape[1] == x.shape[1]
            and 1 <= x.shape[1] <= 128 and W.shape[0] % 128 == 0)


def linear_rows(x: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """x (B,C,N), W (O,C) -> (B,N,O) = (W x)^T per cloud."""
    if not x.is_cuda:
        raise _lib.SambleError("samble_amd.linear.linear_rows runs on the GPU only (no CPU fallback)")
    return _Linear.apply(x, W)


class _LinearMax(torch.autograd.Function):
    """Conv1d(128->O, no bias)(x).max(dim=-1)[0]: values only; the gradient goes to the arg-max column (torch.max(dim)
    routes it to the one index it returned: the first maximum on the CPU reference)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, w, forced_arg=None):
        """forced_arg (B, O), parity-test hook like the samplers' `forced_idx`: the point each (cloud, output) routes its
        gradient to -- where two points tie to fp32 rounding, two valid evaluations pick different ones -- instead of this
        evaluation's own arg-max; the returned values stay this evaluation's maxima."""
        O = w.shape[0]
        W = _f32c(w.reshape(O, 128))
        w_rm, _ = weight_images(W, want_tr=False)
        y, arg = stage_linear_amax(x, w_rm, O)
        if forced_arg is not None:
            arg = forced_arg.to(device=arg.device, dtype=arg.dtype).contiguous()
        ctx.save_for_backward(x, arg, W)
        ctx.mark_non_differentiable(arg)
        return y, arg

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy, _):
        x, arg, W = ctx.saved_tensors
        dx, dW = stage_amax_bwd(x, arg, gy, W)
        return dx, dW.reshape(W.shape[0], 128, 1), None


class _LinearMaxSplit(torch.autograd.Function):
    """(x, w) -> (Conv1d(128->O, no bias)(x).max(dim=-1)[0], arg, x): the pooled head of a level TOGETHER with the tensor
    that goes on to the sampler (models/cls_model.py:113, 132-136: the level's features feed both).  Forward is _LinearMax
    plus a view; the point is the backward: autograd hands over both gradients at once, so the head's sparse gradient --
    O columns per cloud -- is added INTO the gradient that came back from the sampler, where two separate consumers cost a
    zero-filled (B,128,N) tensor and a dense add per level (stock fill + add kernels, 0.05 ms per block step).  Same sums:
    column = other gradient + the head's column sum."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, w, forced_arg=None):
        O = w.shape[0]
        W = _f32c(w.reshape(O, 128))
        w_rm, _ = weight_images(W, want_tr=False)
        y, arg = stage_linear_amax(x, w_rm, O)
        if forced_arg is not None:
            arg = forced_arg.to(device=arg.device, dtype=arg.dtype).contiguous()
        ctx.save_for_backward(x, arg, W)
        ctx.mark_non_differentiable(arg)
        return y, arg, x.view_as(x)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy, _, gx):
        x, arg, W = ctx.saved_tensors
        into = None
        if gx is not None:
            # the sampler's gradient is this node's to keep (a fresh tensor from the node behind): accumulate in place unless
            # it is not a plain dense float32 tensor
            into = gx if (gx.dtype == torch.float32 and gx.is_contiguous() and not gx.requires_grad) else gx.float().contiguous()
        if gy is None:
            return into, None, None
        dx, dW = stage_amax_bwd(x, arg, gy, W, into=into)
        return dx, dW.reshape(W.shape[0], 128, 1), None


class _PointwiseCM(torch.autograd.Function):
    """y (B,128,N) = sum_i W[:, 128 i : 128 i + 128] x_i: a bias-free Conv1d(128 k -> 128, kernel 1) on the channel-wise
    concatenation of k tensors (B,128,N) that is never formed (reference models/upsample.py:142-150: `conv` k = 1,
    `res_conv(torch.cat((pcd_up, interpolated), dim=1))` k = 2).  Channel-major in and out: forward and input gradients
    on lin_fwd's channel-major store, weight gradients on lin_dw's channel-major staging."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, w, *xs):
        W = w.reshape(w.shape[0], -1)
        parts = [_f32c(W[:, 128 * i:128 * i + 128]) for i in range(len(xs))]
        y = None
        for i, (x, Wi) in enumerate(zip(xs, parts)):
            rm, _ = weight_images(Wi, want_tr=False)
            y = stage_linear_fwd_cm(x, rm, W.shape[0], out=y, accumulate=i > 0)
        ctx.save_for_backward(*xs, *parts)
        ctx.k = len(xs)
        ctx.wshape = w.shape
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        k = ctx.k
        saved = ctx.saved_tensors   # (read once: torch.utils.checkpoint's unpack hooks refuse a second access)
        xs, parts = saved[:k], saved[k:]
        gy = _f32c(gy)
        dxs = []
        for i in range(k):
            if ctx.needs_input_grad[1 + i]:
                rm_t, _ = weight_images(parts[i], want_tr=False, transposed=True)     # the image of W_i^T, from W_i as it is
                dxs.append(stage_linear_fwd_cm(gy, rm_t, 128))
            else:
                dxs.append(None)
        dw = None
        if ctx.needs_input_grad[0]:
            dws = [stage_linear_dw_cm(gy, x) for x in xs]
            dw = (dws[0] if k == 1 else torch.cat(dws, dim=1)).reshape(ctx.wshape)
        return (dw, *dxs)


def pointwise_cm_supported(w: torch.Tensor, *xs: torch.Tensor) -> bool:
    return (len(xs) in (1, 2) and w.dim() == 3 and w.shape[0] == 128 and w.shape[2] == 1 and w.shape[1] == 128 * len(xs)
            and all(x.is_cuda and x.dim() == 3 and x.shape[1] == 128 and x.dtype == torch.float32 and x.shape == xs[0].shape
                    for x in xs) and w.is_cuda and bool(_lib.query("samble_linear_two_plane_build")))


def pointwise_cm(w: torch.Tensor, *xs: torch.Tensor) -> torch.Tensor:
    """Conv1d(128 k -> 128, no bias)(cat(xs, dim=1)) with the Conv1d weight w (128, 128 k, 1)."""
    if not w.is_cuda:
        raise _lib.SambleError("samble_amd.linear.pointwise_cm runs on the GPU only (no CPU fallback)")
    return _PointwiseCM.apply(w, *xs)


MAX_OUT_CHANNELS = 4096   # csrc/abi.hip lin_shape_ok: wider layers take the stock Conv1d path, as documented


def ffn_supported(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor) -> bool:
    return (x.is_cuda and x.dim() == 3 and x.shape[1] == 128 and w1.dim() == 3 and w2.dim() == 3 and w1.shape[1] == 128
            and w1.shape[2] == 1 and w2.shape[2] == 1 and w2.shape[0] == 128 and w2.shape[1] == w1.shape[0]
            and w1.shape[0] % 256 == 0 and w1.shape[0] <= MAX_OUT_CHANNELS and x.dtype == torch.float32)


def ffn(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """models/attention.py:187-192 `self.ff(x)` with the two Conv1d weights."""
    if not x.is_cuda:
        raise _lib.SambleError("samble_amd.linear.ffn runs on the GPU only (no CPU fallback)")
    return _FFN.apply(x, w1, w2)


def linear_max_supported(x: torch.Tensor, w: torch.Tensor) -> bool:
    return (x.is_cuda and x.dim() == 3 and x.shape[1] == 128 and w.dim() == 3 and w.shape[1] == 128 and w.shape[2] == 1
            and w.shape[0] % 32 == 0 and w.shape[0] <= MAX_OUT_CHANNELS and x.shape[2] + 3 * w.shape[0] < 36000
            and x.dtype == torch.float32)


def linear_max(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """models/cls_model.py:113 `conv(x).max(dim=-1)[0]` -> (B, O)."""
    if not x.is_cuda:
        raise _lib.SambleError("samble_amd.linear.linear_max runs on the GPU only (no CPU fallback)")
    return _LinearMax.apply(x, w)[0]

"""Host-side mirror of the reference's `utils/ops.py` for the sampler path, on the HIP kernels.

Same function names, argument meaning and error behaviour as the reference (file:line cited per
function) so call sites and tests read alike; the arithmetic runs in libsamble_hip.so through
`samble_amd._lib` (no CPU path: tensors must live on the GPU).  The lower half of the file holds
the stage-level wrappers (`stage_*`) that `samble_amd.downsample` chains without materialising the
reference's dense (B,N,N) tensors.
"""
from __future__ import annotations

import numbers
import os
from typing import List, Optional, Tuple

import torch

from . import _lib

SCORE_MODES = {"sparse_col_sum": 0, "sparse_col_avg": 1, "sparse_col_sqr": 2, "sparse_row_sum": 3,
               "sparse_row_std": 4}
SAMPLE_MODES = {"topk": 0, "uniform": 1, "random": 2, "top_raw": 3, "bottom_raw": 4}
KNN_SIZES = (1, 3, 8, 16, 20, 32, 40, 64)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.SambleError("samble_amd ops need GPU tensors (there is no CPU fallback)")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor) -> torch.Tensor:
    t = t if t.dtype == torch.float32 else t.float()
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------
# stage wrappers
# ------------------------------------------------------------------------------------------------
KNN_FP32_MFMA, KNN_TWO_KERNEL = 1, 2  # include/samble.h SAMBLE_KNN_*


KNN_LIST_SIZES = (1, 3, 8, 16, 20, 32, 40, 64)   # neighbour-list lengths the kNN kernels are built for


def _knn_expression(xq: torch.Tensor, xk: torch.Tensor, k: int):
    """utils/ops.py:23-43 as it stands, on the device: centre on the queries' mean, divide by their mean channel deviation,
    cdist, top-k.  (B,C,Nq), (B,C,Nk) -> idx (B,Nq,k) int32 nearest first, positive distance (B,Nq,k)."""
    a, b = xq.transpose(1, 2), xk.transpose(1, 2)
    mean = a.mean(dim=1, keepdim=True)
    a, b = a - mean, b - mean
    std = torch.std(a, dim=1, keepdim=True).mean(dim=2, keepdim=True)
    dist, idx = (-torch.cdist(a / std, b / std, compute_mode="donot_use_mm_for_euclid_dist")).topk(k=k, dim=-1)
    return idx.int(), -dist


def stage_knn(xq: torch.Tensor, xk: torch.Tensor, k: int, want_dist: bool = False, variant: Optional[int] = None):
    """xq (B,C,Nq), xk (B,C,Nk) channel-major -> idx (B,Nq,k) int32 nearest first
    [, positive reference-normalised distance (B,Nq,k)].  variant: kernel choice (KNN_*); None = the
    feature-space kNN (C = 128) follows MATRIX_MODE."""
    _need_gpu(xq, xk)
    xq, xk = _f32c(xq), _f32c(xk)
    B, C, Nq = xq.shape
    Nk = xk.shape[2]
    if xk.shape[0] != B or xk.shape[1] != C:
        raise ValueError("knn: the two point sets must share batch and channel sizes")
    if k not in KNN_LIST_SIZES:
        # the kernels keep lists of these sizes; the reference takes any k (utils/ops.py:17-44).  Lists are nearest
        # first, so any other k is the head of the next size's list; beyond the kernels' range (k > 64, or a key set
        # shorter than the next size) the reference's own expression runs in torch on the device
        if not 1 <= k <= Nk:
            raise ValueError(f"knn: need 1 <= k <= {Nk} keys, got k = {k}")
        k2 = next((s for s in KNN_LIST_SIZES if s >= k), None)
        if k2 is None or k2 > Nk:
            idx, dist = _knn_expression(xq, xk, k)
        else:
            res = stage_knn(xq, xk, k2, want_dist=want_dist, variant=variant)
            idx, dist = (res if want_dist else (res, None))
            idx = idx[:, :, :k].contiguous()
            dist = dist[:, :, :k].contiguous() if want_dist else None
        return (idx, dist) if want_dist else idx
    with torch.cuda.device(xq.device):
        idx = torch.empty((B, Nq, k), dtype=torch.int32, device=xq.device)
        dist = torch.empty((B, Nq, k), dtype=torch.float32, device=xq.device) if want_dist else None
        if variant is None:
            variant = 0 if MATRIX_MODE == "tri" else KNN_FP32_MFMA
        nbytes = _lib.query("samble_knn_workspace_bytes", B, C, Nq, Nk, k, variant)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xq.device)
        _lib.call("samble_knn_f32", xq.data_ptr(), C * Nq, Nq, xk.data_ptr(), C * Nk, Nk, B, C, k, variant,
                  idx.data_ptr(), _p(dist), ws.data_ptr(), nbytes, _stream())
    return (idx, dist) if want_dist else idx


def _three_weights(w_qkv):
    """w_qkv as (Wq, Wk, Wv) tensors of their own -> three contiguous fp32 (C,C) matrices; a single tensor -> None."""
    if not isinstance(w_qkv, (tuple, list)):
        return None
    ws = tuple(_f32c(w.reshape(w.shape[0], w.shape[1])) for w in w_qkv)
    if len(ws) != 3 or any(w.shape != ws[0].shape or w.shape[0] != w.shape[1] for w in ws):
        raise ValueError("three weights: (Wq, Wk, Wv), each (C, C)")
    return ws


def stage_proj_fwd(x: torch.Tensor, tokens: torch.Tensor, w_qkv, images: str = "", q_only: bool = False):
    """x (B,C,N), tokens (C,nt), w_qkv (3C,C) -> qkv (B,N+nt,3C) point-major rows [Q|K|V].
    w_qkv may be the tuple (Wq, Wk, Wv) of (C,C) tensors: with images the kernels read the three where they are (no
    concatenation launch); without images they are concatenated here.
    images "fwd" / "fwd+bwd" (MATRIX_MODE "tri"): -> (qkv, operand images as stage_tri_split_qkv returns them), written by
    the projection kernel itself (the split pass then covers only the tiles with token rows / a ragged end).
    q_only (with images): the K / V columns of qkv's point rows stay unwritten where the images carry the tile
    (include/samble.h SAMBLE_PROJ_ROWS_Q_ONLY) -- for callers that read the Q rows, the token rows and the images only."""
    w3 = _three_weights(w_qkv)
    if w3 is not None and not images:
        w_qkv, w3 = torch.cat(w3, dim=0), None
    _need_gpu(x, tokens, *(w3 or (w_qkv,)))
    x, tokens = _f32c(x), _f32c(tokens)
    w_qkv = None if w3 else _f32c(w_qkv)
    B, C, N = x.shape
    nt = tokens.shape[1]
    if images:
        if MATRIX_MODE != "tri" or images not in ("fwd", "fwd+bwd"):
            raise ValueError("operand images come with MATRIX_MODE 'tri' only: images in ('fwd', 'fwd+bwd')")
        with torch.cuda.device(x.device):
            qkv = torch.empty((B, N + nt, 3 * C), dtype=torch.float32, device=x.device)
            img = lambda rows, tr: torch.empty(_lib.query("samble_tri_image_bytes", B, rows, tr), dtype=torch.uint8,
                                               device=x.device)
            q_img, k_img, v_img = img(N, 0), img(N + nt, 0), img(N + nt, 1)
            k_tr, v_rm = (img(N + nt, 1), img(N + nt, 0)) if images == "fwd+bwd" else (None, None)
            # with the backward pair also the transposed image of W, for stage_proj_bwd(w_tr=...): last of the tuple
            w_tr = torch.empty(_lib.query("samble_proj_w_image_bytes"), dtype=torch.uint8, device=x.device) \
                if images == "fwd+bwd" else None
            nbytes = _lib.query("samble_proj_fwd_tri_workspace_bytes")
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            wp = [w.data_ptr() for w in w3] if w3 else [w_qkv.data_ptr(), None, None]
            _lib.call("samble_proj_fwd_split_tri_f32", x.data_ptr(), C * N, B, C, N, tokens.data_ptr(), nt, *wp,
                      qkv.data_ptr(), qkv.stride(0), qkv.stride(1), q_img.data_ptr(), k_img.data_ptr(), v_img.data_ptr(),
                      _p(k_tr), _p(v_rm), 1 if q_only else 0, _p(w_tr), ws.data_ptr(), nbytes, _stream())
        return qkv, ((q_img, k_img, v_img, k_tr, v_rm, w_tr) if images == "fwd+bwd" else (q_img, k_img, v_img))
    with torch.cuda.device(x.device):
        qkv = torch.empty((B, N + nt, 3 * C), dtype=torch.float32, device=x.device)
        tri = MATRIX_MODE == "tri"
        nbytes = _lib.query("samble_proj_fwd_tri_workspace_bytes") if tri else 8 * 384 * 4
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        _lib.call("samble_proj_fwd_tri_f32" if tri else "samble_proj_fwd_f32", x.data_ptr(), C * N, B, C, N,
                  tokens.data_ptr(), nt, w_qkv.data_ptr(), qkv.data_ptr(), qkv.stride(0), qkv.stride(1), ws.data_ptr(),
                  nbytes, _stream())
    return qkv


def stage_proj_bwd(dqkv, x, tokens, w_qkv, need_dx: bool, need_dw: bool, w_tr: Optional[torch.Tensor] = None,
                   dx_residual: Optional[torch.Tensor] = None):
    """-> (dx (B,C,N) | None, dW (3C,C) | None, dtokens (C,nt) | None).  w_tr: the transposed operand image of w_qkv as
    stage_proj_fwd(images="fwd+bwd") returned it (MATRIX_MODE "tri"; the weights must not have changed since).
    w_qkv may be the tuple (Wq, Wk, Wv): read where they are when w_tr comes along, concatenated here otherwise.
    dx_residual (B,C,N) (MATRIX_MODE "tri"): added to dx by the kernel's epilogue -- the gradient that reaches x along a
    residual branch beside the projection."""
    w3 = _three_weights(w_qkv)
    if w3 is not None and (w_tr is None or MATRIX_MODE != "tri"):
        w_qkv, w3 = torch.cat(w3, dim=0), None
    _need_gpu(dqkv, x, tokens, *(w3 or (w_qkv,)))
    x, tokens = _f32c(x), _f32c(tokens)
    w_qkv = None if w3 else _f32c(w_qkv)
    if dqkv.stride(2) != 1:
        dqkv = dqkv.contiguous()
    B, C, N = x.shape
    nt = tokens.shape[1]
    with torch.cuda.device(x.device):
        dx = torch.empty_like(x) if need_dx else None
        dw = torch.empty((3 * C, C), dtype=torch.float32, device=x.device) if need_dw else None
        dtok = torch.empty_like(tokens) if need_dw else None
        tri = MATRIX_MODE == "tri"
        nbytes = _lib.query("samble_proj_bwd_tri_workspace_bytes" if tri else "samble_proj_workspace_bytes", B, N)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        if tri:
            _lib.call("samble_proj_bwd_tri_f32", dqkv.data_ptr(), dqkv.stride(0), dqkv.stride(1), x.data_ptr(), C * N, B, C, N,
                      tokens.data_ptr(), nt, *([w.data_ptr() for w in w3] if w3 else [w_qkv.data_ptr(), None, None]),
                      _p(w_tr), _p(dx), C * N, _p(dw), _p(dtok), _p(_f32c(dx_residual)) if dx_residual is not None else None,
                      ws.data_ptr(), nbytes, _stream())
        else:
            if dx_residual is not None:
                raise ValueError("dx_residual comes with MATRIX_MODE 'tri' only")
            _lib.call("samble_proj_bwd_f32", dqkv.data_ptr(), dqkv.stride(0), dqkv.stride(1), x.data_ptr(), C * N, B, C, N,
                      tokens.data_ptr(), nt, w_qkv.data_ptr(), _p(dx), C * N, _p(dw), _p(dtok), ws.data_ptr(), nbytes,
                      _stream())
    return dx, dw, dtok


def stage_n2p_attn_fwd(qkv: torch.Tensor, nn_idx: torch.Tensor, heads: int, diff: bool, want_att: bool = False,
                       residual: Optional[torch.Tensor] = None):
    """qkv (B,N,3C) point-major [Q|K|V], nn_idx (B,N,K) int32 -> (B,C,N) attention output
    [, (B,N,K) softmax probabilities when want_att (single head only)].  residual (B,C,N): added on the way out."""
    _need_gpu(qkv, nn_idx, residual)
    if residual is not None:
        residual = _f32c(residual)
    B, N, C3 = qkv.shape
    C = C3 // 3
    with torch.cuda.device(qkv.device):
        out = torch.empty((B, C, N), dtype=torch.float32, device=qkv.device)
        att = torch.empty((B, N, nn_idx.shape[2]), dtype=torch.float32, device=qkv.device) if want_att else None
        _lib.call("samble_n2p_attn_fwd_f32", qkv.data_ptr(), qkv.stride(0), qkv.stride(1), nn_idx.data_ptr(), B, N,
                  nn_idx.shape[2], C, heads, int(bool(diff)), out.data_ptr(), _p(att), _p(residual), _stream())
    return (out, att) if want_att else out


def inverse_neighbors(nn_idx: torch.Tensor):
    """Inverse neighbour lists of a kNN table nn_idx (B,N,K): (order (B*N*K) int32 = edge ids
    e = (b*N + i)*K + k grouped by target b*N + nn[e], ascending e inside a group; offsets (B*N + 1) int32).
    = a stable sort of the table by target, built on the device without sorting (samble_inverse_neighbors); counts (B*N)
    int32 = the in-degrees.
    Precondition: every row of nn_idx holds K distinct indices in [0, N) (what stage_knn returns).  A table that
    breaks it yields lists with fewer than N*K edges per cloud and zero-filled slack -- wrong sums downstream, but
    no out-of-bounds access."""
    _need_gpu(nn_idx)
    B, N, K = nn_idx.shape
    nn_idx = nn_idx.contiguous()
    if nn_idx.dtype != torch.int32:
        nn_idx = nn_idx.to(torch.int32)
    with torch.cuda.device(nn_idx.device):
        order = torch.empty(B * N * K, dtype=torch.int32, device=nn_idx.device)
        offsets = torch.empty(B * N + 1, dtype=torch.int32, device=nn_idx.device)
        counts = torch.empty(B * N, dtype=torch.int32, device=nn_idx.device)
        nbytes = _lib.query("samble_inverse_neighbors_workspace_bytes", B, N)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=nn_idx.device)
        _lib.call("samble_inverse_neighbors", nn_idx.data_ptr(), B, N, K, order.data_ptr(), offsets.data_ptr(),
                  counts.data_ptr(), ws.data_ptr(), nbytes, _stream())
    return order, offsets, counts


def stage_segment_sum_rows(src: torch.Tensor, order: torch.Tensor, offsets: torch.Tensor, K: int, per_edge: bool):
    """out[t] = sum over the incoming edges e of target t of src[e] (per_edge) or src[e // K]; src (*, 64),
    list order: deterministic replacement of index_add_."""
    _need_gpu(src, order, offsets)
    # (rows may be a column block of a wider matrix -- the a half of EdgeConv's [a | b] projection: read where they are)
    if not (src.dim() == 2 and src.dtype == torch.float32 and src.stride(1) == 1 and src.stride(0) % 2 == 0
            and src.data_ptr() % 8 == 0):
        src = _f32c(src)
    T = offsets.numel() - 1
    with torch.cuda.device(src.device):
        out = torch.empty((T, 64), dtype=torch.float32, device=src.device)
        _lib.call("samble_segment_sum_rows_f32", src.data_ptr(), src.stride(0), order.data_ptr(), offsets.data_ptr(), K, 64,
                  int(bool(per_edge)), T, out.data_ptr(), _stream())
    return out


def stage_bn_train(x: torch.Tensor, gamma, beta, running_mean, running_var, momentum: float, eps: float, group=None,
                   act_slope: float = 1.0):
    """nn.BatchNorm1d.forward in training mode on x (B,C,N): -> (y, batch mean (C), 1 / sqrt(batch var + eps) (C), count);
    the running estimates (may be None) get torch's momentum update in place.  csrc/batchnorm.hip.
    group (nn.SyncBatchNorm, reference train_modelnet.py:245-246): the process group whose ranks pool the statistics --
    the per-channel float64 sums and the element count are all-reduced between the two launches; `count` is then the
    device scalar holding the pooled element count (the backward divides by it), None on one rank.
    act_slope: LeakyReLU(act_slope) of the result in the same pass (1.0 = none)."""
    _need_gpu(x, gamma, beta, running_mean, running_var)
    x = _f32c(x)
    B, C, N = x.shape
    with torch.cuda.device(x.device):
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        nbytes = _lib.query("samble_bn_train_workspace_bytes", B, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        if group is None:
            _lib.call("samble_bn_train_fwd_f32", x.data_ptr(), B, C, N, _p(gamma), _p(beta), float(eps), float(momentum),
                      _p(running_mean), _p(running_var), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), float(act_slope),
                      ws.data_ptr(), nbytes, _stream())
            return y, mean, invstd, None
        pooled = torch.empty(2 * C + 1, dtype=torch.float64, device=x.device)
        _lib.call("samble_bn_train_stats_f32", x.data_ptr(), B, C, N, pooled.data_ptr(), ws.data_ptr(), nbytes, _stream())
        torch.distributed.all_reduce(pooled, group=group)
        _lib.call("samble_bn_train_apply_f32", x.data_ptr(), B, C, N, pooled.data_ptr(), _p(gamma), _p(beta), float(eps),
                  float(momentum), _p(running_mean), _p(running_var), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                  float(act_slope), _stream())
    return y, mean, invstd, pooled[2 * C:]


def stage_bn_train_bwd(x: torch.Tensor, dy: torch.Tensor, gamma, mean: torch.Tensor, invstd: torch.Tensor, count=None, group=None,
                       out: Optional[torch.Tensor] = None, beta=None, act_slope: float = 1.0):
    """Backward of stage_bn_train: -> (dx, dgamma, dbeta).  x is the forward's INPUT, mean / invstd what it saved.  With
    `group` the two per-channel sums are all-reduced between the launches and divided by `count` (the forward's pooled
    element count); dgamma / dbeta stay this rank's sums, as torch's SyncBatchNorm leaves them for DDP to average.
    out: where dx goes (may be dy itself).  act_slope != 1 (with beta): dy is the gradient of LeakyReLU(bn(x))."""
    _need_gpu(x, dy, gamma, mean, invstd, beta)
    x, dy = _f32c(x), _f32c(dy)
    B, C, N = x.shape
    with torch.cuda.device(x.device):
        dx = out if out is not None else torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        nbytes = _lib.query("samble_bn_train_workspace_bytes", B, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        if group is None:
            _lib.call("samble_bn_train_bwd_f32", x.data_ptr(), dy.data_ptr(), B, C, N, mean.data_ptr(), invstd.data_ptr(),
                      _p(gamma), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _p(beta), float(act_slope), ws.data_ptr(),
                      nbytes, _stream())
            return dx, dgamma, dbeta
        pooled = torch.empty(2 * C, dtype=torch.float64, device=x.device)
        _lib.call("samble_bn_train_bwd_sums_f32", x.data_ptr(), dy.data_ptr(), B, C, N, mean.data_ptr(), invstd.data_ptr(),
                  pooled.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), _p(gamma), _p(beta), float(act_slope), ws.data_ptr(),
                  nbytes, _stream())
        torch.distributed.all_reduce(pooled, group=group)
        _lib.call("samble_bn_train_bwd_apply_f32", x.data_ptr(), dy.data_ptr(), B, C, N, mean.data_ptr(), invstd.data_ptr(),
                  _p(gamma), pooled.data_ptr(), count.data_ptr(), dx.data_ptr(), _p(beta), float(act_slope), _stream())
    return dx, dgamma, dbeta


def stage_segment_sum_rows_pair(src_edge: torch.Tensor, src_point: torch.Tensor, order: torch.Tensor, offsets: torch.Tensor,
                                K: int):
    """(sum over the incoming edges e of src_edge[e], of src_point[e // K]) per target, in ONE pass over the lists: the two
    reverse-neighbour sums of EdgeConv's backward, bit for bit `stage_segment_sum_rows` twice."""
    _need_gpu(src_edge, src_point, order, offsets)
    ok = lambda t: (t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(0) % 2 == 0
                    and t.data_ptr() % 8 == 0)
    src_edge = src_edge if ok(src_edge) else _f32c(src_edge)
    src_point = src_point if ok(src_point) else _f32c(src_point)
    T = offsets.numel() - 1
    with torch.cuda.device(src_edge.device):
        out_e = torch.empty((T, 64), dtype=torch.float32, device=src_edge.device)
        out_p = torch.empty((T, 64), dtype=torch.float32, device=src_edge.device)
        _lib.call("samble_segment_sum_rows_pair_f32", src_edge.data_ptr(), src_edge.stride(0), src_point.data_ptr(),
                  src_point.stride(0), order.data_ptr(), offsets.data_ptr(), K, 64, T, out_e.data_ptr(), out_p.data_ptr(), _stream())
    return out_e, out_p


def stage_n2p_attn_bwd(qkv: torch.Tensor, nn_idx: torch.Tensor, g: torch.Tensor, heads: int, diff: bool,
                       use_inverse_lists: bool = True) -> torch.Tensor:
    """g (B,C,N) -> dqkv (B,N,3C) of the gather-attention (deterministic)."""
    _need_gpu(qkv, nn_idx, g)
    g = _f32c(g)
    B, N, C3 = qkv.shape
    K = nn_idx.shape[2]
    with torch.cuda.device(qkv.device):
        dqkv = torch.empty_like(qkv)
        nbytes = _lib.query("samble_n2p_attn_bwd_workspace_bytes", B, N, K)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=qkv.device)
        order = offsets = None
        if use_inverse_lists:
            order, offsets, _ = inverse_neighbors(nn_idx)
        _lib.call("samble_n2p_attn_bwd_f32", qkv.data_ptr(), qkv.stride(0), qkv.stride(1), nn_idx.data_ptr(),
                  g.data_ptr(), B, N, K, C3 // 3, heads, int(bool(diff)), dqkv.data_ptr(), dqkv.stride(0),
                  dqkv.stride(1), _p(order), _p(offsets), ws.data_ptr(), nbytes, _stream())
    return dqkv


HEADS_ASM = {"dot": 1.0, "l2": 2.0, "l2+": -2.0}   # qk_mul of samble_attn_heads_*_f32 (include/samble.h)


def _heads_operands(qkv: torch.Tensor, heads: int):
    if qkv.dim() != 3 or qkv.shape[2] % (3 * heads) or qkv.dtype != torch.float32 or not qkv.is_contiguous():
        raise ValueError("qkv must be contiguous fp32 (B, N, 3 * H * D) rows [Q|K|V]")
    C = qkv.shape[2] // 3
    return qkv[:, :, 0:C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], C // heads


def stage_attn_heads_fwd(qkv: torch.Tensor, heads: int, asm: str = "dot", key_bias: Optional[torch.Tensor] = None):
    """qkv (B,N,3C) rows [Q|K|V], C = heads * D -> (O (B,N,C), lse (B,heads,N)): per head
    softmax((qk_mul q k^T + key_bias_j) / sqrt(D)) v over all N points (reference models/attention.py:317-355).
    key_bias (B,heads,N): -|k_j|^2 for asm "l2", +|k_j|^2 for "l2+" (the caller forms it), None for "dot"."""
    _need_gpu(qkv, key_bias)
    q, k, v, D = _heads_operands(qkv, heads)
    B, N, _ = qkv.shape
    if key_bias is not None:
        key_bias = _f32c(key_bias)
        if key_bias.shape != (B, heads, N):
            raise ValueError(f"key_bias must be (B, heads, N) = {(B, heads, N)}")
    with torch.cuda.device(qkv.device):
        out = torch.empty((B, N, heads * D), dtype=torch.float32, device=qkv.device)
        lse = torch.empty((B, heads, N), dtype=torch.float32, device=qkv.device)
        _lib.call("samble_attn_heads_fwd_f32", q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                  v.data_ptr(), v.stride(0), v.stride(1), _p(key_bias), HEADS_ASM[asm], B, N, heads, D, out.data_ptr(),
                  out.stride(0), out.stride(1), lse.data_ptr(), _stream())
    return out, lse


def stage_attn_heads_bwd(qkv: torch.Tensor, heads: int, out: torch.Tensor, lse: torch.Tensor, g: torch.Tensor,
                         asm: str = "dot", key_bias: Optional[torch.Tensor] = None):
    """g (B,N,C) = gradient of stage_attn_heads_fwd's O -> (dqkv (B,N,3C), bias_grad (B,heads,N) | None).
    dK in dqkv is the part through q k^T only; bias_grad = d loss / d key_bias."""
    _need_gpu(qkv, out, lse, g, key_bias)
    q, k, v, D = _heads_operands(qkv, heads)
    B, N, _ = qkv.shape
    g, out, lse = _f32c(g), _f32c(out), _f32c(lse)
    if key_bias is not None:
        key_bias = _f32c(key_bias)
    with torch.cuda.device(qkv.device):
        dqkv = torch.empty_like(qkv)
        C = heads * D
        dq, dk, dv = dqkv[:, :, 0:C], dqkv[:, :, C:2 * C], dqkv[:, :, 2 * C:]
        bias_grad = torch.empty((B, heads, N), dtype=torch.float32, device=qkv.device) if key_bias is not None else None
        nbytes = _lib.query("samble_attn_heads_bwd_workspace_bytes", B, N, heads)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=qkv.device)
        _lib.call("samble_attn_heads_bwd_f32", q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                  v.data_ptr(), v.stride(0), v.stride(1), _p(key_bias), HEADS_ASM[asm], B, N, heads, D, out.data_ptr(),
                  out.stride(0), out.stride(1), lse.data_ptr(), g.data_ptr(), g.stride(0), g.stride(1), dq.data_ptr(),
                  dq.stride(0), dq.stride(1), dk.data_ptr(), dk.stride(0), dk.stride(1), dv.data_ptr(), dv.stride(0),
                  dv.stride(1), _p(bias_grad), ws.data_ptr(), nbytes, _stream())
    return dqkv, bias_grad


def stage_attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, n_points: int, n_tokens: int,
                   want_row_std: bool = False):
    """q (B,N,D), k/v (B,N+nt,D) (any row/batch stride, unit channel stride) ->
    O (B,N,D), lse (B,N), token logits (B,N,nt) [, row_std (B,N) when want_row_std]."""
    _need_gpu(q, k, v)
    B, N, D = q.shape
    assert N == n_points and k.shape[1] == n_points + n_tokens and v.shape[1] == k.shape[1]
    for t in (q, k, v):
        if t.stride(2) != 1 or t.dtype != torch.float32:
            raise ValueError("attention operands must be fp32 with unit channel stride")
    with torch.cuda.device(q.device):
        O = torch.empty((B, N, D), dtype=torch.float32, device=q.device)
        lse = torch.empty((B, N), dtype=torch.float32, device=q.device)
        tok = torch.empty((B, N, max(n_tokens, 1)), dtype=torch.float32, device=q.device)
        rstd = torch.empty((B, N), dtype=torch.float32, device=q.device) if want_row_std else None
        _lib.call("samble_attn_fwd_f32", q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0),
                  k.stride(1), v.data_ptr(), v.stride(0), v.stride(1), B, N, n_tokens, D, O.data_ptr(),
                  lse.data_ptr(), tok.data_ptr(), _p(rstd), _stream())
    if want_row_std:
        return O, lse, tok[:, :, :n_tokens], rstd
    return O, lse, tok[:, :, :n_tokens]


def stage_attn_colsum(q: torch.Tensor, k: torch.Tensor, lse: torch.Tensor) -> torch.Tensor:
    """Column sums (B,N) of the point-to-point attention block (idx_mode col_sum)."""
    _need_gpu(q, k, lse)
    B, N, D = q.shape
    with torch.cuda.device(q.device):
        out = torch.empty((B, N), dtype=torch.float32, device=q.device)
        _lib.call("samble_attn_colsum_f32", q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0),
                  k.stride(1), lse.data_ptr(), B, N, D, out.data_ptr(), _stream())
    return out


def attn_map_row_stride(n_points: int, n_tokens: int) -> int:
    """Row stride (floats) of a logit map over n_points + n_tokens keys (samble_attn_map_row_stride)."""
    return int(_lib.query("samble_attn_map_row_stride", n_points, n_tokens))


def stage_stat_score(stat: torch.Tensor):
    """Dense-mode statistic (B,N) -> (score with NaN -> 0, z-score)."""
    _need_gpu(stat)
    stat = _f32c(stat)
    B, N = stat.shape
    with torch.cuda.device(stat.device):
        score = torch.empty_like(stat)
        z = torch.empty_like(stat)
        _lib.call("samble_stat_score_f32", stat.data_ptr(), B, N, score.data_ptr(), z.data_ptr(), _stream())
    return score, z


def stage_sparse_score(q, k, lse, nn_idx, idx_mode: str):
    """-> score (B,N), z (B,N), in-degree (B,N) int32 for the sparse_* idx modes."""
    if idx_mode not in SCORE_MODES:
        raise ValueError("Please check the setting of idx mode!")
    _need_gpu(q, k, lse, nn_idx)
    B, N, D = q.shape
    with torch.cuda.device(q.device):
        score = torch.empty((B, N), dtype=torch.float32, device=q.device)
        z = torch.empty_like(score)
        indeg = torch.empty((B, N), dtype=torch.int32, device=q.device)
        nbytes = _lib.query("samble_score_workspace_bytes", B, N)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
        _lib.call("samble_sparse_score_f32", q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0),
                  k.stride(1), lse.data_ptr(), nn_idx.data_ptr(), B, N, nn_idx.shape[2], D, SCORE_MODES[idx_mode],
                  score.data_ptr(), z.data_ptr(), indeg.data_ptr(), ws.data_ptr(), nbytes, _stream())
    return score, z, indeg


def stage_zscore(score: torch.Tensor) -> torch.Tensor:
    _need_gpu(score)
    score = _f32c(score)
    B, N = score.shape
    with torch.cuda.device(score.device):
        z = torch.empty_like(score)
        _lib.call("samble_zscore_f32", score.data_ptr(), B, N, z.data_ptr(), _stream())
    return z


def stage_batch_quantiles(z: torch.Tensor, num_bins: int) -> torch.Tensor:
    _need_gpu(z)
    z = _f32c(z)
    with torch.cuda.device(z.device):
        out = torch.empty((num_bins - 1,), dtype=torch.float32, device=z.device)
        nbytes = _lib.query("samble_quantiles_workspace_bytes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=z.device)
        _lib.call("samble_batch_quantiles_f32", z.data_ptr(), z.numel(), num_bins, out.data_ptr(), ws.data_ptr(), nbytes,
                  _stream())
    return out


def stage_bin_assign(z, tok_logits, upper, lower, relu_first: bool):
    """-> member bits (B,N) uint8, cap (B,nb) int32, w_pre (B,nb), w (B,nb)."""
    _need_gpu(z, tok_logits, upper, lower)
    z, tok_logits = _f32c(z), _f32c(tok_logits)
    upper = _f32c(upper.reshape(-1))
    lower = _f32c(lower.reshape(-1))
    B, N = z.shape
    nb = upper.numel()
    with torch.cuda.device(z.device):
        member = torch.empty((B, N), dtype=torch.uint8, device=z.device)
        cap = torch.empty((B, nb), dtype=torch.int32, device=z.device)
        w_pre = torch.empty((B, nb), dtype=torch.float32, device=z.device)
        w = torch.empty_like(w_pre)
        _lib.call("samble_bin_assign_f32", z.data_ptr(), tok_logits.data_ptr(), tok_logits.shape[-1], upper.data_ptr(),
                  lower.data_ptr(), B, N, nb, int(bool(relu_first)), member.data_ptr(), cap.data_ptr(),
                  w_pre.data_ptr(), w.data_ptr(), _stream())
    return member, cap, w_pre, w


def stage_alloc_counts(w: torch.Tensor, cap: torch.Tensor, total: int) -> torch.Tensor:
    _need_gpu(w, cap)
    w = _f32c(w)
    cap = cap.to(torch.int32).contiguous()
    B, nb = w.shape
    with torch.cuda.device(w.device):
        counts = torch.empty((B, nb), dtype=torch.int32, device=w.device)
        _lib.call("samble_alloc_counts_f32", w.data_ptr(), cap.data_ptr(), B, nb, int(total), counts.data_ptr(),
                  _stream())
    return counts


def boltzmann_temperature(boltzmann_t, n_points: int, num_bins: int) -> Tuple[int, float]:
    """(temp_mode, temp) for samble_bin_select_f32 from the reference's boltzmann_T setting
    (utils/ops.py:524-550)."""
    if boltzmann_t == "mode_1":
        return 1, 100.0
    if boltzmann_t == "mode_3":
        return 1, 200.0
    if boltzmann_t == "mode_2":
        return 0, n_points / (100.0 * num_bins)
    if boltzmann_t == "mode_4":
        return 0, n_points / (200.0 * num_bins)
    if isinstance(boltzmann_t, numbers.Number):
        return 0, 1 / boltzmann_t
    raise NotImplementedError


def philox_state(device, advance: bool = True):
    """(seed, offset) of the device's default torch generator, the offset advanced by one Philox block (4 outputs):
    what the kernels that draw their own random numbers are keyed with (torch.manual_seed reproduces them)."""
    gen = torch.cuda.default_generators[torch.device(device).index if torch.device(device).index is not None
                                        else torch.cuda.current_device()]
    seed, offset = gen.initial_seed(), gen.get_offset()
    if advance:
        gen.set_offset(offset + 4)
    return seed & 0xFFFFFFFFFFFFFFFF, offset


def stage_exp1_noise(seed: int, offset: int, rows: int, n_points: int, device) -> torch.Tensor:
    """The (rows, n_points) Exp(1) tensor stage_bin_select draws from under philox=(seed, offset), written out."""
    with torch.cuda.device(device):
        out = torch.empty((rows, n_points), dtype=torch.float32, device=device)
        _lib.call("samble_exp1_noise_f32", seed, offset, rows, n_points, out.data_ptr(), _stream())
    return out


def stage_bin_select(score, z, member, counts, M: int, sample_mode: str, boltzmann_t, noise=None, philox=None):
    """-> idx (B,M) int64: bins ascending, inside a bin by descending key.
    uniform / random: `noise` (B*num_bins, N) is the Exp(1) draw torch.multinomial makes inside; None: the kernel draws
    it itself under philox = (seed, offset) (default: the state of torch's device generator, advanced)."""
    if sample_mode not in SAMPLE_MODES:
        raise ValueError("Please check the setting of bin sample mode. It must be topk, multinomial or random!")
    _need_gpu(score, z, member, counts, noise)
    B, N = score.shape
    nb = counts.shape[1]
    temp_mode, temp = (0, 1.0)
    if sample_mode == "random":
        temp_mode, temp = boltzmann_temperature(boltzmann_t, N, nb)
    if sample_mode in ("uniform", "random"):
        if noise is None and philox is None and torch.cuda.is_current_stream_capturing():
            # under stream capture a host-side bump of the generator offset is not replayed (every replay would draw
            # the same numbers): draw the Exp(1) tensor with torch's graph-safe generator path instead
            noise = torch.empty((B * nb, N), dtype=torch.float32, device=score.device).exponential_()
        if noise is None:
            seed, offset = philox if philox is not None else philox_state(score.device)
            with torch.cuda.device(score.device):
                idx = torch.empty((B, M), dtype=torch.int64, device=score.device)
                _lib.call("samble_bin_select_seeded_f32", score.data_ptr(), z.data_ptr(), member.data_ptr(),
                          counts.data_ptr(), seed, offset, B, N, nb, M, SAMPLE_MODES[sample_mode], temp_mode, float(temp),
                          idx.data_ptr(), _stream())
            return idx
        noise = _f32c(noise)
        if noise.shape != (B * nb, N):
            raise ValueError(f"noise must be (B*num_bins, N) = {(B * nb, N)}, got {tuple(noise.shape)}")
    with torch.cuda.device(score.device):
        idx = torch.empty((B, M), dtype=torch.int64, device=score.device)  # every slot is written (counts sum to M)
        _lib.call("samble_bin_select_f32", score.data_ptr(), z.data_ptr(), member.data_ptr(), counts.data_ptr(),
                  _p(noise), B, N, nb, M, SAMPLE_MODES[sample_mode], temp_mode, float(temp), idx.data_ptr(),
                  _stream())
    return idx


def stage_topk_indices(score: torch.Tensor, k: int, largest: bool = True) -> torch.Tensor:
    """(B,N) score -> (B,k) int64 indices of the k largest (or smallest) entries, best first;
    exact ties break by ascending index (torch.topk leaves them unspecified)."""
    _need_gpu(score)
    score = _f32c(score)
    B, N = score.shape
    member = torch.ones((B, N), dtype=torch.uint8, device=score.device)
    counts = torch.full((B, 1), k, dtype=torch.int32, device=score.device)
    return stage_bin_select(score, score, member, counts, k, "top_raw" if largest else "bottom_raw", None)


def stage_gather_rows(O: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """O (B,N,D), idx (B,M) int64 -> (B,D,M)."""
    _need_gpu(O, idx)
    B, N, D = O.shape
    M = idx.shape[1]
    with torch.cuda.device(O.device):
        out = torch.empty((B, D, M), dtype=torch.float32, device=O.device)
        _lib.call("samble_gather_rows_f32", O.data_ptr(), O.stride(0), O.stride(1), idx.data_ptr(), B, M, D,
                  out.data_ptr(), _stream())
    return out


def stage_attn_bwd(q, k, v, O, lse, idx, g, n_points: int, n_tokens: int, dq, dk, dv, variant: int = 0) -> None:
    """Fills dq (rows idx, other rows zeroed), dk, dv (views with their own strides).
    variant 1: two kernels (7 matrix products per tile) instead of the fused one (5)."""
    _need_gpu(q, k, v, O, lse, idx, g)
    B, N, D = q.shape
    M = idx.shape[1]
    g = _f32c(g)
    with torch.cuda.device(q.device):
        nbytes = _lib.query("samble_attn_bwd_workspace_bytes", B, n_points, M, D)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
        _lib.call("samble_attn_bwd_f32", q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0),
                  k.stride(1), v.data_ptr(), v.stride(0), v.stride(1), O.data_ptr(), lse.data_ptr(), idx.data_ptr(),
                  g.data_ptr(), B, n_points, n_tokens, M, D, dq.data_ptr(), dq.stride(0), dq.stride(1),
                  dk.data_ptr(), dk.stride(0), dk.stride(1), dv.data_ptr(), dv.stride(0), dv.stride(1), int(variant),
                  ws.data_ptr(), nbytes, _stream())


# Matrix instruction of the attention passes: "tri" = bf16 MFMAs on split fp32 operands (three bf16
# planes, six partial products, fp32 accumulation: fp32-equivalent, csrc/tri_dev.h), "f32" = the
# fp32 MFMA kernels.  Both produce the same map layout; the tests run both against the oracle.
MATRIX_MODE = os.environ.get("SAMBLE_MATRIX_MODE", "tri")


def stage_k_logit_form(k_image: torch.Tensor, k_rows: torch.Tensor) -> torch.Tensor:
    """A row image from stage_tri_split -> its logit form, IN PLACE (include/samble.h samble_tri_k_logit_form):
    what the `k_image` arguments of stage_attn_stats / stage_attn_stats_nl / stage_attn_rows_recompute and the V row
    image of stage_attn_rows_bwd (`images[1]`) expect.  stage_tri_split_qkv and stage_proj_fwd(images=...) hand their K
    and V row images over in this form already."""
    B, rows = k_rows.shape[0], k_rows.shape[1]
    with torch.cuda.device(k_image.device):
        _lib.call("samble_tri_k_logit_form", k_image.data_ptr(), B, rows, _stream())
    return k_image


def stage_tri_split(rows: torch.Tensor, want_rm: bool = True, want_tr: bool = False):
    """fp32 rows (B,R,128) (any row / batch stride) -> operand images (uint8 tensors) for the tri
    kernels: rm (contraction over channels), tr (contraction over the rows of a 32-row tile)."""
    _need_gpu(rows)
    B, R, D = rows.shape
    if rows.stride(2) != 1 or rows.dtype != torch.float32:
        raise ValueError("operands must be fp32 with unit channel stride")
    with torch.cuda.device(rows.device):
        rm = tr = None
        if want_rm:
            rm = torch.empty(_lib.query("samble_tri_image_bytes", B, R, 0), dtype=torch.uint8, device=rows.device)
        if want_tr:
            tr = torch.empty(_lib.query("samble_tri_image_bytes", B, R, 1), dtype=torch.uint8, device=rows.device)
        _lib.call("samble_tri_split_f32", rows.data_ptr(), rows.stride(0), rows.stride(1), B, R, D, _p(rm), _p(tr),
                  _stream())
    return rm, tr


def stage_tri_split_qkv(qkv: torch.Tensor, n_points: int, for_backward: bool = False):
    """qkv (B,N+nt,3*128) point-major rows [Q|K|V] -> (q_image, k_image, v_tr_image[, k_tr_image, v_rm_image])
    in one launch; the last two are what the backward kernels read."""
    _need_gpu(qkv)
    B, NK, D3 = qkv.shape
    D = D3 // 3
    if qkv.stride(2) != 1 or qkv.dtype != torch.float32:
        raise ValueError("operands must be fp32 with unit channel stride")
    with torch.cuda.device(qkv.device):
        q_img = torch.empty(_lib.query("samble_tri_image_bytes", B, n_points, 0), dtype=torch.uint8, device=qkv.device)
        k_img = torch.empty(_lib.query("samble_tri_image_bytes", B, NK, 0), dtype=torch.uint8, device=qkv.device)
        v_img = torch.empty(_lib.query("samble_tri_image_bytes", B, NK, 1), dtype=torch.uint8, device=qkv.device)
        k_tr = v_rm = None
        if for_backward:
            k_tr = torch.empty(_lib.query("samble_tri_image_bytes", B, NK, 1), dtype=torch.uint8, device=qkv.device)
            v_rm = torch.empty(_lib.query("samble_tri_image_bytes", B, NK, 0), dtype=torch.uint8, device=qkv.device)
        _lib.call("samble_tri_split_qkv_f32", qkv.data_ptr(), qkv.stride(0), qkv.stride(1), B, n_points, NK - n_points, D,
                  q_img.data_ptr(), k_img.data_ptr(), v_img.data_ptr(), _p(k_tr), _p(v_rm), _stream())
    return (q_img, k_img, v_img, k_tr, v_rm) if for_backward else (q_img, k_img, v_img)


def stage_attn_stats(q: torch.Tensor, k: torch.Tensor, n_points: int, n_tokens: int, asm: str = "dot",
                     images=None):
    """Pass 1 of the two-pass forward: q (B,N,D), k (B,N+nt,D) -> logit map (B,N,ld) kept in HBM,
    lse (B,N), token logits (B,N,nt).  Columns >= N+nt of the map are -inf.
    asm "dot": S = <q,k>/sqrt(D); "l2": S = -|q-k|^2/sqrt(D) (reference downsample.py:154-175); "l2+": S = +|q-k|^2/sqrt(D)
    (downsample.py:1349, n_tokens = 0).
    images: optional (q_image, k_image) already split (MATRIX_MODE "tri")."""
    _need_gpu(q, k)
    B, N, D = q.shape
    assert N == n_points and k.shape[1] == n_points + n_tokens
    for t in (q, k):
        if t.stride(2) != 1 or t.dtype != torch.float32:
            raise ValueError("attention operands must be fp32 with unit channel stride")
    with torch.cuda.device(q.device):
        ld = _lib.query("samble_attn_map_row_stride", N, n_tokens)
        smap = torch.empty((B, N, ld), dtype=torch.float32, device=q.device)
        lse = torch.empty((B, N), dtype=torch.float32, device=q.device)
        tok = torch.empty((B, N, max(n_tokens, 1)), dtype=torch.float32, device=q.device)
        qn = kn = None
        if asm in ("l2", "l2+"):
            # the kernels form S = (2 <a, k> - qn_i - kn_j) / sqrt(D) from the three inputs as given: a = q with the
            # squared norms gives -|q - k|^2 (l2); a = -q with NEGATED norms gives +|q - k|^2 (l2+, reference
            # models/downsample.py:1349: attends to the farthest keys)
            sgn = 1.0 if asm == "l2" else -1.0
            qn = (sgn * (q * q).sum(-1)).contiguous()
            kn = torch.zeros((B, ld), dtype=torch.float32, device=q.device)
            kn[:, :n_points + n_tokens] = sgn * (k * k).sum(-1)
            if asm == "l2+":
                q, images = (-q).contiguous(), None
        elif asm != "dot":
            raise NotImplementedError
        # l2 scoring keeps the cloud's scaled key norms in LDS beside the tile ring: very long clouds use the fp32 kernel
        if MATRIX_MODE == "tri" and not (asm in ("l2", "l2+") and ld * 4 > 24 * 1024):
            q_img, k_img = images if images is not None else (stage_tri_split(q)[0], stage_k_logit_form(stage_tri_split(k)[0], k))
            _lib.call("samble_attn_stats_tri_f32", q_img.data_ptr(), k_img.data_ptr(), B, N, n_tokens, D,
                      smap.data_ptr(), ld, lse.data_ptr(), tok.data_ptr(), _p(qn), _p(kn), _stream())
        else:
            _lib.call("samble_attn_stats_f32", q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0),
                      k.stride(1), B, N, n_tokens, D, smap.data_ptr(), ld, lse.data_ptr(), tok.data_ptr(), _p(qn),
                      _p(kn), _stream())
    return smap, lse, tok[:, :, :n_tokens]


def stage_attn_rows(smap: torch.Tensor, lse: torch.Tensor, v: torch.Tensor, idx: torch.Tensor, n_points: int,
                    n_tokens: int, v_image=None) -> torch.Tensor:
    """Pass 2: the M sampled rows idx (B,M) of softmax(map) times v (B,N+nt,D) -> x_ds (B,D,M).
    v_image: optional transposed operand image of v already split (MATRIX_MODE "tri")."""
    _need_gpu(smap, lse, v, idx)
    B, N, ld = smap.shape
    M, D = idx.shape[1], v.shape[2]
    assert N == n_points and v.shape[1] == n_points + n_tokens and idx.dtype == torch.int64 and idx.is_contiguous()
    with torch.cuda.device(smap.device):
        out = torch.empty((B, D, M), dtype=torch.float32, device=smap.device)
        if MATRIX_MODE == "tri":
            if v_image is None:
                v_image = stage_tri_split(v, want_rm=False, want_tr=True)[1]
            _lib.call("samble_attn_rows_fwd_tri_f32", smap.data_ptr(), ld, lse.data_ptr(), v_image.data_ptr(),
                      idx.data_ptr(), B, N, n_tokens, M, D, out.data_ptr(), _stream())
        else:
            _lib.call("samble_attn_rows_fwd_f32", smap.data_ptr(), ld, lse.data_ptr(), v.data_ptr(), v.stride(0),
                      v.stride(1), idx.data_ptr(), B, N, n_tokens, M, D, out.data_ptr(), _stream())
    return out


ROWS_BWD_FUSED_DKDV, ROWS_BWD_PMAP = 1, 2   # include/samble.h: variants of samble_attn_rows_bwd_tri_f32
BWD_DQ_TOKEN_ROWS = 8                       # SAMBLE_BWD_DQ_TOKEN_ROWS


def score_workspace(B: int, n_points: int, num_bins, device):
    """The (uninitialised) workspace stage_attn_stats_nl(score=...) accumulates into: hand it to stage_nn_prepare(clear=)
    and on to stage_attn_stats_nl(cleared_ws=) so that no memset launch sits between the two."""
    nbytes = _lib.query("samble_select_chain_workspace_bytes", B, n_points) if num_bins else \
        _lib.query("samble_score_workspace_bytes", B, n_points)
    with torch.cuda.device(device):
        return torch.empty(nbytes, dtype=torch.uint8, device=device)


def stage_nn_prepare(nn_idx: torch.Tensor, clear: Optional[torch.Tensor] = None):
    """Neighbour lists (B,N,K) int32 -> (the same lists in ascending index order, the per-(tile, query) membership
    words (B, ceil(N/32), N) int32) for the map-free forward (stage_attn_stats_nl).
    clear: a contiguous byte tensor the kernel zeroes on its way (score_workspace)."""
    _need_gpu(nn_idx, clear)
    nn_idx = nn_idx.to(torch.int32).contiguous()
    B, N, K = nn_idx.shape
    with torch.cuda.device(nn_idx.device):
        nn_sorted = torch.empty_like(nn_idx)
        masks = torch.empty((B, (N + 31) // 32, N), dtype=torch.int32, device=nn_idx.device)
        assert masks.numel() * 4 == _lib.query("samble_nn_masks_bytes", B, N)
        _lib.call("samble_nn_prepare", nn_idx.data_ptr(), B, N, K, nn_sorted.data_ptr(), masks.data_ptr(), _p(clear),
                  clear.numel() * clear.element_size() if clear is not None else 0, _stream())
    return nn_sorted, masks


def stage_attn_stats_nl(q_image, k_image, masks, B: int, n_points: int, n_tokens: int, n_neighbors: int, D: int = 128,
                        want_nl: bool = True, score=None, cleared_ws: Optional[torch.Tensor] = None):
    """Pass 1 without the logit map (MATRIX_MODE "tri", asm "dot"): -> neighbour logits (B,N,K) in the order of
    stage_nn_prepare's sorted lists (None unless want_nl), lse (B,N), token logits (B,N,nt).
    score = (nn_sorted, idx_mode, num_bins or None): the pass also accumulates the sparse_* score statistics into a
    fresh workspace (returned as 4th value; hand it to stage_sparse_score_map / stage_score_quantiles with
    smap=None): no separate score pass, no neighbour-logit array.  cleared_ws: that workspace, already zeroed
    (score_workspace + stage_nn_prepare(clear=)); else a fresh one is allocated and zeroed by a memset launch."""
    _need_gpu(q_image, k_image, masks)
    dev = q_image.device
    with torch.cuda.device(dev):
        nl = torch.empty((B, n_points, n_neighbors), dtype=torch.float32, device=dev) if want_nl else None
        lse = torch.empty((B, n_points), dtype=torch.float32, device=dev)
        tok = torch.empty((B, n_points, max(n_tokens, 1)), dtype=torch.float32, device=dev)
        nn_sorted, mode, ws, nbytes = None, 0, None, 0
        if score is not None:
            nn_sorted, idx_mode, num_bins = score
            if idx_mode not in SCORE_MODES:
                raise ValueError("Please check the setting of idx mode!")
            mode = SCORE_MODES[idx_mode]
            nbytes = _lib.query("samble_select_chain_workspace_bytes", B, n_points) if num_bins else \
                _lib.query("samble_score_workspace_bytes", B, n_points)
            if cleared_ws is not None:
                if cleared_ws.numel() != nbytes or cleared_ws.dtype != torch.uint8 or not cleared_ws.is_contiguous():
                    raise ValueError(f"cleared_ws must be the {nbytes} bytes of score_workspace")
                ws = cleared_ws
            else:
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.call("samble_attn_stats_nl_tri_f32", q_image.data_ptr(), k_image.data_ptr(), B, n_points, n_tokens, D,
                  masks.data_ptr(), n_neighbors, _p(nl), lse.data_ptr(), tok.data_ptr(), _p(nn_sorted), mode, _p(ws),
                  nbytes, 1 if (ws is not None and ws is cleared_ws) else 0, _stream())
    return nl, lse, tok[:, :, :n_tokens], ws


def stage_attn_rows_recompute(q_image, k_image, v_image, lse, idx, n_points: int, n_tokens: int, want_pmap: bool,
                              D: int = 128):
    """Pass 2 without the logit map: the M sampled rows idx (B,M) recompute their logits from the images.
    -> x_ds (B,D,M), P map (B,M,ld) of the sampled rows (what stage_attn_rows_bwd takes with ROWS_BWD_PMAP) or None."""
    _need_gpu(q_image, k_image, v_image, lse, idx)
    B, M = idx.shape
    assert idx.dtype == torch.int64 and idx.is_contiguous()
    dev = lse.device
    with torch.cuda.device(dev):
        out = torch.empty((B, D, M), dtype=torch.float32, device=dev)
        ld = _lib.query("samble_attn_map_row_stride", n_points, n_tokens)
        pmap = torch.empty((B, M, ld), dtype=torch.float32, device=dev) if want_pmap else None
        _lib.call("samble_attn_rows_fwd_recompute_tri_f32", q_image.data_ptr(), k_image.data_ptr(), v_image.data_ptr(),
                  lse.data_ptr(), idx.data_ptr(), B, n_points, n_tokens, M, D, out.data_ptr(), _p(pmap), ld, _stream())
    return out, pmap


def stage_sparse_score_map(smap, lse, nn_idx, idx_mode: str, compact: bool = False, ws=None):
    """stage_sparse_score with A_ij read from the logit map instead of recomputed.
    compact: `smap` is the (B,N,K) neighbour-logit array of stage_attn_stats_nl and nn_idx the sorted lists.
    smap None: the statistics were accumulated into `ws` by stage_attn_stats_nl(score=...) already."""
    if idx_mode not in SCORE_MODES:
        raise ValueError("Please check the setting of idx mode!")
    _need_gpu(lse, nn_idx)
    B, N = lse.shape
    ld = 0
    if smap is not None:
        _need_gpu(smap)
        ld = smap.shape[2]
        if compact:
            assert smap.shape == nn_idx.shape
            ld = 0
    elif ws is None:
        raise ValueError("smap=None needs the workspace stage_attn_stats_nl filled")
    with torch.cuda.device(lse.device):
        score = torch.empty((B, N), dtype=torch.float32, device=lse.device)
        z = torch.empty_like(score)
        indeg = torch.empty((B, N), dtype=torch.int32, device=lse.device)
        nbytes = _lib.query("samble_score_workspace_bytes", B, N)
        if smap is not None:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=lse.device)
        _lib.call("samble_sparse_score_map_f32", _p(smap), ld, lse.data_ptr(), nn_idx.data_ptr(), B, N,
                  nn_idx.shape[2], SCORE_MODES[idx_mode], score.data_ptr(), z.data_ptr(), indeg.data_ptr(),
                  ws.data_ptr(), ws.numel(), _stream())
    return score, z, indeg


def chain_supported(B: int, N: int, num_bins: int) -> bool:
    """True when the fused select chain (csrc/chain.hip: two launches for score/z/quantiles and
    boundaries/bins/counts) takes this shape; otherwise the stand-alone stage kernels run."""
    return bool(_lib.query("samble_select_chain_supported", B, N, num_bins))


def stage_score_quantiles(smap, lse, nn_idx, idx_mode: str, num_bins: int, want_quantiles: bool, compact: bool = False,
                          ws=None, watch=None, counted: bool = False):
    """stage_sparse_score_map + stage_batch_quantiles in two launches (compact / smap None + ws: as in
    stage_sparse_score_map; then it is ONE launch).
    -> score (B,N), z (B,N), in-degree (B,N) int32, quantiles (nb-1,) or None, chain workspace (hand it to
    stage_bin_plan).  counted: the quantiles come as the kernel writes them, (nb,) = the nb-1 quantiles and a validity
    count of 1.0 (all zeros if the chain gave up) -- what `world_sum` exchanges and stage_bin_plan(counted=True) takes."""
    if idx_mode not in SCORE_MODES:
        raise ValueError("Please check the setting of idx mode!")
    _need_gpu(lse, nn_idx)
    B, N = lse.shape
    ld = 0
    if smap is not None:
        _need_gpu(smap)
        ld = smap.shape[2]
        if compact:
            assert smap.shape == nn_idx.shape
            ld = 0
    elif ws is None:
        raise ValueError("smap=None needs the workspace stage_attn_stats_nl filled")
    with torch.cuda.device(lse.device):
        score = torch.empty((B, N), dtype=torch.float32, device=lse.device)
        z = torch.empty_like(score)
        indeg = torch.empty((B, N), dtype=torch.int32, device=lse.device)
        quant = torch.empty((num_bins,), dtype=torch.float32, device=lse.device) if want_quantiles else None
        nbytes = _lib.query("samble_select_chain_workspace_bytes", B, N)
        if smap is not None:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=lse.device)
        assert ws.numel() >= nbytes
        _lib.call("samble_sparse_score_map_quantiles_f32", _p(smap), ld, lse.data_ptr(), nn_idx.data_ptr(), B, N,
                  nn_idx.shape[2], SCORE_MODES[idx_mode], num_bins, score.data_ptr(), z.data_ptr(), indeg.data_ptr(),
                  _p(quant), ws.data_ptr(), ws.numel(), CHAIN_SPIN_BUDGET, _mailbox(watch), _stream())
    if quant is not None and not counted:
        quant = quant[:num_bins - 1]
    return score, z, indeg, quant, ws


# poll rounds a grid barrier of the fused select chain waits before it gives up (include/samble.h `spin_budget`):
# 0 = the library's default (2^20, about a second); 0xFFFFFFFF injects the give-up (tests)
CHAIN_SPIN_BUDGET = 0


class ChainWatch:
    """The fused select chain's status, watched without a synchronisation and without a copy: one int32 in pinned host
    memory that a grid barrier which gives up sets to 1 with a system-scope store (include/samble.h `host_status`); the
    host looks at it before the NEXT chain launch (`poll`), or any time through `timed_out()`.  1 = SAMBLE_E_TIMEOUT: the
    barrier's workgroups were not all resident and the step that ran it produced placeholder selections -- the watch then
    stays `observed` and the caller uses the stand-alone stage kernels (stage_sparse_score_map, stage_batch_quantiles,
    ...).  `observed` (the word was seen; switches the chain off) and `reported` (the layer has raised for it; owned by
    DownSampleToken.forward) are separate, so that whichever call site sees the word first, the error is raised once."""

    def __init__(self):
        self.flag = None
        self.observed = False
        self.reported = False

    # the mailbox is pinned host memory the kernels store to: a copy of the module (copy.deepcopy for an EMA / SWA
    # model, pickling) must not carry a pageable clone of it -- the copy starts without one and pins its own lazily
    def __getstate__(self):
        return {"observed": self.observed, "reported": self.reported}

    def __setstate__(self, state):
        self.flag = None
        self.observed = bool(state.get("observed", False))
        self.reported = bool(state.get("reported", False))

    def __deepcopy__(self, memo):
        twin = ChainWatch()
        twin.observed, twin.reported = self.observed, self.reported
        return twin

    @property
    def tripped(self) -> bool:
        return self.observed

    @tripped.setter
    def tripped(self, value: bool) -> None:
        self.observed = bool(value)

    def host_ptr(self) -> int:
        if self.flag is None or not self.flag.is_pinned():
            self.flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        return self.flag.data_ptr()

    def poll(self, sync: bool = False) -> bool:
        """True exactly once: at the call that finds the word raised."""
        if sync:
            torch.cuda.synchronize()
        if self.flag is not None and int(self.flag[0]) != 0:
            self.flag[0] = 0
            fresh = not self.observed
            self.observed = True
            return fresh
        return False

    def timed_out(self, sync: bool = False) -> bool:
        self.poll(sync)
        return self.observed


def _mailbox(watch: Optional["ChainWatch"]):
    return None if watch is None else watch.host_ptr()


def _fresh_boundary_state(nb: int, dev) -> List[torch.Tensor]:
    """The two (1,1,1,nb) tensors of a first dynamic call, for a kernel to fill.  NaN, not uninitialised: a fused chain
    that gives up (csrc/chain.hip chain_bail) leaves the state unwritten, and every blending kernel takes a NaN state
    for "no state yet" (include/samble.h) -- so the next call initialises it from its quantiles, like the reference's
    first call (utils/ops.py:214-233), instead of blending garbage for ever.  One (2, nb) fill, first call only."""
    both = torch.full((2, 1, 1, 1, nb), float("nan"), dtype=torch.float32, device=dev)
    return [both[0], both[1]]


def stage_select_chain(lse, tok_logits, nn_idx, idx_mode: str, num_bins: int, want_quantiles: bool, boundaries,
                       momentum_update_factor: float, relu_first: bool, M: int, smap=None, compact: bool = False, ws=None,
                       watch=None):
    """stage_score_quantiles + stage_bin_plan as ONE launch (a single rank: no all-reduce of the quantiles stands between
    them; csrc/chain.hip select_chain_kernel).  smap / compact / ws as in stage_score_quantiles.
    -> score, z, in-degree, quantiles | None, boundaries [upper, lower], member, cap, w_pre, w, counts, workspace."""
    if idx_mode not in SCORE_MODES:
        raise ValueError("Please check the setting of idx mode!")
    _need_gpu(lse, nn_idx, tok_logits)
    B, N = lse.shape
    nb = num_bins
    dev = lse.device
    tok_logits = _f32c(tok_logits)
    ld = 0
    if smap is not None:
        _need_gpu(smap)
        ld = 0 if compact else smap.shape[2]
    elif ws is None:
        raise ValueError("smap=None needs the workspace stage_attn_stats_nl filled")
    first = boundaries is None
    if first:
        if not want_quantiles:
            raise ValueError("static boundaries must be given")
        boundaries = _fresh_boundary_state(nb, dev)
    else:
        boundaries = [boundaries[0].detach(), boundaries[1].detach()]
        if not all(t.is_contiguous() and t.dtype == torch.float32 and t.device == dev for t in boundaries):
            boundaries = [t.to(device=dev, dtype=torch.float32).contiguous() for t in boundaries]
    with torch.cuda.device(dev):
        score = torch.empty((B, N), dtype=torch.float32, device=dev)
        z = torch.empty_like(score)
        indeg = torch.empty((B, N), dtype=torch.int32, device=dev)
        quant = torch.empty((nb - 1,), dtype=torch.float32, device=dev) if want_quantiles else None
        member = torch.empty((B, N), dtype=torch.uint8, device=dev)
        cap = torch.empty((B, nb), dtype=torch.int32, device=dev)
        w_pre = torch.empty((B, nb), dtype=torch.float32, device=dev)
        w = torch.empty_like(w_pre)
        counts = torch.empty((B, nb), dtype=torch.int32, device=dev)
        nbytes = _lib.query("samble_select_chain_workspace_bytes", B, N)
        if smap is not None:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        assert ws.numel() >= nbytes
        _lib.call("samble_select_chain_f32", _p(smap), ld, lse.data_ptr(), nn_idx.data_ptr(), nn_idx.shape[2],
                  SCORE_MODES[idx_mode], tok_logits.data_ptr(), tok_logits.shape[-1], int(bool(want_quantiles)), _p(quant),
                  boundaries[0].data_ptr(), boundaries[1].data_ptr(), int(first), float(momentum_update_factor),
                  float(1 - momentum_update_factor), B, N, nb, int(bool(relu_first)), int(M), score.data_ptr(),
                  z.data_ptr(), indeg.data_ptr(), member.data_ptr(), cap.data_ptr(), w_pre.data_ptr(), w.data_ptr(),
                  counts.data_ptr(), ws.data_ptr(), ws.numel(), CHAIN_SPIN_BUDGET, _mailbox(watch), _stream())
    return score, z, indeg, quant, boundaries, member, cap, w_pre, w, counts, ws


def single_rank() -> bool:
    """No process group, or a group of one: nothing is exchanged between the quantiles and the bin plan."""
    return not (torch.distributed.is_available() and torch.distributed.is_initialized()
                and torch.distributed.get_world_size() > 1)


def stage_bin_plan(z, tok_logits, quantiles, boundaries, num_bins: int, momentum_update_factor: float, relu_first: bool,
                   M: int, ws, watch=None, counted: bool = False):
    """blend_boundaries + stage_bin_assign + stage_alloc_counts in one launch (`ws` from stage_score_quantiles).
    quantiles None: `boundaries` are used as they are (static); else they are initialised (boundaries None) or
    blended IN PLACE (reference utils/ops.py:201-233).  counted: `quantiles` is the (nb,) tensor of
    stage_score_quantiles(counted=True) after `world_sum`: the kernel divides the nb-1 sums by element nb-1, the
    number of ranks that contributed (utils/ops.py:199's `/ world_size` without a launch of its own).
    -> boundaries [upper, lower], member (B,N) uint8, cap (B,nb), w_pre (B,nb), w (B,nb), counts (B,nb) int32."""
    _need_gpu(z, tok_logits)
    z, tok_logits = _f32c(z), _f32c(tok_logits)
    B, N = z.shape
    nb = num_bins
    dev = z.device
    first = boundaries is None
    if first:
        if quantiles is None:
            raise ValueError("static boundaries must be given")
        boundaries = _fresh_boundary_state(nb, dev)
    else:
        boundaries = [boundaries[0].detach(), boundaries[1].detach()]
        if not all(t.is_contiguous() and t.dtype == torch.float32 and t.device == dev for t in boundaries):
            boundaries = [t.to(device=dev, dtype=torch.float32).contiguous() for t in boundaries]
    qd = None
    if quantiles is not None:
        quantiles = _f32c(quantiles)
        if counted:
            assert quantiles.numel() == nb
            qd = quantiles.data_ptr() + 4 * (nb - 1)
    with torch.cuda.device(dev):
        member = torch.empty((B, N), dtype=torch.uint8, device=dev)
        cap = torch.empty((B, nb), dtype=torch.int32, device=dev)
        w_pre = torch.empty((B, nb), dtype=torch.float32, device=dev)
        w = torch.empty_like(w_pre)
        counts = torch.empty((B, nb), dtype=torch.int32, device=dev)
        _lib.call("samble_bin_plan_f32", z.data_ptr(), tok_logits.data_ptr(), tok_logits.shape[-1],
                  _p(quantiles), qd, boundaries[0].data_ptr(), boundaries[1].data_ptr(), int(first), float(momentum_update_factor),
                  float(1 - momentum_update_factor), B, N, nb, int(bool(relu_first)), int(M), member.data_ptr(),
                  cap.data_ptr(), w_pre.data_ptr(), w.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel(),
                  CHAIN_SPIN_BUDGET, _mailbox(watch), _stream())
    return boundaries, member, cap, w_pre, w, counts


def stage_attn_rows_bwd(q, k, v, smap, lse, x_ds, idx, g, n_points: int, n_tokens: int, dq, dk, dv,
                        asm: str = "dot", images=None, variant: int = 0, dq_token_rows: bool = False) -> None:
    """stage_attn_bwd for the two-pass forward: S comes from the map, O from x_ds (B,D,M).
    dq_token_rows: dq is the first n_points rows of a (B, n_points + n_tokens, .) block and the token rows behind them
    are to receive zeros (SAMBLE_BWD_DQ_TOKEN_ROWS).
    asm "l2": the kernels also return the column sums of dS and the gradients are finished here:
    dS/dq_i = scale (2 k_j - 2 q_i), dS/dk_j = scale (2 q_i - 2 k_j), rows of dS sum to zero.
    images: optional (k_tr_image, v_rm_image) of the forward's split (MATRIX_MODE "tri")."""
    _need_gpu(q, k, v, smap, lse, x_ds, idx, g)
    B, N, D = q.shape
    M = idx.shape[1]
    g = _f32c(g)
    x_ds = _f32c(x_ds)
    if dq_token_rows and n_tokens:
        if MATRIX_MODE != "tri":
            raise ValueError("dq_token_rows comes with MATRIX_MODE 'tri' (samble_attn_rows_bwd_tri_f32's variant bits)")
        variant = int(variant) | BWD_DQ_TOKEN_ROWS
    if asm == "l2+":  # the forward's query operand was a = -q (stage_attn_stats)
        q = (-q).contiguous()
    with torch.cuda.device(q.device):
        cs = torch.zeros((B, n_points + n_tokens), dtype=torch.float32, device=q.device) if asm in ("l2", "l2+") else None
        if MATRIX_MODE == "tri":
            if images is None:
                images = (stage_tri_split(k, want_rm=False, want_tr=True)[1], stage_k_logit_form(stage_tri_split(v)[0], v))
            nbytes = _lib.query("samble_attn_rows_bwd_tri_workspace_bytes", B, n_points, M, D)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
            _lib.call("samble_attn_rows_bwd_tri_f32", q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0),
                      k.stride(1), v.data_ptr(), v.stride(0), v.stride(1), images[0].data_ptr(), images[1].data_ptr(),
                      smap.data_ptr(), smap.shape[2], lse.data_ptr(), x_ds.data_ptr(), idx.data_ptr(), g.data_ptr(), B,
                      n_points, n_tokens, M, D, dq.data_ptr(), dq.stride(0), dq.stride(1), dk.data_ptr(), dk.stride(0),
                      dk.stride(1), dv.data_ptr(), dv.stride(0), dv.stride(1), _p(cs), int(variant), ws.data_ptr(),
                      nbytes, _stream())
        else:
            nbytes = _lib.query("samble_attn_bwd_workspace_bytes", B, n_points, M, D)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
            _lib.call("samble_attn_rows_bwd_f32", q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0),
                      k.stride(1), v.data_ptr(), v.stride(0), v.stride(1), smap.data_ptr(), smap.shape[2], lse.data_ptr(),
                      x_ds.data_ptr(), idx.data_ptr(), g.data_ptr(), B, n_points, n_tokens, M, D, dq.data_ptr(),
                      dq.stride(0), dq.stride(1), dk.data_ptr(), dk.stride(0), dk.stride(1), dv.data_ptr(), dv.stride(0),
                      dv.stride(1), _p(cs), ws.data_ptr(), nbytes, _stream())
        if asm == "l2":
            dq.mul_(2.0)
            dk.mul_(2.0).sub_(2.0 * cs.unsqueeze(-1) * k)
        elif asm == "l2+":
            # S = scale (2 <a, k> + |q|^2 + |k|^2), a = -q: dq = -(2 dA), dk_j = 2 dK_j + 2 c_j k_j (rows of dS sum to 0)
            dq.mul_(-2.0)
            dk.mul_(2.0).add_(2.0 * cs.unsqueeze(-1) * k)


# ------------------------------------------------------------------------------------------------
# reference-named functions (utils/ops.py)
# ------------------------------------------------------------------------------------------------
def _channel_major(a: torch.Tensor) -> torch.Tensor:
    """(B,N,C) -> contiguous (B,C,N); free when `a` is the usual permute(0,2,1) view."""
    return _f32c(a.permute(0, 2, 1))


def knn(a: torch.Tensor, b: torch.Tensor, k: int):
    """utils/ops.py:17-44: a (B,N,C), b (B,M,C) -> (distance (B,N,k) NEGATED like the
    reference's topk of -cdist, idx (B,N,k) int64)."""
    idx, dist = stage_knn(_channel_major(a), _channel_major(b), k, want_dist=True)
    return -dist, idx.long()


def index_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """utils/ops.py:5-14: points (B,N,C), idx (B,M,K) -> (B,M,K,C)."""
    shape = idx.shape
    B, N, C = points.shape
    if points.is_cuda and points.requires_grad:
        # same rows, picked as an embedding lookup over the (B*N, C) table: its backward is ATen's sorted segment
        # reduction instead of torch.gather's element-wise atomic scatter (1.09 ms -> 0.2 ms for the seg block's
        # interpolation: 6 144 picks of 128 channels per cloud)
        flat = (idx.reshape(B, -1).long() + torch.arange(B, device=idx.device)[:, None] * N).reshape(-1)
        res = torch.nn.functional.embedding(flat, points.reshape(B * N, C))
        return res.view(*shape, C)
    flat = idx.reshape(shape[0], -1).long()
    res = torch.gather(points, 1, flat[..., None].expand(-1, -1, points.shape[-1]))
    return res.view(*shape, -1)


GROUP_MODES = {"neighbor": 0, "diff": 1, "center_neighbor": 2, "center_diff": 3}


def _group_apply(pcd, nn_idx, mode):
    """fp32 -> the HIP gather; any other dtype (the fp64 verification runs of the tests) -> the same
    expression in torch on the same device."""
    if pcd.dtype == torch.float32:
        return _GroupGather.apply(pcd, nn_idx, mode)
    pts = pcd.permute(0, 2, 1)
    nb = index_points(pts, nn_idx)
    g = (nb - pts[:, :, None, :] if mode in (1, 3) else nb).permute(0, 3, 1, 2)
    if mode >= 2:
        g = torch.cat([pcd[:, :, :, None].expand(-1, -1, -1, nn_idx.shape[2]), g], dim=1)
    return g


class _GroupGather(torch.autograd.Function):
    """x (B,C,N), nn (B,N,K) int32 -> (B, C or 2C, N, K) on the HIP gather kernel; backward = the transpose
    (scatter-add over neighbours, centre terms summed over K)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, nn_idx, mode):
        _need_gpu(x, nn_idx)
        x = _f32c(x)
        nn_idx = nn_idx.to(torch.int32).contiguous()
        B, C, N = x.shape
        K = nn_idx.shape[2]
        with torch.cuda.device(x.device):
            out = torch.empty((B, 2 * C if mode >= 2 else C, N, K), dtype=torch.float32, device=x.device)
            _lib.call("samble_group_gather_f32", x.data_ptr(), nn_idx.data_ptr(), B, C, N, K, mode, out.data_ptr(),
                      _stream())
        ctx.save_for_backward(nn_idx)
        ctx.mode = mode
        ctx.C = C
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        (nn_idx,) = ctx.saved_tensors
        mode, C = ctx.mode, ctx.C
        B, _, N, K = g.shape
        gn = g[:, C:] if mode >= 2 else g                      # gradient of the neighbour half
        dx = torch.zeros((B, C, N), dtype=g.dtype, device=g.device)
        dx.scatter_add_(2, nn_idx.long().view(B, 1, N * K).expand(-1, C, -1), gn.reshape(B, C, N * K))
        if mode in (1, 3):
            dx -= gn.sum(-1)
        if mode >= 2:
            dx += g[:, :C].sum(-1)
        return dx, None, None


def select_neighbors(pcd, K, neighbor_type, normal_channel=False):
    """utils/ops.py:47-65: pcd (B,C,N) -> (neighbours (B,C,N,K), idx (B,N,K))."""
    if neighbor_type not in ("neighbor", "diff"):
        raise ValueError(f'neighbor_type should be "neighbor" or "diff", but got {neighbor_type}')
    src = pcd[:, :3, :] if (normal_channel and pcd.shape[1] == 6) else pcd
    nn_idx = stage_knn(src.detach(), src.detach(), K)
    return _group_apply(pcd, nn_idx, GROUP_MODES[neighbor_type]), nn_idx.long()


def select_neighbors_interpolate(unknown, known, known_feature, K=3):
    """utils/ops.py:68-80: cross-set kNN with positive distances."""
    idx, d = stage_knn(unknown, known, K, want_dist=True)
    idx = idx.long()
    neighbors = index_points(known_feature.permute(0, 2, 1), idx).permute(0, 3, 1, 2)
    return neighbors, idx, d


def group(pcd, K, group_type, normal_channel=False):
    """utils/ops.py:83-112."""
    if group_type not in GROUP_MODES:
        raise ValueError(
            f"group_type should be neighbor, diff, center_neighbor or center_diff, but got {group_type}")
    src = pcd[:, :3, :] if (normal_channel and pcd.shape[1] == 6) else pcd
    nn_idx = stage_knn(src.detach(), src.detach(), K)
    return _group_apply(pcd, nn_idx, GROUP_MODES[group_type]), nn_idx.long()


def neighbor_mask(pcd, K):
    """utils/ops.py:125-133: dense 0/1 (B,N,N) mask.  Kept for API parity only; the sampler
    itself never builds it."""
    idx = stage_knn(pcd, pcd, K).long()
    B, N, _ = idx.shape
    mask = torch.zeros(B, N, N, dtype=torch.float32, device=idx.device)
    mask.scatter_(2, idx, 1.0)
    return mask


def farthest_point_sample(xyz: torch.Tensor, npoint: int, start: Optional[torch.Tensor] = None) -> torch.Tensor:
    """utils/ops.py:622-643: xyz (B,N,3) -> centroid indices (B,npoint) int64 in selection order.
    `start` (B,) int64 is the first centroid per cloud; the reference draws it with torch.randint,
    which is what happens here too when it is omitted."""
    _need_gpu(xyz)
    B, N, C = xyz.shape
    if C != 3:
        raise ValueError("farthest_point_sample expects (B, N, 3) coordinates")
    if start is None:
        start = torch.randint(0, N, (B,), dtype=torch.long, device=xyz.device)
    start = start.to(device=xyz.device, dtype=torch.long).contiguous()
    cm = _channel_major(xyz)  # (B,3,N); free when xyz is the usual permute(0,2,1) view
    with torch.cuda.device(xyz.device):
        out = torch.empty((B, npoint), dtype=torch.long, device=xyz.device)
        _lib.call("samble_fps_f32", cm.data_ptr(), start.data_ptr(), B, N, npoint, out.data_ptr(), _stream())
    return out


def gather_by_idx(pcd, idx):
    """utils/ops.py:136-145: pcd (B,C,N), idx (B,H=1,K) -> (B,C,K)."""
    _need_gpu(pcd, idx)
    pcd = _f32c(pcd)
    B, C, N = pcd.shape
    idx2 = idx.reshape(B, -1).long().contiguous()
    M = idx2.shape[1]
    with torch.cuda.device(pcd.device):
        out = torch.empty((B, C, M), dtype=torch.float32, device=pcd.device)
        _lib.call("samble_gather_points_f32", pcd.data_ptr(), B, C, N, idx2.data_ptr(), M, out.data_ptr(), _stream())
    return out


def index_points_for_fps(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """utils/ops.py:646-667: points (B,N,C), idx (B,S) -> (B,S,C)."""
    return torch.gather(points, 1, idx.long().unsqueeze(-1).expand(-1, -1, points.shape[-1]))


def fps(x: torch.Tensor, xyz: torch.Tensor, npoint: int):
    """utils/ops.py:670-692: x (B,C,N), xyz (B,3,N) -> ((x at the npoint farthest points (B,C,npoint), idx (B,1,npoint)),
    (None, None)) -- the sampler-shaped return of the reference, on the HIP farthest-point kernel + the HIP column gather."""
    fps_idx = farthest_point_sample(xyz.permute(0, 2, 1), npoint)
    return (gather_by_idx(x, fps_idx.unsqueeze(1)), fps_idx.unsqueeze(1)), (None, None)


def l2_global(q: torch.Tensor, k: torch.Tensor) -> torch.Tensor:
    """utils/ops.py:115-122: q (B,H,N,D), k (B,H,D,N) -> |q_i - k_j|^2 as the dense (B,H,N,N) tensor, the reference's own
    three-term expression.  API parity only: the samplers and Point2PointAttention never build this tensor (the norm
    terms enter the logits inside attn_stats / attn_heads)."""
    _need_gpu(q, k)
    inner = -2 * torch.matmul(q, k)
    qq = torch.sum(q ** 2, dim=-1, keepdim=True)
    kk = torch.sum(k.transpose(-2, -1) ** 2, dim=-1, keepdim=True)
    return qq + inner + kk.transpose(-2, -1)


def norm_range(x: torch.Tensor, dim: int = -1, n_min=0, n_max=1, mode: str = "minmax") -> torch.Tensor:
    """utils/ops.py:148-171 (minmax / sigmoid / tanh rescaled to [n_min, n_max]; z-score shifted by n_min)."""
    _need_gpu(x)
    if mode == "minmax":
        lo = torch.min(x, dim=dim, keepdim=True)[0]
        x_norm = (x - lo) / (torch.max(x, dim=dim, keepdim=True)[0] - lo + 1e-8)
    elif mode == "sigmoid":
        x_norm = torch.sigmoid(x)
    elif mode == "tanh":
        x_norm = (torch.tanh(x) + 1.0) / 2
    elif mode == "z-score":
        if x.dtype == torch.float32 and dim in (-1, x.dim() - 1) and x.is_contiguous():
            return stage_zscore(x.reshape(-1, x.shape[-1])).view_as(x) + n_min     # the sampler's own z-score kernel
        return (x - torch.mean(x, dim=dim, keepdim=True)) / torch.std(x, dim=dim, unbiased=False, keepdim=True) + n_min
    else:
        raise ValueError(f"norm_range mode should be minmax, sigmoid or tanh, but got {mode}")
    return x_norm * (n_max - n_min) + n_min


def sort_chunk(attention_point_score: torch.Tensor, num_bins: int, dim: int = -1, descending: bool = False):
    """utils/ops.py:239-259: score (B,H,N) -> (num_bins value chunks, num_bins index chunks) of the sorted scores
    (torch.chunk's sizes: ceil(N / num_bins) each, the last one shorter).  fp32 along the last dimension, N <= 8192: the
    HIP select kernel's full ordering (exact ties by ascending index, where torch.sort leaves them unspecified)."""
    _need_gpu(attention_point_score)
    x = attention_point_score
    if x.dtype == torch.float32 and dim in (-1, x.dim() - 1) and x.shape[-1] <= 8192:
        flat = _f32c(x.reshape(-1, x.shape[-1]))
        order = stage_topk_indices(flat, flat.shape[1], largest=descending)
        idx_sorted = order.view(*x.shape[:-1], x.shape[-1])
        x_sorted = torch.gather(x, -1, idx_sorted)
    else:
        x_sorted, idx_sorted = torch.sort(x, dim=dim, descending=descending)
    return torch.chunk(x_sorted, num_bins, dim=dim), torch.chunk(idx_sorted, num_bins, dim=dim)


# True (tests: SAMBLE_POOL_SINGLE_RANK=1): a SyncBatchNorm pools its statistics through the process group even when the
# group has ONE rank -- the all-reduce is then the identity, and the pooled kernels + the collective library (RCCL on the
# one GPU of a test box) run where otherwise only an 8-GPU node would reach them
POOL_SINGLE_RANK = os.environ.get("SAMBLE_POOL_SINGLE_RANK", "0") == "1"


def sync_group(bn):
    """The process group over which `bn` pools its statistics, or None: nn.SyncBatchNorm in training mode with an
    initialised group of more than one rank (the reference trainer converts every BatchNorm, train_modelnet.py:245-246)."""
    if not isinstance(bn, torch.nn.SyncBatchNorm) or not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return None
    group = bn.process_group if bn.process_group is not None else torch.distributed.group.WORLD
    return group if (torch.distributed.get_world_size(group) > 1 or POOL_SINGLE_RANK) else None


def world_average(t: torch.Tensor) -> torch.Tensor:
    """The exchange step of utils/ops.py:191-199: all_reduce(SUM) / world_size when a process
    group exists (RCCL on the GPU, gloo in the CPU tests), identity otherwise."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.all_reduce(t)
        t = t / torch.distributed.get_world_size()
    return t


def world_sum(t: torch.Tensor) -> torch.Tensor:
    """The exchange step of utils/ops.py:191-197 alone: all_reduce(SUM), in place, when a process group exists.  The
    division by the world size is the consumer's (stage_bin_plan(counted=True) divides by the all-reduced validity
    count inside its kernel)."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.all_reduce(t)
    return t


def reshape_gathered_variable(gathered_variable):
    """utils/ops.py:262-286 (test_modelnet.py:289-297): a per-layer list of per-cloud entries -> a per-cloud list of
    per-layer entries; the entries themselves (tensors or per-bin lists) pass through untouched."""
    clouds = len(gathered_variable[0])
    return [[layer[b] for layer in gathered_variable] for b in range(clouds)]


def gather_variable_from_gpus(downsample_module, variable_name, rank, world_size, device):
    """utils/ops.py:289-384: collect one published variable of a sampler layer (`output_variables(name)`) from every rank
    of the evaluation job on rank 0 (the other ranks return None).  Three shapes, as the reference distinguishes them:
      a tensor (B, ...)                      -> the ranks' tensors concatenated along the clouds;
      a list over bins of (B, H, n) tensors  -> (world B, num_bins, H, n);
      a list over bins of per-cloud (H, n_ij) tensors of ragged length (`idx_chunks`)
                                             -> world B lists over the bins of (1, n_ij) tensors, ranks in order.
    The ragged case travels as one flat tensor per rank plus its (B, num_bins) table of lengths (all_gather of the table,
    gather of the values, padded to the longest rank's, to rank 0): RCCL and gloo both carry it."""
    dist = torch.distributed
    value = downsample_module.output_variables(variable_name)

    def all_ranks(t):
        parts = [torch.empty_like(t).to(device) for _ in range(world_size)]
        dist.all_gather(parts, t)
        return torch.cat(parts, dim=0) if rank == 0 else None

    if isinstance(value, torch.Tensor):
        return all_ranks(value)
    if isinstance(value[0], torch.Tensor):
        return all_ranks(torch.stack(list(value), dim=0).permute(1, 0, 2, 3).contiguous())
    num_bins, clouds = len(value), len(value[0])
    first = value[0][0]
    lengths = torch.tensor([[value[t][b].numel() for t in range(num_bins)] for b in range(clouds)], dtype=torch.float32,
                           device=first.device)                      # (float: the reference's table is one too)
    flat = torch.cat([value[t][b].reshape(-1) for b in range(clouds) for t in range(num_bins)])
    tables = [torch.empty_like(lengths).to(device) for _ in range(world_size)]
    dist.all_gather(tables, lengths)
    # every rank's values padded to the longest rank's length: gloo's gather wants equal sizes (RCCL's does not care)
    totals = [int(tb.sum().item()) for tb in tables]
    padded = torch.zeros((max(max(totals), 1),), dtype=flat.dtype, device=flat.device)
    padded[: flat.numel()] = flat
    sinks = [torch.empty_like(padded).to(device) for _ in range(world_size)] if rank == 0 else None
    dist.gather(padded, gather_list=sinks, dst=0)
    if rank != 0:
        return None
    out = []
    for tb, values, total in zip(tables, sinks, totals):
        pieces = torch.split(values[:total], [int(v) for v in tb.reshape(-1).tolist()])
        for b in range(clouds):
            out.append([pieces[b * num_bins + t].reshape(1, -1) for t in range(num_bins)])
    return out


def blend_boundaries(old: Optional[List[torch.Tensor]], quantiles: torch.Tensor, num_bins: int,
                     momentum_update_factor: float) -> List[torch.Tensor]:
    """utils/ops.py:201-233 on the (already rank-averaged) nb-1 quantiles: first call stores them,
    later calls blend `old * mu + (1 - mu) * new` IN PLACE into both (1,1,1,nb) tensors."""
    if quantiles.is_cuda:  # one HIP launch instead of five tiny tensor ops
        first = old is None
        if first:
            new = _fresh_boundary_state(num_bins, quantiles.device)
        else:
            new = [old[0].detach(), old[1].detach()]
            if not (new[0].is_contiguous() and new[1].is_contiguous() and new[0].dtype == torch.float32):
                new = [t.float().contiguous() for t in new]
        q = _f32c(quantiles)
        with torch.cuda.device(q.device):
            _lib.call("samble_blend_boundaries_f32", q.data_ptr(), new[0].data_ptr(), new[1].data_ptr(), num_bins,
                      float(momentum_update_factor), float(1 - momentum_update_factor), int(first), _stream())
        return new
    if old is not None:
        new = [old[0].detach(), old[1].detach()]
        mixed = new[0][0, 0, 0, 1:] * momentum_update_factor + (1 - momentum_update_factor) * quantiles
        new[0][0, 0, 0, 1:] = mixed
        new[1][0, 0, 0, :-1] = mixed
        return new
    upper = torch.empty((num_bins,), device=quantiles.device)
    upper[0] = float("inf")
    upper[1:] = quantiles
    lower = torch.empty((num_bins,), device=quantiles.device)
    lower[-1] = float("-inf")
    lower[:-1] = quantiles
    return [upper.reshape(1, 1, 1, num_bins), lower.reshape(1, 1, 1, num_bins)]


def update_sampling_score_bin_boundary(old_bin_boundaries, attention_point_score, num_bins, momentum_update_factor):
    """utils/ops.py:174-236 (the argument is the z-scored score, any shape)."""
    q = stage_batch_quantiles(attention_point_score.reshape(-1), num_bins)
    q = world_average(q)
    return blend_boundaries(old_bin_boundaries, q, num_bins, momentum_update_factor)


def _member_to_mask(member: torch.Tensor, num_bins: int) -> torch.Tensor:
    bits = torch.arange(num_bins, device=member.device, dtype=torch.uint8)
    return ((member.unsqueeze(-1) >> bits) & 1).bool().unsqueeze(1)  # (B,1,N,nb)


def bin_partition(attention_point_score, bin_boundaries, dynamic_boundaries_enable, momentum_update_factor,
                  num_bins):
    """utils/ops.py:435-464: score (B,1,N) -> (boundaries, bool mask (B,1,N,nb))."""
    B, H, N = attention_point_score.shape
    if bin_boundaries is not None:
        bin_boundaries = [item.to(attention_point_score.device) for item in bin_boundaries]
    z = stage_zscore(attention_point_score.reshape(B * H, N))
    if dynamic_boundaries_enable:
        bin_boundaries = update_sampling_score_bin_boundary(bin_boundaries, z, num_bins, momentum_update_factor)
    dummy = torch.zeros((B * H, N, 1), dtype=torch.float32, device=z.device)
    member, _, _, _ = stage_bin_assign(z, dummy, bin_boundaries[0], bin_boundaries[1], False)
    return bin_boundaries, _member_to_mask(member, num_bins).reshape(B, H, N, num_bins)


def calculate_num_points_to_choose(bin_prob, max_num_points, total_points_to_choose):
    """utils/ops.py:385-432."""
    return stage_alloc_counts(bin_prob, max_num_points, total_points_to_choose)


def generating_downsampled_index(M, attention_point_score, bin_points_mask, bin_sample_mode, boltzmann_t,
                                 k_point_to_choose, noise=None):
    """utils/ops.py:467-619: score (B,1,N), mask (B,1,N,nb) bool, k (B,nb) -> idx (B,1,M) int64.
    `noise` (B*nb, N) is the Exp(1) draw of torch.multinomial; drawn on the device when None."""
    if bin_sample_mode not in SAMPLE_MODES:
        raise ValueError("Please check the setting of bin sample mode. It must be topk, multinomial or random!")
    B, _, N, nb = bin_points_mask.shape
    weights = (1 << torch.arange(nb, device=bin_points_mask.device, dtype=torch.int32))
    member = (bin_points_mask.reshape(B, N, nb).to(torch.int32) * weights).sum(-1).to(torch.uint8)
    score = _f32c(attention_point_score.reshape(B, N))
    z = stage_zscore(score)
    idx = stage_bin_select(score, z, member, k_point_to_choose.to(torch.int32).contiguous(), M, bin_sample_mode,
                           boltzmann_t, noise)
    return idx.reshape(B, 1, M)

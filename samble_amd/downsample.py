"""Drop-in `DownSampleToken` (reference models/downsample.py:15-378) on the MI355X kernels.

Same constructor (`Cls(config.downsample, layer)`), `forward(x, x_xyz=None)` contract
(`((x_ds (B,C,M), idx (B,1,M) int64), (None, None))`), parameter names / state_dict keys
(`bin_tokens`, `q_conv.weight`, `k_conv.weight`, `v_conv.weight`, + `bn1/ffn/bn2` with `res`) and
side-effect attributes (`attention_point_score`, `bin_boundaries`, `bin_points_mask`,
`bin_weights_beforerelu`, `k_point_to_choose`, `idx`, `attention_bins_beforesoftmax`) as the
reference, so `cls_model` / `seg_model` can use it unchanged and reference checkpoints load.

What differs is how the work is done: two autograd nodes over hand-written HIP kernels, the QKV
projection (one kernel producing point-major rows; backward dx / dW / dtokens) and the sampler core:
kNN build -> attention pass 1 over all rows (softmax statistics, token logits, the exact sparse column score
of the K neighbour entries of every row) -> batch quantiles -> [RCCL all-reduce of nb-1 floats] -> bins /
counts / per-bin selection -> attention pass 2 over the M sampled rows; backward over the M sampled rows only.

Memory held between forward and backward, per layer (B=32, N=2048 -> 1024; stress B=16, N=8192 -> 4096):
  default (split-bf16, asm dot, sparse_* score: MAP_FREE)   qkv 101 MB + P rows of the sampled points (B,M,ld) 269 MB
      + two operand images 101 MB (stress: 202 MB + 2.2 GB + 202 MB); nothing beyond the outputs under no_grad
  logit-map pipeline (MAP_FREE = False, fp32-MFMA mode, asm l2)   the (B,N,ld) logit map 545 MB (stress 4.3 GB)
      exists during the forward in any case and is kept for the backward only when a gradient is wanted
  dense idx modes (col_sum, row_std)   the single-pass flash kernels: O (B,N,D) 34 MB, no N x N tensor
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import ops


def _check_sizes(name: str, N: int, M: int, K: Optional[int] = None, strict: bool = False) -> None:
    """A layer call's sizes, checked on the host before any launch: the reference fails inside topk for these (M > N:
    "selected index k out of range"); kernels handed such sizes would write past their outputs.
    strict: the samplers that also return the dropped points need M < N (the reference's reshape of the empty dropped
    set fails: "cannot reshape tensor of 0 elements")."""
    if not 1 <= M <= N or (strict and M == N):
        raise ValueError(f"{name}: M = {M} points to keep out of N = {N} (need 1 <= M {'<' if strict else '<='} N)")
    if K is not None and N < K:
        raise ValueError(f"{name}: N = {N} points, fewer than the K = {K} neighbours of the score")


def _check_forced(name: str, forced_idx, N: int) -> None:
    """The parity-test hook hands row indices to gather kernels: refuse out-of-range ones here (one host sync, on the hook's
    path only) instead of faulting in a kernel."""
    if forced_idx is None:
        return
    for t in (forced_idx if isinstance(forced_idx, (tuple, list)) else (forced_idx,)):
        if t.numel() and (int(t.min()) < 0 or int(t.max()) >= N):
            raise ValueError(f"{name}: forced_idx holds indices outside 0..{N - 1}")


def _res_ffn(ffn: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """the residual link's Conv1d 128->512, LeakyReLU(0.2), Conv1d 512->128 (models/downsample.py:75-83) on the HIP
    1x1-convolution kernels when the shape is theirs"""
    from . import linear
    w1, w2 = ffn[0].weight, ffn[2].weight
    if ffn[0].bias is None and ffn[2].bias is None and linear.ffn_supported(x, w1, w2):
        return linear.ffn(x, w1, w2)
    return ffn(x)


class _Projection(torch.autograd.Function):
    """q_conv / k_conv / v_conv of the reference (bias-free 1x1 Conv1d, models/downsample.py:54-56,
    124-137) as ONE fp32-MFMA kernel producing point-major [Q|K|V] rows for x and the bin tokens."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, tokens, wq, wk, wv, images="", q_only=False):
        """images "fwd" / "fwd+bwd": also returns the split-bf16 operand images of [Q|K|V] (non-differentiable byte
        tensors), written by the projection kernel itself instead of a split pass over the fp32 rows.
        q_only: the caller reads K and V from the images only (the map-free sampler): the kernel then leaves the K / V
        columns of qkv's point rows unwritten (ops.stage_proj_fwd)."""
        tok = tokens[0]                                  # (C, nt)
        ctx.splits = (wq.shape[0], wk.shape[0], wv.shape[0])
        # the kernels read the three weights where the Conv1d modules hold them (no concatenation launch); the
        # backward reads them again for the token rows only, and dx from the transposed image saved beside them
        w3 = (wq.squeeze(-1), wk.squeeze(-1), wv.squeeze(-1))
        ctx.save_for_backward(x, tok, *w3)
        if not images:
            return ops.stage_proj_fwd(x, tok, w3)
        qkv, imgs = ops.stage_proj_fwd(x, tok, w3, images=images, q_only=q_only)
        if len(imgs) == 6:  # the transposed image of W stays with this node: its own backward reads it
            ctx.save_for_backward(x, tok, *w3, imgs[5])
            imgs = imgs[:5]
        ctx.mark_non_differentiable(*imgs)
        ctx.set_materialize_grads(False)  # (else autograd zero-fills a 50 MB "gradient" per image on the way back)
        return (qkv,) + tuple(imgs)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dqkv, *_):
        if dqkv is None:
            return None, None, None, None, None, None, None
        saved = ctx.saved_tensors   # (read once: torch.utils.checkpoint's unpack hooks refuse a second access)
        x, tok = saved[:2]
        w = tuple(saved[2:5])
        w_tr = saved[5] if len(saved) > 5 else None
        need_dx = ctx.needs_input_grad[0]
        need_dw = any(ctx.needs_input_grad[1:])
        dx, dw, dtok = ops.stage_proj_bwd(dqkv, x, tok, w, need_dx, need_dw, w_tr=w_tr)
        if not need_dw:
            return dx, None, None, None, None, None, None
        a, b, c = ctx.splits
        dtokens = dtok.unsqueeze(0) if ctx.needs_input_grad[1] else None
        return (dx, dtokens, dw[:a].unsqueeze(-1), dw[a:a + b].unsqueeze(-1), dw[a + b:].unsqueeze(-1), None, None)


# Two-pass forward with the logit map kept in HBM (csrc/attn_map.hip) for the sparse_* score modes;
# False selects the single-pass flash kernel (csrc/attn_fwd.hip) for A/B runs.
TWO_PASS = True
# With the split-bf16 kernels, asm "dot" and a sparse_* score mode the N x (N+nt) logit map is not built at all
# (csrc/attn_tri.hip: attn_stats_nl_tri / attn_rows_rc_tri): pass 1 keeps the K neighbour logits of each row, pass 2
# recomputes the M sampled rows and hands their P rows to the backward.  False keeps the map (A/B runs, tests).
MAP_FREE = True
# A single rank runs the integer tail (score + z + batch quantiles + boundaries + bins + counts) as ONE launch
# (csrc/chain.hip select_chain_kernel); False keeps the two launches a process group needs (A/B runs, tests).
FUSED_CHAIN = True


class _SamplerCore(torch.autograd.Function):
    """qkv (B,N+nt,3D) [differentiable], x (B,C,N) [kNN only] -> x_ds (B,D,M), token logits (B,N,nt)
    [both differentiable] + the integer / score by-products."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, qkv, x, mod, noise, images=None, forced_idx=None):
        B, C, N = x.shape
        D = qkv.shape[2] // 3          # (128: a narrower layer arrives zero-padded, DownSampleToken.forward)
        nt = qkv.shape[1] - N
        nb = mod.num_bins
        q = qkv[:, :N, 0:D]
        k = qkv[:, :, D:2 * D]
        v = qkv[:, :, 2 * D:3 * D]

        smap = None
        map_free = False
        plan = None  # (member, cap, w_pre, w, counts) when the fused select chain ran
        if mod.bin_boundaries is not None:
            mod.bin_boundaries = [item.to(x.device) for item in mod.bin_boundaries]
        imgs = None
        if images is not None and mod.idx_mode in ("col_sum", "row_std"):
            raise ops._lib.SambleError("qkv came from the image-writing projection but a dense score mode is selected")
        if mod.idx_mode in ("col_sum", "row_std") and mod.asm == "l2":
            # dense statistics with l2 scoring (models/downsample.py:154-189 + 315-320): the logit map is in HBM on
            # this path anyway; its column sums / row deviations are two torch reductions over exp(S - lse)
            imgs = ops.stage_tri_split_qkv(qkv, N, for_backward=ctx.needs_input_grad[0]) if ops.MATRIX_MODE == "tri" else None
            smap, lse, tok = ops.stage_attn_stats(q, k, N, nt, "l2", images=imgs[:2] if imgs else None)
            A = torch.exp(smap[:, :, :N] - lse.unsqueeze(-1))
            stat = A.sum(dim=1) if mod.idx_mode == "col_sum" else torch.std(A, dim=-1)
            del A
            score, z = ops.stage_stat_score(stat.contiguous())
            nn_idx = torch.empty((B, N, 0), dtype=torch.int32, device=x.device)
            indeg = torch.empty((B, 0), dtype=torch.int32, device=x.device)
        elif mod.idx_mode in ("col_sum", "row_std"):
            # dense statistics of the attention map: no neighbour lists involved
            if mod.idx_mode == "row_std":
                O, lse, tok, stat = ops.stage_attn_fwd(q, k, v, N, nt, want_row_std=True)
            else:
                O, lse, tok = ops.stage_attn_fwd(q, k, v, N, nt)
                stat = ops.stage_attn_colsum(q, k, lse)
            score, z = ops.stage_stat_score(stat)
            nn_idx = torch.empty((B, N, 0), dtype=torch.int32, device=x.device)
            indeg = torch.empty((B, 0), dtype=torch.int32, device=x.device)
        else:
            nn_idx = ops.stage_knn(x, x, mod.K)
            map_free_ok = MAP_FREE and TWO_PASS and ops.MATRIX_MODE == "tri" and mod.asm == "dot" and mod.K in (16, 32)
            if images is not None and not map_free_ok:
                # the projection left the K / V columns of qkv's point rows unwritten (q_only) because the module decided
                # on the map-free forward; a switch flipped between the two calls must not send such a qkv down a
                # pipeline that reads them
                raise ops._lib.SambleError("qkv came from the image-writing projection (K / V rows unwritten) but the "
                                           "map-free forward is no longer selected")
            if map_free_ok:
                need_bwd = ctx.needs_input_grad[0]
                if images is not None and (len(images) == 5 or not need_bwd):
                    imgs = images
                elif images is not None:  # (qkv came with its K / V point rows unwritten: nothing to split from)
                    raise ops._lib.SambleError("the projection's images lack the backward pair although a gradient is wanted")
                else:
                    imgs = ops.stage_tri_split_qkv(qkv, N, for_backward=need_bwd)
                chain = mod._chain_usable(B, N, nb)
                # the pass also accumulates the score statistics of the K neighbour entries of every row
                fused = N <= 8192   # LDS accumulators of the pass; longer clouds take the neighbour-logit array
                sws = ops.score_workspace(B, N, nb if chain else None, x.device) if fused else None
                nn_sorted, masks = ops.stage_nn_prepare(nn_idx, clear=sws)  # (zeroes the score workspace on its way)
                nl, lse, tok, sws = ops.stage_attn_stats_nl(
                    imgs[0], imgs[1], masks, B, N, nt, mod.K, D, want_nl=not fused,
                    score=(nn_sorted, mod.idx_mode, nb if chain else None) if fused else None, cleared_ws=sws)
                del masks
                if chain and ops.single_rank() and FUSED_CHAIN:
                    # one launch: score + z + batch quantiles + boundaries + bins + counts
                    (score, z, indeg, quant, mod.bin_boundaries, *plan, cws) = ops.stage_select_chain(
                        lse, tok, nn_sorted, mod.idx_mode, nb, mod.dynamic_boundaries_enable, mod.bin_boundaries,
                        mod.momentum_update_factor, mod.relu_mean_order == "relu_mean", mod.M, smap=nl,
                        compact=not fused, ws=sws, watch=mod._chain_watch)
                elif chain:
                    # two launches with the ranks' exchange between them (reference utils/ops.py:191-199): the nb-1
                    # quantile sums travel with a validity count, bin_plan divides by it -- no launch for the `/ world`
                    score, z, indeg, quant, cws = ops.stage_score_quantiles(nl, lse, nn_sorted, mod.idx_mode, nb,
                                                                            mod.dynamic_boundaries_enable,
                                                                            compact=not fused, ws=sws,
                                                                            watch=mod._chain_watch, counted=True)
                    if quant is not None:
                        quant = ops.world_sum(quant)
                    mod.bin_boundaries, *plan = ops.stage_bin_plan(z, tok, quant, mod.bin_boundaries, nb,
                                                                   mod.momentum_update_factor,
                                                                   mod.relu_mean_order == "relu_mean", mod.M, cws,
                                                                   watch=mod._chain_watch, counted=True)
                else:
                    score, z, indeg = ops.stage_sparse_score_map(nl, lse, nn_sorted, mod.idx_mode, compact=not fused,
                                                                 ws=sws)
                map_free = True
            elif TWO_PASS or mod.asm == "l2":
                # S once into HBM; the sampled rows' P V (pass 2) and the backward re-read it
                imgs = ops.stage_tri_split_qkv(qkv, N, for_backward=ctx.needs_input_grad[0]) if ops.MATRIX_MODE == "tri" else None
                smap, lse, tok = ops.stage_attn_stats(q, k, N, nt, mod.asm, images=imgs[:2] if imgs else None)
                if mod._chain_usable(B, N, nb):
                    # score + z + batch quantiles, then boundaries + bins + counts: two launches, the rank
                    # average of the quantiles (reference utils/ops.py:191-199) in between
                    score, z, indeg, quant, cws = ops.stage_score_quantiles(smap, lse, nn_idx, mod.idx_mode, nb,
                                                                            mod.dynamic_boundaries_enable,
                                                                            watch=mod._chain_watch, counted=True)
                    if quant is not None:
                        quant = ops.world_sum(quant)
                    mod.bin_boundaries, *plan = ops.stage_bin_plan(z, tok, quant, mod.bin_boundaries, nb,
                                                                   mod.momentum_update_factor,
                                                                   mod.relu_mean_order == "relu_mean", mod.M, cws,
                                                                   watch=mod._chain_watch, counted=True)
                else:
                    score, z, indeg = ops.stage_sparse_score_map(smap, lse, nn_idx, mod.idx_mode)
            else:
                O, lse, tok = ops.stage_attn_fwd(q, k, v, N, nt)
                score, z, indeg = ops.stage_sparse_score(q, k, lse, nn_idx, mod.idx_mode)

        if plan is not None:
            member, cap, w_pre, w, counts = plan
        else:
            if mod.dynamic_boundaries_enable:
                quant = ops.world_average(ops.stage_batch_quantiles(z, nb))
                mod.bin_boundaries = ops.blend_boundaries(mod.bin_boundaries, quant, nb, mod.momentum_update_factor)
            member, cap, w_pre, w = ops.stage_bin_assign(z, tok, mod.bin_boundaries[0], mod.bin_boundaries[1],
                                                         mod.relu_mean_order == "relu_mean")
            counts = ops.stage_alloc_counts(w, cap, mod.M)
        idx = ops.stage_bin_select(score, z, member, counts, mod.M, mod.bin_sample_mode, mod.boltzmann_T, noise)
        if forced_idx is not None:
            # parity-test hook: the rows to gather are given (the reference's own sampled indices), everything before
            # this line still ran and is returned as measured
            idx = forced_idx.reshape(B, mod.M).to(device=x.device, dtype=torch.int64).contiguous()
        # saved for backward: (qkv, O | x_ds, lse, idx[, map[, K transposed image, V row image]]) -- everything through
        # save_for_backward (hooks and version checks see it), nothing when no gradient is wanted
        ctx.pmap = False
        if map_free:
            x_ds, pmap = ops.stage_attn_rows_recompute(imgs[0], imgs[1], imgs[2], lse, idx, N, nt, need_bwd, D)
            if need_bwd:  # the P rows of the sampled points stand in for the logit map in the backward
                ctx.save_for_backward(qkv, x_ds, lse, idx, pmap, imgs[3], imgs[4])
                ctx.pmap = True
        elif smap is not None:
            x_ds = ops.stage_attn_rows(smap, lse, v, idx, N, nt, v_image=imgs[2] if imgs else None)
            if ctx.needs_input_grad[0]:
                extra = (imgs[3], imgs[4]) if imgs is not None and len(imgs) == 5 else ()
                ctx.save_for_backward(qkv, x_ds, lse, idx, smap, *extra)
        else:
            x_ds = ops.stage_gather_rows(O, idx)
            ctx.save_for_backward(qkv, O, lse, idx)
        ctx.dims = (N, nt, D)
        ctx.asm = mod.asm
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(idx, score, z, member, cap, w_pre, counts, indeg, nn_idx)
        return x_ds, tok, idx, score, z, member, cap, w_pre, counts, indeg, nn_idx

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_xds, g_tok, *_):
        saved = ctx.saved_tensors   # (read once: torch.utils.checkpoint's unpack hooks refuse a second access)
        qkv, O, lse, idx = saved[:4]
        smap = saved[4] if len(saved) > 4 else None
        images = tuple(saved[5:7]) if len(saved) > 6 else None
        N, nt, D = ctx.dims
        q = qkv[:, :N, 0:D]
        k = qkv[:, :, D:2 * D]
        v = qkv[:, :, 2 * D:3 * D]
        if g_xds is not None:
            dqkv = torch.empty_like(qkv)
            tok_by_kernel = smap is not None and ops.MATRIX_MODE == "tri"  # (that backward clears dQ's token rows)
            if smap is not None:  # O is x_ds (B,D,M) here
                ops.stage_attn_rows_bwd(q, k, v, smap, lse, O, idx, g_xds, N, nt, dqkv[:, :N, 0:D],
                                        dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:3 * D], ctx.asm,
                                        images=images,
                                        variant=ops.ROWS_BWD_PMAP if ctx.pmap else 0, dq_token_rows=tok_by_kernel)
            else:
                ops.stage_attn_bwd(q, k, v, O, lse, idx, g_xds, N, nt, dqkv[:, :N, 0:D], dqkv[:, :, D:2 * D],
                                   dqkv[:, :, 2 * D:3 * D])
            if nt and not tok_by_kernel:
                dqkv[:, N:, 0:D].zero_()
        else:
            dqkv = torch.zeros_like(qkv)
        if g_tok is not None and nt:
            # token logits = scale * Q K_tok^T (attention_bins_beforesoftmax feeds the optional
            # token loss, reference utils/loss.py:17-27): tiny (N x nt) products
            scale = 1.0 / math.sqrt(D)
            if ctx.asm == "l2":  # logits -|q - k_tok|^2 / sqrt(D)
                dqkv[:, :N, 0:D] += 2 * scale * (torch.matmul(g_tok, k[:, N:, :]) - q * g_tok.sum(-1, keepdim=True))
                dqkv[:, N:, D:2 * D] += 2 * scale * (torch.matmul(g_tok.transpose(1, 2), q)
                                                     - k[:, N:, :] * g_tok.sum(1).unsqueeze(-1))
            else:
                dqkv[:, :N, 0:D] += scale * torch.matmul(g_tok, k[:, N:, :])
                dqkv[:, N:, D:2 * D] += scale * torch.matmul(g_tok.transpose(1, 2), q)
        return dqkv, None, None, None, None, None


class DownSampleToken(nn.Module):
    """Shape-specific point cloud downsampling (SAMBLE) — see module docstring.

    Inputs:  x (B, C, N) features; x_xyz unused (as in the reference).
    Outputs: ((x_ds (B, C, M), index_down (B, H=1, M) int64), (None, None))."""

    def __init__(self, config_ds, layer):
        super().__init__()
        self.M = config_ds.M[layer]
        self.K = config_ds.K
        self.asm = config_ds.asm[layer]
        self.res = config_ds.res.enable[layer]
        self.ff = config_ds.res.ff[layer]
        self.num_heads = config_ds.num_heads[layer]
        self.idx_mode = config_ds.idx_mode[layer]
        self.relu_mean_order = config_ds.bin.relu_mean_order[layer]
        self.num_bins = config_ds.bin.num_bins[layer]

        q_in, q_out = config_ds.q_in[layer], config_ds.q_out[layer]
        k_in, k_out = config_ds.k_in[layer], config_ds.k_out[layer]
        v_in, v_out = config_ds.v_in[layer], config_ds.v_out[layer]
        self.q_depth = int(q_out / self.num_heads)
        self.k_depth = int(k_out / self.num_heads)
        self.v_depth = int(v_out / self.num_heads)
        self.q_conv = nn.Conv1d(q_in, q_out, 1, bias=False)
        self.k_conv = nn.Conv1d(k_in, k_out, 1, bias=False)
        self.v_conv = nn.Conv1d(v_in, v_out, 1, bias=False)

        self.token_mode = config_ds.bin.token_mode[layer]
        if self.token_mode == "multi_token":
            self.bin_tokens = nn.Parameter(torch.normal(mean=0, std=1 / math.sqrt(q_in), size=(1, q_in, self.num_bins)))
        elif self.token_mode == "one_token":
            self.bin_tokens = nn.Parameter(torch.normal(mean=0, std=1 / math.sqrt(q_in), size=(1, q_in, 1)))
        else:
            raise NotImplementedError

        if self.res:
            self.bn1 = nn.BatchNorm1d(v_out)
            if self.ff:
                self.ffn = nn.Sequential(nn.Conv1d(128, 512, 1, bias=False), nn.LeakyReLU(negative_slope=0.2),
                                         nn.Conv1d(512, 128, 1, bias=False))
                self.bn2 = nn.BatchNorm1d(v_out)

        self.scaling_factor = config_ds.bin.scaling_factor[layer]
        self.bin_sample_mode = config_ds.bin.sample_mode[layer]
        self.bin_norm_mode = config_ds.bin.norm_mode[layer]
        self.momentum_update_factor = config_ds.bin.momentum_update_factor[layer]
        self.dynamic_boundaries_enable = config_ds.bin.dynamic_boundaries_enable
        if config_ds.bin.dynamic_boundaries_enable:
            self.bin_boundaries = None
        else:
            values = list(config_ds.bin.bin_boundaries[layer])
            self.bin_boundaries = [
                torch.asarray([float("inf")] + values).reshape(1, 1, 1, self.num_bins),
                torch.asarray(values + [float("-inf")]).reshape(1, 1, 1, self.num_bins),
            ]
        self.boltzmann_enable = config_ds.boltzmann.enable[layer]
        self.boltzmann_T = config_ds.bin.boltzmann_T[layer]
        self.boltzmann_norm_mode = config_ds.boltzmann.norm_mode[layer]
        self.token_orthognonal_loss_factor = config_ds.bin.token_orthognonal_loss_factor

        if self.asm not in ("dot", "l2"):
            raise NotImplementedError
        if self.num_heads != 1:
            raise NotImplementedError("DownSampleToken requires num_heads == 1 (reference utils/check_config.py:158)")
        if not 2 <= self.num_bins <= 8:
            # the kernels carry a point's bin as one bit of a byte and a cloud's bins in one workgroup's registers; the
            # shipped configs have 6 (cls) and 4 (seg) -- refused here, by name, rather than by a kernel argument check
            raise NotImplementedError(f"DownSampleToken: num_bins must be in 2..8 (got {self.num_bins})")
        self._member_bits = None
        self._chain_watch = ops.ChainWatch()

    # -- reference-visible state ---------------------------------------------------------------
    @property
    def bin_points_mask(self):
        """(B,1,N,nb) bool, built on demand from the kernel's membership bits."""
        if self._member_bits is None:
            return None
        return ops._member_to_mask(self._member_bits, self.num_bins)

    def _chain_usable(self, B: int, N: int, nb: int) -> bool:
        """The fused select chain takes this shape and has not given up on this layer.  The call that first finds the
        give-up word also repairs the state the give-up may have left: a chain that bails never writes the boundaries
        (csrc/chain.hip), so after a later call they are the last valid ones; after a FIRST call they are still the NaN
        they were allocated with -- back to "no boundaries yet" (one synchronising read, on this path only)."""
        watch = self._chain_watch
        if watch.poll() and self.dynamic_boundaries_enable and self.bin_boundaries is not None:
            if bool(torch.isnan(self.bin_boundaries[0]).any()):
                self.bin_boundaries = None
        return ops.chain_supported(B, N, nb) and not watch.observed

    def forward(self, x, x_xyz=None, noise: Optional[torch.Tensor] = None, forced_idx: Optional[torch.Tensor] = None):
        """noise: the (B*nb, N) Exp(1) draw torch.multinomial makes inside (None: drawn on the device).
        forced_idx (B,1,M) | (B,M): parity-test hook -- gather these rows instead of the ones the selection produced
        (scores, bins, counts and boundaries are still computed and published as usual)."""
        B, C, N = x.shape
        if not x.is_cuda:
            raise ops._lib.SambleError("samble_amd.DownSampleToken runs on the GPU only (no CPU fallback)")
        _check_sizes("DownSampleToken", N, self.M, self.K)
        _check_forced("DownSampleToken", forced_idx, N)
        self._chain_usable(B, N, self.num_bins)   # (looks at the mailbox; repairs the boundary state if it finds the word)
        if self._chain_watch.observed and not self._chain_watch.reported:
            # raised ONCE, at the first call after the status word arrived -- whichever call site saw it first; from
            # here on the layer runs the stand-alone stage kernels (no grid barrier) on a boundary state that is either
            # the last valid one or none (first call again), so a caller that catches this and carries on gets valid results
            self._chain_watch.reported = True
            raise ops._lib.SambleError(
                "SAMBLE_E_TIMEOUT: a grid barrier of the fused select chain gave up in an earlier forward of this layer "
                "(its workgroups were not all resident); that forward's selection was a placeholder. The layer has "
                "switched to the stand-alone stage kernels.")
        if not (self.q_depth == self.k_depth == self.v_depth == C and C <= 128):
            # any other width (the reference's constructor takes any q_in / q_out, models/downsample.py:33-56; no shipped
            # config has one): the matrix part as the reference's expression in torch on the device, the neighbour search,
            # the scores and the whole selection on the HIP stage kernels
            return self._forward_wide(x, noise, forced_idx)
        wq, wk, wv, tokens = self.q_conv.weight, self.k_conv.weight, self.v_conv.weight, self.bin_tokens
        x_in = x
        if C < 128:
            # a narrower layer runs on the 128-channel kernels with zero channels behind its own: distances, logits and
            # products are unchanged by them; the logits' 1 / sqrt(C) enters through the weights (the kernels divide by
            # sqrt(128)): dot: q.k is linear in W_q, which takes all of sqrt(128 / C); l2: -|q - k|^2 is quadratic in
            # (q, k) TOGETHER, so W_q and W_k (token rows included: they go through W_k) take (128 / C)^(1/4) each
            pad = 128 - C
            grow = lambda w: F.pad(w, (0, 0, 0, pad, 0, pad))            # (C,C,1) -> (128,128,1)
            if self.asm == "l2":
                s4 = (128.0 / C) ** 0.25
                wq, wk, wv = grow(wq * s4), grow(wk * s4), grow(wv)
            else:
                wq, wk, wv = grow(wq * math.sqrt(128.0 / C)), grow(wk), grow(wv)
            tokens = F.pad(tokens, (0, 0, 0, pad))                      # (1,C,nt) -> (1,128,nt)
            x = F.pad(x, (0, 0, 0, pad))                                # (B,C,N) -> (B,128,N)
        # (B, N+nt, 3D) point-major rows [Q|K|V]; rows N.. are the bin tokens
        # the map-free forward takes its operand images straight from the projection kernel (no split pass over qkv)
        fused_images = (MAP_FREE and TWO_PASS and ops.MATRIX_MODE == "tri" and self.asm == "dot" and self.K in (16, 32)
                        and self.idx_mode not in ("col_sum", "row_std"))
        if fused_images:
            need_bwd = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
            # (everything downstream of the projection reads K and V from the images: their fp32 point rows stay unwritten)
            qkv, *images = _Projection.apply(x, tokens, wq, wk, wv, "fwd+bwd" if need_bwd else "fwd", True)
        else:
            images = None
            qkv = _Projection.apply(x, tokens, wq, wk, wv)

        (x_ds, tok, idx, score, z, member, cap, w_pre, counts, indeg, nn_idx) = _SamplerCore.apply(
            qkv, x.detach(), self, noise, tuple(images) if images is not None else None, forced_idx)

        index_down = idx.unsqueeze(1)
        if C < 128:
            x_ds = x_ds[:, :C, :].contiguous()
        if self.res is True:
            x_ds = self.res_block(x_in, x_ds, index_down)

        self.attention_point_score = score.unsqueeze(1)
        self._member_bits = member
        self.bin_weights_beforerelu = w_pre
        self.k_point_to_choose = counts
        self.idx = index_down
        self.attention_bins_beforesoftmax = tok.unsqueeze(1)
        self.knn_idx = nn_idx
        self.knn_indegree = indeg
        self.normalized_score = z
        self.max_num_points = cap
        return (x_ds, index_down), (None, None)

    def _forward_wide(self, x, noise, forced_idx):
        """Channel widths the attention kernels are not built for (> 128, or q_out != k_out ...): models/downsample.py:
        112-290 with Q K^T, the softmax of the sampled rows and P V as torch expressions (torch's autograd carries their
        gradients; the (B, N, N + nt) logits live in HBM as in the reference), and everything that decides WHICH rows --
        kNN, the seven score modes, z-scores, batch quantiles, boundaries, bins, counts, the draw -- on the same HIP stage
        kernels as the 128-channel path."""
        B, C, N = x.shape
        nt, nb, D = self.bin_tokens.shape[2], self.num_bins, self.q_depth
        if not (self.q_depth == self.k_depth):
            raise ValueError("q_out and k_out must agree (the logits contract over them)")
        xt = torch.cat((x, self.bin_tokens.expand(B, -1, -1)), dim=2)
        q, k, v = self.q_conv(x), self.k_conv(xt), self.v_conv(xt)               # (B,D,N), (B,D,N+nt), (B,Dv,N+nt)
        qk = torch.matmul(q.transpose(1, 2), k)                                  # (B,N,N+nt)
        if self.asm == "l2":
            qk = -(q.square().sum(1).unsqueeze(2) + k.square().sum(1).unsqueeze(1) - 2.0 * qk)
        logits = qk / math.sqrt(D)
        with torch.no_grad():
            lse = torch.logsumexp(logits, dim=-1).contiguous()
            tok = logits[:, :, N:].contiguous()
            if self.idx_mode in ("col_sum", "row_std"):
                A = torch.exp(logits[:, :, :N] - lse.unsqueeze(-1))
                stat = A.sum(dim=1) if self.idx_mode == "col_sum" else torch.std(A, dim=-1)
                del A
                score, z = ops.stage_stat_score(stat.contiguous())
                nn_idx = torch.empty((B, N, 0), dtype=torch.int32, device=x.device)
                indeg = torch.empty((B, 0), dtype=torch.int32, device=x.device)
            else:
                nn_idx = ops.stage_knn(x.detach(), x.detach(), self.K)
                ld = ops.attn_map_row_stride(N, nt)
                smap = logits.new_zeros((B, N, ld))
                smap[:, :, :N + nt] = logits
                score, z, indeg = ops.stage_sparse_score_map(smap, lse, nn_idx, self.idx_mode)
                del smap
            if self.bin_boundaries is not None:
                self.bin_boundaries = [item.to(x.device) for item in self.bin_boundaries]
            if self.dynamic_boundaries_enable:
                quant = ops.world_average(ops.stage_batch_quantiles(z, nb))
                self.bin_boundaries = ops.blend_boundaries(self.bin_boundaries, quant, nb, self.momentum_update_factor)
            member, cap, w_pre, w = ops.stage_bin_assign(z, tok, self.bin_boundaries[0], self.bin_boundaries[1],
                                                         self.relu_mean_order == "relu_mean")
            counts = ops.stage_alloc_counts(w, cap, self.M)
            idx = ops.stage_bin_select(score, z, member, counts, self.M, self.bin_sample_mode, self.boltzmann_T, noise)
            if forced_idx is not None:
                idx = forced_idx.reshape(B, self.M).to(device=x.device, dtype=torch.int64).contiguous()
        rows = torch.gather(logits, 1, idx.unsqueeze(-1).expand(-1, -1, N + nt))    # (B,M,N+nt)
        x_ds = torch.matmul(torch.softmax(rows, dim=-1), v.transpose(1, 2)).transpose(1, 2).contiguous()   # (B,Dv,M)
        index_down = idx.unsqueeze(1)
        if self.res is True:
            x_ds = self.res_block(x, x_ds, index_down)
        self.attention_point_score = score.unsqueeze(1)
        self._member_bits = member
        self.bin_weights_beforerelu = w_pre
        self.k_point_to_choose = counts
        self.idx = index_down
        self.attention_bins_beforesoftmax = logits[:, :, N:].unsqueeze(1)
        self.knn_idx = nn_idx
        self.knn_indegree = indeg
        self.normalized_score = z
        self.max_num_points = cap
        return (x_ds, index_down), (None, None)

    def res_block(self, x, x_ds, idx):
        """models/downsample.py:292-298 (the gather picks channel 0 only, as in the reference)."""
        x_tmp = torch.gather(x, dim=-1, index=idx)
        x_res = self.bn1(x_ds + x_tmp)
        if self.ff == True:  # noqa: E712  (reference semantics)
            x_tmp = _res_ffn(self.ffn, x_res)
            x_res = self.bn2(x_ds + x_tmp)
        return x_res

    def output_variable_calculatio(self):
        """models/downsample.py:346-362 (name kept, typo included: scripts call it)."""
        mask = self.bin_points_mask
        B, _, _, num_bins = mask.shape
        index_batch, _, index_point, index_bin = torch.where(mask)
        self.idx_chunks = [
            [index_point[(index_bin == i) & (index_batch == j)].reshape(1, -1) for j in range(B)]
            for i in range(num_bins)
        ]
        self.bin_prob = self.bin_weights_beforerelu

    def output_variables(self, *args):
        """models/downsample.py:364-378."""
        variables = None
        for i, key in enumerate(args):
            if i == 0:
                variables = getattr(self, key)
            elif i == 1:
                variables = (variables,) + (getattr(self, key),)
            else:
                variables = variables + (getattr(self, key),)
        return variables


GLOBAL_SPARSE_MODES = ("sparse_row_sum", "sparse_row_std", "sparse_col_sum", "sparse_col_avg", "sparse_col_sqr",
                       "sparse_col_sum_sqr")


def _global_sparse_statistic(mode, N, K, stat_of):
    """DownSampleGlobal.idx_selection's sparse branch (reference models/downsample.py:1383-1401) from the kernels'
    DownSampleToken statistics.  stat_of(token_mode) -> (statistic (B,N), in-degree (B,N) int32).  The reference's Global
    formulas differ from DownSampleToken's: the row deviation is torch.std over ALL N entries of the masked row (zeros
    included), the column averages divide by the raw in-degree (>= 1: every point lists itself), and
    sparse_col_sum_sqr exists."""
    if mode == "sparse_row_sum":
        return stat_of("sparse_row_sum")[0]
    if mode == "sparse_row_std":
        # the kernel gives s1 = sum and the unbiased deviation over the K picked entries; the other N-K entries are 0
        s1 = stat_of("sparse_row_sum")[0].double()
        sk = stat_of("sparse_row_std")[0].double()
        s2 = sk * sk * (K - 1) + s1 * s1 / K
        return torch.sqrt(torch.clamp((s2 - s1 * s1 / N) / (N - 1), min=0.0)).float()
    cs, indeg = stat_of("sparse_col_sum")
    num = indeg.float()
    if mode == "sparse_col_sum":
        return cs
    if mode == "sparse_col_avg":
        return cs / num
    if mode == "sparse_col_sqr":
        return cs / num / num
    if mode == "sparse_col_sum_sqr":
        return 0.5 * (cs / num / num) + 0.5 * cs
    raise ValueError("Please check the setting of idx mode!")



class _GlobalCore(torch.autograd.Function):
    """qkv (B,N,3D) -> x_ds (B,D,M), x_dropped (B,D,N-M) + indices, for DownSampleGlobal."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, qkv, x, mod):
        B, C, N = x.shape
        D = mod.q_depth
        q, k, v = qkv[:, :, 0:D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:3 * D]
        if mod.idx_mode == "row_std":
            O, lse, _, stat = ops.stage_attn_fwd(q, k, v, N, 0, want_row_std=True)
            col = ops.stage_attn_colsum(q, k, lse)
        else:
            O, lse, _ = ops.stage_attn_fwd(q, k, v, N, 0)
            col = ops.stage_attn_colsum(q, k, lse)
            if mod.idx_mode == "col_sum":
                stat = col
            elif mod.idx_mode in GLOBAL_SPARSE_MODES:
                nn_idx = ops.stage_knn(x, x, mod.K)
                stat = _global_sparse_statistic(
                    mod.idx_mode, N, mod.K,
                    lambda m: (lambda r: (r[0], r[2]))(ops.stage_sparse_score(q, k, lse, nn_idx, m))).contiguous()
            else:
                raise ValueError("Please check the setting of idx mode!")
        idx = ops.stage_topk_indices(stat, mod.M, largest=True)
        idx_dropped = ops.stage_topk_indices(col, N - mod.M, largest=False)
        if mod._forced_idx is not None:
            idx, idx_dropped = mod._forced_idx
        x_ds = ops.stage_gather_rows(O, idx)
        x_dropped = ops.stage_gather_rows(O, idx_dropped)
        ctx.save_for_backward(qkv, O, lse, idx, idx_dropped)
        ctx.dims = (N, D)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(idx, idx_dropped, stat)
        return x_ds, x_dropped, idx, idx_dropped, stat

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_ds, g_dropped, *_):
        qkv, O, lse, idx, idx_dropped = ctx.saved_tensors
        N, D = ctx.dims
        B = qkv.shape[0]
        # both outputs gather rows of the same attention output: fold their gradients into one
        # (B,D,N) gradient over all rows (a row can be in both sets unless idx_mode is col_sum)
        g_all = torch.zeros((B, D, N), dtype=torch.float32, device=qkv.device)
        if g_ds is not None:
            g_all.scatter_add_(2, idx.unsqueeze(1).expand(-1, D, -1), g_ds)
        if g_dropped is not None:
            g_all.scatter_add_(2, idx_dropped.unsqueeze(1).expand(-1, D, -1), g_dropped)
        rows = torch.arange(N, device=qkv.device, dtype=torch.int64).unsqueeze(0).expand(B, -1).contiguous()
        dqkv = torch.empty_like(qkv)
        ops.stage_attn_bwd(qkv[:, :, 0:D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], O, lse, rows, g_all, N, 0,
                           dqkv[:, :, 0:D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:])
        return dqkv, None, None


class _GlobalMapCore(torch.autograd.Function):
    """_GlobalCore for asm l2 / l2+ (models/downsample.py:1347-1350): S = -/+ |q - k|^2 / sqrt(D) through the logit map
    (pass 1 with the norm terms, pass 2 over ALL rows), statistics from the map, backward from the map."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, qkv, x, mod):
        B, C, N = x.shape
        D = mod.q_depth
        q, k, v = qkv[:, :, 0:D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:3 * D]
        smap, lse, _ = ops.stage_attn_stats(q, k, N, 0, mod.asm)
        rows = torch.arange(N, device=qkv.device, dtype=torch.int64).unsqueeze(0).expand(B, -1).contiguous()
        o_all = ops.stage_attn_rows(smap, lse, v, rows, N, 0)                 # (B,D,N): every row's A V
        A = torch.exp(smap[:, :, :N] - lse.unsqueeze(-1))
        col = A.sum(dim=1).contiguous()
        if mod.idx_mode == "col_sum":
            stat = col
        elif mod.idx_mode == "row_std":
            stat = torch.std(A, dim=-1).contiguous()
        elif mod.idx_mode in GLOBAL_SPARSE_MODES:
            nn_idx = ops.stage_knn(x, x, mod.K)
            stat = _global_sparse_statistic(
                mod.idx_mode, N, mod.K,
                lambda m: (lambda r: (r[0], r[2]))(ops.stage_sparse_score_map(smap, lse, nn_idx, m))).contiguous()
        else:
            raise ValueError("Please check the setting of idx mode!")
        del A
        idx = ops.stage_topk_indices(stat, mod.M, largest=True)
        idx_dropped = ops.stage_topk_indices(col, N - mod.M, largest=False)
        if mod._forced_idx is not None:
            idx, idx_dropped = mod._forced_idx
        x_ds = torch.gather(o_all, 2, idx.unsqueeze(1).expand(-1, D, -1))
        x_dropped = torch.gather(o_all, 2, idx_dropped.unsqueeze(1).expand(-1, D, -1))
        ctx.save_for_backward(qkv, o_all, lse, idx, idx_dropped, smap, rows)
        ctx.dims = (N, D, mod.asm)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(idx, idx_dropped, stat)
        return x_ds, x_dropped, idx, idx_dropped, stat

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_ds, g_dropped, *_):
        qkv, o_all, lse, idx, idx_dropped, smap, rows = ctx.saved_tensors
        N, D, asm = ctx.dims
        B = qkv.shape[0]
        g_all = torch.zeros((B, D, N), dtype=torch.float32, device=qkv.device)
        if g_ds is not None:
            g_all.scatter_add_(2, idx.unsqueeze(1).expand(-1, D, -1), g_ds)
        if g_dropped is not None:
            g_all.scatter_add_(2, idx_dropped.unsqueeze(1).expand(-1, D, -1), g_dropped)
        dqkv = torch.empty_like(qkv)
        ops.stage_attn_rows_bwd(qkv[:, :, 0:D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], smap, lse, o_all, rows, g_all, N, 0,
                                dqkv[:, :, 0:D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], asm)
        return dqkv, None, None


class DownSampleGlobal(nn.Module):
    """Drop-in for the reference's APES-style global sampler (models/downsample.py:1232-1405, asm
    'dot' / 'dot-sub' / 'l2' / 'l2+'): softmax(QK^T/sqrt(D)) over the N points, score per `idx_mode`, top-M rows kept and the
    N-M rows with the smallest column sum returned as the dropped set.
    Outputs: ((x_ds (B,C,M), idx (B,1,M)), (x_dropped (B,C,N-M), idx_dropped (B,1,N-M)))."""

    def __init__(self, config_ds, layer):
        super().__init__()
        self.M = config_ds.M[layer]
        self.K = 32
        self.asm = config_ds.asm[layer]
        self.res = config_ds.res.enable[layer]
        self.ff = config_ds.res.ff[layer]
        self.num_heads = config_ds.num_heads[layer]
        self.idx_mode = config_ds.idx_mode[layer]
        q_in, q_out = config_ds.q_in[layer], config_ds.q_out[layer]
        k_in, k_out = config_ds.k_in[layer], config_ds.k_out[layer]
        v_in, v_out = config_ds.v_in[layer], config_ds.v_out[layer]
        self.q_depth = int(q_out / self.num_heads)
        self.k_depth = int(k_out / self.num_heads)
        self.v_depth = int(v_out / self.num_heads)
        if self.res:
            self.bn1 = nn.BatchNorm1d(v_out)
            if self.ff:
                self.ffn = nn.Sequential(nn.Conv1d(128, 512, 1, bias=False), nn.LeakyReLU(negative_slope=0.2),
                                         nn.Conv1d(512, 128, 1, bias=False))
                self.bn2 = nn.BatchNorm1d(v_out)
        self.q_conv = nn.Conv1d(q_in, q_out, 1, bias=False)
        self.k_conv = nn.Conv1d(k_in, k_out, 1, bias=False)
        self.v_conv = nn.Conv1d(v_in, v_out, 1, bias=False)
        self.softmax = nn.Softmax(dim=-1)
        if self.asm not in ("dot", "dot-sub", "l2", "l2+"):
            raise ValueError("Please check the setting of asm!")
        # the attention kernels are built for one head of 128 channels (every shipped config); any other width or head
        # count the reference constructs (models/downsample.py:1248-1279) runs the same expressions in torch on the device
        self._hip_attention = self.num_heads == 1 and q_in == q_out == k_out == v_out == 128
        self._forced_idx = None

    def forward(self, x, x_xyz=None, forced_idx=None):
        """forced_idx = (idx (B,H,M), idx_dropped (B,H,N-M)): parity-test hook -- gather these rows instead of the
        selected ones (the statistic is still computed and published as `attention`)."""
        if not x.is_cuda:
            raise ops._lib.SambleError("samble_amd.DownSampleGlobal runs on the GPU only (no CPU fallback)")
        _check_sizes("DownSampleGlobal", x.shape[2], self.M, self.K if self.idx_mode in GLOBAL_SPARSE_MODES else None,
                     strict=True)
        _check_forced("DownSampleGlobal", forced_idx, x.shape[2])
        if not self._hip_attention:
            return self._forward_generic(x, forced_idx)
        B, N = x.shape[0], x.shape[2]
        self._forced_idx = None if forced_idx is None else (
            forced_idx[0].reshape(B, self.M).to(device=x.device, dtype=torch.int64).contiguous(),
            forced_idx[1].reshape(B, N - self.M).to(device=x.device, dtype=torch.int64).contiguous())
        no_tokens = self.q_conv.weight.new_zeros((1, x.shape[1], 0))
        # attention_scoring (models/downsample.py:1338-1358).  dot-sub: energy = Q (Q^T - K) with Q^T, K both (D, N):
        # energy_ij = <q_i, q_j - k_j> -- dot attention with the keys Q - K, i.e. the key weights W_q - W_k (the convs
        # are linear).  l2 / l2+: the logit-map pipeline (the norm terms enter the logits there).
        wk = (self.q_conv.weight - self.k_conv.weight) if self.asm == "dot-sub" else self.k_conv.weight
        qkv = _Projection.apply(x, no_tokens, self.q_conv.weight, wk, self.v_conv.weight)
        core = _GlobalMapCore if self.asm in ("l2", "l2+") else _GlobalCore
        x_ds, x_dropped, idx, idx_dropped, stat = core.apply(qkv, x.detach(), self)
        self.idx = idx.unsqueeze(1)
        self.attention = stat.unsqueeze(1)
        idx_dropped = idx_dropped.unsqueeze(1)
        if self.res == True:  # noqa: E712
            x_ds = self.res_block(x, x_ds)
        return (x_ds, self.idx), (x_dropped, idx_dropped)

    def _forward_generic(self, x, forced_idx):
        """models/downsample.py:1281-1405 for the widths and head counts the attention kernels are not built for: the
        projections, the (B,H,N,N) softmax and A V as torch expressions on the device (torch's autograd carries their
        gradients; the map lives in HBM as in the reference); the neighbour search of the sparse_* statistics and both
        top-k selections (exact ties by ascending index) on the HIP stage kernels of the 128-channel path."""
        B, _, N = x.shape
        H, M = self.num_heads, self.M
        q = self.q_conv(x).view(B, H, self.q_depth, N).permute(0, 1, 3, 2)                 # (B,H,N,D)
        k = self.k_conv(x).view(B, H, self.k_depth, N)                                     # (B,H,D,N)
        v = self.v_conv(x).view(B, H, self.v_depth, N)
        if self.asm == "dot":
            energy = q @ k
        elif self.asm == "dot-sub":
            energy = q @ (q.transpose(-1, -2) - k)
        else:  # l2 / l2+: utils/ops.py:115-122
            sq = q.square().sum(-1, keepdim=True) - 2.0 * (q @ k) + k.square().sum(-2, keepdim=True)
            energy = -sq if self.asm == "l2" else sq
        A = torch.softmax(energy / math.sqrt(q.shape[-1]), dim=-1)                         # (B,H,N,N)
        with torch.no_grad():
            col = A.sum(dim=-2)
            if self.idx_mode == "col_sum":
                stat = col
            elif self.idx_mode == "row_std":
                stat = torch.std(A, dim=-1)
            elif self.idx_mode in GLOBAL_SPARSE_MODES:
                # idx_selection's sparse branch (1383-1401): the row deviation runs over all N entries of the masked row,
                # the in-degree is used as it is (a column nobody lists: 0/0 = NaN, which top-k ranks first)
                nn_idx = ops.stage_knn(x.detach(), x.detach(), self.K).long()
                mask = torch.zeros((B, N, N), dtype=torch.float32, device=x.device).scatter_(2, nn_idx, 1.0).unsqueeze(1)
                sam = A * mask
                num = mask.sum(dim=-2).expand(-1, H, -1)
                if self.idx_mode == "sparse_row_sum":
                    stat = sam.sum(dim=-1)
                elif self.idx_mode == "sparse_row_std":
                    stat = torch.std(sam, dim=-1)
                else:
                    cs = sam.sum(dim=-2)
                    stat = {"sparse_col_sum": lambda: cs, "sparse_col_avg": lambda: cs / num,
                            "sparse_col_sqr": lambda: cs / num / num,
                            "sparse_col_sum_sqr": lambda: 0.5 * (cs / num / num) + 0.5 * cs}[self.idx_mode]()
                del sam, mask
            else:
                raise ValueError("Please check the setting of idx mode!")
            idx = ops.stage_topk_indices(stat.reshape(B * H, N), M, largest=True).view(B, H, M)
            idx_dropped = ops.stage_topk_indices(col.reshape(B * H, N), N - M, largest=False).view(B, H, N - M)
            if forced_idx is not None:
                idx = forced_idx[0].reshape(B, H, M).to(device=x.device, dtype=torch.int64)
                idx_dropped = forced_idx[1].reshape(B, H, N - M).to(device=x.device, dtype=torch.int64)

        def rows(ix):   # (B,H,m) -> (B, H Dv, m): the selected rows of A times V, heads concatenated along the channels
            a = torch.gather(A, 2, ix.unsqueeze(-1).expand(-1, -1, -1, N))
            o = (a @ v.transpose(-1, -2)).permute(0, 2, 1, 3)                               # (B,m,H,Dv)
            return o.reshape(B, ix.shape[2], -1).permute(0, 2, 1)

        self.idx = idx
        self.attention = stat
        x_ds = rows(idx)
        if self.res == True:  # noqa: E712
            x_ds = self.res_block(x, x_ds)
        return (x_ds, self.idx), (rows(idx_dropped), idx_dropped)

    def res_block(self, x, x_ds):
        x_tmp = torch.gather(x, dim=-1, index=self.idx)
        x_res = self.bn1(x_ds + x_tmp)
        if self.ff == True:  # noqa: E712
            x_tmp = _res_ffn(self.ffn, x_res)
            x_res = self.bn2(x_ds + x_tmp)
        return x_res


class _LocalCore(torch.autograd.Function):
    """x (B,C,N), Conv2d weights -> local-attention output of EVERY point (B,C,N) [differentiable],
    the (B,N,K) attention probabilities and the neighbour lists, for DownSampleLocal."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, wq, wk, wv, K, diff):
        C = x.shape[1]
        w = torch.cat((wq, wk, wv), dim=0).reshape(3 * C, C)
        qkv = ops.stage_proj_fwd(x, x.new_zeros((C, 0)), w)
        nn_idx = ops.stage_knn(x, x, K)
        out, att = ops.stage_n2p_attn_fwd(qkv, nn_idx, 1, diff, want_att=True)
        ctx.save_for_backward(x, w, qkv, nn_idx)
        ctx.cfg = (diff, wq.shape[0], wk.shape[0])
        ctx.mark_non_differentiable(att, nn_idx)
        return out, att, nn_idx

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g, *_):
        x, w, qkv, nn_idx = ctx.saved_tensors
        diff, a, b = ctx.cfg
        dqkv = ops.stage_n2p_attn_bwd(qkv, nn_idx, g, 1, diff)
        need_dx = ctx.needs_input_grad[0]
        need_dw = any(ctx.needs_input_grad[1:4])
        dx, dw, _ = ops.stage_proj_bwd(dqkv, x, x.new_zeros((x.shape[1], 0)), w, need_dx, need_dw)
        if not need_dw:
            return dx, None, None, None, None, None
        C = x.shape[1]
        return (dx, dw[:a].reshape(a, C, 1, 1), dw[a:a + b].reshape(b, C, 1, 1), dw[a + b:].reshape(-1, C, 1, 1),
                None, None)


class _N2PAttention(torch.autograd.Function):
    """qkv (B,N,3C) point-major rows [Q|K|V], neighbour lists -> one-head attention of every point over its neighbours'
    K / V rows ("neighbor" grouping): (B,C,N) [differentiable w.r.t. qkv] and the (B,N,K) probabilities."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, qkv, nn_idx):
        qkv = qkv.contiguous()
        out, att = ops.stage_n2p_attn_fwd(qkv, nn_idx, 1, False, want_att=True)
        ctx.save_for_backward(qkv, nn_idx)
        ctx.mark_non_differentiable(att)
        return out, att

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g, _):
        qkv, nn_idx = ctx.saved_tensors
        return ops.stage_n2p_attn_bwd(qkv, nn_idx, g, 1, False), None


class DownSampleLocal(nn.Module):
    """Drop-in for the reference's local-attention sampler (models/downsample.py:818-1229): 1 x K
    attention of every point over its K = 32 nearest neighbours in feature space, a per-point score
    from that map (`local_std` or the sparse_* statistics), the top-M points kept and the N-M points
    with the smallest map std returned as the dropped set.

    On the kernels of Neighbor2PointAttention run with one head of 128 channels: one projection per
    point (linearity of the 1x1 convs), the gather-attention kernel (which also emits the
    probabilities), its atomics-free backward.  Because both outputs gather rows of the same
    per-point attention result, x_ds / x_dropped are column gathers of one (B,C,N) tensor.
    Outputs: ((x_ds (B,C,M), idx (B,1,M)), (x_dropped (B,C,N-M), idx_dropped (B,1,N-M)))."""

    def __init__(self, config_ds, layer):
        super().__init__()
        self.M = config_ds.M[layer]
        self.K = 32
        self.asm = config_ds.asm[layer]
        self.res = config_ds.res.enable[layer]
        self.ff = config_ds.res.ff[layer]
        self.num_heads = config_ds.num_heads[layer]
        self.idx_mode = config_ds.idx_mode[layer]
        q_in, q_out = config_ds.q_in[layer], config_ds.q_out[layer]
        k_in, k_out = config_ds.k_in[layer], config_ds.k_out[layer]
        v_in, v_out = config_ds.v_in[layer], config_ds.v_out[layer]
        self.group_type = "diff" if self.asm == "dot" else "neighbor"
        self.q_depth = int(q_out / self.num_heads)
        self.k_depth = int(k_out / self.num_heads)
        self.v_depth = int(v_out / self.num_heads)
        self.q_conv = nn.Conv2d(q_in, q_out, 1, bias=False)
        self.k_conv = nn.Conv2d(k_in, k_out, 1, bias=False)
        self.v_conv = nn.Conv2d(v_in, v_out, 1, bias=False)
        self.softmax = nn.Softmax(dim=-1)
        if self.res:
            self.bn1 = nn.BatchNorm1d(v_out)
            if self.ff:
                self.ffn = nn.Sequential(nn.Conv1d(128, 512, 1, bias=False), nn.LeakyReLU(negative_slope=0.2),
                                         nn.Conv1d(512, 128, 1, bias=False))
                self.bn2 = nn.BatchNorm1d(v_out)
        self.num_bins = config_ds.bin.num_bins[layer]
        self.scaling_factor = config_ds.bin.scaling_factor[layer]
        self.bin_sample_mode = config_ds.bin.sample_mode[layer]
        self.bin_norm_mode = config_ds.bin.norm_mode[layer]
        self.boltzmann_enable = config_ds.boltzmann.enable[layer]
        self.boltzmann_T = config_ds.boltzmann.boltzmann_T[layer]
        self.boltzmann_norm_mode = config_ds.boltzmann.norm_mode[layer]
        if self.asm not in ("dot", "dot-neighbor", "dot-sub", "l2", "l2+"):
            raise ValueError("Please check the setting of asm!")
        # the gather-attention kernels are built for 128 channels (every shipped config); any other width the reference
        # constructs (models/downsample.py:834-878) runs the same expression in torch on the device
        self._hip_attention = q_in == q_out == k_in == k_out == v_in == v_out == 128
        if self.idx_mode not in ("local_std", "sparse_row_std", "sparse_col_sum", "sparse_col_avg", "sparse_col_sqr"):
            raise ValueError("Please check the setting of idx mode!")

    def forward(self, x, x_xyz=None, forced_idx=None):
        """forced_idx = (idx (B,1,M), idx_dropped (B,1,N-M)): parity-test hook -- gather these columns instead of the
        selected ones (score and attention map are still computed and published)."""
        if self.num_heads != 1:
            # The reference constructs with any head count and then fails in its first forward, every idx_mode: its
            # get_sparse_attention_map views the (B, N, K) neighbour indices as (B, H, N, K) (models/downsample.py:1041-1044;
            # verified on the unmodified reference: RuntimeError).  A drop-in fails the same way, with the same message.
            B, _, N = x.shape
            raise RuntimeError(f"shape '[{B}, {self.num_heads}, {N}, {self.K}]' is invalid for input of size "
                               f"{B * N * self.K}")
        if not x.is_cuda:
            raise ops._lib.SambleError("samble_amd.DownSampleLocal runs on the GPU only (no CPU fallback)")
        B, C, N = x.shape
        _check_sizes("DownSampleLocal", N, self.M, self.K, strict=True)
        _check_forced("DownSampleLocal", forced_idx, N)
        if not self._hip_attention:
            x_all, att, nn_idx = self._attention_generic(x)
            C = x_all.shape[1]
        elif self.asm in ("l2", "l2+"):
            x_all, att, nn_idx = self._l2_attention(x)
        else:
            # attention_scoring (models/downsample.py:977-1000).  dot-sub: q (q^T - k_j) = |q|^2 - <q, k_j>: the first
            # term cancels in the softmax over the K neighbours -> the dot kernels on -K
            wk = -self.k_conv.weight if self.asm == "dot-sub" else self.k_conv.weight
            x_all, att, nn_idx = _LocalCore.apply(x, self.q_conv.weight, wk, self.v_conv.weight, self.K,
                                                  self.group_type == "diff")
        self.neighbors_idx = nn_idx.long()
        self.attention_map = att.view(B, 1, N, 1, self.K)
        std = torch.std(att, dim=-1, unbiased=False)                      # (B,N)
        if self.idx_mode == "local_std":
            score = std
        elif self.idx_mode == "sparse_row_std":
            score = torch.std(att, dim=-1)                                   # the K scattered entries of the row
        else:
            flat = self.neighbors_idx.reshape(B, N * self.K)
            colsum = torch.zeros((B, N), dtype=torch.float32, device=x.device).scatter_add_(1, flat, att.reshape(B, -1))
            num = torch.zeros((B, N), dtype=torch.float32, device=x.device).scatter_add_(
                1, flat, torch.ones_like(flat, dtype=torch.float32)) + 1e-8
            score = {"sparse_col_sum": colsum, "sparse_col_avg": colsum / num,
                     "sparse_col_sqr": colsum / num / num}[self.idx_mode]
        self.attention_point_score = score.unsqueeze(1)
        idx = ops.stage_topk_indices(score, self.M, largest=True)
        if self.boltzmann_enable:
            idx = self.boltzmann_idx_selection()[:, 0]
        idx_dropped = ops.stage_topk_indices(std, N - self.M, largest=False)
        if forced_idx is not None:
            idx = forced_idx[0].reshape(B, self.M).to(device=x.device, dtype=torch.int64)
            idx_dropped = forced_idx[1].reshape(B, N - self.M).to(device=x.device, dtype=torch.int64)
        self.idx = idx.unsqueeze(1)
        x_ds = torch.gather(x_all, 2, self.idx.expand(-1, C, -1))
        x_dropped = torch.gather(x_all, 2, idx_dropped.unsqueeze(1).expand(-1, C, -1))
        if self.res == True:  # noqa: E712
            x_ds = self.res_block(x, x_ds)
        return (x_ds, self.idx), (x_dropped, idx_dropped.unsqueeze(1))

    def _attention_generic(self, x):
        """models/downsample.py:885-1000 for the widths the gather-attention kernels are not built for: neighbours by the HIP
        kNN, then the grouping, the three 1x1 Conv2d over the (B,C,N,K) tensor, the 1 x K softmax and A V as torch
        expressions on the device (torch's autograd carries their gradients).  -> (B,Cv,N), (B,N,K), (B,N,K) int32"""
        B, C, N = x.shape
        K = self.K
        nn_idx = ops.stage_knn(x.detach(), x.detach(), K)
        nb = torch.gather(x, 2, nn_idx.long().reshape(B, 1, N * K).expand(-1, C, -1)).view(B, C, N, K)
        if self.group_type == "diff":
            nb = nb - x.unsqueeze(-1)
        q = self.q_conv(x.unsqueeze(-1))                                  # (B,D,N,1)
        k, v = self.k_conv(nb), self.v_conv(nb)                           # (B,D,N,K), (B,Dv,N,K)
        if self.asm in ("dot", "dot-neighbor"):
            energy = (q * k).sum(dim=1)                                   # (B,N,K)
        elif self.asm == "dot-sub":
            energy = (q * (q - k)).sum(dim=1)
        else:  # l2 / l2+: the K x K matrix (q - k_a)(q - k_b) averaged over a = <q - kbar, q - k_b>
            e = ((q - k.mean(dim=-1, keepdim=True)) * (q - k)).sum(dim=1)
            energy = -e if self.asm == "l2" else e
        att = torch.softmax(energy / math.sqrt(self.q_depth), dim=-1)
        out = (att.unsqueeze(1) * v).sum(dim=-1)                          # (B,Dv,N)
        return out, att, nn_idx

    def _l2_attention(self, x):
        """asm l2 / l2+ (models/downsample.py:984-996): the reference forms the K x K matrix (q - k_a)(q - k_b) and
        averages it over a: energy_b = -/+ <q - kbar, q - k_b>, kbar = mean of the K neighbour keys.  The part that does
        not depend on b cancels in the softmax over b, which leaves a dot-product attention of the query q - kbar
        (negated for l2+) over the neighbour keys: the N2P kernels, with kbar a gather-mean of the projected keys."""
        B, C, N = x.shape
        no_tokens = self.q_conv.weight.new_zeros((1, C, 0))
        qkv = _Projection.apply(x, no_tokens, self.q_conv.weight.view(C, C, 1), self.k_conv.weight.view(C, C, 1),
                                self.v_conv.weight.view(C, C, 1))
        nn_idx = ops.stage_knn(x.detach(), x.detach(), self.K)
        q, k, v = qkv[:, :, 0:C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
        # kbar = mean of the K neighbour keys, as an embedding-bag lookup over the (B*N, C) table of projected keys: no
        # (B, N, K, C) intermediate (1.07 GB at B=32, N=2048) in the forward, a segment reduction in the backward
        flat = (nn_idx.long() + torch.arange(B, device=x.device).view(B, 1, 1) * N).reshape(B * N, self.K)
        kbar = F.embedding_bag(flat, k.reshape(B * N, C), mode="mean").view(B, N, C)
        qe = (q - kbar) if self.asm == "l2" else (kbar - q)
        out, att = _N2PAttention.apply(torch.cat((qe, k, v), dim=-1), nn_idx)
        return out, att, nn_idx

    def boltzmann_idx_selection(self):
        """reference models/downsample.py:1205-1229: softmax(norm_range(score) / T) -> multinomial without replacement."""
        s = self.attention_point_score
        mode = self.boltzmann_norm_mode
        if mode == "minmax":
            lo = torch.min(s, dim=-1, keepdim=True)[0]
            sn = (s - lo) / (torch.max(s, dim=-1, keepdim=True)[0] - lo + 1e-8)
        elif mode == "z-score":
            sn = (s - torch.mean(s, dim=-1, keepdim=True)) / torch.std(s, dim=-1, unbiased=False, keepdim=True)
        elif mode == "sigmoid":
            sn = torch.sigmoid(s)
        elif mode == "tanh":
            sn = (torch.tanh(s) + 1.0) / 2
        else:
            raise ValueError(f"norm_range mode should be minmax, sigmoid or tanh, but got {mode}")
        p = F.softmax(sn / self.boltzmann_T, dim=-1)
        B, H, N = p.shape
        return torch.multinomial(p.reshape(B * H, N), self.M, replacement=False).view(B, H, self.M)

    def res_block(self, x, x_ds):
        x_tmp = torch.gather(x, dim=-1, index=self.idx)
        x_res = self.bn1(x_ds + x_tmp)
        if self.ff == True:  # noqa: E712
            x_tmp = _res_ffn(self.ffn, x_res)
            x_res = self.bn2(x_ds + x_tmp)
        return x_res

"""Drop-in `UpSampleInterpolation` (reference models/upsample.py:136-213), the decoder layer of the shipped
segmentation preset (`us_which: interpolation`, `distance_type: xyz`, K = 3).

Contract kept from the reference: the constructor's config reads, the parameter names (`conv.*`, `res_conv.*`)
and `forward(fine_features, ((coarse_features, indices, coarse_xyz), dropped), fine_xyz)`.  The work is laid out
this package's way: one cross-set K-nearest-neighbour search on the HIP kNN kernels (exact sum (a-b)^2 path for
xyz, reference-normalised distances), then an inverse-distance blend of the projected coarse features."""
from __future__ import annotations

import os

import torch
from torch import nn

from . import ops
from .attention import batch_norm


# how the layers' two 1x1 convolutions run on the GPU (plain fp32 GEMMs, 128 outputs: library work).  "bmm" (default) =
# rocBLAS through torch.bmm, weight gradient = batched GEMM + sum over the clouds; "conv" = the stock Conv1d (MIOpen: an
# implicit-GEMM weight gradient between two batched transposes, +0.18 ms per seg-block step); "lin" = csrc/linear.hip's
# channel-major entries, the concatenation never formed (deterministic split-operand sums; +0.12 ms over bmm: at 128
# outputs the weight-gradient kernel fills half the chip).  Same-box A/B in DESIGN 7.
POINTWISE = os.environ.get("SAMBLE_INTERP_CONV", "bmm")


class _PointwiseConv(nn.Conv1d):
    """nn.Conv1d(kernel 1, no bias): same parameter, same state_dict entry; forward(*xs) takes the channel-wise pieces
    of the input (128 channels each) so that their concatenation is not built on the linear.hip path."""

    def forward(self, *xs):
        from . import linear
        if POINTWISE == "lin" and xs[0].is_cuda and linear.pointwise_cm_supported(self.weight, *xs):
            return linear.pointwise_cm(self.weight, *xs)
        x = xs[0] if len(xs) == 1 else torch.cat(xs, dim=1)
        if POINTWISE == "bmm" and x.is_cuda and x.dim() == 3:
            # (weight.view: its backward is free; weight[:, :, 0]'s was a fill + copy.  W0 x0 + W1 x1 through bmm + baddbmm
            # instead of the concatenation: measured, no faster)
            return torch.bmm(self.weight.view(self.weight.shape[0], -1).unsqueeze(0).expand(x.shape[0], -1, -1), x)
        return super().forward(x)


def _unit_block(c_in: int, c_out: int) -> nn.Sequential:
    return nn.Sequential(_PointwiseConv(c_in, c_out, 1, bias=False), nn.BatchNorm1d(c_out), nn.LeakyReLU(negative_slope=0.2))


def inverse_distance_blend(values: torch.Tensor, dist: torch.Tensor) -> torch.Tensor:
    """values (B,C,N,K) neighbour features, dist (B,N,K) positive distances -> (B,C,N): weights 1/(d + 1e-8)
    normalised over the K neighbours (reference models/upsample.py:205-213)."""
    inv = (dist + 1e-8).reciprocal()
    share = inv / inv.sum(dim=-1, keepdim=True)
    return (values * share.unsqueeze(1)).sum(dim=-1)


def _normalised_pair_distances(query: torch.Tensor, keys: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """Differentiable distances between every query point and its K chosen keys, in the reference's kNN
    normalisation (utils/ops.py:23-29: both sets centred on the query mean, divided by the mean unbiased
    per-channel std of the queries).  query (B,C,N), keys (B,C,M), idx (B,N,K) -> (B,N,K)."""
    q = query.transpose(1, 2)
    k = keys.transpose(1, 2)
    centre = q.mean(dim=1, keepdim=True)
    q, k = q - centre, k - centre
    scale = q.std(dim=1, keepdim=True).mean(dim=2, keepdim=True)
    q, k = q / scale, k / scale
    return torch.linalg.vector_norm(q.unsqueeze(2) - ops.index_points(k, idx), dim=-1)


class _InterpBlend(torch.autograd.Function):
    """feat (B,C,M) coarse features, idx (B,N,K) int32 nearest coarse points, dist (B,N,K) -> (B,C,N): the reference's
    inverse-distance blend (models/upsample.py:205-213) on two HIP gather kernels (csrc/interp.hip); the distances carry
    no gradient (xyz search)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, feat, idx, dist):
        feat, dist = feat.contiguous(), dist.contiguous()
        idx = idx.to(torch.int32).contiguous()
        B, C, M = feat.shape
        N, K = idx.shape[1], idx.shape[2]
        with torch.cuda.device(feat.device):
            out = torch.empty((B, C, N), dtype=torch.float32, device=feat.device)
            w = torch.empty((B, N, K), dtype=torch.float32, device=feat.device)
            ops._lib.call("samble_interp_blend_fwd_f32", feat.data_ptr(), B, C, M, idx.data_ptr(), dist.data_ptr(), N, K,
                          w.data_ptr(), out.data_ptr(), ops._stream())
        ctx.save_for_backward(idx, w)
        ctx.M = M
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        idx, w = ctx.saved_tensors
        g = g.float().contiguous()
        B, C, N = g.shape
        K, M = idx.shape[2], ctx.M
        order, offsets, _ = ops.inverse_neighbors(idx)      # edges grouped by coarse point, ascending edge id
        with torch.cuda.device(g.device):
            dfeat = torch.empty((B, C, M), dtype=torch.float32, device=g.device)
            nbytes = ops._lib.query("samble_interp_blend_bwd_workspace_bytes", B, C, N, M)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
            ops._lib.call("samble_interp_blend_bwd_f32", g.data_ptr(), B, C, N, w.data_ptr(), order.data_ptr(),
                          offsets.data_ptr(), K, M, dfeat.data_ptr(), ws.data_ptr(), nbytes, ops._stream())
        return dfeat, None, None


FUSED_BLEND = True  # False: the (B,C,N,K) neighbour tensor and torch expressions (A/B runs)


class UpSampleInterpolation(nn.Module):
    def __init__(self, config_upsample, layer):
        super().__init__()
        q_in = config_upsample.q_in[layer]
        v_out = config_upsample.v_out[layer]
        self.distance_type = config_upsample.interpolation.distance_type[layer]
        self.K = config_upsample.interpolation.K[layer]
        self.conv = _unit_block(q_in, v_out)
        self.res_conv = _unit_block(2 * v_out, v_out)

    def forward(self, pcd_up, pcd_down, pcd_up_xyz):
        (coarse, _, coarse_xyz), _ = pcd_down
        filled = self.interpolate(pcd_up, coarse, pcd_up_xyz, coarse_xyz, self.distance_type, self.K)
        conv, norm, act = self.res_conv
        return batch_norm(norm, conv(pcd_up, filled), act)

    def interpolate(self, pcd_up, points_select, pcd_up_xyz, points_select_xyz, distance_type="feature", K=3):
        """Features of the coarse set spread onto the fine set (same signature as the reference's method)."""
        if distance_type not in ("xyz", "feature"):
            raise ValueError(f"upsample interpolation distance type can only be feature or xyz! Got: {distance_type}")
        conv, norm, act = self.conv
        projected = batch_norm(norm, conv(points_select), act)
        if distance_type == "xyz":
            if FUSED_BLEND and projected.is_cuda and K <= 8 and points_select_xyz.shape[2] <= pcd_up_xyz.shape[2]:
                idx, dist = ops.stage_knn(pcd_up_xyz, points_select_xyz, K, want_dist=True)
                return _InterpBlend.apply(projected, idx, dist)
            picked, _, dist = ops.select_neighbors_interpolate(pcd_up_xyz, points_select_xyz, projected, K=K)
            return inverse_distance_blend(picked, dist)
        # feature space: the reference back-propagates through its cdist; here the SEARCH runs on the HIP kNN
        # kernels (indices carry no gradient) and only the K chosen distances per point are recomputed in torch
        _, idx, _ = ops.select_neighbors_interpolate(pcd_up.detach(), points_select.detach(), projected.detach(), K=K)
        picked = ops.index_points(projected.transpose(1, 2), idx).permute(0, 3, 1, 2)
        return inverse_distance_blend(picked, _normalised_pair_distances(pcd_up, points_select, idx))


def upsample_config(preset: str = "seg"):
    from .config import to_attr
    return to_attr(dict(us_which="interpolation", interpolation=dict(distance_type=["xyz", "xyz"], K=[3, 3]),
                        q_in=[128, 128], q_out=[128, 128], k_in=[128, 128], k_out=[128, 128], v_in=[128, 128],
                        v_out=[128, 128], num_heads=[4, 4]))

"""Drop-in `EdgeConv` (reference models/embedding.py:7-39).

The neighbour build (`ops.group`'s kNN) runs on the HIP kNN kernels (exact (a-b)^2 path for xyz, fused
MFMA Gram + top-K for 64-d features).  For the shipped shape (K = 32 neighbours, 64 hidden / output
channels) the body -- conv1 + BN + LReLU + conv2 + BN + LReLU + max over K -- runs fused on HIP
(csrc/edgeconv.hip) without any (B, C, N, K) tensor in the forward pass:

* the 1x1 conv1 over [x_i ; x_j - x_i] is `a_i + b_j` with two per-point projections (plain torch
  matmuls, so autograd delivers dx and dW1 from da, db);
* BatchNorm-1's batch statistics over the B.N.K edges are closed forms of per-point sums;
* conv2 is one fp32-MFMA sweep over the edges (wave = point = 32 edges);
* LReLU o BN2 is monotone per channel, so max_k is taken on the raw conv2 output (max or min per the
  sign of gamma2) and the activation applied afterwards.

Backward recomputes the edge tensors in a second sweep (gradient of the pre-activation per edge +
dW2 partials); the BatchNorm corrections, d gamma / d beta and the per-point gradients da, db are closed
forms.  Other shapes (K != 32, other widths) use the stock torch composition.
"""
from __future__ import annotations

import torch
from torch import nn

from . import linear, ops
from . import _lib

FUSED_PROJECTIONS = True  # False: torch.matmul for the two per-point projections of conv1 (A/B runs)


_sync_group = ops.sync_group


def _all_sum(t: torch.Tensor, group) -> torch.Tensor:
    if group is not None:
        torch.distributed.all_reduce(t, group=group)
    return t


def _edge_weights(w1: torch.Tensor, group_type: str):
    """conv1 weight (Cout, Cin_total, 1, 1) -> (Wa, Wb) with conv1(group(x))_ij = Wa x_i + Wb x_j."""
    w = w1[:, :, 0, 0]
    if group_type == "neighbor":
        return torch.zeros_like(w), w
    if group_type == "diff":
        return -w, w
    c = w.shape[1] // 2
    if group_type == "center_neighbor":
        return w[:, :c], w[:, c:]
    if group_type == "center_diff":
        return w[:, :c] - w[:, c:], w[:, c:]
    raise ValueError(f"group_type should be neighbor, diff, center_neighbor or center_diff, but got {group_type}")


class _EdgeWeights(torch.autograd.Function):
    """conv1 weight (Cout, Cin_total, 1, 1) -> (2 Cout, C) = [Wa ; Wb] of _edge_weights, with its gradient in closed form: two
    small launches each way where autograd's slice / cat / neg nodes took a dozen (each one a launch of its own)."""

    @staticmethod
    def forward(ctx, w1, group_type):
        wa, wb = _edge_weights(w1.detach(), group_type)
        ctx.group_type = group_type
        return torch.cat((wa, wb), dim=0)

    @staticmethod
    def backward(ctx, dwab):
        half = dwab.shape[0] // 2
        da, db = dwab[:half], dwab[half:]
        gt = ctx.group_type
        if gt == "neighbor":
            dw = db
        elif gt == "diff":
            dw = db - da
        elif gt == "center_neighbor":
            dw = torch.cat((da, db), dim=1)
        else:  # center_diff: Wa = Wc - Wd, Wb = Wd
            dw = torch.cat((da, db - da), dim=1)
        return dw.reshape(dw.shape[0], dw.shape[1], 1, 1), None


class _EdgeMLP(torch.autograd.Function):
    """a, b (B,N,64) per-point projections, nn (B,N,32) -> (B,64,N)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, a, b, nn_idx, g1, b1, w2, g2, b2, bn1, bn2, training):
        B, N, C = a.shape
        K = nn_idx.shape[2]
        E = B * N * K
        dev = a.device
        # SyncBatchNorm: the edge statistics (and, backward, the gradient sums) are pooled over the ranks
        grp1 = _sync_group(bn1) if training else None
        grp2 = _sync_group(bn2) if training else None
        E1 = E2 = float(E)
        a = a.contiguous()
        b = b.contiguous()
        w2m = w2[:, :, 0, 0].contiguous()
        with torch.cuda.device(dev):
            S = torch.empty_like(b)
            Q = torch.empty_like(b)
            _lib.call("samble_edge_gather_sums_f32", b.data_ptr(), nn_idx.data_ptr(), B, N, K, C, S.data_ptr(),
                      Q.data_ptr(), ops._stream())
            if training:
                ad, Sd = a.double(), S.double()
                sum_z = K * ad.sum((0, 1)) + Sd.sum((0, 1))
                sum_z2 = (K * ad * ad + 2 * ad * Sd + Q.double()).sum((0, 1))
                if grp1 is not None:
                    pooled = _all_sum(torch.cat((sum_z, sum_z2, sum_z.new_tensor([float(E)]))), grp1)
                    sum_z, sum_z2, E1 = pooled[:C], pooled[C:2 * C], pooled[2 * C]  # 0-d tensor: no host sync
                mu1 = sum_z / E1
                var1 = (sum_z2 / E1 - mu1 * mu1).clamp_min(0)
            else:
                mu1, var1 = bn1.running_mean.double(), bn1.running_var.double()
            sig1 = torch.sqrt(var1 + bn1.eps)
            sc1 = (g1.double() / sig1)
            ap = (a * sc1.float() + (b1.double() - mu1 * sc1).float()).contiguous()
            bp = (b * sc1.float()).contiguous()
            nparts = _lib.query("samble_edge_partial_count")
            ymax = torch.empty_like(a)
            ymin = torch.empty_like(a)
            kmax = torch.empty((B, N, C), dtype=torch.uint8, device=dev)
            kmin = torch.empty((B, N, C), dtype=torch.uint8, device=dev)
            part = torch.empty((nparts, 2, C), dtype=torch.float64, device=dev)
            _lib.call("samble_edge_mlp_fwd_f32", ap.data_ptr(), bp.data_ptr(), nn_idx.data_ptr(), w2m.data_ptr(), B, N, K,
                      C, ymax.data_ptr(), ymin.data_ptr(), kmax.data_ptr(), kmin.data_ptr(), part.data_ptr(),
                      ops._stream())
            if training:
                tot = part.sum(0)
                if grp2 is not None:
                    pooled = _all_sum(torch.cat((tot.reshape(-1), tot.new_tensor([float(E)]))), grp2)
                    tot, E2 = pooled[:2 * C].view(2, C), pooled[2 * C]
                mu2 = tot[0] / E2
                var2 = (tot[1] / E2 - mu2 * mu2).clamp_min(0)
            else:
                mu2, var2 = bn2.running_mean.double(), bn2.running_var.double()
            sig2 = torch.sqrt(var2 + bn2.eps)
            sc2 = g2.double() / sig2
            ext = torch.where(g2 >= 0, ymax, ymin)
            kext = torch.where(g2 >= 0, kmax, kmin).contiguous()
            v = ((ext - mu2.float()) * sc2.float() + b2).contiguous()
            out = torch.maximum(v, 0.2 * v)
            if training:
                with torch.no_grad():
                    for bn, mu, var, En in ((bn1, mu1, var1, E1), (bn2, mu2, var2, E2)):
                        if bn.track_running_stats and bn.running_mean is not None:
                            bn.num_batches_tracked += 1
                            m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                            bn.running_mean.mul_(1 - m).add_(m * mu.to(bn.running_mean.dtype))
                            bn.running_var.mul_(1 - m).add_(m * (var * En / (En - 1)).to(bn.running_var.dtype))
        ctx.save_for_backward(a, b, nn_idx, S, ap, bp, w2m, ext, v, g1, g2, kext)
        ctx.stats = (mu1, sig1, sc1, mu2, sig2, sc2)
        ctx.pool = (grp1, E1, grp2, E2)
        ctx.training = training
        return out.permute(0, 2, 1)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        a, b, nn_idx, S, ap, bp, w2m, ext, v, g1, g2, kext = ctx.saved_tensors
        mu1, sig1, sc1, mu2, sig2, sc2 = ctx.stats
        grp1, E1, grp2, E2 = ctx.pool
        B, N, C = a.shape
        K = nn_idx.shape[2]
        E = B * N * K
        dev = a.device
        gt = g.permute(0, 2, 1)
        dv = (gt * torch.where(v > 0, 1.0, 0.2)).contiguous()
        yhat = (ext.double() - mu2) / sig2
        sum_dv = dv.double().sum((0, 1))
        sum_dvy = (dv.double() * yhat).sum((0, 1))
        dbeta2, dgamma2 = sum_dv, sum_dvy  # local sums: DDP averages parameter gradients over the ranks
        if ctx.training:
            if grp2 is not None:
                pooled = _all_sum(torch.cat((sum_dv, sum_dvy)), grp2)
                m1, m2 = pooled[:C] / E2, pooled[C:] / E2
            else:
                m1, m2 = sum_dv / E, sum_dvy / E
        else:
            m1 = m2 = torch.zeros_like(sum_dv)
        c1 = -sc2 * m2 / sig2
        c0 = -sc2 * m1 - c1 * mu2
        c0c1 = torch.stack((c0, c1)).float().contiguous()
        sdv = (dv * sc2.float()).contiguous()
        with torch.cuda.device(dev):
            nparts = _lib.query("samble_edge_partial_count")
            du = torch.empty((B, N, K, C), dtype=torch.float32, device=dev)
            dwp = torch.empty((nparts, C, C), dtype=torch.float32, device=dev)
            _lib.call("samble_edge_mlp_bwd_f32", ap.data_ptr(), bp.data_ptr(), nn_idx.data_ptr(), w2m.data_ptr(),
                      kext.data_ptr(), sdv.data_ptr(), c0c1.data_ptr(), B, N, K, C, du.data_ptr(), None, dwp.data_ptr(),
                      ops._stream())
        dw2 = dwp.sum(0)
        dusum = du.sum(2)                                                # sum_k du_ik
        # reverse-neighbour sums in a fixed order (inverse lists from a stable sort), not index_add_'s atomics
        order, offsets, counts = ops.inverse_neighbors(nn_idx)
        D = ops.stage_segment_sum_rows(du.view(-1, C), order, offsets, K, per_edge=True).view(B, N, C)
        sum_du = dusum.double().sum((0, 1))
        sum_duz = ((a.double() * dusum).sum((0, 1)) + (b.double() * D).sum((0, 1)) - mu1 * sum_du) / sig1
        dbeta1, dgamma1 = sum_du, sum_duz
        if ctx.training:
            if grp1 is not None:
                pooled = _all_sum(torch.cat((sum_du, sum_duz)), grp1)
                m1p, m2p = (pooled[:C] / E1).float(), (pooled[C:] / E1).float()
            else:
                m1p, m2p = (sum_du / E).float(), (sum_duz / E).float()
            indeg = counts.view(B, N, 1).float()
            R = ops.stage_segment_sum_rows(a.view(-1, C), order, offsets, K, per_edge=False).view(B, N, C)
            mu1f, sig1f = mu1.float(), sig1.float()
            Zs = (K * a + S - K * mu1f) / sig1f
            Zr = (R + indeg * (b - mu1f)) / sig1f
            da = sc1.float() * (dusum - K * m1p - m2p * Zs)
            db = sc1.float() * (D - indeg * m1p - m2p * Zr)
        else:
            da = sc1.float() * dusum
            db = sc1.float() * D
        return (da, db, None, dgamma1.to(g1.dtype), dbeta1.to(g1.dtype), dw2.view(C, C, 1, 1), dgamma2.to(g2.dtype),
                dbeta2.to(g2.dtype), None, None, None)


class _EdgeMLPFused(torch.autograd.Function):
    """_EdgeMLP with its closed forms on HIP (csrc/edge_glue.hip): BatchNorm batch statistics, their backward
    corrections, the activation and the per-point gradients as a dozen launches instead of ~250 torch ones.  Training
    mode (evaluation takes _EdgeMLP).  Under nn.SyncBatchNorm with more than one rank each glue entry runs in two halves
    around an all-reduce of its 2 x 64 + 1 float64 totals (include/samble.h SAMBLE_EDGE_SUMS / _APPLY): the same kernels
    on one rank and on eight.  ab (B,N,128) = the two per-point projections
    [a | b] as the ONE 1x1 convolution that forms them wrote them (read where they are, row stride 128; the gradient
    leaves as one (B,N,128) tensor the same way: no slice copies, no zero-filled halves), nn (B,N,32) -> (B,64,N)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, ab, nn_idx, g1, b1, w2, g2, b2, bn1, bn2):
        B, N, C2 = ab.shape
        C = C2 // 2
        K = nn_idx.shape[2]
        dev = ab.device
        ab = ab.contiguous()
        a, b = ab[..., :C], ab[..., C:]          # views: a.data_ptr() = ab's, b's 4 C bytes behind it, row stride 2 C
        g1, b1, g2, b2 = (t.detach().float().contiguous() for t in (g1, b1, g2, b2))
        w2m = w2.detach()[:, :, 0, 0].float().contiguous()
        run = lambda bn: (bn.running_mean, bn.running_var) if (bn.track_running_stats and bn.running_mean is not None) \
            else (None, None)
        # nn.BatchNorm2d.num_batches_tracked (int64 on the device): the statistics kernels count the batch themselves
        count = lambda bn: bn.num_batches_tracked if (bn.track_running_stats and bn.num_batches_tracked is not None
                                                       and bn.num_batches_tracked.is_cuda) else None
        grp1, grp2 = _sync_group(bn1), _sync_group(bn2)

        def glue(name, group, *args):
            """one glue entry: whole on one rank; statistics -> all-reduce of the pooled totals -> the rest over a group"""
            if group is None:
                _lib.call(name, *args, 0, None, ops._stream())
                return
            pooled = torch.empty(_lib.query("samble_edge_glue_pooled_bytes") // 8, dtype=torch.float64, device=dev)
            _lib.call(name, *args, 1, pooled.data_ptr(), ops._stream())
            torch.distributed.all_reduce(pooled, group=group)
            _lib.call(name, *args, 2, pooled.data_ptr(), ops._stream())

        ctx.glue = glue
        ctx.groups = (grp1, grp2)
        with torch.cuda.device(dev):
            f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
            S, Q, ap, bp = f32(B, N, C), f32(B, N, C), f32(B, N, C), f32(B, N, C)
            cst = torch.empty(_lib.query("samble_edge_glue_constants_bytes") // 4, dtype=torch.float32, device=dev)
            st = torch.empty(_lib.query("samble_edge_glue_statistics_bytes") // 8, dtype=torch.float64, device=dev)
            part = torch.empty(_lib.query("samble_edge_glue_partials_bytes") // 8, dtype=torch.float64, device=dev)
            rm1, rv1 = run(bn1)
            glue("samble_edge_bn1_f32", grp1, a.data_ptr(), b.data_ptr(), C2, nn_idx.data_ptr(), B, N, K, C, g1.data_ptr(),
                 b1.data_ptr(), float(bn1.eps), ops._p(rm1), ops._p(rv1), float(bn1.momentum), ops._p(count(bn1)), S.data_ptr(),
                 Q.data_ptr(), ap.data_ptr(), bp.data_ptr(), cst.data_ptr(), st.data_ptr(), part.data_ptr())
            nparts = _lib.query("samble_edge_partial_count")
            ymax, ymin = f32(B, N, C), f32(B, N, C)
            kmax = torch.empty((B, N, C), dtype=torch.uint8, device=dev)
            kmin = torch.empty((B, N, C), dtype=torch.uint8, device=dev)
            mpart = torch.empty((nparts, 2, C), dtype=torch.float64, device=dev)
            _lib.call("samble_edge_mlp_fwd_f32", ap.data_ptr(), bp.data_ptr(), nn_idx.data_ptr(), w2m.data_ptr(), B, N, K,
                      C, ymax.data_ptr(), ymin.data_ptr(), kmax.data_ptr(), kmin.data_ptr(), mpart.data_ptr(),
                      ops._stream())
            ext = f32(B, N, C)
            kext = torch.empty((B, N, C), dtype=torch.uint8, device=dev)
            out = f32(B, C, N)
            rm2, rv2 = run(bn2)
            glue("samble_edge_bn2_out_f32", grp2, ymax.data_ptr(), ymin.data_ptr(), kmax.data_ptr(), kmin.data_ptr(),
                 mpart.data_ptr(), nparts, B, N, C, g2.data_ptr(), b2.data_ptr(), float(bn2.eps), ops._p(rm2),
                 ops._p(rv2), float(bn2.momentum), ops._p(count(bn2)), cst.data_ptr(), st.data_ptr(), ext.data_ptr(),
                 kext.data_ptr(), out.data_ptr())
            with torch.no_grad():
                for bn in (bn1, bn2):   # (a counter that does not live on this device: the launch the kernels save otherwise)
                    if bn.track_running_stats and bn.num_batches_tracked is not None and count(bn) is None:
                        bn.num_batches_tracked += 1
        ctx.save_for_backward(ab, nn_idx, S, ap, bp, w2m, ext, kext, cst, st, g2)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        ab, nn_idx, S, ap, bp, w2m, ext, kext, cst, st, g2 = ctx.saved_tensors
        B, N, C2 = ab.shape
        C = C2 // 2
        a, b = ab[..., :C], ab[..., C:]
        K = nn_idx.shape[2]
        dev = a.device
        g = g.float().contiguous()
        glue = ctx.glue
        grp1, grp2 = ctx.groups
        with torch.cuda.device(dev):
            f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
            cst = cst.clone()   # (the backward adds its correction terms: the saved block stays as the forward left it)
            part = torch.empty(_lib.query("samble_edge_glue_partials_bytes") // 8, dtype=torch.float64, device=dev)
            sdv, dg2, db2 = f32(B, N, C), f32(C), f32(C)
            glue("samble_edge_bwd_pre_f32", grp2, g.data_ptr(), ext.data_ptr(), B, N, C, g2.data_ptr(), cst.data_ptr(),
                 st.data_ptr(), sdv.data_ptr(), dg2.data_ptr(), db2.data_ptr(), part.data_ptr())
            nparts = _lib.query("samble_edge_partial_count")
            du = f32(B, N, K, C)
            dwp = f32(nparts, C, C)
            dusum = f32(B, N, C)
            _lib.call("samble_edge_mlp_bwd_f32", ap.data_ptr(), bp.data_ptr(), nn_idx.data_ptr(), w2m.data_ptr(),
                      kext.data_ptr(), sdv.data_ptr(), cst.data_ptr() + 256 * 4, B, N, K, C, du.data_ptr(),
                      dusum.data_ptr(), dwp.data_ptr(), ops._stream())
            # reverse-neighbour sums in a fixed order (inverse lists), not index_add_'s atomics
            order, offsets, counts = ops.inverse_neighbors(nn_idx)
            D, R = ops.stage_segment_sum_rows_pair(du.view(-1, C), ab.view(-1, C2)[:, :C], order, offsets, K)   # one pass
            dab = f32(B, N, C2)
            dg1, db1, dw2 = f32(C), f32(C), f32(C, C)
            glue("samble_edge_bwd_post_f32", grp1, a.data_ptr(), b.data_ptr(), C2, S.data_ptr(), R.data_ptr(), dusum.data_ptr(),
                 D.data_ptr(), counts.data_ptr(), B, N, K, C, cst.data_ptr(), st.data_ptr(), dwp.data_ptr(), nparts,
                 dab.data_ptr(), dab.data_ptr() + 4 * C, C2, dg1.data_ptr(), db1.data_ptr(), dw2.data_ptr(),
                 part.data_ptr())
        return dab, None, dg1, db1, dw2.view(C, C, 1, 1), dg2, db2, None, None


FUSED_GLUE = True  # False: the closed forms as torch expressions (_EdgeMLP; A/B runs)


class EdgeConv(nn.Module):
    def __init__(self, config_embedding, layer):
        super().__init__()
        self.K = config_embedding.K[layer]
        self.group_type = config_embedding.group_type[layer]
        self.normal_channel = config_embedding.normal_channel
        c1_in, c1_out = config_embedding.conv1_in[layer], config_embedding.conv1_out[layer]
        c2_in, c2_out = config_embedding.conv2_in[layer], config_embedding.conv2_out[layer]
        self.conv1 = nn.Sequential(nn.Conv2d(c1_in, c1_out, kernel_size=1, bias=False), nn.BatchNorm2d(c1_out),
                                   nn.LeakyReLU(negative_slope=0.2))
        self.conv2 = nn.Sequential(nn.Conv2d(c2_in, c2_out, kernel_size=1, bias=False), nn.BatchNorm2d(c2_out),
                                   nn.LeakyReLU(negative_slope=0.2))
        self.fused = True  # False: the stock torch composition (A/B checks)

    def _fusable(self, x):
        bn1, bn2 = self.conv1[1], self.conv2[1]
        return (self.fused and x.is_cuda and self.K == 32 and self.conv1[0].out_channels == 64
                and self.conv2[0].in_channels == 64 and self.conv2[0].out_channels == 64 and bn1.affine and bn2.affine
                and not (self.normal_channel and x.shape[1] == 6)
                and (self.training or (bn1.running_mean is not None and bn2.running_mean is not None)))

    def forward(self, x):
        if not self._fusable(x):
            x, _ = ops.group(x, self.K, self.group_type, self.normal_channel)
            x = self.conv1(x)
            x = self.conv2(x)
            return x.max(dim=-1, keepdim=False)[0]
        nn_idx = ops.stage_knn(x.detach(), x.detach(), self.K)
        wab = _EdgeWeights.apply(self.conv1[0].weight, self.group_type)   # (128, C): both per-point projections at once
        if FUSED_PROJECTIONS and linear.linear_supported(x, wab):
            ab = linear.linear_rows(x, wab)                           # HIP 1x1 convolution (csrc/linear.hip): (B,N,128)
        else:
            ab = torch.matmul(x.permute(0, 2, 1), wab.t())
        bn1, bn2 = self.conv1[1], self.conv2[1]
        use_batch_stats = self.training or not bn1.track_running_stats
        if FUSED_GLUE and use_batch_stats and bn1.momentum is not None and bn2.momentum is not None:
            return _EdgeMLPFused.apply(ab, nn_idx, bn1.weight, bn1.bias, self.conv2[0].weight, bn2.weight, bn2.bias,
                                       bn1, bn2)
        half = ab.shape[-1] // 2
        return _EdgeMLP.apply(ab[..., :half], ab[..., half:], nn_idx, bn1.weight, bn1.bias, self.conv2[0].weight, bn2.weight,
                              bn2.bias, bn1, bn2, use_batch_stats)


def embedding_config(preset: str = "cls"):
    """`config.feature_learning_block.embedding` of the shipped presets (two EdgeConv layers)."""
    from .config import to_attr
    return to_attr(dict(K=[32, 32], group_type=["center_diff", "center_diff"], normal_channel=False,
                        conv1_in=[6, 128], conv1_out=[64, 64], conv2_in=[64, 64], conv2_out=[64, 64]))

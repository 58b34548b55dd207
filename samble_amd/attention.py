"""Drop-in `Neighbor2PointAttention` (reference models/attention.py:130-250) on the MI355X kernels.

Same constructor (`Cls(config.attention, layer)`), `forward(x (B,C,N)) -> (B,C,N)` and state_dict
keys (`q_conv/k_conv/v_conv.weight` (C,C,1,1), `ff.0/ff.2.weight`, `bn1.*`, `bn2.*`) as the
reference.  The neighbour build is the fused Gram + top-K HIP kernel, the three 1x1 Conv2d collapse
(by linearity) into one per-point fp32-MFMA projection, and the K-neighbour softmax attention is
one HIP gather kernel: the (B,C,N,K) tensors of the reference are never built in forward.
The BatchNorms (training mode, nn.SyncBatchNorm included) and the FFN around it run on csrc/batchnorm.hip / csrc/linear.hip.

Backward of the attention part is two HIP kernels (per-point pass, then an ordered gather over
64-row target blocks: no atomics, run-to-run identical) followed by the HIP projection backward.
"""
from __future__ import annotations

import math
import os
import threading

import torch
from torch import nn

from . import linear, ops


def _attention_from_projection(qkv, nn_idx, heads: int, diff: bool):
    """Differentiable torch restatement used only for the backward pass: qkv (b,N,3C) -> (b,C,N)."""
    b, N, C3 = qkv.shape
    C = C3 // 3
    D = C // heads
    K = nn_idx.shape[2]
    q, kp, vp = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    gi = nn_idx.long().reshape(b, N * K, 1).expand(-1, -1, C)
    kg = torch.gather(kp, 1, gi).view(b, N, K, C)
    vg = torch.gather(vp, 1, gi).view(b, N, K, C)
    if diff:
        kg = kg - kp[:, :, None, :]
        vg = vg - vp[:, :, None, :]
    logits = (q.view(b, N, 1, heads, D) * kg.view(b, N, K, heads, D)).sum(-1) / math.sqrt(D)
    att = torch.softmax(logits, dim=2)
    out = (att.unsqueeze(-1) * vg.view(b, N, K, heads, D)).sum(2).reshape(b, N, C)
    return out.permute(0, 2, 1)


FUSED_FFN = True  # False: the stock Conv1d modules (A/B runs)


def _feed_forward(ff: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """`self.ff(x)` of the reference's attention layers (models/attention.py:187-192: Conv1d 128->512, LeakyReLU(0.2),
    Conv1d 512->128, no biases) on the HIP 1x1-convolution kernels (csrc/linear.hip) when the shape is theirs; any other
    configuration runs the stock modules."""
    w1, w2 = ff[0].weight, ff[2].weight
    if (FUSED_FFN and ff[0].bias is None and ff[2].bias is None and abs(ff[1].negative_slope - 0.2) < 1e-12
            and linear.ffn_supported(x, w1, w2)):
        return linear.ffn(x, w1, w2)
    return ff(x)


class _N2PCore(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, wq, wk, wv, K, heads, diff):
        C = x.shape[1]
        w = torch.cat((wq, wk, wv), dim=0).reshape(3 * C, C)
        no_tokens = x.new_zeros((C, 0))
        qkv = ops.stage_proj_fwd(x, no_tokens, w)
        nn_idx = ops.stage_knn(x, x, K)
        out = ops.stage_n2p_attn_fwd(qkv, nn_idx, heads, diff)
        ctx.save_for_backward(x, w, qkv, nn_idx)
        ctx.cfg = (heads, diff, wq.shape[0], wk.shape[0])
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x, w, qkv, nn_idx = ctx.saved_tensors
        heads, diff, a, b = ctx.cfg
        if nn_idx.shape[2] <= 32:
            dqkv = ops.stage_n2p_attn_bwd(qkv, nn_idx, g, heads, diff)  # HIP, deterministic
        else:  # K > 32: torch restatement, chunked over clouds
            dqkv = torch.empty_like(qkv)
            step = 4
            for s in range(0, qkv.shape[0], step):
                with torch.enable_grad():
                    part = qkv[s:s + step].detach().requires_grad_(True)
                    out = _attention_from_projection(part, nn_idx[s:s + step], heads, diff)
                dqkv[s:s + step] = torch.autograd.grad(out, part, g[s:s + step])[0]
        need_dx = ctx.needs_input_grad[0]
        need_dw = any(ctx.needs_input_grad[1:4])
        dx, dw, _ = ops.stage_proj_bwd(dqkv, x, x.new_zeros((x.shape[1], 0)), w, need_dx, need_dw)
        if not need_dw:
            return dx, None, None, None, None, None, None
        C = x.shape[1]
        return (dx, dw[:a].reshape(a, C, 1, 1), dw[a:a + b].reshape(b, C, 1, 1), dw[a + b:].reshape(-1, C, 1, 1),
                None, None, None)


# False: the layer as separate autograd nodes (attention core, adds, BatchNorm1d, FFN): the A/B reference
FUSED_LAYER = os.environ.get("SAMBLE_FUSED_LAYER", "1") != "0"


class _N2PLayer(torch.autograd.Function):
    """The whole Neighbor2PointAttention layer (reference models/attention.py:165-192) as ONE autograd node, training mode:
        s1 = x + attention(x);  y1 = bn1(s1);  s2 = y1 + ff(y1);  y2 = bn2(s2)
    The four elementwise passes of that expression and of its gradient (two residual adds forward, two gradient
    accumulations backward: 4 x 33 MB in and out at N = 2048, a launch each) ride on the epilogues of the kernels that
    produce the other summand -- the gather attention, the FFN's second product, the FFN's input gradient, the
    projection's input gradient (`residual` of include/samble.h) -- and the weights' `cat`, the autograd bookkeeping
    between eight nodes and their saved intermediates go with them.  Same kernels, same BatchNorm (csrc/batchnorm.hip, both
    directions; under nn.SyncBatchNorm its per-channel sums are all-reduced over the module's process group), same sums in
    the same order: outputs, all gradients and the running statistics are bit-identical to the node-by-node composition
    (tests/test_gpu_layers.py)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, wq, wk, wv, w1, w2, g1, b1, g2, b2, bn1, bn2, K, heads, diff):
        B, C, N = x.shape
        x = x.contiguous()
        H = w1.shape[0]
        w = torch.cat((wq, wk, wv), dim=0).reshape(3 * C, C)
        qkv = ops.stage_proj_fwd(x, x.new_zeros((C, 0)), w)
        nn_idx = ops.stage_knn(x, x, K)
        s1 = ops.stage_n2p_attn_fwd(qkv, nn_idx, heads, diff, residual=x)                 # x + attention(x)
        grp1, grp2 = _sync_group(bn1), _sync_group(bn2)     # nn.SyncBatchNorm: the statistics are pooled over the ranks
        y1, m1, v1, n1 = _bn_train(bn1, s1, g1, b1, grp1)
        w1_rm, w1_tr, w2t_rm, w2t_tr = linear.ffn_weight_images(w1.reshape(H, C), w2.reshape(C, H))
        if linear.chain_supported(y1, H):                                                  # both products in one sweep
            s2, hr, hbits = linear.stage_linear_chain(y1, w1_rm, w2t_tr, H, linear.LIN_LEAKY_BITS, residual=y1)
        else:
            hr, hbits = linear.stage_linear_fwd(y1, w1_rm, H, linear.LIN_LEAKY_BITS)      # leaky(W1 y1), (B,N,H) + its sign bits
            s2 = linear.stage_linear_dx(hr, w2t_tr, H, residual=y1)                       # y1 + W2 h
        y2, m2, v2, n2 = _bn_train(bn2, s2, g2, b2, grp2)
        ctx.save_for_backward(x, w, qkv, nn_idx, s1, m1, v1, y1, hr, s2, m2, v2, g1, g2, w1_tr, w2t_rm, hbits)
        ctx.cfg = (heads, diff, wq.shape[0], wk.shape[0], H)
        ctx.pool = (grp1, n1, grp2, n2)
        return y2

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy2):
        x, w, qkv, nn_idx, s1, m1, v1, y1, hr, s2, m2, v2, g1, g2, w1_tr, w2t_rm, hbits = ctx.saved_tensors
        heads, diff, a, b, H = ctx.cfg
        grp1, n1, grp2, n2 = ctx.pool
        C = x.shape[1]
        dy2 = dy2.float().contiguous()
        ds2, dg2, db2 = ops.stage_bn_train_bwd(s2, dy2, g2, m2, v2, n2, grp2)
        if linear.chain_supported(ds2, H):
            dy1, dh, _ = linear.stage_linear_chain(ds2, w2t_rm, w1_tr, H, linear.LIN_LEAKY_MASK_BITS, bits=hbits, residual=ds2)
        else:
            dh = linear.stage_linear_fwd(ds2, w2t_rm, H, linear.LIN_LEAKY_MASK_BITS, bits=hbits)   # (W2^T ds2) * leaky'(h)
        dw2 = linear.stage_linear_dw(hr, ds2, H, transposed=True).reshape(C, H, 1)
        dw1 = linear.stage_linear_dw(dh, y1, H).reshape(H, C, 1)
        if not linear.chain_supported(ds2, H):
            dy1 = linear.stage_linear_dx(dh, w1_tr, H, residual=ds2, out=ds2)              # ds2 + W1^T dh, in place
        ds1, dg1, db1 = ops.stage_bn_train_bwd(s1, dy1, g1, m1, v1, n1, grp1, out=dy1)   # (dy1 is this node's own buffer)
        dqkv = ops.stage_n2p_attn_bwd(qkv, nn_idx, ds1, heads, diff)
        dx, dw, _ = ops.stage_proj_bwd(dqkv, x, x.new_zeros((C, 0)), w, True, True, dx_residual=ds1)   # ds1 + W^T dqkv
        return (dx, dw[:a].reshape(a, C, 1, 1), dw[a:a + b].reshape(b, C, 1, 1), dw[a + b:].reshape(-1, C, 1, 1), dw1, dw2,
                dg1, db1, dg2, db2, None, None, None, None, None)


_counts = threading.local()   # .pending: this thread's open deferred_batch_counts map (None outside the context)


class deferred_batch_counts:
    """Inside this context the fused layers' `num_batches_tracked += 1` (one single-thread launch per BatchNorm and
    call: ten per step of the segmentation block) are collected and applied as ONE multi-tensor add on the way out.  The
    blocks wrap their forward in it; a layer called on its own counts at once, as before.  The map is per THREAD (two
    replicas under nn.DataParallel, an evaluation thread beside the training one: each sees its own), and a forward that
    raised leaves the counters where they were."""

    def __enter__(self):
        self.outer = getattr(_counts, "pending", None)
        _counts.pending = {}
        return self

    def __exit__(self, *exc):
        mine, _counts.pending = _counts.pending, self.outer
        if exc[0] is not None:
            return False
        once = [t for t, k in mine.values() if k == 1]
        if once:
            torch._foreach_add_(once, 1)
        for t, k in mine.values():
            if k > 1:       # (the same module called twice inside one forward)
                t.add_(k)
        return False


_sync_group = ops.sync_group


def _bn_train(bn, s: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, group=None, act_slope: float = 1.0):
    """nn.BatchNorm1d.forward in training mode on (B,C,N) -> (y, saved mean, saved invstd, pooled count | None), with the
    module's own bookkeeping (momentum, running statistics, num_batches_tracked).  csrc/batchnorm.hip; `group`: the ranks an
    nn.SyncBatchNorm pools its statistics over."""
    factor = 0.0
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        pending = getattr(_counts, "pending", None)
        if pending is not None and bn.momentum is not None:      # (the count is only bookkeeping then)
            t, k = pending.get(id(bn.num_batches_tracked), (bn.num_batches_tracked, 0))
            pending[id(bn.num_batches_tracked)] = (t, k + 1)
        else:
            bn.num_batches_tracked.add_(1)
        factor = (1.0 / float(bn.num_batches_tracked)) if bn.momentum is None else bn.momentum
    elif bn.momentum is not None:
        factor = bn.momentum
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    return ops.stage_bn_train(s, gamma, beta, rm, rv, factor, bn.eps, group, act_slope)


# csrc/batchnorm.hip for bn1 / bn2 in training mode, forward and backward; "0": the stock modules (A/B runs)
OWN_BATCHNORM = os.environ.get("SAMBLE_OWN_BATCHNORM", "1") != "0"


class _BNTrain(torch.autograd.Function):
    """nn.BatchNorm1d / nn.SyncBatchNorm in training mode as the fused layer runs it (`_bn_train` forward,
    `ops.stage_bn_train_bwd` backward): what the node-by-node form of the layer calls, so that both forms stay
    bit-identical."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, s, gamma, beta, bn, act_slope=1.0):
        s = s.contiguous()
        group = _sync_group(bn)
        y, m, v, n = _bn_train(bn, s, gamma, beta, group, act_slope)
        ctx.save_for_backward(s, gamma, m, v, beta)
        ctx.pool = (group, n, float(act_slope))
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        s, gamma, m, v, beta = ctx.saved_tensors
        group, n, slope = ctx.pool
        ds, dg, db = ops.stage_bn_train_bwd(s, dy.float().contiguous(), gamma, m, v, n, group, beta=beta, act_slope=slope)
        return ds, dg, db, None, None


def _plain_bn(bn) -> bool:
    """A BatchNorm the own kernels take: nn.BatchNorm1d or nn.SyncBatchNorm (what convert_sync_batchnorm makes of it), affine,
    float32 parameters and buffers, and either a momentum or no running statistics (the cumulative average needs the host's
    copy of the counter)."""
    if type(bn) not in (nn.BatchNorm1d, nn.SyncBatchNorm) or not bn.affine:
        return False
    if bn.track_running_stats and bn.momentum is None:
        return False
    tensors = [bn.weight, bn.bias] + ([bn.running_mean, bn.running_var] if bn.track_running_stats else [])
    return all(t is not None and t.dtype == torch.float32 and t.is_contiguous() for t in tensors)


FUSED_BN_ACT = os.environ.get("SAMBLE_FUSED_BN_ACT", "1") != "0"   # "0": the LeakyReLU behind a BatchNorm as a stock op (A/B)


def batch_norm(bn, s: torch.Tensor, act=None) -> torch.Tensor:
    """`bn(s)` [`act(bn(s))`] -- through `_BNTrain` where the fused layer would take the same route, the module itself
    otherwise (evaluation, a BatchNorm frozen with bn.eval() inside a training layer, other dtypes).  act: an
    nn.LeakyReLU behind the normalisation (models/upsample.py:142-150) rides in the BatchNorm kernels' passes: no
    elementwise launch of its own, forward or backward."""
    if OWN_BATCHNORM and bn.training and _plain_bn(bn) and s.is_cuda and s.dtype == torch.float32 and s.dim() == 3:
        if act is None:
            return _BNTrain.apply(s, bn.weight, bn.bias, bn)
        if FUSED_BN_ACT and type(act) is nn.LeakyReLU and not act.inplace and 0.0 < act.negative_slope < 1.0:
            return _BNTrain.apply(s, bn.weight, bn.bias, bn, float(act.negative_slope))
        return act(_BNTrain.apply(s, bn.weight, bn.bias, bn))
    y = bn(s)
    return y if act is None else act(y)


def _layer_fusable(mod, x) -> bool:
    bn1, bn2 = mod.bn1, mod.bn2
    # bn.training, not only mod.training: a BatchNorm frozen with bn.eval() inside a layer in train() normalises with its
    # running estimates and leaves them alone (the reference's behaviour, and `batch_norm`'s) -- the fused node would not
    return (FUSED_LAYER and FUSED_FFN and OWN_BATCHNORM and mod.training and bn1.training and bn2.training and x.is_cuda
            and x.dtype == torch.float32 and ops.MATRIX_MODE == "tri"
            and mod.hip_attention and mod.K <= 32 and mod.attention_mode == "scalar_dot"
            and not mod.group_type.startswith("center_")
            and _plain_bn(bn1) and _plain_bn(bn2) and mod.ff[0].bias is None and mod.ff[2].bias is None
            and abs(mod.ff[1].negative_slope - 0.2) < 1e-12 and linear.ffn_supported(x, mod.ff[0].weight, mod.ff[2].weight))


class Neighbor2PointAttention(nn.Module):
    def __init__(self, config_attention, layer):
        super().__init__()
        self.K = config_attention.K[layer]
        self.group_type = config_attention.group_type[layer]
        self.num_heads = config_attention.num_heads[layer]
        self.attention_mode = config_attention.attention_mode[layer]
        q_in, q_out = config_attention.q_in[layer], config_attention.q_out[layer]
        k_in, k_out = config_attention.k_in[layer], config_attention.k_out[layer]
        v_in, v_out = config_attention.v_in[layer], config_attention.v_out[layer]
        self.asm = config_attention.asm[layer]
        self.q_depth = int(q_out / self.num_heads)
        self.k_depth = int(k_out / self.num_heads)
        self.v_depth = int(v_out / self.num_heads)
        self.q_conv = nn.Conv2d(q_in, q_out, 1, bias=False)
        self.k_conv = nn.Conv2d(k_in, k_out, 1, bias=False)
        self.v_conv = nn.Conv2d(v_in, v_out, 1, bias=False)
        self.softmax = nn.Softmax(dim=-1)
        self.ff = nn.Sequential(
            nn.Conv1d(config_attention.ff_conv1_channels_in[layer], config_attention.ff_conv1_channels_out[layer], 1,
                      bias=False),
            nn.LeakyReLU(negative_slope=0.2),
            nn.Conv1d(config_attention.ff_conv2_channels_in[layer], config_attention.ff_conv2_channels_out[layer], 1,
                      bias=False),
        )
        self.bn1 = nn.BatchNorm1d(v_out)
        self.bn2 = nn.BatchNorm1d(v_out)
        if self.attention_mode not in ("scalar_dot", "vector_sub"):
            raise ValueError(f"attention_mode can only be scalar_dot or vector_sub, but got: {self.attention_mode}")
        if self.asm not in ("dot", "dot-sub"):
            raise ValueError("Please check the setting of asm in feature learning layer!")
        if self.group_type not in ("diff", "neighbor", "center_neighbor", "center_diff"):
            raise ValueError(
                f"group_type should be neighbor, diff, center_neighbor or center_diff, but got {self.group_type}")
        grouped = 2 * q_in if self.group_type.startswith("center_") else q_in   # channels of the grouped tensor
        if k_in != grouped or v_in != grouped:
            raise ValueError(f"group_type {self.group_type} gives the key / value convolutions {grouped} input channels")
        if q_out % self.num_heads or k_out % self.num_heads or v_out % self.num_heads or q_out != k_out:
            raise ValueError("q_out = k_out and v_out must be multiples of num_heads")
        # the gather-attention kernels (csrc/n2p.hip): 128 channels, 4, 2 or 1 head(s) (the shipped configs: 4).  Any other
        # width or head count (the reference's constructor takes them, models/attention.py:131-163; no shipped config
        # has one) runs the same expression in torch on the device, on neighbour lists from the HIP kNN
        self.hip_attention = q_in == q_out == k_out == v_out == 128 and self.num_heads in (1, 2, 4)

    def forward(self, x):
        if not x.is_cuda:
            raise ops._lib.SambleError("samble_amd.Neighbor2PointAttention runs on the GPU only (no CPU fallback)")
        if _layer_fusable(self, x):
            wk = -self.k_conv.weight if self.asm == "dot-sub" else self.k_conv.weight      # (see below)
            return _N2PLayer.apply(x, self.q_conv.weight, wk, self.v_conv.weight, self.ff[0].weight, self.ff[2].weight,
                                   self.bn1.weight, self.bn1.bias, self.bn2.weight, self.bn2.bias, self.bn1, self.bn2,
                                   self.K, self.num_heads, self.group_type == "diff")
        if self.attention_mode == "vector_sub":
            x_tmp = self._vector_sub(x)
        else:
            # What the gather-attention kernel computes is softmax_j(<q_i, K x_j [- K x_i]>) and the same mix of
            # V x_j [- V x_i].  The other scalar_dot variants of the reference reduce to it (models/attention.py:
            # 203-250): dot-sub: q (q^T - k_j) = |q|^2 - <q, k_j>, and what does not depend on j cancels in the softmax
            # over the neighbours -> the kernel on -K.  center_*: the 2C-channel grouped tensor [x_i ; g_ij] makes
            # k_ij = K1 x_i + K2 g_ij, v_ij = V1 x_i + V2 g_ij; K1 x_i cancels in the softmax, V1 x_i passes through it
            # (the weights sum to one) -> the kernel on (K2, V2) plus the per-point term V1 x_i.
            C = x.shape[1]
            wk, wv = self.k_conv.weight, self.v_conv.weight
            center = self.group_type.startswith("center_")
            wk2, wv2 = (wk[:, C:], wv[:, C:]) if center else (wk, wv)
            if self.asm == "dot-sub":
                wk2 = -wk2
            diff = self.group_type in ("diff", "center_diff")
            if self.hip_attention:
                x_tmp = _N2PCore.apply(x, self.q_conv.weight, wk2.contiguous(), wv2.contiguous(), self.K, self.num_heads,
                                       diff)
            else:
                Cq = self.q_conv.weight.shape[0]
                if self.v_conv.weight.shape[0] != Cq:
                    raise NotImplementedError("v_out != q_out outside the 128-channel kernels")
                w = torch.cat((self.q_conv.weight, wk2, wv2), dim=0).reshape(3 * Cq, C)
                qkv = torch.matmul(x.transpose(1, 2), w.t())                       # (B,N,3Cq) rows [Q|K|V]
                nn_idx = ops.stage_knn(x.detach(), x.detach(), self.K)
                step = max(1, (1 << 28) // (x.shape[2] * self.K * Cq * 4))         # <= 256 MB per gathered tensor
                x_tmp = torch.cat([_attention_from_projection(qkv[s:s + step], nn_idx[s:s + step], self.num_heads, diff)
                                   for s in range(0, x.shape[0], step)])
            if center:
                x_tmp = x_tmp + torch.nn.functional.conv1d(x, wv[:, :C, :, 0])
        x = batch_norm(self.bn1, x + x_tmp)
        x_tmp = _feed_forward(self.ff, x)
        x = batch_norm(self.bn2, x + x_tmp)
        return x

    def _vector_sub(self, x):
        """attention_mode vector_sub (models/attention.py:216-225): a softmax over the D channels of each head for every
        (point, neighbour) pair -- no shipped config uses it.  The neighbour tensor comes from the HIP gather
        (ops.group), the rest is the reference's own expression in torch on the device (the stock composition, as for
        EdgeConv shapes the fused kernels do not take)."""
        B, C, N = x.shape
        H, D = self.num_heads, self.q_depth
        neighbors, _ = ops.group(x, self.K, self.group_type)                       # (B, C | 2C, N, K)
        q = self.q_conv(x[:, :, :, None]).view(B, H, D, N, 1).permute(0, 1, 3, 4, 2)   # (B,H,N,1,D)
        k = self.k_conv(neighbors).view(B, H, D, N, self.K).permute(0, 1, 3, 4, 2)     # (B,H,N,K,D)
        v = self.v_conv(neighbors).view(B, H, D, N, self.K).permute(0, 1, 3, 4, 2)
        att = torch.softmax((q - k) / math.sqrt(D), dim=-1)
        out = (att * v).sum(dim=-2)                                                    # (B,H,N,D)
        return out.permute(0, 1, 3, 2).reshape(B, C, N)


class _P2PCore(torch.autograd.Function):
    """qkv (B,N,3D) point-major -> softmax(QK^T/sqrt(D)) V for every row, (B,D,N): the single-pass flash
    kernels of the sampler (no N x N tensor), backward over all N rows."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, qkv):
        B, N, D3 = qkv.shape
        D = D3 // 3
        q, k, v = qkv[:, :, 0:D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
        O, lse, _ = ops.stage_attn_fwd(q, k, v, N, 0)
        ctx.save_for_backward(qkv, O, lse)
        return O.permute(0, 2, 1).contiguous()

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        qkv, O, lse = ctx.saved_tensors
        B, N, D3 = qkv.shape
        D = D3 // 3
        rows = torch.arange(N, device=qkv.device, dtype=torch.int64).unsqueeze(0).expand(B, -1).contiguous()
        dqkv = torch.empty_like(qkv)
        ops.stage_attn_bwd(qkv[:, :, 0:D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], O, lse, rows, g, N, 0,
                           dqkv[:, :, 0:D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:])
        return dqkv


class _P2PHeads(torch.autograd.Function):
    """qkv (B,N,3C) point-major -> per head softmax(energy / sqrt(D)) v for every point, (B,C,N): the multi-head
    kernels of csrc/attn_heads.hip (a wave owns 32 rows of one head of depth D = C / H; no (B,H,N,N) tensor).
    energy (reference models/attention.py:338-349): q k^T | -|q - k|^2 | +|q - k|^2."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, qkv, heads, asm):
        qkv = qkv.contiguous()
        B, N, C3 = qkv.shape
        C = C3 // 3
        bias = None
        if asm != "dot":  # the key term of -+|q - k|^2 (the query term is constant along a softmax row)
            k_sq = qkv[:, :, C:2 * C].reshape(B, N, heads, C // heads).square().sum(-1).permute(0, 2, 1)  # (B,H,N)
            bias = (-k_sq if asm == "l2" else k_sq).contiguous()
        out, lse = ops.stage_attn_heads_fwd(qkv, heads, asm, bias)
        ctx.save_for_backward(qkv, out, lse, bias)
        ctx.cfg = (heads, asm)
        return out.permute(0, 2, 1).contiguous()

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        qkv, out, lse, bias = ctx.saved_tensors
        heads, asm = ctx.cfg
        B, N, C3 = qkv.shape
        C = C3 // 3
        dqkv, bias_grad = ops.stage_attn_heads_bwd(qkv, heads, out, lse, g.permute(0, 2, 1).contiguous(), asm, bias)
        if bias is not None:  # bias_j = -+|k_j|^2: d bias_j / d k_j = -+2 k_j
            k = qkv[:, :, C:2 * C].reshape(B, N, heads, C // heads)
            sign = -2.0 if asm == "l2" else 2.0
            dqkv[:, :, C:2 * C] += (sign * bias_grad.permute(0, 2, 1).unsqueeze(-1) * k).reshape(B, N, C)
        return dqkv, None, None


class Point2PointAttention(nn.Module):
    """Drop-in for the reference's global self-attention layer (models/attention.py:253-355): same constructor,
    state_dict keys (`q_conv/k_conv/v_conv.weight` (C,C,1), `ff.*`, `bn1/bn2.*`) and forward (B,C,N) -> (B,C,N),
    `asm` dot / l2 / l2+, any head count dividing 128 (reference default: 4 heads of 32).

    Runs on the HIP projection and, per head of depth D = 128 / H, the multi-head attention kernels of
    csrc/attn_heads.hip (forward, dQ, dK / dV; true fp32 MFMA products, no N x N tensor); a single head with asm dot
    takes the sampler's single-pass flash kernels.  A secondary consumer: no shipped config selects the layer."""

    def __init__(self, config_attention, layer):
        num_heads = config_attention.num_heads[layer]
        q_in, q_out = config_attention.q_in[layer], config_attention.q_out[layer]
        k_in, k_out = config_attention.k_in[layer], config_attention.k_out[layer]
        v_in, v_out = config_attention.v_in[layer], config_attention.v_out[layer]
        super().__init__()
        self.attention_mode = config_attention.attention_mode[layer]
        self.asm = config_attention.asm[layer] if hasattr(config_attention, "asm") else "dot"
        if q_in != k_in or q_in != v_in or k_in != v_in:
            raise ValueError(f"q_in, k_in and v_in should be the same! Got q_in:{q_in}, k_in:{k_in}, v_in:{v_in}")
        if q_out != k_out:
            raise ValueError("q_out should be equal to k_out!")
        if q_out % num_heads != 0 or k_out % num_heads != 0 or v_out % num_heads != 0:
            raise ValueError("please set another value for num_heads!")
        if q_in != v_out:
            raise ValueError(f"q_in should be equal to v_out due to ResLink! Got q_in: {q_in}, v_out: {v_out}")
        self.num_heads = num_heads
        self.q_depth = int(q_out / num_heads)
        self.k_depth = int(k_out / num_heads)
        self.v_depth = int(v_out / num_heads)
        self.q_conv = nn.Conv1d(q_in, q_out, 1, bias=False)
        self.k_conv = nn.Conv1d(k_in, k_out, 1, bias=False)
        self.v_conv = nn.Conv1d(v_in, v_out, 1, bias=False)
        self.softmax = nn.Softmax(dim=-1)
        self.ff = nn.Sequential(
            nn.Conv1d(config_attention.ff_conv1_channels_in[layer], config_attention.ff_conv1_channels_out[layer], 1,
                      bias=False),
            nn.LeakyReLU(negative_slope=0.2),
            nn.Conv1d(config_attention.ff_conv2_channels_in[layer], config_attention.ff_conv2_channels_out[layer], 1,
                      bias=False),
        )
        self.bn1 = nn.BatchNorm1d(v_out)
        self.bn2 = nn.BatchNorm1d(v_out)
        if self.asm not in ("dot", "l2", "l2+"):
            raise ValueError("Please check the setting of asm in feature learning layer!")
        # the multi-head kernels: 128 channels, head depth a multiple of 4 (rows move in 16-byte pieces); any other shape
        # (the reference's constructor takes it; no shipped config has one) runs the expression in torch on the device
        self.hip_attention = q_in == q_out == v_out == 128 and self.q_depth % 4 == 0

    def forward(self, x):
        if not x.is_cuda:
            raise ops._lib.SambleError("samble_amd.Point2PointAttention runs on the GPU only (no CPU fallback)")
        if not self.hip_attention:
            x_tmp = self._attention_in_torch(x)
        else:
            from .downsample import _Projection
            no_tokens = self.q_conv.weight.new_zeros((1, x.shape[1], 0))
            qkv = _Projection.apply(x, no_tokens, self.q_conv.weight, self.k_conv.weight, self.v_conv.weight)
            if self.num_heads == 1 and self.asm == "dot":
                x_tmp = _P2PCore.apply(qkv)
            else:
                x_tmp = _P2PHeads.apply(qkv, self.num_heads, self.asm)
        x = batch_norm(self.bn1, x + x_tmp)
        x_tmp = _feed_forward(self.ff, x)
        x = batch_norm(self.bn2, x + x_tmp)
        return x


def _p2p_attention_in_torch(self, x):
    """models/attention.py:317-355 for shapes outside the kernels: per head softmax_j(logit_ij / sqrt(D)) V_j with logit =
    <q_i, k_j> (dot), -|q_i - k_j|^2 (l2) or +|q_i - k_j|^2 (l2+); the |q_i|^2 term is constant along a row and drops out
    of the softmax."""
    B, _, N = x.shape
    H, D = self.num_heads, self.q_depth
    q = self.q_conv(x).view(B, H, D, N)
    k = self.k_conv(x).view(B, H, D, N)
    v = self.v_conv(x).view(B, H, self.v_depth, N)
    qk = torch.matmul(q.transpose(2, 3), k)                                       # (B,H,N,N)
    if self.asm == "l2":
        qk = 2.0 * qk - k.square().sum(2).unsqueeze(2)
    elif self.asm == "l2+":
        qk = k.square().sum(2).unsqueeze(2) - 2.0 * qk
    att = torch.softmax(qk / math.sqrt(D), dim=-1)
    return torch.matmul(v, att.transpose(2, 3)).reshape(B, H * self.v_depth, N)


Point2PointAttention._attention_in_torch = _p2p_attention_in_torch


def attention_config(preset: str = "cls"):
    """`config.feature_learning_block.attention` of the shipped presets (three N2P layers, K=32,
    diff grouping, 4 heads, 128 channels, FFN 128-512-128)."""
    from .config import to_attr
    n = 3 if preset == "cls" else 5
    rep = lambda v: [v] * n  # noqa: E731
    return to_attr(dict(K=rep(32), attention_mode=rep("scalar_dot"), group_type=rep("diff"), q_in=rep(128),
                        q_out=rep(128), k_in=rep(128), k_out=rep(128), v_in=rep(128), v_out=rep(128),
                        num_heads=rep(4), ff_conv1_channels_in=rep(128), ff_conv1_channels_out=rep(512),
                        ff_conv2_channels_in=rep(512), ff_conv2_channels_out=rep(128), asm=rep("dot")))

"""CPU oracle for SAMBLE's attention-score downsampling path (TEST INFRASTRUCTURE ONLY).

This file is a staged, pure-torch CPU restatement of the reference algorithm
(stevenczwu/SAMBLE: models/downsample.py:112-344, utils/ops.py:17-44, 125-133,
174-236, 385-619).  It exists so that `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` have something to check / time the HIP path
against.  Nothing under `samble_amd/` may import it: the product path has no CPU
fallback and fails loudly when the HIP library is missing.

Pinning: every stage below is checked bit-for-bit against the imported reference
in the build container by `tests/golden/make_golden.py` (same torch build, same
ATen kernels) and the outputs are committed as fixtures under `tests/golden/`;
`tests/test_oracle_golden.py` re-checks the oracle against those fixtures on any
box.  The reference ships no tests of its own, so those
generated fixtures are the only pin there is.

Every stage is a free function so that a test can inject the oracle's stage
inputs into the matching HIP kernel (stage-wise integer exactness) as well as
compare end to end.
"""
from __future__ import annotations

import math
import numbers
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
# configuration
# --------------------------------------------------------------------------- #
@dataclass
class SamplerSpec:
    """Flat view of one layer of the reference's `config.downsample` subtree
    (configs/default.yaml:183-220 + configs/cls.yaml:119-158)."""

    M: int = 1024
    K: int = 32
    C: int = 128
    num_bins: int = 6
    asm: str = "dot"
    idx_mode: str = "sparse_col_sqr"
    sample_mode: str = "random"
    boltzmann_T: object = 0.1
    relu_mean_order: str = "mean_relu"
    token_mode: str = "multi_token"
    dynamic_boundaries: bool = True
    momentum: float = 0.99
    static_boundaries: Optional[List[float]] = None  # nb-1 descending values


# --------------------------------------------------------------------------- #
# neighbour ops  (utils/ops.py:17-44, 125-133)
# --------------------------------------------------------------------------- #
def knn(a: torch.Tensor, b: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """k nearest rows of `b` for each row of `a`; a (B,N,C), b (B,M,C).

    Follows utils/ops.py:23-43: both sets are centred on a's mean over points and
    divided by one scalar per cloud (mean over channels of a's unbiased per-channel
    std), distances come from torch.cdist, and the k largest of the negated
    distances are returned nearest first (self included when a is b).
    Returns (negated distance (B,N,k), index (B,N,k) int64)."""
    centre = a.mean(dim=1, keepdim=True)
    a0 = a - centre
    b0 = b - centre
    scale = torch.std(a0, dim=1, keepdim=True).mean(dim=2, keepdim=True)
    a0 = a0 / scale
    b0 = b0 / scale
    neg = -torch.cdist(a0, b0)
    return neg.topk(k=k, dim=-1)


def knn_mask(x: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Dense 0/1 neighbour mask of the layer input x (B,C,N) -> (B,N,N) and the
    index tensor it was scattered from (utils/ops.py:125-133).  Row i has ones at
    the k nearest neighbours of point i in *feature* space."""
    pts = x.permute(0, 2, 1)
    _, idx = knn(pts, pts, k)
    B, N, _ = idx.shape
    mask = torch.zeros(B, N, N, dtype=torch.float32)
    mask.scatter_(2, idx, 1.0)
    return mask, idx


def index_rows(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """points (B,N,C), idx (B,M,K) -> (B,M,K,C)  (utils/ops.py:5-14)."""
    shape = idx.shape
    flat = idx.reshape(shape[0], -1)
    out = torch.gather(points, 1, flat[..., None].expand(-1, -1, points.shape[-1]))
    return out.view(*shape, -1)


def group_neighbors(x: torch.Tensor, k: int, group_type: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """utils/ops.py:47-112 for the four layouts; x (B,C,N) -> ((B,C|2C,N,k), idx)."""
    pts = x.permute(0, 2, 1)
    _, idx = knn(pts, pts, k)
    nb = index_rows(pts, idx)  # (B,N,k,C)
    if group_type in ("neighbor", "center_neighbor"):
        g = nb.permute(0, 3, 1, 2)
    elif group_type in ("diff", "center_diff"):
        g = (nb - pts[:, :, None, :]).permute(0, 3, 1, 2)
    else:
        raise ValueError(f"unknown group_type {group_type}")
    if group_type.startswith("center_"):
        g = torch.cat([x[:, :, :, None].repeat(1, 1, 1, k), g], dim=1)
    return g, idx


def gather_points(pcd: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """pcd (B,C,N), idx (B,1,M) -> (B,C,M)  (utils/ops.py:136-145)."""
    return torch.gather(pcd, 2, idx.expand(-1, pcd.shape[1], -1))


def interpolate_neighbors(unknown, known, known_feature, k=3):
    """utils/ops.py:68-80: cross-set kNN; returns (neighbours (B,C,N,k), idx, +distance)."""
    kn = known.permute(0, 2, 1)
    kf = known_feature.permute(0, 2, 1)
    un = unknown.permute(0, 2, 1)
    d, idx = knn(un, kn, k)
    nb = index_rows(kf, idx).permute(0, 3, 1, 2)
    return nb, idx, -1 * d


# --------------------------------------------------------------------------- #
# attention map  (models/downsample.py:116-153)
# --------------------------------------------------------------------------- #
def project_qkv(x, tokens, wq, wk, wv):
    """x (B,C,N), tokens (1,C,nb|1), w* (C,C,1).  Returns q (B,1,N,D),
    k (B,1,D,N+nt), v (B,1,D,N+nt) exactly as downsample.py:116-137 (H=1)."""
    B = x.shape[0]
    xt = torch.cat((x, tokens.expand(B, -1, -1)), dim=2)
    q = F.conv1d(x, wq)
    k = F.conv1d(xt, wk)
    v = F.conv1d(xt, wv)
    q = q.view(B, 1, q.shape[1], q.shape[2]).permute(0, 1, 3, 2)
    k = k.view(B, 1, k.shape[1], k.shape[2])
    v = v.view(B, 1, v.shape[1], v.shape[2])
    return q, k, v


def l2_global(q, k):
    """reference utils/ops.py:115-122: q (B,H,N,D), k (B,H,D,N') -> |q_i - k_j|^2 (B,H,N,N')."""
    inner = -2 * torch.matmul(q, k)
    qq = torch.sum(q ** 2, dim=-1, keepdim=True)
    kk = torch.sum(k.transpose(-2, -1) ** 2, dim=-1, keepdim=True)
    return qq + inner + kk.transpose(-2, -1)


def attention_logits(q, k, asm: str = "dot"):
    """downsample.py:139-143 (dot) / 154-175 (l2: -|q-k|^2; the reference also projects the token
    columns through q_conv and drops those rows again, which changes nothing for rows :N)."""
    if asm == "dot":
        return (q @ k) / math.sqrt(q.shape[-1])
    if asm == "l2":
        return (-1 * l2_global(q, k)) / math.sqrt(q.shape[-1])
    raise NotImplementedError


def attention_map(q, k, n_points: int, asm: str = "dot"):
    """energy / sqrt(D); softmax over all N+nt columns (downsample.py:139-153, 154-189).
    Returns (A (B,1,N,N+nt), A_points (B,1,N,N), token logits (B,1,N,nt))."""
    logits = attention_logits(q, k, asm)
    A = torch.softmax(logits, dim=-1)
    _, tok_logits = torch.split(logits, n_points, dim=-1)
    A_pts, _ = torch.split(A, n_points, dim=-1)
    return A, A_pts, tok_logits


# --------------------------------------------------------------------------- #
# per-point score  (models/downsample.py:300-344)
# --------------------------------------------------------------------------- #
def point_score(x, A_pts, k: int, idx_mode: str):
    """Sampling score (B,1,N) from the point-to-point attention block and the
    feature-space kNN mask.  Returns (score, knn idx (B,N,k), in-degree (B,N))."""
    mask, idx = knn_mask(x, k)
    mask4 = mask.unsqueeze(1).expand(-1, A_pts.shape[1], -1, -1)
    sparse = A_pts * mask4
    indeg = torch.sum(mask4, dim=-2) + 1e-8
    if idx_mode == "col_sum":
        s = torch.sum(A_pts, dim=-2)
    elif idx_mode == "row_std":
        s = torch.std(A_pts, dim=-1)
    elif idx_mode == "sparse_row_sum":
        s = torch.sum(sparse, dim=-1)
    elif idx_mode == "sparse_row_std":
        picked = sparse.masked_select(mask4 != 0).view(sparse.shape[:-1] + (k,))
        s = torch.std(picked, dim=-1)
    elif idx_mode == "sparse_col_sum":
        s = torch.sum(sparse, dim=-2)
    elif idx_mode == "sparse_col_avg":
        s = torch.sum(sparse, dim=-2) / indeg
    elif idx_mode == "sparse_col_sqr":
        s = torch.sum(sparse, dim=-2) / indeg / indeg
    else:
        raise ValueError("Please check the setting of idx mode!")
    s[torch.isnan(s)] = 0
    return s, idx, mask.sum(dim=-2)


# --------------------------------------------------------------------------- #
# bins  (utils/ops.py:174-236, 435-464; models/downsample.py:264-284)
# --------------------------------------------------------------------------- #
def zscore(score):
    """Per-cloud z-score with population std (utils/ops.py:450-452, 517-520)."""
    return (score - score.mean(dim=2, keepdim=True)) / score.std(dim=2, unbiased=False, keepdim=True)


def quantile_ranks(num_bins: int, numel: int) -> torch.Tensor:
    """Ranks (into the descending sort) of the nb-1 boundaries: fp32 arithmetic
    then truncation, utils/ops.py:182-183."""
    return (torch.arange(1, num_bins) / num_bins * numel).int()


def batch_quantiles(z, num_bins: int) -> torch.Tensor:
    """nb-1 order statistics of ALL B*H*N z-scores, descending (utils/ops.py:180-189)."""
    ranks = quantile_ranks(num_bins, z.nelement())
    ordered, _ = torch.sort(z.flatten(), dim=0, descending=True)
    return ordered[ranks.long()]


def blend_boundaries(state, quant, num_bins: int, momentum: float):
    """utils/ops.py:201-233.  state is None (first call: raw quantiles) or
    [upper, lower] (1,1,1,nb) tensors which are UPDATED IN PLACE, as the reference
    does.  Returns the [upper, lower] pair."""
    if state is not None:
        up, lo = state[0].detach(), state[1].detach()
        mixed = up[0, 0, 0, 1:] * momentum + (1 - momentum) * quant
        up[0, 0, 0, 1:] = mixed
        lo[0, 0, 0, :-1] = mixed
        return [up, lo]
    up = torch.empty((num_bins,))
    up[0] = float("inf")
    up[1:] = quant
    lo = torch.empty((num_bins,))
    lo[-1] = float("-inf")
    lo[:-1] = quant
    return [up.reshape(1, 1, 1, num_bins), lo.reshape(1, 1, 1, num_bins)]


def static_boundary_state(values: List[float], num_bins: int):
    """models/downsample.py:96-103."""
    up = torch.asarray([float("inf")] + list(values)).reshape(1, 1, 1, num_bins)
    lo = torch.asarray(list(values) + [float("-inf")]).reshape(1, 1, 1, num_bins)
    return [up, lo]


def bin_membership(z, state):
    """(B,1,N) z-scores against [upper, lower] -> bool (B,1,N,nb); bin t holds
    lower[t] <= z < upper[t]  (utils/ops.py:454-463)."""
    z4 = z.reshape(z.shape[0], z.shape[1], z.shape[2], 1)
    return (z4 < state[0]) & (z4 >= state[1])


def bin_weights(tok_logits, member, order: str):
    """Masked mean of the token logits per bin (models/downsample.py:264-284).
    Returns (weights (B,nb), weights before relu (B,nb))."""
    masked = tok_logits * member
    if order == "mean_relu":
        pre = torch.sum(masked, dim=2) / (torch.count_nonzero(member, dim=2) + 1e-8)
        pre = pre.squeeze(1)
        return F.relu(pre), pre
    if order == "relu_mean":
        masked = F.relu(masked)
        pre = torch.sum(masked, dim=2) / (torch.count_nonzero(member, dim=2) + 1e-8)
        pre = pre.squeeze(1)
        return pre, pre
    raise NotImplementedError


def allocate_counts(weights, cap, total: int):
    """Water-filling of `total` picks over the bins of each cloud
    (utils/ops.py:385-432): weights (B,nb) fp32, cap (B,nb) int64 -> (B,nb) int32.
    The early exit is a whole-batch condition, as in the reference."""
    B, nb = weights.shape
    p = weights * cap
    p += 1e-10
    chosen = torch.zeros_like(p)
    for _ in range(nb):
        p = p / torch.sum(p, dim=1, keepdim=True)
        left = total - torch.sum(chosen, dim=1, keepdim=True)
        if torch.all(left == 0):
            break
        chosen += p * left
        chosen = torch.where(chosen >= cap, cap, chosen)
        p = p * torch.where(chosen >= cap, 0, 1)
    chosen = chosen.int()
    fix = torch.argmax(cap - chosen, dim=1)
    chosen[torch.arange(0, B), fix] += total - torch.sum(chosen, dim=1)
    return chosen


def selection_probabilities(score, member, sample_mode: str, boltzmann_T):
    """Per-(cloud,bin) sampling weights (B*nb, N), rows ordered b*nb+t
    (utils/ops.py:507-592).  `topk` has no probabilities; see selection_keys."""
    B, _, N, nb = member.shape
    if sample_mode == "uniform":
        p = member.float().squeeze(dim=1)
        p = p + (torch.sum(p, dim=1, keepdim=True) == 0)
    elif sample_mode == "random":
        t = torch.tanh(zscore(score))
        if boltzmann_T == "mode_1":
            inv_t = torch.sum(member, dim=2, keepdim=True).float() / 100.0
        elif boltzmann_T == "mode_2":
            inv_t = N / (100.0 * nb)
        elif boltzmann_T == "mode_3":
            inv_t = torch.sum(member, dim=2, keepdim=True).float() / 200.0
        elif boltzmann_T == "mode_4":
            inv_t = N / (200.0 * nb)
        elif isinstance(boltzmann_T, numbers.Number):
            inv_t = 1 / boltzmann_T
        else:
            raise NotImplementedError
        p = torch.exp(t.unsqueeze(3) * inv_t) * member
        p = p / torch.sum(p, dim=2, keepdim=True)
        p = p.squeeze(dim=1)
        p[torch.isnan(p)] = 1e-8
    else:
        raise ValueError("sample mode must be topk, uniform or random")
    return p.permute(0, 2, 1).reshape(-1, N)


def draw_noise(rows: int, n: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """The Exp(1) tensor torch.multinomial draws internally (same generator
    consumption: one exponential_ over a (rows, n) fp32 tensor)."""
    return torch.empty(rows, n, dtype=torch.float32).exponential_(1, generator=generator)


def select_indices(score, member, counts, M: int, sample_mode: str, boltzmann_T, noise=None):
    """utils/ops.py:467-619 -> idx (B,1,M) int64: bins in ascending order, inside a
    bin by descending key.  For uniform/random the key is p/noise, which is what
    torch.multinomial(p, M) without replacement evaluates (topk of p / Exp(1));
    `noise` (B*nb, N) makes the draw an explicit input."""
    B, _, N, nb = member.shape
    if sample_mode == "topk":
        keyed = (score + 1e-8).unsqueeze(3) * member
        _, order = torch.sort(keyed, dim=2, descending=True)
        order = order.squeeze(dim=1)  # (B,N,nb)
        rows = [
            torch.cat([order[b, : counts[b, t], t] for t in range(nb)]) for b in range(B)
        ]
        return torch.stack(rows).reshape(B, 1, M)
    p = selection_probabilities(score, member, sample_mode, boltzmann_T)
    if noise is None:
        picked = torch.multinomial(p, M)
    else:
        _, picked = torch.topk(p / noise, M, dim=1)
    picked = picked.reshape(B, nb, M)
    rows = [torch.cat([picked[b, t, : counts[b, t]] for t in range(nb)]) for b in range(B)]
    return torch.stack(rows).reshape(B, 1, M)


def gather_attend(A, v, idx):
    """Rows idx of the attention map times V^T -> x_ds (B,C,M)  (downsample.py:242-252)."""
    rows = torch.gather(A, dim=2, index=idx.unsqueeze(3).expand(-1, -1, -1, A.shape[-1]))
    out = (rows @ v.permute(0, 1, 3, 2)).permute(0, 2, 1, 3)
    return out.reshape(out.shape[0], out.shape[1], -1).permute(0, 2, 1)


# --------------------------------------------------------------------------- #
# the layer
# --------------------------------------------------------------------------- #
@dataclass
class SamplerState:
    """Learnable tensors + the persistent boundary state of one layer."""

    wq: torch.Tensor
    wk: torch.Tensor
    wv: torch.Tensor
    tokens: torch.Tensor
    boundaries: Optional[List[torch.Tensor]] = None
    trace: Dict[str, torch.Tensor] = field(default_factory=dict)


def sampler_forward(spec: SamplerSpec, st: SamplerState, x: torch.Tensor,
                    noise: Optional[torch.Tensor] = None,
                    world_mean=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """One DownSampleToken.forward (models/downsample.py:112-262), H=1, asm=dot.

    `noise` is the (B*nb, N) Exp(1) draw (None: torch.multinomial draws it from the
    global generator, exactly like the reference).  `world_mean` stands in for the
    all_reduce/world_size of utils/ops.py:191-199: a callable applied to the raw
    quantiles (None = single process).  Every intermediate lands in st.trace."""
    if spec.asm not in ("dot", "l2"):
        raise NotImplementedError("oracle covers asm=dot (the shipped configs) and l2")
    B, C, N = x.shape
    nb = spec.num_bins
    q, k, v = project_qkv(x, st.tokens, st.wq, st.wk, st.wv)
    A, A_pts, tok_logits = attention_map(q, k, N, spec.asm)
    score, nn_idx, indeg = point_score(x, A_pts, spec.K, spec.idx_mode)

    z = zscore(score)
    if spec.dynamic_boundaries:
        quant = batch_quantiles(z.reshape(B, 1, N, 1), nb)
        if world_mean is not None:
            quant = world_mean(quant)
        st.boundaries = blend_boundaries(st.boundaries, quant, nb, spec.momentum)
    else:
        quant = None
        if st.boundaries is None:
            st.boundaries = static_boundary_state(spec.static_boundaries, nb)
    member = bin_membership(z, st.boundaries)
    w, w_pre = bin_weights(tok_logits, member, spec.relu_mean_order)
    cap = torch.sum(member.squeeze(dim=1), dim=1)
    counts = allocate_counts(w, cap, spec.M)
    idx = select_indices(score, member, counts, spec.M, spec.sample_mode, spec.boltzmann_T, noise)
    x_ds = gather_attend(A, v, idx)

    st.trace = dict(
        q=q, k=k, v=v, tok_logits=tok_logits, score=score, knn_idx=nn_idx, indeg=indeg,
        z=z, quantiles=quant, upper=st.boundaries[0].clone(), lower=st.boundaries[1].clone(),
        member=member, w=w, w_pre=w_pre, cap=cap, counts=counts, idx=idx, x_ds=x_ds,
        lse=torch.logsumexp(attention_logits(q, k, spec.asm), dim=-1),
    )
    return x_ds, idx


def sampler_grads(spec: SamplerSpec, st: SamplerState, x: torch.Tensor, g: torch.Tensor,
                  noise: Optional[torch.Tensor] = None):
    """Forward + autograd backward for an upstream gradient g (B,C,M).
    Returns dict(dx, dwq, dwk, dwv, dtokens, x_ds, idx)."""
    leaves = [t.detach().clone().requires_grad_(True) for t in (x, st.wq, st.wk, st.wv, st.tokens)]
    st2 = SamplerState(leaves[1], leaves[2], leaves[3], leaves[4],
                       None if st.boundaries is None else [b.clone() for b in st.boundaries])
    x_ds, idx = sampler_forward(spec, st2, leaves[0], noise)
    x_ds.backward(g)
    return dict(dx=leaves[0].grad, dwq=leaves[1].grad, dwk=leaves[2].grad, dwv=leaves[3].grad,
                dtokens=leaves[4].grad, x_ds=x_ds.detach(), idx=idx, boundaries=st2.boundaries,
                trace=st2.trace)


# --------------------------------------------------------------------------- #
# Neighbor2PointAttention  (models/attention.py:130-250)
# --------------------------------------------------------------------------- #
@dataclass
class N2PState:
    """Learnable tensors of one Neighbor2PointAttention layer (Conv2d 1x1 weights (C,C,1,1),
    FFN Conv1d weights, two BatchNorm1d)."""

    wq: torch.Tensor
    wk: torch.Tensor
    wv: torch.Tensor
    ff1: torch.Tensor
    ff2: torch.Tensor
    bn1_w: torch.Tensor
    bn1_b: torch.Tensor
    bn2_w: torch.Tensor
    bn2_b: torch.Tensor


def n2p_attention(x, wq, wk, wv, k: int, heads: int, group_type: str = "diff", asm: str = "dot",
                  attention_mode: str = "scalar_dot"):
    """The attention part of Neighbor2PointAttention.forward (models/attention.py:167-185,
    203-250; scalar_dot / vector_sub, asm dot / dot-sub, all four groupings): x (B,C,N) -> (B,C,N) before the
    residual BatchNorm."""
    neighbors, idx = group_neighbors(x, k, group_type)
    B, C, N = x.shape
    D = wq.shape[0] // heads

    def split(t):  # (B,C,N,K') -> (B,H,N,K',D)
        return t.view(t.shape[0], heads, D, t.shape[2], t.shape[3]).permute(0, 1, 3, 4, 2)

    q = split(F.conv2d(x[:, :, :, None], wq))
    kk = split(F.conv2d(neighbors, wk)).permute(0, 1, 2, 4, 3)
    v = split(F.conv2d(neighbors, wv))
    if attention_mode == "scalar_dot":
        if asm == "dot":
            energy = q @ kk
        elif asm == "dot-sub":
            energy = q @ (q.transpose(-1, -2) - kk)
        else:
            raise ValueError("Please check the setting of asm in feature learning layer!")
        att = torch.softmax(energy / math.sqrt(q.shape[-1]), dim=-1)
        out = (att @ v)[:, :, :, 0, :].permute(0, 2, 1, 3)
    elif attention_mode == "vector_sub":
        energy = q.repeat(1, 1, 1, kk.shape[-1], 1) - kk.permute(0, 1, 2, 4, 3)   # (B,H,N,K,D)
        att = torch.softmax(energy / math.sqrt(q.shape[-1]), dim=-1)
        out = (att * v).permute(0, 2, 1, 3, 4).sum(dim=-2)
    else:
        raise ValueError(f"attention_mode can only be scalar_dot or vector_sub, but got: {attention_mode}")
    return out.reshape(out.shape[0], out.shape[1], -1).permute(0, 2, 1), idx


def n2p_forward(st: N2PState, x, k: int, heads: int, group_type: str = "diff", training: bool = True, asm: str = "dot",
                attention_mode: str = "scalar_dot"):
    """Whole layer (models/attention.py:165-193): attention, bn1(x + .), FFN, bn2(x + .)."""
    a, idx = n2p_attention(x, st.wq, st.wk, st.wv, k, heads, group_type, asm, attention_mode)
    y = F.batch_norm(x + a, None, None, st.bn1_w, st.bn1_b, training=True if training else False)
    f = F.conv1d(F.leaky_relu(F.conv1d(y, st.ff1), negative_slope=0.2), st.ff2)
    return F.batch_norm(y + f, None, None, st.bn2_w, st.bn2_b, training=True if training else False), a, idx


def edgeconv_forward(x, k: int, group_type: str, w1, bn1, w2, bn2):
    """models/embedding.py:7-39 (training-mode BatchNorm2d): group -> 2x(Conv2d 1x1 + BN + LeakyReLU)
    -> max over K.  bn* = (weight, bias)."""
    g, _ = group_neighbors(x, k, group_type)
    y = F.leaky_relu(F.batch_norm(F.conv2d(g, w1), None, None, bn1[0], bn1[1], training=True), negative_slope=0.2)
    y = F.leaky_relu(F.batch_norm(F.conv2d(y, w2), None, None, bn2[0], bn2[1], training=True), negative_slope=0.2)
    return y.max(dim=-1, keepdim=False)[0]


def upsample_interpolation(pcd_up, points_select, pcd_up_xyz, points_select_xyz, k, conv_w, conv_bn, res_w, res_bn):
    """models/upsample.py:164-213 with distance_type xyz, training-mode BatchNorm1d; *_bn = (weight, bias)."""
    sel = F.leaky_relu(F.batch_norm(F.conv1d(points_select, conv_w), None, None, conv_bn[0], conv_bn[1], training=True),
                       negative_slope=0.2)
    nb, _, d = interpolate_neighbors(pcd_up_xyz, points_select_xyz, sel, k)
    wts = 1.0 / (d + 1e-8)
    wts = wts / torch.sum(wts, dim=-1, keepdim=True)
    interp = torch.sum(nb * wts.unsqueeze(dim=1), dim=-1)
    x = torch.concat([pcd_up, interp], dim=1)
    return F.leaky_relu(F.batch_norm(F.conv1d(x, res_w), None, None, res_bn[0], res_bn[1], training=True),
                        negative_slope=0.2)


def norm_range(x, dim=-1, n_min=0.0, n_max=1.0, mode="minmax"):
    """reference utils/ops.py:148-171."""
    if mode == "minmax":
        lo = torch.min(x, dim=dim, keepdim=True)[0]
        xn = (x - lo) / (torch.max(x, dim=dim, keepdim=True)[0] - lo + 1e-8)
    elif mode == "sigmoid":
        xn = torch.sigmoid(x)
    elif mode == "tanh":
        xn = (torch.tanh(x) + 1.0) / 2
    elif mode == "z-score":
        return (x - torch.mean(x, dim=dim, keepdim=True)) / torch.std(x, dim=dim, unbiased=False, keepdim=True) + n_min
    else:
        raise ValueError(f"norm_range mode should be minmax, sigmoid or tanh, but got {mode}")
    return xn * (n_max - n_min) + n_min


def sort_chunk(score, num_bins: int, dim: int = -1, descending: bool = False):
    """reference utils/ops.py:239-259: sorted scores and their indices, each cut into num_bins chunks."""
    x_sorted, idx_sorted = torch.sort(score, dim=dim, descending=descending)
    return torch.chunk(x_sorted, num_bins, dim=dim), torch.chunk(idx_sorted, num_bins, dim=dim)


def fps(x, xyz, npoint: int, start: torch.Tensor):
    """reference utils/ops.py:646-692 (index_points_for_fps + fps) with the first centroid given:
    x (B,C,N), xyz (B,3,N) -> ((x at the farthest points (B,C,npoint), idx (B,1,npoint)), (None, None))."""
    idx = farthest_point_sample(xyz.permute(0, 2, 1), npoint, start)
    pts = x.permute(0, 2, 1)
    batch = torch.arange(x.shape[0], dtype=torch.long).view(-1, 1).repeat(1, npoint)
    return (pts[batch, idx, :].permute(0, 2, 1), idx.unsqueeze(1)), (None, None)


def local_sampler_forward(x, wq, wk, wv, M: int, idx_mode: str = "local_std", k: int = 32, asm: str = "dot"):
    """DownSampleLocal (reference models/downsample.py:818-1229) for one head, no boltzmann draw:
    local 1 x K attention of every point over its K nearest neighbours (in feature space), a per-point
    score from that map, top-M kept, bottom-(N-M) by the map's std dropped.  wq/wk/wv (C,C,1,1).
    -> (x_ds (B,C,M), idx (B,1,M)), (x_dropped (B,C,N-M), idx_dropped (B,1,N-M)), score (B,1,N), att (B,1,N,1,K)"""
    B, C, N = x.shape
    group_type = "diff" if asm == "dot" else "neighbor"
    neighbors, nidx = group_neighbors(x, k, group_type)          # (B,C,N,K), (B,N,K)
    q = torch.nn.functional.conv2d(x[:, :, :, None], wq)          # (B,C,N,1)
    kk = torch.nn.functional.conv2d(neighbors, wk)                # (B,C,N,K)
    vv = torch.nn.functional.conv2d(neighbors, wv)

    def heads(t):  # (B,C,N,K) -> (B,1,N,K,D)
        return t.view(B, 1, C, t.shape[2], t.shape[3]).permute(0, 1, 3, 4, 2)
    q, kk, vv = heads(q), heads(kk), heads(vv)
    kt = kk.permute(0, 1, 2, 4, 3)                                 # (B,1,N,D,K)
    # attention_scoring, models/downsample.py:977-1000
    if asm in ("dot", "dot-neighbor"):
        energy = q @ kt                                            # (B,1,N,1,K)
    elif asm == "dot-sub":
        energy = q @ (q.transpose(-1, -2) - kt)
    elif asm in ("l2", "l2+"):
        energy = (q - kt.transpose(-1, -2)) @ (q.transpose(-1, -2) - kt)   # (B,1,N,K,K)
        if asm == "l2":
            energy = -1 * energy
        energy = torch.mean(energy, dim=-2).unsqueeze(-2)
    else:
        raise ValueError("Please check the setting of asm!")
    att = torch.softmax(energy / math.sqrt(q.shape[-1]), dim=-1)
    a2 = att.squeeze(-2)                                           # (B,1,N,K)
    idx4 = nidx.view(B, 1, N, k)
    sparse = torch.zeros(B, 1, N, N, dtype=torch.float32).scatter_(-1, idx4, a2)
    mask = torch.zeros(B, 1, N, N, dtype=torch.float32).scatter_(-1, idx4, 1.0)
    num = torch.sum(mask, dim=-2) + 1e-8
    if idx_mode == "local_std":
        score = torch.std(att, dim=-1, unbiased=False)[:, :, :, 0]
    elif idx_mode == "sparse_row_std":
        score = torch.std(sparse.masked_select(mask != 0).view(B, 1, N, k), dim=-1)
    elif idx_mode == "sparse_col_sum":
        score = torch.sum(sparse, dim=-2)
    elif idx_mode == "sparse_col_avg":
        score = torch.sum(sparse, dim=-2) / num
    elif idx_mode == "sparse_col_sqr":
        score = torch.sum(sparse, dim=-2) / num / num
    else:
        raise ValueError("Please check the setting of idx mode!")
    idx = score.topk(M, dim=-1)[1]
    idx_dropped = torch.std(att, dim=-1, unbiased=False)[:, :, :, 0].topk(N - M, dim=-1, largest=False)[1]

    def attend(sel):
        a = torch.gather(att, 2, sel[..., None, None].expand(-1, -1, -1, -1, k))
        v = torch.gather(vv, 2, sel[..., None, None].expand(-1, -1, -1, k, C))
        o = (a @ v)[:, :, :, 0, :].permute(0, 2, 1, 3)             # (B,M,1,D)
        return o.reshape(B, o.shape[1], -1).permute(0, 2, 1)
    return (attend(idx), idx), (attend(idx_dropped), idx_dropped), score, att


def farthest_point_sample(xyz: torch.Tensor, npoint: int, start: torch.Tensor) -> torch.Tensor:
    """reference utils/ops.py:622-643 with the random first centroid made an input (`start`, (B,)
    int64; the reference draws it with torch.randint).  xyz (B,N,3) -> (B,npoint) int64."""
    B, N, _ = xyz.shape
    centroids = torch.zeros(B, npoint, dtype=torch.long)
    distance = torch.ones(B, N) * 1e10
    farthest = start.clone().long()
    batch = torch.arange(B, dtype=torch.long)
    for i in range(npoint):
        centroids[:, i] = farthest
        centroid = xyz[batch, farthest, :].view(B, 1, 3)
        dist = torch.sum((xyz - centroid) ** 2, -1)
        mask = dist < distance
        distance[mask] = dist[mask]
        farthest = torch.max(distance, -1)[1]
    return centroids


def global_sampler_forward(x, wq, wk, wv, M: int, idx_mode: str = "col_sum", asm: str = "dot", knn_k: int = 32,
                           num_heads: int = 1):
    """DownSampleGlobal.forward (models/downsample.py:1281-1330, no res block) with split_heads (1332-1336) and
    attention_scoring (models/downsample.py:1338-1358) for asm dot / dot-sub / l2 / l2+.
    Returns ((x_ds, idx (B,H,M)), (x_dropped, idx_dropped (B,H,N-M)), score (B,H,N))."""
    B, C, N = x.shape
    H = num_heads
    q = F.conv1d(x, wq).view(B, H, wq.shape[0] // H, N).permute(0, 1, 3, 2)
    k = F.conv1d(x, wk).view(B, H, wk.shape[0] // H, N)
    v = F.conv1d(x, wv).view(B, H, wv.shape[0] // H, N)
    if asm == "dot":
        energy = q @ k
    elif asm == "dot-sub":
        energy = q @ (q.transpose(-1, -2) - k)
    elif asm == "l2":
        energy = -1 * l2_global(q, k)
    elif asm == "l2+":
        energy = l2_global(q, k)
    else:
        raise ValueError("Please check the setting of asm!")
    A = torch.softmax(energy / math.sqrt(q.shape[-1]), dim=-1)
    if idx_mode == "col_sum":
        score = torch.sum(A, dim=-2)
    elif idx_mode == "row_std":
        score = torch.std(A, dim=-1)
    else:
        # idx_selection's sparse branch (models/downsample.py:1383-1401).  NOT DownSampleToken's formulas: the row
        # deviation runs over all N entries of the masked row (zeros included), the in-degree is used as it is (no
        # 1e-8, no NaN -> 0: a column nobody lists gives 0/0 = NaN, which topk ranks first)
        mask, _ = knn_mask(x, knn_k)
        mask4 = mask.unsqueeze(1).expand(-1, A.shape[1], -1, -1)
        sam = A * mask4
        num = torch.sum(mask4, dim=-2)
        if idx_mode == "sparse_row_sum":
            score = torch.sum(sam, dim=-1)
        elif idx_mode == "sparse_row_std":
            score = torch.std(sam, dim=-1)
        elif idx_mode == "sparse_col_sum":
            score = torch.sum(sam, dim=-2)
        elif idx_mode == "sparse_col_avg":
            score = torch.sum(sam, dim=-2) / num
        elif idx_mode == "sparse_col_sqr":
            score = torch.sum(sam, dim=-2) / num / num
        elif idx_mode == "sparse_col_sum_sqr":
            cs = torch.sum(sam, dim=-2)
            score = 0.5 * (cs / num / num) + 0.5 * cs
        else:
            raise ValueError("Please check the setting of idx mode!")
    idx = score.topk(M, dim=-1)[1]
    idx_dropped = torch.sum(A, dim=-2).topk(N - M, dim=-1, largest=False)[1]
    def rows(ix):
        a = torch.gather(A, dim=2, index=ix[..., None].expand(-1, -1, -1, N))
        o = (a @ v.permute(0, 1, 3, 2)).permute(0, 2, 1, 3)
        return o.reshape(o.shape[0], o.shape[1], -1).permute(0, 2, 1)
    return (rows(idx), idx), (rows(idx_dropped), idx_dropped), score

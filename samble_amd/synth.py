"""Deterministic synthetic point clouds and weights.

Real ModelNet40 / ShapeNet-part files are not available (no network), so every
benchmark, test and golden fixture uses inputs from this counter-based generator:
value i of stream `seed` depends only on (seed, i), is built from integer hashes
with one final rounding, and is therefore bit-identical on every box and for
every shard of a batch (a rank can generate exactly its own clouds).

Shapes follow SURVEY.md section 8(d): features ~ N(0,1) (the layer's real input is
a BatchNorm output, reference models/attention.py:191), xyz = jittered,
anisotropically scaled unit-sphere samples (reference utils/data_augmentation.py:4-10, 56-75).
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays."""
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _bits(n: int, seed: int, lane: int = 0, start: int = 0) -> np.ndarray:
    """n uint64 hashes for counters start..start+n-1 of stream (seed, lane)."""
    with np.errstate(over="ignore"):
        ctr = np.arange(start, start + n, dtype=np.uint64)
        key = _mix(np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(lane) + np.uint64(1))
        return _mix(ctr * np.uint64(0xD1342543DE82EF95) + key)


def uniform(shape, seed: int, start: int = 0) -> np.ndarray:
    """U(0,1] with 24-bit resolution, float32, never 0."""
    n = int(np.prod(shape))
    h = _bits(n, seed, 0, start) >> np.uint64(40)
    return ((h.astype(np.float64) + 1.0) / 16777216.0).astype(np.float32).reshape(shape)


def normal(shape, seed: int, start: int = 0) -> np.ndarray:
    """Approximately N(0,1): sum of 12 24-bit uniforms minus 6 (exact integer sum,
    one rounding to float32)."""
    n = int(np.prod(shape))
    acc = np.zeros(n, dtype=np.int64)
    for lane in range(6):
        h = _bits(n, seed, lane + 1, start)
        acc += (h >> np.uint64(40)).astype(np.int64)
        acc += ((h >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.int64)
    return ((acc.astype(np.float64) + 6.0) / 16777216.0 - 6.0).astype(np.float32).reshape(shape)


def exp1(shape, seed: int, start: int = 0) -> np.ndarray:
    """Exp(1) noise for the Boltzmann/uniform selection (the draw torch.multinomial
    makes internally).  -log(u) is taken in float64 and rounded once."""
    u = uniform(shape, seed, start).astype(np.float64)
    return (-np.log(u) + 2.0 ** -30).astype(np.float32)


def features(batch: int, channels: int, n_points: int, seed: int, first_cloud: int = 0) -> np.ndarray:
    """(batch, channels, n_points) float32 layer input; cloud b of the global batch
    is the same whichever shard generates it."""
    per = channels * n_points
    out = np.empty((batch, channels, n_points), dtype=np.float32)
    for b in range(batch):
        out[b] = normal((channels, n_points), seed, start=(first_cloud + b) * per)
    return out


def xyz_clouds(batch: int, n_points: int, seed: int, first_cloud: int = 0) -> np.ndarray:
    """(batch, 3, n_points) float32 coordinates: points on the unit sphere, N(0,0.01^2)
    jitter clipped to +-0.05, per-cloud anisotropic scale U(0.66,1.5)."""
    out = np.empty((batch, 3, n_points), dtype=np.float32)
    for b in range(batch):
        c = first_cloud + b
        g = normal((3, n_points), seed, start=c * 3 * n_points).astype(np.float64)
        g /= np.sqrt((g * g).sum(axis=0, keepdims=True)) + 1e-12
        jit = np.clip(0.01 * normal((3, n_points), seed + 1, start=c * 3 * n_points), -0.05, 0.05)
        sc = 0.66 + (1.5 - 0.66) * uniform((3, 1), seed + 2, start=c * 3).astype(np.float64)
        out[b] = ((g + jit) * sc).astype(np.float32)
    return out


def sampler_weights(channels: int, num_tokens: int, seed: int):
    """(wq, wk, wv (C,C,1), tokens (1,C,nt)) with the reference's init scales:
    Conv1d default init is U(-1/sqrt(C), 1/sqrt(C)) (kaiming_uniform a=sqrt(5));
    tokens ~ N(0, 1/sqrt(C)) (reference models/downsample.py:60-69)."""
    bound = 1.0 / np.sqrt(channels)
    ws = []
    for i in range(3):
        u = uniform((channels, channels, 1), seed * 16 + i).astype(np.float64)
        ws.append(((2.0 * u - 1.0) * bound).astype(np.float32))
    tok = (normal((1, channels, num_tokens), seed * 16 + 3).astype(np.float64) * bound).astype(np.float32)
    return ws[0], ws[1], ws[2], tok

#!/usr/bin/env python3
"""Golden vectors for the small free functions of the reference's utils/ops.py that the samplers' variants call
(norm_range 148-171, sort_chunk 239-259, l2_global 115-122, fps / index_points_for_fps 646-692), produced by the
UNMODIFIED reference on CPU and checked bit-for-bit against oracle/torch_oracle.py.  Run from the repo root:
    python tests/golden/make_golden_ops.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("SAMBLE_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch

from samble_amd import synth
from oracle import torch_oracle as O
from tests.golden.make_golden import same

from utils import ops as ref_ops  # noqa: E402  (the reference)


def main():
    torch.set_num_threads(8)
    seed = 9100
    B, H, N, D, nb = 3, 1, 500, 16, 6
    score = torch.from_numpy(synth.normal((B, H, N), seed)) * 0.7 + 0.1
    score[0, 0, 17] = score[0, 0, 400]          # an exact tie (its order is unspecified in torch.sort: tests compare values)
    out = dict(meta=np.array([B, H, N, D, nb, seed], dtype=np.int64), torch_version=np.array(torch.__version__))
    for mode in ("minmax", "sigmoid", "tanh", "z-score"):
        r = ref_ops.norm_range(score, dim=-1, n_min=0.25, n_max=2.0, mode=mode)
        same(O.norm_range(score, dim=-1, n_min=0.25, n_max=2.0, mode=mode), r, f"norm_range {mode}")
        out["norm_" + mode.replace("-", "")] = r.numpy()
    for desc in (False, True):
        xs, ids = ref_ops.sort_chunk(score, nb, dim=-1, descending=desc)
        oxs, oids = O.sort_chunk(score, nb, dim=-1, descending=desc)
        assert len(xs) == len(oxs) == nb
        for a, b_, c, d in zip(xs, oxs, ids, oids):
            same(b_, a, "sort_chunk values"); same(d, c, "sort_chunk indices")
        tag = "desc" if desc else "asc"
        out[f"chunk_sizes_{tag}"] = np.array([t.shape[-1] for t in xs], dtype=np.int64)
        out[f"sorted_{tag}"] = torch.cat(xs, dim=-1).numpy()
        out[f"order_{tag}"] = torch.cat(ids, dim=-1).numpy()
    q = torch.from_numpy(synth.normal((B, H, 40, D), seed + 1))
    k = torch.from_numpy(synth.normal((B, H, D, 40), seed + 2))
    r = ref_ops.l2_global(q, k)
    same(O.l2_global(q, k), r, "l2_global")
    out["l2_global"] = r.numpy()
    # fps(x, xyz, npoint): the reference draws the first centroid with torch.randint under the global seed
    Bf, Nf, Cf, npnt = 2, 300, 8, 64
    xyz = torch.from_numpy(synth.xyz_clouds(Bf, Nf, seed + 3))          # (B,3,N)
    x = torch.from_numpy(synth.normal((Bf, Cf, Nf), seed + 4))
    torch.manual_seed(seed)
    (xf, idf), rest = ref_ops.fps(x, xyz, npnt)
    assert rest == (None, None)
    start = idf[:, 0, 0].clone()
    (oxf, oidf), _ = O.fps(x, xyz, npnt, start)
    same(oidf, idf, "fps idx"); same(oxf, xf, "fps x")
    out.update(fps_meta=np.array([Bf, Nf, Cf, npnt, seed], dtype=np.int64), fps_start=start.numpy(), fps_idx=idf.numpy(),
               fps_x=xf.numpy())
    path = os.path.join(HERE, "layer_ops_small.npz")
    np.savez_compressed(path, **out)
    print(f"layer_ops_small: ok, {os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Which switch moves the mid-size block's gradient error (tests/golden/block_cls_mid.npz): the same comparison as
tests/test_gpu_block.py::test_cls_block_mid_size_against_an_unpicked_reference_fixture under the package's A/B switches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from samble_amd import synth, attention as A, blocks as BK, embedding as E, linear as L, ops
from tests.test_gpu_block import _reference_block, _stored_rows

def run(label, **sw):
    olds = {}
    for k, v in sw.items():
        mod, name = k.split("__")
        m = {"A": A, "BK": BK, "E": E, "L": L, "ops": ops}[mod]
        olds[k] = getattr(m, name)
        setattr(m, name, v)
    try:
        d, blk, xyz, noise = _reference_block("cls", "mid")
        seed = int(d["meta"][5])
        forced = [torch.from_numpy(d["idx0"]).cuda(), torch.from_numpy(d["idx1"]).cuda()]
        feat, res = blk(xyz, noise_list=noise, forced_idx_list=forced)
        ferr = float((feat.detach().cpu() - torch.from_numpy(d["feat"])).abs().max())
        feat.backward(torch.from_numpy(synth.normal(tuple(feat.shape), seed + 900)).cuda())
        params = dict(blk.named_parameters())
        errs = {}
        for name in [str(k) for k in d["grad_keys"]]:
            ref = torch.from_numpy(d["grad/" + name])
            got = _stored_rows(params[name].grad, d["grad/" + name])
            errs[name.replace("feature_learning_layer_list", "fl").replace("downsample_list", "ds").replace("embedding_list", "emb")] = float((got - ref).abs().max() / ref.abs().max())
        print(f"{label:34s} feat {ferr:.1e} | " + " ".join(f"{k.split('.')[0]}.{k.split('.')[1]}.{k.split('.')[2][:4]}={v:.0e}" for k, v in errs.items()), flush=True)
    finally:
        for k, v in olds.items():
            mod, name = k.split("__")
            setattr({"A": A, "BK": BK, "E": E, "L": L, "ops": ops}[mod], name, v)

run("default")
run("FUSED_LAYER off", A__FUSED_LAYER=False)
run("OWN_BATCHNORM off (+layer off)", A__OWN_BATCHNORM=False, A__FUSED_LAYER=False)
run("FUSED_FFN off (+layer off)", A__FUSED_FFN=False, A__FUSED_LAYER=False)
run("FUSED_HEADS off", BK__FUSED_HEADS=False)
run("EdgeConv FUSED_GLUE off", E__FUSED_GLUE=False)
run("EdgeConv FUSED_PROJECTIONS off", E__FUSED_PROJECTIONS=False)
run("matrix mode f32", ops__MATRIX_MODE="f32")

#!/usr/bin/env python3
"""Golden fixture for the trainer-side state: a checkpoint WRITTEN BY THE REFERENCE and what the
reference computes after restoring it.

1. The unmodified reference classification block (models/cls_model.py:10-145, cls.yaml: dynamic boundaries) trains for
   three steps (forward, backward, SGD) on three different batches, so that the samplers' `bin_boundaries` have gone
   through "first call = raw quantiles" and two momentum blends (utils/ops.py:201-233), BatchNorm's running statistics
   have moved and the weights are no longer the initial ones.
2. The dict the trainer saves is built exactly as train_modelnet.py:497-505 builds it:
   {"model_state_dict": model.state_dict(), "bin_boundaries": [layer.bin_boundaries for layer in downsample_list]}.
3. It is restored the way test_modelnet.py:158-171 restores it: `bin.dynamic_boundaries_enable` off,
   `bin.bin_boundaries = [b[0][0, 0, 0, 1:].tolist() for b in saved]`, a block constructed from THAT config, the
   state_dict loaded, eval mode.  (The script as shipped edits the config after it has constructed the model, and reads
   the key `dynamic_boundaries` where the config has `dynamic_boundaries_enable` -- SURVEY.md section 5; this fixture
   follows what the edit is for: the saved boundaries as the static ones of the evaluated model.)
4. The restored block's eval-mode forward on a fresh batch is recorded: both samplers' indices, the boundaries each
   sampler holds after the call, the block's output.

Stored as plain arrays (no pickled classes): "sd/<key>" = every state_dict entry, "bb/<layer>/upper|lower", the eval
batch's noise and outputs.  Run from the repo root:
    python tests/golden/make_golden_ckpt.py"""
import copy
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
from tests.golden.make_golden_layers import block_config
from tests.util import fill_parameters

from models import cls_model as ref_cls  # noqa: E402

TRAIN_STEPS = 3
LR = 1e-3


def main():
    torch.set_num_threads(8)
    B, N, M, seed = 2, 256, [128, 64], 9500
    cfg = block_config("cls")
    cfg.downsample.M = list(M)
    assert cfg.downsample.bin.dynamic_boundaries_enable is True
    blk = ref_cls.FeatureLearningBlock(cfg)
    fill_parameters(blk, seed)
    blk.train()
    opt = torch.optim.SGD(blk.parameters(), lr=LR)
    nb = cfg.downsample.bin.num_bins[0]
    history = []
    for step in range(TRAIN_STEPS):
        xyz = torch.from_numpy(synth.xyz_clouds(B, N, seed + 500 + step))
        torch.manual_seed(seed + step)
        feat, _ = blk(xyz)
        opt.zero_grad()
        (feat * torch.from_numpy(synth.normal(tuple(feat.shape), seed + 900 + step))).sum().backward()
        opt.step()
        history.append([m.bin_boundaries[0][0, 0, 0, 1:].clone() for m in blk.downsample_list])
    for a, b in zip(history[0], history[-1]):
        assert not torch.equal(a, b), "the boundaries did not move"

    # train_modelnet.py:497-505
    state_dict = {
        "model_state_dict": blk.state_dict(),
        "bin_boundaries": [downsample_module.bin_boundaries for downsample_module in blk.downsample_list],
    }

    # test_modelnet.py:158-171, in the order that gives the edit its effect: config first, then the model
    cfg_eval = block_config("cls")
    cfg_eval.downsample.M = list(M)
    cfg_eval.downsample.bin.dynamic_boundaries_enable = False
    cfg_eval.downsample.bin.bin_boundaries = [
        bin_boundaries[0][0, 0, 0, 1:].tolist() for bin_boundaries in state_dict["bin_boundaries"]
    ]
    restored = ref_cls.FeatureLearningBlock(cfg_eval)
    restored.load_state_dict(copy.deepcopy(state_dict["model_state_dict"]))
    restored.eval()
    xyz_eval = torch.from_numpy(synth.xyz_clouds(B, N, seed + 700))
    with torch.no_grad():
        torch.manual_seed(seed + 70)
        feat_eval, _ = restored(xyz_eval)
    torch.manual_seed(seed + 70)
    noise0 = O.draw_noise(B * nb, N)
    noise1 = O.draw_noise(B * nb, M[0])

    out = dict(meta=np.array([B, N, M[0], M[1], nb, seed, TRAIN_STEPS], dtype=np.int64),
               names=np.array(list(state_dict["model_state_dict"].keys())),
               feat=feat_eval.numpy(), noise0=noise0.numpy(), noise1=noise1.numpy(),
               idx0=restored.downsample_list[0].idx.numpy(), idx1=restored.downsample_list[1].idx.numpy(),
               score0=restored.downsample_list[0].attention_point_score.numpy(),
               torch_version=np.array(torch.__version__))
    for k, v in state_dict["model_state_dict"].items():
        out["sd/" + k] = v.detach().numpy()
    for i, (upper, lower) in enumerate(state_dict["bin_boundaries"]):
        out[f"bb/{i}/upper"] = upper.detach().numpy()
        out[f"bb/{i}/lower"] = lower.detach().numpy()
        # the static boundaries the restored sampler holds after its eval call: unchanged by the call
        after = restored.downsample_list[i].bin_boundaries
        assert torch.equal(after[0], upper) and torch.equal(after[1], lower), "static boundaries moved"
    path = os.path.join(HERE, "ckpt_cls_block.npz")
    np.savez_compressed(path, **out)
    print("ckpt_cls_block: ok,", os.path.getsize(path) // 1024, "KiB;", len(state_dict["model_state_dict"]),
          "state_dict entries; boundaries layer 0:", state_dict["bin_boundaries"][0][0].flatten().tolist())


if __name__ == "__main__":
    main()

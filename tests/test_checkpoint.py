"""a checkpoint written by the REFERENCE trainer's recipe (train_modelnet.py:497-505) after three
training steps of the unmodified reference block, restored into this package's block the way the reference's
evaluation script restores it (test_modelnet.py:158-171), against what the restored reference block computes
(tests/golden/make_golden_ckpt.py -> tests/golden/ckpt_cls_block.npz)."""
import numpy as np
import pytest
import torch

from samble_amd import checkpoint, synth
from tests.util import layer_fixture, set_agreement

DEV = "cuda:0"


def _reference_checkpoint():
    """The fixture as the dict torch.load() of the reference's checkpoint.pt returns."""
    d = layer_fixture("ckpt_cls_block")
    sd = {str(k): torch.from_numpy(np.array(d["sd/" + str(k)])) for k in d["names"]}
    nlayers = len([k for k in d.files if k.startswith("bb/") and k.endswith("/upper")])
    bounds = [[torch.from_numpy(d[f"bb/{i}/upper"]), torch.from_numpy(d[f"bb/{i}/lower"])] for i in range(nlayers)]
    return d, {"model_state_dict": sd, "bin_boundaries": bounds}


def _block(d):
    from samble_amd.blocks import FeatureLearningBlock, block_config
    B, N, M0, M1, nb, seed, steps = [int(v) for v in d["meta"]]
    return FeatureLearningBlock(block_config("cls", M=(M0, M1)))


def test_reference_checkpoint_loads_strictly_on_the_host():
    """state_dict keys (parameters AND BatchNorm buffers) are the reference's, so its checkpoint loads with
    strict=True; the boundaries arrive as [upper, lower] of (1,1,1,nb) with the +-inf sentinels; freeze=True makes them
    static; `static_boundary_values` is the reference's `b[0][0,0,0,1:].tolist()`."""
    d, state = _reference_checkpoint()
    blk = _block(d)
    assert list(blk.state_dict().keys()) == [str(k) for k in d["names"]]
    checkpoint.load_checkpoint(blk, state, freeze=True)
    for k, v in blk.state_dict().items():
        assert torch.equal(v, state["model_state_dict"][k]), k
    for layer, (upper, lower) in zip(blk.downsample_list, state["bin_boundaries"]):
        assert layer.dynamic_boundaries_enable is False
        assert torch.equal(layer.bin_boundaries[0].cpu(), upper) and torch.equal(layer.bin_boundaries[1].cpu(), lower)
        assert layer.bin_boundaries[0].shape == (1, 1, 1, layer.num_bins)
        assert float(upper[0, 0, 0, 0]) == float("inf") and float(lower[0, 0, 0, -1]) == float("-inf")
        assert layer.bin_boundaries[0].data_ptr() != upper.data_ptr()   # the module owns its copy
    vals = checkpoint.static_boundary_values(state)
    assert vals == [b[0][0, 0, 0, 1:].tolist() for b in state["bin_boundaries"]]
    # the boundaries have been through the momentum blend: three different values per layer are not raw quantiles of
    # one batch, but they are still strictly descending
    for v in vals:
        assert all(a > b for a, b in zip(v, v[1:]))
    # what our own writer produces from the loaded block is the same dict again
    again = checkpoint.checkpoint_dict(blk)
    assert list(again["model_state_dict"].keys()) == list(state["model_state_dict"].keys())
    for (u, l), (u2, l2) in zip(again["bin_boundaries"], state["bin_boundaries"]):
        assert torch.equal(u.cpu(), u2) and torch.equal(l.cpu(), l2)


def test_resume_keeps_the_momentum_state_dynamic():
    d, state = _reference_checkpoint()
    blk = _block(d)
    checkpoint.load_checkpoint(blk, state, freeze=False)
    for layer in blk.downsample_list:
        assert layer.dynamic_boundaries_enable is True and layer.bin_boundaries is not None


@pytest.mark.gpu
def test_eval_after_restore_matches_the_restored_reference():
    """load_checkpoint(freeze=True) -> eval() -> forward with the reference's multinomial noise: both samplers' indices
    and the block output against the restored reference block's; then, unconditionally, the output through the
    reference's indices.  The static boundaries are not touched by the call (utils/ops.py:458-463)."""
    d, state = _reference_checkpoint()
    B, N, M0, M1, nb, seed, steps = [int(v) for v in d["meta"]]
    blk = _block(d)
    checkpoint.load_checkpoint(blk, state, freeze=True)
    blk = blk.to(DEV).eval()
    xyz = torch.from_numpy(synth.xyz_clouds(B, N, seed + 700)).to(DEV)
    noise = [torch.from_numpy(d["noise0"]).to(DEV), torch.from_numpy(d["noise1"]).to(DEV)]
    with torch.no_grad():
        feat, _ = blk(xyz, noise_list=noise)
    ds0, ds1 = blk.downsample_list
    torch.testing.assert_close(ds0.attention_point_score.cpu(), torch.from_numpy(d["score0"]), rtol=2e-3, atol=1e-8)
    for layer, (upper, lower) in zip(blk.downsample_list, state["bin_boundaries"]):
        assert torch.equal(layer.bin_boundaries[0].cpu(), upper) and torch.equal(layer.bin_boundaries[1].cpu(), lower)
    idx0, idx1 = ds0.idx.cpu()[:, 0], ds1.idx.cpu()[:, 0]
    ref0, ref1 = torch.from_numpy(d["idx0"])[:, 0], torch.from_numpy(d["idx1"])[:, 0]
    same0, same1 = int((idx0 == ref0).all(1).sum()), int((idx1 == ref1).all(1).sum())
    print(f"restored block, eval: clouds with the reference's exact index tensor: layer 0 {same0}/{B}, layer 1 "
          f"{same1}/{B}; set agreement {set_agreement(idx0, ref0):.4f} / {set_agreement(idx1, ref1):.4f}")
    assert set_agreement(idx0, ref0) >= 0.97
    # the fixed boundaries put every point of the first layer in the reference's bin: the per-bin counts match
    forced = [torch.from_numpy(d["idx0"]).to(DEV), torch.from_numpy(d["idx1"]).to(DEV)]
    with torch.no_grad():
        feat_f, _ = blk(xyz, noise_list=noise, forced_idx_list=forced)
    torch.testing.assert_close(feat_f.cpu(), torch.from_numpy(d["feat"]), rtol=2e-3, atol=2e-3)
    if same0 == B and same1 == B:
        assert torch.equal(feat, feat_f)     # the own-selection pass went through the same rows: same kernels, same bits


@pytest.mark.gpu
def test_dynamic_resume_continues_the_blend_on_the_device():
    """freeze=False: the next training call blends the loaded boundaries with the new batch's quantiles,
    new = 0.99 old + 0.01 q (utils/ops.py:201-213) -- it does not re-initialise them."""
    d, state = _reference_checkpoint()
    B, N, M0, M1, nb, seed, steps = [int(v) for v in d["meta"]]
    blk = _block(d)
    checkpoint.load_checkpoint(blk, state, freeze=False)
    blk = blk.to(DEV).train()
    xyz = torch.from_numpy(synth.xyz_clouds(B, N, seed + 700)).to(DEV)
    blk(xyz)
    for layer, (upper, lower) in zip(blk.downsample_list, state["bin_boundaries"]):
        new = layer.bin_boundaries[0].cpu()[0, 0, 0, 1:]
        old = upper[0, 0, 0, 1:]
        z = layer.normalized_score.flatten().cpu()
        q = torch.sort(z, descending=True)[0][(torch.arange(1, nb) / nb * z.numel()).int().long()]
        torch.testing.assert_close(new, old * 0.99 + (1 - 0.99) * q, rtol=1e-6, atol=1e-7)
        assert torch.equal(layer.bin_boundaries[1].cpu()[0, 0, 0, :-1], new)

"""Trainer-side state of the sampler: `bin_boundaries` is a plain attribute of each DownSampleToken,
not a buffer, so the reference saves it next to the state_dict (train_modelnet.py:493-509) and the
evaluation scripts turn it into static boundaries (test_modelnet.py:161-171).  These helpers keep
that on-disk format: {"model_state_dict": ..., "bin_boundaries": [[upper, lower], ...]}."""
from __future__ import annotations

from typing import Dict, Iterable, List

import torch


def sampler_layers(module: torch.nn.Module) -> List[torch.nn.Module]:
    from .downsample import DownSampleToken
    return [m for m in module.modules() if isinstance(m, DownSampleToken)]


def checkpoint_dict(model: torch.nn.Module) -> Dict:
    """What the reference trainer writes when dynamic boundaries are on."""
    return {"model_state_dict": model.state_dict(),
            "bin_boundaries": [layer.bin_boundaries for layer in sampler_layers(model)]}


def load_checkpoint(model: torch.nn.Module, state: Dict, freeze: bool = False) -> None:
    """Load a reference-format checkpoint.  freeze=False resumes training (boundaries keep their
    momentum state); freeze=True reproduces the evaluation scripts: boundaries become static."""
    model.load_state_dict(state["model_state_dict"] if "model_state_dict" in state else state)
    layers = sampler_layers(model)
    saved: Iterable = state.get("bin_boundaries", []) if isinstance(state, dict) else []
    for layer, bounds in zip(layers, saved):
        if bounds is None:
            continue
        layer.bin_boundaries = [bounds[0].clone(), bounds[1].clone()]
        if freeze:
            layer.dynamic_boundaries_enable = False


def static_boundary_values(state: Dict) -> List[List[float]]:
    """The nb-1 interior boundaries per layer, as test_modelnet.py:166-169 feeds them back into
    `config.downsample.bin.bin_boundaries`."""
    return [b[0][0, 0, 0, 1:].tolist() for b in state["bin_boundaries"]]

"""Attribute-style config helpers.

The reference builds its modules from a hydra/omegaconf tree (`config.feature_learning_block.
downsample`, reference configs/default.yaml:183-220 overlaid by configs/cls.yaml:119-158 or
configs/seg.yaml:102-121).  hydra/omegaconf are not needed here: any object with attribute access
works (an omegaconf node from the reference's own loader included).  `sampler_config()` returns
the merged values of the shipped classification / segmentation presets for the sampler subtree.
"""
from __future__ import annotations

import copy


class AttrDict(dict):
    """dict with attribute access, recursively."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as exc:  # pragma: no cover - mirrors omegaconf's behaviour
            raise AttributeError(name) from exc

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_attr(obj):
    if isinstance(obj, dict):
        return AttrDict({k: to_attr(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_attr(v) for v in obj]
    return obj


def sampler_config(preset: str = "cls", **overrides):
    """Sampler subtree of the shipped presets: 'cls' (ModelNet40, 6 bins) or 'seg'
    (ShapeNet-part, 4 bins); two layers, M = [1024, 512], C = 128, K = 32, dot scoring,
    sparse_col_sqr score, dynamic boundaries, Boltzmann-random selection with T = 0.1.

    Keyword overrides replace per-layer lists or scalars, dotted keys with '__' (e.g.
    `bin__sample_mode=["topk", "topk"]`, `M=[128, 64]`)."""
    if preset not in ("cls", "seg"):
        raise ValueError("preset must be 'cls' or 'seg'")
    nb = 6 if preset == "cls" else 4
    cfg = dict(
        ds_which="token",
        K=32,
        M=[1024, 512],
        asm=["dot", "dot"],
        res=dict(enable=[False, False], ff=[False, False]),
        bin=dict(
            token_orthognonal_loss_factor=0,
            dynamic_boundaries_enable=True,
            bin_boundaries=[[6.065e-06, 3.737e-07, -2.851e-06, -5.421e-06, -8.08e-06][: nb - 1],
                            [5.914e-05, -2.619e-05, -5.652e-05, -7.882e-05, -0.0001078][: nb - 1]],
            num_bins=[nb, nb],
            scaling_factor=[1.0, 1.0],
            sample_mode=["random", "random"],
            norm_mode=["tanh", "tanh"],
            relu_mean_order=["mean_relu", "mean_relu"],
            token_mode=["multi_token", "multi_token"],
            momentum_update_factor=[0.99, 0.99],
            boltzmann_T=[0.1, 0.1],
        ),
        boltzmann=dict(enable=[False, False], boltzmann_T=[1.0, 1.0], norm_mode=["minmax", "minmax"]),
        q_in=[128, 128], q_out=[128, 128], k_in=[128, 128], k_out=[128, 128], v_in=[128, 128], v_out=[128, 128],
        num_heads=[1, 1],
        idx_mode=["sparse_col_sqr", "sparse_col_sqr"],
    )
    for key, value in overrides.items():
        node = cfg
        parts = key.split("__")
        for part in parts[:-1]:
            node = node[part]
        node[parts[-1]] = value
    return to_attr(cfg)

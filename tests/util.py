"""Shared helpers for the test-suite (fixtures, oracle access)."""
import glob
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_DIR = os.path.join(HERE, "golden")


def golden_names():
    """Sampler fixtures (make_golden.py); the layer_* fixtures come from make_golden_layers.py."""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if not n.startswith(("layer_", "block_", "ckpt_", "headline_", "stress_"))]


def layer_fixture(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


class Golden:
    """One committed fixture (tests/golden/<name>.npz, produced by make_golden.py from the reference)."""

    def __init__(self, name):
        self.name = name
        self.d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        (self.B, self.C, self.N, self.M, self.nb, self.K, self.calls, self.seed) = [int(v) for v in self.d["meta"]]
        self.sample_mode = str(self.d["sample_mode"])
        self.idx_mode = str(self.d["idx_mode"])
        self.asm = str(self.d["asm"]) if "asm" in self.d else "dot"
        bt = self.d["boltzmann_T"]
        self.boltzmann_T = str(bt) if bt.dtype.kind in "US" else float(bt)
        self.token_mode = str(self.d["token_mode"]) if "token_mode" in self.d else "multi_token"
        self.relu_mean_order = str(self.d["relu_mean_order"]) if "relu_mean_order" in self.d else "mean_relu"
        self.momentum = float(self.d["momentum"])
        self.dynamic = bool(self.d["dynamic"])
        self.static = [float(v) for v in self.d["static"]]

    def t(self, key, call=0):
        return torch.from_numpy(np.array(self.d[f"c{call}_{key}"]))

    def has(self, key, call=0):
        return f"c{call}_{key}" in self.d

    def weights(self):
        from samble_amd import synth
        return synth.sampler_weights(self.C, self.nb if self.token_mode == "multi_token" else 1, self.seed)

    def x(self, call=0):
        from samble_amd import synth
        return torch.from_numpy(synth.features(self.B, self.C, self.N, self.seed + 10 + call))

    def upstream(self):
        from samble_amd import synth
        return torch.from_numpy(synth.normal((self.B, self.C, self.M), self.seed + 99))

    def spec(self):
        from oracle import torch_oracle as O
        return O.SamplerSpec(M=self.M, K=self.K, C=self.C, num_bins=self.nb, idx_mode=self.idx_mode, asm=self.asm,
                             token_mode=self.token_mode, relu_mean_order=self.relu_mean_order,
                             sample_mode=self.sample_mode, boltzmann_T=self.boltzmann_T,
                             dynamic_boundaries=self.dynamic, momentum=self.momentum,
                             static_boundaries=self.static or None)

    def oracle_state(self):
        from oracle import torch_oracle as O
        return O.SamplerState(*(torch.from_numpy(a.copy()) for a in self.weights()))

    def config(self):
        from samble_amd import sampler_config
        preset = "cls" if self.nb == 6 else "seg"
        cfg = sampler_config(preset, M=[self.M, max(self.M // 2, 1)])
        cfg.bin.sample_mode = [self.sample_mode] * 2
        cfg.idx_mode = [self.idx_mode] * 2
        cfg.asm = [self.asm] * 2
        cfg.bin.token_mode = [self.token_mode] * 2
        cfg.bin.relu_mean_order = [self.relu_mean_order] * 2
        cfg.bin.boltzmann_T = [self.boltzmann_T] * 2
        cfg.bin.momentum_update_factor = [self.momentum] * 2
        for key in ("q_in", "q_out", "k_in", "k_out", "v_in", "v_out"):
            cfg[key] = [self.C] * 2
        if not self.dynamic:
            cfg.bin.dynamic_boundaries_enable = False
            cfg.bin.bin_boundaries = [list(self.static), list(self.static)]
        return cfg

    def module(self, device):
        from samble_amd.downsample import DownSampleToken
        mod = DownSampleToken(self.config(), 0)
        wq, wk, wv, tok = self.weights()
        with torch.no_grad():
            mod.q_conv.weight.copy_(torch.from_numpy(wq))
            mod.k_conv.weight.copy_(torch.from_numpy(wk))
            mod.v_conv.weight.copy_(torch.from_numpy(wv))
            mod.bin_tokens.copy_(torch.from_numpy(tok))
        return mod.to(device)


def set_agreement(a: torch.Tensor, b: torch.Tensor) -> float:
    """mean over rows of |set(a_row) & set(b_row)| / len(row); a, b (..., K) integer tensors."""
    a2 = np.sort(a.reshape(-1, a.shape[-1]).cpu().numpy().astype(np.int64), axis=1)
    b2 = np.sort(b.reshape(-1, b.shape[-1]).cpu().numpy().astype(np.int64), axis=1)
    hits = 0
    for r in range(a2.shape[0]):
        hits += len(np.intersect1d(a2[r], b2[r], assume_unique=False))
    return hits / a2.size


def fill_parameters(module, seed):
    """Deterministic values for every parameter, keyed by its position in named_parameters() order;
    identical to tests/golden/make_golden_block.py (which fills the reference block)."""
    from samble_amd import synth
    with torch.no_grad():
        for i, (name, p) in enumerate(module.named_parameters()):
            n = synth.normal(tuple(p.shape), seed + i).astype(np.float64)
            if name.endswith("bn1.weight") or name.endswith("bn2.weight") or (p.dim() == 1 and "weight" in name):
                v = 1.0 + 0.1 * n
            elif p.dim() == 1:
                v = 0.1 * n
            else:
                fan_in = int(np.prod(p.shape[1:]))
                v = n / np.sqrt(fan_in)
            p.copy_(torch.from_numpy(v.astype(np.float32)))

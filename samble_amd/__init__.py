"""samble_amd: MI355X-native (gfx950) implementation of SAMBLE's attention-score point-cloud
downsampling path (reference stevenczwu/SAMBLE: models/downsample.py, models/attention.py,
utils/ops.py).  Hand-written HIP kernels behind a C ABI (include/samble.h), a thin PyTorch host
layer that keeps the reference's nn.Module / utils.ops interface.  No CPU fallback."""
from . import _lib  # noqa: F401
from .config import AttrDict, sampler_config, to_attr  # noqa: F401

__all__ = ["DownSampleToken", "ops", "sampler_config", "AttrDict", "to_attr"]


def __getattr__(name):
    # lazy: importing the package must work on a box without the built library
    import importlib
    if name == "DownSampleToken":
        return importlib.import_module(".downsample", __name__).DownSampleToken
    if name in ("ops", "downsample", "synth"):
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)

"""s_memtime marks of workgroup 0 inside score_quantiles / bin_plan / bin_select (scratch library built with -DSAMBLE_STAMPS)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import samble_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scratch", "lib_stamps.so")
import torch
from samble_amd import sampler_config, synth
from samble_amd.downsample import DownSampleToken
B, C, N, M, NB = 32, 128, 2048, 1024, 6
dev = "cuda:0"
mod = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0).to(dev)
x = torch.from_numpy(synth.features(B, C, N, 2001)).to(dev)
noise = torch.from_numpy(synth.exp1((B * NB, N), 2002)).to(dev)
for _ in range(3):
    mod(x, noise=noise)
torch.cuda.synchronize()
lib = L.load()
for name, n in (("samble_scratch_chain_stamps", 64), ("samble_scratch_select_stamps", 16)):
    buf = (ctypes.c_ulonglong * n)()
    fn = getattr(lib, name)
    fn.argtypes = [ctypes.c_void_p]
    assert fn(buf) == 0
    v = list(buf)
    print(name)
    prev = None
    for i, t in enumerate(v):
        if t == 0: continue
        print(f"  stamp {i:2d}: {t}  +{(t - prev) if prev else 0} ticks = {((t - prev) / 100.0) if prev else 0:.2f} us")
        prev = t

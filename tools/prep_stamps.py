"""s_memtime marks inside bwd_prep_tri (first and last workgroup of the grid), scratch library with -DSAMBLE_STAMPS."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import samble_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scratch", "lib_rc_stamps.so")
import torch
from samble_amd import sampler_config, synth
from samble_amd.downsample import DownSampleToken
B, C, N, M, NB = 32, 128, 2048, 1024, 6
dev = "cuda:0"
mod = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0).to(dev)
x = torch.from_numpy(synth.features(B, C, N, 2001)).to(dev).requires_grad_(True)
noise = torch.from_numpy(synth.exp1((B * NB, N), 2002)).to(dev)
for _ in range(3):
    (x_ds, idx), _ = mod(x, noise=noise)
    x_ds.sum().backward()
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_ulonglong * 32)()
lib.samble_scratch_prep_stamps.argtypes = [ctypes.c_void_p]
assert lib.samble_scratch_prep_stamps(buf) == 0
v = list(buf)
names = ["start", "dO/O tile in LDS", "Q rows gathered", "images written", "token logits", "token partials"]
t0 = v[0]
for wg, base in (("first", 0), ("last", 16)):
    s = v[base:base + 6]
    print(f"{wg} workgroup: starts at +{s[0] - t0} | " + " ".join(f"{names[i]}:{s[i] - s[i - 1]}" for i in range(1, 6)) + f" | total {s[5] - s[0]}")

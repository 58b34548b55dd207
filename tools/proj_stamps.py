"""s_memtime marks inside proj_dx_tri (workgroup (0,0), every wave, tiles 5 and 6), scratch library with -DSAMBLE_STAMPS."""
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
buf = (ctypes.c_ulonglong * 64)()
lib.samble_scratch_proj_stamps.argtypes = [ctypes.c_void_p]
assert lib.samble_scratch_proj_stamps(buf) == 0
v = list(buf)
t0 = min(x for x in v if x)
for wave in range(8):
    for it in range(2):
        s = v[(wave * 2 + it) * 4:(wave * 2 + it) * 4 + 4]
        print(f"wave {wave} tile {5 + it}: top at +{s[0] - t0}  issue+products {s[1] - s[0]}  vmcnt wait {s[2] - s[1]}  barrier {s[3] - s[2]}")

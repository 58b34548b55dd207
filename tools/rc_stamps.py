"""s_memtime marks inside attn_rows_rc_tri's iteration (workgroup 0, tiles 20 and 21, every wave), from a scratch
library built with -DSAMBLE_STAMPS (tools/build_scratch_libs.sh): where an iteration's cycles go, on the real step."""
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
buf = (ctypes.c_ulonglong * 128)()
lib.samble_scratch_rc_stamps.argtypes = [ctypes.c_void_p]
assert lib.samble_scratch_rc_stamps(buf) == 0
v = list(buf)
names = ["top", "dma issued"] + [f"k-step {i}" for i in range(8)] + ["P map out", "vmcnt wait", "barrier"]
for wave in range(4):
    for it in range(2):
        s = v[(wave * 2 + it) * 16:(wave * 2 + it) * 16 + 13]
        print(f"wave {wave} tile {20 + it}: " + " ".join(f"{names[i]}:{s[i] - s[i - 1]}" for i in range(1, 13)) + f"  | total {s[12] - s[0]}")

buf = (ctypes.c_ulonglong * 128)()
lib.samble_scratch_nl_stamps.argtypes = [ctypes.c_void_p]
assert lib.samble_scratch_nl_stamps(buf) == 0
v = list(buf)
names = ["top", "dma issued", "phase 1", "phase 2", "vmcnt wait", "barrier"]
print("attn_stats_nl_tri (waves 0-3: products then epilogue; waves 4-7: epilogue then products)")
for wave in range(8):
    for it in range(2):
        s = v[(wave * 2 + it) * 8:(wave * 2 + it) * 8 + 6]
        print(f"wave {wave} tile {20 + it}: " + " ".join(f"{names[i]}:{s[i] - s[i - 1]}" for i in range(1, 6)) + f"  | total {s[5] - s[0]}")

if hasattr(lib, "samble_scratch_ka_stamps"):
    buf = (ctypes.c_ulonglong * 128)()
    lib.samble_scratch_ka_stamps.argtypes = [ctypes.c_void_p]
    assert lib.samble_scratch_ka_stamps(buf) == 0
    v = list(buf)
    names = ["top", "dma issued", "map values", "products", "vmcnt wait", "barrier"]
    print("bwd_kacc_tri (last launch = dK), tiles 10 and 11")
    for wave in range(8):
        for it in range(2):
            s = v[(wave * 2 + it) * 8:(wave * 2 + it) * 8 + 6]
            print(f"wave {wave} tile {10 + it}: " + " ".join(f"{names[i]}:{s[i] - s[i - 1]}" for i in range(1, 6)) + f"  | total {s[5] - s[0]}")

if hasattr(lib, "samble_scratch_pc_stamps"):
    buf = (ctypes.c_ulonglong * 128)()
    lib.samble_scratch_pc_stamps.argtypes = [ctypes.c_void_p]
    assert lib.samble_scratch_pc_stamps(buf) == 0
    v = list(buf)
    print("attn_rows_pc_tri: S waves 0-3 (P work | logit products | lgkm | barrier), O waves 4-7 (P V products | DMA issue | vmcnt | barrier)")
    for wave in range(8):
        for it in range(2):
            s = v[(wave * 2 + it) * 8:(wave * 2 + it) * 8 + 5]
            print(f"wave {wave} tile {20 + it}: " + " ".join(str(s[i] - s[i - 1]) for i in range(1, 5)) + f"  | total {s[4] - s[0]}")

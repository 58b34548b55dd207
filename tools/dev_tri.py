"""Dev check of the split-bf16 attention kernels against the fp32-MFMA ones (GPU)."""
import math, sys, time
import torch
sys.path.insert(0, ".")
from samble_amd import ops, _lib

dev = torch.device("cuda:0")
B, N, nt, D = 32, 2048, 6, 128
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B, N + nt, 3 * D, generator=g).to(dev)
q, k, v = qkv[:, :N, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
res = {}
for mode in ("f32", "tri"):
    ops.MATRIX_MODE = mode
    smap, lse, tok = ops.stage_attn_stats(q, k, N, nt)
    torch.cuda.synchronize()
    res[mode] = (smap, lse, tok)
s64 = (q[:2].double() @ k[:2].double().transpose(1, 2)) / math.sqrt(D)
for mode in res:
    d = res[mode][0][:2, :, :N + nt].double() - s64
    print(mode, "rms err", d.pow(2).mean().sqrt().item(), "max", d.abs().max().item(),
          "lse err", (res[mode][1][:2].double() - torch.logsumexp(s64, -1)).abs().max().item())
print("neg-inf padding ok", torch.isneginf(res["tri"][0][:, :, N + nt:]).all().item(),
      "tok==map", torch.equal(res["tri"][2], res["tri"][0][:, :, N:N + nt]))
for mode in ("f32", "tri"):
    ops.MATRIX_MODE = mode
    if mode == "tri":
        imgs = (ops.stage_tri_split(q)[0], ops.stage_tri_split(k)[0])
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            if mode == "tri":
                ops.stage_attn_stats(q, k, N, nt, images=imgs)
            else:
                ops.stage_attn_stats(q, k, N, nt)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(mode, "attn_stats %.1f us" % (dt * 1e6))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    ops.stage_tri_split(q); ops.stage_tri_split(k)
torch.cuda.synchronize(); print("split q+k %.1f us" % ((time.perf_counter() - t0) / 10 * 1e6))
# pass 2
M = 1024
idx = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(bb))[:M] for bb in range(B)]).to(dev)
outs = {}
for mode in ("f32", "tri"):
    ops.MATRIX_MODE = mode
    smap, lse, _ = res[mode]
    vimg = ops.stage_tri_split(v, want_rm=False, want_tr=True)[1] if mode == "tri" else None
    outs[mode] = ops.stage_attn_rows(smap, lse, v, idx, N, nt, v_image=vimg)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ops.stage_attn_rows(smap, lse, v, idx, N, nt, v_image=vimg)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(mode, "attn_rows %.1f us" % (dt * 1e6))
ref = torch.gather(torch.softmax(s64, -1) @ v[:2].double(), 1, idx[:2, :, None].expand(-1, -1, D)).permute(0, 2, 1)
for mode in outs:
    d = outs[mode][:2].double() - ref
    print(mode, "x_ds rms err", d.pow(2).mean().sqrt().item(), "max", d.abs().max().item())

import torch, sys
sys.path.insert(0, '.')
from samble_amd import ops as o_, synth
DEV='cuda:0'
o_.MATRIX_MODE='tri'
B,N,nt,M,K=2,256,6,128,32
q = torch.from_numpy(synth.normal((B, N, 128), 1)); k = torch.from_numpy(synth.normal((B, N + nt, 128), 2)); v = torch.from_numpy(synth.normal((B, N + nt, 128), 3))
qkv = torch.cat((torch.cat((q, torch.zeros(B, nt, 128)), 1) * 0.3, k * 0.3, v), dim=2).to(DEV).contiguous()
qd, kd, vd = qkv[:, :N, :128], qkv[:, :, 128:256], qkv[:, :, 256:]
g = torch.Generator().manual_seed(N)
idx = torch.stack([torch.randperm(N, generator=g)[:M] for _ in range(B)]).to(DEV)
imgs = o_.stage_tri_split_qkv(qkv, N, for_backward=True)
smap, lse, tok = o_.stage_attn_stats(qd, kd, N, nt, images=imgs[:2])
x_ds = o_.stage_attn_rows(smap, lse, vd, idx, N, nt, v_image=imgs[2])
x_ds2, pmap = o_.stage_attn_rows_recompute(imgs[0], imgs[1], imgs[2], lse, idx, N, nt, True)
x_ds3, _ = o_.stage_attn_rows_recompute(imgs[0], imgs[1], imgs[2], lse, idx, N, nt, False)
print('xds2 maxdiff', (x_ds2-x_ds).abs().max().item(), 'xds3', (x_ds3-x_ds).abs().max().item(), 'x2 vs x3', (x_ds2-x_ds3).abs().max().item())
rows = torch.gather(smap, 1, idx[:, :, None].expand(-1, -1, smap.shape[2]))
p_ref = torch.exp(rows - torch.gather(lse, 1, idx)[:, :, None])
d = (pmap - p_ref).abs()
print('pmap maxdiff', d.max().item(), 'rel', (d / p_ref.clamp_min(1e-30)).max().item())
bad = (d / p_ref.clamp_min(1e-30)) > 1e-5
print('bad count', bad.sum().item(), bad.nonzero()[:10].tolist())
# which rows differ in x_ds
dd = (x_ds2 - x_ds).abs().amax(1)
print('rows with diff', (dd > 0).sum().item(), 'of', dd.numel(), (dd>0).nonzero()[:10].tolist())

"""One rank of the 2-rank DistributedDataParallel(DownSampleToken) test (tests/test_gpu_ddp.py starts two of
these as child processes).  Both ranks run on cuda:0 (the GPU box has one device), so the process group is
gloo -- the code path is the one RCCL takes on an 8-GPU node: DDP's bucketed gradient all-reduce plus the
module's own all-reduce of the nb-1 boundary quantiles inside forward (reference utils/ops.py:191-199,
train_modelnet.py:162-166, 245-250).

    RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT in the environment;  argv: OUT_DIR [BACKEND [sampler | edgeconv | block | block_seg]]

`block`: DistributedDataParallel(SyncBatchNorm.convert_sync_batchnorm(FeatureLearningBlock)) -- the reference trainer's
recipe around the whole block (train_modelnet.py:245-250 with configs/default.yaml `syn_bn: true`).  With WORLD_SIZE=1 and
BACKEND=nccl the same code runs over RCCL on the one GPU of the test box (RCCL refuses two ranks on one device).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

B, C, N, M, NB, SEED = 4, 128, 256, 128, 6, 4100


def build_module(device):
    from samble_amd import sampler_config, synth
    from samble_amd.downsample import DownSampleToken
    mod = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0)
    wq, wk, wv, tok = synth.sampler_weights(C, NB, SEED)
    with torch.no_grad():
        mod.q_conv.weight.copy_(torch.from_numpy(wq))
        mod.k_conv.weight.copy_(torch.from_numpy(wk))
        mod.v_conv.weight.copy_(torch.from_numpy(wv))
        mod.bin_tokens.copy_(torch.from_numpy(tok))
    return mod.to(device)


def shard(rank, device):
    from samble_amd import synth
    x = torch.from_numpy(synth.features(B, C, N, SEED + 1, first_cloud=rank * B)).to(device)
    noise = torch.from_numpy(synth.exp1((B * NB, N), SEED + 2 + rank)).to(device)
    g = torch.from_numpy(synth.normal((B, C, M), SEED + 10 + rank)).to(device)
    return x, noise, g


def edgeconv_syncbn(rank, dev, out_dir):
    """Fused EdgeConv under nn.SyncBatchNorm vs the stock torch composition under nn.SyncBatchNorm, same weights,
    this rank's shard: BatchNorm statistics (forward) and gradient sums (backward) must be pooled over the ranks
    (reference train_modelnet.py:245-246 converts every BatchNorm)."""
    import copy
    from samble_amd import synth
    from samble_amd.embedding import EdgeConv, embedding_config
    Bs, Ns = 2, 512
    torch.manual_seed(11)
    fused = EdgeConv(embedding_config("cls"), 1)  # 128 -> 64 -> 64, K = 32, center_diff
    with torch.no_grad():
        for p in fused.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    stock = copy.deepcopy(fused)
    stock.fused = False
    fused = torch.nn.SyncBatchNorm.convert_sync_batchnorm(fused).to(dev).train()
    stock = torch.nn.SyncBatchNorm.convert_sync_batchnorm(stock).to(dev).train()
    fused.fused, stock.fused = True, False
    x = torch.from_numpy(synth.features(Bs, 64, Ns, SEED + 50, first_cloud=rank * Bs)).to(dev)
    x = x * (1.0 + 0.5 * rank) + 0.3 * rank  # the two shards have different statistics
    g = torch.from_numpy(synth.normal((Bs, 64, Ns), SEED + 60 + rank)).to(dev)
    res = {}
    for name, m in (("fused", fused), ("stock", stock)):
        xin = x.detach().requires_grad_(True)
        y = m(xin)
        y.backward(g)
        torch.cuda.synchronize()
        res[name] = {"y": y.detach().cpu(), "dx": xin.grad.cpu(),
                     "grads": {n: p.grad.detach().cpu() for n, p in m.named_parameters()},
                     "bufs": {n: b.detach().cpu() for n, b in m.named_buffers()}}
    torch.save(res, os.path.join(out_dir, f"edge{rank}.pt"))


BLK_B, BLK_N, BLK_M, BLK_SEED = 2, 256, (128, 64), 6100


def build_block(kind="cls"):
    """The classification (or segmentation) feature-learning block at a small size with seeded parameters (CPU)."""
    from samble_amd.blocks import FeatureLearningBlock, SegFeatureLearningBlock, block_config, seg_block_config
    from tests.util import fill_parameters
    blk = (FeatureLearningBlock(block_config("cls", M=BLK_M)) if kind == "cls"
           else SegFeatureLearningBlock(seg_block_config(M=BLK_M)))
    fill_parameters(blk, BLK_SEED)
    return blk


def block_shard(rank, device, kind="cls"):
    """xyz clouds, the two samplers' Exp(1) noise and the upstream gradient of this rank's shard."""
    from samble_amd import synth
    nb = 6 if kind == "cls" else 4
    xyz = torch.from_numpy(synth.xyz_clouds(BLK_B, BLK_N, BLK_SEED + 1, first_cloud=rank * BLK_B)).to(device)
    noise = [torch.from_numpy(synth.exp1((BLK_B * nb, n), BLK_SEED + 2 + 10 * rank + i)).to(device)
             for i, n in enumerate((BLK_N, BLK_M[0]))]
    shape = (BLK_B, 3 * 1024) if kind == "cls" else (BLK_B, 128, BLK_N)
    g = torch.from_numpy(synth.normal(shape, BLK_SEED + 20 + rank)).to(device)
    return xyz, noise, g


def block_step(model, blk, xyz, noise, g, kind="cls", forced=None):
    """One forward + backward of the block; what a comparison needs, on the CPU."""
    blk.zero_grad(set_to_none=True)
    xin = xyz.detach().requires_grad_(True)
    out = model(xin, noise_list=noise, forced_idx_list=forced)
    y = out[0] if kind == "cls" else out
    y.backward(g)
    torch.cuda.synchronize()
    return {
        "y": y.detach().cpu(), "dx": xin.grad.cpu(),
        "grads": {n: p.grad.detach().cpu().clone() for n, p in blk.named_parameters()},
        "bufs": {n: b.detach().cpu().clone() for n, b in blk.named_buffers()},
        "idx": [layer.idx.cpu().clone() for layer in blk.downsample_list],
        "bounds": [[t.detach().cpu().clone() for t in layer.bin_boundaries] for layer in blk.downsample_list],
    }


def block_syncbn(rank, dev, out_dir, kind="cls"):
    """BASELINE configs[3]'s recipe around the block (reference train_modelnet.py:245-250):
    DistributedDataParallel(SyncBatchNorm.convert_sync_batchnorm(block)) on this rank's shard, two steps."""
    blk = torch.nn.SyncBatchNorm.convert_sync_batchnorm(build_block(kind)).to(dev).train()
    ddp = torch.nn.parallel.DistributedDataParallel(blk, device_ids=[dev.index])
    xyz, noise, g = block_shard(rank, dev, kind)
    log = [block_step(ddp, blk, xyz, noise, g, kind) for _ in range(2)]
    kinds = sorted({type(m).__name__ for m in blk.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)})
    torch.save({"log": log, "backend": dist.get_backend(), "world": dist.get_world_size(), "bn_types": kinds},
               os.path.join(out_dir, f"block{rank}.pt"))


def main():
    out_dir = sys.argv[1]
    backend = sys.argv[2] if len(sys.argv) > 2 else "gloo"
    what = sys.argv[3] if len(sys.argv) > 3 else "sampler"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    if what == "edgeconv":
        edgeconv_syncbn(rank, dev, out_dir)
        dist.barrier()
        dist.destroy_process_group()
        return
    if what in ("block", "block_seg"):
        block_syncbn(rank, dev, out_dir, "seg" if what == "block_seg" else "cls")
        dist.barrier()
        dist.destroy_process_group()
        return
    from samble_amd import ops
    mod = build_module(dev)
    ddp = torch.nn.parallel.DistributedDataParallel(mod, device_ids=[dev.index])
    x, noise, g = shard(rank, dev)
    log = []
    for call in range(2):  # call 0 initialises the boundaries from the rank-averaged quantiles, call 1 blends
        mod.zero_grad(set_to_none=True)
        xin = x.detach().requires_grad_(True)
        (x_ds, idx), _ = ddp(xin, noise=noise)
        x_ds.backward(g)
        torch.cuda.synchronize()
        local_q = ops.stage_batch_quantiles(mod.normalized_score, NB)
        log.append({
            "upper": mod.bin_boundaries[0].detach().cpu().clone(), "lower": mod.bin_boundaries[1].detach().cpu().clone(),
            "local_q": local_q.cpu(), "idx": idx.cpu(), "x_ds": x_ds.detach().cpu(), "dx": xin.grad.cpu(),
            "grads": {n: p.grad.detach().cpu().clone() for n, p in mod.named_parameters()},
        })
    # the evaluation script's collection of what the layer published (test_modelnet.py:236-297, utils/ops.py:289-384):
    # every rank calls, rank 0 receives
    mod.output_variable_calculatio()
    names = ("attention_point_score", "idx", "bin_prob", "idx_chunks", "k_point_to_choose")
    own = {n: mod.output_variables(n) for n in names}
    gathered = {n: ops.gather_variable_from_gpus(mod, n, rank, world, dev) for n in names}

    def cpu(v):
        return v.cpu() if isinstance(v, torch.Tensor) else (None if v is None else [cpu(u) for u in v])
    torch.save({"log": log, "backend": dist.get_backend(), "world": dist.get_world_size(),
                "published": {n: cpu(v) for n, v in own.items()}, "gathered": {n: cpu(v) for n, v in gathered.items()}},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: point-clouds/sec of ONE DownSampleToken layer, forward + backward
(+ SGD step), on the metric configuration of BASELINE.json / SURVEY.md section 8(d):
ModelNet40-shaped synthetic input, B=32 clouds per GPU, C=128, N=2048 -> M=1024, 6 bin tokens,
K=32 feature-space kNN, sparse_col_sqr score, dynamic boundaries, Boltzmann-random selection.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; batches shard over ranks (weak scaling, 32 clouds per rank); the only
data-path collective is the reference's own all-reduce of the nb-1 boundary quantiles, plus DDP's
gradient all-reduce (RCCL over xGMI).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
# split-bf16 kernels (samble_amd/csrc/tri_dev.h): every fp32 product is 6 bf16 MFMA products, so the
# ceiling for ALGORITHMIC (fp32) flops is the dense bf16 peak / 6
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_TRI_TFLOPS = round(PEAK_BF16_MFMA_TFLOPS / 6, 1)
PEAK_HBM_GBS = 8000.0

B_PER_GPU, C, N, M, NB, KNN = 32, 128, 2048, 1024, 6, 32


def algorithmic_flops_per_cloud():
    """proj 2C^2(3N+2nb), dist 2N^2C, qk 2N(N+nb)D, av 2M(N+nb)D forward;
    backward 4*av + 2*proj.  Recomputation is not counted."""
    proj = 2 * C * C * (3 * N + 2 * NB)
    dist_ = 2 * N * N * C
    qk = 2 * N * (N + NB) * C
    av = 2 * M * (N + NB) * C
    return dict(proj=proj, dist=dist_, qk=qk, av=av, fwd=proj + dist_ + qk + av, bwd=4 * av + 2 * proj)


def time_region(fn, iters):
    """Average milliseconds per call of fn(), HIP events on torch's current stream (the stream
    every kernel of the path is enqueued on)."""
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / iters


# ids of the library's timing hook (samble_debug_time_kernel); both matrix modes use the same ids for the
# kernels that play the same part (3 = dominant backward kernel: bwd_rows or bwd_dkdv_tri)
KERNEL_IDS = {"attn_stats": 1, "attn_rows": 2, "bwd_rows": 3, "knn_stream": 4, "bwd_dq": 6, "bwd_dk": 7}


def kernel_ms(kernel, fn, iters=5):
    """Mean duration (ms) of one named kernel inside fn(), from the library's own HIP events."""
    from samble_amd import _lib
    lib = _lib.load()
    fn()
    lib.samble_debug_time_kernel(KERNEL_IDS[kernel])
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    ms = float(lib.samble_debug_kernel_ms())
    lib.samble_debug_time_kernel(0)
    return ms


def kernel_breakdown(mod, x, noise, g, iters=5):
    """Per-stage device time (ms) of one step, each stage timed alone with events."""
    import math
    from samble_amd import ops
    with torch.no_grad():
        B = x.shape[0]
        nt = mod.bin_tokens.shape[2]
        w = torch.cat((mod.q_conv.weight, mod.k_conv.weight, mod.v_conv.weight), 0).squeeze(-1)
        tokm = mod.bin_tokens[0]
        qkv = ops.stage_proj_fwd(x, tokm, w)
        q, k, v = qkv[:, :N, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
        out = {}
        out["proj_fwd"] = time_region(lambda: ops.stage_proj_fwd(x, tokm, w), iters)
        out["knn"] = time_region(lambda: ops.stage_knn(x, x, KNN), iters)
        nn_idx = ops.stage_knn(x, x, KNN)
        imgs = None
        if ops.MATRIX_MODE == "tri":  # the module splits Q, K, V into operand images once per step
            out["split_qkv"] = time_region(lambda: ops.stage_tri_split_qkv(qkv, N, for_backward=True), iters)
            imgs = ops.stage_tri_split_qkv(qkv, N, for_backward=True)
        out["attn_stats"] = time_region(lambda: ops.stage_attn_stats(q, k, N, nt, images=imgs[:2] if imgs else None), iters)
        smap, lse, tok = ops.stage_attn_stats(q, k, N, nt, images=imgs[:2] if imgs else None)
        out["sparse_score"] = time_region(lambda: ops.stage_sparse_score_map(smap, lse, nn_idx, "sparse_col_sqr"), iters)
        score, z, _ = ops.stage_sparse_score_map(smap, lse, nn_idx, "sparse_col_sqr")
        out["batch_quantiles"] = time_region(lambda: ops.stage_batch_quantiles(z, NB), iters)
        up, lo = mod.bin_boundaries
        out["bin_assign"] = time_region(lambda: ops.stage_bin_assign(z, tok, up, lo, False), iters)
        member, cap, w_pre, wts = ops.stage_bin_assign(z, tok, up, lo, False)
        out["alloc_counts"] = time_region(lambda: ops.stage_alloc_counts(wts, cap, M), iters)
        counts = ops.stage_alloc_counts(wts, cap, M)
        out["bin_select"] = time_region(
            lambda: ops.stage_bin_select(score, z, member, counts, M, "random", 0.1, noise), iters)
        idx = ops.stage_bin_select(score, z, member, counts, M, "random", 0.1, noise)
        vimg = imgs[2] if imgs else None
        out["attn_rows"] = time_region(lambda: ops.stage_attn_rows(smap, lse, v, idx, N, nt, v_image=vimg), iters)
        x_ds = ops.stage_attn_rows(smap, lse, v, idx, N, nt, v_image=vimg)
        dqkv = torch.empty_like(qkv)
        out["attn_bwd"] = time_region(
            lambda: ops.stage_attn_rows_bwd(q, k, v, smap, lse, x_ds, idx, g, N, nt, dqkv[:, :N, :C],
                                            dqkv[:, :, C:2 * C], dqkv[:, :, 2 * C:],
                                            images=imgs[3:] if imgs else None), iters)
        out["proj_bwd"] = time_region(lambda: ops.stage_proj_bwd(dqkv, x, tokm, w, True, True), iters)
    return out


def cpu_baseline(seed):
    """The CPU oracle (pure-torch restatement, bit-identical to the reference in the build
    container) timed on this host's cores on a bounded sample of the same workload."""
    from oracle import torch_oracle as O
    from samble_amd import synth
    spec = O.SamplerSpec(M=M, K=KNN, C=C, num_bins=NB)
    wq, wk, wv, tok = synth.sampler_weights(C, NB, seed)
    st = O.SamplerState(*(torch.from_numpy(a) for a in (wq, wk, wv, tok)))
    sample_b = 8
    x = torch.from_numpy(synth.features(sample_b, C, N, seed + 1))
    g = torch.from_numpy(synth.normal((sample_b, C, M), seed + 2))
    noise = torch.from_numpy(synth.exp1((sample_b * NB, N), seed + 3))
    # give the CPU path its best thread count (all cores is slower than fewer on many-core hosts)
    ncpu = os.cpu_count() or 1
    best = None
    for threads in sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu}):
        torch.set_num_threads(threads)
        O.sampler_grads(spec, st, x[:1], g[:1], noise[:NB])  # warm-up
        t0 = time.perf_counter()
        O.sampler_grads(spec, st, x[:2], g[:2], noise[: 2 * NB])
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, threads)
    cores = best[1]
    torch.set_num_threads(cores)
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        O.sampler_grads(spec, st, x, g, noise)
    dt = (time.perf_counter() - t0) / reps
    return dict(value=round(sample_b / dt, 3), unit="clouds/s", cores=cores, kind="port",
                sample=f"{reps} x fwd+bwd of {sample_b} clouds (N={N}->{M}), torch CPU oracle, {cores} threads")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="metric", choices=["metric", "stress"],
                    help="metric = BASELINE.json's B=32 N=2048->1024 (default); stress = configs[4]: B=16 N=8192->4096")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) or gloo (single-GPU smoke test of the N>1 path)")
    args = ap.parse_args()

    global B_PER_GPU, N, M
    if args.workload == "stress":
        B_PER_GPU, N, M = 16, 8192, 4096
        args.no_cpu_baseline = True  # the CPU oracle needs minutes per cloud at this size
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    local = local % max(torch.cuda.device_count(), 1)  # (gloo smoke test: ranks may share one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend)

    from samble_amd import sampler_config, synth
    from samble_amd.downsample import DownSampleToken

    seed = 1000 * 2
    mod = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0)
    wq, wk, wv, tok = synth.sampler_weights(C, NB, seed)
    with torch.no_grad():
        mod.q_conv.weight.copy_(torch.from_numpy(wq))
        mod.k_conv.weight.copy_(torch.from_numpy(wk))
        mod.v_conv.weight.copy_(torch.from_numpy(wv))
        mod.bin_tokens.copy_(torch.from_numpy(tok))
    mod = mod.to(dev)
    model = mod
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(mod, device_ids=[local])
    opt = torch.optim.SGD(mod.parameters(), lr=1e-4)

    # this rank's shard of the global batch (cloud ids rank*32 .. rank*32+31), resident in HBM
    x = torch.from_numpy(synth.features(B_PER_GPU, C, N, seed + 1, first_cloud=rank * B_PER_GPU)).to(dev)
    g = torch.from_numpy(synth.normal((B_PER_GPU, C, M), seed + 2 + rank)).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        xin = x.detach().requires_grad_(True)  # the layer sits mid-network: dL/dx is part of the work
        (x_ds, idx), _ = model(xin)
        x_ds.backward(g)
        opt.step()

    for _ in range(args.warmup):
        step()
    # HIP events around every launch of the dominant kernel (bwd_rows) during the timed steps, recorded
    # by the library on the stream it launches on (two event records per step: no host sync, no effect
    # on the timed region); read back after the final synchronize
    from samble_amd import _lib
    lib = _lib.load()
    from samble_amd import ops as _ops0
    dominant = "knn_stream" if _ops0.MATRIX_MODE == "tri" else "bwd_rows"  # id 4 is knn_tri in the split-bf16 mode
    lib.samble_debug_time_kernel(KERNEL_IDS[dominant])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    dominant_ms = float(lib.samble_debug_kernel_ms())
    lib.samble_debug_time_kernel(0)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        total_clouds = B_PER_GPU * world * args.steps
        value = total_clouds / elapsed
        fl = algorithmic_flops_per_cloud()
        result = {
            "metric": ("point-clouds/sec (downsample fwd+bwd), ModelNet40 B=32 N=2048->1024" if args.workload == "metric"
                       else "point-clouds/sec (downsample fwd+bwd), synthetic dense clouds B=16 N=8192->4096"),
            "value": round(value, 2),
            "unit": "clouds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (hash-generated N(0,1) features, random-init weights; no ModelNet40 files offline)",
            "config": {"workload": f"one DownSampleToken layer fwd+bwd+SGD, cls layer 0: B={B_PER_GPU}/GPU C=128 N={N}->M={M} "
                                   "nb=6 K=32 sparse_col_sqr random T=0.1 dynamic boundaries",
                       "global_batch": B_PER_GPU * world, "parallelism": f"dp{world}", "backend": args.backend if world > 1 else None},
            "step_fraction_of_mfma_roofline": round(
                (fl["fwd"] + fl["bwd"]) * B_PER_GPU / (ms_per_step * 1e-3) / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
        }
        # dominant kernel: bwd_rows (attention backward over the N point keys: dP, dV, dK, dQ = 4 products of
        # 2*M*N*D flop per cloud); duration = mean over the timed steps' launches, HIP events on its stream
        def pmc_traffic(kernel):
            # fabric-side bytes per launch from the newest committed rocprofv3 --pmc summary
            # (profiles/*_pmc.json: (2*FETCH_SIZE + WRITE_SIZE)*1024, separate passes as the counters require)
            try:
                import glob
                pmc = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))[-1]))["kernels"]
                key = next(k for k in pmc if k == kernel or k.startswith(kernel + "<"))  # template arguments vary
                return pmc[key]["traffic_bytes_per_launch"]
            except Exception:
                return None

        from samble_amd import ops as _ops
        tri = _ops.MATRIX_MODE == "tri"
        peak = PEAK_TRI_TFLOPS if tri else PEAK_FP32_MFMA_TFLOPS

        def roof(kernel, alg_flops, ms, pmc_name):
            ach = alg_flops / (ms * 1e-3) / 1e12
            out = {"kernel": kernel, "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                   "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": pmc_traffic(pmc_name),
                   "us_per_launch": round(ms * 1e3, 1), "algorithmic_flops_per_launch": alg_flops}
            if tri:
                out["peak_note"] = ("fp32 products as 6 bf16 MFMA products (3 bf16 planes per operand, fp32 accumulate): "
                                    "peak = dense bf16 2500 TFLOP/s / 6; executed bf16 flop = 6 x algorithmic")
                out["frac_of_fp32_mfma_peak"] = round(ach / PEAK_FP32_MFMA_TFLOPS, 4)
                # register-resident loop of the same 6-product scheme on random operands (tools/micro/split_mfma_bench.hip):
                # 1771 TFLOP/s executed at the 1.69 GHz the chip holds under that load
                out["frac_of_sustained_scheme_rate_295"] = round(ach / 295.0, 4)
            return out

        if tri:
            # dominant kernel: knn_tri (fused Gram + top-K of the feature-space kNN; algorithmic flops = the Gram)
            result["roofline"] = roof("knn_tri_kernel", fl["dist"] * B_PER_GPU, dominant_ms, "samble::knn_tri_kernel")
        else:
            bwd_alg = 4 * 2 * M * N * C * B_PER_GPU
            result["roofline"] = roof("bwd_rows_kernel", bwd_alg, dominant_ms, "samble::bwd_rows_kernel")
        result["matrix_mode"] = _ops.MATRIX_MODE
        if not args.no_breakdown:
            noise = torch.from_numpy(synth.exp1((B_PER_GPU * NB, N), seed + 3)).to(dev)
            br = kernel_breakdown(mod, x, noise, g)
            result["stage_ms"] = {k: round(v, 4) for k, v in br.items()}
            # the other MFMA kernels, each timed the same way over 5 more full steps (same cache state as the timed region)
            def in_step_ms(kernel):
                lib.samble_debug_time_kernel(KERNEL_IDS[kernel])
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
                ms = float(lib.samble_debug_kernel_ms())
                lib.samble_debug_time_kernel(0)
                return ms
            sfx = "_tri_kernel" if tri else "_kernel"
            others = {
                "attn_stats" + sfx: (fl["qk"] * B_PER_GPU, in_step_ms("attn_stats"), "samble::attn_stats" + sfx),
                "attn_rows" + sfx: (fl["av"] * B_PER_GPU, in_step_ms("attn_rows"), "samble::attn_rows" + sfx),
            }
            if tri:
                # backward: dQ kernel = dP + dQ (2 products over N + nt keys), then dV and dK (1 product each, N keys)
                others["bwd_dq_tri_kernel"] = (2 * fl["av"] * B_PER_GPU, in_step_ms("bwd_dq"), "samble::bwd_dq_tri_kernel")
                others["bwd_kacc_tri_kernel<0> (dV)"] = (2 * M * N * C * B_PER_GPU, in_step_ms("bwd_rows"),
                                                        "samble::bwd_kacc_tri_kernel<0, false>")
                others["bwd_kacc_tri_kernel<1> (dK)"] = (2 * M * N * C * B_PER_GPU, in_step_ms("bwd_dk"),
                                                        "samble::bwd_kacc_tri_kernel<1, false>")
            else:
                others["knn_stream_kernel"] = (fl["dist"] * B_PER_GPU, in_step_ms("knn_stream"), "samble::knn_stream_kernel")
            result["roofline_other_kernels"] = [roof(kk, a_, ms_, pn) for kk, (a_, ms_, pn) in others.items()]
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(seed)
            result["gpu_over_cpu"] = round(value / result["cpu_baseline"]["value"], 2)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

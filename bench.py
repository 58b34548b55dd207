#!/usr/bin/env python3
"""Headline benchmark: point-clouds/sec of ONE DownSampleToken layer, forward + backward
(+ SGD step), on the metric configuration of BASELINE.json / SURVEY.md section 8(d):
ModelNet40-shaped synthetic input, B=32 clouds per GPU, C=128, N=2048 -> M=1024, 6 bin tokens,
K=32 feature-space kNN, sparse_col_sqr score, dynamic boundaries, Boltzmann-random selection.

    python bench.py --gpus N --steps K --warmup W          (starts its own N ranks when N > 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; batches shard over ranks (weak scaling, 32 clouds per rank); the only
data-path collective is the reference's own all-reduce of the nb-1 boundary quantiles, plus DDP's
gradient all-reduce (RCCL over xGMI).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
# split-plane kernels (samble_amd/csrc/tri_dev.h): an fp32 product is formed from 16-bit MFMA products of operand planes
# -- six with three bf16 planes per operand, three with two fp16 planes -- so the ceiling for ALGORITHMIC (fp32) flops of
# a kernel is the dense 16-bit peak (bf16 = fp16 = 2500 TFLOP/s) / the 16-bit products it EXECUTES per algorithmic product
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
# 16-bit MFMA products executed per ALGORITHMIC fp32 product, per kernel of the default (split-plane, map-free) step
EXECUTED_PRODUCTS = {
    "knn": (4.0, "two fp16 planes: 1 seed product (h h) over all key tiles + 3 exact products (hh + hm + mh)"),
    "attn_stats": (3.0, "logits on two fp16 planes under per-tile / per-row power-of-two scales: 3 products"),
    "attn_rows": (9.0, "algorithmic work = P V (SURVEY 8d: recomputation is not counted); executed = the sampled rows' "
                       "logits again (3 fp16 products) + P V on three bf16 planes (6)"),
    "bwd_dq": (4.5, "two algorithmic products (dP, dQ): dP on two fp16 planes (3) + dQ += dS K on three bf16 planes (6) "
                    "= 9 executed per 2"),
    "bwd_dv": (3.0, "dV += dO^T P on two fp16 planes: 3 products"),
    "bwd_dk": (3.0, "dK += Q^T dS on two fp16 planes: 3 products"),
    "proj_fwd": (6.0, "three bf16 planes per operand: 6 products"),
    "proj_dx": (6.0, "three bf16 planes per operand: 6 products"),
    "proj_dw": (6.0, "three bf16 planes per operand: 6 products"),
}

B_PER_GPU, C, N, M, NB, KNN = 32, 128, 2048, 1024, 6, 32


def algorithmic_flops_per_cloud():
    """proj 2C^2(3N+2nb), dist 2N^2C, qk 2N(N+nb)D, av 2M(N+nb)D forward;
    backward 4*av + 2*proj.  Recomputation is not counted."""
    proj = 2 * C * C * (3 * N + 2 * NB)
    dist_ = 2 * N * N * C
    qk = 2 * N * (N + NB) * C
    av = 2 * M * (N + NB) * C
    return dict(proj=proj, dist=dist_, qk=qk, av=av, fwd=proj + dist_ + qk + av, bwd=4 * av + 2 * proj)


def algorithmic_bytes_per_cloud():
    """SURVEY.md section 8(d) per-kernel algorithmic HBM bytes (fp32 inputs read once, outputs written once)."""
    return dict(
        knn=4 * N * C + 4 * N * KNN,                          # points in, neighbour lists out
        select=4 * N + 4 * N * NB + 4 * NB + 8 * M,           # score, Exp(1) noise, boundaries in; idx out
        sparse_score=8 * N * KNN + 4 * N + 12 * N,            # neighbour ids + one logit each, lse in; score, z, in-degree out
                                                              # (map-free forward: folded into attn_stats_nl_tri, no launch)
        knn_prep=4 * N * C + 4 * N * C + 4 * N,               # points in; operand image (two fp16 planes: 4 B / element) + norms out
        nn_prepare=4 * N * KNN + 4 * N * KNN + 4 * N * ((N + 31) // 32),  # lists in; sorted lists + one word per (tile, row) out
        tri_split=4 * 3 * C * (N + NB) + 5 * 6 * C * (N + NB),  # [Q|K|V] rows in; five operand images out
        bwd_prep=8 * M + 8 * M * C + 3 * 6 * M * C,           # idx, Q rows, dO in; three operand images of the sampled rows out
        fused_step=4 * N * C + 4 * M * C + 8 * M + 4 * M * C + 8 * N * C,  # ideal fused layer: x, x_ds, idx; g, dx(+x)
    )


# ------------------------------------------------------------------------------------------------
# N > 1 without torch.distributed.run: start the ranks ourselves, BEFORE anything touches the GPU
# ------------------------------------------------------------------------------------------------
EADDRINUSE_EXIT = 98  # a rank that loses the race for MASTER_PORT exits with this code: the parent retries


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _visible_gpus() -> int:
    """GPUs this process may use, WITHOUT bringing up the HIP runtime (torch.cuda.device_count() goes through
    hipGetDeviceCount on ROCm): the visibility variables, else the KFD topology."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    n = 0
    try:
        for node in os.listdir("/sys/class/kfd/kfd/topology/nodes"):
            props = open(f"/sys/class/kfd/kfd/topology/nodes/{node}/properties").read().split()
            if "simd_count" in props and int(props[props.index("simd_count") + 1]) > 0:
                n += 1
    except OSError:
        n = torch.cuda.device_count()
    return n


def launch_ranks(n: int, argv) -> int:
    """Parent of `python bench.py --gpus N` (N > 1, no RANK in the environment): one child process per
    rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would.  The ranks are fresh child
    processes (never an exec of this one); the parent counts the GPUs without initialising the runtime, forwards
    SIGINT / SIGTERM to the ranks and reaps them on every way out; rank 0's stdout is the JSON line."""
    import signal
    ndev = _visible_gpus()
    if "--backend" not in " ".join(argv) and ndev < n:
        # RCCL refuses two ranks on one device: on a box with fewer GPUs than ranks the ranks share GPUs over gloo
        # (a functional check of the N > 1 path, flagged in the JSON line; never a scaling number)
        argv = list(argv) + ["--backend", "gloo"]
    procs = []

    def stop(*_):
        for p_ in procs:
            if p_.poll() is None:
                p_.terminate()

    old = {sig: signal.signal(sig, lambda *_: (stop(), sys.exit(130))) for sig in (signal.SIGINT, signal.SIGTERM)}
    rc = 0
    try:
        for attempt in range(3):  # (the port is free when probed, not necessarily when rank 0 binds it)
            env0 = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                        HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs[:] = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv),
                                         env=dict(env0, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]
            rc = 0
            alive = set(range(n))
            while alive:
                for r in list(alive):
                    code = procs[r].poll()
                    if code is None:
                        continue
                    alive.discard(r)
                    if code != 0:
                        rc = rc or code
                        for o in alive:  # a dead rank would leave the others waiting at the rendezvous / next collective
                            procs[o].terminate()
                time.sleep(0.05)
            if rc != EADDRINUSE_EXIT:
                break
    finally:
        stop()
        for p_ in procs:
            try:
                p_.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p_.kill()
        for sig, h in old.items():
            signal.signal(sig, h)
    return rc


def measure_collectives(dev, world, c2_floats=None, c2_label="c2_ddp_bucket_399KB_allreduce_us"):
    """After the timed region, every rank: which library carried the collectives, how many ranks it formed, and the
    latency of the two messages on the data path -- C1, the all-reduce of the nb-1 boundary quantiles that sits between
    score_quantiles and bin_plan in every forward (reference utils/ops.py:191-199), and C2, one DDP bucket with the
    sampler layer's 99 840 parameters (reference train_modelnet.py:245-250) -- so that a scaling curve explains itself."""
    backend = dist.get_backend()
    one = torch.ones(1, device=dev)
    dist.all_reduce(one)
    torch.cuda.synchronize()
    out = {"backend": backend, "ranks_formed": int(round(float(one.item()))), "world_size": dist.get_world_size()}
    try:
        v = torch.cuda.nccl.version()
        out["rccl_version"] = ".".join(str(i) for i in v) if isinstance(v, tuple) else str(v)
    except Exception as e:  # noqa: BLE001
        out["rccl_version"] = f"unavailable ({e!r})"
    # (C1 carries nb floats since round 5: the nb-1 quantiles and the validity count bin_plan divides by)
    c2_floats = c2_floats or 3 * C * C + C * NB
    for label, n in (("c1_boundary_allreduce_5_floats_us", NB), (c2_label, c2_floats)):
        buf = torch.zeros(n, device=dev)
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        dist.barrier()
        reps = 20
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for i in range(reps):
            dist.all_reduce(buf)
            ev[i + 1].record()
        torch.cuda.synchronize()
        out[label] = round(1e3 * statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(reps)), 1)
    worst = torch.tensor([out["c1_boundary_allreduce_5_floats_us"], out[c2_label]], device=dev)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    out["c1_boundary_allreduce_5_floats_us"], out[c2_label] = [round(float(v), 1) for v in worst]
    out["note"] = ("median of 20 back-to-back all-reduces timed by HIP events on the compute stream, max over ranks; C1 is "
                   "on the forward's critical path once per layer, C2 overlaps the backward under DDP")
    return out


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seed):
    """The CPU oracle (pure-torch restatement, bit-identical to the reference in the build
    container) timed on this host's cores on a bounded sample of the same workload."""
    from oracle import torch_oracle as O
    from samble_amd import synth
    spec = O.SamplerSpec(M=M, K=KNN, C=C, num_bins=NB)
    wq, wk, wv, tok = synth.sampler_weights(C, NB, seed)
    st = O.SamplerState(*(torch.from_numpy(a) for a in (wq, wk, wv, tok)))
    sample_b = B_PER_GPU
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
    reps = 3
    O.sampler_grads(spec, st, x, g, noise)  # warm-up at the full batch (first-touch of the 2 x 538 MB maps)
    times = []
    ref = None
    for _ in range(reps):
        t0 = time.perf_counter()
        ref = O.sampler_grads(spec, st, x, g, noise)
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    line = dict(value=round(sample_b / dt, 3), unit="clouds/s", cores=cores, kind="port", cpu=cpu_model(),
                host_cpus=ncpu,
                sample=f"median of {reps} x fwd+bwd of {sample_b} clouds (N={N}->{M}) after 1 warm-up, torch CPU oracle "
                       f"(bit-identical restatement of the reference), {cores} threads")
    return line, dict(idx=ref["idx"], x_ds=ref["x_ds"], dx=ref["dx"], noise=noise, g=g)


def parity_vs_oracle(seed, dev, ref):
    """One UNTIMED step of a fresh layer (the initial weights, the same 32 clouds, the oracle's Exp(1) noise injected)
    against what the oracle produced in cpu_baseline(): the end-to-end sampled-index match at the headline
    configuration (reference utils/ops.py:467-619; SURVEY section 7: stage-wise exactness is what the tests assert,
    this is the reported end-to-end rate)."""
    from samble_amd import sampler_config, synth
    from samble_amd.downsample import DownSampleToken
    mod = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0)
    wq, wk, wv, tok = synth.sampler_weights(C, NB, seed)
    with torch.no_grad():
        mod.q_conv.weight.copy_(torch.from_numpy(wq))
        mod.k_conv.weight.copy_(torch.from_numpy(wk))
        mod.v_conv.weight.copy_(torch.from_numpy(wv))
        mod.bin_tokens.copy_(torch.from_numpy(tok))
    mod = mod.to(dev)
    x = torch.from_numpy(synth.features(B_PER_GPU, C, N, seed + 1)).to(dev).requires_grad_(True)
    (x_ds, idx), _ = mod(x, noise=ref["noise"].to(dev))
    x_ds.backward(ref["g"].to(dev))
    torch.cuda.synchronize()
    a, b = idx.cpu()[:, 0], ref["idx"][:, 0]
    same = (a == b).all(1)
    inter = sum(len(set(a[i].tolist()) & set(b[i].tolist())) for i in range(a.shape[0]))
    out = {"clouds_identical": int(same.sum()), "of": int(a.shape[0]),
           "positions_identical": round(float((a == b).float().mean()), 6),
           "set_agreement": round(inter / a.numel(), 6),
           "max_points_differing_in_a_cloud": max(len(set(a[i].tolist()) ^ set(b[i].tolist())) // 2 for i in range(a.shape[0])),
           "against": "the CPU oracle's fwd+bwd of the same 32 clouds in cpu_baseline (same weights, same Exp(1) noise)"}
    if bool(same.any()):
        xr, dr = ref["x_ds"][same], ref["dx"][same]
        out["x_ds_max_err_on_identical_clouds"] = float((x_ds.detach().cpu()[same] - xr).abs().max())
        out["dx_max_rel_err_on_identical_clouds"] = float((x.grad.cpu()[same] - dr).abs().max() / dr.abs().max())
    return out


def quick_sampler(Bq, Nq, Mq, steps, warmup, label="BASELINE configs[4]"):
    """ms/step and the longest kernel of one DownSampleToken layer fwd+bwd+SGD at another geometry (the stress workload
    behind the headline line; `--workload stress` prints its full line)."""
    from samble_amd import _lib, sampler_config, synth
    from samble_amd.downsample import DownSampleToken
    dev = torch.device("cuda", torch.cuda.current_device())
    seed = 1000 * 5
    mod = DownSampleToken(sampler_config("cls", M=[Mq, Mq // 2]), 0)
    wq, wk, wv, tok = synth.sampler_weights(C, NB, seed)
    with torch.no_grad():
        mod.q_conv.weight.copy_(torch.from_numpy(wq))
        mod.k_conv.weight.copy_(torch.from_numpy(wk))
        mod.v_conv.weight.copy_(torch.from_numpy(wv))
        mod.bin_tokens.copy_(torch.from_numpy(tok))
    mod = mod.to(dev)
    opt = torch.optim.SGD(mod.parameters(), lr=1e-4)
    x = torch.from_numpy(synth.features(Bq, C, Nq, seed + 1)).to(dev)
    g = torch.from_numpy(synth.normal((Bq, C, Mq), seed + 2)).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        (x_ds, _), _ = mod(x.detach().requires_grad_(True))
        x_ds.backward(g)
        opt.step()

    for _ in range(warmup):
        step()
    cand = ["knn", "attn_stats", "attn_rows", "bwd_dq", "bwd_dv", "bwd_dk"]
    _lib.timing_select(cand)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    seen = {n: _lib.timing_read(n) for n in cand}
    _lib.timing_select([])
    dom = max((n for n in cand if seen.get(n)), key=lambda n: seen[n][1], default=None)
    fl = (2 * C * C * (3 * Nq + 2 * NB) * 3 + 2 * Nq * Nq * C + 2 * Nq * (Nq + NB) * C + 5 * 2 * Mq * (Nq + NB) * C) * Bq
    return {"config": f"{label}: one DownSampleToken layer fwd+bwd+SGD, B={Bq} C=128 N={Nq}->{Mq}",
            "ms_per_step": round(ms, 4), "clouds_per_s": round(Bq / (ms * 1e-3), 1), "steps": steps, "warmup": warmup,
            "dominant_kernel": dom, "dominant_kernel_us": round(seen[dom][1] * 1e3, 1) if dom else None,
            "kernel_us": {n: round(v[1] * 1e3, 1) for n, v in seen.items() if v},
            "step_fraction_of_fp32_mfma_roofline": round(fl / (ms * 1e-3) / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4)}


def metric_f32_mfma(steps, warmup):
    """The headline geometry with every product on the true fp32 matrix instruction (v_mfma_f32_32x32x2_f32;
    SAMBLE_MATRIX_MODE=f32, the arithmetic baseline the split-plane default is tested against): a driver clock for it."""
    from samble_amd import ops
    old = ops.MATRIX_MODE
    ops.MATRIX_MODE = "f32"
    try:
        r = quick_sampler(B_PER_GPU, N, M, steps, warmup)
    finally:
        ops.MATRIX_MODE = old
    r["config"] = (f"the headline layer (B={B_PER_GPU} C=128 N={N}->{M}) in matrix mode f32: v_mfma_f32_32x32x2_f32 "
                   "for every product (peak 157.3 TFLOP/s), logit-map pipeline")
    return r


def config0(steps, warmup, with_cpu=True):
    """BASELINE configs[0] -- ModelNet40 cls, B=8 N=1024->512, a single downsample layer, the reference's own CPU-runnable
    case -- on the GPU, with the CPU oracle (the bit-identical restatement of the reference) timed beside it on the host's
    cores in the same run."""
    Bq, Nq, Mq = 8, 1024, 512
    r = quick_sampler(Bq, Nq, Mq, steps, warmup, label="BASELINE configs[0]")
    if with_cpu:
        from oracle import torch_oracle as O
        from samble_amd import synth
        seed = 1000 * 0 + 1
        spec = O.SamplerSpec(M=Mq, K=KNN, C=C, num_bins=NB)
        st = O.SamplerState(*(torch.from_numpy(a) for a in synth.sampler_weights(C, NB, seed)))
        x = torch.from_numpy(synth.features(Bq, C, Nq, seed + 1))
        g = torch.from_numpy(synth.normal((Bq, C, Mq), seed + 2))
        noise = torch.from_numpy(synth.exp1((Bq * NB, Nq), seed + 3))
        O.sampler_grads(spec, st, x, g, noise)
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.sampler_grads(spec, st, x, g, noise)
            times.append(time.perf_counter() - t0)
        dt = statistics.median(times)
        r["cpu_baseline"] = dict(value=round(Bq / dt, 3), unit="clouds/s", cores=torch.get_num_threads(), kind="port",
                                 sample=f"median of 3 x fwd+bwd of {Bq} clouds (N={Nq}->{Mq}) after 1 warm-up, torch CPU oracle")
        r["gpu_over_cpu"] = round(r["clouds_per_s"] / r["cpu_baseline"]["value"], 1)
    return r


def extra_workloads(args):
    out = {}
    for name, fn in (("config0", lambda: config0(steps=20, warmup=5, with_cpu=not args.no_cpu_baseline)),
                     ("metric_f32_mfma", lambda: metric_f32_mfma(steps=8, warmup=3)),
                     ("stress", lambda: quick_sampler(16, 8192, 4096, steps=8, warmup=3)),
                     ("block_cls", lambda: measure_block("block_cls", steps=8, warmup=4)),
                     ("block_seg", lambda: measure_block("block_seg", steps=8, warmup=4))):
        try:
            r = fn()
            if name.startswith("block_"):
                r = {"config": r["config"]["workload"], "ms_per_step": r["ms_per_step"], "clouds_per_s": r["value"],
                     "steps": r["steps"], "warmup": r["warmup"], "dominant_kernel": r["roofline"]["kernel"],
                     "dominant_kernel_us": r["roofline"]["us_per_launch"],
                     "dominant_launches_per_step": r["roofline"]["launches_per_step"],
                     "kernel_family_ms_per_step": r["kernel_family_ms_per_step"]}
            out[name] = r
        except Exception as e:  # noqa: BLE001  (the headline line must survive)
            out[name] = {"error": repr(e)}
        torch.cuda.empty_cache()
    return out


def init_ranks(args):
    """(rank, world, device, shared_gpus): this process's place in the job, the process group formed when WORLD_SIZE > 1
    (RCCL = backend "nccl", as the reference's trainer forms it, train_modelnet.py:162-166; gloo when the ranks share GPUs)."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = max(torch.cuda.device_count(), 1)
    shared_gpus = world > ndev
    if shared_gpus and args.backend == "nccl":
        # (started by torch.distributed.run on a box with fewer GPUs than ranks: RCCL refuses two ranks on one device --
        # the ranks share GPUs over gloo, as launch_ranks arranges for `python bench.py --gpus N`; flagged in the line)
        args.backend = "gloo"
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or getattr(args, "force_process_group", False):
        # (--force-process-group: a ONE-rank group -- the N > 1 code path end to end, DDP's reducer, the in-forward
        # all-reduce and the collectives' report, over RCCL on a single GPU; tests/test_gpu_ddp.py)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
            else:
                dist.init_process_group(args.backend)
        except Exception as e:  # noqa: BLE001
            if "EADDRINUSE" in str(e) or "address already in use" in str(e).lower():
                sys.exit(EADDRINUSE_EXIT)  # (launch_ranks picks another port)
            raise
    return rank, world, dev, shared_gpus, ndev


def run_block(args):
    """`--workload block_cls | block_seg [--gpus N]`: with N > 1 every rank wraps the block as the reference's trainer does
    -- DistributedDataParallel(SyncBatchNorm.convert_sync_batchnorm(block)), train_modelnet.py:245-250 -- on its own 32
    clouds (BASELINE configs[3]: global batch 32 N)."""
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank, world, dev, shared_gpus, ndev = init_ranks(args)
    grouped = dist.is_initialized()
    result = measure_block(args.workload, args.steps, args.warmup, rank=rank, world=world, dev=dev,
                           breakdown=not args.no_breakdown)
    if rank == 0:
        result["config"]["backend"] = dist.get_backend() if grouped else None
        result["config"]["visible_gpus"] = ndev
        if shared_gpus:
            result["note"] = (f"{world} ranks share {ndev} GPU(s) over {args.backend}: functional check of the N>1 path "
                              "(DDP + SyncBatchNorm + boundary all-reduce), not a scaling measurement")
        print(json.dumps(result), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    return 0


BLOCK_FAMILIES = ["n2p_bwd", "edge_bwd", "knn", "edge_fwd", "n2p_fwd", "knn_small", "inv_nn", "seg_sum", "bwd_dq", "attn_rows",
                  "attn_stats", "lin_fwd", "lin_dx", "lin_dw", "lin_amax", "lin_amax_bwd", "bn_fwd", "bn_bwd", "lin_chain"]
# the families that have ever led a block step: these carry their events through the TIMED region (two event records per
# library call, ~20 calls per step), and the one with the most time per step in that region is the dominant one
BLOCK_CANDIDATES = ["n2p_bwd", "edge_bwd", "knn", "seg_sum", "lin_dw"]


def measure_block(workload, steps, warmup, rank=0, world=1, dev=None, breakdown=True):
    """BASELINE.json configs[1] (block_cls: EdgeConv x2 -> N2P -> sampler 2048->1024 -> N2P -> sampler 1024->512 -> N2P)
    and configs[2] (block_seg: the same path down with 4 bins, interpolation + N2P back up to 2048): one step = forward +
    backward + SGD of the whole block on B=32 clouds per rank of N=2048 xyz points resident in HBM.  grouped =
    configs[3]'s recipe: DDP(SyncBatchNorm(block)), one shard of the global batch per rank, timing bracketed by barriers,
    max over ranks.  The JSON line has the contract's shape; `roofline` is for the kernel family that takes the most time
    per step, timed by the library's HIP events on its launch stream over the timed steps themselves."""
    from types import SimpleNamespace
    args = SimpleNamespace(workload=workload, steps=steps, warmup=warmup)
    grouped = dist.is_available() and dist.is_initialized()   # (world > 1, or a forced one-rank group)
    from samble_amd import _lib, synth
    from samble_amd.blocks import FeatureLearningBlock, SegFeatureLearningBlock, block_config, seg_block_config
    if dev is None:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
    Bb, Nb = 32, 2048
    torch.manual_seed(1000 * (1 if args.workload == "block_cls" else 3))
    seg = args.workload == "block_seg"
    blk = SegFeatureLearningBlock(seg_block_config()) if seg else FeatureLearningBlock(block_config("cls"))
    if grouped:
        blk = torch.nn.SyncBatchNorm.convert_sync_batchnorm(blk)
    blk = blk.to(dev).train()
    model = torch.nn.parallel.DistributedDataParallel(blk, device_ids=[dev.index]) if grouped else blk
    xyz = torch.from_numpy(synth.xyz_clouds(Bb, Nb, 77, first_cloud=rank * Bb)).to(dev)
    opt = torch.optim.SGD(blk.parameters(), lr=1e-4)
    # a fixed upstream gradient of the block's output, as the metric workload drives its layer (rounds 3-5 put a
    # `feat.square().mean()` loss behind the block: five stock elementwise launches per step that are the harness's, not the
    # block's -- 0.05 ms of 7.5); scaled like d(mean of squares) so that the SGD steps stay as small as they were
    gshape = (Bb, 128, Nb) if seg else (Bb, 3 * 1024)
    g_up = torch.from_numpy(synth.normal(gshape, 78 + rank)).to(dev) * (2.0 / (gshape[0] * gshape[1] * (gshape[2] if seg else 1)))

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(xyz)
        feat = out if seg else out[0]
        feat.backward(g_up)
        opt.step()

    for _ in range(args.warmup):
        step()
    _lib.timing_select(BLOCK_CANDIDATES)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    host_ms = []
    for i in range(args.steps):
        h0 = time.perf_counter()
        step()
        marks[i + 1].record()
        host_ms.append(1e3 * (time.perf_counter() - h0))
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    seen = {n: _lib.timing_read(n) for n in BLOCK_CANDIDATES}
    _lib.timing_select([])
    timed_per_step = {n: v[0] * v[2] / args.steps for n, v in seen.items() if v}   # mean ms x launches / steps
    dominant = max(timed_per_step, key=timed_per_step.get)
    dom = seen[dominant]
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    comm = None
    if grouped:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        nparam = sum(p.numel() for p in blk.parameters())
        comm = measure_collectives(dev, world, c2_floats=nparam, c2_label="c2_ddp_all_parameters_allreduce_us")
        comm["c2_floats"] = nparam
        comm["syncbatchnorm_layers"] = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in blk.modules())
        comm["note"] += ("; every SyncBatchNorm adds one all-reduce of 2C+1 float64 per forward and one of 2C per backward "
                         "(csrc/batchnorm.hip between its two launches; EdgeConv: each glue entry in two halves around the all-reduce of its totals)")
    # every family of the step over further untimed steps (EVERY rank: a step holds collectives)
    per_step = dict(timed_per_step)
    if breakdown:
        _lib.timing_select(BLOCK_FAMILIES)
        extra = 3
        for _ in range(extra):
            step()
        torch.cuda.synchronize()
        allseen = {n: _lib.timing_read(n) for n in BLOCK_FAMILIES}
        _lib.timing_select([])
        per_step = {n: v[0] * v[2] / extra for n, v in allseen.items() if v}
        per_step.update(timed_per_step)   # (the candidates: the timed region's own figures)
    ms = 1e3 * elapsed / args.steps
    launches_per_step = dom[2] / args.steps if dom else 0
    # algorithmic work of the dominant family per launch, averaged over its launches of a step (layers of 2048 / 1024 /
    # 512 points): SURVEY 8(d) style, inputs read once and outputs written once, recomputation not counted
    K, Cc = 32, 128
    layers_n = [2048, 1024, 512] + ([1024, 2048] if seg else [])
    if dominant in ("n2p_bwd", "n2p_fwd"):
        # per point: its qkv row, the upstream gradient row, the K neighbour ids in; a dqkv row out
        by = sum(Bb * n * (3 * Cc * 4 * 2 + Cc * 4 + K * 4) for n in layers_n) / len(layers_n)
        roof = {"kernel": "n2p backward (transpose + n2p_bwd_point + n2p_bwd_gather)" if dominant == "n2p_bwd" else "n2p_attn_fwd",
                "bound": "hbm", "achieved": round(by / (dom[0] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "note": ("gather kernel: every point reads the K / V rows of its 32 neighbours (32 KB per point) through "
                         "L2; the algorithmic HBM bytes are the rows read once, so the fraction is small by construction")}
    elif dominant in ("edge_bwd", "edge_fwd"):
        # conv2 of the EdgeConv body on B*N*K edges, 64 -> 64 channels: forward 1 product, backward 2 (dh, dW2)
        fl = Bb * Nb * K * 2 * 64 * 64 * (2 if dominant == "edge_bwd" else 1)
        # split-bf16 kernels (csrc/edgeconv.hip): six bf16 products per product executed; the backward executes four
        # (y in both orientations, dh, dW2) for its two algorithmic ones, the forward one for one
        executed = 12.0 if dominant == "edge_bwd" else 6.0
        roof = {"kernel": "edge_mlp_bwd_tri_kernel" if dominant == "edge_bwd" else "edge_mlp_fwd_tri_kernel", "bound": "mfma",
                "achieved": round(fl / (dom[0] * 1e-3) / 1e12, 2), "peak": PEAK_BF16_MFMA_TFLOPS,
                "unit": "TFLOP/s", "executed_products": executed,
                "note": ("three bf16 planes per operand: 6 MFMA products per fp32 product; the backward recomputes the edge "
                         "activations in both orientations (2 products, not counted) beside dh and dW2; the kernel is bound "
                         "by vector issue (operand splits of tensors that are used once) and its waits, not by the matrix "
                         "pipe (DESIGN 7)")}
    elif dominant == "seg_sum":
        by = Bb * Nb * K * 64 * 4 + 2 * Bb * Nb * 64 * 4 + Bb * Nb * K * 4
        roof = {"kernel": "seg_sum_rows64_pair_kernel", "bound": "hbm", "achieved": round(by / (dom[0] * 1e-3) / 1e9, 1),
                "peak": PEAK_HBM_GBS, "unit": "GB/s"}
    elif dominant == "lin_dw":
        fl = sum(2.0 * Bb * n * Cc * 512 for n in layers_n) / len(layers_n)
        roof = {"kernel": "lin_dw_tri_kernel", "bound": "mfma", "achieved": round(fl / (dom[0] * 1e-3) / 1e12, 2),
                "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "executed_products": 3.0}
    else:
        fl = sum(2.0 * Bb * n * n * Cc for n in layers_n) / len(layers_n)
        roof = {"kernel": dominant, "bound": "mfma", "achieved": round(fl / (dom[0] * 1e-3) / 1e12, 2),
                "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "executed_products": 4.0}
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    if roof["bound"] == "mfma":
        roof["frac_executed"] = round(roof["achieved"] * roof["executed_products"] / PEAK_BF16_MFMA_TFLOPS, 4)
        roof["frac_note"] = ("frac = ALGORITHMIC fp32 flops / launch time / the dense 16-bit MFMA peak (2500 TFLOP/s); "
                             "frac_executed counts the 16-bit products the split-plane kernel issues per fp32 product")
    roof["us_per_launch"] = round(dom[0] * 1e3, 1)
    roof["launches_per_step"] = round(launches_per_step, 1)
    roof["chosen_from_ms_per_step"] = {n: round(v, 3) for n, v in timed_per_step.items()}
    roof["chosen_how"] = f"most time per step over the {args.steps} timed steps (HIP events on the launch stream)"
    roof["traffic"] = None
    try:  # fabric bytes per launch of the dominant family's kernels from the newest committed rocprofv3 --pmc summary
        import glob
        pmc_path = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}_pmc.json")))[-1]
        pmc = json.load(open(pmc_path))
        keys = {"n2p_bwd": ("n2p_bwd",), "n2p_fwd": ("n2p_attn_fwd",), "edge_bwd": ("edge_mlp_bwd",), "edge_fwd": ("edge_mlp_fwd",),
                "knn": ("knn_duo",), "seg_sum": ("seg_sum_rows64",), "lin_dw": ("lin_dw_tri",)}.get(dominant, (dominant,))
        hit = [e for k_, e in pmc["kernels"].items() if any(s_ in k_ for s_ in keys) and "traffic_bytes_per_launch" in e]
        if hit:
            roof["traffic"] = int(sum(e["traffic_bytes_per_launch"] * e.get("launches_per_step", 1) for e in hit)
                                  / max(sum(e.get("launches_per_step", 1) for e in hit), 1))
            roof["traffic_source"] = f"profiles/{os.path.basename(pmc_path)} (rocprofv3 --pmc of an earlier run of this command)"
        if pmc.get("step_traffic_bytes"):
            roof["step_traffic"] = int(pmc["step_traffic_bytes"])
    except Exception:  # noqa: BLE001
        pass
    result = {
        "metric": ("point-clouds/sec (feature-learning block fwd+bwd), " + ("ShapeNet-part seg block" if seg else "ModelNet40 cls block")
                   + " B=32 N=2048->1024->512" + ("->1024->2048" if seg else "")),
        "value": round(Bb * world * args.steps / elapsed, 2), "unit": "clouds/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "ms_per_step_median": round(statistics.median(step_ms), 4),
        "step_ms": [round(v, 3) for v in step_ms], "host_enqueue_ms": [round(v, 3) for v in host_ms],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (unit-sphere xyz clouds with jitter and anisotropic scale, random-init weights; no dataset files offline)",
        "config": {"workload": ("BASELINE configs[2]: SegFeatureLearningBlock" if seg else
                                ("BASELINE configs[1]: FeatureLearningBlock" if world == 1 else
                                 "BASELINE configs[3]: DDP(SyncBatchNorm(FeatureLearningBlock))"))
                               + " fwd+bwd+SGD, B=32/GPU xyz (32,3,2048), EdgeConv x2, N2P x" + ("5" if seg else "3")
                               + ", DownSampleToken x2 (" + ("4" if seg else "6") + " bins, random T=0.1, dynamic boundaries)"
                               + (", UpSampleInterpolation x2" if seg else ""),
                   "global_batch": Bb * world, "parallelism": f"dp{world}", "ranks": world},
        "roofline": roof,
        "kernel_family_ms_per_step": {n: round(v, 3) for n, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
    }
    if comm is not None:
        result["comm"] = comm
    return result

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="metric", choices=["metric", "stress", "block_cls", "block_seg"],
                    help="metric = BASELINE.json's B=32 N=2048->1024 (default); stress = configs[4]: B=16 N=8192->4096; "
                         "block_cls / block_seg = configs[1] / configs[2]: the whole feature-learning block, B=32 N=2048")
    ap.add_argument("--prewarm-steps", type=int, default=0,
                    help="untimed steps before the --warmup steps, with the parameters put back afterwards (metric / "
                         "stress workloads; worth ~1 %% on an idle GPU, off by default)")
    ap.add_argument("--lr", type=float, default=1e-4,
                    help="SGD learning rate of the synthetic step (0 keeps the weights where they are: long runs for "
                         "power / clock probes, where 1e-4 against a fixed random gradient would blow the weights up)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph replay of the step behind the timed region")
    ap.add_argument("--graph-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip the short stress / block_cls / block_seg runs behind the headline line's `workloads`")
    ap.add_argument("--logit-map", action="store_true",
                    help="A/B: keep the N x (N+nt) logit map in HBM (the round-1 pipeline) instead of the map-free forward")
    ap.add_argument("--force-process-group", action="store_true",
                    help="form a process group even for --gpus 1 (a one-rank group: the N > 1 code path over RCCL on one GPU)")
    ap.add_argument("--backend", default="nccl",
                    help="nccl (= RCCL, default) or gloo (ranks sharing a GPU: functional check of the N>1 path)")
    args = ap.parse_args()

    if args.workload.startswith("block_"):
        return run_block(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    global B_PER_GPU, N, M
    if args.workload == "stress":
        B_PER_GPU, N, M = 16, 8192, 4096
        args.no_cpu_baseline = True  # the CPU oracle needs minutes per cloud at this size
    rank, world, dev, shared_gpus, ndev = init_ranks(args)
    grouped = dist.is_initialized()
    local = dev.index

    from samble_amd import _lib, ops, sampler_config, synth
    from samble_amd.downsample import DownSampleToken
    if args.logit_map:
        import samble_amd.downsample as _ds
        _ds.MAP_FREE = False

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
    if grouped:
        model = torch.nn.parallel.DistributedDataParallel(mod, device_ids=[local])
    opt = torch.optim.SGD(mod.parameters(), lr=args.lr)

    # this rank's shard of the global batch (cloud ids rank*32 .. rank*32+31), resident in HBM
    x = torch.from_numpy(synth.features(B_PER_GPU, C, N, seed + 1, first_cloud=rank * B_PER_GPU)).to(dev)
    g = torch.from_numpy(synth.normal((B_PER_GPU, C, M), seed + 2 + rank)).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        xin = x.detach().requires_grad_(True)  # the layer sits mid-network: dL/dx is part of the work
        (x_ds, idx), _ = model(xin)
        x_ds.backward(g)
        opt.step()

    # Optional device pre-warm (untimed, before the W warm-up steps; off by default): a fresh process starts on a GPU
    # that has been idle, and the first tenths of a second run ~1 % slower.  Reported as prewarm_steps.
    prewarm_steps = max(args.prewarm_steps, 0)  # (a count, not a time: every rank runs the same collectives)
    if prewarm_steps:
        # the pre-warm must not change the workload: 250 SGD steps against a fixed random upstream gradient blow the
        # projection weights up (|W_q| 6.5 -> 18 after 300 steps), the attention saturates and the bins collapse onto
        # one.  Parameters and boundary state are put back afterwards.
        saved = {k_: v_.detach().clone() for k_, v_ in mod.state_dict().items()}
        saved_bounds = None if mod.bin_boundaries is None else [b_.clone() for b_ in mod.bin_boundaries]
        for _ in range(prewarm_steps):
            step()
        with torch.no_grad():
            mod.load_state_dict(saved)
        mod.bin_boundaries = saved_bounds
    if args.graph_child:
        # EVERY step before the capture runs on a side stream: the parameters' AccumulateGrad nodes are bound to the stream
        # of the first backward and stay alive (the module keeps `attention_bins_beforesoftmax`, which holds the graph);
        # bound to the legacy default stream they make the engine synchronise the capture stream with the NULL stream,
        # and hipStreamEndCapture answers that with SIGSEGV instead of an error (tools/experiments/graph_capture_probe.py)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(args.warmup, 1)):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
    else:
        for _ in range(args.warmup):
            step()
    if args.graph_child:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        for _ in range(args.steps):
            graph.replay()
        torch.cuda.synchronize()
        graph_ms = 1e3 * (time.perf_counter() - tg) / args.steps
        print(json.dumps({"ms_per_step": round(graph_ms, 4), "clouds_per_s": round(B_PER_GPU / (graph_ms * 1e-3), 1),
                          "steps": args.steps,
                          "note": "forward + backward + SGD of the headline step captured once (hipStreamBeginCapture via "
                                  "torch.cuda.graph) in a child process and replayed; not the headline value"}), flush=True)
        return
    tri = ops.MATRIX_MODE == "tri"
    # HIP events around every launch of the dominant kernel during the timed steps, recorded by the
    # library on the stream it launches on (two event records per step: no host sync); read back after
    # the final synchronize
    # the dominant kernel is whichever of the big matrix kernels takes longest on this box (rounds 1-2: the kNN; round 3
    # brought it under the sampled-row kernels)
    # (round 5: all four candidates carry their events through the TIMED region itself and the longest MEAN over its K
    # steps is the dominant kernel -- the two untimed steps rounds 3-4 chose from made it a coin toss between
    # `knn` and `bwd_dq`; eight event records per step, no host synchronisation)
    dominant = "knn" if tri else "bwd_rows_f32"
    cand = ["knn", "attn_stats", "attn_rows", "bwd_dq"] if tri else [dominant]
    _lib.timing_select(cand)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    host_ms = []
    for i in range(args.steps):
        h0 = time.perf_counter()
        step()
        marks[i + 1].record()
        host_ms.append(1e3 * (time.perf_counter() - h0))
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    seen = {n: _lib.timing_read(n) for n in cand}
    dominant = max((n for n in cand if seen.get(n)), key=lambda n: seen[n][0], default=dominant)
    dom = seen.get(dominant)
    _lib.timing_select([])
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    comm = None
    if grouped:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        comm = measure_collectives(dev, world)
    # every other kernel of the step, timed the same way (library-side HIP events on the launch stream) over 5 further
    # full steps: same cache state as the timed region.  EVERY rank runs them -- a step holds collectives (the boundary
    # all-reduce, DDP's buckets): rank 0 alone would wait for the others forever
    kt = None
    if not args.no_breakdown:
        names = [n for n in _lib.TIMED_KERNELS if n != "attn_fwd"]
        _lib.timing_select(names)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        kt = {n: _lib.timing_read(n) for n in names}
        _lib.timing_select([])

    # the same step as ONE hipGraph (torch.cuda.graph around forward + backward + SGD; the selection noise then comes from
    # torch's graph-safe generator path, one extra launch), replayed K times and reported beside the eager headline.  Eager
    # stays the headline at every N: DDP's reducer is not captured, and a scaling curve must not mix the two.  The capture
    # runs in a CHILD process (`--graph-child`): on this ROCm 7.2 / torch 2.10 image hipStreamEndCapture takes the process
    # down with SIGSEGV where CUDA would return an error -- always under rocprofv3, and whenever a backward has run on the
    # legacy default stream before the capture (the parameters' AccumulateGrad nodes stay bound to it), which the timed
    # region above has done.  A crash that cannot be caught must not be able to cost the headline line.
    graph_report = None
    profiled = any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and rank == 0 and not args.no_graph and args.workload == "metric":
        if profiled:
            graph_report = {"skipped": "running under rocprofv3: stream capture crashes inside the profiler's tool library"}
        else:
            try:
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
                child = subprocess.run([sys.executable, os.path.abspath(__file__), "--graph-child", "--steps", str(args.steps),
                                        "--warmup", "3", "--lr", str(args.lr)], capture_output=True, text=True, timeout=600, env=env)
                lines = [l for l in child.stdout.splitlines() if l.startswith("{")]
                if child.returncode == 0 and lines:
                    graph_report = json.loads(lines[-1])
                else:
                    graph_report = {"error": f"capture child exited with code {child.returncode}",
                                    "stderr_tail": child.stderr[-200:]}
            except Exception as e:  # noqa: BLE001
                graph_report = {"error": repr(e)[:300]}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        total_clouds = B_PER_GPU * world * args.steps
        value = total_clouds / elapsed
        fl = algorithmic_flops_per_cloud()
        by = algorithmic_bytes_per_cloud()
        result = {
            "metric": ("point-clouds/sec (downsample fwd+bwd), ModelNet40 B=32 N=2048->1024" if args.workload == "metric"
                       else "point-clouds/sec (downsample fwd+bwd), synthetic dense clouds B=16 N=8192->4096"),
            "value": round(value, 2),
            "unit": "clouds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_steps": prewarm_steps,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_median": round(statistics.median(step_ms), 4),
            # the timed region starts on a drained device (the contract's synchronize): while the host is still ahead of
            # nothing, a step takes what its launches take to enqueue -- these two lists show where mean and median part
            "step_ms": [round(v, 3) for v in step_ms],
            "host_enqueue_ms": [round(v, 3) for v in host_ms],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (hash-generated N(0,1) features, random-init weights; no ModelNet40 files offline)",
            "config": {"workload": f"one DownSampleToken layer fwd+bwd+SGD, cls layer 0: B={B_PER_GPU}/GPU C=128 N={N}->M={M} "
                                   "nb=6 K=32 sparse_col_sqr random T=0.1 dynamic boundaries",
                       "global_batch": B_PER_GPU * world, "parallelism": f"dp{world}",
                       "backend": (dist.get_backend() if grouped else None),
                       "ranks": (dist.get_world_size() if grouped else 1),
                       "visible_gpus": ndev},
            "step_fraction_of_mfma_roofline": round(
                (fl["fwd"] + fl["bwd"]) * B_PER_GPU / (ms_per_step * 1e-3) / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
        }
        if graph_report is not None:
            result["graph_replay"] = graph_report
        if comm is not None:
            result["comm"] = comm
        if shared_gpus:
            result["note"] = (f"{world} ranks share {ndev} GPU(s) over {args.backend}: functional check of the N>1 path "
                              "(DDP + boundary all-reduce), not a scaling measurement")

        def pmc_entry(kernel):
            # per-launch counters from the newest committed rocprofv3 --pmc summary (profiles/*_pmc.json: fabric bytes
            # = (2*FETCH_SIZE + WRITE_SIZE)*1024, separate passes as the counters require; matrix-pipe busy =
            # SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x 1024 SIMDs x 2.4 GHz))
            try:
                import glob
                path = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{args.workload}_pmc.json")))[-1]
                pmc = json.load(open(path))["kernels"]
                key = next(k for k in pmc if k == kernel or k.startswith(kernel + "<"))  # template arguments vary
                return pmc[key], "profiles/" + os.path.basename(path)
            except Exception:
                return None, None

        def pmc_traffic(kernel):
            e, src = pmc_entry(kernel)
            return (e.get("traffic_bytes_per_launch"), src) if e else (None, None)

        def roof(kernel, alg_flops, ms, pmc_name, timed_as=None):
            """MFMA roofline of one kernel: achieved = ALGORITHMIC fp32 flops / launch time; peak = the dense
            MFMA peak of the instruction the kernel issues (16-bit: 2500 TFLOP/s; matrix mode f32: 157.3); frac = achieved /
            peak.  frac_executed = the 16-bit products actually issued / 2500 (what the matrix pipe is busy with)."""
            ach = alg_flops / (ms * 1e-3) / 1e12
            e, src = pmc_entry(pmc_name)
            if tri:
                products, why = EXECUTED_PRODUCTS.get(timed_as, (6.0, "three bf16 planes per operand: 6 products"))
                peak = PEAK_BF16_MFMA_TFLOPS
            else:
                products, why, peak = 1.0, "v_mfma_f32_32x32x2_f32 (true fp32 products)", PEAK_FP32_MFMA_TFLOPS
            out = {"kernel": kernel, "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                   "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                   "traffic": e.get("traffic_bytes_per_launch") if e else None,
                   "traffic_source": (f"{src} (rocprofv3 --pmc of an earlier run of this command, not measured in this run)"
                                      if src else None),
                   "us_per_launch": round(ms * 1e3, 1), "algorithmic_flops_per_launch": alg_flops,
                   "executed_products": products,
                   "executed_tflops": round(ach * products, 1),
                   "frac_executed": round(ach * products / peak, 4),
                   "mfma_busy": e.get("mfma_busy_frac_at_2.4GHz") if e else None}
            if tri:
                out["peak_note"] = (f"frac = ALGORITHMIC fp32 TFLOP/s (SURVEY 8d's flops, no credit for split products or "
                                    f"recomputation) / the dense 16-bit MFMA peak 2500; the kernel issues {products:g} 16-bit "
                                    f"products per algorithmic fp32 product ({why}): frac_executed = executed 16-bit TFLOP/s / "
                                    "2500; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES of the committed PMC run / (kernel time x 1024 "
                                    "SIMDs x 2.4 GHz)")
                out["frac_of_fp32_mfma_peak"] = round(ach / PEAK_FP32_MFMA_TFLOPS, 4)
            return out

        def hbm(kernel, alg_bytes, ms, pmc_name):
            ach = alg_bytes / (ms * 1e-3) / 1e9
            traffic, src = pmc_traffic(pmc_name)
            return {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic,
                    "traffic_source": (f"{src} (earlier rocprofv3 --pmc run)" if src else None),
                    "us_per_launch": round(ms * 1e3, 1), "algorithmic_bytes_per_launch": alg_bytes}

        dominant_ms = dom[0] if dom else float("nan")
        import samble_amd.downsample as _dsm1
        map_free1 = tri and _dsm1.MAP_FREE

        def tri_desc(n_):
            """(kernel name, algorithmic flops per launch, rocprof kernel name, note) of a split-bf16 step kernel"""
            if n_ == "knn":
                return ("knn_duo_kernel", fl["dist"] * B_PER_GPU, "samble::knn_duo_kernel", None)
            if n_ == "attn_stats":
                nm = "attn_stats_nl_tri_kernel" if map_free1 else "attn_stats_tri_kernel"
                return (nm, fl["qk"] * B_PER_GPU, "samble::" + nm, None)
            if n_ == "attn_rows":
                nm = "attn_rows_rc_tri_kernel" if map_free1 else "attn_rows_tri_kernel"
                return (nm, fl["av"] * B_PER_GPU, "samble::" + nm, None)
            nm = "bwd_dq_pm_tri_kernel" if map_free1 else "bwd_dq_tri_kernel"
            return (nm, 2 * fl["av"] * B_PER_GPU, "samble::" + nm, None)

        if tri:
            nm, flops, pmcn, note = tri_desc(dominant)
            result["roofline"] = roof(nm, flops, dominant_ms, pmcn, timed_as=dominant)
            if note:
                result["roofline"]["note"] = note
            result["roofline"]["chosen_from_us"] = {n_: round(v[0] * 1e3, 1) for n_, v in seen.items() if v}
            result["roofline"]["chosen_how"] = f"longest mean launch time over the {args.steps} timed steps (HIP events on the launch stream)"
        else:
            bwd_alg = 4 * 2 * M * N * C * B_PER_GPU
            result["roofline"] = roof("bwd_rows_kernel", bwd_alg, dominant_ms, "samble::bwd_rows_kernel")
        result["roofline"]["launches_timed"] = dom[2] if dom else 0
        try:  # sum of fabric traffic over one step's launches (committed PMC summary) / the ideal fused layer's bytes
            import glob
            pmc_path = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{args.workload}_pmc.json")))[-1]
            pmc_all = json.load(open(pmc_path))
            step_traffic = pmc_all.get("step_traffic_bytes")
            if step_traffic is None:
                # (summaries of rounds 1-4 carry no launch counts: every kernel once per step, the key-stationary
                # accumulation twice -- dV and dK)
                step_traffic = sum(k_.get("traffic_bytes_per_launch", 0) * k_.get("launches_per_step", 2 if "bwd_kacc" in n_ else 1)
                                   for n_, k_ in pmc_all["kernels"].items())
            result["roofline"]["step_traffic"] = int(step_traffic)
            result["roofline"]["traffic_ratio"] = round(step_traffic / (by["fused_step"] * B_PER_GPU), 2)
            result["roofline"]["traffic_ratio_note"] = (
                f"sum of fabric bytes over one step's launches (profiles/{os.path.basename(pmc_path)}) / the ideal fused "
                "layer's algorithmic bytes (4.20 MB/cloud): the two M-row maps P and dS are 1.36 GB of it (DESIGN 8; "
                "tools/experiments/map_alias.md: keeping them on-die was measured and buys nothing)")
        except Exception:  # noqa: BLE001
            result["roofline"]["traffic_ratio"] = None
        result["matrix_mode"] = ops.MATRIX_MODE
        import samble_amd.downsample as _dsm0
        result["forward"] = ("map-free (K neighbour logits per row in pass 1, sampled rows recomputed in pass 2)"
                             if (tri and _dsm0.MAP_FREE) else "logit map in HBM (two-pass)")
        if kt is not None:
            result["kernel_us"] = {n: round(v[1] * 1e3, 1) for n, v in kt.items() if v}
            sfx = "_tri_kernel" if tri else "_kernel"
            import samble_amd.downsample as _dsm
            map_free = tri and _dsm.MAP_FREE  # no N x (N+nt) logit map: pass 1 keeps K logits per row, pass 2 recomputes
            mf = []
            if tri:
                for n_ in ("knn", "attn_stats", "attn_rows", "bwd_dq"):
                    if kt.get(n_):
                        nm, flops, pmcn, note = tri_desc(n_)
                        r_ = roof(nm, flops, kt[n_][0], pmcn, timed_as=n_)
                        if note:
                            r_["note"] = note
                        mf.append(r_)
                # backward: dV and dK accumulated key-stationary from the two maps (1 product each, N keys)
                if kt.get("bwd_dv"):
                    mf.append(roof("bwd_kacc_pm_tri_kernel (dV)" if map_free else "bwd_kacc_tri_kernel<0> (dV)",
                                   2 * M * N * C * B_PER_GPU, kt["bwd_dv"][0],
                                   "samble::bwd_kacc_pm_tri_kernel<false>" if map_free else
                                   "samble::bwd_kacc_tri_kernel<0, false>", timed_as="bwd_dv"))
                if kt.get("bwd_dk"):
                    mf.append(roof("bwd_kacc_pm_tri_kernel (dK)", 2 * M * N * C * B_PER_GPU, kt["bwd_dk"][0],
                                   "samble::bwd_kacc_pm_tri_kernel<false>", timed_as="bwd_dk"))
            else:
                if kt.get("attn_stats"):
                    mf.append(roof("attn_stats" + sfx, fl["qk"] * B_PER_GPU, kt["attn_stats"][0], "samble::attn_stats" + sfx))
                if kt.get("attn_rows"):
                    mf.append(roof("attn_rows" + sfx, fl["av"] * B_PER_GPU, kt["attn_rows"][0], "samble::attn_rows" + sfx))
                if kt.get("knn"):
                    mf.append(roof("knn_stream_kernel", fl["dist"] * B_PER_GPU, kt["knn"][0], "samble::knn_stream_kernel"))
            pj = 2 * C * C * 3 * N * B_PER_GPU
            for n_, name, flops in (("proj_fwd", "proj_fwd" + sfx, pj), ("proj_dx", "proj_dx" + sfx, pj),
                                    ("proj_dw", ("proj_dw_tri_kernel (+ reduce and token gradients)" if tri else
                                                 "proj_dw_kernel (+ reduce; fp32 MFMA)"), pj)):
                if kt.get(n_):
                    r_ = roof(name, flops, kt[n_][0], "samble::" + name.split(" ")[0], timed_as=n_)
                    mf.append(r_)
            result["roofline_other_kernels"] = mf
            # HBM-bound kernels: algorithmic bytes / launch duration against 8 TB/s
            hb = []
            if kt.get("sparse_score"):
                hb.append(hbm("sparse_score_map + finalize_score", by["sparse_score"] * B_PER_GPU, kt["sparse_score"][0],
                              "samble::sparse_score_map_kernel"))
            sel = {n_: kt.get(n_) for n_ in ("quantiles", "bin_assign", "alloc_counts", "bin_select")}
            if sel["quantiles"] and sel["bin_assign"] and sel["bin_select"]:
                # fused chain: "quantiles" = score + z + batch quantiles, "bin_assign" = boundaries + bins + counts
                hb.append(hbm("select chain (" + " + ".join(k_ for k_, v in sel.items() if v) + ")",
                              by["select"] * B_PER_GPU, sum(v[0] for v in sel.values() if v), "samble::bin_select_kernel"))
            if kt.get("nn_prepare"):
                hb.append(hbm("nn_prepare (ascending neighbour lists + membership words)", by["nn_prepare"] * B_PER_GPU,
                              kt["nn_prepare"][0], "nn_prepare_kernel"))
            for n_, label, key in (("knn_prep", "cloud_mean_amax + duo_split_cm (kNN operand image + norms)", "knn_prep"),
                                   ("tri_split", "tri_split_qkv (operand images of Q, K, V)", "tri_split"),
                                   ("bwd_prep", "bwd_prep_tri (gather of the sampled rows + their operand images)", "bwd_prep")):
                if kt.get(n_):
                    hb.append(hbm(label, by[key] * B_PER_GPU, kt[n_][0], "samble::" + label.split(" ")[0]))
            if kt.get("knn"):
                hb.append(hbm("knn (SURVEY 8d per-kernel bytes; compute-bound by construction)", by["knn"] * B_PER_GPU,
                              kt["knn"][0], "samble::knn_duo_kernel"))
            hb.append(hbm("whole step vs the ideal fused layer's bytes (4.20 MB/cloud)", by["fused_step"] * B_PER_GPU,
                          ms_per_step, "-"))
            result["roofline_hbm_kernels"] = hb
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"], oracle_out = cpu_baseline(seed)
            result["gpu_over_cpu"] = round(value / result["cpu_baseline"]["value"], 2)
            try:
                result["parity"] = parity_vs_oracle(seed, dev, oracle_out)
            except Exception as e:  # noqa: BLE001  (never lose the headline line to the extra check)
                result["parity"] = {"error": repr(e)}
        if world == 1 and args.workload == "metric" and not args.no_extra_workloads:
            # BASELINE configs[4], [1], [2] with a clock of this run on them: ms/step + the kernel (family) that takes the
            # most time, a few steps each after the headline's timed region (their own full lines: --workload <name>)
            result["workloads"] = extra_workloads(args)
        print(json.dumps(result), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

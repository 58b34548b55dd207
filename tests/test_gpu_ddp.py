"""Two ranks through torch.nn.parallel.DistributedDataParallel(DownSampleToken) on the GPU: the N>1 path of
BASELINE.json configs[3] (reference train_modelnet.py:66-71, 162-166, 245-250; utils/ops.py:191-199).

The GPU box has one device, so the two ranks share cuda:0 over gloo; on an 8-GPU node the same code runs
over RCCL (backend "nccl").  Checked: both ranks end every step with IDENTICAL bin boundaries equal to the
blend of the rank-averaged batch quantiles; the sampled indices of each rank equal a single-process run of
its shard under those boundaries; parameter gradients on both ranks equal the MEAN of the two single-process
gradients (DDP's bucketed all-reduce); bench.py --gpus 2 starts its own ranks and reports n_gpus = 2."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(tmp_path, world=2, backend="gloo", what="sampler", extra_env=None):
    env0 = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_worker.py"), str(tmp_path), backend, what],
                              env=dict(env0, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    prefix = {"sampler": "rank", "edgeconv": "edge"}.get(what, "block")
    return [torch.load(os.path.join(tmp_path, f"{prefix}{r}.pt")) for r in range(world)]


def test_ddp_two_ranks_boundaries_indices_and_gradients(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ddp_worker as W
    from samble_amd import ops
    res = _run_ranks(tmp_path)
    assert res[0]["world"] == 2 and res[1]["world"] == 2
    state = None
    for call in range(2):
        r0, r1 = res[0]["log"][call], res[1]["log"][call]
        # identical state on both ranks, = blend of the MEAN of the two local quantile vectors
        assert torch.equal(r0["upper"], r1["upper"]) and torch.equal(r0["lower"], r1["lower"])
        assert not torch.equal(r0["local_q"], r1["local_q"]), "ranks hold different shards"
        mean_q = ((r0["local_q"] + r1["local_q"]) / 2).to(DEV)
        state = ops.blend_boundaries(state, mean_q, W.NB, 0.99)
        torch.cuda.synchronize()
        assert torch.equal(state[0].cpu(), r0["upper"]) and torch.equal(state[1].cpu(), r0["lower"])
        # single-process runs of each shard under those boundaries
        grads = []
        for rank, rr in enumerate((r0, r1)):
            mod = W.build_module(DEV)
            mod.dynamic_boundaries_enable = False
            mod.bin_boundaries = [rr["upper"].to(DEV).clone(), rr["lower"].to(DEV).clone()]
            x, noise, g = W.shard(rank, DEV)
            xin = x.detach().requires_grad_(True)
            (x_ds, idx), _ = mod(xin, noise=noise)
            x_ds.backward(g)
            assert torch.equal(idx.cpu(), rr["idx"]), f"rank {rank}: sampled indices differ from the single-process run"
            torch.testing.assert_close(x_ds.detach().cpu(), rr["x_ds"], rtol=0, atol=0)
            torch.testing.assert_close(xin.grad.cpu(), rr["dx"], rtol=0, atol=0)  # dx is per shard, not averaged
            grads.append({n: p.grad.detach().cpu() for n, p in mod.named_parameters()})
        for name in grads[0]:
            want = (grads[0][name] + grads[1][name]) / 2
            for rr in (r0, r1):
                got = rr["grads"][name]
                err = float((got - want).abs().max())
                assert err <= 1e-6 * float(want.abs().max()) + 1e-9, (name, err)
            assert torch.equal(r0["grads"][name], r1["grads"][name]), "DDP leaves the same gradient on every rank"
    _check_collected(res)


def _check_collected(res):
    """rank 0 holds every rank's published variables in rank order (utils/ops.py:289-384); the other ranks None"""
    world = len(res)
    for r in range(1, world):
        assert all(v is None for v in res[r]["gathered"].values())
    got = res[0]["gathered"]
    for name in ("attention_point_score", "idx", "bin_prob", "k_point_to_choose"):
        assert torch.equal(got[name], torch.cat([res[r]["published"][name] for r in range(world)], dim=0)), name
    chunks = got["idx_chunks"]                     # world B clouds x num_bins x (1, n)
    B = len(res[0]["published"]["idx_chunks"][0])
    assert len(chunks) == world * B
    for r in range(world):
        pub = res[r]["published"]["idx_chunks"]    # num_bins x B x (1, n)
        for b in range(B):
            for t in range(len(pub)):
                assert torch.equal(chunks[r * B + b][t], pub[t][b]), (r, b, t)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no launcher): the parent spawns the ranks before touching the GPU and rank 0
    prints the JSON line with n_gpus = 2 (on this 1-GPU box the ranks share the device over gloo)."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                          "--prewarm-steps", "4", "--no-breakdown", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["ranks"] == 2 and rec["config"]["global_batch"] == 64
    assert rec["value"] > 0 and rec["scaling"] == "weak"


def test_fused_edgeconv_pools_syncbatchnorm_statistics(tmp_path):
    """the fused EdgeConv computed BatchNorm statistics from this rank's edges only.  Under
    nn.SyncBatchNorm with 2 ranks it must equal the stock composition under nn.SyncBatchNorm: outputs, dx, every
    parameter gradient and the running buffers."""
    res = _run_ranks(tmp_path, what="edgeconv")
    for rank, r in enumerate(res):
        f, s_ = r["fused"], r["stock"]
        scale = float(s_["y"].abs().max())
        # arg-max flips at near-ties are inherent to any fp32 evaluation: compare in relative L2
        rel = float((f["y"] - s_["y"]).norm() / s_["y"].norm())
        assert rel <= 1e-4, (rank, rel, scale)
        rel = float((f["dx"] - s_["dx"]).norm() / s_["dx"].norm())
        assert rel <= 2e-3, (rank, "dx", rel)
        for n in s_["grads"]:
            a, b2 = f["grads"][n], s_["grads"][n]
            assert float((a - b2).norm()) <= 2e-3 * float(b2.norm()) + 1e-7, (rank, n)
        for n in s_["bufs"]:
            if s_["bufs"][n].dtype.is_floating_point:
                torch.testing.assert_close(f["bufs"][n], s_["bufs"][n], rtol=1e-4, atol=1e-6)
    # pooled statistics: both ranks hold the same running buffers
    for n in res[0]["fused"]["bufs"]:
        assert torch.equal(res[0]["fused"]["bufs"][n], res[1]["fused"]["bufs"][n]), n


def test_bench_two_ranks_finishes_and_reports_the_collectives():
    """`python bench.py --gpus 2` (the driver's N > 1 contract) on this box: two ranks share the GPU over gloo.  The run
    must FINISH -- every phase that holds collectives (timed steps, the per-kernel breakdown steps) runs on every rank --
    and the line carries the communication report (library, ranks formed, latencies of the two data-path messages)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=420, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["config"]["global_batch"] == 64
    assert line["scaling"] == "weak" and line["value"] > 0 and "kernel_us" in line
    comm = line["comm"]
    assert comm["ranks_formed"] == 2 and comm["c1_boundary_allreduce_5_floats_us"] > 0 and comm["c2_ddp_bucket_399KB_allreduce_us"] > 0


def test_bench_eight_ranks_over_gloo_on_one_gpu():
    """BASELINE.json configs[3]'s rank count on the one GPU this box has: `bench.py --gpus 8 --backend gloo --steps 2`.
    Eight child processes (spawned before the parent touches the GPU), eight DDP replicas of the layer, eight shards of the
    global batch of 256, the boundary exchange and DDP's buckets over an 8-rank group.  A functional check of the N = 8
    path -- the line says so itself -- never a scaling number."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--no-breakdown"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["config"]["ranks"] == 8 and line["config"]["global_batch"] == 256
    assert line["config"]["parallelism"] == "dp8" and line["scaling"] == "weak" and line["value"] > 0
    assert "not a scaling measurement" in line["note"]
    comm = line["comm"]
    assert comm["ranks_formed"] == 8 and comm["world_size"] == 8 and comm["backend"] == "gloo"


def _block_single_process(W, kind, ranks, res, fused_glue):
    """The same block in ONE process (plain BatchNorm, no process group) on the concatenation of the ranks' shards, under
    the boundaries and the sampled indices the ranks ended each call with: per call what `ddp_worker.block_step` returns."""
    from samble_amd import embedding
    blk = W.build_block(kind).to(DEV).train()
    shards = [W.block_shard(r, DEV, kind) for r in ranks]
    xyz = torch.cat([s_[0] for s_ in shards])
    noise = [torch.cat([s_[1][i] for s_ in shards]) for i in range(2)]
    g = torch.cat([s_[2] for s_ in shards])
    old = embedding.FUSED_GLUE
    embedding.FUSED_GLUE = fused_glue
    out = []
    try:
        for call in range(2):
            for i, layer in enumerate(blk.downsample_list):
                layer.dynamic_boundaries_enable = False
                layer.bin_boundaries = [t.to(DEV).clone() for t in res[ranks[0]]["log"][call]["bounds"][i]]
            forced = [torch.cat([res[r]["log"][call]["idx"][i] for r in ranks]).to(DEV) for i in range(2)]
            out.append(W.block_step(blk, blk, xyz, noise, g, kind, forced=forced))
    finally:
        embedding.FUSED_GLUE = old
    return out


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("kind", ["cls", "seg"])
def test_block_under_ddp_and_syncbatchnorm_two_ranks(tmp_path, kind):
    """BASELINE configs[3]'s recipe at block level (reference train_modelnet.py:245-250, configs/default.yaml:94
    `syn_bn: true`): DistributedDataParallel(SyncBatchNorm.convert_sync_batchnorm(block)) on two ranks against ONE process
    on the concatenated batch -- the output rows, dx, every parameter gradient (DDP leaves the MEAN over the ranks: half
    the single-process gradient of the summed loss), every BatchNorm buffer, and both samplers' boundaries.

    Under SyncBatchNorm the attention layers stay on the fused node and the own BatchNorm kernels (their per-channel
    float64 sums all-reduced between the two launches); the single process runs plain BatchNorm over the whole batch.  The
    samplers' boundaries are rank-averaged quantiles by the reference's design (utils/ops.py:191-199), not the whole
    batch's: the single process takes the ranks' boundaries and sampled indices as given (as the sampler test above does)
    and must then reproduce everything else."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ddp_worker as W
    what = "block" if kind == "cls" else "block_seg"
    res = _run_ranks(tmp_path, what=what)
    assert res[0]["world"] == 2 and res[0]["bn_types"] == ["SyncBatchNorm"], res[0]["bn_types"]
    # (EdgeConv: the same fused kernels on the ranks -- their totals all-reduced between the two halves of every glue entry --
    # and in the single process)
    single = _block_single_process(W, kind, (0, 1), res, fused_glue=True)
    Bs = W.BLK_B
    for call in range(2):
        r0, r1 = res[0]["log"][call], res[1]["log"][call]
        for i in range(2):   # both samplers: identical state on both ranks after the in-forward all-reduce
            assert torch.equal(r0["bounds"][i][0], r1["bounds"][i][0]) and torch.equal(r0["bounds"][i][1], r1["bounds"][i][1])
            assert not torch.equal(r0["idx"][i], r1["idx"][i])
        one = single[call]
        for r, rr in enumerate((r0, r1)):
            rows = slice(r * Bs, (r + 1) * Bs)
            assert _rel(rr["y"], one["y"][rows]) <= 2e-5, (call, r, "y", _rel(rr["y"], one["y"][rows]))
            assert _rel(rr["dx"], one["dx"][rows]) <= 2e-4, (call, r, "dx", _rel(rr["dx"], one["dx"][rows]))
        worst = {}
        # a bias whose layer feeds a normalisation has a gradient that cancels to ~0 (sum of dy over the batch): the error is
        # measured against the tensor's own norm OR the norm of the same layer's companion weight gradient, whichever is larger
        norms = {n: float(w_.double().norm()) for n, w_ in one["grads"].items()}
        for name, want in one["grads"].items():
            assert torch.equal(r0["grads"][name], r1["grads"][name]), f"DDP leaves the same gradient on every rank: {name}"
            companion = norms.get(name[:-len("bias")] + "weight", 0.0) if name.endswith(".bias") else 0.0
            worst[name] = float((r0["grads"][name].double() * 2 - want.double()).norm()) / max(norms[name], companion, 1e-30)
        bad = {k: (v, norms[k]) for k, v in worst.items() if not v <= 5e-4}
        assert not bad, (call, bad)
        for name, want in one["bufs"].items():
            for rr in (r0, r1):
                got = rr["bufs"][name]
                if got.dtype.is_floating_point:
                    torch.testing.assert_close(got, want, rtol=2e-5, atol=1e-6, msg=lambda m: f"{name}: {m}")
                else:
                    assert torch.equal(got, want), name
        print(f"call {call}: worst gradient rel-L2 {max(worst.values()):.2e} ({max(worst, key=worst.get)})")


def test_rccl_world_size_one_sampler_step_is_the_single_process_step(tmp_path):
    """RCCL on the one GPU of this box (it refuses two ranks on a device, so the group has ONE rank): init_process_group
    ("nccl", device_id=...) as bench.py and the reference (train_modelnet.py:162-166) form it, the in-forward all-reduce of
    the boundary quantiles on device memory (utils/ops.py:191-199) and DistributedDataParallel's reducer all execute over
    RCCL.  With one rank every collective is the identity: boundaries, indices, output and gradients must be bit for bit
    those of the same two steps without a process group."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ddp_worker as W
    res = _run_ranks(tmp_path, world=1, backend="nccl")
    assert res[0]["world"] == 1 and res[0]["backend"] == "nccl"
    mod = W.build_module(DEV)
    x, noise, g = W.shard(0, DEV)
    for call in range(2):
        mod.zero_grad(set_to_none=True)
        xin = x.detach().requires_grad_(True)
        (x_ds, idx), _ = mod(xin, noise=noise)
        x_ds.backward(g)
        rr = res[0]["log"][call]
        assert torch.equal(mod.bin_boundaries[0].cpu(), rr["upper"]) and torch.equal(mod.bin_boundaries[1].cpu(), rr["lower"])
        assert torch.equal(idx.cpu(), rr["idx"])
        assert torch.equal(x_ds.detach().cpu(), rr["x_ds"]) and torch.equal(xin.grad.cpu(), rr["dx"])
        for n, p in mod.named_parameters():
            assert torch.equal(p.grad.cpu(), rr["grads"][n]), n
    _check_collected(res)   # (all_gather + gather of the published variables over RCCL)


def test_rccl_world_size_one_block_with_pooled_syncbatchnorm(tmp_path):
    """DDP(SyncBatchNorm(block)) over a one-rank RCCL group with SAMBLE_POOL_SINGLE_RANK=1: every SyncBatchNorm takes the
    POOLED route (statistics kernel -> all-reduce of the float64 sums over RCCL -> normalisation kernel, and the same in
    the backward; EdgeConv's glue entries in their SUMS / APPLY halves) although one rank is all there is.  An all-reduce
    over one rank is the identity and the pooled kernels add the same partials in the same order, so the result must equal
    the un-pooled single process."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ddp_worker as W
    res = _run_ranks(tmp_path, world=1, backend="nccl", what="block", extra_env={"SAMBLE_POOL_SINGLE_RANK": "1"})
    assert res[0]["world"] == 1 and res[0]["backend"] == "nccl" and res[0]["bn_types"] == ["SyncBatchNorm"]
    single = _block_single_process(W, "cls", (0,), res, fused_glue=True)
    for call in range(2):
        rr, one = res[0]["log"][call], single[call]
        assert _rel(rr["y"], one["y"]) <= 1e-6 and _rel(rr["dx"], one["dx"]) <= 1e-5, (_rel(rr["y"], one["y"]), _rel(rr["dx"], one["dx"]))
        for name, want in one["grads"].items():
            assert _rel(rr["grads"][name], want) <= 1e-5, (name, _rel(rr["grads"][name], want))
        for name, want in one["bufs"].items():
            if want.dtype.is_floating_point:
                torch.testing.assert_close(rr["bufs"][name], want, rtol=1e-6, atol=1e-7)
            else:
                assert torch.equal(rr["bufs"][name], want), name


@pytest.mark.parametrize("workload", ["metric", "block_cls"])
def test_bench_multi_rank_path_over_rccl_with_one_rank(workload):
    """`bench.py --force-process-group`: the N > 1 code path of the benchmark -- init_process_group("nccl", device_id=...),
    DistributedDataParallel (around SyncBatchNorm(block) for the block workload), the barriers, the max-over-ranks
    all-reduce, the collectives' report -- on a ONE-rank RCCL group, which is all a one-GPU box can form (RCCL refuses two
    ranks on a device).  What the driver's 8-GPU launch will run, minus the other seven ranks."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-process-group", "--steps", "3", "--warmup", "2",
           "--no-extra-workloads", "--no-cpu-baseline", "--no-graph", "--workload", workload]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    comm = line["comm"]
    assert comm["backend"] == "nccl" and comm["ranks_formed"] == 1 and comm["world_size"] == 1
    assert comm["c1_boundary_allreduce_5_floats_us"] > 0
    if workload == "block_cls":
        assert comm["syncbatchnorm_layers"] == 10 and line["config"]["backend"] == "nccl"

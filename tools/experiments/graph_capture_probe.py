"""Which launch of the sampler step breaks hipStreamEndCapture?  ("tried and dropped (i)": torch.cuda.graph around the
step segfaulted in round 3 on this ROCm 7.2 / torch 2.10 build; the cause was never isolated.)  Answer (round 5): none of
them -- every configuration of the library's launches captures and replays bit-identically; the crash needs a backward
on the legacy default stream before the capture (last configuration), or rocprofv3.

    python tools/experiments/graph_capture_probe.py            # runs every configuration in a child process
    python tools/experiments/graph_capture_probe.py <config>   # one configuration in this process

A configuration = what is captured (fwd | fwd+bwd | step) x switches (chain: fused select chain with its grid barrier and
pinned-memory mailbox | nochain: stand-alone stage kernels | nomailbox: chain without the host_status pointer).
Each child captures, replays twice and compares the replays with an eager run bit for bit."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CONFIGS = ["fwd:nochain", "fwd:nomailbox", "fwd:chain", "fwdbwd:nochain", "fwdbwd:chain", "step:chain",
           # the bench's form of the step, one difference at a time: the noise drawn inside (torch's graph-safe generator
           # path), gradients freed and re-allocated inside the capture, a fresh leaf for the input every step
           "step:chain+ownnoise", "step:chain+setnone", "step:chain+freshleaf", "step:chain+ownnoise+setnone+freshleaf",
           # THE CAUSE (expected: rc -11): one backward on the legacy default stream before the capture.  The parameters'
           # AccumulateGrad nodes are bound to the stream of the first backward and stay alive across steps (the module
           # keeps `attention_bins_beforesoftmax`, whose grad_fn holds the graph); in the capture the autograd engine then
           # synchronises the capture stream with the NULL stream, and hipStreamEndCapture dereferences a null pointer
           # instead of returning hipErrorStreamCaptureImplicit / Unjoined.  Also segfaults: ANY configuration under
           # rocprofv3.  Remedy (torch's documented recipe): every step before the capture on a side stream.
           "step:chain+defaultwarm"]


def one(config: str) -> None:
    import torch
    from samble_amd import ops, sampler_config, synth
    from samble_amd.downsample import DownSampleToken
    what, switch = config.split(":")
    switch, *extras = switch.split("+")
    dev = torch.device("cuda:0")
    B, C, N, M, nb = 32, 128, 2048, 1024, 6
    if os.environ.get("PROBE_SMALL"):
        B, N, M = 4, 512, 256
    mod = DownSampleToken(sampler_config("cls", M=[M, M // 2]), 0)
    wq, wk, wv, tok = synth.sampler_weights(C, nb, 2000)
    with torch.no_grad():
        mod.q_conv.weight.copy_(torch.from_numpy(wq)); mod.k_conv.weight.copy_(torch.from_numpy(wk))
        mod.v_conv.weight.copy_(torch.from_numpy(wv)); mod.bin_tokens.copy_(torch.from_numpy(tok))
    mod = mod.to(dev)
    if switch == "nochain":
        mod._chain_watch.observed = mod._chain_watch.reported = True
    if switch == "nomailbox":
        mod._chain_watch.host_ptr = lambda: None
    opt = torch.optim.SGD(mod.parameters(), lr=0.0)
    x = torch.from_numpy(synth.features(B, C, N, 2001)).to(dev)
    noise = torch.from_numpy(synth.exp1((B * nb, N), 2002)).to(dev)
    g = torch.from_numpy(synth.normal((B, C, M), 2003)).to(dev)
    xin = x.clone().requires_grad_(what != "fwd")

    def step():
        if what == "fwd":
            with torch.no_grad():
                (x_ds, idx), _ = mod(xin, noise=noise)
            return x_ds, idx, None
        opt.zero_grad(set_to_none="setnone" in extras)
        xi = xin
        if "freshleaf" in extras:
            xi = x.detach().requires_grad_(True)
        elif xin.grad is not None:
            xin.grad.zero_()
        (x_ds, idx), _ = mod(xi, noise=None if "ownnoise" in extras else noise)
        x_ds.backward(g)
        if what == "step":
            opt.step()
        return x_ds, idx, xi.grad

    if "defaultwarm" in extras:
        step()      # a backward on the legacy default stream: binds the AccumulateGrad nodes to it
        torch.cuda.synchronize()
    # warm-up on a side stream, as torch.cuda.graph wants it (allocator pools, one-time library initialisation,
    # the first call's boundary state)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    state = [t.clone() for t in mod.bin_boundaries]
    eager = [t.clone() if t is not None else None for t in step()]
    eager_w = mod.q_conv.weight.grad.clone() if what != "fwd" else None
    torch.cuda.synchronize()
    mod.bin_boundaries = [t.clone() for t in state]      # the same boundary state for the captured run
    print(f"[{config}] eager ok; capturing", flush=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    print(f"[{config}] capture ended", flush=True)
    results = []
    for rep in range(2):
        for t, t0 in zip(mod.bin_boundaries, state):      # replays blend into the captured state tensors: reset them
            t.copy_(t0)
        graph.replay()
        torch.cuda.synchronize()
        results.append([t.clone() if t is not None else None for t in out])
    if "ownnoise" in extras:   # every replay draws new noise: nothing to compare bit for bit
        results[1] = results[0]
        eager = results[0]
        eager_w = None
    same = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(results[0], results[1]))
    vs_eager = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(results[0], eager))
    if eager_w is not None:
        vs_eager = vs_eager and torch.equal(mod.q_conv.weight.grad, eager_w)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        graph.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 20
    print(f"[{config}] OK replays identical: {same}; replay == eager: {vs_eager}; ms per call graph {1e3 * t_graph:.4f} "
          f"eager {1e3 * t_eager:.4f}", flush=True)


def main():
    if len(sys.argv) > 1:
        return one(sys.argv[1])
    for config in CONFIGS:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), config], capture_output=True, text=True, timeout=600)
        tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip()][-4:]
        print(f"=== {config}: rc {r.returncode}")
        for l in tail:
            print("   ", l[:300])


if __name__ == "__main__":
    main()

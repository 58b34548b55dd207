"""BatchNorm1d training forward / backward on (B, 128, N): MIOpen (the aten entry nn.BatchNorm1d dispatches to on this
build) against torch's native kernels (cudnn disabled).  GPU time by events over 50 calls.
    python tools/experiments/batchnorm_probe.py"""
import torch

dev = torch.device("cuda:0")


def gpu_us(fn, n=50):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for N in (2048, 1024, 512):
    x = torch.randn(32, 128, N, device=dev)
    g = torch.randn(32, 128, N, device=dev)
    w, b = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev)
    rm, rv = torch.zeros(128, device=dev), torch.ones(128, device=dev)
    for mode in ("miopen", "native"):
        torch.backends.cudnn.enabled = mode == "miopen"
        if mode == "miopen":
            fwd = lambda: torch.miopen_batch_norm(x, w, b, rm, rv, True, 0.1, 1e-5)
            y, m, v = fwd()
            bwd = lambda: torch.ops.aten.miopen_batch_norm_backward(x, g, w, rm, rv, m, v, 1e-5)
        else:
            fwd = lambda: torch.native_batch_norm(x, w, b, rm, rv, True, 0.1, 1e-5)
            y, m, v = fwd()
            bwd = lambda: torch.ops.aten.native_batch_norm_backward(g, x, w, rm, rv, m, v, True, 1e-5, [True, True, True])
        print(f"N={N} {mode:7s} fwd {gpu_us(fwd):7.1f} us   bwd {gpu_us(bwd):7.1f} us", flush=True)

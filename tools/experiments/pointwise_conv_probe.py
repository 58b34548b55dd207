"""Which torch formulation of a bias-free 1x1 Conv1d is cheapest forward + backward on this ROCm build?  (round 5: the
interpolation layers' four convs ran MIOpen's igemm_wrw + two batched transposes per weight gradient.)
    python tools/experiments/pointwise_conv_probe.py"""
import time
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")


def bench(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for (B, C, N, O) in ((32, 256, 2048, 128), (32, 128, 1024, 128), (32, 256, 1024, 128), (32, 128, 512, 128)):
    x = torch.randn(B, C, N, device=dev, requires_grad=True)
    w = torch.randn(O, C, 1, device=dev, requires_grad=True)
    g = torch.randn(B, O, N, device=dev)

    def run(form):
        x.grad = w.grad = None
        if form == "conv1d":
            y = F.conv1d(x, w)
        elif form == "bmm":
            y = torch.bmm(w[:, :, 0].unsqueeze(0).expand(B, -1, -1), x)
        elif form == "matmul":
            y = torch.matmul(w[:, :, 0], x)
        elif form == "einsum":
            y = torch.einsum("oc,bcn->bon", w[:, :, 0], x)
        y.backward(g)
        return y

    ref = run("conv1d").detach().clone(); rw = w.grad.clone(); rx = x.grad.clone()
    for form in ("conv1d", "bmm", "matmul", "einsum"):
        y = run(form)
        err = ((y - ref).abs().max().item(), (w.grad - rw).abs().max().item() / rw.abs().max().item(),
               (x.grad - rx).abs().max().item())
        print(f"B{B} C{C} N{N} O{O} {form:8s} {bench(lambda: run(form)):8.1f} us fwd+bwd   contiguous {y.is_contiguous()}  "
              f"err {err[0]:.1e} {err[1]:.1e} {err[2]:.1e}", flush=True)

"""HBM write / copy bandwidth at the size of the logit map (reference point for attn_stats)."""
import time, torch
dev = torch.device("cuda:0")
for mb in (545, 2180):
    x = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    for name, fn, traffic in (("fill", lambda: x.fill_(1.0), 1), ("copy", lambda: y.copy_(x), 2), ("read-sum", lambda: x.sum(), 1)):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print("%5d MB %-9s %7.1f us  %.2f TB/s" % (mb, name, dt * 1e6, traffic * mb * 1.048576e6 / dt / 1e12))

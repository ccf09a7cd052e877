"""Dev aid: pinned D2H / H2D bandwidth on this box."""
import time, torch
dev = torch.device("cuda:0")
for mb in (1, 8, 64, 512):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device=dev); h = torch.empty(n, dtype=torch.uint8).pin_memory()
    for name, (dst, src) in (("D2H", (h, d)), ("H2D", (d, h))):
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"{name} {mb:4d} MiB: {n/dt/1e9:6.1f} GB/s  ({dt*1e3:.3f} ms)")

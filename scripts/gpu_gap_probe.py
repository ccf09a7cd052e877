"""Dev aid (GPU box): where does the ~100-ms-cadence stall of a bench step come from?  Four tight loops in ONE process that has the
GPU open, 2 s each, every iteration timed; iterations slower than 3x the median are listed with their time stamps:
  A  D2H copy of 8 MB into pinned memory + wait, the waiter SPINS (plain event)              -> the bus / the copy path itself
  B  the same, the waiter SLEEPS (hipEventBlockingSync event: interrupt + wake-up)            -> the interrupt / wake-up path
  C  an empty-ish kernel + wait (spin)                                                        -> the dispatch path
  D  pure CPU: a clock loop in this process                                                   -> the process / its cgroup
python scripts/gpu_gap_probe.py [seconds]"""
import sys
import time

import numpy as np
import torch

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
dev = torch.device("cuda:0")
src = torch.empty(8 << 20, dtype=torch.uint8, device=dev)
dst = torch.empty(8 << 20, dtype=torch.uint8).pin_memory()
small = torch.zeros(1024, device=dev)
torch.cuda.synchronize()


def loop(body, name):
    ts = []
    t_end = time.perf_counter() + secs
    t0 = time.perf_counter()
    while True:
        a = time.perf_counter()
        if a > t_end:
            break
        body()
        ts.append((a - t0, time.perf_counter() - a))
    d = np.array([x[1] for x in ts]) * 1e3
    med = float(np.median(d))
    slow = [(round(t * 1e3, 1), round(float(x), 2)) for (t, _), x in zip(ts, d) if x > max(3 * med, med + 0.5)]
    print(f"{name}: {len(d)} iterations, median {med * 1e3:.0f} us, p99 {np.percentile(d, 99) * 1e3:.0f} us, max {d.max():.2f} ms; "
          f"{len(slow)} slower than max(3x, +0.5 ms) the median, {sum(x for _, x in slow):.1f} ms in all")
    print("   at ms:", slow[:40])


ev_spin = torch.cuda.Event(blocking=False)
ev_sleep = torch.cuda.Event(blocking=True)


def copy_wait(ev):
    dst.copy_(src, non_blocking=True)
    ev.record()
    ev.synchronize()


def kern():
    small.add_(1.0)
    ev_spin.record()
    ev_spin.synchronize()


def cpu():
    t = time.perf_counter()
    while time.perf_counter() - t < 100e-6:
        pass


for rep in range(2):
    loop(lambda: copy_wait(ev_spin), "A copy 8 MB D2H, spin-wait ")
    loop(lambda: copy_wait(ev_sleep), "B copy 8 MB D2H, sleep-wait")
    loop(kern, "C small kernel, spin-wait  ")
    loop(cpu, "D pure CPU 100 us           ")

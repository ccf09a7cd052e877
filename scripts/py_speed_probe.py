"""Dev aid (any box, no GPU needed): how fast does ONE thread run interpreter work on each CPU of this box, right now?
bench.py's `between_calls` (the calling thread's Python around the native calls) is 0.46 ms in one process and 1.1 ms in the next on
the same box while every native phase is the same: is it the CPU the thread happens to sit on (a busy SMT sibling, a slow core)?
A fixed piece of interpreter work (~0.25 ms: list / dict / ctypes traffic like the wrappers') timed 40 times on each of a sample of the
allowed CPUs, the thread pinned there; then 3 s unpinned, with the CPU it was on.   python scripts/py_speed_probe.py [cpus to sample]"""
import ctypes as C
import os
import statistics
import sys
import time


def work():
    a = [i * 3 for i in range(600)]
    d = {i: str(i) for i in range(300)}
    arr = (C.c_uint64 * 96)(*range(96))
    s = 0
    for i in range(96):
        s += arr[i] + len(d[i]) + a[i]
    return s


def timed(n=40):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        work()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts) * 1e3, min(ts) * 1e3


_libc = C.CDLL(None)
allowed = sorted(os.sched_getaffinity(0))
k = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sample = allowed[:: max(1, len(allowed) // k)][:k]
print(f"{len(allowed)} allowed CPUs; sampling {len(sample)}; work() median / min per CPU (ms):")
res = []
for c in sample:
    os.sched_setaffinity(0, {c})
    time.sleep(0.002)
    timed(10)
    med, mn = timed()
    res.append((c, med, mn))
os.sched_setaffinity(0, set(allowed))
meds = sorted(r[1] for r in res)
print("  " + "  ".join(f"cpu{c}:{med:.3f}/{mn:.3f}" for c, med, mn in res))
print(f"  across CPUs: fastest {meds[0]:.3f}  median {statistics.median(meds):.3f}  slowest {meds[-1]:.3f}  (slowest / fastest {meds[-1] / meds[0]:.2f})")
t_end = time.perf_counter() + 3.0
seen = {}
while time.perf_counter() < t_end:
    med, _ = timed(20)
    seen.setdefault(_libc.sched_getcpu(), []).append(med)
print("unpinned, 3 s: " + "  ".join(f"cpu{c}: {len(v)} x {statistics.median(v):.3f}" for c, v in sorted(seen.items())))

"""Dev aid (GPU): does restricting the process to N distinct physical cores (one SMT thread each) of the GPU's NUMA node
change the step time?  And a different number of host workers?
python scripts/affinity_ab.py <n_cores|0> [n_threads]   (0: leave the node's whole CPU list / the default 16 workers)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n_cores = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 0
import numpy as np, torch
from flashgmm_amd import parallel as P
print(P.bind_to_gpu_numa_node(0))
if n_cores:
    allowed = sorted(os.sched_getaffinity(0))
    seen, pick = set(), []
    for c in allowed:
        sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        if sib in seen: continue
        seen.add(sib); pick.append(c)
    os.sched_setaffinity(0, set(pick[:n_cores]))
    print("pinned to", sorted(os.sched_getaffinity(0)))
from flashgmm_amd import GaussianMixtureConditional, _lib, testing as T
dev = torch.device("cuda:0")
lat = [T.make_latent(i) for i in range(48)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
_lib.ctx(0, n_threads)
def step():
    t0 = time.perf_counter()
    res = gmc.compress_batch(ys, ss, ms, ws)
    t1 = time.perf_counter()
    for s in range(2):
        idx = range(s, 48, 2)
        gmc.decompress_batch([res[i][0][0] for i in idx], [res[i][0][1] for i in idx], [res[i][0][2] for i in idx], ss[s::2], ms[s::2], ws[s::2])
    torch.cuda.synchronize()
    return (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3
def cpu_stat():
    try:
        return dict((l.split()[0], int(l.split()[1])) for l in open("/sys/fs/cgroup/cpu.stat"))
    except OSError:
        return {}
for _ in range(5): step()
s0, w0 = cpu_stat(), time.perf_counter()
t = np.array([step() for _ in range(200)])
s1, w1 = cpu_stat(), time.perf_counter()
if s0:
    print(f"timed region: {w1 - w0:.2f} s wall, {(s1['usage_usec'] - s0['usage_usec']) / 1e6:.2f} s CPU = {(s1['usage_usec'] - s0['usage_usec']) / 1e6 / (w1 - w0):.1f} CPUs, "
          f"periods {s1['nr_periods'] - s0['nr_periods']}, throttled {s1['nr_throttled'] - s0['nr_throttled']} ({(s1['throttled_usec'] - s0['throttled_usec']) / 1e3:.0f} ms of thread time)")
print(f"threads {_lib.lib().fgmm_ctx_threads(_lib.ctx(0))} cores {n_cores or 'node'}: encode {np.median(t[:,0]):.3f} decode {np.median(t[:,1]):.3f} step median {np.median(t.sum(1)):.3f} ms (min {t.sum(1).min():.3f})")

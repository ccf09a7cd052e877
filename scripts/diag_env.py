"""Dev aid (GPU box): what the process may use and where its memory lands — CPU affinity, cgroup quota + throttling counters,
NUMA node of the GPU, and (after a few steps of the Kodak batch) on which NUMA nodes the big mappings of this process live.
python scripts/diag_env.py [--bind all|thread|none|other|early] [--threads N] [--steps K]"""
import argparse, os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read(p):
    try:
        return open(p).read().strip()
    except Exception as e:
        return f"<{e.__class__.__name__}>"


def cgroup_cpu():
    out = {}
    for name in ("cpu.max", "cpu.stat", "cpuset.cpus.effective", "cpuset.mems.effective"):
        for base in ("/sys/fs/cgroup", "/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpuset"):
            v = read(os.path.join(base, name))
            if not v.startswith("<"):
                out[name] = v.replace("\n", " | ")
                break
    if "cpu.max" not in out:
        q, p = read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
        out["cfs_quota/period"] = f"{q}/{p}"
    return out


def numa_summary(min_mb=32):
    """big mappings of /proc/self/numa_maps: pages per node"""
    rows = []
    try:
        for line in open("/proc/self/numa_maps"):
            nodes = {int(m.group(1)): int(m.group(2)) for m in re.finditer(r"N(\d+)=(\d+)", line)}
            ps = re.search(r"kernelpagesize_kB=(\d+)", line)
            kb = int(ps.group(1)) if ps else 4
            tot = sum(nodes.values()) * kb / 1024
            if tot >= min_mb:
                what = re.search(r"file=(\S+)", line)
                rows.append((tot, {n: round(c * kb / 1024) for n, c in nodes.items()}, what.group(1) if what else line.split()[1]))
    except Exception as e:
        return [f"numa_maps: {e}"]
    return [f"{t:8.0f} MB  {n}  {w}" for t, n, w in sorted(rows, reverse=True)[:12]]


ap = argparse.ArgumentParser()
ap.add_argument("--bind", default="all", choices=["all", "thread", "none", "other", "early"],
                help="all: every thread of the process, after the first GPU call (bench.py's); thread: the calling thread only (round 2); "
                     "early: before anything touches the GPU; other: to the OTHER node's CPUs; none")
ap.add_argument("--threads", type=int, default=0)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()

print("nproc(os.cpu_count)", os.cpu_count(), " affinity", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:4], "...")
print("cgroup", cgroup_cpu())
print("nodes", read("/sys/devices/system/node/online"), {n: read(f"/sys/devices/system/node/{n}/cpulist") for n in sorted(os.listdir("/sys/devices/system/node")) if re.fullmatch(r"node\d+", n)})
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib, parallel as P
from tests import synth as T
if a.bind == "early":
    print("bind early:", P.bind_to_gpu_numa_node(0))
p = torch.cuda.get_device_properties(0)
print("gpu", p.name, f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0", "numa", read(f"/sys/bus/pci/devices/{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0/numa_node"))
bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
try:
    print("kfd topology says", P.gpu_pci_address(0), "runtime says", bdf)
except LookupError as e:
    print("kfd topology:", e)
torch.zeros(1, device="cuda:0").item()  # the runtime's threads exist now
if a.bind == "all":
    print("bind:", P.bind_to_gpu_numa_node(0))
elif a.bind == "thread":
    print("bind:", P.bind_to_gpu_numa_node(0, all_threads=False))
elif a.bind == "other":
    node = int(read(f"/sys/bus/pci/devices/{bdf}/numa_node"))
    cpus = set()
    for part in read(f"/sys/devices/system/node/node{1 - node}/cpulist").split(","):
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    for tid in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(tid), cpus)
        except OSError:
            pass
    print("bound to the OTHER node", 1 - node, len(cpus), "CPUs")
print("threads of the process:", len(os.listdir("/proc/self/task")), " this thread's affinity:", len(os.sched_getaffinity(0)))
dev = torch.device("cuda:0")
if a.threads:
    _lib.ctx(0, a.threads)
print("host threads", _lib.lib().fgmm_ctx_threads(_lib.ctx(0)))
lat = [T.make_latent(i) for i in range(48)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")


def step(codec=True):
    t0 = time.perf_counter()
    res = gmc.compress_batch(ys, ss, ms, ws)
    t1 = time.perf_counter()
    td = []
    if codec:
        for s in range(2):
            idx = range(s, 48, 2)
            gmc.decompress_batch([res[i][0][0] for i in idx], [res[i][0][1] for i in idx], [res[i][0][2] for i in idx], ss[s::2], ms[s::2], ws[s::2])
            td.append(time.perf_counter())
    else:
        gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
        td.append(time.perf_counter())
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return [(t1 - t0) * 1e3] + [(b - a_) * 1e3 for a_, b in zip([t1] + td[:-1], td)] + [(t2 - t0) * 1e3]


import gc
gc.disable()
for codec in (True, False):
    for _ in range(5):
        step(codec)
    c0 = cgroup_cpu().get("cpu.stat")
    t = np.array([step(codec) for _ in range(a.steps)])
    c1 = cgroup_cpu().get("cpu.stat")
    names = ["encode", "dec anchors", "dec non-anchors", "step"] if codec else ["encode", "decode", "step"]
    print("codec" if codec else "all-at-once", " ".join(f"{n} med {np.median(t[:, i]):.3f} min {t[:, i].min():.3f} p90 {np.percentile(t[:, i], 90):.3f}" for i, n in enumerate(names)))
    print("  cpu.stat before", c0)
    print("  cpu.stat after ", c1)
print("numa_maps (big mappings):")
for r in numa_summary():
    print("  ", r)

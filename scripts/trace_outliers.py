"""Dev aid (GPU): many decode calls of 24 bitstreams with the job timeline on (option trace = 2); prints the timeline of the calls
that took much longer than the median - where does a stalled call lose its time?   python scripts/trace_outliers.py [calls]"""
import os, sys, io, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
dev = torch.device("cuda:0")
lat = [T.make_latent(i) for i in range(48)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
res = gmc.compress_batch(ys, ss, ms, ws)
idx = range(0, 48, 2)
args = ([res[i][0][0] for i in idx], [res[i][0][1] for i in idx], [res[i][0][2] for i in idx], ss[0::2], ms[0::2], ws[0::2])
for _ in range(5):
    gmc.decompress_batch(*args)
_lib.set_option(0, "trace", 2)
n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
# stderr of the native library goes to a file per call
tmp = tempfile.TemporaryFile(mode="w+b")
saved = os.dup(2)
times, logs = [], []
for c in range(n_calls):
    tmp.seek(0); tmp.truncate()
    os.dup2(tmp.fileno(), 2)
    t0 = time.perf_counter()
    gmc.compress_batch(ys, ss, ms, ws) if c % 3 == 2 else gmc.decompress_batch(*args)
    dt = (time.perf_counter() - t0) * 1e3
    os.dup2(saved, 2)
    if c % 3 != 2:
        tmp.seek(0)
        times.append(dt); logs.append(tmp.read().decode(errors="replace"))
times = np.array(times)
med = np.median(times)
print(f"{len(times)} decode calls: median {med:.3f} ms, p90 {np.percentile(times, 90):.3f}, max {times.max():.3f}; calls over 1.5 x median: {(times > 1.5 * med).sum()}")
for k in np.argsort(-times)[:4]:
    if times[k] < 1.5 * med: break
    print(f"==== call {k}: {times[k]:.3f} ms")
    lines = [ln for ln in logs[k].splitlines() if "fgmm decode" in ln]
    items = [ln for ln in lines if " item " in ln]
    for ln in lines:
        if " item " not in ln: print(ln)
    # the slowest items by end time
    def endt(ln):
        try: return float(ln.split("..")[1].split()[0])
        except Exception: return 0
    for ln in sorted(items, key=endt)[-5:]: print(ln)

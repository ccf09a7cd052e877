"""Dev aid (GPU): A/B of context options inside ONE process on ONE box (boxes and runs differ by several percent): every
configuration is timed in every round, medians over rounds.   python scripts/ab_options.py [codec|all] "name=value,..." ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
dev = torch.device("cuda:0")
schedule = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("codec", "all") else "codec"
cfgs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[2:]] or [{}]
lat = [T.make_latent(i) for i in range(48)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=int(os.environ.get("CKPT", "0")))
defaults = {k: _lib.get_option(0, k) for c in cfgs for k in c}
def step():
    t0 = time.perf_counter()
    res = gmc.compress_batch(ys, ss, ms, ws)
    t1 = time.perf_counter()
    if schedule == "codec":
        for s in range(2):
            idx = range(s, 48, 2)
            gmc.decompress_batch([res[i][0][0] for i in idx], [res[i][0][1] for i in idx], [res[i][0][2] for i in idx], ss[s::2], ms[s::2], ws[s::2])
    else:
        gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3
import gc
gc.disable()
times = [[] for _ in cfgs]
for rnd in range(int(os.environ.get("ROUNDS", "8"))):
    for ci, c in enumerate(cfgs):
        for k, v in defaults.items(): _lib.set_option(0, k, v)
        for k, v in c.items(): _lib.set_option(0, k, v)
        step()
        for _ in range(5): times[ci].append(step())
for c, t in zip(cfgs, times):
    e, d = np.array(t).T
    print(f"{schedule:6s} {str(c):60s} encode {np.median(e):6.3f}  decode {np.median(d):6.3f}  step median {np.median(e + d):6.3f} ms  (p25 {np.percentile(e + d, 25):.3f}, min {np.min(e + d):.3f}, p90 {np.percentile(e + d, 90):.3f}, max {np.max(e + d):.3f}, mean {np.mean(e + d):.3f})")

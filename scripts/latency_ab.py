"""Dev aid (GPU): one image (two bitstreams) encode + decode latency, as the codec schedules it (one decode call per
bitstream) and all at once, against context options.   python scripts/latency_ab.py "name=value,..." ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib, testing as T
dev = torch.device("cuda:0")
cfgs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[1:]] or [{}]
lat = [T.make_latent(i) for i in range(2)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
defaults = {k: _lib.get_option(0, k) for c in cfgs for k in c}
def step(codec):
    t0 = time.perf_counter()
    res = gmc.compress_batch(ys, ss, ms, ws)
    t1 = time.perf_counter()
    if codec:
        for s in range(2):
            gmc.decompress_batch([res[s][0][0]], [res[s][0][1]], [res[s][0][2]], ss[s:s + 1], ms[s:s + 1], ws[s:s + 1])
    else:
        gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
    torch.cuda.synchronize()
    return (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3
res = {}
for rnd in range(int(os.environ.get("ROUNDS", "6"))):
    for ci, c in enumerate(cfgs):
        for k, v in defaults.items(): _lib.set_option(0, k, v)
        for k, v in c.items(): _lib.set_option(0, k, v)
        for codec in (True, False):
            step(codec)
            for _ in range(10): res.setdefault((ci, codec), []).append(step(codec))
for ci, c in enumerate(cfgs):
    for codec in (True, False):
        e, d = np.array(res[(ci, codec)]).T
        print(f"{'as codec  ' if codec else 'all at once'} {str(c):40s} encode {np.median(e):6.3f}  decode {np.median(d):6.3f}  total median {np.median(e + d):6.3f} ms (min {np.min(e + d):.3f})")

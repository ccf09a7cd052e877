"""Dev aid (GPU): how much of a compress_batch / decompress_batch call is Python glue (everything except the native call)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
dev = torch.device("cuda:0")
lat = [T.make_latent(i) for i in range(48)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=int(os.environ.get("CKPT", "0")))
L = _lib.lib()
native = {"c": [], "d": []}
for name, key in (("fgmm_gmc_compress_batch", "c"), ("fgmm_gmc_decompress_batch", "d")):
    f = getattr(L, name)
    def wrap(*a, _f=f, _k=key):
        t0 = time.perf_counter(); r = _f(*a); native[_k].append(time.perf_counter() - t0); return r
    setattr(L, name, wrap)
import gc; gc.disable()
tot = {"c": [], "d": []}
for it in range(30):
    t0 = time.perf_counter(); res = gmc.compress_batch(ys, ss, ms, ws); tot["c"].append(time.perf_counter() - t0)
    for s in range(2):
        idx = range(s, 48, 2)
        t0 = time.perf_counter()
        gmc.decompress_batch([res[i][0][0] for i in idx], [res[i][0][1] for i in idx], [res[i][0][2] for i in idx], ss[s::2], ms[s::2], ws[s::2])
        tot["d"].append(time.perf_counter() - t0)
for k, nm in (("c", "compress (48)"), ("d", "decompress (24)")):
    t, n = np.array(tot[k][6:]) * 1e3, np.array(native[k][6:]) * 1e3
    print(f"{nm:16s} call {np.median(t):6.3f} ms   native {np.median(n):6.3f} ms   python glue {np.median(t - n):6.3f} ms")

"""Dev aid (GPU): ONE image (two bitstreams) as the codec schedules it — encode call, anchor decode call, non-anchor decode
call: Python total, native call, Python glue; then the native phases of one call of each kind (option trace=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
dev = torch.device("cuda:0")
ck_stride = 0
for kv in sys.argv[1:]:
    if kv.startswith("ckpt="):
        ck_stride = int(kv.split("=")[1])
    else:
        _lib.set_option(0, kv.split("=")[0], int(kv.split("=")[1]))
lat = [T.make_latent(i) for i in range(2)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=ck_stride)
L = _lib.lib()
native = {"c": [], "d": []}
for name, key in (("fgmm_gmc_compress_batch", "c"), ("fgmm_gmc_decompress_batch", "d")):
    f = getattr(L, name)
    def wrap(*a, _f=f, _k=key):
        t0 = time.perf_counter(); r = _f(*a); native[_k].append(time.perf_counter() - t0); return r
    setattr(L, name, wrap)
import gc; gc.disable()
tot = {"c": [], "d0": [], "d1": [], "all": []}
def image():
    t0 = time.perf_counter(); res = gmc.compress_batch(ys, ss, ms, ws); t1 = time.perf_counter()
    tot["c"].append(t1 - t0)
    for s in range(2):
        ta = time.perf_counter()
        gmc.decompress_batch([res[s][0][0]], [res[s][0][1]], [res[s][0][2]], ss[s:s + 1], ms[s:s + 1], ws[s:s + 1])
        tot[f"d{s}"].append(time.perf_counter() - ta)
    torch.cuda.synchronize()
    tot["all"].append(time.perf_counter() - t0)
    return res
for _ in range(40): image()
med = lambda v: float(np.median(np.array(v[10:]) * 1e3))
nc = np.array(native["c"][10:]) * 1e3; nd = np.array(native["d"][20:]) * 1e3
print(f"one image as the codec schedules it: {med(tot['all']):.3f} ms = encode {med(tot['c']):.3f} (native {np.median(nc):.3f}) + anchors {med(tot['d0']):.3f} (native {np.median(nd[0::2]):.3f}) + non-anchors {med(tot['d1']):.3f} (native {np.median(nd[1::2]):.3f})")
print(f"symbols per bitstream: {[int(r[0][2].sum()) * 768 for r in image()]}")
_lib.set_option(0, "trace", 2)
image()

"""Dev aid (GPU): where a segdec_kernel wave spends its cycles.  Needs a library built with -DFGMM_SEG_PROF:
    OUT=$PWD/ab/lib_segprof.so bash flashgmm_amd/csrc/build.sh -DFGMM_SEG_PROF
    FGMM_LIB=ab/lib_segprof.so python scripts/segprof.py [stride] [images]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device("cuda:0")
L = _lib.lib()
L.fgmm_debug_segprof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lat = [T.make_latent(i) for i in range(nimg)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=stride)
_lib.set_option(0, "gpu_decode", 1)
res = gmc.compress_batch(ys, ss, ms, ws)
args = ([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
for _ in range(3): gmc.decompress_batch(*args)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8)()
L.fgmm_debug_segprof(buf, 1)
n = 5
t0 = time.perf_counter()
for _ in range(n): gmc.decompress_batch(*args)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n * 1e3
L.fgmm_debug_segprof(buf, 0)
a, b, m, c, syms, batches, total, waves = (int(v) / n for v in buf)
print(f"stride {stride} items {nimg}: call {dt:.3f} ms; waves {waves:.0f}, batches/wave {batches / waves:.1f}, symbols/batch {syms / batches:.1f}")
print(f"cycles per symbol: producer A {a / syms:.0f}  B+per-latent {b / syms:.0f} | consumer C {c / syms:.0f} | both, at the barrier {m / syms:.0f} | whole segment {total / syms:.0f}   (per segment {total / waves / 1e3:.0f} k cycles)")

if hasattr(L, "fgmm_debug_segtimes") and len(sys.argv) > 3:
    tb = (C.c_ulonglong * (2 * 16384))()
    L.fgmm_debug_segtimes(tb)
    t = np.array(tb, dtype=np.float64).reshape(-1, 2)[: int(min(waves, 16384))]
    t0 = t[:, 0].min()
    beg, end = (t[:, 0] - t0) / 1e5, (t[:, 1] - t0) / 1e5   # ms: wall_clock64 counts at 100 MHz
    dur = end - beg
    print(f"segments: duration ms min {dur.min():.3f} p10 {np.percentile(dur, 10):.3f} median {np.median(dur):.3f} p90 {np.percentile(dur, 90):.3f} max {dur.max():.3f}; last start {beg.max():.3f}, last end {end.max():.3f}")
    order = np.argsort(beg)
    print("start time of every 500th segment in start order:", np.round(beg[order][::500], 3).tolist())
    print("end time of every 250th segment in launch order:", np.round(end[::250], 3).tolist())
    print("start / duration of every 250th segment in launch order:", [(round(float(b), 3), round(float(x), 3)) for b, x in zip(beg[::250], dur[::250])])
    busy = [(int(((beg <= x) & (end > x)).sum())) for x in np.arange(0, end.max(), 0.1)]
    print("segments in flight every 0.1 ms:", busy)

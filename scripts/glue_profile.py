"""Dev aid (GPU): cProfile of the Python glue around one batched encode call of the ELIC-4K workload (80 separate tensors:
the list path of GaussianMixtureConditional.compress_batch / decompress_batch).   python scripts/glue_profile.py [images]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from flashgmm_amd import GaussianMixtureConditional, _lib
dev = torch.device("cuda:0")
images = int(sys.argv[1]) if len(sys.argv) > 1 else 8
workload = sys.argv[2] if len(sys.argv) > 2 else "elic4k"   # kodak24: one stacked tensor per operand (the other glue path)
host, devt, pix = B.make_workload(0, images, dev, workload, workload == "elic4k")
if workload == "kodak24":
    ys, ss, ms, ws = (torch.cat([t[k] for t in devt]) for k in range(4))
else:
    ys, ss, ms, ws = ([t[k] for t in devt] for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=int(os.environ.get("CKPT", "1024")))
for _ in range(2):
    res = gmc.compress_batch(ys, ss, ms, ws)
native = []
f = _lib.lib().fgmm_gmc_compress_batch
def wrap(*a):
    t0 = time.perf_counter(); r = f(*a); native.append(time.perf_counter() - t0); return r
_lib.lib().fgmm_gmc_compress_batch = wrap
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    res = gmc.compress_batch(ys, ss, ms, ws)
pr.disable()
tot = (time.perf_counter() - t0) / 5 * 1e3
print(f"compress_batch of {len(ys)} items: {tot:.2f} ms per call under the profiler, native {sum(native) / 5 * 1e3:.2f}")
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
# ... and the decode glue (all bitstreams in one call)
fd = _lib.lib().fgmm_gmc_decompress_batch
nd = []
def wrapd(*a):
    t0 = time.perf_counter(); r = fd(*a); nd.append(time.perf_counter() - t0); return r
_lib.lib().fgmm_gmc_decompress_batch = wrapd
args = ([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
gmc.decompress_batch(*args)
nd.clear()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    gmc.decompress_batch(*args)
pr.disable()
print(f"decompress_batch: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call under the profiler, native {sum(nd) / 5 * 1e3:.2f}")
pstats.Stats(pr).sort_stats("tottime").print_stats(10)

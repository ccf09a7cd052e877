"""Dev aid: A/B of decode-pipeline settings inside ONE process on ONE box (boxes differ by several percent):
every configuration is timed in every round, medians are reported."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flashgmm_amd import GaussianMixtureConditional, _lib, testing as T

dev = torch.device("cuda:0")
_lib.ctx(0, 16)
devt = [[torch.from_numpy(a).to(dev) for a in T.make_latent(i)] for i in range(48)]
ys, ss, ms, ws = ([t[k] for t in devt] for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
configs = [dict(c.split("=") for c in arg.split(",")) if arg != "default" else {} for arg in sys.argv[1:]] or [{}]
res = gmc.compress_batch(ys, ss, ms, ws)
args = ([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
keys = sorted({k for c in configs for k in c})
times = [[] for _ in configs]
for rnd in range(12):
    for ci, c in enumerate(configs):
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(c)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = gmc.decompress_batch(*args)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if rnd >= 2:
            times[ci].append(1e3 * (t1 - t0))
for c, t in zip(configs, times):
    print(f"{statistics.median(t):7.3f} ms (min {min(t):7.3f})  {c or 'default'}")

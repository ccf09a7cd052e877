"""Dev aid (GPU): phase timeline of one encode call of 48 bitstreams.   python scripts/trace_encode.py [name=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
dev = torch.device("cuda:0")
lat = [T.make_latent(i) for i in range(48)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
for kv in sys.argv[1:]:
    _lib.set_option(0, kv.split("=")[0], int(kv.split("=")[1]))
for _ in range(4):
    gmc.compress_batch(ys, ss, ms, ws)
_lib.set_option(0, "trace", 2)
for _ in range(3):
    t0 = time.perf_counter()
    gmc.compress_batch(ys, ss, ms, ws)
    sys.stderr.write(f"---- python total {(time.perf_counter() - t0) * 1e3:.3f} ms\n")

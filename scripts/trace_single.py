"""Dev aid (GPU): phase timeline of a ONE-bitstream decode call (what the one-image latency is made of)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flashgmm_amd import GaussianMixtureConditional, _lib, testing as T
dev = torch.device("cuda:0")
lat = [T.make_latent(i) for i in range(2)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
for kv in sys.argv[1:]:
    _lib.set_option(0, kv.split("=")[0], int(kv.split("=")[1]))
res = gmc.compress_batch(ys, ss, ms, ws)
args = ([res[0][0][0]], [res[0][0][1]], [res[0][0][2]], ss[0:1], ms[0:1], ws[0:1])
for _ in range(5):
    gmc.decompress_batch(*args)
_lib.set_option(0, "trace", 2)
for _ in range(3):
    t0 = time.perf_counter()
    gmc.decompress_batch(*args)
    torch.cuda.synchronize()
    sys.stderr.write(f"---- python total {(time.perf_counter() - t0) * 1e3:.3f} ms\n")

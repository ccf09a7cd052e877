"""Dev aid: how much of a compress_batch / decompress_batch call is Python glue (everything except the native call)."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flashgmm_amd import GaussianMixtureConditional, _lib, testing as T
dev = torch.device("cuda:0")
_lib.ctx(0, 16)
devt = [[torch.from_numpy(a).to(dev) for a in T.make_latent(i)] for i in range(48)]
ys, ss, ms, ws = ([t[k] for t in devt] for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
L = _lib.lib()
native = {"enc": [], "dec": []}
orig_c, orig_d = L.fgmm_gmc_compress_batch, L.fgmm_gmc_decompress_batch
class Wrap:
    def __init__(self, f, key): self.f, self.key = f, key
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.f(*a); native[self.key].append(time.perf_counter() - t0); return r
L.fgmm_gmc_compress_batch = Wrap(orig_c, "enc"); L.fgmm_gmc_decompress_batch = Wrap(orig_d, "dec")
form = sys.argv[1] if len(sys.argv) > 1 else "list"
if form == "stacked":
    ys, ss, ms, ws = (torch.cat(t) for t in (ys, ss, ms, ws))
enc, dec = [], []
for it in range(14):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = gmc.compress_batch(ys, ss, ms, ws)
    t1 = time.perf_counter()
    a = ([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
    t2 = time.perf_counter()
    outs = gmc.decompress_batch(*a)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    enc.append(t1 - t0); dec.append(t3 - t2)
m = lambda v: 1e3 * statistics.median(v[4:])
print(form); print(f"encode: wall {m(enc):.3f} ms  native {m(native['enc']):.3f} ms  glue {m(enc)-m(native['enc']):.3f} ms")
print(f"decode: wall {m(dec):.3f} ms  native {m(native['dec']):.3f} ms  glue {m(dec)-m(native['dec']):.3f} ms")

"""Dev aid: run the symtab batch a few times and the saturation selftest, for PMC comparison under rocprofv3."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flashgmm_amd import GaussianMixtureConditional, _lib, testing as T
dev = torch.device("cuda:0")
devt = [[torch.from_numpy(a).to(dev) for a in T.make_latent(i)] for i in range(48)]
ys, ss, ms, ws = ([t[k] for t in devt] for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
for _ in range(3): gmc.compress_batch(ys, ss, ms, ws)
bad = C.c_uint64()
_lib.lib().fgmm_selftest_saturation(_lib.ctx(0), 0, C.byref(bad))

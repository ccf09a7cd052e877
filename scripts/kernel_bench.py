"""Dev aid: mean duration of the symtab / cdftab kernels on the kodak24 workload (HIP events in the library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib, testing as T

dev = torch.device("cuda:0")
devt = [[torch.from_numpy(a).to(dev) for a in T.make_latent(i)] for i in range(48)]
ys, ss, ms, ws = ([t[k] for t in devt] for k in range(4))
_lib.set_profiling(0, True)
modes = sys.argv[1:] or ["polya", "as", "logistic"]
for mode in modes:
    gmc = GaussianMixtureConditional(K=4, mode=mode)
    sym, tab = [], []
    for it in range(12):
        res = gmc.compress_batch(ys, ss, ms, ws)
        sym.append(_lib.kernel_ms(0, 0))
        if it % 4 == 3:
            gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
            tab.append(_lib.kernel_ms(0, 1))
    n = sum(int(r[0][2].sum()) * 768 for r in res)
    s = float(np.median(sym[2:]))
    print(f"{mode:9s} symtab {s*1e3:7.1f} us  -> {n*56/s/1e6:7.1f} GB/s ({n*56/s/1e6/8000:.3f} of 8 TB/s)   cdftab(all groups, overlapped) {np.median(tab):.3f} ms")

"""Dev aid: wall time of the encode and decode halves of a bench step, and table sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib, testing as T

dev = torch.device("cuda:0")
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 0
_lib.ctx(0, nthreads)
devt = [[torch.from_numpy(a).to(dev) for a in T.make_latent(i)] for i in range(48)]
ys, ss, ms, ws = ([t[k] for t in devt] for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
_lib.set_profiling(0, True)
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = gmc.compress_batch(ys, ss, ms, ws)
    t1 = time.perf_counter()
    outs = gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: encode {1e3*(t1-t0):.2f} ms  decode {1e3*(t2-t1):.2f} ms   kernels: symtab {_lib.kernel_ms(0,0):.3f} cdftab count {_lib.kernel_ms(0,1):.3f} fill {_lib.kernel_ms(0,3):.3f} qs {_lib.kernel_ms(0,2):.3f}  threads {_lib.lib().fgmm_ctx_threads(_lib.ctx(0))}")

"""Dev aid (GPU): segdec_kernel's launch time (HIP events around the launch: fgmm_ctx_kernel_ms 3) and the decode call's wall time for a
codec call of Kodak halves, for the library FGMM_LIB names.    python scripts/segdec_ab.py [stride] [halves ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
counts = [int(a) for a in sys.argv[2:]] or [24, 48, 2]
dev = torch.device("cuda:0")
lat = [T.make_latent(i) for i in range(max(counts))]
_lib.set_option(0, "gpu_decode", 1)
_lib.set_profiling(0, True)
for nimg in counts:
    ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat[:nimg]]).to(dev) for k in range(4))
    gmc = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=stride)
    res = gmc.compress_batch(ys, ss, ms, ws)
    args = ([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
    ref = [r[1] for r in res]
    for _ in range(3): out = gmc.decompress_batch(*args)
    assert all(torch.equal(a, b) for a, b in zip(out, ref))
    torch.cuda.synchronize()
    k, wall = [], []
    for _ in range(15):
        t0 = time.perf_counter(); gmc.decompress_batch(*args); wall.append((time.perf_counter() - t0) * 1e3)
        k.append(_lib.kernel_ms(0, 3))
    print(f"{os.environ.get('FGMM_LIB', 'default'):28s} stride {stride} halves {nimg:3d}: segdec launch median {np.median(k):.3f} ms (min {min(k):.3f}), call {np.median(wall):.3f} ms", flush=True)

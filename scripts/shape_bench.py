"""Dev aid: symtab kernel rate for different item shapes (is the per-symbol rate shape-independent?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib, testing as T
dev = torch.device("cuda:0")
_lib.set_profiling(0, True)
gmc = GaussianMixtureConditional(K=4, mode="polya")
def run(name, shapes):
    devt = [[torch.from_numpy(a).to(dev) for a in T.make_latent(i, M=M, h=h, w=w)] for i, (M, h, w) in enumerate(shapes)]
    ys, ss, ms, ws = ([t[k] for t in devt] for k in range(4))
    sym = []
    for it in range(8):
        res = gmc.compress_batch(ys, ss, ms, ws)
        sym.append(_lib.kernel_ms(0, 0))
    n = sum(int(r[0][2].sum()) * y.shape[2] * y.shape[3] for r, y in zip(res, ys))
    s = float(np.median(sym[2:]))
    print(f"{name:34s} {n/1e6:6.2f} M symbols  symtab {s*1e3:7.1f} us  {n/s/1e6:6.1f} G sym/s")
import math
for h, w in [(32, 24), (32, 32), (64, 48), (64, 64), (64, 96), (128, 64), (96, 128), (136, 120), (128, 128), (256, 128)]:
    cnt = max(1, round(6.2e6 / (192 * h * w)))
    run(f"{cnt} x [192,{h},{w}] hw={h*w}", [(192, h, w)] * cnt)
run("1 x [48,256,256]", [(48, 256, 256)])
run("48 x [24,64,96]", [(24, 64, 96)] * 48)

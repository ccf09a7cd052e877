"""Dev aid (GPU): per-step times of the first steps after a pause (the bench checks its warm-up results before the timed
region: do the first timed steps pay for the pause?).  python scripts/step_drift.py [pause_ms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
dev = torch.device("cuda:0")
pause = float(sys.argv[1]) / 1e3 if len(sys.argv) > 1 else 0.15
lat = [T.make_latent(i) for i in range(48)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
def step():
    t0 = time.perf_counter()
    res = gmc.compress_batch(ys, ss, ms, ws)
    for s in range(2):
        idx = range(s, 48, 2)
        gmc.decompress_batch([res[i][0][0] for i in idx], [res[i][0][1] for i in idx], [res[i][0][2] for i in idx], ss[s::2], ms[s::2], ws[s::2])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
import gc; gc.disable()
rows = []
for rep in range(6):
    for _ in range(3): step()
    time.sleep(pause)
    rows.append([step() for _ in range(20)])
a = np.array(rows)
print("pause %.0f ms; median over 6 repetitions of step k after the pause:" % (pause * 1e3))
print(" ".join(f"{v:.2f}" for v in np.median(a, axis=0)))
print("mean of the 20: %.3f   mean of steps 5..19: %.3f" % (np.median(a.mean(1)), np.median(a[:, 5:].mean(1))))

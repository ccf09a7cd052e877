"""Dev aid: per-step wall times of the bench workload inside one process (how much of the run-to-run spread is per step?)."""
import gc, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flashgmm_amd import GaussianMixtureConditional, _lib, testing as T
from flashgmm_amd import parallel as P
print(P.bind_to_gpu_numa_node(0))
dev = torch.device("cuda:0")
devt = [[torch.from_numpy(a).to(dev) for a in T.make_latent(i)] for i in range(48)]
ys, ss, ms, ws = (torch.cat([t[k] for t in devt]) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
def step():
    res = gmc.compress_batch(ys, ss, ms, ws)
    return gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
for _ in range(5): step()
gc.collect(); gc.freeze(); gc.disable()
ts = []
for _ in range(80):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("steps: " + " ".join(f"{t:.2f}" for t in ts))
print(f"median {statistics.median(ts):.3f}  mean {statistics.mean(ts):.3f}  min {min(ts):.3f}  max {max(ts):.3f}  p90 {sorted(ts)[71]:.3f}")

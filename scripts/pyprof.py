"""Dev aid: where the Python-side glue of compress_batch / decompress_batch spends its time."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flashgmm_amd import GaussianMixtureConditional, testing as T
dev = torch.device("cuda:0")
devt = [[torch.from_numpy(a).to(dev) for a in T.make_latent(i)] for i in range(48)]
ys, ss, ms, ws = ([t[k] for t in devt] for k in range(4))
if len(sys.argv) > 1 and sys.argv[1] == "stacked":
    ys, ss, ms, ws = (torch.cat(t) for t in (ys, ss, ms, ws))
gmc = GaussianMixtureConditional(K=4, mode="polya")
def step():
    res = gmc.compress_batch(ys, ss, ms, ws)
    return gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
for _ in range(3): step()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)

"""Dev aid (GPU): cProfile of the Python side of compress_batch / decompress_batch (48 / 24 stacked items)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flashgmm_amd import GaussianMixtureConditional, testing as T
dev = torch.device("cuda:0")
lat = [T.make_latent(i) for i in range(48)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
res = gmc.compress_batch(ys, ss, ms, ws)
def run(n):
    for _ in range(n):
        r = gmc.compress_batch(ys, ss, ms, ws)
        for s in range(2):
            idx = range(s, 48, 2)
            gmc.decompress_batch([r[i][0][0] for i in idx], [r[i][0][1] for i in idx], [r[i][0][2] for i in idx], ss[s::2], ms[s::2], ws[s::2])
run(5)
pr = cProfile.Profile(); pr.enable(); run(40); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

"""Dev aid (GPU): phase / job timeline of one decode call of 24 bitstreams (the codec schedule's unit).
python scripts/trace_decode.py [name=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
dev = torch.device("cuda:0")
lat = [T.make_latent(i) for i in range(48)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
for kv in sys.argv[1:]:
    _lib.set_option(0, kv.split("=")[0], int(kv.split("=")[1]))
res = gmc.compress_batch(ys, ss, ms, ws)
idx = range(0, 48, 2)
args = ([res[i][0][0] for i in idx], [res[i][0][1] for i in idx], [res[i][0][2] for i in idx], ss[0::2], ms[0::2], ws[0::2])
for _ in range(4):
    gmc.decompress_batch(*args)
_lib.set_option(0, "trace", 2)
for _ in range(3):
    gmc.decompress_batch(*args)
    sys.stderr.write("----\n")

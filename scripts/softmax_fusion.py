"""Dev aid (GPU): what fusing the softmax over K into the kernels saves on the encode side of the kodak24 batch.
   un-fused: torch.softmax over the [N, K, M, h, w] view of the logits (reads 16 B, writes 16 B per latent), then compress
             (the symtab kernel reads the pi plane: 16 of its 56 B per symbol)
   fused   : compress(..., weights_are_logits=True): the logits are read once, by the symtab kernel"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
dev = torch.device("cuda:0")
lat = [T.make_latent(i) for i in range(48)]
ys, ss, ms = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(3))
lg = torch.log(torch.cat([torch.from_numpy(l[3]) for l in lat]).to(dev))  # logits whose softmax is (about) the workload's pi
gmc = GaussianMixtureConditional(K=4, mode="polya")
_lib.set_profiling(0, True)
N, KM, h, w = lg.shape
def unfused():
    pi = torch.softmax(lg.view(N, 4, KM // 4, h, w), dim=1).view(N, KM, h, w)
    return gmc.compress_batch(ys, ss, ms, pi)
def fused():
    return gmc.compress_batch(ys, ss, ms, lg, weights_are_logits=True)
for name, fn in (("un-fused", unfused), ("fused", fused), ("un-fused", unfused), ("fused", fused)):
    for _ in range(3): r = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ks = []
    for _ in range(10):
        r = fn(); ks.append(_lib.kernel_ms(0, 0))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    n = sum(int(x[0][2].sum()) * h * w for x in r)
    print(f"{name:9s} encode step {dt*1e3:6.3f} ms   symtab kernel {np.median(ks)*1e3:6.1f} us ({n*56/np.median(ks)/1e6/8000:.3f} of 8 TB/s at 56 B/symbol)")
sm = []
for _ in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); pi = torch.softmax(lg.view(N, 4, KM // 4, h, w), dim=1); e1.record(); torch.cuda.synchronize(); sm.append(e0.elapsed_time(e1))
print(f"torch.softmax alone: {np.median(sm)*1e3:.1f} us for {lg.numel()*4/1e6:.0f} MB in + out each (the traffic the fusion removes: 32 B per latent)")

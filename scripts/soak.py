"""Dev aid: a few thousand bench steps in one process: host RSS, free GPU memory and step time must stay flat."""
import gc, os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flashgmm_amd import GaussianMixtureConditional, testing as T
dev = torch.device("cuda:0")
devt = [[torch.from_numpy(a).to(dev) for a in T.make_latent(i)] for i in range(48)]
ys, ss, ms, ws = (torch.cat([t[k] for t in devt]) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=int(os.environ.get("CKPT", "0")))  # CKPT=1024: the GPU segment decoder
def rss(): return int(open("/proc/self/statm").read().split()[1]) * 4096 / 2**20
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
t_last = time.perf_counter()
for it in range(n):
    res = gmc.compress_batch(ys, ss, ms, ws)
    outs = gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
    if it % 500 == 0 or it == n - 1:
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        now = time.perf_counter()
        print(f"step {it:5d}  rss {rss():8.1f} MiB  gpu free {free/2**30:7.2f} GiB  {1e3*(now-t_last)/max(1,(500 if it else 1)):6.2f} ms/step", flush=True)
        t_last = now
assert all(torch.equal(o, r[1]) for o, r in zip(outs, res))
print("soak ok")

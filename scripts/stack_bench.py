"""Dev aid: symtab kernel time for the Kodak batch given as 48 separate tensors vs ONE stacked tensor per operand."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib, testing as T
dev = torch.device("cuda:0")
_lib.set_profiling(0, True)
gmc = GaussianMixtureConditional(K=4, mode="polya")
devt = [[torch.from_numpy(a).to(dev) for a in T.make_latent(i)] for i in range(48)]
lists = [[t[k] for t in devt] for k in range(4)]
stacked = [torch.cat(l) for l in lists]
pad = [torch.empty(1 << 20, device=dev) for _ in range(3)]  # shift the next allocations
stacked2 = [torch.cat(l) for l in lists]
def run(name, args):
    sym = []
    for it in range(10):
        gmc.compress_batch(*args); sym.append(_lib.kernel_ms(0, 0))
    print(f"{name:28s} symtab {1e3*float(np.median(sym[2:])):6.1f} us   ptrs {[hex(a.data_ptr() if hasattr(a,'data_ptr') else a[0].data_ptr()) for a in args]}")
for rep in range(2):
    run("list of 48", lists); run("stacked", stacked); run("stacked (other addresses)", stacked2)

"""Dev aid: where compress_batch (stacked form) spends its wall time, by monkey-patching timers around its steps."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib, testing as T
dev = torch.device("cuda:0"); _lib.ctx(0, 16)
devt = [[torch.from_numpy(a).to(dev) for a in T.make_latent(i)] for i in range(48)]
y, s, m, w = (torch.cat([t[k] for t in devt]) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
pc = time.perf_counter
rows = []
for it in range(14):
    torch.cuda.synchronize(); t0 = pc()
    items, keep, N, M, h, ww, d = gmc._stacked_items(y, s, m, w); t1 = pc()
    yq = torch.empty((N, 1, M, h, ww), dtype=torch.float32, device=d); zb = torch.empty((N, M), dtype=torch.int64)
    items["yq_out"] = np.uint64(yq.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(M * h * ww * 4)
    items["zero_bitmap"] = np.uint64(zb.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(M * 8)
    stream = torch.cuda.current_stream(d).cuda_stream; t2 = pc()
    rc = _lib.lib().fgmm_gmc_compress_batch(_lib.ctx(0), stream, C.cast(items.ctypes.data, C.POINTER(_lib.fgmm_item)), N, 0, 1); t3 = pc()
    ptrs, lens, amax = items["bytes"].tolist(), items["bytes_len"].tolist(), items["abs_max"].tolist()
    out = []
    for i, (q, b) in enumerate(zip(yq.unbind(0), zb.unbind(0))):
        data = C.string_at(ptrs[i], lens[i]); _lib.lib().fgmm_free(ptrs[i]); out.append(((data, amax[i], b), q))
    t4 = pc()
    torch.cuda.synchronize(); t5 = pc()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
    del out
med = [1e3 * statistics.median(r[k] for r in rows[4:]) for k in range(5)]
print("items %.3f  alloc+ptrs %.3f  native %.3f  results %.3f  final sync %.3f ms" % tuple(med))

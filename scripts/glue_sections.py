"""Dev aid (GPU): the sections of the Python glue of one stacked compress_batch of 48 Kodak halves (CKPT=stride for checkpointed
streams), timed in place.   CKPT=1024 python scripts/glue_sections.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from flashgmm_amd import GaussianMixtureConditional, _lib, entropy_models as EM
from tests import synth as T
dev = torch.device("cuda:0")
lat = [T.make_latent(i) for i in range(48)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=int(os.environ.get("CKPT", "1024")))
import gc; gc.disable()
self = gmc
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc.setdefault(name, []).append(t - t0); return t
for it in range(30):
    t = time.perf_counter()
    items, keep, N, M, h, w, d = self._stacked_items(ys, ss, ms, ws, 0)
    t = tick("stacked_items", t)
    yq = torch.empty((N, 1, M, h, w), dtype=torch.float32, device=d)
    zb = torch.empty((N, M), dtype=torch.int64)
    items["yq_out"] = np.uint64(yq.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(M * h * w * 4)
    items["zero_bitmap"] = np.uint64(zb.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(M * 8)
    items["ckpt_stride"] = self.checkpoint_stride
    stream = torch.cuda.current_stream(d).cuda_stream
    t = tick("alloc+fields", t)
    rc = _lib.lib().fgmm_gmc_compress_batch(_lib.ctx(d.index if d.index is not None else -1), stream, C.cast(items.ctypes.data, C.POINTER(_lib.fgmm_item)), N, self._mode(), int(self.clamp_scales))
    t = tick("native", t)
    ptrs, lens, amax = items["bytes"].tolist(), items["bytes_len"].tolist(), items["abs_max"].tolist()
    t = tick("tolist", t)
    datas = _lib.take_bytes_many(d.index if d.index is not None else -1, ptrs, lens, EM.CheckpointedBytes if self.checkpoint_stride else None)
    t = tick("take_bytes_many", t)
    cks = EM._take_ckpts_many(d.index if d.index is not None else -1, items["ckpt"].tolist(), items["n_ckpt"].tolist()) if self.checkpoint_stride else None
    out = []
    qs, bs = yq.unbind(0), zb.unbind(0)
    t = tick("unbind", t)
    for i, (q, b) in enumerate(zip(qs, bs)):
        data = datas[i]
        if cks is not None:
            data = EM.CheckpointedBytes._adopt(data, cks[i][0], self.checkpoint_stride, cks[i][1])
        out.append(((data, amax[i], b), q))
    t = tick("ckpt+tuples", t)
for k, v in acc.items():
    print(f"{k:18s} {np.median(v[5:]) * 1e3:7.3f} ms")

# ---- the decode call of 24 halves (the codec schedule's first call), the same way
res = gmc.compress_batch(ys, ss, ms, ws)
idx = list(range(0, 48, 2))
strings, ams, zbs = [res[i][0][0] for i in idx], [res[i][0][1] for i in idx], [res[i][0][2] for i in idx]
sc, me, we = ss[0::2], ms[0::2], ws[0::2]
acc = {}
for it in range(30):
    t = time.perf_counter()
    items, keep, N, M, h, w, d = self._stacked_items(None, sc, me, we, 0)
    t = tick("stacked_items", t)
    zb = torch.stack([z.to("cpu", torch.int64) for z in zbs])
    t = tick("zero bitmaps", t)
    data = [s_ if isinstance(s_, bytes) else bytes(s_) for s_ in strings]
    bufs = (C.c_char_p * N)(*data)
    t = tick("byte pointers", t)
    y_hat = torch.empty((N, 1, M, h, w), dtype=torch.float32, device=d)
    t = tick("torch.empty", t)
    items["bytes"] = np.frombuffer(bufs, dtype=np.uint64)
    items["bytes_len"] = [len(x) for x in data]
    if any(isinstance(x, EM.CheckpointedBytes) for x in data):
        cks = [(x._ckpt_addr, len(x.ckpt), x.ckpt_stride) if isinstance(x, EM.CheckpointedBytes) else (0, 0, 0) for x in data]
        items["ckpt"], items["n_ckpt"], items["ckpt_stride"] = (np.array(c, dtype=np.uint64) for c in zip(*cks))
    items["abs_max"] = np.asarray(ams, dtype=np.int64)
    items["yq_out"] = np.uint64(y_hat.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(M * h * w * 4)
    items["zero_bitmap"] = np.uint64(zb.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(M * 8)
    stream = torch.cuda.current_stream(d).cuda_stream
    t = tick("fields", t)
    rc = _lib.lib().fgmm_gmc_decompress_batch(_lib.ctx(d.index if d.index is not None else -1), stream, C.cast(items.ctypes.data, C.POINTER(_lib.fgmm_item)), N, self._mode(), int(self.clamp_scales))
    t = tick("native", t)
    outs = list(y_hat.unbind(0))
    t = tick("unbind", t)
    del outs, y_hat
    t = tick("free", t)
print("decompress (24):")
for k, v in acc.items():
    print(f"{k:18s} {np.median(v[5:]) * 1e3:7.3f} ms")

"""Dev aid (GPU): the Python glue of the three calls of a Kodak step, timed IN the step loop (the sections of glue_sections.py, but
with the other calls of the step between them: after a 4 ms native call the interpreter's state is cold)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from flashgmm_amd import GaussianMixtureConditional, _lib, entropy_models as EM
from tests import synth as T
dev = torch.device("cuda:0")
lat = [T.make_latent(i) for i in range(48)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
stage_params = [(ss[s::2], ms[s::2], ws[s::2]) for s in range(2)]  # (the caller's: a network's outputs per stage)
import gc; gc.disable()
self = gmc
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc.setdefault(name, []).append(t - t0); return t
L = _lib.lib()
for it in range(40):
    t = time.perf_counter()
    items, keep, N, M, h, w, d = self._stacked_items(ys, ss, ms, ws, 0)
    t = tick("c.stacked_items", t)
    yq = torch.empty((N, 1, M, h, w), dtype=torch.float32, device=d)
    zb = torch.empty((N, M), dtype=torch.int64)
    items["yq_out"] = np.uint64(yq.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(M * h * w * 4)
    items["zero_bitmap"] = np.uint64(zb.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(M * 8)
    items["ckpt_stride"] = 0
    stream = torch.cuda.current_stream(d).cuda_stream
    t = tick("c.alloc+fields", t)
    rc = L.fgmm_gmc_compress_batch(_lib.ctx(0), stream, C.cast(items.ctypes.data, C.POINTER(_lib.fgmm_item)), N, self._mode(), int(self.clamp_scales))
    t = tick("c.NATIVE", t)
    ptrs, lens, amax = items["bytes"].tolist(), items["bytes_len"].tolist(), items["abs_max"].tolist()
    t = tick("c.tolist", t)
    datas = _lib.take_bytes_many(0, ptrs, lens, None)
    t = tick("c.take_bytes_many", t)
    res = EM.CompressedBatch(datas, amax, zb, yq)
    t = tick("c.result", t)
    for s in range(2):
        strings, ams, zbs = res.strings[s::2], res.abs_maxes[s::2], res.zero_bitmaps[s::2]
        sc, me, we = stage_params[s]
        t = tick("d.caller slices", t)
        items, keep, N, M, h, w, d = self._stacked_items(None, sc, me, we, 0)
        t = tick("d.stacked_items", t)
        zb2 = zbs  # (rows of a strided view: no copy)
        t = tick("d.zero bitmaps", t)
        data = [s_ if isinstance(s_, bytes) else bytes(s_) for s_ in strings]
        bufs = (C.c_char_p * N)(*data)
        y_hat = torch.empty((N, 1, M, h, w), dtype=torch.float32, device=d)
        items["bytes"] = np.frombuffer(bufs, dtype=np.uint64)
        items["bytes_len"] = [len(x) for x in data]
        items["abs_max"] = np.asarray(ams, dtype=np.int64)
        items["yq_out"] = np.uint64(y_hat.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(M * h * w * 4)
        items["zero_bitmap"] = np.uint64(zb2.data_ptr()) + np.arange(N, dtype=np.uint64) * np.uint64(zb2.stride(0) * 8)
        stream = torch.cuda.current_stream(d).cuda_stream
        t = tick("d.pointers+fields", t)
        rc = L.fgmm_gmc_decompress_batch(_lib.ctx(0), stream, C.cast(items.ctypes.data, C.POINTER(_lib.fgmm_item)), N, self._mode(), int(self.clamp_scales))
        t = tick("d.NATIVE", t)
        outs = y_hat
        t = tick("d.result", t)
    del outs, y_hat, res, datas, yq
    t = tick("free", t)
tot = 0
for k, v in acc.items():
    m = np.median(v[8 * (2 if k.startswith('d.') else 1):]) * 1e3 * (2 if k.startswith('d.') else 1)
    tot += 0 if "NATIVE" in k else m
    print(f"{k:24s} {m:7.3f} ms per step")
print(f"python glue per step     {tot:7.3f} ms")

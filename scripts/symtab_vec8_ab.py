"""Dev aid (GPU): the encode-side kernel on fp16 planes with 8-byte (VEC = 4, the default) against 16-byte (VEC = 8, option enc_vec = 8)
loads per lane and plane (the default since round 5), alternating in ONE process on ELIC-4K batches (2 and 4 images: 20 / 40 bitstreams); the bytes must not change.
python scripts/symtab_vec8_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
dev = torch.device("cuda:0")
_lib.set_profiling(0, True)
ELIC = [(g, 136, 120) for g in (16, 16, 32, 64, 192) for _ in range(2)]
for n_img in (2, 4):
    devt = []
    for i in range(10 * n_img):
        M, h, w = ELIC[i % 10]
        y, sg, mu, pi = T.make_latent(i, M=M, h=h, w=w)
        sg, mu, pi = T.to_float16_planes(sg, mu, pi)
        devt.append([torch.from_numpy(a).to(dev) for a in (y, sg, mu, pi)])
    ys, ss, ms, ws = ([t[k] for t in devt] for k in range(4))
    gmc = GaussianMixtureConditional(K=4, mode="polya")
    times, ref = {4: [], 8: []}, None
    for rep in range(8):
        for vec in (4, 8):
            _lib.set_option(0, "enc_vec", vec)
            res = gmc.compress_batch(ys, ss, ms, ws)
            times[vec].append(_lib.kernel_ms(0, 0))
            b = [bytes(r[0][0]) for r in res]
            ref = ref or b
            assert b == ref, (vec, rep)
    n = sum(int(r[0][2].sum()) * t[0].shape[2] * t[0].shape[3] for r, t in zip(res, devt))
    for vec in (4, 8):
        s = float(np.median(times[vec][2:]))
        print(f"{n_img} images ({n} symbols)  VEC={vec}: {s * 1e3:7.1f} us  {n / s / 1e6:6.1f} G symbols/s  {n * 32 / s / 1e6 / 8000:.3f} of 8 TB/s   all: {[round(t * 1e3) for t in times[vec]]}")
_lib.set_option(0, "enc_vec", 0)

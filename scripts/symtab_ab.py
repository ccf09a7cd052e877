"""Dev aid: A/B builds of the library on the symtab kernel (kodak24 batch and one elic4k-like fp16 batch).
    python scripts/symtab_ab.py lib1.so lib2.so ...      (each in its own process via FGMM_LIB)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        env = dict(os.environ, FGMM_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(f"{os.path.basename(lib):28s} {out.stdout.strip()}" + (("\n" + out.stderr[-800:]) if out.returncode else ""), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
dev = torch.device("cuda:0")
_lib.set_profiling(0, True)
res_txt = []
ELIC = [(g, 136, 120) for g in (16, 16, 32, 64, 192) for _ in range(2)]  # the ten bitstreams of one 4K image
for name, n_items, shape, f16, mode in (("kodak f32 polya", 48, (192, 32, 24), False, "polya"), ("kodak f32 as", 48, (192, 32, 24), False, "as"),
                                         ("kodak f32 logistic", 48, (192, 32, 24), False, "logistic"), ("elic f16 polya", 4, (192, 136, 120), True, "polya"),
                                         ("elic 2 images f16", 20, None, True, "polya")):
    devt = []
    for i in range(n_items):
        shape = ELIC[i % 10] if name.startswith("elic 2") else shape
        y, sg, mu, pi = T.make_latent(i, M=shape[0], h=shape[1], w=shape[2])
        if f16:
            sg, mu, pi = T.to_float16_planes(sg, mu, pi)
        devt.append([torch.from_numpy(a).to(dev) for a in (y, sg, mu, pi)])
    ys, ss, ms, ws = ([t[k] for t in devt] for k in range(4))
    gmc = GaussianMixtureConditional(K=4, mode=mode)
    sym = []
    for it in range(10):
        res = gmc.compress_batch(ys, ss, ms, ws)
        sym.append(_lib.kernel_ms(0, 0))
    n = sum(int(r[0][2].sum()) * t[0].shape[2] * t[0].shape[3] for r, t in zip(res, devt))
    s = float(np.median(sym[3:]))
    bps = 32 if f16 else 56
    res_txt.append(f"{name}: {s*1e3:6.1f} us {n/s/1e6:6.1f} Gsym/s {n*bps/s/1e6/8000:.3f}")
print(" | ".join(res_txt))

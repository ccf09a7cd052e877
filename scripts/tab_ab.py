"""Dev aid: A/B builds of the library on the single-pass table kernel alone (fgmm_build_tab_hip on ~3 M kodak-like latents).
    python scripts/tab_ab.py lib1.so lib2.so ...      (each in its own process via FGMM_LIB; FGMM_TAB_CAP_E is honoured)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        env = dict(os.environ, FGMM_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(f"{os.path.basename(lib):24s} {out.stdout.strip()}" + (("\n" + out.stderr[-800:]) if out.returncode else ""), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
from flashgmm_amd import _lib
from tests import synth as T
dev = torch.device("cuda:0")
L, ctx = _lib.lib(), _lib.ctx(0)
S, M_, W_, mb = [], [], [], 0
for i in range(int(os.environ.get("TAB_AB_ITEMS", "24"))):
    y, sg, mu, pi = T.make_latent(i)
    sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
    S.append(np.ascontiguousarray(s)); M_.append(np.ascontiguousarray(m)); W_.append(np.ascontiguousarray(w)); mb = max(mb, abs_max + 1)
s, m, w = (torch.from_numpy(np.concatenate(a)).to(dev) for a in (S, M_, W_))
n = s.size(0)
form = 2 if 2 * mb + 2 <= 254 else 4
cap = n * (2 * (2 * mb + 2) + 4)
hdr = torch.zeros(n * form, dtype=torch.uint8, device=dev)
bo = torch.zeros(n // 16 + 2, dtype=torch.int32, device=dev)
rows = torch.zeros(cap + 128, dtype=torch.uint8, device=dev)
used = torch.zeros(2, dtype=torch.int64, device=dev)
tl = C.c_int32(0)
ts = []
for it in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    rc = L.fgmm_build_tab_hip(ctx, None, s.data_ptr(), m.data_ptr(), w.data_ptr(), n, s.stride(0), s.stride(1), 0, mb, int(os.environ.get('TAB_AB_FLAGS', '2')),
                              hdr.data_ptr(), bo.data_ptr(), rows.data_ptr(), cap, used.data_ptr(), C.byref(tl))
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = float(np.median(ts[2:]))
edges = _lib.ctx_stat(0, 3)
print(f"rc {rc} n {n} max_bs {mb} tl {tl.value}: {t:7.3f} ms  {n/t/1e6:6.2f} G latents/s  edges/latent {edges/n:5.1f}  rows {int(used[0])/n:5.1f} B/latent"
      f"  -> kodak24 step (6.22 M latents) ~ {6.223/ (n/t/1e3):5.2f} ms")

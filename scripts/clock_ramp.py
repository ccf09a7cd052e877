"""Dev aid: does the symtab kernel speed up when launched back to back (clock ramp)?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import _lib, testing as T
dev = torch.device("cuda:0")
L, ctx = _lib.lib(), _lib.ctx(0)
n = 6_291_456
rng = np.random.default_rng(0)
e = np.exp(rng.uniform(-3, 2.5, n)).astype(np.float32)
mu = torch.from_numpy((rng.standard_normal((4, n)) * e).astype(np.float32)).to(dev)
sg = torch.from_numpy(np.clip((rng.uniform(0, 2, (4, n)) + 0.05) * e, 0.11, 256).astype(np.float32)).to(dev)
lg = rng.standard_normal((4, n)); pi = torch.from_numpy((np.exp(lg) / np.exp(lg).sum(0)).astype(np.float32)).to(dev)
v = torch.from_numpy(np.round(rng.standard_normal(n) * 1.5 * e).astype(np.int32)).to(dev)
out = torch.empty(n, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
s = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(L.fgmm_build_symtab_hip(ctx, s, v.data_ptr(), sg.data_ptr(), mu.data_ptr(), pi.data_ptr(), n, 1, n, 0, out.data_ptr()))
for idle_ms in (0, 20):
    ts = []
    for i in range(60):
        if idle_ms: time.sleep(idle_ms / 1e3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"idle {idle_ms} ms between launches: first 3 {ts[:3]}  median of last 30: {np.median(ts[30:]):.1f} us  min {min(ts):.1f} us (n={n}, unclamped IEEE kernel)")

"""Dev aid: VALU issue-rate calibration with the pure-compute saturation selftest kernel."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flashgmm_amd import _lib
L, ctx = _lib.lib(), _lib.ctx(0)
for mode in (0, 1, 2):
    bad = C.c_uint64()
    L.fgmm_selftest_saturation(ctx, mode, C.byref(bad))
    t0 = time.perf_counter()
    for _ in range(3): L.fgmm_selftest_saturation(ctx, mode, C.byref(bad))
    dt = (time.perf_counter() - t0) / 3
    print(f"mode {mode}: {dt*1e3:.2f} ms per scan (bad={bad.value})")

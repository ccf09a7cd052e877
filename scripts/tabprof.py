"""Dev aid (GPU): cycles per phase of tab_kernel (a library built with -DFGMM_TAB_PROF):
    OUT=$PWD/ab/lib_tabprof.so bash flashgmm_amd/csrc/build.sh -DFGMM_TAB_PROF
    FGMM_LIB=ab/lib_tabprof.so [FGMM_TAB_CAP_E=..] [TAB_AB_FLAGS=..] python scripts/tabprof.py"""
import ctypes as C, os, runpy, sys
sys.argv = [sys.argv[0], "--child"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flashgmm_amd import _lib
L = _lib.lib()
L.fgmm_debug_tabprof.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tab_ab.py"), run_name="__main__")
L.fgmm_debug_tabprof(buf, 0)
v = [int(x) for x in buf][:6]
tot = sum(v)
print("block-cycles per phase (share): " + "  ".join(f"{n} {x / tot:.3f}" for n, x in zip(("0+1", "2", "3", "4", "5a", "5b"), v)))

"""ctypes front-end of the CPU oracle (oracle/fgmm_oracle.c) and of the real reference built into
oracle/_ref/.  TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never from flashgmm_amd/.

All arrays are numpy; params are (n, K=4) float32 arrays with arbitrary strides (the reference's
accessor<float,2> contract, rans_interface.cpp:478-480).
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import subprocess
import sysconfig

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfgmm_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")

MODES = {"polya": 0, "as": 1, "logistic": 2}  # rans_interface.cpp:224-232 (code numbering)
MODE_NAMES = {v: k for k, v in MODES.items()}

_lib = None


def build(force: bool = False) -> None:
    """Compile the C restatement (and, when /root/reference is present, the real reference)."""
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
        os.path.join(HERE, "fgmm_oracle.c")
    ):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.isdir("/root/reference/compressai/cpp_exts/rans"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        i64, i32, p = C.c_int64, C.c_int32, C.c_void_p
        L.fgo_exp.restype = C.c_float
        L.fgo_exp.argtypes = [C.c_float]
        L.fgo_gmm_cdf.argtypes = [C.c_int, i64, p, p, p, p, i64, i64, p, p]
        L.fgo_gmm_cdf_x.argtypes = [C.c_int, i64, p, p, p, p, p, i64, i64, p, p]
        L.fgo_symtab.argtypes = [C.c_int, i64, p, p, p, p, i64, i64, p]
        L.fgo_cdftab.argtypes = [C.c_int, i64, p, p, p, i64, i64, i32, p]
        L.fgo_encode_gmm.argtypes = [C.c_int, i64, p, p, p, p, i64, i64, C.POINTER(p), C.POINTER(C.c_size_t), C.POINTER(i64)]
        L.fgo_encode_gmm.restype = C.c_int
        L.fgo_rans_encode_symtab.argtypes = [i64, p, p, C.POINTER(p), C.POINTER(C.c_size_t)]
        L.fgo_rans_encode_symtab.restype = C.c_int
        L.fgo_decode_gmm.argtypes = [C.c_int, p, C.c_size_t, i64, p, p, p, i64, i64, i32, p]
        L.fgo_decode_gmm.restype = C.c_int
        L.fgo_rans_decode_cdftab.argtypes = [p, C.c_size_t, i64, p, i32, p]
        L.fgo_rans_decode_cdftab.restype = C.c_int
        L.fgo_free.argtypes = [p]
        L.fgo_encode_table.argtypes = [i64, p, p, p, i64, p, p, C.POINTER(p), C.POINTER(C.c_size_t)]
        L.fgo_encode_table.restype = C.c_int
        L.fgo_decode_table.argtypes = [p, C.c_size_t, i64, p, p, i64, p, p, p]
        L.fgo_decode_table.restype = C.c_int
        L.fgo_pmf_to_quantized_cdf.argtypes = [p, C.c_int, C.c_int, p]
        L.fgo_pmf_to_quantized_cdf.restype = C.c_int
        L.fgo_head_params.argtypes = [C.c_int, C.c_int, i64, p, p, p, p]
        L.fgo_head_params.restype = None
        _lib = L
    return _lib


def _mode(mode) -> int:
    return MODES[mode] if isinstance(mode, str) else int(mode)


def _params(scales, means, weights):
    """Return three float32 arrays sharing one (stride_n, stride_k) in elements."""
    arrs = [np.asarray(a, dtype=np.float32) for a in (scales, means, weights)]
    n, k = arrs[0].shape
    assert k == 4 and all(a.shape == (n, 4) for a in arrs)
    st = {tuple(s // 4 for s in a.strides) for a in arrs}
    if len(st) != 1 or n == 0:
        arrs = [np.ascontiguousarray(a) for a in arrs]
        st = {(4, 1)}
    (sn, sk), = st
    return arrs, n, sn, sk


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _take_bytes(out_p, out_len) -> bytes:
    data = C.string_at(out_p, out_len.value)
    lib().fgo_free(out_p)
    return data


def exp(x: float) -> float:
    return float(lib().fgo_exp(C.c_float(x)))


def gmm_cdf(mode, v, scales, means, weights):
    (s, m, w), n, sn, sk = _params(scales, means, weights)
    v = np.ascontiguousarray(v, dtype=np.int32)
    c1 = np.empty(n, np.float32)
    c2 = np.empty(n, np.float32)
    lib().fgo_gmm_cdf(_mode(mode), n, _ptr(v), _ptr(s), _ptr(m), _ptr(w), sn, sk, _ptr(c1), _ptr(c2))
    return c1, c2


def gmm_cdf_x(mode, x1, x2, scales, means, weights):
    (s, m, w), n, sn, sk = _params(scales, means, weights)
    x1 = np.ascontiguousarray(x1, dtype=np.float32)
    x2 = np.ascontiguousarray(x2, dtype=np.float32)
    c1 = np.empty(n, np.float32)
    c2 = np.empty(n, np.float32)
    lib().fgo_gmm_cdf_x(_mode(mode), n, _ptr(x1), _ptr(x2), _ptr(s), _ptr(m), _ptr(w), sn, sk, _ptr(c1), _ptr(c2))
    return c1, c2


def symtab(mode, v, scales, means, weights) -> np.ndarray:
    (s, m, w), n, sn, sk = _params(scales, means, weights)
    v = np.ascontiguousarray(v, dtype=np.int32)
    out = np.empty(n, np.uint32)
    lib().fgo_symtab(_mode(mode), n, _ptr(v), _ptr(s), _ptr(m), _ptr(w), sn, sk, _ptr(out))
    return out


def cdftab(mode, scales, means, weights, max_bs: int) -> np.ndarray:
    (s, m, w), n, sn, sk = _params(scales, means, weights)
    out = np.empty((n, 2 * max_bs + 2), np.uint16)
    lib().fgo_cdftab(_mode(mode), n, _ptr(s), _ptr(m), _ptr(w), sn, sk, int(max_bs), _ptr(out))
    return out


def encode_gmm(mode, symbols, scales, means, weights, return_bypass: bool = False):
    (s, m, w), n, sn, sk = _params(scales, means, weights)
    v = np.ascontiguousarray(symbols, dtype=np.int32)
    assert v.shape == (n,)
    out_p, out_len, nb = C.c_void_p(), C.c_size_t(), C.c_int64()
    rc = lib().fgo_encode_gmm(_mode(mode), n, _ptr(v), _ptr(s), _ptr(m), _ptr(w), sn, sk,
                              C.byref(out_p), C.byref(out_len), C.byref(nb))
    if rc:
        raise RuntimeError(f"fgo_encode_gmm rc={rc}")
    data = _take_bytes(out_p, out_len)
    return (data, nb.value) if return_bypass else data


def rans_encode_symtab(packed, symbols) -> bytes:
    packed = np.ascontiguousarray(packed, dtype=np.uint32)
    v = np.ascontiguousarray(symbols, dtype=np.int32)
    out_p, out_len = C.c_void_p(), C.c_size_t()
    rc = lib().fgo_rans_encode_symtab(len(packed), _ptr(packed), _ptr(v), C.byref(out_p), C.byref(out_len))
    if rc:
        raise RuntimeError(f"fgo_rans_encode_symtab rc={rc}")
    return _take_bytes(out_p, out_len)


def decode_gmm(mode, encoded: bytes, scales, means, weights, max_bs_value: int) -> np.ndarray:
    (s, m, w), n, sn, sk = _params(scales, means, weights)
    out = np.empty(n, np.int32)
    buf = np.frombuffer(encoded, dtype=np.uint8)
    rc = lib().fgo_decode_gmm(_mode(mode), _ptr(buf), len(encoded), n, _ptr(s), _ptr(m), _ptr(w), sn, sk,
                              int(max_bs_value), _ptr(out))
    if rc:
        raise RuntimeError(f"fgo_decode_gmm rc={rc}")
    return out


def rans_decode_cdftab(encoded: bytes, tab: np.ndarray, max_bs_value: int) -> np.ndarray:
    tab = np.ascontiguousarray(tab, dtype=np.uint16)
    n = tab.shape[0]
    assert tab.shape[1] == 2 * max_bs_value + 2
    out = np.empty(n, np.int32)
    buf = np.frombuffer(encoded, dtype=np.uint8)
    rc = lib().fgo_rans_decode_cdftab(_ptr(buf), len(encoded), n, _ptr(tab), int(max_bs_value), _ptr(out))
    if rc:
        raise RuntimeError(f"fgo_rans_decode_cdftab rc={rc}")
    return out


def _table_args(indexes, cdfs, cdfs_sizes, offsets):
    idx = np.ascontiguousarray(indexes, dtype=np.int32)
    sizes = np.ascontiguousarray(cdfs_sizes, dtype=np.int32)
    offs = np.ascontiguousarray(offsets, dtype=np.int32)
    width = max(len(c) for c in cdfs)
    mat = np.zeros((len(cdfs), width), np.int32)
    for i, c in enumerate(cdfs):
        mat[i, : len(c)] = c
    return idx, mat, sizes, offs


def encode_table(symbols, indexes, cdfs, cdfs_sizes, offsets) -> bytes:
    """RansEncoder.encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets)  (rans_interface.cpp:587-598)"""
    sym = np.ascontiguousarray(symbols, dtype=np.int32)
    idx, mat, sizes, offs = _table_args(indexes, cdfs, cdfs_sizes, offsets)
    out_p, out_len = C.c_void_p(), C.c_size_t()
    rc = lib().fgo_encode_table(len(sym), _ptr(sym), _ptr(idx), _ptr(mat), mat.shape[1], _ptr(sizes), _ptr(offs),
                                C.byref(out_p), C.byref(out_len))
    if rc:
        raise RuntimeError(f"fgo_encode_table rc={rc}")
    return _take_bytes(out_p, out_len)


def decode_table(encoded: bytes, indexes, cdfs, cdfs_sizes, offsets) -> np.ndarray:
    """RansDecoder.decode_with_indexes(encoded, indexes, cdfs, cdfs_sizes, offsets)  (rans_interface.cpp:619-688)"""
    idx, mat, sizes, offs = _table_args(indexes, cdfs, cdfs_sizes, offsets)
    out = np.empty(len(idx), np.int32)
    buf = np.frombuffer(encoded, dtype=np.uint8)
    rc = lib().fgo_decode_table(_ptr(buf), len(encoded), len(idx), _ptr(idx), _ptr(mat), mat.shape[1], _ptr(sizes),
                                _ptr(offs), _ptr(out))
    if rc:
        raise RuntimeError(f"fgo_decode_table rc={rc}")
    return out


def pmf_to_quantized_cdf(pmf, precision: int = 16):
    """compressai._CXX.pmf_to_quantized_cdf  (ops.cpp:40-109)"""
    pmf = np.ascontiguousarray(pmf, dtype=np.float32)
    out = np.zeros(len(pmf) + 1, np.uint32)
    rc = lib().fgo_pmf_to_quantized_cdf(_ptr(pmf), len(pmf), precision, _ptr(out))
    if rc:
        raise ValueError(f"fgo_pmf_to_quantized_cdf rc={rc}")
    return out.tolist()


# ---------------------------------------------------------------------------------------------------
# The REAL reference (oracle/_ref): only present where `make ref` ran (this container), or where the
# prebuilt files travelled to (the GPU box).  APPROX_MODE is latched once per process by the reference
# (rans_interface.cpp:100), so callers choose the mode through the environment BEFORE first use.
# ---------------------------------------------------------------------------------------------------

def head_params(weight, bias, x) -> np.ndarray:
    """the parameter head's last layer as the product defines its arithmetic (one fmaf chain per output, k ascending, from the bias):
    weight [n_out, c_in], bias [n_out] or None, x [c_in, hw] -> [n_out, hw] float32"""
    w = np.ascontiguousarray(weight, dtype=np.float32)
    xx = np.ascontiguousarray(x, dtype=np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    n_out, c_in = w.shape
    assert xx.shape[0] == c_in and (b is None or b.shape == (n_out,))
    out = np.empty((n_out, xx.shape[1]), np.float32)
    lib().fgo_head_params(n_out, c_in, xx.shape[1], _ptr(w), _ptr(b) if b is not None else None, _ptr(xx), _ptr(out))
    return out


def ref_available(flavour: str = "") -> bool:
    d = os.path.join(REF_DIR, flavour)
    return os.path.exists(os.path.join(d, "ans" + sysconfig.get_config_var("EXT_SUFFIX")))


def ref_ans(flavour: str = ""):
    """Import the unmodified reference extension `compressai.ans` from oracle/_ref (needs torch)."""
    import torch  # noqa: F401  (the extension links libtorch)

    path = os.path.join(REF_DIR, flavour, "ans" + sysconfig.get_config_var("EXT_SUFFIX"))
    spec = importlib.util.spec_from_file_location("ans", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_cxx(flavour: str = ""):
    """Import the unmodified reference extension `compressai._CXX` (pmf_to_quantized_cdf) from oracle/_ref."""
    path = os.path.join(REF_DIR, flavour, "_CXX" + sysconfig.get_config_var("EXT_SUFFIX"))
    if not os.path.exists(path):  # only the x86-64-v3 flavour of this (pybind11-only, scalar) module is built
        path = os.path.join(REF_DIR, "_CXX" + sysconfig.get_config_var("EXT_SUFFIX"))
    spec = importlib.util.spec_from_file_location("_CXX", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_probe(flavour: str = "") -> C.CDLL:
    import torch  # noqa: F401

    L = C.CDLL(os.path.join(REF_DIR, flavour, "libref_probe.so"))
    p = C.c_void_p
    L.ref_probe_gmm_cdf.argtypes = [C.c_long, p, p, p, p, p, p]
    L.ref_probe_gmm_cdf_x.argtypes = [C.c_long, p, p, p, p, p, p, p]
    L.ref_probe_mode.restype = C.c_int
    return L


def ref_gmm_cdf(L, v, scales, means, weights):
    s, m, w = (np.ascontiguousarray(a, dtype=np.float32) for a in (scales, means, weights))
    v = np.ascontiguousarray(v, dtype=np.int32)
    n = len(v)
    c1 = np.empty(n, np.float32)
    c2 = np.empty(n, np.float32)
    L.ref_probe_gmm_cdf(n, _ptr(v), _ptr(m), _ptr(s), _ptr(w), _ptr(c1), _ptr(c2))
    return c1, c2


def ref_gmm_cdf_x(L, x1, x2, scales, means, weights):
    s, m, w = (np.ascontiguousarray(a, dtype=np.float32) for a in (scales, means, weights))
    x1 = np.ascontiguousarray(x1, dtype=np.float32)
    x2 = np.ascontiguousarray(x2, dtype=np.float32)
    n = len(x1)
    c1 = np.empty(n, np.float32)
    c2 = np.empty(n, np.float32)
    L.ref_probe_gmm_cdf_x(n, _ptr(x1), _ptr(x2), _ptr(m), _ptr(s), _ptr(w), _ptr(c1), _ptr(c2))
    return c1, c2

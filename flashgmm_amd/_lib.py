"""ctypes binding of libflashgmm_amd.so (include/flashgmm_amd.h).

The library is built in-tree by ``flashgmm_amd/csrc/build.sh`` (``__graft_entry__.build()``).  There is no
Python or CPU fallback for it: if the shared object is missing or no HIP device is usable, every entry point
raises — loudly — instead of silently doing the work some other way.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FGMM_LIB") or os.path.join(HERE, "libflashgmm_amd.so")  # FGMM_LIB: A/B builds (dev)

FGMM_OK = 0
FGMM_HOST, FGMM_DEVICE = 0, 1
FGMM_K = 4
FGMM_F32, FGMM_F16 = 0, 1
FGMM_HEAD_BF16X6 = 1  # fgmm_head_create_ex flags
FGMM_PARAMS_LOGITS = 1  # fgmm_params.flags: the weights planes hold logits, softmax over K runs in the kernels
MODES = {"polya": 0, "as": 1, "logistic": 2}  # numbering of the reference CODE (rans_interface.cpp:224-232)

STATUS_NAMES = {1: "FGMM_ERR_INVALID", 2: "FGMM_ERR_NO_DEVICE", 3: "FGMM_ERR_HIP", 4: "FGMM_ERR_NOMEM",
                5: "FGMM_ERR_STREAM", 6: "FGMM_ERR_UNSUPPORTED"}


class FgmmError(RuntimeError):
    """Raised for every non-zero fgmm_status (the reference raises RuntimeError through pybind11 as well)."""


class fgmm_params(C.Structure):
    _fields_ = [("scales", C.c_void_p), ("means", C.c_void_p), ("weights", C.c_void_p),
                ("stride_k", C.c_int64), ("stride_c", C.c_int64), ("dtype", C.c_int32), ("flags", C.c_int32)]


class fgmm_ckpt(C.Structure):
    _fields_ = [("x", C.c_uint64), ("pos", C.c_uint64)]


class fgmm_call_marks(C.Structure):
    _fields_ = [("kind", C.c_int32), ("count", C.c_int32), ("t_begin_ms", C.c_double), ("ms", C.c_double * 6),
                ("worker_busy_ms", C.c_double), ("worker_wait_ms", C.c_double), ("head_ms", C.c_double * 3)]


class fgmm_item(C.Structure):
    _fields_ = [("y", C.c_void_p), ("params", fgmm_params), ("M", C.c_int32), ("K", C.c_int32), ("hw", C.c_int64),
                ("yq_out", C.c_void_p), ("zero_bitmap", C.c_void_p), ("abs_max", C.c_int32),
                ("bytes", C.c_void_p), ("bytes_len", C.c_size_t), ("status", C.c_int32), ("ckpt_stride", C.c_int32),
                ("ckpt", C.c_void_p), ("n_ckpt", C.c_int64)]


def _item_dtype():
    """numpy view of ``fgmm_item[]`` (offsets taken from the ctypes declaration): lets a batch be filled column by
    column instead of field by field."""
    import numpy as np
    names, formats, offsets = [], [], []
    kinds = {C.c_void_p: "<u8", C.c_int64: "<i8", C.c_int32: "<i4", C.c_size_t: "<u8"}
    for name, typ in fgmm_item._fields_:
        base = getattr(fgmm_item, name).offset
        if typ is fgmm_params:
            for pn, pt in fgmm_params._fields_:
                names.append(pn); formats.append(kinds[pt]); offsets.append(base + getattr(fgmm_params, pn).offset)
        else:
            names.append(name); formats.append(kinds[typ]); offsets.append(base)
    return np.dtype({"names": names, "formats": formats, "offsets": offsets, "itemsize": C.sizeof(fgmm_item)})


ITEM_DTYPE = _item_dtype()

SINK_ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_size_t)  # fgmm_sink.alloc(user, item, nbytes) -> address


class fgmm_sink(C.Structure):
    """where a batched compress call puts its bitstreams (include/flashgmm_amd.h); alloc is called on the calling thread, inside
    the native call (ctypes takes the GIL back for a Python callback by itself)"""
    _fields_ = [("alloc", SINK_ALLOC), ("user", C.c_void_p)]


# every symbol include/flashgmm_amd.h declares: (restype, argtypes)
_p, _i, _i32, _i64, _sz = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_size_t
_pp, _psz = C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)
SIGNATURES = {
    "fgmm_abi_version": (_i, []),
    "fgmm_last_error": (C.c_char_p, []),
    "fgmm_host_cpu_budget": (_i, [C.POINTER(C.c_double), C.POINTER(_i), C.POINTER(C.c_double)]),
    "fgmm_host_thread_budget": (_i, [_i]),
    "fgmm_ctx_create": (_i, [_i, _i, _pp]),
    "fgmm_ctx_destroy": (None, [_p]),
    "fgmm_ctx_device": (_i, [_p]),
    "fgmm_ctx_threads": (_i, [_p]),
    "fgmm_ctx_set_threads": (_i, [_p, _i]),
    "fgmm_ctx_worker_cpus": (_i, [_p, C.c_char_p, _sz]),
    "fgmm_free": (None, [_p]),
    "fgmm_ctx_take_buffers": (_i, [_p, _p, _p, _p, _i]),
    "fgmm_ctx_set_profiling": (_i, [_p, _i]),
    "fgmm_ctx_kernel_ms": (_i, [_p, _i, C.POINTER(C.c_float)]),
    "fgmm_ctx_stat": (_i, [_p, _i, C.POINTER(C.c_uint64)]),
    "fgmm_ctx_call_log": (_i, [_p, C.POINTER(fgmm_call_marks), _i, C.POINTER(_i)]),
    "fgmm_encode_with_indexes_gmm": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i, _i, _i, _i32, _pp, _psz]),
    "fgmm_decode_with_indexes_gmm": (_i, [_p, _p, _sz, _p, _p, _p, _i64, _i64, _i64, _i, _i, _i, _i32, _p]),
    "fgmm_gmc_compress": (_i, [_p, _p, _p, C.POINTER(fgmm_params), _i, _i, _i64, _i, _i, _p, C.POINTER(_i32), _p, _pp, _psz]),
    "fgmm_gmc_decompress": (_i, [_p, _p, _p, _sz, _i32, _p, C.POINTER(fgmm_params), _i, _i, _i64, _i, _i, _p]),
    "fgmm_gmc_compress_batch": (_i, [_p, _p, C.POINTER(fgmm_item), _i, _i, _i]),
    "fgmm_gmc_decompress_batch": (_i, [_p, _p, C.POINTER(fgmm_item), _i, _i, _i]),
    "fgmm_gmc_compress_batch_to": (_i, [_p, _p, C.POINTER(fgmm_item), _i, _i, _i, C.POINTER(fgmm_sink)]),
    "fgmm_head_create": (_i, [_p, _p, _p, _p, _i, _i, _i, _pp]),
    "fgmm_head_create_ex": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _pp]),
    "fgmm_head_destroy": (None, [_p]),
    "fgmm_head_params_batch": (_i, [_p, _p, _p, _p, _p, _p, _i]),
    "fgmm_gmc_compress_head_batch": (_i, [_p, _p, C.POINTER(fgmm_item), _p, _i, _p, _i, _i]),
    "fgmm_gmc_compress_head_batch_to": (_i, [_p, _p, C.POINTER(fgmm_item), _p, _i, _p, _i, _i, C.POINTER(fgmm_sink)]),
    "fgmm_gmm_cdf_hip": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i, _p, _p]),
    "fgmm_softmax4_hip": (_i, [_p, _p, _p, _p, _i64]),
    "fgmm_build_symtab_hip": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "fgmm_build_cdftab_hip": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i, _i32, _i, _p, _p, C.c_uint64, _p]),
    "fgmm_selftest_saturation": (_i, [_p, _i, C.POINTER(C.c_uint64)]),
    "fgmm_selftest_fastmath": (_i, [_p, _i, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "fgmm_rans_encode_symtab": (_i, [_p, _p, _i64, _pp, _psz]),
    "fgmm_ckpt_count": (_i64, [_i64, _i64]),
    "fgmm_rans_encode_symtab_ckpt": (_i, [_p, _p, _i64, _i64, _pp, _psz, _p]),
    "fgmm_rans_decode_tab_ckpt": (_i, [_p, _sz, _p, _i, _p, _i32, _p, C.c_uint64, _i64, _i32, _i, _p, _i64, _i64, _p, C.POINTER(_i32)]),
    "fgmm_rans_encode_symtab2": (_i, [_p, _p, _i64, _p, _p, _i64, _pp, _psz, _pp, _psz]),
    "fgmm_rans_encode_symtab_n": (_i, [_i, _p, _p, _p, _p, _p]),
    "fgmm_rans_encode_symtab_segs": (_i, [_p, _i, _i64, _p, _i64, _i64, _pp, _psz, _p]),
    "fgmm_rans_decode_cdftab": (_i, [_p, _sz, _p, _p, C.c_uint64, _i64, _i32, _i, _p]),
    "fgmm_rans_decode_tab": (_i, [_p, _sz, _p, _i, _p, _i32, _p, C.c_uint64, _i64, _i32, _i, _p]),
    "fgmm_build_tab_hip": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i, _i32, _i, _p, _p, _p, C.c_uint64, _p, C.POINTER(_i32)]),
    "fgmm_ctx_set_option": (_i, [_p, C.c_char_p, _i64]),
    "fgmm_ctx_get_option": (_i, [_p, C.c_char_p, C.POINTER(_i64)]),
    "fgmm_ctx_trim": (_i, [_p]),
    # table path (z hyper-latent coder), host only
    "fgmm_encode_with_indexes": (_i, [_p, _p, _i64, _p, _i64, _i32, _p, _p, _pp, _psz]),
    "fgmm_decode_with_indexes": (_i, [_p, _sz, _p, _i64, _p, _i64, _i32, _p, _p, _p]),
    "fgmm_symbuf_create": (_i, [_pp]),
    "fgmm_symbuf_destroy": (None, [_p]),
    "fgmm_symbuf_size": (_i64, [_p]),
    "fgmm_symbuf_append_table": (_i, [_p, _p, _p, _i64, _p, _i64, _i32, _p, _p]),
    "fgmm_symbuf_append_symtab": (_i, [_p, _p, _p, _i64]),
    "fgmm_symbuf_flush": (_i, [_p, _pp, _psz]),
    "fgmm_symbuf_append_gmm": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i, _i, _i]),
    "fgmm_decstream_create": (_i, [_p, _sz, _pp]),
    "fgmm_decstream_destroy": (None, [_p]),
    "fgmm_decstream_decode": (_i, [_p, _p, _i64, _p, _i64, _i32, _p, _p, _p]),
    "fgmm_pmf_to_quantized_cdf": (_i, [_p, _i, _i, _p]),
    "fgmm_ckbd_unembed": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i, _i]),
    "fgmm_ckbd_embed": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i, _i]),
}

_lib = None
_lock = threading.Lock()
_ctxs = {}
_native = None  # flashgmm_amd._native (csrc/fgmm_pybind.cpp), False when it cannot be used


def native():
    """The compiled boundary (``flashgmm_amd._native``, pybind11 over the same C ABI: items built in C++, the GIL released across
    the call) or None - then the ctypes path of this module does the same work.  Not used with FGMM_LIB (an A/B build of the
    library: the module is linked against the in-tree one) or FGMM_NATIVE=0."""
    global _native
    if _native is None:
        _native = False
        if not os.environ.get("FGMM_LIB") and os.environ.get("FGMM_NATIVE", "1") != "0":
            try:
                lib()  # (the library first, by its path: the module's own dependency then resolves to the same mapping)
                from . import _native as mod

                if mod.abi_version == lib().fgmm_abi_version():
                    _native = mod
            except ImportError:
                pass
    return _native or None


def ctx_addr(device: int = -1) -> int:
    return ctx(device).value


def lib() -> C.CDLL:
    """Load the shared object (no GPU needed for this step) and bind every declared symbol."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise FgmmError(
                    f"{LIB_PATH} is missing: build it with flashgmm_amd/csrc/build.sh (or __graft_entry__.build()). "
                    "flashgmm_amd has no CPU / pure-Python fallback for the GMM entropy-coding path."
                )
            L = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(L, name)  # AttributeError here = header and library out of sync
                fn.restype = res
                fn.argtypes = args
            _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != FGMM_OK:
        msg = lib().fgmm_last_error().decode(errors="replace")
        raise FgmmError(f"{what or 'libflashgmm_amd'}: {STATUS_NAMES.get(rc, rc)}: {msg}")


def ctx(device: int = -1, n_threads: int = 0) -> C.c_void_p:
    """Process-wide context of a GPU (created on first use; needs a real HIP device)."""
    L = lib()
    key = int(device)
    with _lock:
        h = _ctxs.get(key)
    if h is None:
        out = C.c_void_p()
        if n_threads <= 0:  # this process's share of the host's CPU budget (affinity mask, cgroup quota, ranks on the node)
            n_threads = L.fgmm_host_thread_budget(ranks_on_node())
        check(L.fgmm_ctx_create(key, int(n_threads), C.byref(out)), "fgmm_ctx_create")
        with _lock:
            if key in _ctxs:  # lost a race: keep the first
                L.fgmm_ctx_destroy(out)
            else:
                _ctxs[key] = out
            h = _ctxs[key]
    return h


def take_bytes(ptr: C.c_void_p, length: int) -> bytes:
    data = C.string_at(ptr, length)
    lib().fgmm_free(ptr)
    return data


_py = C.pythonapi
_py.PyBytes_FromStringAndSize.restype = C.py_object
_py.PyBytes_FromStringAndSize.argtypes = [C.c_void_p, C.c_ssize_t]
_py.PyBytes_AsString.restype = C.c_void_p
_py.PyBytes_AsString.argtypes = [C.py_object]


def take_bytes_many(device: int, ptrs, lens, cls=None) -> list:
    """library-allocated buffers -> ``bytes`` objects (or instances of the bytes subclass ``cls``), the buffers released.
    The copies are done by the context's host workers, straight into the objects' own storage: ``bytes`` objects allocated
    uninitialised (``PyBytes_FromStringAndSize(NULL, n)``: the C API's way to fill a bytes object before anyone else sees it),
    subclass instances as ``bytes.__new__(cls, n)`` (zero pages until written) - one copy, not one per layer of glue."""
    n = len(ptrs)
    if n == 0:
        return []
    if cls is None:
        if sum(lens) < (1 << 18):
            return [take_bytes(p, l) for p, l in zip(ptrs, lens)]
        objs = [_py.PyBytes_FromStringAndSize(None, int(l)) for l in lens]
    else:
        objs = [bytes.__new__(cls, int(l)) for l in lens]
    dst = (C.c_void_p * n)(*[_py.PyBytes_AsString(o) if l else None for o, l in zip(objs, lens)])
    src = (C.c_void_p * n)(*ptrs)
    ln = (C.c_size_t * n)(*lens)
    check(lib().fgmm_ctx_take_buffers(ctx(device), dst, src, ln, n), "fgmm_ctx_take_buffers")
    return objs


def take_buffers_into(device: int, dst_addrs, src_ptrs, lens) -> None:
    """library-allocated buffers copied to the given addresses and released (one native call)"""
    n = len(src_ptrs)
    if n:
        check(lib().fgmm_ctx_take_buffers(ctx(device), (C.c_void_p * n)(*dst_addrs), (C.c_void_p * n)(*src_ptrs), (C.c_size_t * n)(*lens), n),
              "fgmm_ctx_take_buffers")


def mode_id(mode) -> int:
    if isinstance(mode, str):
        return MODES[mode.lower()]
    m = int(mode)
    if m not in (0, 1, 2):
        raise ValueError(f"mode {mode!r}")
    return m


def default_mode() -> int:
    """APPROX_MODE env var, parsed as the reference does (rans_interface.cpp:99-115): 0/1/2, anything else -> 0."""
    try:
        m = int(os.environ.get("APPROX_MODE", "0"))
    except ValueError:
        return 0
    return m if 0 <= m <= 2 else 0


def set_option(device: int, name: str, value: int) -> None:
    """a tuning knob of the device's context (include/flashgmm_amd.h: fgmm_ctx_set_option)"""
    check(lib().fgmm_ctx_set_option(ctx(device), name.encode(), int(value)), f"fgmm_ctx_set_option({name})")


def get_option(device: int, name: str) -> int:
    out = C.c_int64()
    check(lib().fgmm_ctx_get_option(ctx(device), name.encode(), C.byref(out)), f"fgmm_ctx_get_option({name})")
    return int(out.value)


def trim(device: int) -> None:
    check(lib().fgmm_ctx_trim(ctx(device)), "fgmm_ctx_trim")


def ranks_on_node() -> int:
    """processes (one per GPU) that share this host's CPU budget: LOCAL_WORLD_SIZE as torch.distributed.run exports it (bench.py's
    own launcher exports it too), FGMM_RANKS_ON_NODE to say it by hand; 1 otherwise"""
    for name in ("FGMM_RANKS_ON_NODE", "LOCAL_WORLD_SIZE"):
        try:
            v = int(os.environ.get(name, ""))
            if v >= 1:
                return v
        except ValueError:
            pass
    return 1


def host_cpu_budget() -> dict:
    """-> {"cpus", "affinity", "quota"} (fgmm_host_cpu_budget; quota None when the cgroup sets none)"""
    cpus, aff, quota = C.c_double(), C.c_int(), C.c_double()
    check(lib().fgmm_host_cpu_budget(C.byref(cpus), C.byref(aff), C.byref(quota)), "fgmm_host_cpu_budget")
    return {"cpus": cpus.value, "affinity": aff.value, "quota": quota.value if quota.value > 0 else None}


def set_threads(device: int, n_threads: int) -> None:
    """Resize the pool of host rANS workers of the device's context in place (0: the default for this process's share of the
    host).  Options, profiling state and buffers of the context are kept; the handle from ``ctx()`` stays valid."""
    if n_threads <= 0:
        n_threads = lib().fgmm_host_thread_budget(ranks_on_node())
    check(lib().fgmm_ctx_set_threads(ctx(device), int(n_threads)), "fgmm_ctx_set_threads")


def set_profiling(device: int, enable: bool) -> None:
    check(lib().fgmm_ctx_set_profiling(ctx(device), int(enable)), "fgmm_ctx_set_profiling")


def kernel_ms(device: int, which: int) -> float:
    """Duration (ms) of the most recent launch of kernel `which` (0 symtab, 1 decode-side table kernels of the last call,
    first to last, 2 quant_stats), from HIP events recorded on the stream the kernel ran on."""
    out = C.c_float()
    check(lib().fgmm_ctx_kernel_ms(ctx(device), which, C.byref(out)), "fgmm_ctx_kernel_ms")
    return float(out.value)


def ctx_stat(device: int, which: int) -> int:
    """Counters of the most recent batched call (0 encode tables D2H bytes, 1 decode tables D2H bytes, 2 decode latents,
    3 edges the decode-side kernels evaluated)."""
    out = C.c_uint64()
    check(lib().fgmm_ctx_stat(ctx(device), which, C.byref(out)), "fgmm_ctx_stat")
    return int(out.value)


CALL_KINDS = {0: "encode", 1: "decode", 2: "decode_gpu"}


def worker_cpus(device: int) -> str:
    """cpulist of the CPUs the context's host workers may run on ("" = the creating thread's mask): fgmm_ctx_worker_cpus"""
    buf = C.create_string_buffer(4096)
    check(lib().fgmm_ctx_worker_cpus(ctx(device), buf, len(buf)), "fgmm_ctx_worker_cpus")
    return buf.value.decode()


def call_log(device: int, last: int = 64) -> list:
    """phase marks of the context's most recent batched calls, oldest first (include/flashgmm_amd.h: fgmm_ctx_call_log)"""
    buf = (fgmm_call_marks * 64)()
    n = C.c_int()
    check(lib().fgmm_ctx_call_log(ctx(device), buf, min(int(last), 64), C.byref(n)), "fgmm_ctx_call_log")
    return [{"kind": CALL_KINDS.get(m.kind, m.kind), "count": m.count, "t_begin_ms": m.t_begin_ms, "ms": list(m.ms),
             "worker_busy_ms": m.worker_busy_ms, "worker_wait_ms": m.worker_wait_ms, "head_ms": list(m.head_ms)} for m in buf[:n.value]]

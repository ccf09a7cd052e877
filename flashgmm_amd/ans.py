"""``flashgmm_amd.ans`` — host-side mirror of the reference's pybind11 module ``compressai.ans`` for the GMM
path (compressai/cpp_exts/rans/rans_interface.cpp:961-1036), over the C ABI of libflashgmm_amd.so.

Same class names, method names, keyword names and argument meaning:

    RansEncoder().encode_with_indexes_gmm(symbols, scales, means, weights, max_value) -> bytes
    BufferedRansEncoder().encode_with_indexes_gmm(...same...) -> None ;  .flush() -> bytes
    RansDecoder().decode_with_indexes_gmm(encoded, scales, means, weights, max_bs_value) -> IntTensor[n]

plus the table path used for the `z` hyper-latent (host, integer only), same signatures as the reference:

    RansEncoder().encode_with_indexes(symbols, indexes, cdfs, cdfs_sizes, offsets) -> bytes
    BufferedRansEncoder().encode_with_indexes(...) ; RansDecoder().decode_with_indexes(encoded, indexes, cdfs, ...)
    RansDecoder().set_stream(encoded) ; .decode_stream(indexes, cdfs, cdfs_sizes, offsets) -> list[int]

Differences, all deliberate:
  * tensors may live on the GPU (then nothing crosses PCIe except the 4 B/symbol table); CPU tensors — what the
    reference is handed (entropy_models.py:859-865) — are staged to the GPU by the library;
  * wrong dtype / rank / K raise RuntimeError with a message (the reference has its checks commented out and
    mis-reads silently, rans_interface.cpp:465-474);
  * the Phi approximation is an explicit ``mode`` (default: the APPROX_MODE environment variable, read as the
    reference reads it).
The float work runs in HIP kernels only; there is no CPU implementation behind these calls.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib

__all__ = ["RansEncoder", "BufferedRansEncoder", "RansDecoder"]


def _check_rows(name: str, t: torch.Tensor, n: Optional[int]) -> None:
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name} must be a torch.Tensor")
    if t.dtype != torch.float32 or t.dim() != 2 or t.size(1) != _lib.FGMM_K:
        raise RuntimeError(f"{name} must be a float32 tensor of shape (n, {_lib.FGMM_K}); got {t.dtype} {tuple(t.shape)}")
    if n is not None and t.size(0) != n:
        raise RuntimeError(f"{name} has {t.size(0)} rows, expected {n}")


def _common_layout(scales, means, weights):
    """The C ABI takes one (stride_n, stride_k) for the three arrays; make them agree (copy only if they do not)."""
    st = {tuple(t.stride()) for t in (scales, means, weights)}
    dev = {t.device for t in (scales, means, weights)}
    if len(dev) != 1:
        raise RuntimeError("scales, means and weights must be on the same device")
    if len(st) != 1 or scales.size(0) <= 1:
        scales, means, weights = (t.contiguous() for t in (scales, means, weights))
    return scales, means, weights


def _device_index(t: torch.Tensor) -> int:
    return t.device.index if t.is_cuda and t.device.index is not None else -1


def _prep_gmm(symbols, scales, means, weights):
    if not isinstance(symbols, torch.Tensor) or symbols.dtype != torch.int32 or symbols.dim() != 1:
        raise RuntimeError("symbols must be a 1-D int32 tensor")
    n = symbols.numel()
    for name, t in (("scales", scales), ("means", means), ("weights", weights)):
        _check_rows(name, t, n)
    scales, means, weights = _common_layout(scales, means, weights)
    if symbols.device != scales.device:
        symbols = symbols.to(scales.device)
    symbols = symbols.contiguous()
    on_gpu = scales.is_cuda
    if on_gpu:
        torch.cuda.current_stream(scales.device).synchronize()  # the raw boundary runs on the default stream
    return symbols, scales, means, weights, n, on_gpu


def _encode(symbols, scales, means, weights, max_value, mode) -> bytes:
    symbols, scales, means, weights, n, on_gpu = _prep_gmm(symbols, scales, means, weights)
    L = _lib.lib()
    ctx = _lib.ctx(_device_index(scales) if on_gpu else -1)
    out, out_len = C.c_void_p(), C.c_size_t()
    rc = L.fgmm_encode_with_indexes_gmm(
        ctx, symbols.data_ptr(), scales.data_ptr(), means.data_ptr(), weights.data_ptr(), n,
        scales.stride(0) if n else 4, scales.stride(1) if n else 1, _lib.FGMM_K,
        _lib.default_mode() if mode is None else _lib.mode_id(mode),
        _lib.FGMM_DEVICE if on_gpu else _lib.FGMM_HOST, int(max_value), C.byref(out), C.byref(out_len))
    _lib.check(rc, "encode_with_indexes_gmm")
    return _lib.take_bytes(out, out_len.value)


def _i32(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.int32).reshape(-1)


class _Tables:
    """cdfs (list of lists, ragged, or an int matrix), cdfs_sizes, offsets -> what the C ABI takes.  Built once per
    call like the reference's pybind conversion; pass a prepared ``_Tables`` again to skip the conversion."""

    def __init__(self, cdfs, cdfs_sizes, offsets):
        self.sizes = _i32(cdfs_sizes)
        self.offsets = _i32(offsets)
        if isinstance(cdfs, torch.Tensor):
            cdfs = cdfs.detach().cpu().numpy()
        if isinstance(cdfs, np.ndarray) and cdfs.ndim == 2:
            self.mat = np.ascontiguousarray(cdfs, dtype=np.int32)
        else:
            width = max((len(c) for c in cdfs), default=2)
            self.mat = np.zeros((len(cdfs), max(width, 2)), np.int32)
            for i, c in enumerate(cdfs):
                self.mat[i, : len(c)] = c
        if not (len(self.sizes) == len(self.offsets) == self.mat.shape[0]):
            raise RuntimeError("cdfs, cdfs_sizes and offsets must describe the same number of tables")

    def args(self):
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        return p(self.mat), self.mat.shape[1], self.mat.shape[0], p(self.sizes), p(self.offsets)


def _tables(cdfs, cdfs_sizes, offsets) -> _Tables:
    return cdfs if isinstance(cdfs, _Tables) else _Tables(cdfs, cdfs_sizes, offsets)


def _check(rc: int, what: str) -> None:
    _lib.check(rc, what)


class RansEncoder:
    """compressai.ans.RansEncoder (rans_interface.hpp:88-112)."""

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes=None, offsets=None) -> bytes:
        """table rANS, rans_interface.cpp:587-598"""
        sym, idx, t = _i32(symbols), _i32(indexes), _tables(cdfs, cdfs_sizes, offsets)
        if sym.size != idx.size:
            raise RuntimeError("symbols and indexes must have the same length")
        out, out_len = C.c_void_p(), C.c_size_t()
        _check(_lib.lib().fgmm_encode_with_indexes(sym.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p),
                                                   sym.size, *t.args(), C.byref(out), C.byref(out_len)),
               "encode_with_indexes")
        return _lib.take_bytes(out, out_len.value)

    def encode_with_indexes_gmm(self, symbols, scales, means, weights, max_value, *, mode=None) -> bytes:
        return _encode(symbols, scales, means, weights, max_value, mode)


class BufferedRansEncoder:
    """compressai.ans.BufferedRansEncoder (rans_interface.hpp:57-86): calls accumulate symbols — table calls and
    GMM calls may be mixed — and ``flush`` emits ONE stream covering them in call order (rans_interface.cpp:557-585)."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(_lib.lib().fgmm_symbuf_create(C.byref(self._h)), "BufferedRansEncoder")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().fgmm_symbuf_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes=None, offsets=None) -> None:
        sym, idx, t = _i32(symbols), _i32(indexes), _tables(cdfs, cdfs_sizes, offsets)
        if sym.size != idx.size:
            raise RuntimeError("symbols and indexes must have the same length")
        _check(_lib.lib().fgmm_symbuf_append_table(self._h, sym.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p),
                                                   sym.size, *t.args()), "encode_with_indexes")

    def encode_with_indexes_gmm(self, symbols, scales, means, weights, max_value, *, mode=None) -> None:
        symbols, scales, means, weights, n, on_gpu = _prep_gmm(symbols, scales, means, weights)
        ctx = _lib.ctx(_device_index(scales) if on_gpu else -1)
        _check(_lib.lib().fgmm_symbuf_append_gmm(
            ctx, self._h, symbols.data_ptr(), scales.data_ptr(), means.data_ptr(), weights.data_ptr(), n,
            scales.stride(0) if n else 4, scales.stride(1) if n else 1, _lib.FGMM_K,
            _lib.default_mode() if mode is None else _lib.mode_id(mode),
            _lib.FGMM_DEVICE if on_gpu else _lib.FGMM_HOST), "encode_with_indexes_gmm")

    def flush(self) -> bytes:
        out, out_len = C.c_void_p(), C.c_size_t()
        _check(_lib.lib().fgmm_symbuf_flush(self._h, C.byref(out), C.byref(out_len)), "flush")
        return _lib.take_bytes(out, out_len.value)


class RansDecoder:
    """compressai.ans.RansDecoder (rans_interface.hpp:114-154)."""

    def __init__(self):
        self._stream = None

    def __del__(self):
        try:
            if getattr(self, "_stream", None):
                _lib.lib().fgmm_decstream_destroy(self._stream)
                self._stream = None
        except Exception:
            pass

    def decode_with_indexes(self, encoded, indexes, cdfs, cdfs_sizes=None, offsets=None) -> List[int]:
        """table rANS, rans_interface.cpp:619-688; returns a list of ints like the reference"""
        encoded = bytes(encoded)
        idx, t = _i32(indexes), _tables(cdfs, cdfs_sizes, offsets)
        out = np.empty(idx.size, np.int32)
        _check(_lib.lib().fgmm_decode_with_indexes(encoded, len(encoded), idx.ctypes.data_as(C.c_void_p), idx.size,
                                                   *t.args(), out.ctypes.data_as(C.c_void_p)), "decode_with_indexes")
        return out.tolist()

    def set_stream(self, encoded) -> None:
        """rans_interface.cpp:886-892"""
        encoded = bytes(encoded)
        if self._stream:
            _lib.lib().fgmm_decstream_destroy(self._stream)
            self._stream = None
        h = C.c_void_p()
        _check(_lib.lib().fgmm_decstream_create(encoded, len(encoded), C.byref(h)), "set_stream")
        self._stream = h

    def decode_stream(self, indexes, cdfs, cdfs_sizes=None, offsets=None) -> List[int]:
        """rans_interface.cpp:894-956"""
        if not self._stream:
            raise RuntimeError("decode_stream called before set_stream")
        idx, t = _i32(indexes), _tables(cdfs, cdfs_sizes, offsets)
        out = np.empty(idx.size, np.int32)
        _check(_lib.lib().fgmm_decstream_decode(self._stream, idx.ctypes.data_as(C.c_void_p), idx.size, *t.args(),
                                                out.ctypes.data_as(C.c_void_p)), "decode_stream")
        return out.tolist()

    def decode_with_indexes_gmm(self, encoded, scales, means, weights, max_bs_value, *, mode=None) -> torch.Tensor:
        if not isinstance(encoded, (bytes, bytearray, memoryview)):
            raise RuntimeError("encoded must be bytes")
        encoded = bytes(encoded)
        for name, t in (("scales", scales), ("means", means), ("weights", weights)):
            _check_rows(name, t, scales.size(0) if isinstance(scales, torch.Tensor) else None)
        scales, means, weights = _common_layout(scales, means, weights)
        n = scales.size(0)
        on_gpu = scales.is_cuda
        L = _lib.lib()
        ctx = _lib.ctx(_device_index(scales) if on_gpu else -1)
        if on_gpu:
            torch.cuda.current_stream(scales.device).synchronize()
        out = torch.empty(n, dtype=torch.int32)  # the reference returns a CPU int32 tensor (rans_interface.cpp:782)
        rc = L.fgmm_decode_with_indexes_gmm(
            ctx, encoded, len(encoded), scales.data_ptr(), means.data_ptr(), weights.data_ptr(), n,
            scales.stride(0) if n else 4, scales.stride(1) if n else 1, _lib.FGMM_K,
            _lib.default_mode() if mode is None else _lib.mode_id(mode),
            _lib.FGMM_DEVICE if on_gpu else _lib.FGMM_HOST, int(max_bs_value), out.data_ptr())
        _lib.check(rc, "decode_with_indexes_gmm")
        return out

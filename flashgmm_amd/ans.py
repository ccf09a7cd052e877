"""``flashgmm_amd.ans`` — host-side mirror of the reference's pybind11 module ``compressai.ans`` for the GMM
path (compressai/cpp_exts/rans/rans_interface.cpp:961-1036), over the C ABI of libflashgmm_amd.so.

Same class names, method names, keyword names and argument meaning:

    RansEncoder().encode_with_indexes_gmm(symbols, scales, means, weights, max_value) -> bytes
    BufferedRansEncoder().encode_with_indexes_gmm(...same...) -> None ;  .flush() -> bytes
    RansDecoder().decode_with_indexes_gmm(encoded, scales, means, weights, max_bs_value) -> IntTensor[n]

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
from typing import List, Optional

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


def _encode(symbols, scales, means, weights, max_value, mode) -> bytes:
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
    L = _lib.lib()
    ctx = _lib.ctx(_device_index(scales) if on_gpu else -1)
    if on_gpu:
        torch.cuda.current_stream(scales.device).synchronize()  # the raw boundary runs on the default stream
    out, out_len = C.c_void_p(), C.c_size_t()
    rc = L.fgmm_encode_with_indexes_gmm(
        ctx, symbols.data_ptr(), scales.data_ptr(), means.data_ptr(), weights.data_ptr(), n,
        scales.stride(0) if n else 4, scales.stride(1) if n else 1, _lib.FGMM_K,
        _lib.default_mode() if mode is None else _lib.mode_id(mode),
        _lib.FGMM_DEVICE if on_gpu else _lib.FGMM_HOST, int(max_value), C.byref(out), C.byref(out_len))
    _lib.check(rc, "encode_with_indexes_gmm")
    return _lib.take_bytes(out, out_len.value)


class RansEncoder:
    """compressai.ans.RansEncoder (rans_interface.hpp:88-112), GMM method."""

    def encode_with_indexes_gmm(self, symbols, scales, means, weights, max_value, *, mode=None) -> bytes:
        return _encode(symbols, scales, means, weights, max_value, mode)


class BufferedRansEncoder:
    """compressai.ans.BufferedRansEncoder (rans_interface.hpp:57-86): calls accumulate, ``flush`` emits ONE stream
    covering every buffered symbol in call order (rans_interface.cpp:557-585)."""

    def __init__(self):
        self._parts: List[tuple] = []

    def encode_with_indexes_gmm(self, symbols, scales, means, weights, max_value, *, mode=None) -> None:
        self._parts.append((symbols, scales, means, weights, max_value, mode))

    def flush(self) -> bytes:
        parts, self._parts = self._parts, []
        if not parts:
            return bytes.fromhex("0000008000000000")  # Rans64EncInit state, flushed
        if len(parts) == 1:
            return _encode(*parts[0])
        modes = {p[5] for p in parts}
        if len(modes) != 1:
            raise RuntimeError("one stream cannot mix Phi approximations")
        sym = torch.cat([p[0].reshape(-1) for p in parts])
        s, m, w = (torch.cat([p[i].contiguous() for p in parts]) for i in (1, 2, 3))
        return _encode(sym, s, m, w, parts[-1][4], parts[0][5])


class RansDecoder:
    """compressai.ans.RansDecoder (rans_interface.hpp:114-154), GMM method."""

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

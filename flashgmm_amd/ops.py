"""Operators either side of the path.

``ckbd_unembed`` / ``ckbd_embed`` — ``CheckerboardLatentCodec.unembed / embed`` (compressai/latent_codecs/checkerboard.py:
333-377) as one HIP kernel each (the reference issues four strided slice assignments into a zero-filled tensor).

``pmf_to_quantized_cdf`` — mirror of ``compressai._CXX.pmf_to_quantized_cdf`` (compressai/cpp_exts/ops/ops.cpp:40-109),
the table builder behind ``EntropyBottleneck.update()`` / ``GaussianConditional.update()`` (the `z` hyper-latent path).
Runs once per model; host code in libflashgmm_amd.so."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib

__all__ = ["pmf_to_quantized_cdf", "ckbd_unembed", "ckbd_embed"]


def _ckbd(t, anchor_parity: str, embed: bool):
    import torch

    if anchor_parity not in ("even", "odd"):
        raise ValueError(f"anchor_parity {anchor_parity!r}")
    if not t.is_cuda:
        raise RuntimeError("flashgmm_amd operators run on the GPU only: the tensor must be on a HIP device")
    if t.element_size() not in (2, 4):
        raise RuntimeError(f"checkerboard split/merge handles 2- and 4-byte element types, got {t.dtype}")
    t = t.contiguous()
    if embed:
        if t.dim() != 5 or t.shape[0] != 2:
            raise RuntimeError(f"embed expects [2, n, c, h, w/2], got {tuple(t.shape)}")
        _, n, c, h, w2 = t.shape
        out = torch.empty((n, c, h, 2 * w2), dtype=t.dtype, device=t.device)
        fn, w = _lib.lib().fgmm_ckbd_embed, 2 * w2
    else:
        if t.dim() != 4:
            raise RuntimeError(f"unembed expects [n, c, h, w], got {tuple(t.shape)}")
        n, c, h, w = t.shape
        if w % 2:
            raise RuntimeError("unembed needs an even width (the reference's slicing does too)")
        out = torch.empty((2, n, c, h, w // 2), dtype=t.dtype, device=t.device)
        fn = _lib.lib().fgmm_ckbd_unembed
    dev = t.device.index if t.device.index is not None else -1
    rc = fn(_lib.ctx(dev), torch.cuda.current_stream(t.device).cuda_stream, t.data_ptr(), out.data_ptr(), n * c, h, w,
            t.element_size(), int(anchor_parity == "odd"))
    _lib.check(rc, "ckbd_embed" if embed else "ckbd_unembed")
    return out


def ckbd_unembed(y, anchor_parity: str = "even"):
    """``[n, c, h, w] -> [2, n, c, h, w/2]``: half 0 = anchors, half 1 = non-anchors (checkerboard.py:333-354)."""
    return _ckbd(y, anchor_parity, False)


def ckbd_embed(y_, anchor_parity: str = "even"):
    """``[2, n, c, h, w/2] -> [n, c, h, w]`` (checkerboard.py:356-377)."""
    return _ckbd(y_, anchor_parity, True)


def pmf_to_quantized_cdf(pmf: Sequence[float], precision: int = 16) -> List[int]:
    """Same contract as the reference: list of len(pmf)+1 ints, cdf[0] == 0, cdf[-1] == 1 << precision, strictly
    increasing; ``ValueError`` for negative / non-finite / all-zero input (pybind11 maps std::domain_error to it)."""
    a = np.ascontiguousarray(pmf, dtype=np.float32)
    if a.ndim != 1 or a.size == 0:
        raise ValueError("pmf must be a non-empty 1-D sequence")
    out = np.zeros(a.size + 1, np.uint32)
    rc = _lib.lib().fgmm_pmf_to_quantized_cdf(a.ctypes.data_as(C.c_void_p), a.size, int(precision),
                                               out.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError("Invalid `pmf`: negative, non-finite or all-zero elements (or more symbols than counts)")
    return out.tolist()

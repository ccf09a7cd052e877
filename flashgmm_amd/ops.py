"""``pmf_to_quantized_cdf`` — mirror of ``compressai._CXX.pmf_to_quantized_cdf`` (compressai/cpp_exts/ops/ops.cpp:40-109),
the table builder behind ``EntropyBottleneck.update()`` / ``GaussianConditional.update()`` (the `z` hyper-latent path).
Runs once per model; host code in libflashgmm_amd.so."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib

__all__ = ["pmf_to_quantized_cdf"]


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

"""Test helpers (CPU): the decode-side table FORMAT of include/flashgmm_amd.h, built from the oracle's full
edge table with numpy, so the host rANS code can be tested without a GPU, and the inverse (expand a trimmed
table back to the full one) so GPU-built tables can be compared with the oracle's."""
from __future__ import annotations

import ctypes as C

import numpy as np


def trim_full_table(tab: np.ndarray, max_bs: int):
    """full table [n, W=2*max_bs+2] (F_i[v], v=-max_bs..max_bs+1) -> (hdr uint64[n], pool uint16[...]) exactly as
    cdftab_kernel lays them out (rows padded to 4 entries with their last value, offsets in row order)."""
    n, W = tab.shape
    assert W == 2 * max_bs + 2
    hdr = np.zeros(n, np.uint64)
    rows = []
    off = 0
    for i in range(n):
        F = tab[i].astype(np.int64)
        nzpos = np.nonzero(F)[0]
        lead = (nzpos[0] - 1) if len(nzpos) else W - 1  # index of the last leading zero (-1: none)
        diff = np.nonzero(F != F[-1])[0]
        run_start = (diff[-1] + 1) if len(diff) else 0  # start of the trailing constant run
        a_idx = min(max(lead, 0), run_start)
        cnt = run_start - a_idx + 1
        row = F[a_idx:a_idx + cnt]
        nonmono = int((np.diff(row) < 0).any())
        pad = (-cnt) % 4
        rows.append(np.concatenate([row, np.full(pad, row[-1])]).astype(np.uint16))
        a = a_idx - max_bs
        hdr[i] = np.uint64((a & 0xFFFF) | ((cnt | (nonmono << 15)) << 16) | ((off >> 2) << 32))
        off += cnt + pad
    pool = np.concatenate(rows + [np.zeros(32, np.uint16)]) if rows else np.zeros(32, np.uint16)
    return hdr, pool, off


def expand_trimmed(hdr: np.ndarray, pool: np.ndarray, max_bs: int) -> np.ndarray:
    """(hdr, pool) -> full table [n, 2*max_bs+2] (the virtual F of the header comment)."""
    n = len(hdr)
    W = 2 * max_bs + 2
    out = np.zeros((n, W), np.uint16)
    for i in range(n):
        h = int(hdr[i])
        a = h & 0xFFFF
        a = a - 65536 if a >= 32768 else a
        cnt = (h >> 16) & 0x7FFF
        off = (h >> 32) << 2
        row = pool[off:off + cnt]
        j0 = a + max_bs
        out[i, j0:j0 + cnt] = row
        out[i, j0 + cnt:] = row[-1]
    return out


def host_encode_symtab(lib, packed: np.ndarray, symbols) -> bytes:
    packed = np.ascontiguousarray(packed, np.uint32)
    out, out_len = C.c_void_p(), C.c_size_t()
    sp = None
    if symbols is not None:
        symbols = np.ascontiguousarray(symbols, np.int32)
        sp = symbols.ctypes.data_as(C.c_void_p)
    rc = lib.fgmm_rans_encode_symtab(packed.ctypes.data_as(C.c_void_p), sp, len(packed), C.byref(out), C.byref(out_len))
    assert rc == 0, rc
    data = C.string_at(out, out_len.value)
    lib.fgmm_free(out)
    return data


def host_decode_cdftab(lib, enc: bytes, hdr: np.ndarray, pool: np.ndarray, max_bs: int):
    hdr = np.ascontiguousarray(hdr, np.uint64)
    pool = np.ascontiguousarray(pool, np.uint16)
    out = np.empty(len(hdr), np.int32)
    rc = lib.fgmm_rans_decode_cdftab(enc, len(enc), hdr.ctypes.data_as(C.c_void_p), pool.ctypes.data_as(C.c_void_p),
                                     len(hdr), max_bs, out.ctypes.data_as(C.c_void_p))
    return rc, out

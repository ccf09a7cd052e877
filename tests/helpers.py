"""Test helpers (CPU): the decode-side table FORMAT of include/flashgmm_amd.h, built from the oracle's full
edge table with numpy, so the host rANS code can be tested without a GPU, and the inverse (expand a trimmed
table back to the full one) so GPU-built tables can be compared with the oracle's."""
from __future__ import annotations

import ctypes as C

import numpy as np


EF_MIN = 64


def row_is_ef(cnt: int, nonmono: int) -> bool:
    return cnt >= EF_MIN and not nonmono


def row_bytes(cnt: int, nonmono: int) -> int:
    if row_is_ef(cnt, nonmono):
        return ((cnt + 7) & ~7) + 8 * ((cnt + 256 + 63) >> 6)
    return 2 * ((cnt + 1) & ~1)


def trim_full_table(tab: np.ndarray, max_bs: int):
    """full table [n, W=2*max_bs+2] (F_i[v], v=-max_bs..max_bs+1) -> (hdr uint32[n], pool uint8[...], used bytes),
    format v3 of include/flashgmm_amd.h exactly as the cdftab kernels lay it out."""
    n, W = tab.shape
    assert W == 2 * max_bs + 2
    hdr = np.zeros(n, np.uint32)
    chunks = []
    for i in range(n):
        F = tab[i].astype(np.int64)
        nzpos = np.nonzero(F)[0]
        lead = (nzpos[0] - 1) if len(nzpos) else W - 1  # index of the last leading zero (-1: none)
        diff = np.nonzero(F != F[-1])[0]
        run_start = (diff[-1] + 1) if len(diff) else 0  # start of the trailing constant run
        a_idx = min(lead + 1, run_start)  # first non-zero edge (the zero before it is implied: F[v < a] = 0)
        cnt = run_start - a_idx + 1
        row = F[a_idx:a_idx + cnt]
        nonmono = int((np.diff(row) < 0).any())
        a = a_idx - max_bs
        hdr[i] = (a & 0xFFFF) | (cnt << 16) | (nonmono << 31)
        if row_is_ef(cnt, nonmono):
            lows = np.zeros((cnt + 7) & ~7, np.uint8)
            lows[:cnt] = row & 0xFF
            U = (cnt + 256 + 63) >> 6
            bits = np.zeros(U * 64, np.uint8)
            bits[(row >> 8) + np.arange(cnt)] = 1
            up = np.packbits(bits.reshape(U, 64)[:, ::-1], axis=1).view(">u8").astype("<u8").reshape(-1)
            chunks += [lows, up.view(np.uint8)]
        else:
            pad = (-cnt) % 2
            chunks.append(np.concatenate([row, np.full(pad, row[-1])]).astype("<u2").view(np.uint8))
    used = sum(len(c) for c in chunks)
    pool = np.concatenate(chunks + [np.zeros(128, np.uint8)]) if chunks else np.zeros(128, np.uint8)
    return hdr, pool, used


def expand_trimmed(hdr: np.ndarray, pool: np.ndarray, max_bs: int) -> np.ndarray:
    """(hdr, pool) -> full table [n, 2*max_bs+2] (the virtual F of the header comment)."""
    n = len(hdr)
    W = 2 * max_bs + 2
    out = np.zeros((n, W), np.uint16)
    pool = np.asarray(pool).view(np.uint8)
    off = 0
    for i in range(n):
        h = int(hdr[i])
        a = h & 0xFFFF
        a = a - 65536 if a >= 32768 else a
        cnt = (h >> 16) & 0x7FFF
        nonmono = h >> 31
        if row_is_ef(cnt, nonmono):
            lb = (cnt + 7) & ~7
            U = (cnt + 256 + 63) >> 6
            lows = pool[off:off + cnt].astype(np.int64)
            up = pool[off + lb:off + lb + 8 * U].view("<u8")
            bits = np.unpackbits(up.astype(">u8").view(np.uint8).reshape(U, 8), axis=1)[:, ::-1].reshape(-1)
            pos = np.nonzero(bits)[0]
            assert len(pos) == cnt, (i, len(pos), cnt)
            row = ((pos - np.arange(cnt)) << 8) | lows
        else:
            row = pool[off:off + 2 * cnt].view("<u2").astype(np.int64)
        off += row_bytes(cnt, nonmono)
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
    hdr = np.ascontiguousarray(hdr, np.uint32)
    pool = np.ascontiguousarray(np.asarray(pool).view(np.uint8))
    out = np.empty(len(hdr), np.int32)
    rc = lib.fgmm_rans_decode_cdftab(enc, len(enc), hdr.ctypes.data_as(C.c_void_p), pool.ctypes.data_as(C.c_void_p),
                                     len(hdr), max_bs, out.ctypes.data_as(C.c_void_p))
    return rc, out

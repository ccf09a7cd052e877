"""Test helpers (CPU): the decode-side table FORMAT of include/flashgmm_amd.h, built from the oracle's full
edge table with numpy, so the host rANS code can be tested without a GPU, and the inverse (expand a trimmed
table back to the full one) so GPU-built tables can be compared with the oracle's."""
from __future__ import annotations

import ctypes as C

import numpy as np


EF_MIN = 14
MAX_BS_H4 = 16382


def hdr_form(max_bs: int) -> int:
    return 2 if 2 * max_bs + 2 <= 254 else (4 if max_bs <= MAX_BS_H4 else 8)


def row_is_ef(cnt: int, nonmono: int) -> bool:
    return cnt >= EF_MIN and not nonmono


def ef_l(cnt: int) -> int:
    """low bits of an Elias-Fano row: 12 up to 48 entries, 8 beyond"""
    return 12 if cnt <= 48 else 8


def row_bytes(cnt: int, nonmono: int) -> int:
    if row_is_ef(cnt, nonmono):
        l = ef_l(cnt)
        return 2 * ((cnt * l + ((cnt + (65536 >> l) + 7) & ~7) + 15) >> 4)
    return 2 * cnt


def _trim_row(F: np.ndarray):
    """full row F[0..W) -> (a_idx, cnt, nonmono): first non-zero edge .. start of the trailing constant run"""
    W = len(F)
    nzpos = np.nonzero(F)[0]
    lead = (nzpos[0] - 1) if len(nzpos) else W - 1  # index of the last leading zero (-1: none)
    diff = np.nonzero(F != F[-1])[0]
    run_start = (diff[-1] + 1) if len(diff) else 0  # start of the trailing constant run
    a_idx = min(lead + 1, run_start)  # first non-zero edge (the zero before it is implied: F[v < a] = 0)
    cnt = run_start - a_idx + 1
    nonmono = int((np.diff(F[a_idx:a_idx + cnt]) < 0).any())
    return int(a_idx), int(cnt), nonmono


def _row_payload(row: np.ndarray, nonmono: int):
    cnt = len(row)
    if row_is_ef(cnt, nonmono):
        l = ef_l(cnt)
        HB = cnt + (65536 >> l)
        nbytes = row_bytes(cnt, nonmono)
        bits = np.zeros(8 * nbytes, np.uint8)
        bits[(row >> l) + np.arange(cnt)] = 1
        lows = ((row[:, None] >> np.arange(l)[None, :]) & 1).astype(np.uint8)  # [cnt, l], bit 0 first
        LB = (HB + 7) & ~7
        bits[LB:LB + cnt * l] = lows.reshape(-1)
        return [np.packbits(bits, bitorder="little")]
    return [row.astype("<u2").view(np.uint8)]


def _pack_hdr4(a: int, cnt: int, nonmono: int) -> int:
    return (a & 0xFFFF) | (cnt << 16) | (nonmono << 31)


def trim_full_table(tab: np.ndarray, max_bs: int, form: int = 4, tl: int = 0, shuffle_seed=None):
    """full table [n, W=2*max_bs+2] (F_i[v], v=-max_bs..max_bs+1) -> (hdr, pool uint8[...], used bytes) in format v5 of
    include/flashgmm_amd.h: `form`-byte headers; rows sequential in latent order (tl = 0) or, with tl > 0, per block of tl
    latents at 4 * blk_off[block] — then (hdr, blk_off, pool, used) is returned, the blocks placed in a shuffled order
    when shuffle_seed is given (the single-pass kernel places them in no particular order)."""
    n, W = tab.shape
    assert W == 2 * max_bs + 2
    dt = {2: np.uint16, 4: np.uint32, 8: np.uint64}[form]
    hdr = np.zeros(n, dt)
    rows = []
    for i in range(n):
        F = tab[i].astype(np.int64)
        a_idx, cnt, nonmono = _trim_row(F)
        row = F[a_idx:a_idx + cnt]
        a = a_idx - max_bs
        chunks = _row_payload(row, nonmono)
        if form == 2:
            assert W <= 254
            hdr[i] = a_idx | ((255 if nonmono else cnt) << 8)
            if nonmono:
                chunks = [np.array([_pack_hdr4(a, cnt, 1)], "<u4").view(np.uint8)] + chunks
        elif form == 4:
            hdr[i] = _pack_hdr4(a, cnt, nonmono)
        else:
            hdr[i] = (a & 0xFFFFFFFF) | ((cnt | (nonmono << 31)) << 32)
        rows.append(np.concatenate(chunks))
    if not tl:
        used = sum(len(c) for c in rows)
        pool = np.concatenate(rows + [np.zeros(128, np.uint8)]) if rows else np.zeros(128, np.uint8)
        return hdr, pool, used
    nblk = (n + tl - 1) // tl
    order = np.arange(nblk)
    if shuffle_seed is not None:
        np.random.default_rng(shuffle_seed).shuffle(order)
    blk_off = np.zeros(nblk, np.uint32)
    parts, off = [], 0
    for b in order:
        blk_off[b] = off // 4
        for r in rows[b * tl:(b + 1) * tl]:
            parts.append(r)
            off += len(r)
        if off % 4:  # blocks start 4-byte aligned
            parts.append(np.zeros(2, np.uint8))
            off += 2
    pool = np.concatenate(parts + [np.zeros(128, np.uint8)]) if parts else np.zeros(128, np.uint8)
    return hdr, blk_off, pool, off


def expand_trimmed(hdr: np.ndarray, pool: np.ndarray, max_bs: int, blk_off=None, tl: int = 0) -> np.ndarray:
    """(hdr, pool[, blk_off, tl]) -> full table [n, 2*max_bs+2] (the virtual F of the header comment); the header form is
    taken from hdr.dtype."""
    n = len(hdr)
    W = 2 * max_bs + 2
    form = hdr.dtype.itemsize
    out = np.zeros((n, W), np.uint16)
    pool = np.asarray(pool).view(np.uint8)
    off = 0
    for i in range(n):
        if blk_off is not None and i % tl == 0:
            off = 4 * int(blk_off[i // tl])
        h = int(hdr[i])
        if form == 2:
            a, cnt, nonmono = (h & 0xFF) - max_bs, h >> 8, 0
            if cnt == 255:
                h4 = int(pool[off:off + 4].view("<u4")[0])
                off += 4
                a = h4 & 0xFFFF
                a = a - 65536 if a >= 32768 else a
                cnt, nonmono = (h4 >> 16) & 0x7FFF, h4 >> 31
        elif form == 4:
            a = h & 0xFFFF
            a = a - 65536 if a >= 32768 else a
            cnt, nonmono = (h >> 16) & 0x7FFF, h >> 31
        else:
            a = h & 0xFFFFFFFF
            a = a - (1 << 32) if a >= (1 << 31) else a
            cnt, nonmono = (h >> 32) & 0x7FFFFFFF, h >> 63
        if row_is_ef(cnt, nonmono):
            l = ef_l(cnt)
            HB = cnt + (65536 >> l)
            bits = np.unpackbits(pool[off:off + row_bytes(cnt, nonmono)], bitorder="little").astype(np.int64)
            pos = np.nonzero(bits[:HB])[0]
            assert len(pos) == cnt, (i, len(pos), cnt)
            LB = (HB + 7) & ~7
            lows = (bits[LB:LB + cnt * l].reshape(cnt, l) << np.arange(l)[None, :]).sum(1)
            assert not bits[HB:LB].any() and not bits[LB + cnt * l:].any(), i  # the padding is zero
            row = ((pos - np.arange(cnt)) << l) | lows
        else:
            row = pool[off:off + 2 * cnt].copy().view("<u2").astype(np.int64)
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


def _flags():
    return 4 if EF_MIN > 65536 else 0  # FGMM_TAB_RAW_ROWS when the tables were built without Elias-Fano rows


def host_decode_cdftab(lib, enc: bytes, hdr: np.ndarray, pool: np.ndarray, max_bs: int, pool_len=None):
    """the public 4-byte-header / sequential-rows entry point"""
    hdr = np.ascontiguousarray(hdr, np.uint32)
    pool = np.ascontiguousarray(np.asarray(pool).view(np.uint8))
    out = np.empty(len(hdr), np.int32)
    rc = lib.fgmm_rans_decode_cdftab(enc, len(enc), hdr.ctypes.data_as(C.c_void_p), pool.ctypes.data_as(C.c_void_p),
                                     len(pool) if pool_len is None else pool_len, len(hdr), max_bs, _flags(), out.ctypes.data_as(C.c_void_p))
    return rc, out


def host_decode_tab(lib, enc: bytes, hdr: np.ndarray, pool: np.ndarray, max_bs: int, blk_off=None, tl: int = 0, pool_len=None):
    """any header form (taken from hdr.dtype), rows sequential or block-placed"""
    hdr = np.ascontiguousarray(hdr)
    pool = np.ascontiguousarray(np.asarray(pool).view(np.uint8))
    out = np.empty(len(hdr), np.int32)
    bo = None
    if blk_off is not None:
        blk_off = np.ascontiguousarray(blk_off, np.uint32)
        bo = blk_off.ctypes.data_as(C.c_void_p)
    rc = lib.fgmm_rans_decode_tab(enc, len(enc), hdr.ctypes.data_as(C.c_void_p), hdr.dtype.itemsize, bo, tl,
                                  pool.ctypes.data_as(C.c_void_p), len(pool) if pool_len is None else pool_len, len(hdr), max_bs,
                                  _flags(), out.ctypes.data_as(C.c_void_p))
    return rc, out


def expected_threads(budget: dict, ranks: int = 1) -> int:
    """fgmm_host_thread_budget, restated: the share's CPUs; x3 (up to the share of the affinity mask) when a cgroup quota is the limit"""
    by_time, by_mask = int(budget["cpus"] / ranks + 1e-9), int(budget["affinity"] / ranks + 1e-9)
    import os
    per_cpu = min(max(int(os.environ.get("FGMM_WORKERS_PER_CPU", "3")), 1), 4)
    t = min(by_mask, per_cpu * by_time) if budget.get("quota") and by_mask > by_time else by_time
    return max(1, min(t, 48))

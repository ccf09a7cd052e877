"""CPU (-m "not gpu"): the product's HOST logic and ABI surface, no GPU compute.

  * libflashgmm_amd.so loads and exports every symbol include/flashgmm_amd.h declares;
  * with no HIP device the GPU entry points fail loudly (no CPU fallback exists);
  * the integer-only host rANS coder (fgmm_rans_encode_symtab / fgmm_rans_decode_cdftab) reproduces the
    reference's streams / symbols from tables — tables here come from the oracle, formatted by tests/helpers.py.
"""
import ctypes as C
import hashlib
import json
import os
import re

import numpy as np
import pytest

from flashgmm_amd import _lib
from tests import synth as T
import helpers
from helpers import expand_trimmed, host_decode_cdftab, host_decode_tab, host_encode_symtab, trim_full_table

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = ["polya", "as", "logistic"]


def test_compiled_python_boundary_loads_and_matches_the_library():
    """flashgmm_amd._native (pybind11, csrc/fgmm_pybind.cpp) is built beside the library, resolves it through its $ORIGIN rpath, reports
    the library's ABI version and exposes the three batched calls; FGMM_NATIVE=0 turns it off (the ctypes binding then does the work)"""
    import subprocess
    import sys

    nat = _lib.native()
    assert nat is not None, "flashgmm_amd/_native*.so missing or not loadable: flashgmm_amd/csrc/build.sh builds it"
    assert nat.abi_version == _lib.lib().fgmm_abi_version()
    for fn in ("compress_stacked", "compress_head_stacked", "decompress_stacked"):
        assert callable(getattr(nat, fn))
    out = subprocess.run([sys.executable, "-c", "from flashgmm_amd import _lib; print(_lib.native())"], env=dict(os.environ, FGMM_NATIVE="0"),
                         capture_output=True, text=True, cwd=ROOT)
    assert out.stdout.strip() == "None", out.stderr[-500:]


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    header = open(os.path.join(ROOT, "include", "flashgmm_amd.h")).read()
    declared = set(re.findall(r"\b(fgmm_[a-z0-9_]+)\s*\(", header))
    declared -= {"fgmm_status", "fgmm_mode", "fgmm_memspace"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert L.fgmm_abi_version() == 6  # (v6: + the parameter head, section 2b of the header)


def test_header_is_plain_c_and_links():
    """include/flashgmm_amd.h is the drop-in boundary: it must compile as C99 (no torch / C++ types in the signatures) and
    a C program calling it must link against the library and run (no GPU needed for the host-only entry points)"""
    import shutil
    import subprocess
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "flashgmm_amd.h")
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    src = r"""
#include "flashgmm_amd.h"
#include <stdio.h>
#include <string.h>
int main(void) {
  /* an empty symbol table flushes the initial rANS state: 00 00 00 80 00 00 00 00 (rans_interface.cpp:557-585) */
  uint8_t *out = NULL; size_t len = 0; uint32_t none = 0;
  if (fgmm_rans_encode_symtab(&none, NULL, 0, &out, &len) != FGMM_OK || len != 8) return 1;
  static const unsigned char want[8] = {0, 0, 0, 0x80, 0, 0, 0, 0};
  if (memcmp(out, want, 8)) return 2;
  fgmm_free(out);
  float pmf[4] = {0.1f, 0.2f, 0.f, 0.f}; uint32_t cdf[5];
  if (fgmm_pmf_to_quantized_cdf(pmf, 4, 16, cdf) != FGMM_OK || cdf[1] != 21845 || cdf[4] != 65536) return 3;
  printf("abi %d ok\n", fgmm_abi_version());
  return 0;
}
"""
    lib_dir = os.path.join(root, "flashgmm_amd")
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), c, "-o", exe, "-L", lib_dir,
                        "-lflashgmm_amd", "-Wl,-rpath," + lib_dir], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_no_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.FgmmError, match="NO_DEVICE"):
        _lib.ctx(-1)
    from flashgmm_amd import GaussianMixtureConditional

    y, sg, mu, pi = (torch.from_numpy(a) for a in T.make_latent(3, M=8, h=4, w=4))
    with pytest.raises(RuntimeError, match="GPU only|NO_DEVICE"):
        GaussianMixtureConditional(K=4).compress(y, sg, mu, pi)
    from flashgmm_amd import ans

    s = torch.rand(5, 4) + 0.2
    with pytest.raises(RuntimeError):
        ans.RansEncoder().encode_with_indexes_gmm(torch.zeros(5, dtype=torch.int32), s, s, s, 1)


def test_product_does_not_import_the_oracle():
    """the product path may not route through oracle/ (grep-level guard)"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "flashgmm_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".sh")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "fgmm_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


@pytest.mark.parametrize("mode", MODES)
def test_host_encoder_from_oracle_tables(oracle, mode):
    L = _lib.lib()
    # KA-1 Kodak half: md5 of the reference
    ent = json.load(open(os.path.join(GOLD, "ka1.json")))[mode]["1234"]
    y, sg, mu, pi = T.make_latent(1234)
    sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
    packed = oracle.symtab(mode, sym, s, m, w)
    b = host_encode_symtab(L, packed, None)  # |symbols| < 32768: bypass values come from the table
    assert (len(b), hashlib.md5(b).hexdigest()) == (ent["len"], ent["md5"])
    assert host_encode_symtab(L, packed, sym) == b


@pytest.mark.parametrize("mode", MODES)
def test_host_encoder_small_and_wide_bypass(oracle, mode):
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = json.load(open(os.path.join(GOLD, "g3_small.json")))["cases"]
    L = _lib.lib()
    for name, (sym, s, m, w) in mg.g3_cases().items():
        packed = oracle.symtab(mode, sym, s, m, w)
        assert host_encode_symtab(L, packed, sym).hex() == gold[name][mode]["hex"], name  # incl. 2^30, -2^31 bypass


def test_host_pair_encoder_equals_single(oracle):
    """fgmm_rans_encode_symtab2 (two streams coded in turn by one thread): each output == the single-stream coder's,
    for equal and unequal lengths, empty tables, dense bypass entries and wide (raw-symbol) bypass values"""
    import ctypes as C
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    L = _lib.lib()
    tabs = []
    for seed, (M, h, w) in ((1234, (192, 32, 24)), (3, (40, 9, 7)), (4, (7, 5, 3))):
        y, sg, mu, pi = T.make_latent(seed, M=M, h=h, w=w)
        sym, s, m, wt, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
        tabs.append((oracle.symtab("as", sym, s, m, wt), None))
    for name, (sym, s, m, wt) in mg.g3_cases().items():  # n = 0, 1, 17, 1000; forced bypass incl. 2^30 and -2^31
        tabs.append((oracle.symtab("polya", sym, s, m, wt), sym))
    rng = np.random.default_rng(5)
    dense = rng.integers(0, 1 << 32, 5000, dtype=np.uint64).astype(np.uint32)
    dense[rng.random(5000) < 0.3] &= 0xFFFF  # 30 % bypass entries (range == 0)
    dense[(dense >> 16) != 0] = (dense[(dense >> 16) != 0] & 0xFFFF0FFF)  # keep start + range <= 65535 plausible
    ok = ((dense & 0xFFFF).astype(np.int64) + (dense >> 16).astype(np.int64)) <= 65535
    dense[~ok] = 0x00010000
    tabs.append((dense, None))
    single = [host_encode_symtab(L, p, sy) for p, sy in tabs]

    def ptr(a):
        return None if a is None else np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)

    for i in range(len(tabs)):
        for j in range(len(tabs)):
            (p0, s0), (p1, s1) = tabs[i], tabs[j]
            p0, p1 = np.ascontiguousarray(p0, np.uint32), np.ascontiguousarray(p1, np.uint32)
            s0 = None if s0 is None else np.ascontiguousarray(s0, np.int32)
            s1 = None if s1 is None else np.ascontiguousarray(s1, np.int32)
            o0, o1, l0, l1 = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()
            rc = L.fgmm_rans_encode_symtab2(ptr(p0), ptr(s0), len(p0), ptr(p1), ptr(s1), len(p1), C.byref(o0), C.byref(l0),
                                            C.byref(o1), C.byref(l1))
            assert rc == 0
            b0, b1 = C.string_at(o0, l0.value), C.string_at(o1, l1.value)
            L.fgmm_free(o0); L.fgmm_free(o1)
            assert b0 == single[i] and b1 == single[j], (i, j)
    # fgmm_rans_encode_symtab_n: one to four tables coded in turn (what the batched encoder does with ceil(streams / workers))
    rng = np.random.default_rng(6)
    for ways in (1, 2, 3, 4):
        for _ in range(12):
            pick = rng.integers(0, len(tabs), ways)
            ps = [np.ascontiguousarray(tabs[k][0], np.uint32) for k in pick]
            ss = [None if tabs[k][1] is None else np.ascontiguousarray(tabs[k][1], np.int32) for k in pick]
            P = (C.c_void_p * ways)(*[p_.ctypes.data for p_ in ps])
            S = (C.c_void_p * ways)(*[None if s_ is None else s_.ctypes.data for s_ in ss])
            N = (C.c_int64 * ways)(*[len(p_) for p_ in ps])
            O, LN = (C.c_void_p * ways)(), (C.c_size_t * ways)()
            assert L.fgmm_rans_encode_symtab_n(ways, P, S, N, O, LN) == 0
            for q in range(ways):
                assert C.string_at(O[q], LN[q]) == single[pick[q]], (ways, pick)
                L.fgmm_free(O[q])
    assert L.fgmm_rans_encode_symtab_n(5, P, S, N, O, LN) == 1 and L.fgmm_rans_encode_symtab_n(0, P, S, N, O, LN) == 1


@pytest.mark.parametrize("mode", MODES)
def test_host_decoder_from_oracle_tables(oracle, mode):
    L = _lib.lib()
    y, sg, mu, pi = T.make_latent(11, M=48, h=16, w=8)
    sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
    enc = oracle.encode_gmm(mode, sym, s, m, w)
    max_bs = abs_max + 1
    tab = oracle.cdftab(mode, s, m, w, max_bs)
    hdr, pool, used = trim_full_table(tab, max_bs)
    assert np.array_equal(expand_trimmed(hdr, pool, max_bs), tab)  # the format is lossless
    rc, out = host_decode_cdftab(L, enc, hdr, pool, max_bs)
    assert rc == 0 and np.array_equal(out, sym)
    assert used < tab.size * 2  # and it is a real trim (bytes)


def test_host_decoder_matches_reference_bisection_on_hostile_tables(oracle):
    """Non-monotone rows, zero-width brackets and cum_freq values no interval contains: the host decoder must
    return what the reference's bisection returns (oracle.rans_decode_cdftab is its literal replay)."""
    L = _lib.lib()
    rng = np.random.default_rng(5)
    n, max_bs = 4000, 9
    W = 2 * max_bs + 2
    tab = np.sort(rng.integers(0, 65536, (n, W)), axis=1).astype(np.uint16)
    tab[: n // 2, :3] = 0
    tab[n // 4: n // 2, -4:] = tab[n // 4: n // 2, -5:-4]
    # hostile: swap two entries in a third of the rows (non-monotone), flat rows, rows not starting at 0
    for i in range(0, n, 3):
        j = rng.integers(1, W - 1)
        tab[i, j], tab[i, j - 1] = tab[i, j - 1], tab[i, j]
    tab[5::50] = 1234
    tab[7::50] = 0
    enc = rng.integers(0, 256, 4 * (n + 64), dtype=np.uint8).tobytes()  # garbage stream: exercises every fallback
    hdr, pool, _ = trim_full_table(tab, max_bs)
    assert np.array_equal(expand_trimmed(hdr, pool, max_bs), tab)
    want = oracle.rans_decode_cdftab(enc, tab, max_bs)
    rc, out = host_decode_cdftab(L, enc, hdr, pool, max_bs)
    assert rc == 0
    assert np.array_equal(out, want)


def test_host_decoder_elias_fano_rows(oracle):
    """wide monotone rows are Elias-Fano coded: search, neighbour reconstruction and the expand-and-bisect fallback
    (garbage stream => cum_freq values no interval contains) must all reproduce the reference's bisection"""
    L = _lib.lib()
    rng = np.random.default_rng(8)
    n, max_bs = 3000, 99
    W = 2 * max_bs + 2
    tab = np.sort(rng.integers(0, 65536, (n, W)), axis=1).astype(np.uint16)
    tab[::3, :40] = 0                       # leading zeros: windows start later
    tab[1::3, 150:] = tab[1::3, 149:150]    # trailing constant run
    tab[2::7] = np.sort(rng.integers(0, 300, (len(tab[2::7]), W)), axis=1)  # many duplicates, tiny high parts
    tab[4::11, 1:] = 65535                  # everything in the last bucket
    hdr, pool, used = trim_full_table(tab, max_bs)
    cnt = (hdr >> 16) & 0x7FFF
    assert (cnt >= 48).mean() > 0.9 and not (hdr >> 31).any()
    assert np.array_equal(expand_trimmed(hdr, pool, max_bs), tab)
    enc = rng.integers(0, 256, 4 * (n + 64), dtype=np.uint8).tobytes()
    want = oracle.rans_decode_cdftab(enc, tab, max_bs)
    rc, out = host_decode_cdftab(L, enc, hdr, pool, max_bs)
    assert rc == 0 and np.array_equal(out, want)


@pytest.mark.parametrize("shape", ["uniform", "gaussian"])
def test_host_decoder_elias_fano_rows_of_every_width(oracle, shape):
    """format v5: the low part of an Elias-Fano row has 12 bits (14..48 entries: a one-word unary part) or 8 bits (longer), rows are
    2-byte aligned: rows of every count from 1 to the full window, uniform entries and CDF-shaped ones (dense tails)"""
    L = _lib.lib()
    rng = np.random.default_rng(11)
    max_bs = 99
    W = 2 * max_bs + 2
    n = 6 * W
    tab = np.zeros((n, W), np.uint16)
    for i in range(n):
        cnt = 1 + i % W
        a = int(rng.integers(0, W - cnt + 1))
        if shape == "uniform":
            row = np.sort(rng.integers(1, 65536, cnt))
        else:
            from math import erf
            s = cnt / rng.uniform(3.0, 7.0)
            row = np.array([32768 * (1 + erf((k - cnt / 2) / (s * 2 ** 0.5 + 1e-9))) for k in range(cnt)]).astype(np.int64)
            row = np.clip(np.maximum.accumulate(row), 1, 65535)
        tab[i, a:a + cnt] = row
        tab[i, a + cnt:] = row[-1]
    hdr, pool, used = trim_full_table(tab, max_bs)
    cnts = (hdr >> 16) & 0x7FFF
    from helpers import ef_l
    assert {ef_l(int(k)) for k in cnts if k >= helpers.EF_MIN} == {8, 12}
    assert np.array_equal(expand_trimmed(hdr, pool, max_bs), tab)
    enc = rng.integers(0, 256, 4 * (n + 64), dtype=np.uint8).tobytes()
    want = oracle.rans_decode_cdftab(enc, tab, max_bs)
    rc, out = host_decode_cdftab(L, enc, hdr, pool, max_bs)
    assert rc == 0 and np.array_equal(out, want)
    # and a real bitstream: symbols drawn from each row's own distribution (every row searched where its mass is)
    t64 = tab.astype(np.int64)
    sym = np.empty(n, np.int32)
    packed = np.empty(n, np.uint32)
    for i in range(n):
        pmf = np.diff(t64[i])                      # symbol v = j - max_bs has the interval [F[j], F[j+1])
        j = int(rng.choice(np.nonzero(pmf > 0)[0])) if (pmf > 0).any() else 0
        sym[i] = j - max_bs
        packed[i] = int(t64[i, j]) | (int(max(pmf[j], 1)) << 16)
    enc2 = oracle.rans_encode_symtab(packed, sym)
    rc, out = host_decode_cdftab(L, enc2, hdr, pool, max_bs)
    assert rc == 0 and np.array_equal(out, oracle.rans_decode_cdftab(enc2, tab, max_bs))
    assert np.array_equal(out[np.diff(t64, axis=1).max(1) > 0], sym[np.diff(t64, axis=1).max(1) > 0])


def test_host_decoder_long_elias_fano_rows_mixture_shaped(oracle):
    """rows of 49 .. 254 entries with 8 low bits (a 256-bucket unary part of up to eight words): CDFs of two-component mixtures - a wide component under a narrow heavy one, so that one symbol jumps
    over dozens of EMPTY buckets (its neighbours lie words away) while the tails crowd many entries into one bucket - searched
    with the cum_freq values of garbage streams and of real streams; every answer is the reference bisection's."""
    from math import erf

    L = _lib.lib()
    rng = np.random.default_rng(33)
    max_bs = 126
    W = 2 * max_bs + 2  # 254: the longest row the 2-byte headers carry
    n = 6000
    tab = np.zeros((n, W), np.uint16)
    v = np.arange(W) - max_bs - 0.5
    for i in range(n):
        s_wide, s_nar = rng.uniform(8.0, 45.0), rng.uniform(0.11, 1.5)
        m_wide, m_nar = rng.uniform(-20, 20), rng.uniform(-30, 30)
        w_nar = rng.choice([0.0, 0.2, 0.6, 0.95])
        cdf = np.array([(1 - w_nar) * 0.5 * (1 + erf((x - m_wide) / (s_wide * 2 ** 0.5))) + w_nar * 0.5 * (1 + erf((x - m_nar) / (s_nar * 2 ** 0.5))) for x in v])
        tab[i] = np.minimum((cdf * 65535).astype(np.int64), 65535)
    hdr, bo, pool, used = trim_full_table(tab, max_bs, form=2, tl=32, shuffle_seed=7)
    cnts = (hdr >> 8).astype(np.int64)
    assert (cnts >= 49).mean() > 0.8 and cnts.max() > 200  # the rows this test is about
    assert np.array_equal(expand_trimmed(hdr, pool, max_bs, bo, 32), tab)
    for seed in (1, 2):
        enc = np.random.default_rng(seed).integers(0, 256, 4 * (n + 64), dtype=np.uint8).tobytes()
        rc, out = host_decode_tab(L, enc, hdr, pool, max_bs, bo, 32)
        assert rc == 0 and np.array_equal(out, oracle.rans_decode_cdftab(enc, tab, max_bs))
    t64 = tab.astype(np.int64)
    sym = np.empty(n, np.int32)
    packed = np.empty(n, np.uint32)
    for i in range(n):  # a real stream: every row asked where its mass is (the big jump most of the time)
        pmf = np.diff(t64[i])
        j = int(rng.choice(np.nonzero(pmf > 0)[0], p=pmf[pmf > 0] / pmf[pmf > 0].sum()))
        sym[i] = j - max_bs
        packed[i] = int(t64[i, j]) | (int(pmf[j]) << 16)
    enc2 = oracle.rans_encode_symtab(packed, sym)
    rc, out = host_decode_tab(L, enc2, hdr, pool, max_bs, bo, 32)
    assert rc == 0 and np.array_equal(out, sym)
    # a corrupted unary part is refused or decodes to garbage, never reads out of bounds (run under ASan: scripts/asan_host.sh)
    bad = pool.copy()
    flip = rng.integers(0, used, 4000)
    bad[flip] ^= (1 << rng.integers(0, 8, 4000)).astype(np.uint8)
    rc, _ = host_decode_tab(L, enc2, hdr, bad, max_bs, bo, 32)
    assert rc in (0, 1, 5)


@pytest.mark.parametrize("max_bs,tl", [(9, 16), (9, 48), (99, 32), (200, 32), (70000, 16)])
def test_host_decoder_header_forms_and_block_placement(oracle, max_bs, tl):
    """format v5: 2-byte headers (with the escape for non-monotone rows), 4- and 8-byte headers, rows placed block by
    block in a shuffled order behind blk_off — the decoder must give what the sequential 4-byte form gives, which is what
    the reference's bisection gives (garbage stream: every fallback is exercised)."""
    L = _lib.lib()
    rng = np.random.default_rng(max_bs + tl)
    n = 1500
    W = 2 * max_bs + 2
    if max_bs > 1000:  # a huge half-width: rows live in a small window of it
        tab = np.zeros((n, W), np.uint16)
        for i in range(n):
            c = int(rng.integers(0, W - 80))
            k = int(rng.integers(1, 70))
            tab[i, c:c + k] = np.sort(rng.integers(1, 65536, k))
            tab[i, c + k:] = tab[i, c + k - 1]
    else:
        tab = np.sort(rng.integers(0, 65536, (n, W)), axis=1).astype(np.uint16)
        tab[::3, : W // 5] = 0
        tab[1::3, -W // 4:] = tab[1::3, -W // 4 - 1: -W // 4]
        for i in range(0, n, 9):  # non-monotone rows (escape in the 2-byte form)
            j = rng.integers(1, W - 1)
            tab[i, j], tab[i, j - 1] = tab[i, j - 1], tab[i, j]
    enc = rng.integers(0, 256, 4 * (n + 64), dtype=np.uint8).tobytes()
    want = oracle.rans_decode_cdftab(enc, tab, max_bs)
    from helpers import hdr_form
    form = hdr_form(max_bs)
    assert form == {9: 2, 99: 2, 200: 4, 70000: 8}[max_bs]
    hdr, blk_off, pool, used = trim_full_table(tab, max_bs, form=form, tl=tl, shuffle_seed=3)
    assert hdr.dtype.itemsize == form
    if max_bs < 1000:
        assert np.array_equal(expand_trimmed(hdr, pool, max_bs, blk_off, tl), tab)
    rc, out = host_decode_tab(L, enc, hdr, pool, max_bs, blk_off, tl)
    assert rc == 0 and np.array_equal(out, want)
    hdr_s, pool_s, _ = trim_full_table(tab, max_bs, form=form)  # the same form, rows sequential
    rc, out = host_decode_tab(L, enc, hdr_s, pool_s, max_bs)
    assert rc == 0 and np.array_equal(out, want)


def test_host_decoder_raw_rows_only(oracle, monkeypatch):
    """FGMM_TAB_RAW_ROWS: tables without Elias-Fano rows (what a host-bound call ships) decode to the same symbols"""
    import helpers

    L = _lib.lib()
    rng = np.random.default_rng(12)
    n, max_bs = 2000, 99
    tab = np.sort(rng.integers(0, 65536, (n, 2 * max_bs + 2)), axis=1).astype(np.uint16)
    enc = rng.integers(0, 256, 4 * (n + 64), dtype=np.uint8).tobytes()
    want = oracle.rans_decode_cdftab(enc, tab, max_bs)
    hdr_e, bo_e, pool_e, used_e = trim_full_table(tab, max_bs, form=2, tl=48, shuffle_seed=5)
    monkeypatch.setattr(helpers, "EF_MIN", 1 << 30)
    hdr_r, bo_r, pool_r, used_r = trim_full_table(tab, max_bs, form=2, tl=48, shuffle_seed=5)
    assert used_r > 1.2 * used_e and np.array_equal(expand_trimmed(hdr_r, pool_r, max_bs, bo_r, 48), tab)
    rc, out = host_decode_tab(L, enc, hdr_r, pool_r, max_bs, bo_r, 48)
    assert rc == 0 and np.array_equal(out, want)
    hdr4, pool4, _ = trim_full_table(tab, max_bs)
    rc, out = host_decode_cdftab(L, enc, hdr4, pool4, max_bs)
    assert rc == 0 and np.array_equal(out, want)


def test_host_decoder_long_raw_rows_are_searched_in_logarithmic_time(oracle, monkeypatch):
    """Raw uint16 rows far beyond one vector (wide items of the generic path: max_bs comes out of a stream's side
    information): the search narrows by bisection before its SIMD compare — same symbols as the reference's bisection on
    random garbage streams, for rows of up to 3002 entries incl. long runs of equal entries, and no per-symbol cost
    proportional to the row length (a 2^16-entry row decodes as fast per symbol as a 300-entry one, within a factor)."""
    import time

    import helpers

    L = _lib.lib()
    rng = np.random.default_rng(21)
    monkeypatch.setattr(helpers, "EF_MIN", 1 << 30)
    for max_bs, n in ((1500, 300), (140, 600)):
        W = 2 * max_bs + 2
        tab = np.sort(rng.integers(0, 65536, (n, W)), axis=1).astype(np.uint16)
        tab[::3] = np.sort(rng.integers(0, 40, (len(tab[::3]), W)) * 1600, axis=1).astype(np.uint16)  # long runs of equal entries
        enc = rng.integers(0, 256, 4 * (n + 64), dtype=np.uint8).tobytes()
        want = oracle.rans_decode_cdftab(enc, tab, max_bs)
        hdr4, pool4, _ = trim_full_table(tab, max_bs)
        rc, out = host_decode_cdftab(L, enc, hdr4, pool4, max_bs)
        assert rc == 0 and np.array_equal(out, want), max_bs
    # cost per symbol against the row length (8-byte headers carry rows this long)
    per_symbol = {}
    for max_bs in (150, 32766):
        W, n = 2 * max_bs + 2, 400
        tab = np.sort(rng.integers(0, 65536, (n, W)), axis=1).astype(np.uint16)
        tab[:, 0], tab[:, -1] = 0, 65535
        enc = rng.integers(0, 256, 4 * (n + 64), dtype=np.uint8).tobytes()
        form = helpers.hdr_form(max_bs)
        hdr, bo, pool, used = trim_full_table(tab, max_bs, form=form, tl=16)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            rc, out = host_decode_tab(L, enc, hdr, pool, max_bs, bo, 16)
            best = min(best, time.perf_counter() - t0)
        assert rc == 0 and np.array_equal(out, oracle.rans_decode_cdftab(enc, tab, max_bs))
        per_symbol[max_bs] = best / n
    assert per_symbol[32766] < 20 * per_symbol[150] + 2e-6, per_symbol  # a linear search would be ~200x


def test_host_decoder_rejects_malformed_tables(oracle):
    """memory safety does not depend on the table being well-formed (include/flashgmm_amd.h): cnt = 0, a window outside the
    half-width, rows past the pool, inconsistent Elias-Fano rows -> FGMM_ERR_INVALID (1), never a wild read"""
    L = _lib.lib()
    rng = np.random.default_rng(2)
    n, max_bs = 64, 99
    tab = np.sort(rng.integers(0, 65536, (n, 2 * max_bs + 2)), axis=1).astype(np.uint16)
    hdr, pool, used = trim_full_table(tab, max_bs)
    enc = rng.integers(0, 256, 4 * (n + 64), dtype=np.uint8).tobytes()
    assert host_decode_cdftab(L, enc, hdr, pool, max_bs)[0] == 0
    bad = hdr.copy(); bad[5] &= 0x8000FFFF  # cnt = 0
    assert host_decode_cdftab(L, enc, bad, pool, max_bs)[0] == 1
    bad = hdr.copy(); bad[7] = (bad[7] & 0xFFFF0000) | ((-max_bs - 3) & 0xFFFF)  # a below -max_bs
    assert host_decode_cdftab(L, enc, bad, pool, max_bs)[0] == 1
    bad = hdr.copy(); bad[9] = (bad[9] & 0x8000FFFF) | (0x7FFF << 16)  # cnt far beyond the table
    assert host_decode_cdftab(L, enc, bad, pool, max_bs)[0] == 1
    assert host_decode_cdftab(L, enc, hdr, pool, max_bs, pool_len=used // 2)[0] == 1  # rows past the declared pool
    # Elias-Fano rows whose unary part is all ones / all zeros (no closing zero, no entries)
    for fill in (0xFF, 0x00):
        p2 = pool.copy()
        off = 0
        for i in range(n):
            cnt = (int(hdr[i]) >> 16) & 0x7FFF
            assert helpers.row_is_ef(cnt, 0)
            hb_bytes = (cnt + (65536 >> helpers.ef_l(cnt))) // 8
            if i % 2 == 0:
                p2[off: off + hb_bytes] = fill
            off += helpers.row_bytes(cnt, 0)
        rc, _ = host_decode_cdftab(L, enc, hdr, p2, max_bs)
        assert rc in (0, 1)  # no crash; whatever is decodable decodes, the rest is refused


def test_host_decoder_short_stream_is_an_error(oracle):
    L = _lib.lib()
    y, sg, mu, pi = T.make_latent(12, M=8, h=8, w=8)
    sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
    enc = oracle.encode_gmm("polya", sym, s, m, w)
    hdr, pool, _ = trim_full_table(oracle.cdftab("polya", s, m, w, abs_max + 1), abs_max + 1)
    rc, _ = host_decode_cdftab(L, enc[: len(enc) // 2 // 4 * 4], hdr, pool, abs_max + 1)
    assert rc == 5  # FGMM_ERR_STREAM
    rc, _ = host_decode_cdftab(L, b"\0\0\0", hdr, pool, abs_max + 1)
    assert rc == 5


def test_empty_table_flushes_initial_state():
    assert host_encode_symtab(_lib.lib(), np.zeros(0, np.uint32), None) == bytes.fromhex("0000008000000000")


# ---- the host's CPU budget: affinity mask and cgroup quota (fgmm_host_cpu_budget / fgmm_host_thread_budget) ----------
def _budget_in_child(sysroot, cpus=None, ranks=1):
    """the budget as a fresh process sees it (the affinity is the process's own; FGMM_SYSROOT stands in for "/")"""
    import subprocess
    import sys

    code = ("import os, sys, json\n"
            f"cpus = {sorted(cpus) if cpus else None}\n"
            "if cpus: os.sched_setaffinity(0, cpus)\n"
            f"sys.path.insert(0, {ROOT!r})\n"
            "from flashgmm_amd import _lib\n"
            f"print(json.dumps([_lib.host_cpu_budget(), _lib.lib().fgmm_host_thread_budget({ranks})]))\n")
    env = dict(os.environ, FGMM_SYSROOT=str(sysroot))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    import json
    return json.loads(r.stdout.strip().splitlines()[-1])


def _tree(root, files):
    for rel, text in files.items():
        path = os.path.join(root, rel.lstrip("/"))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)


def test_host_budget_reads_the_cgroup_v2_quota_of_the_group_and_its_ancestors(tmp_path):
    _tree(tmp_path, {"/proc/self/cgroup": "0::/pod/box\n",
                     "/sys/fs/cgroup/pod/box/cpu.max": "max 100000\n",
                     "/sys/fs/cgroup/pod/cpu.max": "1600000 100000\n",
                     "/sys/fs/cgroup/cpu.max": "max 100000\n"})
    n_aff = len(os.sched_getaffinity(0))
    budget, threads = _budget_in_child(tmp_path)
    assert budget["affinity"] == n_aff and budget["quota"] == 16.0 and budget["cpus"] == min(n_aff, 16.0)
    from helpers import expected_threads
    assert threads == expected_threads(budget)
    # eight ranks share the node: each gets an eighth, and at least one worker
    _, per_rank = _budget_in_child(tmp_path, ranks=8)
    assert per_rank == expected_threads(budget, 8)
    # a quota below the mask (a GPU box: 128 CPUs visible, 16 granted): three workers per granted CPU, never more than the mask
    if n_aff >= 4:
        _tree(tmp_path, {"/sys/fs/cgroup/pod/cpu.max": "200000 100000\n"})
        b2, t2 = _budget_in_child(tmp_path)
        assert b2["quota"] == 2.0 and t2 == min(n_aff, 6)
        # FGMM_WORKERS_PER_CPU: the multiplier (1..4; default 3) - a host where the cgroup's other processes need part of the quota
        for per_cpu, want in (("1", 2), ("2", min(n_aff, 4)), ("9", min(n_aff, 8))):
            os.environ["FGMM_WORKERS_PER_CPU"] = per_cpu
            try:
                assert _budget_in_child(tmp_path)[1] == want, per_cpu
            finally:
                del os.environ["FGMM_WORKERS_PER_CPU"]


def test_host_budget_reads_a_cgroup_v1_quota_and_the_affinity_mask(tmp_path):
    _tree(tmp_path, {"/proc/self/cgroup": "3:cpuset:/jobs\n2:cpu,cpuacct:/jobs/j1\n0::/\n",
                     "/sys/fs/cgroup/cpu,cpuacct/jobs/j1/cpu.cfs_quota_us": "350000\n",
                     "/sys/fs/cgroup/cpu,cpuacct/jobs/j1/cpu.cfs_period_us": "100000\n",
                     "/sys/fs/cgroup/cpu,cpuacct/cpu.cfs_quota_us": "-1\n",
                     "/sys/fs/cgroup/cpu,cpuacct/cpu.cfs_period_us": "100000\n"})
    budget, threads = _budget_in_child(tmp_path)
    from helpers import expected_threads
    assert budget["quota"] == 3.5 and budget["cpus"] == min(budget["affinity"], 3.5) and threads == expected_threads(budget)
    one = sorted(os.sched_getaffinity(0))[:1]
    budget, threads = _budget_in_child(tmp_path, cpus=one)  # a one-CPU mask wins over the quota
    assert budget["affinity"] == 1 and budget["cpus"] == 1.0 and threads == 1


def test_host_budget_without_a_quota_is_the_affinity_mask(tmp_path):
    _tree(tmp_path, {"/proc/self/cgroup": "0::/\n", "/sys/fs/cgroup/cpu.max": "max 100000\n"})
    budget, threads = _budget_in_child(tmp_path)
    assert budget["quota"] is None and budget["cpus"] == budget["affinity"] and threads == min(48, budget["affinity"])


# ---- checkpoints: seekable streams without touching the bitstream (fgmm_ckpt) ---------------------------------
def _ckpt_encode(L, packed, symbols, stride):
    from flashgmm_amd._lib import fgmm_ckpt
    n = len(packed)
    n_ck = L.fgmm_ckpt_count(n, stride)
    ck = (fgmm_ckpt * max(n_ck, 1))()
    out, ln = C.c_void_p(), C.c_size_t()
    packed = np.ascontiguousarray(packed, np.uint32)
    sy = np.ascontiguousarray(symbols, np.int32) if symbols is not None else None
    rc = L.fgmm_rans_encode_symtab_ckpt(packed.ctypes.data_as(C.c_void_p), sy.ctypes.data_as(C.c_void_p) if sy is not None else None, n, stride,
                                        C.byref(out), C.byref(ln), C.cast(ck, C.c_void_p))
    assert rc == 0
    data = C.string_at(out, ln.value)
    L.fgmm_free(out)
    return data, ck, n_ck


def _ckpt_decode(L, enc, hdr, pool, max_bs, bo, tl, ck, n_ck, stride):
    hdr = np.ascontiguousarray(hdr)
    pool = np.ascontiguousarray(np.asarray(pool).view(np.uint8))
    bo = np.ascontiguousarray(bo, np.uint32)
    out = np.empty(len(hdr), np.int32)
    ok = C.c_int32(-1)
    rc = L.fgmm_rans_decode_tab_ckpt(enc, len(enc), hdr.ctypes.data_as(C.c_void_p), hdr.dtype.itemsize, bo.ctypes.data_as(C.c_void_p), tl,
                                     pool.ctypes.data_as(C.c_void_p), len(pool), len(hdr), max_bs, 0, C.cast(ck, C.c_void_p), n_ck, stride,
                                     out.ctypes.data_as(C.c_void_p), C.byref(ok))
    return rc, out, ok.value


@pytest.mark.parametrize("mode", MODES)
def test_checkpoints_leave_the_bitstream_alone_and_make_it_seekable(oracle, mode):
    """The encoder notes (coder state, words read) every `stride` symbols OUT OF BAND: the bitstream is the reference's, byte
    for byte, whatever the stride; a decoder that starts every segment from its note and ends it in the next one reproduces
    the sequential decoder's symbols (bypass-coded symbols and the ragged last segment included); block-placed tables in
    shuffled order, every header form."""
    L = _lib.lib()
    y, sg, mu, pi = T.make_latent(5, M=40, h=16, w=8)
    y = y.copy()
    y.reshape(-1)[::37] *= 40  # bypass-coded symbols (several renormalisation words each) between the notes
    sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
    enc = oracle.encode_gmm(mode, sym, s, m, w)
    packed = oracle.symtab(mode, sym, s, m, w)
    max_bs = abs_max + 1
    tab = oracle.cdftab(mode, s, m, w, max_bs)
    n = len(sym)
    for stride, tl in ((256, 16), (1024, 48), (4096, 32)):
        data, ck, n_ck = _ckpt_encode(L, packed, sym, stride)
        assert data == enc and n_ck == (n - 1) // stride and n_ck >= 1
        assert all(ck[k].x >= (1 << 31) for k in range(n_ck)) and all(ck[k].pos <= ck[k + 1].pos for k in range(n_ck - 1))
        hdr, bo, pool, used = trim_full_table(tab, max_bs, form=helpers.hdr_form(max_bs), tl=tl, shuffle_seed=3)
        rc, out, ok = _ckpt_decode(L, enc, hdr, pool, max_bs, bo, tl, ck, n_ck, stride)
        assert rc == 0 and ok == 1 and np.array_equal(out, sym), (stride, tl)
    # the interleaved encoders note the same checkpoints
    from flashgmm_amd._lib import fgmm_ckpt
    assert C.sizeof(fgmm_ckpt) == 16


def test_checkpoints_are_verified_never_trusted(oracle):
    """Wrong notes - a flipped state bit, a shifted position, notes of ANOTHER stream, a wrong count - cannot change the result:
    the segment that starts from a wrong note does not end in the next one, and the stream is decoded sequentially."""
    from flashgmm_amd._lib import fgmm_ckpt

    L = _lib.lib()
    rng = np.random.default_rng(4)
    y, sg, mu, pi = T.make_latent(6, M=24, h=16, w=8)
    sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
    packed = oracle.symtab("polya", sym, s, m, w)
    max_bs = abs_max + 1
    tab = oracle.cdftab("polya", s, m, w, max_bs)
    hdr, bo, pool, used = trim_full_table(tab, max_bs, form=helpers.hdr_form(max_bs), tl=32, shuffle_seed=9)
    stride = 512
    enc, ck, n_ck = _ckpt_encode(L, packed, sym, stride)
    assert n_ck >= 4
    rc, out, ok = _ckpt_decode(L, enc, hdr, pool, max_bs, bo, 32, ck, n_ck, stride)
    assert rc == 0 and ok == 1 and np.array_equal(out, sym)
    for trial in range(12):
        bad = (fgmm_ckpt * n_ck)()
        for k in range(n_ck):
            bad[k].x, bad[k].pos = ck[k].x, ck[k].pos
        k = int(rng.integers(0, n_ck))
        if trial % 4 == 0:
            bad[k].x ^= 1 << int(rng.integers(0, 63))
        elif trial % 4 == 1:
            bad[k].pos = max(0, bad[k].pos + int(rng.choice([-1, 1, 7])))
        elif trial % 4 == 2:
            bad[k].pos = 1 << 40  # far outside the stream
        else:
            bad[k].x = int(rng.integers(1 << 31, 1 << 62))
        rc, out, ok = _ckpt_decode(L, enc, hdr, pool, max_bs, bo, 32, bad, n_ck, stride)
        assert rc == 0 and ok == 0 and np.array_equal(out, sym), trial
    # a wrong count / stride is not even tried
    rc, out, ok = _ckpt_decode(L, enc, hdr, pool, max_bs, bo, 32, ck, n_ck - 1, stride)
    assert rc == 0 and ok == 0 and np.array_equal(out, sym)
    # a truncated stream is an error with or without notes
    rc, out, ok = _ckpt_decode(L, enc[:len(enc) // 2 & ~3], hdr, pool, max_bs, bo, 32, ck, n_ck, stride)
    assert rc == 5


@pytest.mark.parametrize("stride", [0, 256])
def test_encoder_follows_a_table_that_lies_in_segments(oracle, stride):
    """fgmm_rans_encode_symtab_segs: the table in 1..4 segments (what the batched encoder's tables are when they cross PCIe tail
    first): the bitstream and the checkpoints are byte for byte those of the one-piece encoder - for segment lengths that do
    and do not divide the table, bypass-coded symbols on segment borders, a table shorter than its segments' capacity, n = 0"""
    from flashgmm_amd._lib import fgmm_ckpt

    L = _lib.lib()
    y, sg, mu, pi = T.make_latent(77, M=12, h=16, w=8, zero_frac=0.2)
    y = y.copy()
    y.reshape(-1)[::37] *= 40  # bypass-coded symbols
    sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
    packed = np.ascontiguousarray(oracle.symtab("polya", sym, s, m, w), np.uint32)
    n = len(packed)
    assert (packed >> 16 == 0).sum() > 5
    want, ck_want, n_ck = _ckpt_encode(L, packed, sym, stride) if stride else (host_encode_symtab(L, packed, sym), None, 0)
    assert want == oracle.encode_gmm("polya", sym, s, m, w)
    for n_seg, seg_len in ((1, n), (2, (n + 1) // 2), (3, n // 3 + 5), (4, (n + 3) // 4), (4, n // 2 + 1), (4, n)):
        for n_used in (n, 0) if seg_len == (n + 3) // 4 else (n,):
            parts = [np.ascontiguousarray(packed[k * seg_len:(k + 1) * seg_len]) for k in range(n_seg)]
            ptrs = (C.c_void_p * n_seg)(*[p.ctypes.data if len(p) else None for p in parts])
            ck = (fgmm_ckpt * max(L.fgmm_ckpt_count(n_used, stride), 1))()
            out, ln = C.c_void_p(), C.c_size_t()
            rc = L.fgmm_rans_encode_symtab_segs(ptrs, n_seg, seg_len, sym.ctypes.data_as(C.c_void_p), n_used, stride, C.byref(out), C.byref(ln),
                                                C.cast(ck, C.c_void_p))
            assert rc == 0, (n_seg, seg_len)
            data = C.string_at(out, ln.value)
            L.fgmm_free(out)
            if n_used == 0:
                assert data == bytes.fromhex("0000008000000000")
                continue
            assert data == want, (n_seg, seg_len)
            if stride:
                assert all(ck[k].x == ck_want[k].x and ck[k].pos == ck_want[k].pos for k in range(n_ck))
    ptrs = (C.c_void_p * 2)(packed.ctypes.data, None)
    out, ln = C.c_void_p(), C.c_size_t()
    assert L.fgmm_rans_encode_symtab_segs(ptrs, 2, n // 2, None, n, 0, C.byref(out), C.byref(ln), None) == 1  # a segment is missing
    assert L.fgmm_rans_encode_symtab_segs(ptrs, 5, n, None, n, 0, C.byref(out), C.byref(ln), None) == 1

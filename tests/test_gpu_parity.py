"""GPU (-m gpu): parity of the HIP path with the oracle / golden vectors, called through the C ABI.

Bars: bit-exact for every integer / byte / index result; float CDFs within 1e-5 of the reference (north_star) —
and, since the arithmetic is restated op for op, asserted bit-exact as well.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from flashgmm_amd import CheckpointedBytes, GaussianMixtureConditional, _lib, ans
from tests import synth as T
from helpers import expand_trimmed, hdr_form

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MODES = ["polya", "as", "logistic"]
DEV = "cuda:0"


def _mg():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def dv(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def gpu_cdf(mode, v, s, m, w):
    L, ctx = _lib.lib(), _lib.ctx(0)
    v, s, m, w = dv(v.astype(np.int32)), dv(s), dv(m), dv(w)
    n = v.numel()
    c1 = torch.empty(n, dtype=torch.float32, device=DEV)
    c2 = torch.empty_like(c1)
    torch.cuda.synchronize()
    _lib.check(L.fgmm_gmm_cdf_hip(ctx, None, v.data_ptr(), s.data_ptr(), m.data_ptr(), w.data_ptr(), n, s.stride(0),
                                  s.stride(1), _lib.mode_id(mode), c1.data_ptr(), c2.data_ptr()))
    return c1.cpu().numpy(), c2.cpu().numpy()


def gpu_symtab(mode, v, s, m, w):
    L, ctx = _lib.lib(), _lib.ctx(0)
    v, s, m, w = dv(v.astype(np.int32)), dv(s), dv(m), dv(w)
    n = v.numel()
    out = torch.empty(n, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    _lib.check(L.fgmm_build_symtab_hip(ctx, None, v.data_ptr(), s.data_ptr(), m.data_ptr(), w.data_ptr(), n, s.stride(0),
                                       s.stride(1), _lib.mode_id(mode), out.data_ptr()))
    return out.cpu().numpy().view(np.uint32)


def gpu_cdftab(mode, s, m, w, max_bs, flags=0, cap=None):
    """generic two-pass kernels: 4-byte headers, rows sequential -> (hdr, pool, used)"""
    L, ctx = _lib.lib(), _lib.ctx(0)
    s, m, w = dv(s), dv(m), dv(w)
    n = s.size(0)
    cap = n * 2 * (2 * max_bs + 6) if cap is None else cap  # bytes
    hdr = torch.zeros(n, dtype=torch.int32, device=DEV)
    pool = torch.zeros(cap + 128, dtype=torch.uint8, device=DEV)
    used = torch.zeros(2, dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()
    _lib.check(L.fgmm_build_cdftab_hip(ctx, None, s.data_ptr(), m.data_ptr(), w.data_ptr(), n, s.stride(0), s.stride(1),
                                       _lib.mode_id(mode), max_bs, flags, hdr.data_ptr(), pool.data_ptr(), cap,
                                       used.data_ptr()))
    return hdr.cpu().numpy().view(np.uint32), pool.cpu().numpy(), int(used[0].item())


def gpu_tab(mode, s, m, w, max_bs, flags=0, cap=None):
    """the single-pass kernel of the batched decode path: headers in the form of max_bs, rows placed block by block
    -> (hdr, blk_off, rows, used, tl)"""
    L, ctx = _lib.lib(), _lib.ctx(0)
    s, m, w = dv(s), dv(m), dv(w)
    n = s.size(0)
    form = hdr_form(max_bs)
    cap = n * (2 * (2 * max_bs + 2) + 4) if cap is None else cap
    hdr = torch.zeros(max(n, 1) * form, dtype=torch.uint8, device=DEV)
    blk_off = torch.zeros(n // 16 + 2, dtype=torch.int32, device=DEV)
    rows = torch.zeros(cap + 128, dtype=torch.uint8, device=DEV)
    used = torch.zeros(2, dtype=torch.int64, device=DEV)
    tl = C.c_int32(0)
    torch.cuda.synchronize()
    _lib.check(L.fgmm_build_tab_hip(ctx, None, s.data_ptr(), m.data_ptr(), w.data_ptr(), n, s.stride(0), s.stride(1),
                                    _lib.mode_id(mode), max_bs, flags, hdr.data_ptr(), blk_off.data_ptr(), rows.data_ptr(), cap,
                                    used.data_ptr(), C.byref(tl)))
    dt = {2: np.uint16, 4: np.uint32, 8: np.uint64}[form]
    nblk = (n + tl.value - 1) // tl.value
    return (hdr.cpu().numpy()[: n * form].view(dt), blk_off.cpu().numpy().view(np.uint32)[:nblk], rows.cpu().numpy(),
            int(used[0].item()), tl.value)


def gpu_full_table(kernel, mode, s, m, w, max_bs, flags=0):
    """-> (the virtual full table [n, 2*max_bs+2] a kernel's output stands for, bytes of rows)"""
    if kernel == "generic":
        h, p_, u = gpu_cdftab(mode, s, m, w, max_bs, flags)
        return expand_trimmed(h, p_, max_bs), u
    h, bo, rows, u, tl = gpu_tab(mode, s, m, w, max_bs, flags)
    return expand_trimmed(h, rows, max_bs, bo, tl), u


KERNELS = ["generic", "tab"]


# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", MODES)
def test_g1_float_cdf(mode):
    g = np.load(os.path.join(GOLD, "g1_cdf.npz"))
    c1, c2 = gpu_cdf(mode, g["v"], g["scales"], g["means"], g["weights"])
    r1, r2 = g[f"c1_{mode}"].view(np.float32), g[f"c2_{mode}"].view(np.float32)
    assert np.abs(c1 - r1).max() <= 1e-5 and np.abs(c2 - r2).max() <= 1e-5  # the stated tolerance
    assert np.array_equal(c1.view(np.uint32), g[f"c1_{mode}"])  # and in fact bit for bit
    assert np.array_equal(c2.view(np.uint32), g[f"c2_{mode}"])


@pytest.mark.parametrize("mode", MODES)
def test_float_cdf_bit_exact_wide_sweep(oracle, mode):
    """2M rows incl. extreme sigma (1e-5 .. 3e3), far tails, both sides of every clamp."""
    rng = np.random.default_rng(2024)
    n = 2_000_000
    e = np.exp(rng.uniform(-3, 2.5, n)).astype(np.float32)
    mu = (rng.standard_normal((n, 4)) * e[:, None]).astype(np.float32)
    sg = np.exp(rng.uniform(-12, 8, (n, 4))).astype(np.float32)
    sg[: n // 2] = np.clip((rng.uniform(0, 2, (n // 2, 4)) + 0.05) * e[: n // 2, None], 0.11, 256)
    lg = rng.standard_normal((n, 4))
    pi = (np.exp(lg) / np.exp(lg).sum(1, keepdims=True)).astype(np.float32)
    v = np.round(rng.standard_normal(n) * 40).astype(np.int32)
    c1, c2 = gpu_cdf(mode, v, sg, mu, pi)
    o1, o2 = oracle.gmm_cdf(mode, v, sg, mu, pi)
    assert np.array_equal(c1.view(np.uint32), o1.view(np.uint32))
    assert np.array_equal(c2.view(np.uint32), o2.view(np.uint32))


@pytest.mark.parametrize("mode", MODES)
def test_g2_symtab(oracle, mode):
    g = np.load(os.path.join(GOLD, "g1_cdf.npz"))
    packed = gpu_symtab(mode, g["v"], g["scales"], g["means"], g["weights"])
    assert np.array_equal(packed >> 16, g[f"range_{mode}"].astype(np.uint32))
    nb = (packed >> 16) != 0
    assert np.array_equal((packed & 0xFFFF)[nb], g[f"start_{mode}"].astype(np.uint32)[nb])
    assert np.array_equal(packed, oracle.symtab(mode, g["v"], g["scales"], g["means"], g["weights"]))


@pytest.mark.parametrize("mode", MODES)
def test_symtab_strided_and_ragged(oracle, mode):
    """(n,4) row-major, (1,n) planar views, n not a multiple of 4 -> scalar and 16-B kernels agree with the oracle."""
    for seed, n in ((1, 1), (2, 3), (3, 1021), (4, 4096)):
        rng = np.random.default_rng(seed)
        e = np.exp(rng.uniform(-2, 2, n)).astype(np.float32)
        mu = (rng.standard_normal((n, 4)) * e[:, None]).astype(np.float32)
        sg = ((rng.uniform(0, 2, (n, 4)) + 0.11) * e[:, None]).astype(np.float32)
        pi = rng.dirichlet(np.ones(4), n).astype(np.float32)
        v = np.round(rng.standard_normal(n) * 2 * e).astype(np.int32)
        want = oracle.symtab(mode, v, sg, mu, pi)
        assert np.array_equal(gpu_symtab(mode, v, sg, mu, pi), want)
        # planar: strides (1, n)
        L, ctx = _lib.lib(), _lib.ctx(0)
        sp, mp, wp = (dv(np.ascontiguousarray(a.T)) for a in (sg, mu, pi))
        vd = dv(v)
        out = torch.empty(n, dtype=torch.int32, device=DEV)
        torch.cuda.synchronize()
        _lib.check(L.fgmm_build_symtab_hip(ctx, None, vd.data_ptr(), sp.data_ptr(), mp.data_ptr(), wp.data_ptr(), n, 1, n,
                                           _lib.mode_id(mode), out.data_ptr()))
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want)


@pytest.mark.parametrize("mode", MODES)
def test_g3_small_streams_through_ans_mirror(mode):
    gold = json.load(open(os.path.join(GOLD, "g3_small.json")))["cases"]
    for name, (sym, s, m, w) in _mg().g3_cases().items():
        ent = gold[name][mode]
        for dev in ("cpu", DEV):  # the reference hands CPU tensors; GPU tensors are this repo's fast path
            t = [torch.from_numpy(a).to(dev) for a in (sym, s, m, w)]
            b = ans.RansEncoder().encode_with_indexes_gmm(*t, 0, mode=mode)
            assert b.hex() == ent["hex"], (name, dev)
            d = ans.RansDecoder().decode_with_indexes_gmm(b, *t[1:], ent["max_bs"], mode=mode)
            assert d.dtype == torch.int32 and d.device.type == "cpu"
            assert d.tolist() == ent["decoded"], (name, dev)


@pytest.mark.parametrize("mode", MODES)
def test_ka1_boundary_level(mode):
    """KA-1 through the reference's own boundary signature, with the reference's (n,4) strides-(1,n) views."""
    ent = json.load(open(os.path.join(GOLD, "ka1.json")))[mode]
    for seed in ("1234", "0"):
        y, sg, mu, pi = T.make_latent(int(seed))
        sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
        t = [torch.from_numpy(np.ascontiguousarray(a.T)).to(DEV).T for a in (s, m, w)]  # strides (1, n)
        assert t[0].stride() == (1, len(sym))
        b = ans.RansEncoder().encode_with_indexes_gmm(torch.from_numpy(sym).to(DEV), *t, abs_max + 1, mode=mode)
        assert (len(b), hashlib.md5(b).hexdigest()) == (ent[seed]["len"], ent[seed]["md5"])
        d = ans.RansDecoder().decode_with_indexes_gmm(b, *t, abs_max + 1, mode=mode)
        assert np.array_equal(d.numpy(), sym)


@pytest.mark.parametrize("mode", MODES)
def test_g4_api_level(mode):
    """GaussianMixtureConditional.compress / decompress == the reference's own class (un-clamped sigma, zero channels)."""
    gold = json.load(open(os.path.join(GOLD, "g4_api.json")))[mode]
    gmc = GaussianMixtureConditional(K=4, mode=mode)
    for seed, ent in gold.items():
        y, sg, mu, pi = T.make_latent(int(seed), M=ent["M"], h=ent["h"], w=ent["w"], clamp=False, zero_frac=0.15)
        t = [dv(a) for a in (y, sg, mu, pi)]
        (b, abs_max, zb), yq = gmc.compress(*t)
        assert isinstance(b, bytes) and isinstance(abs_max, int)
        assert zb.dtype == torch.int64 and zb.device == t[0].device and zb.tolist() == ent["zero_bitmap"]
        assert (len(b), hashlib.md5(b).hexdigest(), abs_max) == (ent["len"], ent["md5"], ent["abs_max"])
        assert yq.shape == t[0].shape and hashlib.sha256(yq.cpu().numpy().tobytes()).hexdigest() == ent["yq_sha256"]
        y_hat = gmc.decompress(b, abs_max, zb, *t[1:])
        assert y_hat.dtype == torch.float32 and list(y_hat.shape) == ent["y_hat_shape"]
        assert torch.equal(y_hat, yq)


@pytest.mark.parametrize("mode", MODES)
def test_chunked_parameter_views(oracle, mode):
    """params handed as chunk(3, 1) views of one [1, 3*K*M, h, w] head output (latent codec :193-195)."""
    M, h, w = 24, 12, 10  # hw = 120: 16-B path;  M*hw strides stay multiples of 4
    y, sg, mu, pi = T.make_latent(8, M=M, h=h, w=w, clamp=False)
    head = dv(np.concatenate([sg, mu, pi], axis=1))
    s_, m_, w_ = head.chunk(3, 1)
    gmc = GaussianMixtureConditional(K=4, mode=mode)
    (b, abs_max, zb), yq = gmc.compress(dv(y), s_, m_, w_)
    sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(y, sg, mu, pi)
    assert b == oracle.encode_gmm(mode, sym, s, m, wt) and abs_max == am
    assert torch.equal(gmc.decompress(b, abs_max, zb, s_, m_, w_), yq)


@pytest.mark.parametrize("mode", MODES)
def test_ragged_shapes_and_edge_cases(oracle, mode):
    gmc = GaussianMixtureConditional(K=4, mode=mode)
    for seed, (M, h, w), zf in ((1, (3, 5, 7), 0.0), (2, (17, 1, 1), 0.3), (3, (192, 16, 8), 0.5), (4, (5, 3, 3), 1.0)):
        y, sg, mu, pi = T.make_latent(seed, M=M, h=h, w=w, clamp=False, zero_frac=zf)
        t = [dv(a) for a in (y, sg, mu, pi)]
        (b, abs_max, zb), yq = gmc.compress(*t)
        sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(y, sg, mu, pi)
        assert b == oracle.encode_gmm(mode, sym, s, m, wt), (seed, M, h, w)
        assert abs_max == am and zb.cpu().numpy().tolist() == zbm.tolist()
        assert np.array_equal(yq.cpu().numpy(), yqn)
        if zf == 1.0:
            assert b == bytes.fromhex("0000008000000000")  # every channel zero: empty stream
        assert torch.equal(gmc.decompress(b, abs_max, zb, *t[1:]), yq)


def test_wide_bypass_symbols(oracle):
    """latents far outside int16 with tiny sigma: bypass nibbles must carry the full int32 (rans_interface.cpp:525)."""
    M, h, w = 4, 4, 4
    y, sg, mu, pi = T.make_latent(21, M=M, h=h, w=w)
    y = y.copy()
    y[0, 1, 2, 3] = 70000.3
    y[0, 2, 0, 0] = -123456.7
    y[0, 3, 1, 1] = 40000.0
    gmc = GaussianMixtureConditional(K=4, mode="polya")
    t = [dv(a) for a in (y, sg, mu, pi)]
    (b, abs_max, zb), yq = gmc.compress(*t)
    sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(y, sg, mu, pi)
    assert abs_max == am == 123457
    assert b == oracle.encode_gmm("polya", sym, s, m, wt)
    # decoder half-width 123458: 8-byte headers, generic kernels; the result is the reference's
    assert np.array_equal(oracle.decode_gmm("polya", b, s, m, wt, abs_max + 1), sym)
    y_hat = gmc.decompress(b, abs_max, zb, *t[1:])
    assert torch.equal(y_hat, yq) and np.array_equal(y_hat.cpu().numpy(), yqn)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mode", MODES)
def test_cdftab_equals_oracle_full_table(oracle, mode, kernel):
    y, sg, mu, pi = T.make_latent(31, M=24, h=16, w=8)
    sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
    max_bs = abs_max + 1
    full = oracle.cdftab(mode, s, m, w, max_bs)
    got, used = gpu_full_table(kernel, mode, s, m, w, max_bs)
    assert np.array_equal(got, full)
    assert used < 0.6 * 2 * full.size  # trimmed for real (bytes)
    # un-normalised / wild parameters: still exact (the window is found by evaluation, not by assumption)
    rng = np.random.default_rng(3)
    n = 3000
    sgw = np.exp(rng.uniform(-6, 6, (n, 4))).astype(np.float32)
    muw = (rng.standard_normal((n, 4)) * 20).astype(np.float32)
    piw = rng.uniform(0, 0.6, (n, 4)).astype(np.float32)
    for max_bs in (37, 200):  # 2-byte and 4-byte headers
        got, _ = gpu_full_table(kernel, mode, sgw, muw, piw, max_bs)
        assert np.array_equal(got, oracle.cdftab(mode, sgw, muw, piw, max_bs)), max_bs


@pytest.mark.parametrize("mode", MODES)
def test_single_pass_kernel_sizes_and_edge_cases(oracle, mode):
    """ragged block counts (n not a multiple of the block), a single latent, every block size the LDS budget gives,
    non-monotone rows through the 2-byte header's escape; the two kernels agree byte for byte on what they write"""
    rng = np.random.default_rng(41)
    for n, max_bs in ((1, 5), (15, 5), (17, 60), (1000, 126), (257, 127), (300, 255), (4097, 20), (0, 9)):
        sg = np.exp(rng.uniform(-3, 3, (n, 4))).astype(np.float32)
        mu = (rng.standard_normal((n, 4)) * np.exp(rng.uniform(-2, 4, (n, 1)))).astype(np.float32)
        pi = rng.dirichlet(np.ones(4), n).astype(np.float32)
        pi[::5] = rng.uniform(-0.4, 1.2, (len(pi[::5]), 4)).astype(np.float32)  # negative weights: non-monotone rows
        want = oracle.cdftab(mode, sg, mu, pi, max_bs)
        for flags in (0, 2):
            sgc = np.clip(sg, np.float32(0.11), np.float32(256)) if flags & 2 else sg
            want = oracle.cdftab(mode, sgc, mu, pi, max_bs)
            h, bo, rows, used, tl = gpu_tab(mode, sg, mu, pi, max_bs, flags)
            assert np.array_equal(expand_trimmed(h, rows, max_bs, bo, tl), want), (n, max_bs, flags)
            import helpers
            saved = helpers.EF_MIN
            try:  # FGMM_TAB_RAW_ROWS: every row as uint16 entries
                helpers.EF_MIN = 1 << 30
                hr, bor, rr, ur, tlr = gpu_tab(mode, sg, mu, pi, max_bs, flags | 4)
                assert np.array_equal(expand_trimmed(hr, rr, max_bs, bor, tlr), want) and ur >= used
                hq, pq, uq = gpu_cdftab(mode, sg, mu, pi, max_bs, flags | 4)
                assert np.array_equal(expand_trimmed(hq, pq, max_bs), want)
            finally:
                helpers.EF_MIN = saved
            hg, pg, ug = gpu_cdftab(mode, sg, mu, pi, max_bs, flags)
            # same rows; the single-pass kernel adds the escaped rows' headers and pads every block to 4 bytes
            esc = 4 * int(((h >> 8) == 255).sum() if h.dtype == np.uint16 else 0)
            assert 0 <= used - ug - esc <= 2 * len(bo) and (used - ug) % 2 == 0, (n, max_bs, flags)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):  # 2*max_bs+2 beyond the kernel's LDS budget: the generic path's
        gpu_tab(mode, sg[:1], mu[:1], pi[:1], 5000)


def test_row_area_overflow_is_reported_with_the_size_needed(oracle):
    """both building blocks: a row area that is too small -> FGMM_ERR_NOMEM, *used = the bytes needed, and a second call
    with exactly that size succeeds (the batched decoder does the same with its staging area)"""
    y, sg, mu, pi = T.make_latent(33, M=8, h=16, w=8)
    sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
    max_bs = abs_max + 1
    want = oracle.cdftab("polya", s, m, w, max_bs)
    h, p_, need = gpu_cdftab("polya", s, m, w, max_bs)
    with pytest.raises(RuntimeError, match=f"NOMEM.*need {need}"):
        gpu_cdftab("polya", s, m, w, max_bs, cap=need - 4)
    h, p_, u = gpu_cdftab("polya", s, m, w, max_bs, cap=need)
    assert u == need and np.array_equal(expand_trimmed(h, p_, max_bs), want)
    h, bo, rows, need_t, tl = gpu_tab("polya", s, m, w, max_bs)
    with pytest.raises(RuntimeError, match=f"NOMEM.*need {need_t}"):
        gpu_tab("polya", s, m, w, max_bs, cap=need_t - 4)
    h, bo, rows, u, tl = gpu_tab("polya", s, m, w, max_bs, cap=need_t)
    assert u == need_t and np.array_equal(expand_trimmed(h, rows, max_bs, bo, tl), want)


@pytest.mark.parametrize("mode", MODES)
def test_saturation_lemmas_exhaustive(mode):
    """every binary32 beyond the pruning thresholds saturates as fgmm_math.h claims (2e9 values per mode)"""
    bad = C.c_uint64(123)
    _lib.check(_lib.lib().fgmm_selftest_saturation(_lib.ctx(0), _lib.mode_id(mode), C.byref(bad)))
    assert bad.value == 0


def test_fastmath_cores_equal_ieee():
    """the hand-expanded division / reciprocal / sqrt cores give the compiler's correctly-rounded bits:
    sqrt and reciprocal exhaustively over their domains, division on 4e9 hashed pairs; the kernels' slimmed Phi
    (no upper exp clamp, ldexp, fused 0.5*(1+s)) equals the literal one for every binary32 |z| < 2^48, all modes"""
    L, ctx = _lib.lib(), _lib.ctx(0)
    for which, n in ((0, 0), (2, 0), (1, 1 << 32), (3, 0), (4, 0), (5, 0), (6, 0)):
        for seed in ((1, 2) if which == 1 else (0,)):
            bad = C.c_uint64(99)
            _lib.check(L.fgmm_selftest_fastmath(ctx, which, n, seed, C.byref(bad)))
            assert bad.value == 0, (which, seed, bad.value)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mode", MODES)
def test_cdftab_clamped_variant(oracle, mode, kernel):
    """the entropy-model kernel variant (sigma clamp + shared refined reciprocals) == oracle on clamped sigma,
    including NaN / inf / huge means and sigma far outside the clamp"""
    rng = np.random.default_rng(123)
    n = 6000
    sg = np.exp(rng.uniform(-6, 8, (n, 4))).astype(np.float32)
    mu = (rng.standard_normal((n, 4)) * np.exp(rng.uniform(-2, 5, (n, 1)))).astype(np.float32)
    pi = rng.dirichlet(np.ones(4), n).astype(np.float32)
    mu[3::97, 1] = np.inf
    mu[4::97, 2] = -np.inf
    mu[5::97, 0] = 3e38
    mu[6::97, 3] = np.nan
    sg[7::97, 0] = np.nan
    sg[8::97, 1] = 0.0
    sg[9::97, 2] = -5.0
    sg[10::97, 3] = np.inf
    sgc = np.minimum(np.maximum(sg, np.float32(0.11)), np.float32(256))  # torch.clamp semantics (NaN stays)
    sgc[np.isnan(sg)] = np.nan
    for max_bs in (3, 60):
        want = oracle.cdftab(mode, sgc, mu, pi, max_bs)
        for flags in (2, 3):
            got, _ = gpu_full_table(kernel, mode, sg, mu, pi, max_bs, flags=flags)
            assert np.array_equal(got, want), (max_bs, flags)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mode", MODES)
def test_cdftab_pruned_equals_unpruned(oracle, mode, kernel):
    """wide half-widths, tiny and huge sigma, far-off means, weights outside [0,1], NaN/inf/zero sigma: the pruned
    kernel must reproduce the full evaluation exactly (and both the oracle where the oracle is affordable)."""
    rng = np.random.default_rng(77)
    n = 20000
    sg = np.exp(rng.uniform(-4, 6, (n, 4))).astype(np.float32)
    mu = (rng.standard_normal((n, 4)) * np.exp(rng.uniform(-2, 7, (n, 1)))).astype(np.float32)
    pi = rng.dirichlet(np.ones(4), n).astype(np.float32)
    pi[::7] = rng.uniform(-0.5, 1.5, (len(pi[::7]), 4)).astype(np.float32)
    sg[5::101, 1] = 0.0
    sg[6::101, 2] = np.inf
    sg[7::101, 0] = np.nan
    sg[8::101, 3] = -1.0
    mu[9::101, 0] = np.inf
    pi[10::101, 2] = np.nan
    for max_bs in (0, 1, 40, 200) if kernel == "tab" else (0, 1, 40, 700):
        f0, u0 = gpu_full_table(kernel, mode, sg, mu, pi, max_bs, flags=1)
        f1, u1 = gpu_full_table(kernel, mode, sg, mu, pi, max_bs, flags=0)
        assert np.array_equal(f0, f1) and u0 == u1, max_bs
        if kernel == "generic":  # sequential rows: the two runs are the same bytes
            h0, p0, _ = gpu_cdftab(mode, sg, mu, pi, max_bs, flags=1)
            h1, p1, _ = gpu_cdftab(mode, sg, mu, pi, max_bs, flags=0)
            assert np.array_equal(h0, h1) and np.array_equal(p0[:u0], p1[:u1])
        if max_bs <= 40:
            assert np.array_equal(f1, oracle.cdftab(mode, sg, mu, pi, max_bs)), max_bs


@pytest.mark.parametrize("mode", MODES)
def test_decoder_equals_reference_on_garbage_streams(oracle, mode):
    """Desynchronised input: the result must still be the reference's (its bisection fallbacks included)."""
    rng = np.random.default_rng(17)
    n, max_bs = 5000, 12
    e = np.exp(rng.uniform(-2, 1.5, n)).astype(np.float32)
    mu = (rng.standard_normal((n, 4)) * e[:, None]).astype(np.float32)
    sg = ((rng.uniform(0, 2, (n, 4)) + 0.11) * e[:, None]).astype(np.float32)
    pi = rng.dirichlet(np.ones(4), n).astype(np.float32)
    enc = rng.integers(0, 256, 4 * (n + 64), dtype=np.uint8).tobytes()
    want = oracle.decode_gmm(mode, enc, sg, mu, pi, max_bs)
    got = ans.RansDecoder().decode_with_indexes_gmm(enc, dv(sg), dv(mu), dv(pi), max_bs, mode=mode)
    assert np.array_equal(got.numpy(), want)
    with pytest.raises(RuntimeError, match="STREAM"):
        ans.RansDecoder().decode_with_indexes_gmm(enc[:64], dv(sg), dv(mu), dv(pi), max_bs, mode=mode)


@pytest.mark.parametrize("mode", MODES)
def test_header_forms_of_the_batched_decoder(oracle, mode):
    """the batched decoder ships 2-byte headers for an item whose half-width fits and that has no non-monotone row, the
    4-byte form otherwise: a small-width item, the same with negative sigmas (decreasing CDFs: non-monotone rows, the
    raw coder boundary does not clamp) and a wide one, all on a garbage stream, must equal the reference's decoder"""
    rng = np.random.default_rng(23)
    n = 3000
    e = np.exp(rng.uniform(-2, 1.0, n)).astype(np.float32)
    mu = (rng.standard_normal((n, 4)) * e[:, None]).astype(np.float32)
    sg = ((rng.uniform(0, 2, (n, 4)) + 0.11) * e[:, None]).astype(np.float32)
    pi = rng.dirichlet(np.ones(4), n).astype(np.float32)
    sg_neg = sg.copy()
    sg_neg[rng.random(n) < 0.2, 1] *= -1.0
    enc = rng.integers(0, 256, 4 * (n + 64), dtype=np.uint8).tobytes()
    for sgx, max_bs in ((sg, 20), (sg_neg, 20), (sg, 126), (sg, 127), (sg_neg, 400)):
        want = oracle.decode_gmm(mode, enc, sgx, mu, pi, max_bs)
        got = ans.RansDecoder().decode_with_indexes_gmm(enc, dv(sgx), dv(mu), dv(pi), max_bs, mode=mode)
        assert np.array_equal(got.numpy(), want), (max_bs, bool((sgx < 0).any()))


def test_input_validation_is_loud():
    s = torch.rand(8, 4, device=DEV) + 0.2
    v = torch.zeros(8, dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError):
        ans.RansEncoder().encode_with_indexes_gmm(v.long(), s, s, s, 1)
    with pytest.raises(RuntimeError):
        ans.RansEncoder().encode_with_indexes_gmm(v, s[:, :3], s, s, 1)
    with pytest.raises(RuntimeError):
        ans.RansEncoder().encode_with_indexes_gmm(v, s.double(), s, s, 1)
    with pytest.raises(RuntimeError):
        GaussianMixtureConditional(K=3).compress(torch.zeros(1, 2, 2, 2, device=DEV), *(torch.ones(1, 6, 2, 2, device=DEV),) * 3)


def test_buffered_encoder_concatenates(oracle):
    rng = np.random.default_rng(9)
    parts = []
    for n in (100, 37):
        e = np.ones(n, np.float32)
        mu = rng.standard_normal((n, 4)).astype(np.float32)
        sg = (rng.uniform(0.2, 2, (n, 4))).astype(np.float32)
        pi = rng.dirichlet(np.ones(4), n).astype(np.float32)
        v = np.round(rng.standard_normal(n) * 2).astype(np.int32)
        parts.append((v, sg, mu, pi))
    enc = ans.BufferedRansEncoder()
    for v, sg, mu, pi in parts:
        enc.encode_with_indexes_gmm(dv(v), dv(sg), dv(mu), dv(pi), 0, mode="as")
    b = enc.flush()
    cat = [np.concatenate([p[i] for p in parts]) for i in range(4)]
    assert b == oracle.encode_gmm("as", *cat)
    assert enc.flush() == bytes.fromhex("0000008000000000")


@pytest.mark.parametrize("mode", MODES)
def test_batch_of_kodak_halves_full_size(mode):
    """BASELINE configs[1] shape: 24 images x 2 halves of [1,192,32,24], one native call each way.
    EVERY one of the 48 bitstreams against the reference's own bytes (tests/golden/fullsize.json: length + md5 of what the compiled
    reference's RansEncoder returns for these seeds, rans_interface.cpp:609-617), and the whole batch through
    decode(encode(y)) == round(y)."""
    ka = json.load(open(os.path.join(GOLD, "fullsize.json")))[mode]["kodak24"]
    gmc = GaussianMixtureConditional(K=4, mode=mode)
    ys, ss, ms, ws = [], [], [], []
    for seed in range(48):
        y, sg, mu, pi = T.make_latent(seed)
        ys.append(dv(y)); ss.append(dv(sg)); ms.append(dv(mu)); ws.append(dv(pi))
    res = gmc.compress_batch(ys, ss, ms, ws)
    for seed in range(48):
        (b, abs_max, zb), yq = res[seed]
        assert (len(b), hashlib.md5(b).hexdigest(), abs_max, int(zb.sum())) == (ka[str(seed)]["len"], ka[str(seed)]["md5"], ka[str(seed)]["abs_max"],
                                                                                ka[str(seed)]["nz_channels"]), seed
    outs = gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
    for seed in range(48):
        assert torch.equal(outs[seed], res[seed][1]), seed
        assert torch.equal(res[seed][1], torch.round(ys[seed]))


@pytest.mark.parametrize("mode", MODES)
def test_one_kodak_half_alone_with_default_options(mode):
    """The latency case: ONE bitstream per call with the context's default options - the automatic piece plan gives a lone decoder
    two pieces, an eighth first (fgmm_decode.cpp plan_pieces / piece_bound, `lead_small`), a branch no batch reaches.  Two halves
    (seeds 0 and 1), each compressed and decompressed alone, against the reference's bytes and round(y)."""
    ka = json.load(open(os.path.join(GOLD, "fullsize.json")))[mode]["kodak24"]
    gmc = GaussianMixtureConditional(K=4, mode=mode)
    assert _lib.get_option(0, "pieces") == 0
    for seed in (0, 1):
        t = [dv(a) for a in T.make_latent(seed)]
        (b, abs_max, zb), yq = gmc.compress(*t)
        assert (len(b), hashlib.md5(b).hexdigest(), abs_max) == (ka[str(seed)]["len"], ka[str(seed)]["md5"], ka[str(seed)]["abs_max"]), seed
        for _ in range(3):  # (the same buffers reused call after call)
            y_hat = gmc.decompress(b, abs_max, zb, *t[1:])
            assert torch.equal(y_hat, yq) and torch.equal(yq, torch.round(t[0])), seed
        log = _lib.call_log(0, 1)[0]
        assert log["kind"] == "decode" and log["count"] == 1


@pytest.mark.parametrize("stride", [0, 512])
def test_compiled_and_ctypes_bindings_agree(monkeypatch, stride):
    """flashgmm_amd._native (pybind11, csrc/fgmm_pybind.cpp: the mirror of the reference's module definition, rans_interface.cpp:961-1036)
    and the ctypes binding (flashgmm_amd/_lib.py, the documented fallback) drive the same C ABI: the same bytes, side information,
    checkpoints and decoded latents from either - stacked inputs, strided stage views, plain and checkpointed streams."""
    assert _lib.native() is not None, "flashgmm_amd/_native*.so is missing: flashgmm_amd/csrc/build.sh builds it"
    gmc = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=stride)
    lat = [T.make_latent(4200 + i, M=48, h=16, w=12, clamp=False, zero_frac=0.2) for i in range(6)]
    ys, ss, ms, ws = (torch.stack([torch.from_numpy(l[k][0]) for l in lat]).to("cuda:0") for k in range(4))
    nat = gmc.compress_batch(ys, ss, ms, ws)
    out_nat = [gmc.decompress_batch(nat.strings[s::2], nat.abs_maxes[s::2], nat.zero_bitmaps[s::2], ss[s::2], ms[s::2], ws[s::2], stacked_output=True) for s in range(2)]
    # the sequence form (items of any mix of shapes: compress_items / decompress_items of the module)
    mixed = [T.make_latent(4250 + i, M=(24, 40, 8)[i % 3], h=(8, 16, 4)[i % 3], w=(12, 8, 20)[i % 3], clamp=False, zero_frac=0.2 if i != 4 else 1.0) for i in range(7)]
    mixed = [[torch.from_numpy(l[k]).to("cuda:0") for k in range(4)] for l in mixed]
    cols = [[t[k] for t in mixed] for k in range(4)]
    nat_l = gmc.compress_batch(*cols)
    out_nat_l = gmc.decompress_batch([r[0][0] for r in nat_l], [r[0][1] for r in nat_l], [r[0][2] for r in nat_l], *cols[1:])
    monkeypatch.setattr(_lib, "_native", False)
    assert _lib.native() is None
    cty_l = gmc.compress_batch(*cols)
    out_cty_l = gmc.decompress_batch([r[0][0] for r in nat_l], [r[0][1] for r in nat_l], [r[0][2] for r in nat_l], *cols[1:])
    for i in range(7):
        (bn, an, zn), yn = nat_l[i]
        (bc, ac, zc), yc = cty_l[i]
        assert type(bn) is type(bc) is (CheckpointedBytes if stride else bytes) and bytes(bn) == bytes(bc) and an == ac and torch.equal(zn, zc) and torch.equal(yn, yc)
        assert yn.shape == mixed[i][0].shape and torch.equal(yn, torch.round(mixed[i][0])) and torch.equal(out_nat_l[i], yn) and torch.equal(out_cty_l[i], yn)
        if stride:
            assert bn.ckpt_stride == stride and np.array_equal(bn.ckpt, bc.ckpt)
    cty = gmc.compress_batch(ys, ss, ms, ws)
    out_cty = [gmc.decompress_batch(nat.strings[s::2], nat.abs_maxes[s::2], nat.zero_bitmaps[s::2], ss[s::2], ms[s::2], ws[s::2], stacked_output=True) for s in range(2)]
    assert type(nat.strings[0]) is (CheckpointedBytes if stride else bytes) and type(cty.strings[0]) is type(nat.strings[0])
    for i in range(6):
        assert bytes(nat.strings[i]) == bytes(cty.strings[i]) and nat.abs_maxes[i] == cty.abs_maxes[i]
        if stride:
            assert nat.strings[i].ckpt_stride == stride and np.array_equal(nat.strings[i].ckpt, cty.strings[i].ckpt) and len(nat.strings[i].ckpt) == (int(nat.zero_bitmaps[i].sum()) * 192 - 1) // stride
    assert torch.equal(nat.zero_bitmaps, cty.zero_bitmaps) and torch.equal(nat.y_q, cty.y_q) and torch.equal(nat.y_q[:, 0], torch.round(ys))
    for s in range(2):
        assert torch.equal(out_nat[s], nat.y_q[s::2]) and torch.equal(out_cty[s], nat.y_q[s::2])


def test_sink_puts_the_bitstreams_into_the_callers_storage(monkeypatch):
    """include/flashgmm_amd.h: fgmm_sink.  fgmm_gmc_compress_batch_to asks the sink once per item, on the calling thread, for
    storage of the bitstream's exact size and the worker that coded it writes it there (what rans_interface.cpp:557-585 does with its py::bytes): the same
    bytes as the plain call's buffers; a sink that refuses an item fails the call with FGMM_ERR_NOMEM and returns no buffer.  The
    compiled module's `bytes` objects are made that way: they hash, compare and slice like any other."""
    import ctypes as C
    import threading

    L = _lib.lib()
    gmc = GaussianMixtureConditional(K=4, mode="polya")
    lat = [T.make_latent(4300 + i, M=32 + 8 * (i % 2), h=16, w=8 + 4 * (i % 3), clamp=False, zero_frac=0.25 if i != 3 else 1.0) for i in range(7)]
    dev_t = [[torch.from_numpy(l[k]).to("cuda:0") for k in range(4)] for l in lat]
    plain = gmc.compress_batch(*[[t[k] for t in dev_t] for k in range(4)])
    items = (_lib.fgmm_item * 7)()
    keep = []
    for i, t in enumerate(dev_t):
        it, M, hw, _ = gmc._item(t[0], t[1], t[2], t[3], keep)
        yq, zb = torch.empty_like(t[0]), torch.empty(M, dtype=torch.int64)
        it.yq_out, it.zero_bitmap = yq.data_ptr(), zb.data_ptr()
        items[i] = it
        keep += [yq, zb]
    store, asked, threads = {}, [], set()

    def alloc(user, item, nbytes, refuse=-1):
        asked.append(item)
        threads.add(threading.get_ident())
        if item == refuse:
            return None
        store[item] = C.create_string_buffer(nbytes)
        return C.addressof(store[item])

    sink = _lib.fgmm_sink(_lib.SINK_ALLOC(alloc), None)
    rc = L.fgmm_gmc_compress_batch_to(_lib.ctx(0), torch.cuda.current_stream().cuda_stream, items, 7, 0, 1, C.byref(sink))
    assert rc == 0, L.fgmm_last_error()
    assert sorted(asked) == list(range(7)), asked  # once per item
    assert threads == {threading.get_ident()}      # ... on the calling thread (which serves its workers' requests inside the call)
    for i in range(7):
        assert items[i].bytes == C.addressof(store[i]) and items[i].bytes_len == len(store[i].raw)
        assert store[i].raw == plain[i][0][0] and items[i].abs_max == plain[i][0][1]
    # a sink that refuses item 4
    store.clear(), asked.clear()
    sink2 = _lib.fgmm_sink(_lib.SINK_ALLOC(lambda u, i, n: alloc(u, i, n, refuse=4)), None)
    rc = L.fgmm_gmc_compress_batch_to(_lib.ctx(0), torch.cuda.current_stream().cuda_stream, items, 7, 0, 1, C.byref(sink2))
    assert rc == 4 and sorted(asked) == list(range(7))  # FGMM_ERR_NOMEM; the other items' encoders ran to their end
    assert all(not items[i].bytes and not items[i].ckpt for i in range(7))
    # a one-item call (coded on the calling thread itself)
    threads.clear(), asked.clear()
    rc = L.fgmm_gmc_compress_batch_to(_lib.ctx(0), torch.cuda.current_stream().cuda_stream, items, 1, 0, 1, C.byref(sink))
    assert rc == 0 and asked == [0] and threads == {threading.get_ident()} and store[0].raw == plain[0][0][0]
    # the compiled module's objects (stacked inputs go through it)
    ys, ss, ms, ws = (torch.cat([dev_t[i][k] for i in (0, 6)]) for k in range(4))
    for stride in (0, 256):
        g = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=stride)
        res = g.compress_batch(ys, ss, ms, ws)
        assert _lib.native() is not None
        for j, i in enumerate((0, 6)):
            b = res.strings[j]
            assert type(b) is (CheckpointedBytes if stride else bytes) and b == plain[i][0][0] and hash(b) == hash(plain[i][0][0])
            assert b[:8] == plain[i][0][0][:8] and len(b) == len(plain[i][0][0]) and bytes(b) + b"x" == plain[i][0][0] + b"x"
            assert {b: 1}[plain[i][0][0]] == 1


@pytest.fixture
def ctx_options():
    """set options of the process-wide context for one test and restore them afterwards"""
    saved = {}

    def set_(**kw):
        for k, v in kw.items():
            saved.setdefault(k, _lib.get_option(0, k))
            _lib.set_option(0, k, v)

    yield set_
    for k, v in saved.items():
        _lib.set_option(0, k, v)


@pytest.mark.parametrize("pieces", [None, 1, 7])
def test_batch_of_ragged_empty_and_tiny_items(oracle, ctx_options, pieces):
    """one batch mixing full-size items with all-zero ones (empty streams), single-channel / single-position ones and
    odd sizes: every item must equal its own single-item result and the oracle, whatever part of the decode pipeline
    (whole tables, pieces with fewer blocks than pieces) it falls into"""
    if pieces is not None:
        ctx_options(pieces=pieces)
    gmc = GaussianMixtureConditional(K=4, mode="logistic")
    specs = [(1, (192, 32, 24), 0.1), (2, (5, 3, 3), 1.0), (3, (1, 1, 2), 0.0), (4, (3, 5, 7), 0.0), (5, (17, 1, 1), 0.3),
             (6, (192, 32, 24), 0.0), (7, (6, 4, 4), 1.0), (8, (2, 2, 2), 0.0), (9, (2, 64, 66), 0.0)]
    lat = [T.make_latent(seed, M=M, h=h, w=w, clamp=False, zero_frac=zf) for seed, (M, h, w), zf in specs]
    ys, ss, ms, ws = ([dv(l[k]) for l in lat] for k in range(4))
    res = gmc.compress_batch(ys, ss, ms, ws)
    for i, l in enumerate(lat):
        sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(*l)
        (b, abs_max, zb), yq = res[i]
        assert b == oracle.encode_gmm("logistic", sym, s, m, wt) and abs_max == am and zb.tolist() == zbm.tolist(), specs[i]
        assert np.array_equal(yq.cpu().numpy(), yqn)
    outs = gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
    for i in range(len(lat)):
        assert outs[i].shape == ys[i].shape and torch.equal(outs[i], res[i][1]), specs[i]


def test_items_without_positions_in_a_batch(oracle, ctx_options):
    """An item with h * w == 0 (tensors without elements: null device pointers) beside ordinary ones - outside what the reference
    can be handed (`y.max()` of an empty tensor raises, entropy_models.py:834) but inside what a C caller can pass: the empty stream,
    an all-zero bitmap, abs_max 1; the neighbours untouched; also with the segmented table layout asked for (found on the fake
    device, round 5: such an item had no statistics arrays and the segmented encoder refused a zero-length segment)."""
    gmc = GaussianMixtureConditional(K=4, mode="polya")
    lat = [T.make_latent(7400 + i, M=24, h=8, w=12, zero_frac=0.1) for i in range(3)]
    ys, ss, ms, ws = ([dv(l[k]) for l in lat] for k in range(4))
    for k, t in enumerate((ys, ss, ms, ws)):
        t.insert(1, torch.zeros((1, 6 if k == 0 else 24, 0, 12), dtype=torch.float32, device=DEV))
    for segs in (1, 2):
        ctx_options(enc_segs=segs)
        res = gmc.compress_batch(ys, ss, ms, ws)
        (b, abs_max, zb), yq = res[1]
        assert bytes(b) == bytes.fromhex("0000008000000000") and abs_max == 1 and zb.tolist() == [0] * 6 and tuple(yq.shape) == (1, 6, 0, 12)
        for i, l in zip((0, 2, 3), lat):
            sym, s_, m_, w_, am, zbm, yqn = T.to_coder_inputs(*l)
            assert res[i][0][0] == oracle.encode_gmm("polya", sym, s_, m_, w_) and res[i][0][1] == am, (segs, i)
        outs = gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
        assert tuple(outs[1].shape) == (1, 6, 0, 12) and all(torch.equal(outs[i], res[i][1]) for i in (0, 2, 3)), segs


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_stacked_batch_equals_item_lists(dtype):
    """Items of one shape given as ONE tensor each ([N, M, h, w] / [N, K*M, h, w], batch-strided views included)
    must give exactly what the list form gives: same bytes, side information and tensors."""
    gmc = GaussianMixtureConditional(K=4, mode="as")
    N = 7
    lat = [T.make_latent(100 + i, M=24, h=8, w=12, zero_frac=0.2) for i in range(N)]
    if dtype == "f16":
        lat = [(y,) + tuple(T.to_float16_planes(sg, mu, pi)) for y, sg, mu, pi in lat]
    y, sg, mu, pi = (torch.cat([dv(l[k]) for l in lat]) for k in range(4))
    # parameters as chunk(3, 1) views of one [N, 3*K*M, h, w] head output: batch stride != K*M*h*w
    head = torch.cat([sg, mu, pi], dim=1)
    sg_v, mu_v, pi_v = head.chunk(3, 1)
    a = gmc.compress_batch(y, sg_v, mu_v, pi_v)
    b = gmc.compress_batch([y[i:i + 1] for i in range(N)], [sg[i:i + 1] for i in range(N)], [mu[i:i + 1] for i in range(N)],
                           [pi[i:i + 1] for i in range(N)])
    for (sa, qa), (sb, qb) in zip(a, b):
        assert sa[0] == sb[0] and sa[1] == sb[1] and torch.equal(sa[2], sb[2]) and torch.equal(qa, qb)
    outs = gmc.decompress_batch([r[0][0] for r in a], [r[0][1] for r in a], [r[0][2] for r in a], sg_v, mu_v, pi_v)
    outs2 = gmc.decompress_batch([r[0][0] for r in a], [r[0][1] for r in a], torch.stack([r[0][2] for r in a]), sg, mu, pi)
    for i in range(N):
        assert outs[i].shape == (1, 24, 8, 12) and torch.equal(outs[i], a[i][1]) and torch.equal(outs2[i], a[i][1])
        assert torch.equal(a[i][1], torch.round(y[i:i + 1]))
    with pytest.raises(RuntimeError):
        gmc.compress_batch(y, sg, mu[:-1], pi)
    with pytest.raises(RuntimeError):
        gmc.decompress_batch([r[0][0] for r in a][:-1], [r[0][1] for r in a], [r[0][2] for r in a], sg, mu, pi)


@pytest.mark.parametrize("pieces,first,threads", [(4, 2, 0), (1, 2, 0), (2, 1, 3), (3, 9, 0), (8, 2, 2), (7, 1, 16), (8, 2, 1), (16, 2, 0), (32, 2, 0), (0, 3, 5)])
def test_decode_piece_schedule_settings(ctx_options, pieces, first, threads):
    """The tables of a decode batch land on the host in pieces, piece-major, and the host workers take (bitstream, piece) tasks as
    they land (fgmm_decode.cpp): every setting of the schedule, and any number of workers, must give the same symbols; the encoder side codes
    ceil(9 / workers) bitstreams in turn per worker (enc_ways), up to four."""
    if threads:
        _lib.set_threads(0, threads)  # a fresh context: before the options
    try:
        ctx_options(pieces=pieces, dec_first=first, ef_min=14 + 35 * (pieces & 1), ef_rows=(pieces >> 1) & 1)
        _piece_schedule_case()
    finally:
        if threads:
            _lib.set_threads(0, 0)


def _piece_schedule_case():
    gmc = GaussianMixtureConditional(K=4, mode="polya")
    ys, ss, ms, ws = [], [], [], []
    for seed in range(9):
        # a mix of sizes: full Kodak halves (pool >> 1 MiB, land in pieces) and small items (land whole)
        y, sg, mu, pi = T.make_latent(seed) if seed % 3 else T.make_latent(seed, M=24, h=8, w=6)
        ys.append(dv(y)); ss.append(dv(sg)); ms.append(dv(mu)); ws.append(dv(pi))
    res = gmc.compress_batch(ys, ss, ms, ws)
    outs = gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
    for i in range(9):
        assert torch.equal(outs[i], torch.round(ys[i])), i
    # a truncated stream in the window still fails loudly (the decoder must not wait for pieces forever)
    bad = [r[0][0] for r in res]
    bad[8] = bad[8][:64]
    with pytest.raises(RuntimeError):
        gmc.decompress_batch(bad, [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)


@pytest.mark.parametrize("mode", MODES)
def test_decode_paths_agree(ctx_options, mode):
    """one batch decoded (a) by the single-pass kernel into a staging area provisioned for the worst case, (b) into one
    that is far too small, so that every launch overflows and is re-run with the size its cursor reports, (c) with tiny
    and (d) with the largest blocks the kernel takes, (e) by the generic two-pass kernels (the LDS budget set so low
    that no item fits): the same symbols every time.  Then the context gives its buffers back and works again."""
    gmc = GaussianMixtureConditional(K=4, mode=mode)
    lat = [T.make_latent(300 + i, M=48, h=16, w=16, clamp=False, zero_frac=0.1) for i in range(6)]
    ys, ss, ms, ws = ([dv(l[k]) for l in lat] for k in range(4))
    res = gmc.compress_batch(ys, ss, ms, ws)
    args = ([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
    want = [r[1] for r in res]

    def check(tag):
        out = gmc.decompress_batch(*args)
        for i in range(6):
            assert torch.equal(out[i], want[i]), (tag, i)
        return _lib.ctx_stat(0, 1)

    _lib.set_profiling(0, True)  # the kernels count the edges they evaluate only for a profiling caller
    ctx_options(ef_rows=1)
    bytes_a = check("worst case")
    ctx_options(stage_max_mb=1)
    _lib.trim(0)
    assert check("overflow, re-run") == bytes_a
    ctx_options(stage_max_mb=0, tab_cap_e=32768)
    _lib.trim(0)
    check("largest blocks")
    ctx_options(tab_cap_e=16 * (2 * max(r[0][1] + 1 for r in res) + 2))
    check("16-latent blocks")
    ctx_options(ef_rows=1)
    bytes_ef = check("Elias-Fano rows")
    ctx_options(ef_rows=2)
    assert check("uint16 rows only") > 1.15 * bytes_ef
    ctx_options(ef_rows=0, tab_cap_e=256)
    assert _lib.ctx_stat(0, 3) > 0
    check("generic kernels")
    assert _lib.ctx_stat(0, 3) == 0  # the generic path does not count its edges: proof that it ran
    _lib.set_profiling(0, False)
    _lib.trim(0)


@pytest.mark.parametrize("mode", MODES)
def test_fp16_parameter_planes(oracle, mode):
    """BASELINE configs[4]: fp16 (mu, sigma, pi), fp32 CDF.  Result == the reference path fed the widened values."""
    for seed, (M, h, w) in ((41, (32, 16, 12)), (42, (7, 5, 3))):  # 8-B vector loads, and the scalar kernel
        y, sg, mu, pi = T.make_latent(seed, M=M, h=h, w=w, clamp=False, zero_frac=0.1)
        sg16, mu16, pi16 = T.to_float16_planes(sg, mu, pi)  # weights rounded toward zero: sum <= 1 (see helper)
        gmc = GaussianMixtureConditional(K=4, mode=mode)
        t16 = [dv(a) for a in (sg16, mu16, pi16)]
        assert t16[0].dtype == torch.float16
        (b, abs_max, zb), yq = gmc.compress(dv(y), *t16)
        sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(y, *(a.astype(np.float32) for a in (sg16, mu16, pi16)))
        assert b == oracle.encode_gmm(mode, sym, s, m, wt) and abs_max == am
        assert torch.equal(gmc.decompress(b, abs_max, zb, *t16), yq)
    # round-to-nearest fp16 weights can sum above 1: the reference algorithm then desynchronises (quantised edge
    # wraps past 65535); the HIP path must do exactly what the reference does on the widened values: same bytes,
    # and a decode that fails or mis-decodes the same way (here: both run off the end of the stream)
    y, sg, mu, pi = T.make_latent(42, M=7, h=5, w=3, clamp=False, zero_frac=0.1)
    bad16 = [a.astype(np.float16) for a in (sg, mu, pi)]
    sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(y, *(a.astype(np.float32) for a in bad16))
    if wt.sum(1).max() > 1.0:
        (b, abs_max, zb), yq = gmc.compress(dv(y), *[dv(a) for a in bad16])
        assert b == oracle.encode_gmm(mode, sym, s, m, wt)


def test_elic_channel_group_shapes(oracle):
    """BASELINE configs[4] geometry: ELIC's five channel groups 16/16/32/64/192 of a 4K latent (h*w = 136*120 per half), one ragged
    batch: every bitstream - the 2.1 M and 3.1 M symbol ones included - against the reference's own bytes
    (tests/golden/fullsize.json "elic_groups") and against the oracle."""
    gmc = GaussianMixtureConditional(K=4, mode="polya")
    gold = json.load(open(os.path.join(GOLD, "fullsize.json")))["polya"]["elic_groups"]
    ys, ss, ms, ws, host = [], [], [], [], []
    for seed, M in ((51, 16), (52, 16), (53, 32), (54, 64), (55, 192)):
        y, sg, mu, pi = T.make_latent(seed, M=M, h=136, w=120, clamp=False)
        host.append((y, sg, mu, pi))
        ys.append(dv(y)); ss.append(dv(sg)); ms.append(dv(mu)); ws.append(dv(pi))
    res = gmc.compress_batch(ys, ss, ms, ws)  # ragged batch: M differs per item
    for i, seed in enumerate((51, 52, 53, 54, 55)):
        sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(*host[i])
        (b, abs_max, zb), yq = res[i]
        assert (len(b), hashlib.md5(b).hexdigest(), abs_max) == (gold[str(seed)]["len"], gold[str(seed)]["md5"], gold[str(seed)]["abs_max"]), seed
        assert b == oracle.encode_gmm("polya", sym, s, m, wt) and abs_max == am and zb.tolist() == zbm.tolist()
    outs = gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
    for i in range(len(res)):
        assert torch.equal(outs[i], res[i][1]) and torch.equal(res[i][1], torch.round(ys[i]))


def test_elic4k_image_fp16_planes_equals_reference_bytes():
    """BASELINE configs[4] as bench.py runs it: the ten bitstreams of a 4K image (groups 16/16/32/64/192 x two halves, seeds 0..9),
    fp16 (mu, sigma, pi) planes, encoded in ONE call and decoded stage by stage - each stream's bytes against what the compiled
    reference returns when fed the widened planes (tests/golden/fullsize.json "elic4k_image0")."""
    gold = json.load(open(os.path.join(GOLD, "fullsize.json")))["polya"]["elic4k_image0"]
    gmc = GaussianMixtureConditional(K=4, mode="polya")
    ys, ss, ms, ws = [], [], [], []
    k = 0
    for g in (16, 16, 32, 64, 192):
        for _ in range(2):
            y, sg, mu, pi = T.make_latent(k, M=g, h=136, w=120)
            sg, mu, pi = T.to_float16_planes(sg, mu, pi)
            ys.append(dv(y)); ss.append(dv(sg)); ms.append(dv(mu)); ws.append(dv(pi))
            k += 1
    assert ss[0].dtype == torch.float16
    res = gmc.compress_batch(ys, ss, ms, ws)
    for k in range(10):
        (b, abs_max, zb), yq = res[k]
        assert (len(b), hashlib.md5(b).hexdigest(), abs_max, int(zb.sum())) == (gold[str(k)]["len"], gold[str(k)]["md5"], gold[str(k)]["abs_max"],
                                                                                gold[str(k)]["nz_channels"]), k
    for k in range(10):  # the codec's decode schedule: one stage at a time
        out = gmc.decompress_batch([res[k][0][0]], [res[k][0][1]], [res[k][0][2]], [ss[k]], [ms[k]], [ws[k]])
        assert torch.equal(out[0], res[k][1]) and torch.equal(res[k][1], torch.round(ys[k])), k


@pytest.mark.parametrize("mode", MODES)
def test_softmax_over_k_fused_into_the_kernels(oracle, mode):
    """the kernels take the parameter head's LOGITS and compute pi = softmax over K themselves.
    (1) the device sequence is within 1e-6 of torch.softmax (in fact ~2e-7), exact on the {0, -inf} logits of the exact
    networks; (2) a stream coded from logits == the un-fused kernels fed that same pi == the oracle fed it, byte for byte:
    the fusion changes where pi is computed, nothing else; (3) it decodes from logits; (4) through the latent codec."""
    L, ctx = _lib.lib(), _lib.ctx(0)
    rng = np.random.default_rng(91)
    M, h, w = 24, 16, 12
    y, sg, mu, pi = T.make_latent(92, M=M, h=h, w=w, clamp=False, zero_frac=0.1)
    lg = (rng.standard_normal((1, 4 * M, h, w)) * 3).astype(np.float32)
    lg[0, :M, :2] = -np.inf        # a dead component
    lg[0, M:2 * M, 3] = 60.0       # one that takes everything
    lg[0, 2 * M:3 * M, 5] = -45.0  # below the e^-41 cut
    # (1) the probe: rows (n, 4) = the four logits of a latent
    rows = torch.from_numpy(np.ascontiguousarray(lg.reshape(4, -1).T)).to(DEV)
    out = torch.empty_like(rows)
    torch.cuda.synchronize()
    _lib.check(L.fgmm_softmax4_hip(ctx, None, rows.data_ptr(), out.data_ptr(), rows.size(0)))
    ref = torch.softmax(rows.double(), dim=1)
    assert float((out.double() - ref).abs().max()) < 1e-6
    assert float((out.sum(1) - 1).abs().max()) < 3e-7
    exact = torch.tensor([[0.0, -np.inf, -np.inf, -np.inf], [0.0, 0.0, -np.inf, -np.inf], [0.0, 0.0, 0.0, 0.0], [5.0, 5.0, -np.inf, 5.0]], device=DEV)
    oute = torch.empty_like(exact)
    _lib.check(L.fgmm_softmax4_hip(ctx, None, exact.data_ptr(), oute.data_ptr(), 4))
    assert torch.equal(oute[:3].cpu(), torch.tensor([[1.0, 0, 0, 0], [0.5, 0.5, 0, 0], [0.25] * 4]))
    assert float((oute[3].cpu() - torch.tensor([1 / 3, 1 / 3, 0, 1 / 3])).abs().max()) < 1e-7
    # (2) fused == un-fused on the device's pi == oracle on the device's pi
    pi_dev = out.T.reshape(1, 4 * M, h, w).contiguous()
    gmc = GaussianMixtureConditional(K=4, mode=mode)
    t = [dv(a) for a in (y, sg, mu)]
    (b, abs_max, zb), yq = gmc.compress(*t, dv(lg), weights_are_logits=True)
    (b2, abs_max2, zb2), yq2 = gmc.compress(*t, pi_dev)
    assert b == b2 and abs_max == abs_max2 and torch.equal(zb, zb2) and torch.equal(yq, yq2)
    sym, s_, m_, w_, am, zbm, yqn = T.to_coder_inputs(y, sg, mu, pi_dev.cpu().numpy())
    assert b == oracle.encode_gmm(mode, sym, s_, m_, w_)
    # (3) decode from logits (both table kernels: the batched path and, with a tiny LDS budget, the generic one)
    assert torch.equal(gmc.decompress(b, abs_max, zb, t[1], t[2], dv(lg), weights_are_logits=True), yq)
    saved = _lib.get_option(0, "tab_cap_e")
    try:
        _lib.set_option(0, "tab_cap_e", 256)
        assert torch.equal(gmc.decompress(b, abs_max, zb, t[1], t[2], dv(lg), weights_are_logits=True), yq)
    finally:
        _lib.set_option(0, "tab_cap_e", saved)
    # (4) the latent codec with fuse_softmax: [scales | means | logits] in, no pi plane
    from flashgmm_amd.latent_codecs import GaussianMixtureConditionalLatentCodec

    codec = GaussianMixtureConditionalLatentCodec(K=4, mode=mode, fuse_softmax=True)
    head = torch.cat([dv(sg), dv(mu), dv(lg)], dim=1)
    enc = codec.compress(dv(y), head)
    assert enc["strings"][0][0] == b
    assert torch.equal(codec.decompress(enc["strings"], enc["shape"], head)["y_hat"], yq)
    with pytest.raises(ValueError):
        GaussianMixtureConditionalLatentCodec(K=4, quantizer="weighted_mean_ste", fuse_softmax=True)


def test_config4_elic_4k_fp16_through_the_group_codec(oracle):
    """BASELINE configs[4] as stated: ELIC on a 4K image — y [1, 320, 136, 240], channel groups 16/16/32/64/192, each a
    checkerboard codec over the GMM entropy model (models/elic_gmm.py:198-219) — with fp16 (mu, sigma, pi) planes and fp32
    CDF arithmetic, all five groups in their sequential flow through ChannelGroupsLatentCodec on exact networks.  Checked:
    all ten bitstreams byte for byte against the oracle on the (widened) parameters each half was
    coded with; every group through decode(encode(y)) == y_hat; encode in ONE batched call == the group-by-group schedule."""
    from flashgmm_amd.latent_codecs import ChannelGroupsLatentCodec, CheckerboardLatentCodec, GaussianMixtureConditionalLatentCodec

    groups, c_side, h, w = [16, 16, 32, 64, 192], 8, 136, 240
    Ctx, Par = T.exact_modules()
    seen = []  # (y_code, scales, means, weights) of every half, in coding order

    class Spy(GaussianMixtureConditionalLatentCodec):
        def coder_inputs(self, y, ctx_params):
            out = super().coder_inputs(y, ctx_params)
            seen.append(out)
            return out

    def build():
        latent = {f"y{k}": CheckerboardLatentCodec(latent_codec={"y": Spy(K=4, mode="polya", param_dtype=torch.float16)},
                                                   context_prediction=Ctx(g, 2 * g), entropy_parameters=Par(2 * g + (k > 0) * 2 * g + c_side, g))
                  for k, g in enumerate(groups)}
        chctx = {f"y{k}": Ctx(sum(groups[:k]), 2 * groups[k]) for k in range(1, len(groups))}
        return ChannelGroupsLatentCodec(groups=groups, channel_context=chctx, latent_codec=latent)

    y, side = T.exact_codec_inputs(61, sum(groups), c_side, h, w)
    y[:, :2] = 0.3  # two dead channels in the first group
    yd, sided = dv(y), dv(side)
    codec = build()
    enc = codec.compress(yd, sided)
    assert len(enc["strings"]) == 10 and len(seen) == 10 and [tuple(s_) for s_ in enc["shape"]] == [(g, h, w) for g in groups]
    assert all(p[1].dtype == torch.float16 and tuple(p[1].shape) == (1, 4 * g, h, w // 2) for p, g in zip(seen, [g for g in groups for _ in (0, 1)]))
    for i in range(10):  # all five groups, both halves (the 64- and 192-channel ones: 1 M / 3.1 M symbols each): the reference path fed the widened planes
        yc, sg, mu, pi = (t.cpu().numpy() for t in seen[i])
        sym, s_, m_, w_, am, zbm, yqn = T.to_coder_inputs(yc, *(a.astype(np.float32) for a in (sg, mu, pi)))
        b, abs_max, zb = enc["strings"][i]
        assert b == oracle.encode_gmm("polya", sym, s_, m_, w_) and abs_max == am and zb.cpu().tolist() == zbm.tolist(), i
    dec = codec.decompress(enc["strings"], enc["shape"], sided)
    assert torch.equal(dec["y_hat"], enc["y_hat"]) and torch.equal(enc["y_hat"], torch.round(yd))
    # the group-by-group schedule (ten encode calls) gives the same ten bitstreams
    seq = build()
    seq._one_call = lambda codecs: False
    enc2 = seq.compress(yd, sided)
    assert [(b, a) for b, a, _ in enc2["strings"]] == [(b, a) for b, a, _ in enc["strings"]] and torch.equal(enc2["y_hat"], enc["y_hat"])


def test_config4_images_in_flight_stage_major_equals_image_by_image(oracle):
    """configs[4] as a codec runs it on several images: ``compress_many`` codes every bitstream of every image in one
    call, ``decompress_many`` decodes STAGE-MAJOR — stage s (group, half) of every image in one batched call, the
    dependency chain of channel_groups.py:147-154 intact inside each image.  Checked on two images (ELIC's five groups,
    fp16 planes, a reduced plane size): batched == image by image, byte for byte and bit for bit; the bitstreams of
    the three small groups == the oracle fed the widened parameters; one decode call per stage whatever the image count."""
    from flashgmm_amd.latent_codecs import ChannelGroupsLatentCodec, CheckerboardLatentCodec, GaussianMixtureConditionalLatentCodec

    groups, c_side, h, w = [16, 16, 32, 64, 192], 8, 34, 60
    Ctx, Par = T.exact_modules()
    seen = []
    calls = []

    class Spy(GaussianMixtureConditionalLatentCodec):
        def coder_inputs(self, y, ctx_params):
            out = super().coder_inputs(y, ctx_params)
            seen.append(out)
            return out

    def build(checkpoint_stride=0):
        gmc = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=checkpoint_stride)
        real = gmc.decompress_batch
        gmc.decompress_batch = lambda strings, *a, **k: (calls.append(len(strings)), real(strings, *a, **k))[1]
        latent = {f"y{k}": CheckerboardLatentCodec(latent_codec={"y": Spy(K=4, gaussian_mixture_conditional=gmc, param_dtype=torch.float16)},
                                                   context_prediction=Ctx(g, 2 * g), entropy_parameters=Par(2 * g + (k > 0) * 2 * g + c_side, g))
                  for k, g in enumerate(groups)}
        chctx = {f"y{k}": Ctx(sum(groups[:k]), 2 * groups[k]) for k in range(1, len(groups))}
        return ChannelGroupsLatentCodec(groups=groups, channel_context=chctx, latent_codec=latent)

    images = [T.exact_codec_inputs(71 + n, sum(groups), c_side, h, w) for n in range(2)]
    ys, sides = [dv(y) for y, _ in images], [dv(sd) for _, sd in images]
    codec = build()
    one_by_one = [codec.compress(y, sd) for y, sd in zip(ys, sides)]
    seen.clear()
    many = codec.compress_many(ys, sides)
    assert len(seen) == 20
    for a, b in zip(many, one_by_one):
        assert [(s_[0], s_[1], s_[2].tolist()) for s_ in a["strings"]] == [(s_[0], s_[1], s_[2].tolist()) for s_ in b["strings"]]
        assert torch.equal(a["y_hat"], b["y_hat"]) and [tuple(x) for x in a["shape"]] == [tuple(x) for x in b["shape"]]
    for n in range(2):  # the three small groups of both images against the oracle
        for i in range(6):
            yc, sg, mu, pi = (t.cpu().numpy() for t in seen[10 * n + i])
            sym, s_, m_, w_, am, zbm, yqn = T.to_coder_inputs(yc, *(a.astype(np.float32) for a in (sg, mu, pi)))
            b, abs_max, zb = many[n]["strings"][i]
            assert b == oracle.encode_gmm("polya", sym, s_, m_, w_) and abs_max == am and zb.cpu().tolist() == zbm.tolist(), (n, i)
    calls.clear()
    dec = codec.decompress_many([m["strings"] for m in many], many[0]["shape"], sides)
    assert calls == [2] * 10  # ten stages, both images in every call
    calls.clear()
    for n in range(2):
        assert torch.equal(dec[n]["y_hat"], many[n]["y_hat"]) and torch.equal(many[n]["y_hat"], torch.round(ys[n]))
        ref = codec.decompress(many[n]["strings"], many[n]["shape"], sides[n])
        assert torch.equal(ref["y_hat"], dec[n]["y_hat"])
    assert calls == [1] * 20
    # ... and with checkpointed bitstreams: the same bytes (groups of five sizes in one encode call: the library orders its jobs by
    # size), every stage's call decoded by the GPU's segment decoder (fp16 planes; forced: the reduced planes are small)
    from flashgmm_amd import CheckpointedBytes
    ck = build(checkpoint_stride=256)
    many_ck = ck.compress_many(ys, sides)
    for a, b in zip(many_ck, many):
        assert all(isinstance(x[0], CheckpointedBytes) for x in a["strings"])
        assert [(bytes(s_[0]), s_[1], s_[2].tolist()) for s_ in a["strings"]] == [(s_[0], s_[1], s_[2].tolist()) for s_ in b["strings"]]
        assert torch.equal(a["y_hat"], b["y_hat"])
    _lib.set_option(0, "gpu_decode", 1)
    try:
        dec_ck = ck.decompress_many([m["strings"] for m in many_ck], many_ck[0]["shape"], sides)
        assert (_lib.ctx_stat(0, 4), _lib.ctx_stat(0, 5)) == (2, 0)  # (the last stage's call)
    finally:
        _lib.set_option(0, "gpu_decode", 0)
    for n in range(2):
        assert torch.equal(dec_ck[n]["y_hat"], many[n]["y_hat"])


@pytest.mark.parametrize("mode", MODES)
def test_checkpointed_streams_are_the_same_bytes_and_decode_on_all_workers(oracle, mode):
    """``GaussianMixtureConditional(checkpoint_stride=S)``: ``compress`` returns ``CheckpointedBytes`` — the reference's
    bitstream (== plain ``compress``, == the oracle) plus out-of-band checkpoints; ``decompress`` decodes the segments between
    them on all host workers and gives the sequential decoder's result: for a batch, for ONE bitstream (the case checkpoints
    exist for), with zero channels, with bypass-coded symbols, for stacked and listed inputs; plain bytes still decode."""
    from flashgmm_amd import CheckpointedBytes

    lat = [T.make_latent(300 + i, M=M, h=h, w=w, zero_frac=zf) for i, (M, h, w, zf) in
           enumerate([(192, 32, 24, 0.15), (48, 16, 8, 0.0), (192, 32, 24, 0.0), (7, 5, 3, 0.3), (64, 16, 12, 0.5)])]
    lat[2] = (lat[2][0] * np.float32(1.0), *lat[2][1:])
    lat[2][0].reshape(-1)[::53] *= 50  # bypass-coded symbols between the checkpoints
    dev = [[dv(a) for a in l] for l in lat]
    ys, ss, ms, ws = ([d[k] for d in dev] for k in range(4))
    plain = GaussianMixtureConditional(K=4, mode=mode)
    ck = GaussianMixtureConditional(K=4, mode=mode, checkpoint_stride=1024)
    r0 = plain.compress_batch(ys, ss, ms, ws)
    r1 = ck.compress_batch(ys, ss, ms, ws)
    for (a, b, l) in zip(r0, r1, lat):
        assert type(a[0][0]) is bytes and isinstance(b[0][0], CheckpointedBytes) and b[0][0] == a[0][0]
        assert a[0][1] == b[0][1] and torch.equal(a[0][2], b[0][2]) and torch.equal(a[1], b[1])
        n = int(b[0][2].sum()) * l[0].shape[2] * l[0].shape[3]
        assert len(b[0][0].ckpt) == max(n - 1, 0) // 1024 and b[0][0].ckpt_stride == 1024
    sym, s_, m_, w_, am, zbm, yqn = T.to_coder_inputs(*lat[1])
    assert bytes(r1[1][0][0]) == oracle.encode_gmm(mode, sym, s_, m_, w_)
    strings, ams, zbs = [r[0][0] for r in r1], [r[0][1] for r in r1], [r[0][2] for r in r1]
    # on the GPU (option gpu_decode = 1; by default the library takes it when a call has enough segments to make it the
    # faster decoder - these few small items would go to the host's workers): one workgroup per segment, no tables
    _lib.set_option(0, "gpu_decode", 1)
    out = ck.decompress_batch(strings, ams, zbs, ss, ms, ws)
    for o, r in zip(out, r1):
        assert torch.equal(o, r[1])
    # (a bitstream shorter than the stride has no note, and a half-width whose window does not fit the kernel's LDS budget -
    # 2 * (abs_max + 2) + 2 > 2048 edges - is not the kernel's: both take the table path)
    n_gpu = sum(len(b.ckpt) > 0 and 2 * (a + 1) + 2 <= 2048 for b, a in zip(strings, ams))
    assert n_gpu >= 3 and (_lib.ctx_stat(0, 4), _lib.ctx_stat(0, 5)) == (n_gpu, 0)
    one = ck.decompress(strings[0], ams[0], zbs[0], ss[0], ms[0], ws[0])
    assert torch.equal(one, r1[0][1]) and _lib.ctx_stat(0, 4) == 1
    assert 2 * (ams[2] + 1) + 2 > 2048  # the outliers' item is the one with the wide window (bypass-coded symbols on the table path)
    _lib.set_option(0, "gpu_decode", 0)
    # a whole Kodak-sized batch is the GPU's by default: 24 bitstreams x 125 segments
    big = [T.make_latent(700 + i) for i in range(24)]
    bt = [torch.cat([dv(l[k]) for l in big]) for k in range(4)]
    rb = ck.compress_batch(*bt)
    ob = ck.decompress_batch([x[0][0] for x in rb], [x[0][1] for x in rb], [x[0][2] for x in rb], *bt[1:])
    assert all(torch.equal(o, x[1]) for o, x in zip(ob, rb)) and (_lib.ctx_stat(0, 4), _lib.ctx_stat(0, 5)) == (24, 0)
    # ... that launch had 3 048 segments: three waves each (two producers).  With a note every 256 symbols it has 12 120 and takes
    # the two-wave shape (launch_segdec: more than 4 096 segments); the order of the segments (heaviest first) is the host's
    ck256 = GaussianMixtureConditional(K=4, mode=mode, checkpoint_stride=256)
    rb2 = ck256.compress_batch(*bt)
    assert all(bytes(a[0][0]) == bytes(b[0][0]) for a, b in zip(rb2, rb)) and sum(len(x[0][0].ckpt) + 1 for x in rb2) > 4096
    ob2 = ck256.decompress_batch([x[0][0] for x in rb2], [x[0][1] for x in rb2], [x[0][2] for x in rb2], *bt[1:])
    assert all(torch.equal(o, x[1]) for o, x in zip(ob2, rb)) and (_lib.ctx_stat(0, 4), _lib.ctx_stat(0, 5)) == (24, 0)
    # through the table path, segments on the host workers (gpu_decode = 2):
    # (threads, option ckpt_decode: 0 = segments when the call has fewer bitstreams than workers, 1 = always, 2 = never)
    for threads, how in ((16, 0), (3, 1), (1, 1), (3, 0), (16, 2)):
        _lib.set_threads(0, threads)
        _lib.set_option(0, "ckpt_decode", how)
        _lib.set_option(0, "gpu_decode", 2)
        try:
            out = ck.decompress_batch(strings, ams, zbs, ss, ms, ws)  # segments on the workers
            for o, r in zip(out, r1):
                assert torch.equal(o, r[1])
            assert _lib.ctx_stat(0, 4) == 0 and _lib.ctx_stat(0, 1) > 0
            one = ck.decompress(strings[0], ams[0], zbs[0], ss[0], ms[0], ws[0])  # ONE bitstream, 124 segments
            assert torch.equal(one, r1[0][1])
        finally:
            _lib.set_threads(0, 0)
            _lib.set_option(0, "ckpt_decode", 0)
            _lib.set_option(0, "gpu_decode", 0)
    out = plain.decompress_batch([bytes(b) for b in strings], ams, zbs, ss, ms, ws)  # the same streams without their notes
    for o, r in zip(out, r1):
        assert torch.equal(o, r[1])
    # stacked input (one tensor per operand)
    st = [torch.cat([dev[0][k], dev[2][k]]) for k in range(4)]
    rs = ck.compress_batch(*st)
    assert [bytes(x[0][0]) for x in rs] == [bytes(r1[0][0][0]), bytes(r1[2][0][0])] and all(isinstance(x[0][0], CheckpointedBytes) for x in rs)
    outs = ck.decompress_batch([x[0][0] for x in rs], [x[0][1] for x in rs], [x[0][2] for x in rs], *st[1:])
    assert torch.equal(outs[0], rs[0][1]) and torch.equal(outs[1], rs[1][1])


def test_checkpointed_codec_result_through_the_container():
    """the group codec over checkpointed entropy models: the ten strings of an ELIC-shaped latent are CheckpointedBytes
    equal to the plain codec's, decode to the same y_hat (one bitstream per call: the segments run on all workers), and
    survive the byte container with their checkpoints"""
    from flashgmm_amd import CheckpointedBytes, container as Cn
    from flashgmm_amd.latent_codecs import ChannelGroupsLatentCodec, CheckerboardLatentCodec, GaussianMixtureConditionalLatentCodec

    groups, c_side, h, w = [16, 16, 32, 64, 192], 8, 34, 60
    Ctx, Par = T.exact_modules()

    def build(stride):
        latent = {f"y{k}": CheckerboardLatentCodec(latent_codec={"y": GaussianMixtureConditionalLatentCodec(K=4, mode="polya", checkpoint_stride=stride)},
                                                   context_prediction=Ctx(g, 2 * g), entropy_parameters=Par(2 * g + (k > 0) * 2 * g + c_side, g))
                  for k, g in enumerate(groups)}
        chctx = {f"y{k}": Ctx(sum(groups[:k]), 2 * groups[k]) for k in range(1, len(groups))}
        return ChannelGroupsLatentCodec(groups=groups, channel_context=chctx, latent_codec=latent)

    y, side = T.exact_codec_inputs(81, sum(groups), c_side, h, w)
    yd, sided = dv(y), dv(side)
    plain, noted = build(0), build(256)
    e0, e1 = plain.compress(yd, sided), noted.compress(yd, sided)
    assert all(type(a[0]) is bytes and isinstance(b[0], CheckpointedBytes) and a[0] == b[0] and a[1] == b[1] and torch.equal(a[2], b[2])
               for a, b in zip(e0["strings"], e1["strings"]))
    assert sum(len(b[0].ckpt) for b in e1["strings"]) > 100 and torch.equal(e0["y_hat"], e1["y_hat"])
    d1 = noted.decompress(e1["strings"], e1["shape"], sided)
    assert torch.equal(d1["y_hat"], e1["y_hat"])
    blob = Cn.pack(e1["strings"], e1["shape"])
    s2, shape2 = Cn.unpack(blob, device="cuda")
    assert all(isinstance(s_[0], CheckpointedBytes) for s_ in s2)
    assert torch.equal(plain.decompress(s2, shape2, sided)["y_hat"], e1["y_hat"])  # any decoder of this library uses the notes it is given
    assert Cn.side_info_bytes(e1["strings"], e1["shape"]) - Cn.side_info_bytes(e0["strings"], e0["shape"]) == sum(8 + 12 * len(b[0].ckpt) for b in e1["strings"])


@pytest.mark.parametrize("mode", MODES)
def test_gpu_segment_decoder_equals_the_table_path_in_every_variant(oracle, mode):
    """segdec_kernel (checkpointed bitstreams decoded ON the GPU, edges across the lanes, count + ballot for the reference's
    bisection) against the table path and the oracle: fp32 / fp16 planes, sigma clamped or not, weights as logits,
    windows beyond 64 edges (several passes per symbol), bypass-coded symbols (the synthetic latents have ~0.2 % of them), dead channels, tiny items (one segment,
    fewer than 64 latents), and parameters the kernel must hand back (decreasing rows, NaN sigma)."""
    rng = np.random.default_rng(91)
    _lib.set_option(0, "gpu_decode", 1)  # (small items: by default they would be the host workers')
    cases = []
    for f16, clamp, logits in ((False, True, False), (True, True, False), (False, False, False), (False, True, True)):
        y, sg, mu, pi = T.make_latent(400 + len(cases), M=40, h=16, w=12, clamp=False, zero_frac=0.2)
        if not clamp:
            sg = np.maximum(sg, np.float32(0.02))
        w_in = np.log(np.maximum(pi, 1e-6)).astype(np.float32) if logits else pi
        if f16:
            sg, mu, w_in = T.to_float16_planes(sg, mu, w_in)
        cases.append((y, sg, mu, w_in, clamp, logits))
    # wide windows: sigma up to 60 -> more than 64 edges between the saturated tails; and bypass-coded outliers
    y, sg, mu, pi = T.make_latent(410, M=24, h=16, w=12, clamp=False)
    sg = (sg * np.float32(12.0)).astype(np.float32)
    y = (y * np.float32(6.0)).astype(np.float32)
    y.reshape(-1)[::41] *= 2
    cases.append((y, sg, mu, pi, True, False))
    y, sg, mu, pi = T.make_latent(411, M=3, h=3, w=5, clamp=False)  # 45 latents: one short segment
    cases.append((y, sg, mu, pi, True, False))
    for ci, (y, sg, mu, w_in, clamp, logits) in enumerate(cases):
        t = [dv(a) for a in (y, sg, mu, w_in)]
        plain = GaussianMixtureConditional(K=4, mode=mode, clamp_scales=clamp)
        ck = GaussianMixtureConditional(K=4, mode=mode, clamp_scales=clamp, checkpoint_stride=256)
        (b0, am0, zb0), yq0 = plain.compress(*t, weights_are_logits=logits)
        (b1, am1, zb1), yq1 = ck.compress(*t, weights_are_logits=logits)
        assert bytes(b1) == b0 and am0 == am1 and torch.equal(yq0, yq1)
        want = plain.decompress(b0, am0, zb0, *t[1:], weights_are_logits=logits)
        got = ck.decompress(b1, am1, zb1, *t[1:], weights_are_logits=logits)
        assert torch.equal(got, want) and torch.equal(got, yq1), ci
        on_gpu = len(b1.ckpt) > 0 and 2 * (am1 + 1) + 2 <= 2048
        assert (_lib.ctx_stat(0, 4), _lib.ctx_stat(0, 5)) == ((1, 0) if on_gpu else (0, 0)), ci  # decoded by the GPU's segment decoder
        assert on_gpu or ci == 5, ci  # (only the 45-latent item has no note)
    # rows the kernel does not settle itself: negative weights make the CDF decrease, a NaN sigma poisons a latent; a garbage
    # stream asks for intervals that do not exist.  Whatever the kernel does with them, the result is the table path's
    y, sg, mu, pi = T.make_latent(420, M=16, h=16, w=12, clamp=False)
    pi = pi.copy()
    pi[:, 0::4][..., ::3, :] *= np.float32(-0.7)  # some components with negative weight
    sg = sg.copy()
    sg.reshape(-1)[::997] = np.nan
    t = [dv(a) for a in (y, sg, mu, pi)]
    plain = GaussianMixtureConditional(K=4, mode=mode)
    ck = GaussianMixtureConditional(K=4, mode=mode, checkpoint_stride=256)
    (b0, am0, zb0), yq0 = plain.compress(*t)
    (b1, am1, zb1), yq1 = ck.compress(*t)
    assert bytes(b1) == b0
    assert torch.equal(ck.decompress(b1, am1, zb1, *t[1:]), plain.decompress(b0, am0, zb0, *t[1:]))
    from flashgmm_amd import CheckpointedBytes
    junk = bytes(rng.integers(0, 256, len(b0) & ~3, dtype=np.uint8))
    try:
        want = plain.decompress(junk, am0, zb0, *t[1:])
    except RuntimeError:
        want = None
    try:
        got = ck.decompress(CheckpointedBytes(junk, b1.ckpt, 256), am1, zb1, *t[1:])
    except RuntimeError:
        got = None
    assert (want is None and got is None) or torch.equal(got, want)
    _lib.set_option(0, "gpu_decode", 0)


def test_wrong_checkpoints_cost_a_sequential_decode_never_a_wrong_symbol():
    """checkpoints are verified against one another segment by segment: flipped states, shifted positions, another stream's
    notes, a wrong stride — the result is the sequential decoder's every time; a truncated stream is still an error"""
    from flashgmm_amd import CheckpointedBytes

    rng = np.random.default_rng(77)
    y, sg, mu, pi = T.make_latent(311, M=96, h=32, w=24, zero_frac=0.1)
    t = [dv(a) for a in (y, sg, mu, pi)]
    y2, sg2, mu2, pi2 = T.make_latent(312, M=96, h=32, w=24, zero_frac=0.1)
    gmc = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=512)
    (b, am, zb), yq = gmc.compress(*t)
    (b2, _, _), _ = gmc.compress(*[dv(a) for a in (y2, sg2, mu2, pi2)])
    assert len(b.ckpt) > 50
    for trial in range(10):
        ck = b.ckpt.copy()
        k = int(rng.integers(0, len(ck)))
        if trial % 5 == 0:
            ck["x"][k] ^= np.uint64(1 << int(rng.integers(0, 62)))
        elif trial % 5 == 1:
            ck["pos"][k] += np.uint64(1)
        elif trial % 5 == 2:
            ck["pos"][k] = np.uint64(1 << 45)
        elif trial % 5 == 3:
            ck = b2.ckpt[:len(ck)].copy() if len(b2.ckpt) >= len(ck) else ck[::-1].copy()
        else:
            ck["x"][:] = ck["x"][::-1].copy()
        bad = CheckpointedBytes(bytes(b), ck, 512)
        for gpu in (1, 2):  # the GPU's segment decoder hands the bitstream back; the host's segments fall back to a sequential decode
            _lib.set_option(0, "gpu_decode", gpu)
            try:
                assert torch.equal(gmc.decompress(bad, am, zb, *t[1:]), yq), (trial, gpu)
                assert gpu == 2 or (_lib.ctx_stat(0, 4), _lib.ctx_stat(0, 5)) == (0, 1)
            finally:
                _lib.set_option(0, "gpu_decode", 0)
    assert torch.equal(gmc.decompress(CheckpointedBytes(bytes(b), b.ckpt, 1024), am, zb, *t[1:]), yq)  # wrong stride: ignored
    assert torch.equal(gmc.decompress(CheckpointedBytes(bytes(b), b.ckpt[:-1], 512), am, zb, *t[1:]), yq)  # wrong count: ignored
    with pytest.raises(RuntimeError):
        gmc.decompress(CheckpointedBytes(bytes(b)[: len(b) // 2 & ~3], b.ckpt, 512), am, zb, *t[1:])


def test_config0_256x256_plumbing(oracle):
    """BASELINE configs[0]: one 256x256 image -> y [1,192,16,16], halves [1,192,16,8]; all three modes."""
    for mode in MODES:
        gmc = GaussianMixtureConditional(K=4, mode=mode)
        for half in range(2):
            y, sg, mu, pi = T.make_latent(1234 + half, M=192, h=16, w=8)
            t = [dv(a) for a in (y, sg, mu, pi)]
            (b, abs_max, zb), yq = gmc.compress(*t)
            sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(y, sg, mu, pi)
            assert b == oracle.encode_gmm(mode, sym, s, m, wt)
            assert np.array_equal(oracle.decode_gmm(mode, b, s, m, wt, am + 1), sym)  # the reference's own decoder agrees
            assert torch.equal(gmc.decompress(b, abs_max, zb, *t[1:]), yq)


@pytest.mark.parametrize("quantizer", ["noise", "weighted_mean_ste"])
def test_latent_codec_contract(oracle, quantizer):
    """GaussianMixtureConditionalLatentCodec.compress/decompress (latent_codecs/gaussian_mixture_conditional.py:127-181):
    chunk(3,1) + softmax over K feed the entropy model in place; the output structure is the reference's."""
    from flashgmm_amd.latent_codecs import GaussianMixtureConditionalLatentCodec

    M, h, w, K = 24, 12, 10, 4
    rng = np.random.default_rng(61)
    y, sg, mu, pi = T.make_latent(61, M=M, h=h, w=w, clamp=False)
    logits = rng.standard_normal((1, K * M, h, w)).astype(np.float32)
    ctx = dv(np.concatenate([sg, mu, logits], axis=1))  # what entropy_parameters would output: [1, 3*K*M, h, w]
    codec = GaussianMixtureConditionalLatentCodec(K=K, quantizer=quantizer, mode="polya")
    out = codec.compress(dv(y), ctx)
    assert set(out) == {"strings", "shape", "y_hat"} and tuple(out["shape"]) == (h, w)
    (b, abs_max, zb), = out["strings"]
    # independent recomputation of what must have been coded
    s_t, m_t, l_t = ctx.chunk(3, 1)
    w_t = torch.softmax(l_t.reshape(1, K, M, h, w), dim=1).reshape(1, K * M, h, w)
    y_t = dv(y)
    if quantizer == "weighted_mean_ste":
        ws = (m_t.view(1, K, M, h, w) * w_t.view(1, K, M, h, w)).sum(1)
        y_t = torch.round(y_t - ws)
        m_t = (m_t.view(1, K, M, h, w) - ws.unsqueeze(1)).reshape(1, K * M, h, w)
    sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(y_t.cpu().numpy(), s_t.cpu().numpy(), m_t.cpu().numpy(), w_t.cpu().numpy())
    assert b == oracle.encode_gmm("polya", sym, s, m, wt) and abs_max == am
    dec = codec.decompress(out["strings"], out["shape"], ctx)
    want = out["y_hat"] if quantizer == "noise" else out["y_hat"] + ws
    assert torch.equal(dec["y_hat"], want)


# ---------------------------------------------------------------------------------------------------------------
# the codecs above the entropy model
# ---------------------------------------------------------------------------------------------------------------
def _ref_unembed(y, parity):  # restatement of checkerboard.py:333-354 with torch slicing
    n, c, h, w = y.shape
    y_ = y.new_zeros((2, n, c, h, w // 2))
    a, b = (0, 1) if parity == "even" else (1, 0)
    y_[0, ..., 0::2, :] = y[..., 0::2, a::2]
    y_[0, ..., 1::2, :] = y[..., 1::2, b::2]
    y_[1, ..., 0::2, :] = y[..., 0::2, b::2]
    y_[1, ..., 1::2, :] = y[..., 1::2, a::2]
    return y_


@pytest.mark.parametrize("parity", ["even", "odd"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.int32])
def test_checkerboard_split_merge_equals_reference_slicing(parity, dtype):
    from flashgmm_amd.ops import ckbd_embed, ckbd_unembed

    g = torch.Generator().manual_seed(5)
    for shape in [(1, 6, 8, 12), (1, 3, 7, 10), (2, 5, 1, 2), (1, 192, 32, 48), (1, 1, 33, 258), (1, 4, 0, 6)]:
        y = (torch.randn(shape, generator=g) * 50).to(dtype).to("cuda")
        want = _ref_unembed(y, parity)
        got = ckbd_unembed(y, parity)
        assert got.shape == want.shape and torch.equal(got, want), (shape, "unembed")
        back = ckbd_embed(got, parity)
        assert back.shape == y.shape and torch.equal(back, y), (shape, "embed")
    # -0.0, NaN payloads and infinities are data, not numbers, to a copy kernel
    bits = torch.tensor([0x80000000, 0x7FC12345, 0x7F800000, 0xFF800001], dtype=torch.int64).to(torch.int32)
    y = bits.view(torch.float32).reshape(1, 1, 2, 2).cuda()
    assert torch.equal(ckbd_embed(ckbd_unembed(y, parity), parity).view(torch.int32), y.view(torch.int32))
    with pytest.raises(RuntimeError):
        ckbd_unembed(torch.zeros(1, 2, 4, 5, device="cuda"), parity)  # odd width
    with pytest.raises(RuntimeError):
        ckbd_unembed(torch.zeros(1, 2, 4, 6), parity)  # not on the GPU: no CPU fallback


@pytest.mark.parametrize("mode", MODES)
def test_g7_codecs_equal_the_reference_classes(mode):
    """CheckerboardLatentCodec / ChannelGroupsLatentCodec (checkerboard.py:275-330, channel_groups.py:111-158): the
    reference's own classes were run on device-independent networks (tests/synth.py: exact_modules) when
    tests/golden/g7_codecs.json was made; the mirrors must give the same strings, side information and tensors —
    including the reference's -0.0 in y_hat and its encoder/decoder mismatch for quantizer="weighted_mean_ste"
    (compress() feeds the un-recentred residual to the non-anchor context, decompress() the re-centred value)."""
    import importlib.util

    from flashgmm_amd.latent_codecs import (ChannelGroupsLatentCodec, CheckerboardLatentCodec,
                                            GaussianMixtureConditionalLatentCodec)

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)  # the case tables and the wiring helper only: nothing of the reference is touched
    gold = json.load(open(os.path.join(GOLD, "g7_codecs.json")))[mode]
    Ctx, Par = T.exact_modules()
    Gmm = lambda K, quantizer: GaussianMixtureConditionalLatentCodec(K=K, quantizer=quantizer, mode=mode)  # noqa: E731
    sha = lambda t: hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()  # noqa: E731
    for kind, cfgs in (("ckbd", mg.G7_CKBD), ("groups", mg.G7_GROUPS)):
        for cfg in cfgs:
            name, seed = cfg[0], cfg[1]
            if kind == "ckbd":
                y, side = T.exact_codec_inputs(seed, cfg[2], cfg[3], cfg[4], cfg[5], dead=cfg[6])
            else:
                y, side = T.exact_codec_inputs(seed, sum(cfg[2]), cfg[3], cfg[4], cfg[5])
            codec = mg.build_codecs(CheckerboardLatentCodec, ChannelGroupsLatentCodec, Gmm, Ctx, Par, kind, cfg).cuda()
            enc = codec.compress(dv(y), dv(side))
            want = gold[name]
            got = [{"hex": b.hex(), "abs_max": int(a), "zero_bitmap": [int(v) for v in zb.tolist()]} for (b, a, zb) in enc["strings"]]
            assert got == want["strings"], (name, "strings")
            shape = [list(s_) for s_ in enc["shape"]] if kind == "groups" else list(enc["shape"])
            assert shape == want["shape"], name
            assert sha(enc["y_hat"]) == want["y_hat_sha256"], (name, "compress y_hat")
            dec = codec.decompress(enc["strings"], enc["shape"], dv(side))
            assert sha(dec["y_hat"]) == want["decompress_y_hat_sha256"], (name, "decompress y_hat")
            if cfg[-1] == "noise" or cfg[-2] == "noise":
                assert torch.equal(dec["y_hat"], enc["y_hat"]), name  # (-0.0 == 0.0)
            # through the byte container (flashgmm_amd/container.py) and back: same reconstruction
            from flashgmm_amd import container as Cn

            blob = Cn.pack(enc["strings"], enc["shape"])
            s2, shape2 = Cn.unpack(blob, device="cuda")
            assert torch.equal(codec.decompress(s2, shape2, dv(side))["y_hat"], dec["y_hat"]), (name, "container")
            assert len(blob) - sum(len(t[0]) for t in enc["strings"]) == Cn.side_info_bytes(enc["strings"], enc["shape"])


@pytest.mark.parametrize("mode", MODES)
def test_g8_hyperprior_result_equals_the_reference_classes(mode):
    """golden G8: the COMPLETE nested result of a hyperprior model around the GMM path — HyperpriorLatentCodec{y:
    CheckerboardLatentCodec(GMM), hyper: HyperLatentCodec(EntropyBottleneck)} of the reference's own classes, run on
    device-independent networks in the build container — from GPU tensors: strings [anchor, non-anchor, z], shape
    {"y", "hyper"}, y_hat; and back."""
    from flashgmm_amd import EntropyBottleneckCoder
    from flashgmm_amd.latent_codecs import (CheckerboardLatentCodec, GaussianMixtureConditionalLatentCodec, HyperLatentCodec,
                                            HyperpriorLatentCodec)

    mg = _mg()
    g8 = json.load(open(os.path.join(GOLD, "g8_hyperprior.json")))[mode]
    Ctx, Par = T.exact_modules()
    Ha, Hs = T.exact_hyper_modules()
    for name, seed, c, cz, c_side, h, w, quantizer in mg.G8_HYPER:
        ent = g8[name]
        t = ent["tables"]
        med = torch.from_numpy(np.array(t["medians_bits"], np.uint32).view(np.float32).copy())
        coder = EntropyBottleneckCoder(torch.tensor(t["quantized_cdf"], dtype=torch.int32), torch.tensor(t["cdf_length"], dtype=torch.int32),
                                       torch.tensor(t["offset"], dtype=torch.int32), med).to(DEV)
        ycodec = CheckerboardLatentCodec(latent_codec={"y": GaussianMixtureConditionalLatentCodec(K=4, quantizer=quantizer, mode=mode)},
                                         context_prediction=Ctx(c, 2 * c), entropy_parameters=Par(2 * c + c_side, c))
        codec = HyperpriorLatentCodec(latent_codec={"y": ycodec, "hyper": HyperLatentCodec(entropy_bottleneck=coder, h_a=Ha(c, cz), h_s=Hs(cz, c_side))})
        y, _ = T.exact_codec_inputs(seed, c, c_side, h, w)
        enc = codec.compress(dv(y))
        *ys_, zs_ = enc["strings"]
        hashed = name in mg.G8_HASHED
        assert [mg.bytes_to_json(b, hashed) for b in zs_] == ent["z_strings"]
        assert mg.strings_to_json([(b, a, zb.cpu()) for b, a, zb in ys_], hashed) == ent["y_strings"]
        assert list(enc["shape"]["y"]) == ent["shape"]["y"] and list(enc["shape"]["hyper"]) == ent["shape"]["hyper"]
        assert enc["y_hat"].is_cuda
        assert hashlib.sha256(enc["y_hat"].contiguous().cpu().numpy().tobytes()).hexdigest() == ent["y_hat_sha256"]
        dec = codec.decompress(enc["strings"], enc["shape"])
        assert hashlib.sha256(dec["y_hat"].contiguous().cpu().numpy().tobytes()).hexdigest() == ent["decompress_y_hat_sha256"]
        # the container carries the whole nested result
        from flashgmm_amd import container as Cn

        blob = Cn.pack(enc["strings"], {"y": tuple(enc["shape"]["y"]), "hyper": tuple(enc["shape"]["hyper"])})
        s2, shape2 = Cn.unpack(blob)
        dec2 = codec.decompress([(b, a, zb.to(DEV)) if isinstance(zb, torch.Tensor) else (b, a, zb) for (b, a, zb) in s2[:-1]] + [s2[-1]],
                                shape2)
        assert torch.equal(dec2["y_hat"], dec["y_hat"])


def test_rccl_collectives_single_rank():
    """the collectives of flashgmm_amd.parallel through RCCL with GPU buffers (their multi-rank logic: test_parallel_cpu.py)"""
    import socket
    import subprocess
    import sys

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "nccl_worker.py"), str(port)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("mode", MODES)
def test_gpu_segment_decoder_on_a_corrupt_last_segment_equals_the_sequential_decoder(mode):
    """The LAST segment of a checkpointed bitstream is verified against no note.  With every note valid and only the words
    of the last segment corrupted, the GPU's segment decoder, the host's segments and the plain sequential decode must still
    agree symbol for symbol.  Weights that sum to 0.9 put T_sat near 58 981: a corrupt stream then asks, one time in ten, for
    a cf in [T_sat, 0xFFFF) that no interval of a saturated window holds - the case the straight path of segdec_kernel must
    hand back to the table path (which replays the reference's bisection) instead of walking on."""
    from flashgmm_amd import CheckpointedBytes

    rng = np.random.default_rng(5)
    for seed, (M, h, w), stride in ((431, (24, 16, 12), 256), (432, (2, 16, 12), 256), (433, (24, 16, 12), 1024)):
        y, sg, mu, pi = T.make_latent(seed, M=M, h=h, w=w)
        pi = (pi * np.float32(0.9)).astype(np.float32)
        t = [dv(a) for a in (y, sg, mu, pi)]
        plain = GaussianMixtureConditional(K=4, mode=mode)
        ck = GaussianMixtureConditional(K=4, mode=mode, checkpoint_stride=stride)
        (b0, am, zb), yq = plain.compress(*t)
        (b1, _, _), _ = ck.compress(*t)
        assert bytes(b1) == b0 and len(b1.ckpt) >= 1
        if seed == 432:
            assert len(b1.ckpt) == 1  # n_ckpt == 1: two segments, the second unverified
        first = 8 + 4 * int(b1.ckpt["pos"][-1])  # the words the decoder has not read when it stands at the last note
        assert first < len(b0)
        for trial in range(4):
            bad = bytearray(b0)
            k0 = first if trial < 2 else int(rng.integers(first, len(b0) - 3)) & ~3
            bad[k0:] = bytes(rng.integers(0, 256, len(b0) - k0, dtype=np.uint8))
            bad = bytes(bad)
            results = []
            for how, stream in ((None, bad), (1, CheckpointedBytes(bad, b1.ckpt, stride)), (2, CheckpointedBytes(bad, b1.ckpt, stride))):
                if how is not None:
                    _lib.set_option(0, "gpu_decode", how)
                try:
                    results.append((plain if how is None else ck).decompress(stream, am, zb, *t[1:]).cpu().numpy())
                except RuntimeError:  # (a corrupt stream may also run out of words: then it must for every decoder)
                    results.append(None)
                finally:
                    _lib.set_option(0, "gpu_decode", 0)
            want = results[0]
            for got in results[1:]:
                assert (want is None and got is None) or (want is not None and got is not None and np.array_equal(got, want)), (seed, trial)


def _bench_line_and_detail(r, detail_path):
    """bench.py's stdout is ONE summary line under 4 KB (what the driver parses); the whole result lies in the detail file"""
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, r.stdout[-2000:]
    line, full = json.loads(lines[0]), json.load(open(detail_path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "data"):
        assert line[k] == full[k]
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["detail"] == str(detail_path)
    return line, full


def test_bench_two_ranks_rehearsed_on_one_device(tmp_path):
    """BASELINE configs[3]'s code path in the one form a 1-GPU box allows (FGMM_BENCH_ONE_DEVICE): `bench.py --gpus 2` starts two
    fresh rank processes before anything touches the GPU (never a re-exec), both code on GPU 0, the process group is gloo, the
    per-step all-gather of bitstream lengths and one gather of the containers run, every rank checks its results, and rank 0
    prints one line that carries the max-over-ranks legs (upper_bound, checkpointed).  A rehearsal, not a scaling point."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FGMM_BENCH_ONE_DEVICE="1", FGMM_BENCH_DETAIL=str(tmp_path / "detail.json"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "FGMM_BENCH_DRYRUN"):
        env.pop(k, None)
    torch.cuda.synchronize()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--images", "4", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--launch-timeout", "400"], env=env, capture_output=True, text=True, timeout=480)
    assert r.returncode == 0, r.stderr[-3000:]
    line, d = _bench_line_and_detail(r, tmp_path / "detail.json")
    assert line["n_gpus"] == 2 and line["ranks"]["backend"] == "gloo" and len(line["ranks"]["ms_per_step"]) == 2 and line["checkpointed"]["value"] > 0
    assert d["n_gpus"] == 2 and d["config"]["one_device_rehearsal"] is True and d["config"]["images_per_gpu"] == 4
    rk = d["ranks"]
    assert rk["backend"] == "gloo" and rk["rccl_ranks"] == 0 and len(rk["ms_per_step"]) == 2 and rk["result_checked_ranks"] == 2
    budget = d["config"]["host_cpu_budget"]
    from helpers import expected_threads
    assert rk["host_threads_per_gpu"] == [expected_threads(budget, 2)] * 2
    ag = rk["allgather_ms"]  # the overlapped exchange: issued after the encode call, waited for at the end of the step
    assert ag["total_in_flight"] > ag["exposed"] >= 0 and ag["issue"] > 0 and rk["allgather_payload_ms"] > 0
    assert d["ms_per_step"] == max(rk["ms_per_step"]) and d["value"] > 0
    assert d["upper_bound"]["value"] > 0 and d["checkpointed"]["value"] > 0 and d["checkpointed"]["bitstreams_handed_back_last_call"] == 0
    assert "latency_ms" not in d and "modes" not in d  # (N = 1 legs)


def test_table_kernel_blocks_are_placed_by_a_cursor(oracle):
    """tab_kernel places a block's rows by ONE atomic add per block on the launch's cursor: blocks lie in arrival order - any
    order -, the offsets are a permutation of the running sums of the block sizes, and the table is the oracle's whatever the
    order.  (Round 4's decoupled look-back placement - launch order, a third slower - is in the git history.)"""
    y, sg, mu, pi = T.make_latent(35, M=64, h=16, w=12)
    sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
    max_bs = abs_max + 1
    want = oracle.cdftab("polya", s, m, w, max_bs)
    h0, bo0, rows0, u0, tl = gpu_tab("polya", s, m, w, max_bs)
    assert np.array_equal(expand_trimmed(h0, rows0, max_bs, bo0, tl), want)
    o = np.sort(bo0.astype(np.int64)) * 4
    assert len(bo0) > 50 and o[0] == 0 and len(np.unique(o)) == len(o) and o[-1] < u0  # every block has a place of its own
    h1, bo1, rows1, u1, _ = gpu_tab("polya", s, m, w, max_bs)
    assert u1 == u0 and np.array_equal(h1, h0) and np.array_equal(expand_trimmed(h1, rows1, max_bs, bo1, tl), want)
    assert np.array_equal(np.sort(np.diff(np.append(o, u0))), np.sort(np.diff(np.append(np.sort(bo1.astype(np.int64)) * 4, u1))))  # the same blocks


def test_bench_latents_dir_hook(tmp_path):
    """bench.py --latents-dir: saved (y, scales, means, weights) tensors of real images instead of the synthetic batch - the hook for
    repeating the reference's own measurement (eval_ckbd.py:113-143) when a checkpoint and images are supplied.  Here: two small
    "images" of two bitstreams each, the line must say so and its results must have been checked (decode == round(y))."""
    import subprocess
    import sys

    for i in range(2):
        img = []
        for j in range(2):
            y, sg, mu, pi = T.make_latent(900 + 10 * i + j, M=48, h=16, w=12)
            img.append({"y": torch.from_numpy(y), "scales": torch.from_numpy(sg), "means": torch.from_numpy(mu), "weights": torch.from_numpy(pi)})
        img.append({"pixels": 128 * 192})
        torch.save(img, tmp_path / f"image_{i:02d}.pt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--latents-dir", str(tmp_path), "--steps", "2", "--warmup", "1", "--no-extras",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=dict(os.environ, FGMM_BENCH_DETAIL=str(tmp_path / "detail.json")))
    assert r.returncode == 0, r.stderr[-3000:]
    line, d = _bench_line_and_detail(r, tmp_path / "detail.json")
    assert line["data"] == "real-latents" and line["step_ms"]["between_calls"] >= 0
    assert d["data"] == "real-latents" and d["config"]["images_per_gpu"] == 2 and d["config"]["streams_per_gpu"] == 4 and d["value"] > 0
    assert d["config"]["coded_symbols_per_gpu"] > 0 and d["ranks"]["result_checked_ranks"] == 1
    # the region's time by phase, from the library's call log: one encode call and two decode calls (two bitstreams per image) per step,
    # and the phases account for the steps (what is left is the Python between the calls)
    ph = d["step_ms"]["phases_ms"]
    assert ph["steps"] == 2 and {"call0_encode.bus", "call1_decode.head", "call2_decode.host_tail", "between_calls"} <= set(ph)
    in_calls = sum(v for k, v in ph.items() if k.endswith((".head", ".bus", ".host_tail", ".end")))
    assert 0 <= ph["between_calls"] and in_calls < d["step_ms"]["max"] * 1.05


def test_a_failed_call_leaves_nothing_in_flight(ctx_options):
    """A bitstream that fails early (truncated input) finishes its item while the copies of its LATER table pieces are still queued;
    the call must not return before they have drained - the next call hands the same staging area to its kernels and the same pinned
    ranges to its copies (found by ThreadSanitizer on the fake device, round 5: scripts/tsan_host.sh).  Here: a one-bitstream call
    that fails at its first piece with many pieces behind it, straight into a good call, many times over."""
    ctx_options(pieces=16)
    gmc = GaussianMixtureConditional(K=4, mode="polya")
    y, sg, mu, pi = T.make_latent(7100)
    t = [dv(a) for a in (y, sg, mu, pi)]
    (b, abs_max, zb), yq = gmc.compress(*t)
    others = [[dv(a) for a in T.make_latent(7101 + i, M=96, h=32, w=24)] for i in range(3)]
    res = gmc.compress_batch(*[[o[k] for o in others] for k in range(4)])
    good = ([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], *[[o[k] for o in others] for k in (1, 2, 3)])
    for rep in range(12):
        with pytest.raises(RuntimeError):
            gmc.decompress(bytes(b[:16]), abs_max, zb, *t[1:])
        out = gmc.decompress_batch(*good)
        assert all(torch.equal(o, r[1]) for o, r in zip(out, res)), rep
    assert torch.equal(gmc.decompress(b, abs_max, zb, *t[1:]), yq)


def test_an_encode_call_drains_its_segment_copies(ctx_options):
    """Segmented encode tables (tails first): an encoder waits only for the segments it enters, and a bitstream without a coded symbol
    enters none - the call must still not return before its last table copy has landed, or that copy writes into the pinned workspace
    the NEXT call is filling (found under AddressSanitizer on the fake device, round 5: the descriptors of a following decode call
    overwritten).  Here: segmentation forced (enc_segs 2) on batches whose items are all dead, each straight into a decode call."""
    ctx_options(enc_segs=2, pieces=10)
    gmc = GaussianMixtureConditional(K=4, mode="polya")
    dead = [[dv(a) for a in T.make_latent(7300 + i, M=M, h=h, w=w, zero_frac=1.0)] for i, (M, h, w) in enumerate([(9, 16, 24), (17, 1, 1), (24, 8, 8)])]
    live = [[dv(a) for a in T.make_latent(7310 + i, M=48, h=16, w=12)] for i in range(4)]
    res_live = gmc.compress_batch(*[[o[k] for o in live] for k in range(4)])
    good = ([r[0][0] for r in res_live], [r[0][1] for r in res_live], [r[0][2] for r in res_live], *[[o[k] for o in live] for k in (1, 2, 3)])
    for rep in range(10):
        res = gmc.compress_batch(*[[o[k] for o in dead] for k in range(4)])
        assert all(bytes(r[0][0]) == bytes.fromhex("0000008000000000") and int(r[0][2].sum()) == 0 for r in res), rep
        out = gmc.decompress_batch(*good)
        assert all(torch.equal(o, r[1]) for o, r in zip(out, res_live)), rep
        out = gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], *[[o[k] for o in dead] for k in (1, 2, 3)])
        assert all(torch.equal(o, r[1]) and not o.any() for o, r in zip(out, res)), rep


def test_scheduling_options_change_no_byte(oracle, ctx_options):
    """How a call's tables cross PCIe is a matter of scheduling only: encode tables whole (enc_segs 0) or in four segments per bitstream,
    tails first (1, the default: the encoders follow the landing); symbols back to the GPU round by round or bitstream by bitstream.
    Every combination gives the
    oracle's bytes and the encoder's reconstruction - on a batch large enough for the segmented layout (>= 4 MB of tables), with
    an all-zero item and a channel count that is not a multiple of four among them."""
    specs = [(192, 32, 24, 0.1)] * 6 + [(190, 32, 24, 0.3), (192, 32, 24, 1.0), (192, 32, 24, 0.0)]
    lat = [T.make_latent(4400 + i, M=M, h=h, w=w, zero_frac=zf) for i, (M, h, w, zf) in enumerate(specs)]
    dev = [[dv(a) for a in l] for l in lat]
    ys, ss, ms, ws = ([d[k] for d in dev] for k in range(4))
    gmc = GaussianMixtureConditional(K=4, mode="polya")
    want = []
    for y, sg, mu, pi in lat[:3] + lat[6:]:
        sym, s, m, wt, am, zbm, yq = T.to_coder_inputs(y, sg, mu, pi)
        want.append(oracle.encode_gmm("polya", sym, s, m, wt))
    results = {}
    for segs in (0, 1):
        ctx_options(enc_segs=segs)
        res = gmc.compress_batch(ys, ss, ms, ws)
        results[segs] = [bytes(r[0][0]) for r in res]
        assert [results[segs][i] for i in (0, 1, 2, 6, 7, 8)] == want, segs
    assert results[0] == results[1]
    args = ([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
    for rounds in (1, 0):
        ctx_options(scatter_rounds=rounds)
        out = gmc.decompress_batch(*args)
        assert all(torch.equal(o, r[1]) for o, r in zip(out, res)), rounds


def test_round_scatter_redoes_an_item_with_symbols_beyond_int16(oracle, ctx_options):
    """The decoded symbols return to the GPU round by round as int16 (option scatter_rounds).  A bypass-coded symbol beyond int16
    can only reach that path when the caller's abs_max is smaller than the stream's symbols (a hostile or corrupt container: an
    honest half-width of 32768 takes the generic kernels) - the bypass nibbles carry it whatever the half-width
    (rans_interface.cpp:808-824).  Such an item is scattered once more, whole and wide, after the rounds have read: the result
    is the oracle decoder's for that (stream, half-width), in a multi-piece batch beside ordinary items."""
    lat = [T.make_latent(4700 + i, M=192, h=32, w=24, zero_frac=0.1) for i in range(3)]
    y1 = lat[1][0].copy()
    clean_max = T.to_coder_inputs(*lat[1])[4]  # the honest abs_max of the latent without its outliers
    nzc = [c for c in range(192) if np.round(y1[0, c]).any()]  # coded channels: the outliers do not change the zero bitmap
    y1[0, nzc[3], 5, 7], y1[0, nzc[40], 0, 0], y1[0, nzc[-1], 31, 23] = 70000.4, -40000.0, 32768.0
    lat[1] = (y1,) + tuple(lat[1][1:])
    dev = [[dv(a) for a in l] for l in lat]
    ys, ss, ms, ws = ([d[k] for d in dev] for k in range(4))
    gmc = GaussianMixtureConditional(K=4, mode="polya")
    res = gmc.compress_batch(ys, ss, ms, ws)
    sym, s, m, wt, am, zbm, yq = T.to_coder_inputs(*lat[1])
    assert res[1][0][1] == am > 70000 > 32767 > clean_max and res[1][0][0] == oracle.encode_gmm("polya", sym, s, m, wt)
    assert np.array_equal(oracle.decode_gmm("polya", res[1][0][0], s, m, wt, clean_max + 1), sym)  # the small half-width decodes it too
    abs_maxes = [res[0][0][1], clean_max, res[2][0][1]]
    for rounds in (1, 0):
        ctx_options(scatter_rounds=rounds)
        out = gmc.decompress_batch([r[0][0] for r in res], abs_maxes, [r[0][2] for r in res], ss, ms, ws)
        assert all(torch.equal(o, r[1]) for o, r in zip(out, res)), rounds
        assert float(out[1].abs().max()) == 70000.0


def test_stacked_results_are_the_list_forms_sequence(oracle):
    """compress_batch on stacked inputs returns a CompressedBatch: the same sequence of ((bytes, abs_max, zero_bitmap), y_q) as the
    list form - by index, slice and iteration - held stacked; its fields go back into decompress_batch as they are (a stage = a
    stride), and stacked_output gives the one tensor the per-item outputs are views of."""
    from flashgmm_amd import CompressedBatch

    lat = [T.make_latent(5100 + i, M=24, h=8, w=12, zero_frac=0.25) for i in range(6)]
    ys, ss, ms, ws = (torch.cat([dv(l[k]) for l in lat]) for k in range(4))
    gmc = GaussianMixtureConditional(K=4, mode="as")
    stacked = gmc.compress_batch(ys, ss, ms, ws)
    listed = gmc.compress_batch([ys[i:i + 1] for i in range(6)], [ss[i:i + 1] for i in range(6)], [ms[i:i + 1] for i in range(6)],
                                [ws[i:i + 1] for i in range(6)])
    assert isinstance(stacked, CompressedBatch) and isinstance(listed, list) and len(stacked) == 6
    for i, (((b, am, zb), q), ((b2, am2, zb2), q2)) in enumerate(zip(stacked, listed)):
        sym, s, m, wt, am_o, zbm, yq = T.to_coder_inputs(*lat[i])
        assert b == b2 == oracle.encode_gmm("as", sym, s, m, wt) and am == am2 == am_o and torch.equal(zb, zb2) and torch.equal(q, q2)
    assert [x[0][0] for x in stacked[1:5:2]] == [stacked.strings[1], stacked.strings[3]] and stacked[-1][0][1] == stacked.abs_maxes[5]
    with pytest.raises(IndexError):
        stacked[6]
    for s in range(2):  # a codec stage: every second bitstream, fields in, one tensor out
        out = gmc.decompress_batch(stacked.strings[s::2], stacked.abs_maxes[s::2], stacked.zero_bitmaps[s::2], ss[s::2], ms[s::2], ws[s::2],
                                   stacked_output=True)
        assert out.shape == (3, 1, 24, 8, 12) and torch.equal(out, stacked.y_q[s::2])
        views = gmc.decompress_batch(stacked.strings[s::2], stacked.abs_maxes[s::2], stacked.zero_bitmaps[s::2], ss[s::2], ms[s::2], ws[s::2])
        assert isinstance(views, list) and all(torch.equal(v, o) for v, o in zip(views, out))
    with pytest.raises(RuntimeError):
        gmc.decompress_batch([stacked.strings[0]], [stacked.abs_maxes[0]], [stacked.zero_bitmaps[0]], [ss[:1]], [ms[:1]], [ws[:1]], stacked_output=True)
    empty = gmc.compress_batch(ys[:0], ss[:0], ms[:0], ws[:0])
    assert len(empty) == 0 and list(empty) == []
    assert gmc.decompress_batch([], [], empty.zero_bitmaps, ss[:0], ms[:0], ws[:0], stacked_output=True).shape == (0, 1, 24, 8, 12)


def test_workers_are_kept_off_the_creating_threads_l3():
    """The host workers stream the decode-side tables through the L3 of the core complex they run on; the library keeps them off the
    L3 of the thread that created the context (include/flashgmm_amd.h: fgmm_ctx_worker_cpus; profiles/r05_l3_ab.txt) when that leaves
    them at least 32 CPUs and two per worker.  The reported cpulist must be the process's CPUs minus exactly one L3 domain, and every
    worker thread must be confined to it."""
    from flashgmm_amd import parallel as P

    if os.environ.get("FGMM_WORKER_CPUS"):
        pytest.skip("FGMM_WORKER_CPUS is set")
    _lib.ctx(0)
    wc = _lib.worker_cpus(0)
    mask = os.sched_getaffinity(0)
    n_workers = _lib.lib().fgmm_ctx_threads(_lib.ctx(0))
    if not wc:
        assert len(mask) - 16 < max(32, 2 * n_workers), f"{len(mask)} CPUs, {n_workers} workers: an L3 could have been set aside"
        return
    workers = P.cpulist_to_set(wc)
    free = mask - workers
    assert workers <= mask and free and len(workers) >= max(32, 2 * n_workers)
    l3 = P.cpulist_to_set(open(f"/sys/devices/system/cpu/cpu{min(free)}/cache/index3/shared_cpu_list").read()) & mask
    assert free == l3, (sorted(free), sorted(l3))
    seen = 0
    for tid in os.listdir("/proc/self/task"):
        try:
            if not open(f"/proc/self/task/{tid}/comm").read().startswith("fgmm-w"):
                continue
            allowed = [ln.split(":", 1)[1].strip() for ln in open(f"/proc/self/task/{tid}/status") if ln.startswith("Cpus_allowed_list")][0]
        except OSError:
            continue
        # (a process may hold several contexts - device 0 and "the current device" - each with its own decision)
        off = mask - P.cpulist_to_set(allowed)
        assert off and off == P.cpulist_to_set(open(f"/sys/devices/system/cpu/cpu{min(off)}/cache/index3/shared_cpu_list").read()) & mask, (tid, allowed)
        seen += P.cpulist_to_set(allowed) == workers
    assert seen >= n_workers
    # the decode calls' pool (threads fgmm-d*): as many workers, on ONE hardware thread of every core of that list (when the host has
    # second hardware threads and enough cores)
    dec = []
    for tid in os.listdir("/proc/self/task"):
        try:
            if open(f"/proc/self/task/{tid}/comm").read().startswith("fgmm-d"):
                dec.append(P.cpulist_to_set([ln.split(":", 1)[1].strip() for ln in open(f"/proc/self/task/{tid}/status") if ln.startswith("Cpus_allowed_list")][0]))
        except OSError:
            continue

    def siblings(c):
        return P.cpulist_to_set(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read())

    firsts = {c for c in workers if c == min(siblings(c) & workers)}
    if os.environ.get("FGMM_DECODE_SMT") == "1" or len(firsts) == len(workers) or len(firsts) < n_workers:
        assert not dec
    else:
        # (a process may hold several contexts, each with its own list: this context's decoders are on `firsts`, every decoder of the
        # process is on one hardware thread per core of the process's CPUs)
        assert len(dec) >= n_workers and any(d == firsts for d in dec)
        assert all(d <= mask and all(len(siblings(c) & d) == 1 for c in d) for d in dec)

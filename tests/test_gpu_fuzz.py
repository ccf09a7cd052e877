"""GPU (-m gpu): randomized shapes / modes / dtypes / layouts against the oracle, concurrency, envelope edges."""
import threading

import numpy as np
import pytest
import torch

from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODES = ["polya", "as", "logistic"]


def dv(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_fuzz_shapes_modes_dtypes(oracle):
    rng = np.random.default_rng(2025)
    for case in range(48):
        M = int(rng.integers(1, 41))
        h, w = int(rng.integers(1, 21)), int(rng.integers(1, 21))
        mode = MODES[case % 3]
        f16 = bool(case % 4 == 3)
        clamp = bool(case % 5 != 4)
        y, sg, mu, pi = T.make_latent(9000 + case, M=M, h=h, w=w, clamp=False, zero_frac=float(rng.choice([0, 0.2, 0.6])))
        if case % 7 == 0:  # outliers: far symbols -> bypass escape
            y = y.copy()
            y.reshape(-1)[rng.integers(0, y.size, max(1, y.size // 50))] *= 40
        if not clamp:
            sg = np.maximum(sg, np.float32(0.02))  # raw sigma path: keep it positive (the reference divides by it)
        if f16:
            sg, mu, pi = T.to_float16_planes(sg, mu, pi)
        gmc = GaussianMixtureConditional(K=4, mode=mode, clamp_scales=clamp)
        t = [dv(y)] + [dv(a) for a in (sg, mu, pi)]
        if case % 6 == 1:  # non-contiguous channel stride: every second channel of a twice-as-large tensor
            t[1:] = [torch.stack([x, x.flip(1)], 2).reshape(1, 2 * x.size(1), h, w)[:, ::2] for x in t[1:]]
        (b, abs_max, zb), yq = gmc.compress(*t)
        sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(y, *(a.astype(np.float32) for a in (sg, mu, pi)), clamp=clamp)
        tag = (case, M, h, w, mode, f16, clamp)
        assert abs_max == am and zb.cpu().tolist() == zbm.tolist(), tag
        assert b == oracle.encode_gmm(mode, sym, s, m, wt), tag
        if abs_max + 1 <= 16382:
            assert torch.equal(gmc.decompress(b, abs_max, zb, *t[1:]), yq), tag


def test_fuzz_symtab_grid_forms(oracle):
    """the three launch forms of the encode kernel — linear waves (every hw a multiple of 256), per-channel blocks with
    16-byte loads (hw % 4 == 0), one symbol per lane (odd hw) — on batches, both parameter dtypes, all modes"""
    rng = np.random.default_rng(77)
    case = 0
    for (h, w) in [(16, 16), (16, 32), (32, 24), (32, 32), (40, 32), (13, 20), (2, 257), (64, 64), (1, 256), (9, 7)]:
        for f16 in (False, True):
            mode = MODES[case % 3]
            case += 1
            Ms = [int(rng.integers(1, 25)) for _ in range(3)]
            lat = [T.make_latent(4000 + 10 * case + i, M=M, h=h, w=w, clamp=False, zero_frac=float(rng.choice([0, 0.3])))
                   for i, M in enumerate(Ms)]
            if f16:
                lat = [(y,) + tuple(T.to_float16_planes(sg, mu, pi)) for y, sg, mu, pi in lat]
            gmc = GaussianMixtureConditional(K=4, mode=mode)
            ys, ss, ms, ws = ([dv(l[k]) for l in lat] for k in range(4))
            res = gmc.compress_batch(ys, ss, ms, ws)
            for i, l in enumerate(lat):
                sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(l[0], *(a.astype(np.float32) for a in l[1:]))
                (b, abs_max, zb), yq = res[i]
                tag = (h, w, Ms[i], mode, f16)
                assert b == oracle.encode_gmm(mode, sym, s, m, wt) and abs_max == am and zb.tolist() == zbm.tolist(), tag
            outs = gmc.decompress_batch([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
            for i in range(3):
                assert torch.equal(outs[i], res[i][1]), (h, w, Ms[i], mode, f16)


def test_concurrent_callers_share_one_context(oracle):
    """four Python threads compress / decompress on one GPU at once: calls serialise on the context, results stay right"""
    cases = []
    for seed in range(4):
        y, sg, mu, pi = T.make_latent(700 + seed, M=16, h=12, w=8)
        sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(y, sg, mu, pi)
        cases.append(([dv(a) for a in (y, sg, mu, pi)], oracle.encode_gmm("polya", sym, s, m, wt), am))
    errors = []

    def worker(k):
        try:
            gmc = GaussianMixtureConditional(K=4, mode="polya")
            t, want, am = cases[k]
            for _ in range(25):
                (b, abs_max, zb), yq = gmc.compress(*t)
                assert b == want and abs_max == am
                assert torch.equal(gmc.decompress(b, abs_max, zb, *t[1:]), yq)
        except Exception as e:  # pragma: no cover
            errors.append((k, repr(e)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errors, errors


def test_decoder_half_width_envelope(oracle):
    """max_bs_value = abs_max + 1: 2-byte headers up to 126, 4-byte up to 16382, 8-byte beyond (the reference bisects any
    half-width, rans_interface.cpp:826-854); only a latent whose evaluation window exceeds 2^20 edges is refused, loudly"""
    M, h, w = 2, 2, 2
    y, sg, mu, pi = T.make_latent(5, M=M, h=h, w=w)
    gmc = GaussianMixtureConditional(K=4, mode="as")
    for peak in (125.6, 126.6, 254.4, 255.6, 16380.6, 16381.6, 1.0e6, -3.0e8, 1073741000.0):
        y = y.copy()
        y[0, 0, 0, 0] = peak
        t = [dv(a) for a in (y, sg, mu, pi)]
        (b, abs_max, zb), yq = gmc.compress(*t)
        sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(y, sg, mu, pi)
        assert abs_max == am == int(abs(np.float32(peak))) + 1
        assert b == oracle.encode_gmm("as", sym, s, m, wt)
        assert torch.equal(gmc.decompress(b, abs_max, zb, *t[1:]), yq), peak
    # a component 3e6 away from the others: the window between the saturated tails is ~3e6 edges long -> refused
    mu2 = mu.copy()
    mu2[0, 0, 0, 0] = 3.0e6
    y[0, 0, 0, 0] = 4.0e6
    t = [dv(a) for a in (y, sg, mu2, pi)]
    (b, abs_max, zb), yq = gmc.compress(*t)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        gmc.decompress(b, abs_max, zb, *t[1:])
    y[0, 0, 0, 0] = 2.0e9  # abs_max beyond FGMM_MAX_BS
    (b, abs_max, zb), yq = gmc.compress(dv(y), *t[1:])
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        gmc.decompress(b, abs_max, zb, *t[1:])


def test_empty_and_degenerate_batches():
    gmc = GaussianMixtureConditional(K=4, mode="polya")
    assert gmc.compress_batch([], [], [], []) == [] and gmc.decompress_batch([], [], [], [], [], []) == []
    z = torch.zeros(1, 3, 2, 2, device=DEV)
    p = torch.ones(1, 12, 2, 2, device=DEV)
    (b, abs_max, zb), yq = gmc.compress(z, p, p * 0, p / 4)
    assert b == bytes.fromhex("0000008000000000") and abs_max == 1 and zb.tolist() == [0, 0, 0]
    assert torch.equal(gmc.decompress(b, abs_max, zb, p, p * 0, p / 4), z)


def test_staging_overflow_rerun_in_a_batch(oracle):
    """the decode staging area capped at 1 MiB for a batch that needs ~6: every launch unit overflows its share and is
    re-run with the exact size its cursor reports (fgmm_capi.cpp, decode_batch); symbols as without the cap"""
    g = GaussianMixtureConditional(K=4, mode="logistic")
    ts = [[dv(a) for a in T.make_latent(s, M=24, h=16, w=8)] for s in range(20)]
    res = g.compress_batch(*[[t[k] for t in ts] for k in range(4)])
    args = ([r[0][0] for r in res], [r[0][1] for r in res], [r[0][2] for r in res], *[[t[k] for t in ts] for k in (1, 2, 3)])
    saved = _lib.get_option(0, "stage_max_mb")
    try:
        _lib.trim(0)
        _lib.set_option(0, "stage_max_mb", 1)
        out = g.decompress_batch(*args)
    finally:
        _lib.set_option(0, "stage_max_mb", saved)
        _lib.trim(0)
    assert all(torch.equal(o, r[1]) for o, r in zip(out, res))

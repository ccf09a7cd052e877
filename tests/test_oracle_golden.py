"""CPU: the oracle (oracle/fgmm_oracle.c) against the golden vectors captured from the REAL reference
(tests/golden/make_golden.py).  Bit-exact everywhere."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests import synth as T

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MODES = ["polya", "as", "logistic"]


def _g3_cases():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg.g3_cases()


@pytest.mark.parametrize("mode", MODES)
def test_g1_float_cdf_bit_exact(oracle, mode):
    g = np.load(os.path.join(GOLD, "g1_cdf.npz"))
    c1, c2 = oracle.gmm_cdf(mode, g["v"], g["scales"], g["means"], g["weights"])
    # north_star tolerance is 1e-5 on float CDFs; the restatement is in fact bit-exact
    assert np.array_equal(c1.view(np.uint32), g[f"c1_{mode}"])
    assert np.array_equal(c2.view(np.uint32), g[f"c2_{mode}"])
    assert np.abs(c1 - g[f"c1_{mode}"].view(np.float32)).max() <= 1e-5


@pytest.mark.parametrize("mode", MODES)
def test_g2_start_range(oracle, mode):
    g = np.load(os.path.join(GOLD, "g1_cdf.npz"))
    packed = oracle.symtab(mode, g["v"], g["scales"], g["means"], g["weights"])
    rng = (packed >> 16).astype(np.uint16)
    start = (packed & 0xFFFF).astype(np.uint16)
    assert np.array_equal(rng, g[f"range_{mode}"])
    nb = rng != 0
    assert np.array_equal(start[nb], g[f"start_{mode}"][nb])
    # bypass rows carry the low 16 bits of the symbol
    assert np.array_equal(start[~nb], (g["v"][~nb] & 0xFFFF).astype(np.uint16))


@pytest.mark.parametrize("mode", MODES)
def test_g3_small_streams_verbatim(oracle, mode):
    gold = json.load(open(os.path.join(GOLD, "g3_small.json")))["cases"]
    for name, (sym, s, m, w) in _g3_cases().items():
        ent = gold[name]
        assert ent["symbols"] == sym.tolist()
        b = oracle.encode_gmm(mode, sym, s, m, w)
        assert b.hex() == ent[mode]["hex"], name
        d = oracle.decode_gmm(mode, b, s, m, w, ent[mode]["max_bs"])
        assert d.tolist() == ent[mode]["decoded"], name
        # integer-only surfaces: same tables => same bytes / same symbols
        assert oracle.rans_encode_symtab(oracle.symtab(mode, sym, s, m, w), sym) == b
        tab = oracle.cdftab(mode, s, m, w, ent[mode]["max_bs"])
        assert oracle.rans_decode_cdftab(b, tab, ent[mode]["max_bs"]).tolist() == ent[mode]["decoded"]


@pytest.mark.parametrize("mode", MODES)
def test_fullsize_fixture_sample(oracle, mode):
    """tests/golden/fullsize.json (the reference's bytes of every full-size bitstream the GPU tests and bench.py code; its
    generator already held the oracle to ALL of them): here a sample - every eighth Kodak half, and the first ELIC group."""
    gold = json.load(open(os.path.join(GOLD, "fullsize.json")))[mode]
    assert len(gold["kodak24"]) == 48 and all(e["roundtrip"] for st in gold.values() for e in st.values())
    for seed in range(5, 48, 8):
        sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(*T.make_latent(seed))
        b = oracle.encode_gmm(mode, sym, s, m, w)
        assert (len(b), hashlib.md5(b).hexdigest(), abs_max) == (gold["kodak24"][str(seed)]["len"], gold["kodak24"][str(seed)]["md5"],
                                                                 gold["kodak24"][str(seed)]["abs_max"]), seed
    if mode == "polya":
        assert len(gold["elic_groups"]) == 5 and len(gold["elic4k_image0"]) == 10
        sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(*T.make_latent(51, M=16, h=136, w=120, clamp=False))
        b = oracle.encode_gmm(mode, sym, s, m, w)
        assert (len(b), hashlib.md5(b).hexdigest()) == (gold["elic_groups"]["51"]["len"], gold["elic_groups"]["51"]["md5"])


def test_empty_stream_is_8_bytes(oracle):
    sym, s, m, w = _g3_cases()["n0"]
    assert oracle.encode_gmm("polya", sym, s, m, w) == bytes.fromhex("0000008000000000")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("seed", [1234, 0])
def test_ka1_kodak_half(oracle, mode, seed):
    ent = json.load(open(os.path.join(GOLD, "ka1.json")))[mode][str(seed)]
    y, sg, mu, pi = T.make_latent(seed)
    sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
    assert (len(sym), int(zb.sum()), abs_max) == (ent["n"], ent["nz_channels"], ent["abs_max"])
    b = oracle.encode_gmm(mode, sym, s, m, w)
    assert len(b) == ent["len"] and hashlib.md5(b).hexdigest() == ent["md5"]
    tab = oracle.cdftab(mode, s, m, w, abs_max + 1)
    assert np.array_equal(oracle.rans_decode_cdftab(b, tab, abs_max + 1), sym)


def test_ka1_survey_md5s(oracle):
    """SURVEY.md §8c KA-1, literally."""
    want = {"polya": (53164, "e759909d27406fbc0168c33b4509772d"), "as": (52768, "9283e03480f545471e6245b21aa61af5"),
            "logistic": (51784, "ecbe33ac17fe297803909b32db54d1f6")}
    y, sg, mu, pi = T.make_latent(1234)
    sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi)
    assert len(sym) == 125952 and zb.sum() == 164
    for mode, (ln, md5) in want.items():
        b = oracle.encode_gmm(mode, sym, s, m, w)
        assert (len(b), hashlib.md5(b).hexdigest()) == (ln, md5)


@pytest.mark.parametrize("mode", MODES)
def test_g4_api_level_restatement(oracle, mode):
    """numpy restatement of GaussianMixtureConditional.compress's tensor prep (testing.to_coder_inputs) + the
    oracle coder == what the reference's own Python class returned."""
    gold = json.load(open(os.path.join(GOLD, "g4_api.json")))[mode]
    for seed, ent in gold.items():
        y, sg, mu, pi = T.make_latent(int(seed), M=ent["M"], h=ent["h"], w=ent["w"], clamp=False, zero_frac=0.15)
        sym, s, m, w, abs_max, zb, yq = T.to_coder_inputs(y, sg, mu, pi, clamp=True)
        assert abs_max == ent["abs_max"] and zb.tolist() == ent["zero_bitmap"]
        assert hashlib.sha256(yq.tobytes()).hexdigest() == ent["yq_sha256"]
        b = oracle.encode_gmm(mode, sym, s, m, w)
        assert (len(b), hashlib.md5(b).hexdigest()) == (ent["len"], ent["md5"])
        assert ent["decompress_equals_yq"]
        d = oracle.decode_gmm(mode, b, s, m, w, abs_max + 1) if len(sym) < 40000 else \
            oracle.rans_decode_cdftab(b, oracle.cdftab(mode, s, m, w, abs_max + 1), abs_max + 1)
        assert np.array_equal(d, sym)


def test_exp_clamp_edges(oracle):
    """avx_mathfun.h:297-302 builds 2^n as a bit pattern and multiplies: not ldexp at the clamp edges."""
    assert oracle.exp(-1000.0) == oracle.exp(-88.3762626647949)
    assert oracle.exp(0.0) == 1.0
    assert oracle.exp(1000.0) == oracle.exp(88.3762626647949)

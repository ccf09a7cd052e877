"""CPU: the table path (z hyper-latent coder) — host, integer only, so it runs without a GPU.
Checked against vectors captured from the REAL reference (tests/golden/g5_table.json: RansEncoder/RansDecoder table
overloads, BufferedRansEncoder, set_stream/decode_stream, pmf_to_quantized_cdf) and against the oracle."""
import importlib.util
import json
import os

import numpy as np
import pytest

from flashgmm_amd import ans, ops

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _mg():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def test_g6_pmf_to_quantized_cdf_known_answer():
    """the reference's own known-answer test, tests/test_ops.py:104-106"""
    assert ops.pmf_to_quantized_cdf([0.1, 0.2, 0, 0], 16) == [0, 21845, 65534, 65535, 65536]


@pytest.mark.parametrize("bad", [[1, 0, -1], [1, 0, float("inf")], [1, 0, float("-inf")], [1, 0, float("nan")],
                                 [1, 0, float("nan"), 2, 3, 4], [0, 0, 0]])
def test_pmf_to_quantized_cdf_rejects(bad):
    """tests/test_ops.py:108-118"""
    with pytest.raises(ValueError):
        ops.pmf_to_quantized_cdf(bad, 16)


def test_pmf_to_quantized_cdf_matches_reference_rows(oracle):
    gold = json.load(open(os.path.join(GOLD, "g5_table.json")))
    cases = _mg().g5_cases(ops.pmf_to_quantized_cdf)  # the product builds the CDF rows itself
    assert [list(c) for c in cases["t1"][2]] == gold["cdfs"]
    rng = np.random.default_rng(1)
    for L in (1, 2, 7, 64, 300):
        for _ in range(20):
            pmf = rng.dirichlet(np.ones(L) * rng.uniform(0.05, 3)).astype(np.float32)
            pmf[rng.uniform(0, 1, L) < 0.3] = 0
            if pmf.sum() == 0:
                continue
            got = ops.pmf_to_quantized_cdf(pmf.tolist(), 16)
            assert got == oracle.pmf_to_quantized_cdf(pmf, 16)
            assert got[0] == 0 and got[-1] == 65536 and all(b > a for a, b in zip(got, got[1:]))


def test_g5_table_streams_verbatim():
    gold = json.load(open(os.path.join(GOLD, "g5_table.json")))
    for name, (sym, idx, cdfs, sizes, offsets) in _mg().g5_cases(ops.pmf_to_quantized_cdf).items():
        ent = gold["cases"][name]
        b = ans.RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), cdfs, sizes, offsets)
        assert b.hex() == ent["hex"], name
        d = ans.RansDecoder().decode_with_indexes(b, idx.tolist(), cdfs, sizes, offsets)
        assert isinstance(d, list) and d == ent["decoded"] == sym.tolist(), name
        # buffered encoder, two calls -> one stream; streaming decoder, two calls on one stream
        be = ans.BufferedRansEncoder()
        h = len(sym) // 2
        be.encode_with_indexes(sym[:h], idx[:h], cdfs, sizes, offsets)  # arrays are accepted as well as lists
        be.encode_with_indexes(sym[h:], idx[h:], cdfs, sizes, offsets)
        assert be.flush() == b and be.flush() == bytes.fromhex("0000008000000000")
        dec = ans.RansDecoder()
        dec.set_stream(b)
        assert dec.decode_stream(idx[:h], cdfs, sizes, offsets) + dec.decode_stream(idx[h:], cdfs, sizes, offsets) == ent["decoded"]


def test_table_path_matches_oracle_on_entropy_bottleneck_shapes(oracle):
    """z of a Kodak image: [1,192,8,12], one CDF row per channel; symbols a few sigma wide plus outliers"""
    rng = np.random.default_rng(4)
    C_, hw = 192, 96
    cdfs, sizes, offsets = [], [], []
    for c in range(C_):
        L = int(rng.integers(5, 60))
        pmf = np.exp(-0.5 * ((np.arange(L) - L / 2) / (L / 6)) ** 2).astype(np.float32) + 1e-6
        pmf /= pmf.sum()
        cdf = ops.pmf_to_quantized_cdf(np.concatenate([pmf, [1e-5]]).tolist(), 16)
        cdfs.append(cdf); sizes.append(len(cdf)); offsets.append(-(L // 2))
    idx = np.repeat(np.arange(C_, dtype=np.int32), hw)
    sym = np.round(rng.standard_normal(C_ * hw) * np.repeat([s / 7 for s in sizes], hw)).astype(np.int32)
    sym[::501] = 40000
    sym[7::733] = -123456
    b = ans.RansEncoder().encode_with_indexes(sym, idx, cdfs, sizes, offsets)
    assert b == oracle.encode_table(sym, idx, cdfs, sizes, offsets)
    assert ans.RansDecoder().decode_with_indexes(b, idx, cdfs, sizes, offsets) == sym.tolist()
    assert oracle.decode_table(b, idx, cdfs, sizes, offsets).tolist() == sym.tolist()


def test_table_path_validation():
    cdfs, sizes, offsets = [[0, 30000, 65535, 65536]], [4], [0]
    with pytest.raises(RuntimeError):
        ans.RansEncoder().encode_with_indexes([0, 1], [0, 5], cdfs, sizes, offsets)  # index out of range
    with pytest.raises(RuntimeError):
        ans.RansEncoder().encode_with_indexes([0, 1], [0], cdfs, sizes, offsets)  # length mismatch
    with pytest.raises(RuntimeError):
        ans.RansDecoder().decode_with_indexes(b"\x00" * 4, [0], cdfs, sizes, offsets)  # stream too short
    with pytest.raises(RuntimeError):
        ans.RansDecoder().decode_stream([0], cdfs, sizes, offsets)  # no stream set
    b = ans.RansEncoder().encode_with_indexes([0, 1, 1, 0], [0, 0, 0, 0], cdfs, sizes, offsets)
    with pytest.raises(RuntimeError):
        ans.RansDecoder().decode_with_indexes(b, [0] * 400, cdfs, sizes, offsets)  # asks for more than the stream holds


def _g8():
    import json
    import os

    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g8_hyperprior.json")))


def g8_coder(ent):
    """EntropyBottleneckCoder on the tables the reference's EntropyBottleneck.update() built (part of the fixture)"""
    import torch
    from flashgmm_amd import EntropyBottleneckCoder

    t = ent["tables"]
    med = torch.from_numpy(np.array(t["medians_bits"], np.uint32).view(np.float32).copy())
    return EntropyBottleneckCoder(torch.tensor(t["quantized_cdf"], dtype=torch.int32), torch.tensor(t["cdf_length"], dtype=torch.int32),
                                  torch.tensor(t["offset"], dtype=torch.int32), med)


def test_g8_hyper_latent_z_stream_equals_the_reference_classes():
    """golden G8, the `z` half: HyperLatentCodec over EntropyBottleneckCoder == the reference's HyperLatentCodec over its
    own EntropyBottleneck (same tables): z strings (bypass-coded symbols included), shape, z_hat -> params"""
    import hashlib

    import torch
    from tests import synth as T
    from flashgmm_amd.latent_codecs import HyperLatentCodec
    from golden.make_golden import G8_HASHED, G8_HYPER, bytes_to_json

    g8 = _g8()["polya"]
    Ha, Hs = T.exact_hyper_modules()
    for name, seed, c, cz, c_side, h, w, quantizer in G8_HYPER:
        ent = g8[name]
        assert ent["z_bypass_symbols"] > 0
        y, _ = T.exact_codec_inputs(seed, c, c_side, h, w)
        hyper = HyperLatentCodec(entropy_bottleneck=g8_coder(ent), h_a=Ha(c, cz), h_s=Hs(cz, c_side))
        z = hyper.h_a(torch.from_numpy(y))
        assert hashlib.sha256(z.contiguous().numpy().tobytes()).hexdigest() == ent["z_sha256"]
        out = hyper.compress(torch.from_numpy(y))
        assert [bytes_to_json(b, name in G8_HASHED) for b in out["strings"][0]] == ent["z_strings"] and list(out["shape"]) == ent["shape"]["hyper"]
        dec = hyper.decompress(out["strings"], out["shape"])
        assert torch.equal(dec["params"], out["params"]) and tuple(out["params"].shape) == (1, c_side, h, w)
        # compress() gets z_hat without decoding what it has just written: the same bits as decoding it
        zs2, z_hat_short = hyper.entropy_bottleneck.compress(z, return_dequantized=True)
        assert zs2 == out["strings"][0] and torch.equal(z_hat_short, hyper.entropy_bottleneck.decompress(zs2, out["shape"]))
        # z_hat = round(z - medians) + medians, bit for bit
        coder = hyper.entropy_bottleneck
        z_hat = coder.decompress(out["strings"][0], out["shape"])
        med = coder.medians.view(1, -1, 1, 1)
        assert torch.equal(z_hat, torch.round(z - med) + med)
        with pytest.raises(ValueError):
            coder.compress(torch.zeros(1, cz + 1, 2, 2))

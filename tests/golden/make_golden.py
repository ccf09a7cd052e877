#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (oracle/_ref, built by `make -C oracle ref ref-native`).

Runs only where /root/reference exists (this build container).  The reference itself never travels:
what is committed is data (inputs, expected floats / bytes / hashes) plus this script.

The reference latches APPROX_MODE once per process (rans_interface.cpp:100-115), so each mode runs in a
child process (`--worker MODE`).  Before anything is written, the x86-64-v3 build that travels to the GPU
box and the -march=native build (the reference's own setup.py flags) are checked to agree bit-for-bit.

Fixtures:
  g1_cdf.npz        G1/G2  4096 seeded rows x 3 modes: float32 CDF pairs of _fast_gmm_cdf<4> (bit patterns)
                           + the (start,range) they imply
  g3_small.json     G3     encode_with_indexes_gmm bytes, verbatim hex: n in {0,1,17,1000}, forced-bypass rows
                           with positive and negative symbols, plus decoder outputs
  ka1.json          KA-1   Kodak-half [1,192,32,24] cases (seeds 1234, 0..3): len + md5 + bypass count, 3 modes
  g5_table.json     G5/G6  table path (the z hyper-latent coder): encode_with_indexes bytes for ragged CDF rows
                           built by the reference's pmf_to_quantized_cdf, in-range and far out-of-range symbols
                           (bypass), buffered + streaming forms; the known answer of tests/test_ops.py:104-106
  g4_api.json       G4     Python-API level: the reference's own GaussianMixtureConditional.compress /
                           decompress imported IN PLACE from /root/reference (un-clamped sigma, >=10 % zero
                           channels): md5(bytes), len, abs_max, zero_bitmap, sha256(y_q), decompress == y_q
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from tests import synth as T  # noqa: E402

MODE_NAMES = ["polya", "as", "logistic"]


def g1_inputs():
    rng = np.random.default_rng(20251003)
    n = 4096
    e = np.exp(rng.uniform(-3, 2.5, n)).astype(np.float32)
    mu = (rng.standard_normal((n, 4)) * e[:, None]).astype(np.float32)
    sg = np.clip((rng.uniform(0, 2, (n, 4)) + 0.05) * e[:, None], 0.11, 256).astype(np.float32)
    lg = rng.standard_normal((n, 4))
    pi = (np.exp(lg) / np.exp(lg).sum(1, keepdims=True)).astype(np.float32)
    v = np.round(rng.standard_normal(n) * 1.5 * e + rng.standard_normal(n) * 2).astype(np.int32)
    return v, sg, mu, pi


def g3_cases():
    """name -> (symbols, scales, means, weights) small explicit cases."""
    cases = {}
    rng = np.random.default_rng(33)

    def rows(n, bypass_every=0):
        e = np.exp(rng.uniform(-2, 2, n)).astype(np.float32)
        mu = (rng.standard_normal((n, 4)) * e[:, None]).astype(np.float32)
        sg = np.clip((rng.uniform(0, 2, (n, 4)) + 0.05) * e[:, None], 0.11, 256).astype(np.float32)
        lg = rng.standard_normal((n, 4))
        pi = (np.exp(lg) / np.exp(lg).sum(1, keepdims=True)).astype(np.float32)
        v = np.round(rng.standard_normal(n) * 1.5 * e).astype(np.int32)
        if bypass_every:
            # forced bypass: sigma = 0.11 and |v - mu| >> sigma  => pmf == 0 (rans_interface.cpp:513)
            idx = np.arange(0, n, bypass_every)
            sg[idx] = 0.11
            mu[idx] = 0.0
            far = np.array([37, -37, 5, -5, 300, -300, 70000, -70000, 2**30, -(2**31)], np.int64)
            v[idx] = far[np.arange(len(idx)) % len(far)].astype(np.int32)
        return v, sg, mu, pi

    cases["n0"] = rows(0)
    cases["n1"] = rows(1)
    cases["n1_bypass_neg"] = (np.array([-9], np.int32), np.full((1, 4), 0.11, np.float32),
                              np.zeros((1, 4), np.float32), np.full((1, 4), 0.25, np.float32))
    cases["n17"] = rows(17, bypass_every=4)
    cases["n1000"] = rows(1000, bypass_every=97)
    return cases


def g5_cases(pmf_to_cdf):
    """table path: name -> (symbols, indexes, cdfs(list of lists), cdfs_sizes, offsets).  CDF rows come from
    `pmf_to_cdf` (the reference's / the oracle's pmf_to_quantized_cdf) on seeded pmfs of different lengths, so the
    matrix is ragged as EntropyBottleneck's is; symbols fall inside and far outside the tables (bypass)."""
    rng = np.random.default_rng(55)
    n_cdfs = 6
    lengths = [5, 9, 17, 33, 3, 64]
    cdfs, sizes, offsets = [], [], []
    for L in lengths:
        pmf = rng.dirichlet(np.ones(L) * 0.7).astype(np.float32)
        pmf[rng.integers(0, L)] = 0.0  # a zero-probability symbol: exercises the frequency stealing
        tail = np.float32(1e-4)
        cdf = pmf_to_cdf(np.concatenate([pmf, [tail]]).tolist(), 16)  # last entry = the bypass sentinel
        cdfs.append([int(v) for v in cdf])
        sizes.append(len(cdf))
        offsets.append(-(L // 2))
    cases = {}
    for name, n, spread in (("t0", 0, 1), ("t1", 1, 1), ("t_in", 300, 1), ("t_mixed", 2000, 4), ("t_far", 64, 4000)):
        idx = rng.integers(0, n_cdfs, n).astype(np.int32)
        half = np.array([lengths[k] // 2 for k in idx], np.float64)
        sym = np.round(rng.standard_normal(n) * half * 0.6 * spread).astype(np.int32)
        cases[name] = (sym, idx, cdfs, sizes, offsets)
    return cases


def ka1_case(seed):
    y, sg, mu, pi = T.make_latent(seed)
    return T.to_coder_inputs(y, sg, mu, pi)


def import_reference_entropy_models(ans_mod):
    """Import /root/reference/compressai/entropy_models/entropy_models.py IN PLACE.

    `import compressai` fails here with ModuleNotFoundError (torch_geometric / torchvision / pytorch_msssim are
    not installed), so a bare package object with the right __path__ stands in for compressai/__init__.py and
    the two compiled extensions come from oracle/_ref."""
    import importlib.util
    import sysconfig

    pkg = types.ModuleType("compressai")
    pkg.__path__ = ["/root/reference/compressai"]
    pkg.available_entropy_coders = lambda: ["ans"]
    pkg.get_entropy_coder = lambda: "ans"
    pkg.ans = ans_mod
    sys.modules["compressai"] = pkg
    sys.modules["compressai.ans"] = ans_mod
    cxx_path = os.path.join(O.REF_DIR, "_CXX" + sysconfig.get_config_var("EXT_SUFFIX"))
    spec = importlib.util.spec_from_file_location("compressai._CXX", cxx_path)
    cxx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cxx)
    sys.modules["compressai._CXX"] = cxx
    from compressai.entropy_models.entropy_models import GaussianMixtureConditional

    return GaussianMixtureConditional


def import_reference_latent_codecs():
    """The reference's own CheckerboardLatentCodec / ChannelGroupsLatentCodec / GaussianMixtureConditionalLatentCodec,
    imported IN PLACE (after import_reference_entropy_models).  compressai.registry pulls in torch_geometric,
    torchvision and pytorch_msssim, which are not installed: inert stand-ins for those three THIRD-PARTY packages
    (none of their code is on the path) let the import proceed."""

    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            m = _Any(self.__name__ + "." + name)
            sys.modules[m.__name__] = m
            return m

        def __call__(self, *a, **k):
            return None

    for name in ("torch_geometric", "torch_geometric.transforms", "torchvision", "torchvision.transforms", "pytorch_msssim"):
        sys.modules.setdefault(name, _Any(name))
    from compressai.latent_codecs.channel_groups import ChannelGroupsLatentCodec
    from compressai.latent_codecs.checkerboard import CheckerboardLatentCodec
    from compressai.latent_codecs.gaussian_mixture_conditional import GaussianMixtureConditionalLatentCodec

    return CheckerboardLatentCodec, ChannelGroupsLatentCodec, GaussianMixtureConditionalLatentCodec


def import_reference_hyper_codecs():
    """The reference's own HyperpriorLatentCodec / HyperLatentCodec / EntropyBottleneck, in place (after the two imports above)."""
    from compressai.entropy_models.entropy_models import EntropyBottleneck
    from compressai.latent_codecs.hyper import HyperLatentCodec
    from compressai.latent_codecs.hyperprior import HyperpriorLatentCodec

    return HyperpriorLatentCodec, HyperLatentCodec, EntropyBottleneck


# G8: the complete nested result of a hyperprior model around the GMM path (models/ckbd_gmm.py:109-123):
# (name, seed, c, cz, c_side, h, w, quantizer)
# the second case has the MODEL's geometry (Cheng2020AnchorCheckerboardGMMv2, N = 192, on a Kodak image: y [1, 192, 32, 48], the
# EntropyBottleneck over N = 192 channels, 2 N = 384 side channels, models/ckbd_gmm.py:78-131); its bitstreams are stored as
# length + sha256 (G8_HASHED), not verbatim
G8_HYPER = [("hyperprior_ckbd", 31, 6, 4, 8, 8, 12, "noise"), ("hyperprior_ckbd_kodak_n192", 32, 192, 192, 384, 32, 48, "noise")]
G8_HASHED = {"hyperprior_ckbd_kodak_n192"}


def build_entropy_bottleneck(EB, cz: int):
    """the reference's EntropyBottleneck with its random initial density model (seeded) and dyadic, channel-dependent
    medians, tables built by its own update()"""
    import torch

    torch.manual_seed(5)
    eb = EB(cz)
    med = (torch.arange(cz, dtype=torch.float32) % 5 - 2.0) / 4.0  # -0.5 .. 0.5 in steps of 1/4
    q = eb.quantiles.data
    q[:, 0, 0], q[:, 0, 1], q[:, 0, 2] = med - 7.0, med, med + 9.0  # asymmetric supports: offsets differ from -lengths/2
    eb.update(force=True)
    return eb


G7_CKBD = [  # CheckerboardLatentCodec cases: (name, seed, c, c_side, h, w, dead channels, quantizer, anchor_parity)
    ("ckbd_noise_even", 11, 6, 8, 8, 12, 0, "noise", "even"),
    ("ckbd_ste_odd", 12, 5, 6, 6, 10, 1, "weighted_mean_ste", "odd"),
]
G7_GROUPS = [  # ChannelGroupsLatentCodec of checkerboard codecs: (name, seed, groups, c_side, h, w, quantizer)
    ("groups_224", 21, [2, 2, 4], 8, 8, 12, "noise"),
]


def build_codecs(Ckbd, Groups, Gmm, Ctx, Par, kind, cfg):
    """the same wiring for the reference's classes and for flashgmm_amd's (models/ckbd_gmm.py:111, elic_gmm.py:198-219)"""
    if kind == "ckbd":
        name, seed, c, c_side, h, w, dead, quantizer, parity = cfg
        return Ckbd(latent_codec={"y": Gmm(K=4, quantizer=quantizer)}, context_prediction=Ctx(c, 2 * c),
                    entropy_parameters=Par(2 * c + c_side, c), anchor_parity=parity)
    name, seed, groups, c_side, h, w, quantizer = cfg
    latent = {f"y{k}": Ckbd(latent_codec={"y": Gmm(K=4, quantizer=quantizer)}, context_prediction=Ctx(g, 2 * g),
                            entropy_parameters=Par(2 * g + (k > 0) * 2 * g + c_side, g))
              for k, g in enumerate(groups)}
    chctx = {f"y{k}": Ctx(sum(groups[:k]), 2 * groups[k]) for k in range(1, len(groups))}
    return Groups(groups=groups, channel_context=chctx, latent_codec=latent)


def bytes_to_json(b: bytes, hashed: bool):
    return {"len": len(b), "sha256": hashlib.sha256(b).hexdigest()} if hashed else b.hex()


def strings_to_json(strings, hashed: bool = False):
    if hashed:
        return [{**bytes_to_json(b, True), "abs_max": int(a), "zero_bitmap": [int(v) for v in zb.tolist()]} for (b, a, zb) in strings]
    return [{"hex": b.hex(), "abs_max": int(a), "zero_bitmap": [int(v) for v in zb.tolist()]} for (b, a, zb) in strings]


def worker(mode: int, flavour: str):
    import torch

    os.environ["APPROX_MODE"] = str(mode)
    ans = O.ref_ans(flavour)
    probe = O.ref_probe(flavour)
    assert probe.ref_probe_mode() == (mode | 0x100)
    ts = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    out = {}

    v, sg, mu, pi = g1_inputs()
    c1, c2 = O.ref_gmm_cdf(probe, v, sg, mu, pi)
    out["g1_c1_bits"] = c1.view(np.uint32).tolist()
    out["g1_c2_bits"] = c2.view(np.uint32).tolist()

    g3 = {}
    for name, (sym, s, m, w) in g3_cases().items():
        b = ans.RansEncoder().encode_with_indexes_gmm(ts(sym), ts(s), ts(m), ts(w), 0)
        max_bs = int(np.abs(sym.astype(np.int64)).max() + 2) if len(sym) else 1
        max_bs = min(max_bs, 400)  # bisection half-width; bypass symbols do not use it
        d = ans.RansDecoder().decode_with_indexes_gmm(b, ts(s), ts(m), ts(w), max_bs).numpy()
        g3[name] = {"hex": b.hex(), "max_bs": max_bs, "decoded": d.tolist()}
    out["g3"] = g3

    ka = {}
    for seed in (1234, 0, 1, 2, 3):
        sym, s, m, w, abs_max, zb, yq = ka1_case(seed)
        b = ans.RansEncoder().encode_with_indexes_gmm(ts(sym), ts(s), ts(m), ts(w), abs_max + 1)
        d = ans.RansDecoder().decode_with_indexes_gmm(b, ts(s), ts(m), ts(w), abs_max + 1).numpy()
        ka[str(seed)] = {"n": int(len(sym)), "nz_channels": int(zb.sum()), "abs_max": int(abs_max), "len": len(b),
                         "md5": hashlib.md5(b).hexdigest(), "roundtrip": bool((d == sym).all())}
    out["ka1"] = ka

    cxx = O.ref_cxx(flavour)
    g5 = {}
    for name, (sym, idx, cdfs, sizes, offsets) in g5_cases(cxx.pmf_to_quantized_cdf).items():
        b = ans.RansEncoder().encode_with_indexes(sym.tolist(), idx.tolist(), cdfs, sizes, offsets)
        d = ans.RansDecoder().decode_with_indexes(b, idx.tolist(), cdfs, sizes, offsets)
        # buffered form, two calls into one stream + streaming decoder (set_stream / decode_stream)
        be = ans.BufferedRansEncoder()
        h = len(sym) // 2
        be.encode_with_indexes(sym[:h].tolist(), idx[:h].tolist(), cdfs, sizes, offsets)
        be.encode_with_indexes(sym[h:].tolist(), idx[h:].tolist(), cdfs, sizes, offsets)
        b2 = be.flush()
        dec = ans.RansDecoder()
        dec.set_stream(b2)
        d2 = dec.decode_stream(idx[:h].tolist(), cdfs, sizes, offsets) + dec.decode_stream(idx[h:].tolist(), cdfs, sizes, offsets)
        g5[name] = {"hex": b.hex(), "decoded": list(d), "buffered_equal": b2 == b, "stream_decoded_equal": list(d2) == list(d)}
    out["g5"] = g5
    out["g5_cdfs"] = [list(map(int, c)) for c in g5_cases(cxx.pmf_to_quantized_cdf)["t1"][2]]
    out["g6"] = [int(v) for v in cxx.pmf_to_quantized_cdf([0.1, 0.2, 0, 0], 16)]  # tests/test_ops.py:104-106

    if flavour == "":
        GMC = import_reference_entropy_models(ans)
        g4 = {}
        for seed, shape in ((4321, (192, 32, 24)), (77, (192, 16, 8)), (5, (320, 8, 12))):
            M, h, w_ = shape
            y, sg4, mu4, pi4 = T.make_latent(seed, M=M, h=h, w=w_, clamp=False, zero_frac=0.15)
            gmc = GMC(K=4)
            (b, abs_max, zb), yq = gmc.compress(ts(y), ts(sg4), ts(mu4), ts(pi4))
            y_hat = gmc.decompress(b, abs_max, zb, ts(sg4), ts(mu4), ts(pi4))
            g4[str(seed)] = {
                "M": M, "h": h, "w": w_, "len": len(b), "md5": hashlib.md5(b).hexdigest(), "abs_max": int(abs_max),
                "zero_bitmap": zb.tolist(), "yq_sha256": hashlib.sha256(yq.numpy().tobytes()).hexdigest(),
                "decompress_equals_yq": bool(torch.equal(y_hat, yq)),
                "y_hat_dtype": str(y_hat.dtype), "y_hat_shape": list(y_hat.shape),
            }
        out["g4"] = g4

        # G7: the codecs above the entropy model, run with the reference's own classes on exact networks
        import contextlib

        Ckbd, Groups, Gmm = import_reference_latent_codecs()
        Ctx, Par = T.exact_modules()
        g7 = {}
        with contextlib.redirect_stdout(sys.stderr):  # the reference's codecs print timings
            for kind, cfgs in (("ckbd", G7_CKBD), ("groups", G7_GROUPS)):
                for cfg in cfgs:
                    name, seed = cfg[0], cfg[1]
                    if kind == "ckbd":
                        y, side = T.exact_codec_inputs(seed, cfg[2], cfg[3], cfg[4], cfg[5], dead=cfg[6])
                    else:
                        y, side = T.exact_codec_inputs(seed, sum(cfg[2]), cfg[3], cfg[4], cfg[5])
                    codec = build_codecs(Ckbd, Groups, Gmm, Ctx, Par, kind, cfg)
                    enc = codec.compress(ts(y), ts(side))
                    dec = codec.decompress(enc["strings"], enc["shape"], ts(side))
                    g7[name] = {
                        "strings": strings_to_json(enc["strings"]),
                        "shape": [list(s_) for s_ in enc["shape"]] if kind == "groups" else list(enc["shape"]),
                        "y_hat_sha256": hashlib.sha256(enc["y_hat"].contiguous().numpy().tobytes()).hexdigest(),
                        "decompress_y_hat_sha256": hashlib.sha256(dec["y_hat"].contiguous().numpy().tobytes()).hexdigest(),
                    }
        out["g7"] = g7

        # G8: HyperpriorLatentCodec{y: CheckerboardLatentCodec(GMM), hyper: HyperLatentCodec(EntropyBottleneck)} — the
        # reference's own classes end to end; the EntropyBottleneck's tables (float work of its update() on this CPU) are
        # part of the fixture, so that the mirror codes against the same integers
        Hyperprior, Hyper, EB = import_reference_hyper_codecs()
        Ha, Hs = T.exact_hyper_modules()
        g8 = {}
        with contextlib.redirect_stdout(sys.stderr):
            for name, seed, c, cz, c_side, h, w_, quantizer in G8_HYPER:
                y, _ = T.exact_codec_inputs(seed, c, c_side, h, w_)
                eb = build_entropy_bottleneck(EB, cz)
                ycodec = build_codecs(Ckbd, Groups, Gmm, Ctx, Par, "ckbd", (name, seed, c, c_side, h, w_, 0, quantizer, "even"))
                codec = Hyperprior(latent_codec={"y": ycodec, "hyper": Hyper(entropy_bottleneck=eb, h_a=Ha(c, cz), h_s=Hs(cz, c_side))})
                enc = codec.compress(ts(y))
                dec = codec.decompress(enc["strings"], enc["shape"])
                *ys_, zs_ = enc["strings"]
                z = Ha(c, cz)(ts(y))
                g8[name] = {
                    "tables": {"quantized_cdf": eb._quantized_cdf.tolist(), "cdf_length": eb._cdf_length.tolist(),
                               "offset": eb._offset.tolist(), "medians_bits": eb.quantiles[:, 0, 1].detach().numpy().view(np.uint32).tolist()},
                    "y_strings": strings_to_json(ys_, name in G8_HASHED), "z_strings": [bytes_to_json(b, name in G8_HASHED) for b in zs_],
                    "shape": {"y": list(enc["shape"]["y"]), "hyper": list(enc["shape"]["hyper"])},
                    "z_sha256": hashlib.sha256(z.contiguous().numpy().tobytes()).hexdigest(),
                    "z_bypass_symbols": int(sum(int(((zs < o) | (zs >= o + ln - 2)).sum()) for zs, o, ln in zip(
                        (z - eb.quantiles[:, 0, 1].detach().view(1, -1, 1, 1)).round()[0], eb._offset.tolist(), eb._cdf_length.tolist()))),
                    "y_hat_sha256": hashlib.sha256(enc["y_hat"].contiguous().numpy().tobytes()).hexdigest(),
                    "decompress_y_hat_sha256": hashlib.sha256(dec["y_hat"].contiguous().numpy().tobytes()).hexdigest(),
                }
        out["g8"] = g8
    json.dump(out, sys.stdout)


def run_worker(mode, flavour):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(mode), "--flavour", flavour],
                       check=True, capture_output=True, text=True)
    return json.loads(r.stdout[r.stdout.index("{"):])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", type=int, default=None)
    ap.add_argument("--flavour", default="")
    a = ap.parse_args()
    if a.worker is not None:
        return worker(a.worker, a.flavour)

    assert os.path.isdir("/root/reference"), "the reference is only present in the build container"
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle", "ref"])
    have_native = O.ref_available("native")
    res = {}
    for mode, name in enumerate(MODE_NAMES):
        r = run_worker(mode, "")
        if have_native:
            rn = run_worker(mode, "native")
            for key in ("g1_c1_bits", "g1_c2_bits", "g3", "ka1", "g5", "g5_cdfs", "g6"):
                assert r[key] == rn[key], f"x86-64-v3 and native reference builds differ: {name}/{key}"
        res[name] = r
        print(f"[{name}] reference run ok" + (" (v3 == native)" if have_native else ""))

    # --- cross-check this repo's C restatement against the reference outputs before writing anything ---
    v, sg, mu, pi = g1_inputs()
    npz = {"v": v, "scales": sg, "means": mu, "weights": pi}
    for mode, name in enumerate(MODE_NAMES):
        c1 = np.array(res[name]["g1_c1_bits"], np.uint32)
        c2 = np.array(res[name]["g1_c2_bits"], np.uint32)
        o1, o2 = O.gmm_cdf(mode, v, sg, mu, pi)
        assert (o1.view(np.uint32) == c1).all() and (o2.view(np.uint32) == c2).all(), f"oracle float mismatch {name}"
        npz[f"c1_{name}"] = c1
        npz[f"c2_{name}"] = c2
        # G2: (start, range) implied by the reference floats (rans_interface.cpp:509-512)
        lo = (c1.view(np.float32) * np.float32(65535.0)).astype(np.int64) & 0xFFFF
        hi = (c2.view(np.float32) * np.float32(65535.0)).astype(np.int64) & 0xFFFF
        npz[f"start_{name}"] = lo.astype(np.uint16)
        npz[f"range_{name}"] = ((hi - lo) & 0xFFFF).astype(np.uint16)
        for cname, (sym, s, m, w) in g3_cases().items():
            assert O.encode_gmm(mode, sym, s, m, w).hex() == res[name]["g3"][cname]["hex"], (name, cname)
        for seed, ent in res[name]["ka1"].items():
            sym, s, m, w, abs_max, zb, yq = ka1_case(int(seed))
            b = O.encode_gmm(mode, sym, s, m, w)
            assert hashlib.md5(b).hexdigest() == ent["md5"] and ent["roundtrip"], (name, seed)
    np.savez_compressed(os.path.join(HERE, "g1_cdf.npz"), **npz)

    g3_out = {"note": "inputs are regenerated by make_golden.g3_cases(); hex = RansEncoder.encode_with_indexes_gmm output",
              "cases": {}}
    for cname, (sym, s, m, w) in g3_cases().items():
        ent = {"symbols": sym.tolist()}
        if len(sym) <= 17:
            ent.update(scales=s.tolist(), means=m.tolist(), weights=w.tolist())
        for name in MODE_NAMES:
            ent[name] = res[name]["g3"][cname]
        g3_out["cases"][cname] = ent
    json.dump(g3_out, open(os.path.join(HERE, "g3_small.json"), "w"), indent=1)
    json.dump({name: res[name]["ka1"] for name in MODE_NAMES}, open(os.path.join(HERE, "ka1.json"), "w"), indent=1)
    json.dump({name: res[name]["g4"] for name in MODE_NAMES}, open(os.path.join(HERE, "g4_api.json"), "w"), indent=1)
    json.dump({name: res[name]["g7"] for name in MODE_NAMES}, open(os.path.join(HERE, "g7_codecs.json"), "w"), indent=1)
    for name in MODE_NAMES[1:]:  # the z stream and its tables do not depend on the Phi approximation
        for k, v in res[name]["g8"].items():
            assert v["tables"] == res["polya"]["g8"][k]["tables"] and v["z_strings"] == res["polya"]["g8"][k]["z_strings"]
    json.dump({name: res[name]["g8"] for name in MODE_NAMES}, open(os.path.join(HERE, "g8_hyperprior.json"), "w"), indent=1)
    # table path (mode independent): the oracle must rebuild the same CDF rows and the same bytes
    r0 = res["polya"]
    assert r0["g6"] == [0, 21845, 65534, 65535, 65536] == O.pmf_to_quantized_cdf([0.1, 0.2, 0, 0], 16)
    ocases = g5_cases(O.pmf_to_quantized_cdf)
    assert [list(c) for c in ocases["t1"][2]] == r0["g5_cdfs"], "oracle pmf_to_quantized_cdf differs from the reference"
    for name, (sym, idx, cdfs, sizes, offsets) in ocases.items():
        assert O.encode_table(sym, idx, cdfs, sizes, offsets).hex() == r0["g5"][name]["hex"], name
        assert O.decode_table(bytes.fromhex(r0["g5"][name]["hex"]), idx, cdfs, sizes, offsets).tolist() == r0["g5"][name]["decoded"]
        assert r0["g5"][name]["buffered_equal"] and r0["g5"][name]["stream_decoded_equal"], name
        assert r0["g5"][name]["decoded"] == sym.tolist(), name  # the reference round-trips, bypass included
    json.dump({"g6_pmf_to_quantized_cdf": r0["g6"], "cdfs": r0["g5_cdfs"],
               "cases": {k: {"hex": v["hex"], "decoded": v["decoded"]} for k, v in r0["g5"].items()}},
              open(os.path.join(HERE, "g5_table.json"), "w"), indent=1)
    # the survey's KA-1 must be what we just reproduced
    assert res["polya"]["ka1"]["1234"]["md5"] == "e759909d27406fbc0168c33b4509772d"
    assert res["as"]["ka1"]["1234"]["md5"] == "9283e03480f545471e6245b21aa61af5"
    assert res["logistic"]["ka1"]["1234"]["md5"] == "ecbe33ac17fe297803909b32db54d1f6"
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()

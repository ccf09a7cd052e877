#!/usr/bin/env python3
"""Full-size golden vectors from the REAL reference (build container only; oracle/_ref built by oracle/Makefile).

    python tests/golden/make_golden_fullsize.py

Writes tests/golden/fullsize.json - length + md5 of what the reference's `RansEncoder.encode_with_indexes_gmm`
(compressai/cpp_exts/rans/rans_interface.cpp:609-617) returns for EVERY bitstream of the BASELINE configurations the GPU
tests and bench.py run at full size (inputs regenerated from their seeds by tests/synth.py: make_latent):

  kodak24        the 48 checkerboard halves [1,192,32,24] of bench.py's Kodak batch (rank 0: seeds 0..47), modes polya / as /
                 logistic - configs[1], configs[2]
  elic_groups    tests/test_gpu_parity.py::test_elic_channel_group_shapes: five channel groups 16/16/32/64/192 of a 4K latent half
                 (h*w = 136*120; seeds 51..55, sigma not pre-clamped), polya
  elic4k_image0  the ten bitstreams of image 0 of bench.py's ELIC-4K leg (seeds 0..9, groups x halves, fp16 parameter planes
                 widened to fp32: what the reference is fed), polya - configs[4]

One subprocess per approximation mode (the reference latches APPROX_MODE once per process, rans_interface.cpp:99-115); every
stream is also decoded by the reference (`roundtrip`).  Nothing of the reference is copied: the fixture is lengths and hashes."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import synth as T  # noqa: E402
from oracle import oracle as O  # noqa: E402

MODE_NAMES = ["polya", "as", "logistic"]
ELIC_GROUPS = (16, 16, 32, 64, 192)


def cases(mode_name):
    """-> [(set name, key, coder inputs)]"""
    for seed in range(48):
        yield "kodak24", str(seed), T.to_coder_inputs(*T.make_latent(seed))
    if mode_name == "polya":
        for seed, M in zip((51, 52, 53, 54, 55), ELIC_GROUPS):
            yield "elic_groups", str(seed), T.to_coder_inputs(*T.make_latent(seed, M=M, h=136, w=120, clamp=False))
        k = 0
        for g in ELIC_GROUPS:
            for _ in range(2):  # bench.py: stream_seed(rank 0, image 0, stream k, 10) = k
                y, sg, mu, pi = T.make_latent(k, M=g, h=136, w=120)
                sg, mu, pi = (a.astype(np.float32) for a in T.to_float16_planes(sg, mu, pi))
                yield "elic4k_image0", str(k), T.to_coder_inputs(y, sg, mu, pi)
                k += 1


def worker(mode: int):
    import torch

    os.environ["APPROX_MODE"] = str(mode)
    ans = O.ref_ans("")
    out = {}
    for name, key, (sym, s, m, w, abs_max, zb, yq) in cases(MODE_NAMES[mode]):
        # the (n,4) views with strides (1,n) the reference's Python hands its coder (entropy_models.py:810-828)
        ts = [torch.from_numpy(np.ascontiguousarray(a.T)).T for a in (s, m, w)]
        b = ans.RansEncoder().encode_with_indexes_gmm(torch.from_numpy(sym), *ts, abs_max + 1)
        d = ans.RansDecoder().decode_with_indexes_gmm(b, *ts, abs_max + 1).numpy()
        out.setdefault(name, {})[key] = {"n": int(len(sym)), "abs_max": int(abs_max), "nz_channels": int(zb.sum()), "len": len(b),
                                         "md5": hashlib.md5(b).hexdigest(), "roundtrip": bool((d == sym).all())}
    print(json.dumps(out))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        return worker(int(sys.argv[2]))
    assert os.path.isdir("/root/reference"), "the reference is only present in the build container"
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle", "ref"])
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(m)], stdout=subprocess.PIPE, text=True) for m in range(3)]
    res = {}
    for name, p in zip(MODE_NAMES, procs):
        so = p.communicate()[0]
        assert p.returncode == 0, name
        res[name] = json.loads(so[so.index("{"):])
        assert all(e["roundtrip"] for st in res[name].values() for e in st.values()), name
    # must agree with the small fixture written by make_golden.py (seeds 0..3) ...
    ka1 = json.load(open(os.path.join(HERE, "ka1.json")))
    for name in MODE_NAMES:
        for seed in range(4):
            assert res[name]["kodak24"][str(seed)]["md5"] == ka1[name][str(seed)]["md5"], (name, seed)
    # ... and this repo's C restatement must reproduce every one of them before anything is written
    for mode, name in enumerate(MODE_NAMES):
        for set_name, key, (sym, s, m, w, abs_max, zb, yq) in cases(name):
            b = O.encode_gmm(mode, sym, s, m, w)
            assert (len(b), hashlib.md5(b).hexdigest()) == (res[name][set_name][key]["len"], res[name][set_name][key]["md5"]), (name, set_name, key)
        print(f"[{name}] reference == oracle on {sum(len(v) for v in res[name].values())} full-size bitstreams")
    json.dump({"note": "tests/golden/make_golden_fullsize.py: len + md5 of the reference encoder's output per bitstream", **res},
              open(os.path.join(HERE, "fullsize.json"), "w"), indent=0)
    print("written", os.path.join(HERE, "fullsize.json"))


if __name__ == "__main__":
    main()

"""CPU, build container only: the reference's own Python `EntropyBottleneck` (imported in place from /root/reference)
must behave identically on this repo's drop-in `ans` / `pmf_to_quantized_cdf` and on its own compiled extensions."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(
    not os.path.isdir("/root/reference/compressai") or not os.path.exists(os.path.join(HERE, "..", "oracle", "_ref")),
    reason="needs /root/reference and oracle/_ref (build container only)")


def _run(backend):
    r = subprocess.run([sys.executable, os.path.join(HERE, "dropin_worker.py"), backend], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout[r.stdout.index("{"):])


def test_entropy_bottleneck_on_dropin_modules_equals_reference():
    ref, got = _run("ref"), _run("fgmm")
    assert got["cdf"] == ref["cdf"]          # update(): pmf_to_quantized_cdf rows
    assert got["strings"] == ref["strings"]  # compress(): table rANS bytes, bypass included
    assert got["z_hat_sum"] == ref["z_hat_sum"] and got["z_hat_max"] == ref["z_hat_max"] == 500.0

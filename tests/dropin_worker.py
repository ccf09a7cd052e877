"""Helper of tests/test_reference_dropin_cpu.py: run the REFERENCE'S OWN Python EntropyBottleneck (imported in place
from /root/reference — only present in the build container) on top of either the reference's compiled extensions
(oracle/_ref) or this repo's drop-in modules, and print what it produced."""
import importlib.util
import json
import os
import sys
import sysconfig
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(backend: str):
    import torch

    if backend == "ref":
        from oracle import oracle as O

        ans = O.ref_ans()
        cxx = O.ref_cxx()
    else:
        from flashgmm_amd import ans, ops

        cxx = types.ModuleType("compressai._CXX")
        cxx.pmf_to_quantized_cdf = ops.pmf_to_quantized_cdf
    pkg = types.ModuleType("compressai")
    pkg.__path__ = ["/root/reference/compressai"]
    pkg.available_entropy_coders = lambda: ["ans"]
    pkg.get_entropy_coder = lambda: "ans"
    pkg.ans = ans
    sys.modules["compressai"] = pkg
    sys.modules["compressai.ans"] = ans
    sys.modules["compressai._CXX"] = cxx
    from compressai.entropy_models.entropy_models import EntropyBottleneck

    torch.manual_seed(7)
    eb = EntropyBottleneck(24)
    eb.update(force=True)
    z = torch.randn(2, 24, 8, 12) * 3
    z[0, 3, 2, 2] = 500.0   # far outside the table: bypass
    z[1, 5, 1, 1] = -321.0
    strings = eb.compress(z)
    z_hat = eb.decompress(strings, z.size()[2:])
    print(json.dumps({"cdf": eb._quantized_cdf.tolist(), "strings": [s.hex() for s in strings],
                      "z_hat_sum": float(z_hat.double().sum()), "z_hat_max": float(z_hat.max())}))


if __name__ == "__main__":
    main(sys.argv[1])

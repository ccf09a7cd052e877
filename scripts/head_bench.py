"""The f2 ledger, measured: one Kodak batch (48 halves [1,192,32,24], features [48,640,32,24]) encoded three ways -
  torch    torch.nn.functional.conv2d (fp32) -> chunk -> compress_batch(weights_are_logits)       what a caller does today
  unfused  ParameterHead.params (MFMA kernel, planes written) -> compress_batch                     library-owned order, planes in HBM
  fused    compress_head_batch (MFMA kernel with the table entries as its epilogue)                 no parameter planes at all
GPU time of each stage by HIP/CUDA events on the current stream (the library's own events for its kernels), whole-call wall time,
MFMA fraction of the 157.3 TF f32 roof.  usage: python scripts/head_bench.py [images=24] [reps=20] [c_in=640]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from flashgmm_amd import GaussianMixtureConditional, ParameterHead, _lib  # noqa: E402
from tests.synth import make_head  # noqa: E402

images = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
c_in = int(sys.argv[3]) if len(sys.argv) > 3 else 640
N, M, h, w = 2 * images, 192, 32, 24
conv, x, y = make_head(5, M, c_in, h, w, N)
arith = os.environ.get("HEAD_ARITH", "f32")  # or bf16x6 (FGMM_HEAD_BF16X6: fgmm_head16.hip)
head = ParameterHead(conv, arithmetic=arith)
gmc = GaussianMixtureConditional(K=4, mode="polya")
_lib.set_profiling(0, True)
flop = 2.0 * 12 * M * c_in * N * h * w


def ev_time(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    r = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b), r


def torch_params():
    with torch.no_grad():
        return torch.nn.functional.conv2d(x, conv.weight, conv.bias).chunk(3, 1)


out = {"arithmetic": arith, "workload": f"{images} Kodak-sized images = {N} halves [1,{M},{h},{w}], head Conv2d({c_in}, {12 * M}, 1)", "gemm_gflop": round(flop / 1e9, 1)}
for name, prm in (("torch", torch_params), ("unfused", lambda: head.params(x))):
    for _ in range(3):
        p = prm()
        gmc.compress_batch(y, *p, weights_are_logits=True)
    t_prm, t_call, t_sym = [], [], []
    for _ in range(reps):
        ms, p = ev_time(prm)
        t_prm.append(ms)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = gmc.compress_batch(y, *p, weights_are_logits=True)
        t_call.append((time.perf_counter() - t0) * 1e3)
        t_sym.append(_lib.kernel_ms(0, 0))
    out[name] = {"params_ms": round(float(np.median(t_prm)), 4), "params_tflops": round(flop / np.median(t_prm) / 1e9, 1),
                 "symtab_kernel_ms": round(float(np.median(t_sym)), 4), "compress_call_ms": round(float(np.median(t_call)), 3),
                 "gpu_ms_params_plus_symtab": round(float(np.median(t_prm) + np.median(t_sym)), 4)}
    out[name + "_bytes"] = [bytes(b) for b in res.strings]
for _ in range(3):
    gmc.compress_head_batch(y, x, head)
t_call, t_k = [], []
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = gmc.compress_head_batch(y, x, head)
    t_call.append((time.perf_counter() - t0) * 1e3)
    t_k.append(_lib.kernel_ms(0, 0))
k = float(np.median(t_k))
out["fused"] = {"head_symtab_kernel_ms": round(k, 4), "tflops": round(flop / k / 1e9, 1), "mfma_frac_of_157.3TF": round(flop / k / 1e9 / 157.3, 4),
                "compress_call_ms": round(float(np.median(t_call)), 3)}
fb = [bytes(b) for b in res.strings]
out["fused"]["bytes_equal_unfused"] = fb == out.pop("unfused_bytes")
out["fused"]["bytes_equal_torch_conv_params"] = fb == out.pop("torch_bytes")
out["unfused"]["mfma_frac_of_157.3TF"] = round(out["unfused"]["params_tflops"] / 157.3, 4)
out["ledger_gpu_ms"] = {"torch conv + symtab": out["torch"]["gpu_ms_params_plus_symtab"], "head_params + symtab": out["unfused"]["gpu_ms_params_plus_symtab"],
                        "fused head": out["fused"]["head_symtab_kernel_ms"]}
print(json.dumps(out))

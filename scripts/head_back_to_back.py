"""is the head kernel's per-launch fixed part (0.13 ms of 1.07) a property of the launch or of the idle gap before it?  1, 2, 4, 8
launches back to back inside one event bracket.   python scripts/head_back_to_back.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flashgmm_amd import GaussianMixtureConditional, ParameterHead, _lib  # noqa: E402
from tests.synth import make_head  # noqa: E402

conv, x, y = make_head(5, 192, 640, 32, 24, 48)
head = ParameterHead(conv, arithmetic=os.environ.get("HEAD_ARITH", "f32"))
gmc = GaussianMixtureConditional(K=4, mode="polya")
for _ in range(3):
    head.params(x)
for n in (1, 2, 4, 8):
    ts = []
    for rep in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            head.params(x)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"{n} launches back to back: {np.median(ts):7.3f} ms = {np.median(ts) / n:6.3f} per launch")

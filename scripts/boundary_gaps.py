"""Dev aid (GPU): where the interpreter's time between the native calls of a kodak24 step goes (bench.py `phases_ms.between_calls`).
Every piece of the calling thread's work outside the library is bracketed by a clock read; the library's own call durations come from
its call log.  Plain and checkpointed.   python scripts/boundary_gaps.py [steps]"""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import bench as B
from flashgmm_amd import GaussianMixtureConditional, _lib
from flashgmm_amd import entropy_models as EM
from flashgmm_amd import parallel as P

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print(P.bind_to_gpu_numa_node(0))
mine, note = P.plan_l3(0, 1)
host, devt, pix = B.make_workload(0, 24, dev, "kodak24", False)
ys, ss, ms, ws = (torch.cat([t[k] for t in devt]) for k in range(4))
n = ys.shape[0]
_lib.ctx(0, 0)  # (the library's workers before the calling thread narrows itself, as bench.py has it)
if mine:
    os.sched_setaffinity(0, mine)
gc.disable()
nat = _lib.native()
now = time.perf_counter


class Timed:
    """wraps a function of the compiled module: wall time of every call"""

    def __init__(self, f):
        self.f, self.t = f, []

    def __call__(self, *a, **k):
        t0 = now()
        r = self.f(*a, **k)
        self.t.append((t0, now()))
        return r


class NatProxy:
    def __init__(self, nat):
        self.compress_stacked = Timed(nat.compress_stacked)
        self.decompress_stacked = Timed(nat.decompress_stacked)
        self.compress_head_stacked = nat.compress_head_stacked
        self.abi_version = nat.abi_version


proxy = NatProxy(nat) if nat is not None else None  # (FGMM_NATIVE=0: the ctypes binding - only the totals per call below)
if proxy is not None:
    _lib.native = lambda: proxy
print("binding:", "flashgmm_amd._native" if proxy is not None else "ctypes")

for stride in (0, 1024):
    gmc = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=stride)
    ex = P.LengthExchange(n, device=dev, threaded=True)
    rows = []
    for r in range(steps + 5):
        if proxy is not None:
            proxy.compress_stacked.t.clear(), proxy.decompress_stacked.t.clear()
        m = [now()]
        res = gmc.compress_batch(ys, ss, ms, ws)
        m.append(now())
        ex.start([len(b) for b in res.strings])
        m.append(now())
        outs = []
        for s in range(2):
            a = (res.strings[s::2], res.abs_maxes[s::2], res.zero_bitmaps[s::2], ss[s::2], ms[s::2], ws[s::2])
            m.append(now())
            outs.append(gmc.decompress_batch(*a, stacked_output=True))
            m.append(now())
        lengths = ex.wait(to_host=False)
        m.append(now())
        if r < 5:
            continue
        log = _lib.call_log(0, 3)
        lib_ms = [c["ms"][5] for c in log]
        ms_ = lambda a, b: (b - a) * 1e3
        if proxy is None:
            rows.append({
                "step": ms_(m[0], m[-1]),
                "compress: everything outside the library": ms_(m[0], m[1]) - lib_ms[0],
                "exchange start": ms_(m[1], m[2]),
                "stage 0: everything outside the library": ms_(m[2], m[4]) - lib_ms[1],
                "stage 1: everything outside the library": ms_(m[4], m[6]) - lib_ms[2],
                "exchange wait": ms_(m[6], m[7]),
                "library calls": sum(lib_ms),
                "library: compress call": lib_ms[0],
                "library: compress call, host tail": max(log[0]["ms"][4] - max(log[0]["ms"][3], log[0]["ms"][1]), 0.0),
                "library: stage 0": lib_ms[1],
                "library: stage 1": lib_ms[2],
            })
            continue
        (c0, c1), = proxy.compress_stacked.t
        (d0, d1), (e0, e1) = proxy.decompress_stacked.t
        rows.append({
            "step": ms_(m[0], m[-1]),
            "compress: python before the module": ms_(m[0], c0),
            "compress: module - library call": ms_(c0, c1) - lib_ms[0],
            "compress: python after the module": ms_(c1, m[1]),
            "exchange start": ms_(m[1], m[2]),
            "stage 0: slices": ms_(m[2], m[3]),
            "stage 0: python before the module": ms_(m[3], d0),
            "stage 0: module - library call": ms_(d0, d1) - lib_ms[1],
            "stage 0: python after the module": ms_(d1, m[4]),
            "stage 1: slices": ms_(m[4], m[5]),
            "stage 1: python before the module": ms_(m[5], e0),
            "stage 1: module - library call": ms_(e0, e1) - lib_ms[2],
            "stage 1: python after the module": ms_(e1, m[6]),
            "exchange wait": ms_(m[6], m[7]),
            "library calls": sum(lib_ms),
            "library: compress call": lib_ms[0],
            "library: compress call, host tail": max(log[0]["ms"][4] - max(log[0]["ms"][3], log[0]["ms"][1]), 0.0),
            "library: stage 0": lib_ms[1],
            "library: stage 1": lib_ms[2],
        })
    ex.close()
    print(f"\ncheckpoint_stride {stride}: medians over {len(rows)} steps, ms")
    tot = 0.0
    for k in rows[0]:
        v = float(np.median([r_[k] for r_ in rows]))
        if k != "step" and not k.startswith("library"):
            tot += v
        print(f"  {k:40s} {v:7.3f}")
    print(f"  {'sum outside the library':40s} {tot:7.3f}")
dist.destroy_process_group()

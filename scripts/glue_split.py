"""Dev aid (GPU): where the calling thread's Python time BETWEEN the native calls of a Kodak step goes (bench.py's
`step_ms.phases_ms.between_calls`: 0.5 ms on one box, 1.0 - 1.5 ms on others while every in-call phase stays the same).
The step of bench.py's headline leg with clock reads between its statements and around the three native calls:
    python scripts/glue_split.py [steps]          (FGMM_BENCH_PG=0: without the process group of one rank)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench as B
from flashgmm_amd import _lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
use_pg = os.environ.get("FGMM_BENCH_PG", "1") != "0"
dist = None
if use_pg:
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
env = B.Env(0, 1, 0, dev, dist, dev if use_pg else torch.device("cpu"), "nccl" if use_pg else None)
leg = B.Leg(env, "kodak24", 24, "polya", False)
import gc

gc.disable()
native = []  # (enter, exit) of every native batched call
L = _lib.lib()
for name in ("fgmm_gmc_compress_batch", "fgmm_gmc_decompress_batch"):
    f = getattr(L, name)

    def wrap(*a, _f=f):
        t0 = time.perf_counter()
        r = _f(*a)
        native.append((t0, time.perf_counter()))
        return r

    setattr(L, name, wrap)
for _ in range(5):
    leg.step("codec")
torch.cuda.synchronize()
rows = []
for _ in range(steps):
    native.clear()
    t = [time.perf_counter()]
    res = leg.gmc.compress_batch(leg.ys, leg.ss, leg.ms, leg.ws)
    t.append(time.perf_counter())
    if leg.ex is not None:
        leg.ex.start([len(b) for b in res.strings])
    t.append(time.perf_counter())
    outs = []
    for s in range(leg.spi):
        sp, mp_, wp = leg.stage_params[s]
        outs.append(leg.gmc.decompress_batch(res.strings[s::leg.spi], res.abs_maxes[s::leg.spi], res.zero_bitmaps[s::leg.spi], sp, mp_, wp, stacked_output=True))
        t.append(time.perf_counter())
    if leg.ex is not None:
        leg.ex.wait(to_host=False)
    t.append(time.perf_counter())
    (e0, e1), (d0, d1), (f0, f1) = native
    rows.append({
        "step": t[-1] - t[0], "native": (e1 - e0) + (d1 - d0) + (f1 - f0),
        "encode: before the native call": e0 - t[0], "encode: after it (bytes, side information)": t[1] - e1,
        "lengths exchange: start": t[2] - t[1],
        "decode 1: before": d0 - t[2], "decode 1: after": t[3] - d1,
        "decode 2: before": f0 - t[3], "decode 2: after": t[4] - f1,
        "lengths exchange: wait": t[5] - t[4]})
print(f"{steps} steps, process group of one rank: {use_pg}")
for k in rows[0]:
    v = np.asarray([r[k] for r in rows]) * 1e3
    print(f"  {k:46s} median {np.median(v):7.3f}  p90 {np.percentile(v, 90):7.3f}  max {v.max():7.3f} ms")
glue = np.asarray([r["step"] - r["native"] for r in rows]) * 1e3
print(f"  {'all Python between / around the calls':46s} median {np.median(glue):7.3f}  p90 {np.percentile(glue, 90):7.3f}  max {glue.max():7.3f} ms")
if dist is not None:
    dist.destroy_process_group()

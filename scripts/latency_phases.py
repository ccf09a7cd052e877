"""Dev aid (GPU): one Kodak image as the codec schedules it (encode both halves, decode the anchors, decode the non-anchors) - where its
3.5 - 3.9 ms go, from the library's call log.   python scripts/latency_phases.py [repeats]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench as B
from flashgmm_amd import GaussianMixtureConditional, _lib
from flashgmm_amd import parallel as P

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
print(P.bind_to_gpu_numa_node(0))
mine, note = P.plan_l3(0, 1)
print(note)
host, devt, pix = B.make_workload(0, 1, dev, "kodak24", False)
gmc = GaussianMixtureConditional(K=4, mode="polya")
ys, ss, ms, ws = (torch.cat([t[k] for t in devt]) for k in range(4))
if mine and os.environ.get("LATENCY_NO_PIN") != "1":  # (LATENCY_NO_PIN=1: the calling thread stays on the process's CPUs)
    os.sched_setaffinity(0, mine)
import gc

gc.disable()
rows = []
for r in range(reps + 5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = gmc.compress_batch(ys, ss, ms, ws)
    outs = [gmc.decompress_batch(res.strings[s::2], res.abs_maxes[s::2], res.zero_bitmaps[s::2], ss[s::2], ms[s::2], ws[s::2], stacked_output=True) for s in range(2)]
    t1 = time.perf_counter()
    if r >= 5:
        ph = B.D.call_phases(_lib.call_log(0, 3))
        ph["total"] = (t1 - t0) * 1e3
        ph["between_calls"] = ph["total"] - sum(c["ms"][5] for c in _lib.call_log(0, 3))
        rows.append(ph)
for k in sorted(rows[0]):
    print(f"  {k:34s} median {np.median([r_[k] for r_ in rows]):7.3f} ms")

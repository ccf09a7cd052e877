"""Dev aid (GPU): the ELIC-4K step (bench.py --workload elic4k) split into its encode call and its ten stage-major decode
calls, plain or on checkpointed streams.   CKPT=1024 python scripts/elic_split.py [images] [name=value ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as B
from flashgmm_amd import GaussianMixtureConditional, _lib


def main():
    dev = torch.device("cuda:0")
    images = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
    for kv in sys.argv[1:]:
        if "=" in kv:
            _lib.set_option(0, kv.split("=")[0], int(kv.split("=")[1]))
    shapes1, _ = B.workload_shapes("elic4k")
    host, devt, pix = B.make_workload(0, images, dev, "elic4k", True)
    spi = len(shapes1)
    ys, ss, ms, ws = ([t[k] for t in devt] for k in range(4))
    gmc = GaussianMixtureConditional(K=4, mode="polya", checkpoint_stride=int(os.environ.get("CKPT", "0")))
    n = len(devt)
    def step():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = gmc.compress_batch(ys, ss, ms, ws)
        t1 = time.perf_counter()
        td = []
        for s in range(spi):
            idx = list(range(s, n, spi))
            ta = time.perf_counter()
            gmc.decompress_batch([res[i][0][0] for i in idx], [res[i][0][1] for i in idx], [res[i][0][2] for i in idx], ss[s::spi], ms[s::spi], ws[s::spi])
            td.append((time.perf_counter() - ta) * 1e3)
        torch.cuda.synchronize()
        return (t1 - t0) * 1e3, td, (time.perf_counter() - t0) * 1e3
    step()
    r = [step() for _ in range(4)]
    enc = np.median([x[0] for x in r]); tot = np.median([x[2] for x in r]); td = np.median(np.array([x[1] for x in r]), axis=0)
    syms = [sum(int(np.prod(devt[i][0].shape)) for i in range(s, n, spi)) for s in range(spi)]
    print(f"{images} images, CKPT={os.environ.get('CKPT', '0')}: step {tot:.2f} ms = encode {enc:.2f} + decode {td.sum():.2f}")
    print("decode calls ms:", np.round(td, 2).tolist())
    print("M symbols per call:", [round(x / 1e6, 2) for x in syms], " ns/symbol:", [round(t * 1e6 / x, 3) for t, x in zip(td, syms)])


if __name__ == "__main__":  # (spawned workload helpers import this file: nothing at import time)
    main()

"""Dev aid (GPU): randomized stress of the decode scheduler - random batches (1..40 items, tiny to Kodak-sized, some empty,
occasionally one truncated bitstream, plain or checkpointed streams) under random pipeline options and worker counts; every result is checked against the
encoder's y_q, every truncated batch must raise.  python scripts/stress_schedule.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
shapes = [(192, 32, 24), (24, 8, 6), (5, 3, 3), (1, 1, 2), (64, 16, 12), (17, 1, 1), (192, 16, 24)]
pool = {}
def latent(seed, shp, zf):
    key = (seed, shp, zf)
    if key not in pool:
        y, sg, mu, pi = T.make_latent(seed, M=shp[0], h=shp[1], w=shp[2], zero_frac=zf)
        pool[key] = tuple(torch.from_numpy(a).to(dev) for a in (y, sg, mu, pi))
    return pool[key]
t_end = time.time() + budget
rounds = errors_expected = 0
while time.time() < t_end:
    threads = int(rng.choice([1, 2, 3, 8, 16, 48]))
    _lib.set_threads(0, threads)
    opts = dict(pieces=int(rng.integers(0, 33)), dec_first=int(rng.integers(1, 10)),
                ef_min=int(rng.choice([14, 33, 49, 200])), ef_rows=int(rng.integers(0, 3)),
                enc_ways=int(rng.integers(0, 5)), ckpt_decode=int(rng.integers(0, 3)), gpu_decode=int(rng.choice([0, 2])),
                enc_segs=int(rng.integers(0, 3)), scatter_rounds=int(rng.integers(0, 2)))
    ck_stride = int(rng.choice([0, 0, 256, 1024, 4096]))  # checkpointed streams: segments on the workers
    for k, v in opts.items(): _lib.set_option(0, k, v)
    for mode in ("polya", "as", "logistic"):
        gmc = GaussianMixtureConditional(K=4, mode=mode, checkpoint_stride=ck_stride)
        n = int(rng.integers(1, 41))
        items = [latent(int(rng.integers(0, 12)), shapes[int(rng.choice(len(shapes), p=[.3, .2, .1, .1, .1, .1, .1]))], float(rng.choice([0.0, 0.1, 1.0], p=[.2, .6, .2])))
                 for _ in range(n)]
        ys, ss, ms, ws = ([it[k] for it in items] for k in range(4))
        res = gmc.compress_batch(ys, ss, ms, ws)
        strings = [r[0][0] for r in res]
        outs = gmc.decompress_batch(strings, [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
        for i in range(n):
            assert torch.equal(outs[i], res[i][1]), (opts, threads, mode, i)
        victims = [i for i in range(n) if len(strings[i]) > 64]
        if victims and rng.random() < 0.5:
            bad = list(strings); v = int(rng.choice(victims))
            bad[v] = type(bad[v])(bad[v][:16], bad[v].ckpt, bad[v].ckpt_stride) if ck_stride and rng.random() < 0.5 else bytes(bad[v][:16])
            try:
                gmc.decompress_batch(bad, [r[0][1] for r in res], [r[0][2] for r in res], ss, ms, ws)
                raise SystemExit(f"truncated stream decoded: {opts} {threads} {mode}")
            except RuntimeError:
                errors_expected += 1
        rounds += 1
    if rounds % 30 == 0: print(f"{rounds} batches ok ({errors_expected} refused as they must)", flush=True)
_lib.set_threads(0, 0)
print(f"done: {rounds} batches, {errors_expected} truncated batches refused")

"""Dev aid (GPU): the codec schedule with TWO image groups in flight - each group honours its own dependency chain
(anchors, then non-anchors), the two chains run on two contexts / streams / Python threads and fill each other's bubbles.
python scripts/two_lanes.py [threads_per_lane]"""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashgmm_amd import GaussianMixtureConditional, _lib, testing as T
dev = torch.device("cuda:0")
tpl = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lat = [T.make_latent(i) for i in range(48)]
ys, ss, ms, ws = (torch.cat([torch.from_numpy(l[k]) for l in lat]).to(dev) for k in range(4))
gmc = GaussianMixtureConditional(K=4, mode="polya")
L = _lib.lib()
tls = threading.local()
base_ctx = _lib.ctx(0)
orig_ctx = _lib.ctx
_lib.ctx = lambda device=-1, n_threads=0: getattr(tls, "ctx", base_ctx)
lanes = []
for k in range(2):
    h = C.c_void_p()
    _lib.check(L.fgmm_ctx_create(0, tpl, C.byref(h)))
    lanes.append((h, torch.cuda.Stream()))
# images 0..11 -> lane 0, 12..23 -> lane 1; stream index = 2 * image + half
idx = [[[2 * im + s for im in range(12 * k, 12 * k + 12)] for s in range(2)] for k in range(2)]
sel = [[[torch.tensor(ix, device=dev) for ix in lane] for lane in idx]][0]
prm = [[(ss[ix], ms[ix], ws[ix]) for ix in lane] for lane in sel]  # gathered once (stand-in for the network's outputs)
def lane_decode(k, res, out):
    tls.ctx = lanes[k][0]
    with torch.cuda.stream(lanes[k][1]):
        for s in range(2):
            ix = idx[k][s]
            out[(k, s)] = gmc.decompress_batch([res[i][0][0] for i in ix], [res[i][0][1] for i in ix], [res[i][0][2] for i in ix], *prm[k][s])
        lanes[k][1].synchronize()
def step(two):
    t0 = time.perf_counter()
    res = gmc.compress_batch(ys, ss, ms, ws)
    t1 = time.perf_counter()
    out = {}
    if two:
        th = [threading.Thread(target=lane_decode, args=(k, res, out)) for k in range(2)]
        for t in th: t.start()
        for t in th: t.join()
    else:
        for s in range(2):
            ix = range(s, 48, 2)
            out[s] = gmc.decompress_batch([res[i][0][0] for i in ix], [res[i][0][1] for i in ix], [res[i][0][2] for i in ix], ss[s::2], ms[s::2], ws[s::2])
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, res, out
# correctness of the two-lane result
e, d, res, out = step(True)
for k in range(2):
    for s in range(2):
        for j, i in enumerate(idx[k][s]):
            assert torch.equal(out[(k, s)][j], res[i][1]), (k, s, j)
import gc; gc.disable()
times = {True: [], False: []}
for rnd in range(6):
    for two in (False, True):
        step(two)
        for _ in range(5): times[two].append(step(two)[:2])
for two in (False, True):
    e, d = np.array(times[two]).T
    print(f"{'two lanes' if two else 'one lane '} threads/lane {tpl}: encode {np.median(e):6.3f}  decode {np.median(d):6.3f}  step median {np.median(e + d):6.3f} ms (min {np.min(e + d):.3f})")

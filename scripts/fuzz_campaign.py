"""Dev aid: a longer randomized parity campaign than the test suite runs (single items and batches, all modes, both
parameter dtypes, clamped and raw sigma, outliers) against the oracle.  Usage: python scripts/fuzz_campaign.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import oracle as O
from flashgmm_amd import GaussianMixtureConditional, _lib
from tests import synth as T
MODES = ["polya", "as", "logistic"]
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(987654321)
n_items = n_syms = n_gpu = n_back = 0
for r in range(rounds):
    mode = MODES[r % 3]
    f16 = bool(rng.integers(0, 2))
    clamp = bool(rng.integers(0, 4) != 0)
    count = int(rng.choice([1, 2, 3, 5, 9, 20]))
    same_hw = bool(rng.integers(0, 2))
    h0, w0 = int(rng.choice([1, 3, 8, 16, 32])), int(rng.choice([2, 5, 8, 16, 24, 32]))
    lat, shapes = [], []
    for i in range(count):
        M = int(rng.integers(1, 33))
        h, w = (h0, w0) if same_hw else (int(rng.integers(1, 25)), int(rng.integers(1, 25)))
        y, sg, mu, pi = T.make_latent(int(rng.integers(0, 1 << 30)), M=M, h=h, w=w, clamp=False, zero_frac=float(rng.choice([0, 0.2, 0.7])))
        if rng.integers(0, 5) == 0:
            y = y.copy(); y.reshape(-1)[rng.integers(0, y.size, max(1, y.size // 40))] *= 60
        if rng.integers(0, 6) == 0:
            sg = (sg * np.float32(rng.choice([4.0, 12.0]))).astype(np.float32)  # wide windows
            y = (y * np.float32(4.0)).astype(np.float32)
        if not clamp:
            sg = np.maximum(sg, np.float32(0.02))
        if f16:
            sg, mu, pi = T.to_float16_planes(sg, mu, pi)
        lat.append((y, sg, mu, pi))
    gmc = GaussianMixtureConditional(K=4, mode=mode, clamp_scales=clamp)
    ys, ss, ms, ws = ([dv(l[k]) for l in lat] for k in range(4))
    res = gmc.compress_batch(ys, ss, ms, ws)
    ok_dec = True
    for i, l in enumerate(lat):
        sym, s, m, wt, am, zbm, yqn = T.to_coder_inputs(l[0], *(a.astype(np.float32) for a in l[1:]), clamp=clamp)
        (b, abs_max, zb), yq = res[i]
        assert b == O.encode_gmm(mode, sym, s, m, wt) and abs_max == am and zb.tolist() == zbm.tolist(), (r, i, mode, f16, clamp)
        ok_dec = ok_dec and abs_max + 1 <= 16382
        n_items += 1; n_syms += len(sym)
    if ok_dec:
        outs = gmc.decompress_batch([x[0][0] for x in res], [x[0][1] for x in res], [x[0][2] for x in res], ss, ms, ws)
        for i in range(count):
            assert torch.equal(outs[i], res[i][1]), (r, i, "decode")
        # the same items as checkpointed streams: the same bytes, decoded by the GPU's segment decoder (forced: the items are small),
        # by the host workers' segments, or by the library's choice - the sequential decoder's result every time
        stride = int(rng.choice([256, 512, 1024]))
        how = int(rng.integers(0, 3))
        # (wide sigmas now and then: windows beyond 63 edges take the kernel's 64-edges-per-pass loop)
        ck = GaussianMixtureConditional(K=4, mode=mode, clamp_scales=clamp, checkpoint_stride=stride)
        rc = ck.compress_batch(ys, ss, ms, ws)
        assert all(bytes(a[0][0]) == bytes(b[0][0]) for a, b in zip(rc, res)), (r, "checkpointed bytes")
        _lib.set_option(0, "gpu_decode", how)
        try:
            outs = ck.decompress_batch([x[0][0] for x in rc], [x[0][1] for x in rc], [x[0][2] for x in rc], ss, ms, ws)
        finally:
            _lib.set_option(0, "gpu_decode", 0)
        n_gpu += _lib.ctx_stat(0, 4); n_back += _lib.ctx_stat(0, 5)
        for i in range(count):
            assert torch.equal(outs[i], res[i][1]), (r, i, "checkpointed decode", stride, how)
print(f"fuzz campaign ok: {rounds} rounds, {n_items} items, {n_syms} symbols; {n_gpu} bitstreams decoded by segdec_kernel, {n_back} handed back to the table path")

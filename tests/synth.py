"""Seeded synthetic latents for tests and bench (SURVEY.md §8c KA-1 / §8d recipe).

No Kodak files or trained checkpoint exist offline, so every workload is generated from
``numpy.random.Generator(PCG64(seed))`` in a fixed draw order.  numpy only — never torch RNG — so the
same arrays can be regenerated in any process, on CPU or on the GPU box.

Layout returned (the latent codec's layout, latent_codecs/gaussian_mixture_conditional.py:193-195):
    y       float32 [1, M, h, w]
    scales  float32 [1, K*M, h, w]   channel index = k*M + c   (PRE-clamp unless clamp=True)
    means   float32 [1, K*M, h, w]
    weights float32 [1, K*M, h, w]   softmax over k
"""
from __future__ import annotations

import numpy as np


def make_latent(seed: int, M: int = 192, h: int = 32, w: int = 24, K: int = 4, *, clamp: bool = True,
                zero_frac: float = 0.0):
    """KA-1 draw order: e_c, y, mu, sigma, logits.

    ``e_c`` is cast to float32 immediately; every other right-hand side is evaluated in float64 and cast to
    float32 once at the end.  ``clamp=True`` reproduces KA-1 (sigma already clipped to [0.11, 256]);
    ``clamp=False`` leaves sigma un-clamped so the entropy model's own clamp (entropy_models.py:817) is exercised.
    ``zero_frac`` additionally forces that fraction of channels to all-zero y (zero_bitmap coverage).
    """
    rng = np.random.default_rng(seed)
    e_c = np.exp(rng.uniform(-3, 2.5, M)).astype(np.float32)
    y = (rng.standard_normal((M, h, w)) * 1.5 * e_c[:, None, None]).astype(np.float32)
    mu = (rng.standard_normal((K, M, h, w)) * e_c[None, :, None, None]).astype(np.float32)
    sg = (rng.uniform(0, 2, (K, M, h, w)) + 0.05) * e_c[None, :, None, None]
    if clamp:
        sg = np.clip(sg, 0.11, 256)
    sg = sg.astype(np.float32)
    lg = rng.standard_normal((K, M, h, w))
    pi = (np.exp(lg) / np.exp(lg).sum(0)).astype(np.float32)
    if zero_frac > 0:
        kill = rng.uniform(0, 1, M) < zero_frac
        y[kill] = (rng.uniform(-0.49, 0.49, (int(kill.sum()), h, w))).astype(np.float32)
    return (
        y.reshape(1, M, h, w),
        sg.reshape(1, K * M, h, w),
        mu.reshape(1, K * M, h, w),
        pi.reshape(1, K * M, h, w),
    )


def to_coder_inputs(y, scales, means, weights, K: int = 4, clamp: bool = True):
    """numpy restatement of what GaussianMixtureConditional.compress hands the coder
    (entropy_models.py:834-846, :810-828): (symbols int32[n], scales/means/weights (n,K) views with strides
    (1, n) elements, abs_max, zero_bitmap int64[M], y_q)."""
    B, M, h, w = y.shape
    assert B == 1
    ymax, ymin = float(y.max()), float(y.min())
    # torch.abs(y.max()).int().item(): truncation toward zero of |max|, |min|   (:834-837)
    abs_max = max(int(abs(ymax)), int(abs(ymin))) + 1
    abs_max = 1 if abs_max < 1 else abs_max
    yq = np.round(y)  # round-half-even, as torch.round
    zero_bitmap = (np.abs(yq).sum((3, 2))[0] != 0).astype(np.int64)
    nz = np.nonzero(zero_bitmap)[0]
    symbols = yq[0, nz].reshape(-1).astype(np.int32)

    def rs(p):
        return p.reshape(K, M, h * w)[:, nz].reshape(K, -1).T  # (n, K) view-like, strides (1, n) after copy

    s, m, wt = rs(scales), rs(means), rs(weights)
    if clamp:
        s = np.clip(s, np.float32(0.11), np.float32(256))
    return symbols, s, m, wt, abs_max, zero_bitmap, yq


def to_float16_planes(scales, means, weights):
    """fp16 copies of the parameter planes for BASELINE configs[4].  Weights are rounded TOWARD ZERO: the reference
    algorithm needs sum_k pi_k <= 1 after widening (a quantised CDF edge above 65535 wraps, rans_interface.cpp:509-512,
    and the stream desynchronises — in the reference exactly as here); round-to-nearest fp16 weights can sum to more."""
    w16 = weights.astype(np.float16)
    over = w16.astype(np.float32) > weights
    w16[over] = np.nextafter(w16[over], np.float16(0))
    return scales.astype(np.float16), means.astype(np.float16), w16


# ---------------------------------------------------------------------------------------------------------------
# Networks whose outputs are the same bits on every device, for parity tests of the codecs that sit ABOVE the
# entropy model (CheckerboardLatentCodec, ChannelGroupsLatentCodec): a real convolution or softmax differs in the
# last bits between a CPU and a GPU, and the bitstream depends on every bit of (sigma, mu, pi).  Everything below is
# exact arithmetic: dyadic constants, shifts, sums of a few exactly representable terms, and mixture logits in
# {0, -inf} with 1, 2 or 4 active components (softmax = 1, 1/2, 1/4 exactly).  The reference's own codec classes
# run with these modules on the CPU when the golden vectors are made (tests/golden/make_golden.py), ours on the GPU.
# ---------------------------------------------------------------------------------------------------------------
def exact_modules():
    """-> (ExactContext, ExactParams) classes (torch imported lazily: this module stays numpy-only otherwise)."""
    import torch
    import torch.nn as nn

    class ExactContext(nn.Module):
        """[n, c_in, h, w] -> [n, c_out, h, w]: out[o] = x[o % c_in] shifted down / 2 + x[(o+1) % c_in] shifted right / 4
        - x[(3o+2) % c_in] shifted left / 4 (zero fill).  Exact for integer-valued x."""

        def __init__(self, c_in: int, c_out: int):
            super().__init__()
            self.c_in, self.c_out = c_in, c_out

        def forward(self, x):
            n, c, h, w = x.shape
            down = torch.zeros_like(x); down[:, :, 1:, :] = x[:, :, :-1, :]
            right = torch.zeros_like(x); right[:, :, :, 1:] = x[:, :, :, :-1]
            left = torch.zeros_like(x); left[:, :, :, :-1] = x[:, :, :, 1:]
            o = torch.arange(self.c_out, device=x.device)
            return 0.5 * down[:, o % c] + 0.25 * right[:, (o + 1) % c] - 0.25 * left[:, (3 * o + 2) % c]

    class ExactParams(nn.Module):
        """pointwise [n, c_in, h, w] -> [n, 3*K*M, h, w] = [scales | means | logits], channel k*M + c each."""

        def __init__(self, c_in: int, M: int, K: int = 4):
            super().__init__()
            self.c_in, self.M, self.K = c_in, M, K

        def forward(self, x):
            n, c_in, h, w = x.shape
            M, K = self.M, self.K
            c = torch.arange(M, device=x.device)
            scales, means, logits = [], [], []
            a = x[:, (5 * c + 1) % c_in]  # decides how many components are active
            b = x[:, (7 * c + 3) % c_in]
            ninf = torch.full_like(a, float("-inf"))
            zero = torch.zeros_like(a)
            for k in range(K):
                xin = x[:, (c + k) % c_in]
                scales.append(0.0625 + 0.25 * xin.abs() + 0.125 * k)  # some below the 0.11 clamp
                means.append(0.5 * x[:, (c + 2 * k) % c_in] + (k - 1.5))
                if k == 0:
                    logits.append(zero)
                elif k == 1:
                    logits.append(torch.where(a > 0, zero, ninf))
                else:
                    logits.append(torch.where((a > 0) & (b > 0), zero, ninf))
            return torch.cat(scales + means + logits, dim=1)

    return ExactContext, ExactParams


def exact_hyper_modules():
    """-> (ExactHa, ExactHs): the hyper branch's transforms as exact arithmetic (see exact_modules).
    ExactHa  y [n, c, h, w] -> z [n, cz, h/2, w/2]: z[o] = round(2 * (y[o % c] + y[(o+1) % c]) sub-sampled by 2) / 2 + (o % 3 - 1) / 2
             — multiples of 1/2 whatever y is, some beyond the bottleneck's tables (bypass-coded) (round() of an IEEE sum: the same bits on every device)
    ExactHs  z_hat [n, cz, h/2, w/2] -> side [n, c_side, h, w]: nearest-neighbour x2 of 0.5 * z_hat[o % cz] + 0.25 * (o % 5 - 2)"""
    import torch
    import torch.nn as nn

    class ExactHa(nn.Module):
        def __init__(self, c: int, cz: int):
            super().__init__()
            self.c, self.cz = c, cz

        def forward(self, y):
            o = torch.arange(self.cz, device=y.device)
            t = (y[:, o % self.c] + y[:, (o + 1) % self.c])[:, :, ::2, ::2]
            return torch.round(2.0 * t) / 2.0 + ((o % 3).to(y.dtype) - 1.0).view(1, -1, 1, 1) / 2.0

    class ExactHs(nn.Module):
        def __init__(self, cz: int, c_side: int):
            super().__init__()
            self.cz, self.c_side = cz, c_side

        def forward(self, z):
            o = torch.arange(self.c_side, device=z.device)
            t = 0.5 * z[:, o % self.cz] + 0.25 * ((o % 5).to(z.dtype) - 2.0).view(1, -1, 1, 1)
            return t.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)

    return ExactHa, ExactHs


def exact_codec_inputs(seed: int, c: int, c_side: int, h: int, w: int, dead: int = 0):
    """y float32 [1, c, h, w] (arbitrary floats; `dead` leading channels within +-0.4 so that they quantise to all-zero),
    side_params float32 [1, c_side, h, w] (multiples of 1/4, so that everything derived from them stays exact)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    y = (rng.standard_normal((1, c, h, w)) * rng.uniform(0.5, 6.0, (1, c, 1, 1))).astype(np.float32)
    if dead:
        y[:, :dead] = rng.uniform(-0.4, 0.4, (1, dead, h, w)).astype(np.float32)
    side = (rng.integers(-12, 13, (1, c_side, h, w)) / 4.0).astype(np.float32)
    return y, side


def make_head(seed, M, c_in, h, w, N, dead=0, dev="cuda:0"):
    """-> (conv, x [N, c_in, h, w], y [N, M, h, w]): a random 1x1 convolution of the parameter head's shape (nn.Conv2d(c_in, 3*K*M, 1),
    compressai/models/ckbd_gmm.py:115-121) whose outputs look like entropy parameters - sigma and means of the order of a per-channel
    energy e_c, logits of order one - and features / latents to go with it (tests/test_gpu_head.py, bench.py's head_fused leg)"""
    import torch

    rng = np.random.default_rng(seed)
    e_c = np.exp(rng.uniform(-2.5, 2.0, M)).astype(np.float32)
    W = (rng.standard_normal((3, 4, M, c_in)) / np.sqrt(c_in)).astype(np.float32)
    b = np.zeros((3, 4, M), np.float32)
    W[0] *= 0.3 * e_c[None, :, None]
    b[0] = (0.6 + 0.5 * rng.uniform(0, 1, (4, M))) * e_c[None, :]  # sigma: mostly positive, sometimes under the clamp
    W[1] *= e_c[None, :, None]
    conv = torch.nn.Conv2d(c_in, 12 * M, 1)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(W.reshape(12 * M, c_in, 1, 1)))
        conv.bias.copy_(torch.from_numpy(b.reshape(-1)))
    x = rng.standard_normal((N, c_in, h, w)).astype(np.float32)
    x = np.where(x > 0, x, 0.01 * x)  # (the layer before is a LeakyReLU)
    y = (rng.standard_normal((N, M, h, w)) * 1.5 * e_c[None, :, None, None]).astype(np.float32)
    if dead:
        y[:, rng.choice(M, dead, replace=False)] *= 0.0
    return conv.to(dev), torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)

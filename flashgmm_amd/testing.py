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

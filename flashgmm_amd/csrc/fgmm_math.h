// fgmm_math.h — device-side fp32 arithmetic of the GMM CDF, CDNA4 (gfx950).
//
// Every function is a fixed sequence of single IEEE-754 binary32 round-to-nearest-even operations that
// reproduces, bit for bit, what the reference's default (AVX, K == 4) path computes after GCC's FMA
// contraction (SURVEY.md §8a'; reference: compressai/cpp_exts/rans/rans_interface.cpp:133-292,
// compressai/cpp_exts/rans/avx_mathfun.h:250-304).  Consequences for how this file must be built and written:
//   * compile with -ffp-contract=off: the ONLY fused operations are the explicit __builtin_fmaf calls;
//   * '/' and __builtin_sqrtf must be correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt, hipcc's
//     default) — never __fdividef / v_rcp_f32 / v_exp_f32 fast paths;
//   * f32 denormals stay enabled (hipcc's default float mode on gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fgmm {

enum : int { MODE_POLYA = 0, MODE_AS = 1, MODE_LOGISTIC = 2 };

__device__ __forceinline__ float bits2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2bits(float f) { return __float_as_uint(f); }

// exp256_ps as compiled (avx_mathfun.h:250-304): clamp, n = floor(x*log2e + 0.5) by one FMA, two-constant
// Cody-Waite by FMAs, degree-5 Horner by FMAs, y = fma(y, x*x, x) + 1, times the BIT PATTERN 2^n
// ((n+127)<<23: n = -127 -> +0.0, n = 128 -> +inf; this is not ldexp).
__device__ __forceinline__ float exp_ref(float x) {
  x = (x < 88.3762626647949f) ? x : 88.3762626647949f;   // _mm256_min_ps(x, hi): hi when x is NaN
  x = (x > -88.3762626647949f) ? x : -88.3762626647949f; // _mm256_max_ps(x, lo)
  float fx = __builtin_fmaf(x, 1.44269504088896341f, 0.5f);
  fx = __builtin_floorf(fx);
  x = __builtin_fmaf(-fx, 0.693359375f, x);
  x = __builtin_fmaf(-fx, -2.12194440e-4f, x);
  const float z = x * x;
  float y = 1.9875691500E-4f;
  y = __builtin_fmaf(y, x, 1.3981999507E-3f);
  y = __builtin_fmaf(y, x, 8.3334519073E-3f);
  y = __builtin_fmaf(y, x, 4.1665795894E-2f);
  y = __builtin_fmaf(y, x, 1.6666665459E-1f);
  y = __builtin_fmaf(y, x, 5.0000001201E-1f);
  y = __builtin_fmaf(y, z, x);
  y = y + 1.0f;
  const int n = (int)fx; // |fx| <= 128 after the clamps
  return y * bits2f((uint32_t)(n + 127) << 23);
}

// exp_ref for an argument known to be <= 0 (or -0) and not NaN — what Polya's and A&S's exp always get when z is
// finite.  Same value with two operations fewer: the upper clamp cannot act, and y * 2^n (n in [-127, 0]) is
// v_ldexp_f32 except at n = -127, where the reference's bit pattern is +0 and ldexp gives a denormal < 2^-126:
// both callers only use e in 1 - e resp. fma(-c*e, poly, 1) with |c*poly| < 1, which is 1.0f either way.
// Phi<MODE, true> == Phi<MODE, false> is checked for EVERY binary32 |z| < 2^48 by fgmm_selftest_fastmath(3..5).
__device__ __forceinline__ float exp_nonpos(float x) {
  x = (x > -88.3762626647949f) ? x : -88.3762626647949f;
  float fx = __builtin_fmaf(x, 1.44269504088896341f, 0.5f);
  fx = __builtin_floorf(fx);
  x = __builtin_fmaf(-fx, 0.693359375f, x);
  x = __builtin_fmaf(-fx, -2.12194440e-4f, x);
  const float z = x * x;
  float y = 1.9875691500E-4f;
  y = __builtin_fmaf(y, x, 1.3981999507E-3f);
  y = __builtin_fmaf(y, x, 8.3334519073E-3f);
  y = __builtin_fmaf(y, x, 4.1665795894E-2f);
  y = __builtin_fmaf(y, x, 1.6666665459E-1f);
  y = __builtin_fmaf(y, x, 5.0000001201E-1f);
  y = __builtin_fmaf(y, z, x);
  y = y + 1.0f;
  return __builtin_ldexpf(y, (int)fx);
}

// ---------------------------------------------------------------------------------------------------------
// Correctly-rounded division and square root, hand-expanded.
//
// hipcc lowers an IEEE f32 '/' to  v_div_scale x2, v_rcp, 4 fma, mul, v_div_fmas, v_div_fixup  (and f32 sqrt to
// v_sqrt + two fma probes + scaling + class fix-ups).  The scale / fix-up steps only act when an exponent is
// extreme or an operand is 0/inf/NaN.  Where the operands are known to be tame the core below computes the SAME
// sequence of operations — hence the same, correctly rounded, bits — in 5 instructions per quotient once the
// refined reciprocal of the denominator exists; and the denominator (sigma_k) is shared by every abscissa that
// is evaluated for a latent.  Callers guard the domain and fall back to '/' outside it.
// Equality with '/' and sqrtf is checked on the GPU by fgmm_selftest_fastmath (tests/test_gpu_parity.py).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rcp_refined(float d) { // Fma1 of the lowering: rcp + one Newton step
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r0, 1.0f);
  return __builtin_fmaf(e, r0, r0);
}
// a / d given r = rcp_refined(d); exact for a in {+0} U +-[2^-60, 2^60), d in [2^-10, 2^60) (no scaling needed).
// (a = -0 would give +0 instead of -0; a = x - mu with x = v -+ 0.5 is never -0.)
__device__ __forceinline__ float div_core(float a, float d, float r) {
  const float q0 = a * r;
  const float e2 = __builtin_fmaf(-d, q0, a);
  const float q1 = __builtin_fmaf(e2, r, q0);
  const float e3 = __builtin_fmaf(-d, q1, a);
  return __builtin_fmaf(e3, r, q1);
}
__device__ __forceinline__ bool tame(float a) { return __builtin_fabsf(a) < 0x1p60f; } // false for NaN / inf

// a / d for a clamped denominator (sigma in [0.11, 256], or NaN which propagates either way) — guarded form
__device__ __forceinline__ float div_clamped(float a, float d, float r) {
  float q = div_core(a, d, r);
  if (__builtin_expect(!tame(a), 0)) q = a / d;
  return q;
}
// 1 / d for d >= 1 (d = 1 + something non-negative), any magnitude, NaN — guarded form
__device__ __forceinline__ float rcp_ge1(float d) {
  float q = div_core(1.0f, d, rcp_refined(d));
  if (__builtin_expect(!tame(d), 0)) q = 1.0f / d;
  return q;
}
// 1 / d for 1 <= d < 2^60 — unguarded (the caller has established the bound)
__device__ __forceinline__ float rcp_ge1_tame(float d) { return div_core(1.0f, d, rcp_refined(d)); }
// sqrt for x in {+0} U [2^-96, 2^96], negative or NaN (-> NaN): v_sqrt_f32 is within 1 ulp, the two fma probes
// pick the correctly rounded neighbour (the un-scaled core of hipcc's own lowering)
__device__ __forceinline__ float sqrt_core(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float sd = bits2f(f2bits(s) - 1u), su = bits2f(f2bits(s) + 1u);
  const float vp = __builtin_fmaf(-sd, s, x), vs = __builtin_fmaf(-su, s, x);
  s = (vp <= 0.0f) ? sd : s;
  s = (vs > 0.0f) ? su : s;
  return s;
}

// sqrt for x in {+0} U [2^-24, 1] (all Polya's 1 - e can be on the guarded path): Markstein's final step — from
// y ~ 1/sqrt(x) (v_rsq_f32), s0 = x*y and h = y/2, one fused correction s0 + (x - s0^2) * h.  Two instructions fewer
// than the probe form; equality with sqrtf over the whole domain is checked by fgmm_selftest_fastmath(6).
__device__ __forceinline__ float sqrt_unit(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y, h = 0.5f * y;
  const float r = __builtin_fmaf(-s0, s0, x);
  const float s = __builtin_fmaf(r, h, s0);
  return (x == 0.0f) ? 0.0f : s;
}

// FAST = false: plain IEEE '/' and sqrtf, any input.  FAST = true: the cores above, for finite |z| < 2^48 (the
// callers' guard): identical results, checked exhaustively over that domain (fgmm_selftest_fastmath 3..5).
template <int MODE, bool FAST> struct Phi;

// Polya/Watterson, rans_interface.cpp:135-146
template <bool FAST> struct Phi<MODE_POLYA, FAST> {
  static __device__ __forceinline__ float eval(float z) {
    const float c = -2.0f / 3.14159265358979323846f; // folded in binary32, as the reference's constant is
    const float e = FAST ? exp_nonpos(c * (z * z)) : exp_ref(c * (z * z));
    // 1 - e is +0, >= 2^-24, or (NaN path: e = exp(+88.4)) hugely negative -> NaN: always in sqrt_core's domain
    float s = FAST ? sqrt_unit(1.0f - e) : __builtin_sqrtf(1.0f - e); // FAST: e in [0, 1], so 1 - e in {0} U [2^-24, 1]
    s = bits2f((f2bits(z) & 0x80000000u) | (f2bits(s) & 0x7fffffffu)); // copysign_ps(z, s)
    // 0.5 * (1 + s): the product is an exact scaling (1 + s is 0 or >= 2^-24), so one fma rounds the same way
    return FAST ? __builtin_fmaf(0.5f, s, 0.5f) : 0.5f * (1.0f + s);
  }
};

// Abramowitz & Stegun 26.2.17, rans_interface.cpp:154-186
template <bool FAST> struct Phi<MODE_AS, FAST> {
  static __device__ __forceinline__ float eval(float z) {
    const float az = bits2f(f2bits(z) & 0x7fffffffu);
    const float zx = 0.3989422804014327f * (FAST ? exp_nonpos((z * z) * -0.5f) : exp_ref((z * z) * -0.5f));
    const float d = __builtin_fmaf(0.2316419f, az, 1.0f);
    // FAST callers guarantee |z| < 2^48 (|x - mu| < 2^40, sigma >= 0.11), hence d < 2^60
    const float t = FAST ? rcp_ge1_tame(d) : 1.0f / d;
    float poly = __builtin_fmaf(1.330274429f, t, -1.821255978f);
    poly = __builtin_fmaf(poly, t, 1.781477937f);
    poly = __builtin_fmaf(poly, t, -0.356563782f);
    poly = __builtin_fmaf(poly, t, 0.319381530f);
    poly = poly * t;
    const float res_pos = __builtin_fmaf(-zx, poly, 1.0f);
    const float res_neg = 1.0f - res_pos;
    return (f2bits(z) & 0x80000000u) ? res_neg : res_pos; // blendv on the sign BIT (so -0.0 -> res_neg)
  }
};

// logistic, rans_interface.cpp:208-214
template <bool FAST> struct Phi<MODE_LOGISTIC, FAST> {
  static __device__ __forceinline__ float eval(float z) {
    const float e = exp_ref(-1.0f * (1.702f * z));
    const float d = 1.0f + e;
    return FAST ? rcp_ge1(d) : 1.0f / d;
  }
};

template <int MODE> __device__ __forceinline__ float phi(float z) { return Phi<MODE, false>::eval(z); }

// _fast_gmm_cdf<4>, AVX branch (rans_interface.cpp:259-283): (p0 + p1) + (p2 + p3)
template <int MODE>
__device__ __forceinline__ float mix4(float x, const float (&mu)[4], const float (&sg)[4], const float (&pi)[4]) {
  float p[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) p[k] = pi[k] * phi<MODE>((x - mu[k]) / sg[k]);
  return (p[0] + p[1]) + (p[2] + p[3]);
}

// Same value, for CLAMPED sigma (entropy-model path), with the refined reciprocals rs[k] = rcp_refined(sg[k])
// computed once per latent and reused for every abscissa.  Branch-free: `ok` comes back false when some z_k is
// not finite with |z_k| < 2^48 (huge / inf / NaN mean, NaN sigma: then |x - mu_k| < 2^56 is not established and
// the cores' domains do not hold) — the caller then discards the value and takes mix4_slow.
template <int MODE>
__device__ __forceinline__ float mix4_clamped(float x, const float (&mu)[4], const float (&sg)[4], const float (&rs)[4],
                                              const float (&pi)[4], bool &ok) {
  float p[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float z = div_core(x - mu[k], sg[k], rs[k]); // exact when |x - mu| < 2^60; else huge, inf or NaN
    ok = ok && (__builtin_fabsf(z) < 0x1p48f);
    p[k] = pi[k] * Phi<MODE, true>::eval(z);
  }
  return (p[0] + p[1]) + (p[2] + p[3]);
}
// out-of-line IEEE evaluation for the rare latent the fast cores do not cover
template <int MODE>
__device__ __noinline__ float mix4_slow(float x, float m0, float m1, float m2, float m3, float s0, float s1, float s2, float s3,
                                        float w0, float w1, float w2, float w3) {
  const float mu[4] = {m0, m1, m2, m3}, sg[4] = {s0, s1, s2, s3}, pi[4] = {w0, w1, w2, w3};
  return mix4<MODE>(x, mu, sg, pi);
}

// ---------------------------------------------------------------------------------------------------------
// Two abscissae at once: packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).
//
// Measured on MI355X (scripts/valu_peak.hip, profiles/r02_valu_peak.txt): a wave64 v_fma_f32 issues once per ~4.5
// cycles per SIMD at any occupancy (36 T lane-op/s chip-wide), a v_pk_fma_f32 once per ~5.2 — two results for little
// more than the price of one.  Both CDF kernels have a natural pair that shares all twelve parameters: the two edges
// v - 0.5, v + 0.5 of a symbol (encode side) and two consecutive edges of a latent's row (decode side).  Each half of
// a packed operation is the same single IEEE-754 binary32 operation as its scalar form, so every function below is
// the scalar sequence above, lane for lane: the non-arithmetic steps (floor, int conversion, ldexp, rsq / rcp seeds,
// compares, sign transfer) stay scalar per half.  Equality with Phi<MODE, false> is checked for EVERY binary32
// |z| < 2^48 in both halves by fgmm_selftest_fastmath(3..5).
// (The compiler's own SLP pairing of unrelated scalars stays off, build.sh: it pays for its pairs with moves.)
// ---------------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat(float s) { return (f2){s, s}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// exp_nonpos on both halves WITHOUT its lower clamp, for arguments in [-2^30, 0]: below -88.38 the clamped form returns
// less than 2^-126, and so does this one — n = floor(x * log2e + 0.5) <= -127 fits an int, the reduced argument and the
// polynomial stay finite (|x_red| grows with the rounding of x * log2e: <= ~16 at 2^28), and v_ldexp_f32 of a finite y by
// n <= -127 is +-0 or a denormal.  Polya and A&S only use e in 1 - e resp. fma(-c*e, poly, 1) with finite poly: 1.0f for
// every |e| < 2^-25 of either sign.  The callers' guard (|x - mu| < 2^11, sigma >= 0.11: |z| < 2^15) keeps the argument
// above -2^30; Phi2 == Phi<MODE, false> is checked for EVERY binary32 |z| < 2^15 (fgmm_selftest_fastmath 3, 4).
__device__ __forceinline__ f2 exp_nonpos2(f2 x) {
  f2 fx = fma2(x, splat(1.44269504088896341f), splat(0.5f));
  fx.x = __builtin_floorf(fx.x);
  fx.y = __builtin_floorf(fx.y);
  x = fma2(-fx, splat(0.693359375f), x);
  x = fma2(-fx, splat(-2.12194440e-4f), x);
  const f2 z = x * x;
  f2 y = splat(1.9875691500E-4f);
  y = fma2(y, x, splat(1.3981999507E-3f));
  y = fma2(y, x, splat(8.3334519073E-3f));
  y = fma2(y, x, splat(4.1665795894E-2f));
  y = fma2(y, x, splat(1.6666665459E-1f));
  y = fma2(y, x, splat(5.0000001201E-1f));
  y = fma2(y, z, x);
  y = y + splat(1.0f);
  return (f2){__builtin_ldexpf(y.x, (int)fx.x), __builtin_ldexpf(y.y, (int)fx.y)};
}
__device__ __forceinline__ f2 exp_ref2(f2 x) { // exp_ref, both halves (any argument)
  x.x = (x.x < 88.3762626647949f) ? x.x : 88.3762626647949f;
  x.y = (x.y < 88.3762626647949f) ? x.y : 88.3762626647949f;
  x.x = (x.x > -88.3762626647949f) ? x.x : -88.3762626647949f;
  x.y = (x.y > -88.3762626647949f) ? x.y : -88.3762626647949f;
  f2 fx = fma2(x, splat(1.44269504088896341f), splat(0.5f));
  fx.x = __builtin_floorf(fx.x);
  fx.y = __builtin_floorf(fx.y);
  x = fma2(-fx, splat(0.693359375f), x);
  x = fma2(-fx, splat(-2.12194440e-4f), x);
  const f2 z = x * x;
  f2 y = splat(1.9875691500E-4f);
  y = fma2(y, x, splat(1.3981999507E-3f));
  y = fma2(y, x, splat(8.3334519073E-3f));
  y = fma2(y, x, splat(4.1665795894E-2f));
  y = fma2(y, x, splat(1.6666665459E-1f));
  y = fma2(y, x, splat(5.0000001201E-1f));
  y = fma2(y, z, x);
  y = y + splat(1.0f);
  const f2 p2 = {bits2f((uint32_t)((int)fx.x + 127) << 23), bits2f((uint32_t)((int)fx.y + 127) << 23)};
  return y * p2;
}
// a / d for both halves of a, one denominator d with r = rcp_refined(d) (domain as div_core)
__device__ __forceinline__ f2 div_core2(f2 a, float d, float r) {
  const f2 nd = splat(-d), rr = splat(r);
  const f2 q0 = a * rr;
  const f2 e2 = fma2(nd, q0, a);
  const f2 q1 = fma2(e2, rr, q0);
  const f2 e3 = fma2(nd, q1, a);
  return fma2(e3, rr, q1);
}
// 1 / d per half, 1 <= d < 2^60 (rcp_ge1_tame): q0 = 1 * r is r itself
__device__ __forceinline__ f2 rcp_ge1_tame2(f2 d) {
  const f2 r0 = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const f2 e = fma2(-d, r0, splat(1.0f));
  const f2 r = fma2(e, r0, r0);
  const f2 e2 = fma2(-d, r, splat(1.0f));
  const f2 q1 = fma2(e2, r, r);
  const f2 e3 = fma2(-d, q1, splat(1.0f));
  return fma2(e3, r, q1);
}
// sqrt_unit per half: x in {+0} U [2^-24, 1]
__device__ __forceinline__ f2 sqrt_unit2(f2 x) {
  // x == 0: rsq gives +inf and 0 * inf would be NaN.  The seed is taken of x + 2^-100 instead - ONE packed add for both halves
  // (it was a v_min per half): any x >= 2^-24 absorbs the addend exactly (its ulp is >= 2^-47), and for x == 0 the seed
  // is 2^50, with which every step below yields +0, which is sqrt(0)
  const f2 xs = x + splat(0x1p-100f);
  const f2 y = {__builtin_amdgcn_rsqf(xs.x), __builtin_amdgcn_rsqf(xs.y)};
  const f2 s0 = x * y, h = splat(0.5f) * y;
  const f2 r = fma2(-s0, s0, x);
  return fma2(r, h, s0);
}

// Phi<MODE, true> on both halves; same domain (finite |z| < 2^48 in each half)
template <int MODE> struct Phi2;
template <> struct Phi2<MODE_POLYA> {
  static __device__ __forceinline__ f2 eval(f2 z) {
    const float c = -2.0f / 3.14159265358979323846f;
    const f2 e = exp_nonpos2(splat(c) * (z * z));
    f2 s = sqrt_unit2(splat(1.0f) - e);
    s.x = bits2f((f2bits(z.x) & 0x80000000u) | (f2bits(s.x) & 0x7fffffffu));
    s.y = bits2f((f2bits(z.y) & 0x80000000u) | (f2bits(s.y) & 0x7fffffffu));
    return fma2(splat(0.5f), s, splat(0.5f));
  }
};
template <> struct Phi2<MODE_AS> {
  static __device__ __forceinline__ f2 eval(f2 z) {
    const f2 az = {bits2f(f2bits(z.x) & 0x7fffffffu), bits2f(f2bits(z.y) & 0x7fffffffu)};
    const f2 zx = splat(0.3989422804014327f) * exp_nonpos2((z * z) * splat(-0.5f));
    const f2 d = fma2(splat(0.2316419f), az, splat(1.0f));
    const f2 t = rcp_ge1_tame2(d);
    f2 poly = fma2(splat(1.330274429f), t, splat(-1.821255978f));
    poly = fma2(poly, t, splat(1.781477937f));
    poly = fma2(poly, t, splat(-0.356563782f));
    poly = fma2(poly, t, splat(0.319381530f));
    poly = poly * t;
    const f2 res_pos = fma2(-zx, poly, splat(1.0f));
    const f2 res_neg = splat(1.0f) - res_pos;
    return (f2){(f2bits(z.x) & 0x80000000u) ? res_neg.x : res_pos.x, (f2bits(z.y) & 0x80000000u) ? res_neg.y : res_pos.y};
  }
};
template <> struct Phi2<MODE_LOGISTIC> {
  static __device__ __forceinline__ f2 eval(f2 z) {
    const f2 e = exp_ref2(splat(-1.0f) * (splat(1.702f) * z));
    const f2 d = splat(1.0f) + e;
    // d = 1 + e >= 1; e can be huge (+inf at z << 0): the guarded reciprocal, per half
    f2 q = rcp_ge1_tame2(d);
    if (__builtin_expect(!(tame(d.x) && tame(d.y)), 0)) q = (f2){1.0f / d.x, 1.0f / d.y};
    return q;
  }
};

// Parameters of one latent prepared for mix4_clamped2: sigma clamped to [0.11, 256] (entropy_models.py:817) and its
// refined reciprocal, components paired so that the two Newton fmas are packed.  `tame` is false when some sigma is
// NaN (torch.clamp keeps NaN; v_med3_f32 does not): the caller then takes the IEEE path with clamp_scale().
struct Sigma4 {
  float sg[4], rs[4];
  bool tame;
  __device__ __forceinline__ void set(float s0, float s1, float s2, float s3) {
    tame = (s0 == s0) && (s1 == s1) && (s2 == s2) && (s3 == s3);
    const f2 a = {__builtin_amdgcn_fmed3f(s0, 0.11f, 256.0f), __builtin_amdgcn_fmed3f(s1, 0.11f, 256.0f)};
    const f2 b = {__builtin_amdgcn_fmed3f(s2, 0.11f, 256.0f), __builtin_amdgcn_fmed3f(s3, 0.11f, 256.0f)};
    const f2 ra = {__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}, rb = {__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
    const f2 ea = fma2(-a, ra, splat(1.0f)), eb = fma2(-b, rb, splat(1.0f));
    const f2 qa = fma2(ea, ra, ra), qb = fma2(eb, rb, rb); // rcp_refined, two at a time
    sg[0] = a.x; sg[1] = a.y; sg[2] = b.x; sg[3] = b.y;
    rs[0] = qa.x; rs[1] = qa.y; rs[2] = qb.x; rs[3] = qb.y;
  }
};

// mix4_clamped at two abscissae of one latent that lie one apart (x.y == x.x + 1.0f: the two edges of a symbol, two
// consecutive edges of a row).  Guard: `ok` comes back false unless every numerator x.x - mu_k is finite and below
// 2^11 in magnitude — then both halves have |x - mu_k| <= 2^11 + 1 and, sigma being in [0.11, 256], |z| < 2^15: inside
// the domain of div_core2 and of Phi2.  One compare per component; NaN sigmas are Sigma4::tame's to catch.
template <int MODE>
__device__ __forceinline__ f2 mix4_clamped2(f2 x, const float (&mu)[4], const Sigma4 &S, const float (&pi)[4], bool &ok) {
  f2 p[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f2 a = x - splat(mu[k]);
    ok = ok && (__builtin_fabsf(a.x) < 0x1p11f); // false for inf / NaN too
    p[k] = splat(pi[k]) * Phi2<MODE>::eval(div_core2(a, S.sg[k], S.rs[k]));
  }
  return (p[0] + p[1]) + (p[2] + p[3]);
}

// ---------------------------------------------------------------------------------------------------------
// Mixture weights from logits, in the kernels (the parameter head's softmax over K,
// latent_codecs/gaussian_mixture_conditional.py:198-202, fused into the consumers of pi): ONE fixed sequence of
// binary32 operations, the same in the encode-side and the decode-side kernel, so that a stream coded from logits
// decodes from logits whatever device torch.softmax would have run on.
//   m = max(l0..l3);  e_k = exp(l_k - m) (the Cephes sequence above), exactly 0 for l_k - m < -41 (e^-41 = 1.6e-18: nothing a
//   16-bit CDF can see, and it keeps e_k inside the exact division core's domain);  s = (e0 + e1) + (e2 + e3) in [1, 4];
//   pi_k = e_k / s, correctly rounded.          Within 2e-7 of torch.softmax (tests pin 1e-6).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void softmax4(float (&w)[4]) {
  const float m = __builtin_fmaxf(__builtin_fmaxf(w[0], w[1]), __builtin_fmaxf(w[2], w[3]));
  f2 d0 = {w[0] - m, w[1] - m}, d1 = {w[2] - m, w[3] - m};
  const bool z0 = !(d0.x >= -41.0f), z1 = !(d0.y >= -41.0f), z2 = !(d1.x >= -41.0f), z3 = !(d1.y >= -41.0f); // also NaN
  d0.x = __builtin_fmaxf(d0.x, -100.0f); d0.y = __builtin_fmaxf(d0.y, -100.0f);
  d1.x = __builtin_fmaxf(d1.x, -100.0f); d1.y = __builtin_fmaxf(d1.y, -100.0f);
  f2 e0 = exp_nonpos2(d0), e1 = exp_nonpos2(d1);
  e0.x = z0 ? 0.0f : e0.x; e0.y = z1 ? 0.0f : e0.y;
  e1.x = z2 ? 0.0f : e1.x; e1.y = z3 ? 0.0f : e1.y;
  const float s = (e0.x + e0.y) + (e1.x + e1.y); // >= 1: the largest logit contributes exp(0) = 1 (all-NaN / all -inf rows: NaN)
  const float r = rcp_refined(s);
  const f2 p0 = div_core2(e0, s, r), p1 = div_core2(e1, s, r); // e in {0} U [2^-60, 1], s in [1, 4]: the core's domain
  const bool ok = s >= 1.0f;                                    // false for NaN
  w[0] = ok ? p0.x : e0.x / s; w[1] = ok ? p0.y : e0.y / s;
  w[2] = ok ? p1.x : e1.x / s; w[3] = ok ? p1.y : e1.y / s;
}

// static_cast<uint16_t>(float) as x86-64 GCC emits it (cvttss2si r32 ; movzwl), rans_interface.cpp:509-510.
// cvttss2si yields 0x80000000 for NaN / out-of-range, v_cvt_i32_f32 saturates: make the x86 answer explicit.
// Only the low 16 bits are kept, so only ONE case needs the explicit answer: f >= 2^31 saturates to 0x7FFFFFFF (low half
// 0xFFFF) where x86 gives 0x80000000 (low half 0).  NaN converts to 0 and f < -2^31 saturates to 0x80000000: low half 0 like
// x86's; everything in range truncates toward zero on both.
__device__ __forceinline__ uint32_t quant16(float cdf) {
  const float f = cdf * 65535.0f;
  int t; // the instruction itself (a C++ cast of an out-of-range float is undefined, and the optimizer may use that)
  asm("v_cvt_i32_f32 %0, %1" : "=v"(t) : "v"(f));
  t = (f >= 2147483648.0f) ? 0 : t;
  return (uint32_t)t & 0xFFFFu;
}

// sigma clamp of reshape_entropy_parameters (entropy_models.py:817): torch.clamp(x, 0.11, 256) keeps NaN
__device__ __forceinline__ float clamp_scale(float s) {
  s = (s < 0.11f) ? 0.11f : s;
  s = (s > 256.0f) ? 256.0f : s;
  return s;
}

// Saturation thresholds of the three Phi sequences above, used to prune the decode-side table evaluation.
// For every binary32 z (exhaustively scanned on the GPU by fgmm_selftest_saturation, tests/test_gpu_parity.py):
//   z <= -ZL  =>  phi(z) == +0.0f exactly           (logistic: phi(z) <= 2^-20)
//   z >= +ZR  =>  phi(z) == 1.0f exactly
// Measured last non-saturated |z| (CPU scan of the oracle): Polya 5.2172623 / 4.969077, A&S 5.4203954 both,
// logistic 8.145089 (2^-20 bound) / 9.774107.
// weight_ok: what a mixture weight must satisfy for the lemma to give F == 0 / F == quant16(sum pi):
//   exact 0/1 saturation only needs finite weights; the logistic left tail is a bound, so it needs 0 <= pi <= 1
//   (then cdf*65535 <= 4 * 2^-20 * 65535 * (1 + eps) < 1).
template <int MODE> struct Sat;
template <> struct Sat<MODE_POLYA> {
  static constexpr float ZL = 5.25f, ZR = 5.0f;
  static constexpr float LEFT_MAX = 0.0f;
  __device__ static __forceinline__ bool weight_ok(float w) { return fabsf(w) < INFINITY; }
};
template <> struct Sat<MODE_AS> {
  static constexpr float ZL = 5.45f, ZR = 5.45f;
  static constexpr float LEFT_MAX = 0.0f;
  __device__ static __forceinline__ bool weight_ok(float w) { return fabsf(w) < INFINITY; }
};
template <> struct Sat<MODE_LOGISTIC> {
  static constexpr float ZL = 8.2f, ZR = 9.8f;
  static constexpr float LEFT_MAX = 9.5367431640625e-07f; // 2^-20
  __device__ static __forceinline__ bool weight_ok(float w) { return w >= 0.0f && w <= 1.0f; }
};

} // namespace fgmm

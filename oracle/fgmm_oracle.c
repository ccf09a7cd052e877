/*
 * fgmm_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's GMM entropy-coding path (tokkiwa/FlashGMM), used ONLY by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the *checker* for the HIP product
 * path in flashgmm_amd/.  Nothing in flashgmm_amd/ may import, link or call this file.
 *
 * Parity status: PINNED.  tests/golden/make_golden.py compares every function here against the real
 * reference compiled into oracle/_ref/ (oracle/Makefile `make ref`: the unmodified pybind11 extension
 * `compressai.ans` and a probe TU that #includes the reference's own .cpp) — float CDF pairs bit-for-bit,
 * (start,range) pairs, encoder bytes and decoder output in all three APPROX_MODEs — and commits the
 * resulting vectors under tests/golden/.  SURVEY.md §8c's known answer KA-1 (three md5s) is among them.
 *
 * All citations are relative to /root/reference/.  "rans_interface.cpp" = compressai/cpp_exts/rans/
 * rans_interface.cpp, "avx_mathfun.h" = compressai/cpp_exts/rans/avx_mathfun.h, "rans64.h" =
 * third_party/ryg_rans/rans64.h.
 *
 * Parity target = the reference's DEFAULT path (USE_SIMD unset, K == 4): the AVX branch of
 * _fast_gmm_cdf<4> as GCC compiles it at -O3 with FMA available (setup.py:72-76), i.e. with the
 * mul/add intrinsic pairs contracted into FMAs.  Every operation below is one IEEE-754 binary32
 * round-to-nearest-even operation; this file must be compiled with -ffp-contract=off so that the only
 * fused operations are the explicit fmaf() calls.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FGO_MODE_POLYA 0    /* APPROX_MODE=0 (default)            rans_interface.cpp:224-232 */
#define FGO_MODE_AS 1       /* APPROX_MODE=1 Abramowitz & Stegun  (README says 2; the code wins) */
#define FGO_MODE_LOGISTIC 2 /* APPROX_MODE=2 logistic */

#define FGO_PRECISION 16        /* rans_interface.cpp:55 */
#define FGO_MAX_CDF 65535       /* rans_interface.cpp:56 */
#define FGO_BYPASS_PRECISION 4  /* rans_interface.cpp:58 */
#define FGO_MAX_BYPASS_VAL 15   /* rans_interface.cpp:59 */
#define FGO_RANS64_L (1ull << 31) /* rans64.h:59 */

/* ------------------------------------------------------------------------------------------------
 * float helpers
 * ---------------------------------------------------------------------------------------------- */

/* _mm256_min_ps(a,b) / _mm256_max_ps(a,b): "a OP b ? a : b" — returns b when either is NaN. */
static inline float fgo_min_ps(float a, float b) { return a < b ? a : b; }
static inline float fgo_max_ps(float a, float b) { return a > b ? a : b; }

static inline float fgo_bits2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint32_t fgo_f2bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}

/* cvttss2si / _mm256_cvttps_epi32: truncate, "integer indefinite" 0x80000000 when out of range or NaN. */
static inline int32_t fgo_cvtt(float f) {
  if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT32_MIN;
  return (int32_t)f;
}

/* exp256_ps, avx_mathfun.h:250-304, as contracted by GCC -O3 + FMA (SURVEY.md §8a').
 * The 2^n factor is built as the bit pattern (n+127)<<23 and MULTIPLIED in (avx_mathfun.h:297-302);
 * this is not ldexp: n = -127 gives +0.0 and n = 128 gives +inf. */
float fgo_exp(float x) {
  x = fgo_min_ps(x, 88.3762626647949f);  /* :255 */
  x = fgo_max_ps(x, -88.3762626647949f); /* :256 */
  float fx = fmaf(x, 1.44269504088896341f, 0.5f); /* :259-260 */
  float tmp = floorf(fx);                         /* :266 */
  float mask = (tmp > fx) ? 1.0f : 0.0f;          /* :270-271 (never fires for finite fx) */
  fx = tmp - mask;                                /* :272 */
  x = fmaf(-fx, 0.693359375f, x);                 /* :274,276 */
  x = fmaf(-fx, -2.12194440e-4f, x);              /* :275,277 */
  float z = x * x;                                /* :279 */
  float y = 1.9875691500E-4f;                     /* :281 */
  y = fmaf(y, x, 1.3981999507E-3f);               /* :282-283 */
  y = fmaf(y, x, 8.3334519073E-3f);
  y = fmaf(y, x, 4.1665795894E-2f);
  y = fmaf(y, x, 1.6666665459E-1f);
  y = fmaf(y, x, 5.0000001201E-1f);               /* :290-291 */
  y = fmaf(y, z, x);                              /* :292-293 */
  y = y + 1.0f;                                   /* :294 */
  int32_t n = fgo_cvtt(fx);                       /* :297 */
  uint32_t pw = (uint32_t)(n + 0x7f) << 23;       /* :299-300 */
  return y * fgo_bits2f(pw);                      /* :301-302 */
}

/* Polya/Watterson, rans_interface.cpp:135-146 */
static inline float fgo_phi_polya(float z) {
  const float c = -2.0f / 3.14159265358979323846f; /* :138, folded in binary32 */
  float e = fgo_exp(c * (z * z));                  /* :140-141 */
  float s = sqrtf(1.0f - e);                       /* :142 */
  s = fgo_bits2f((fgo_f2bits(z) & 0x80000000u) | (fgo_f2bits(s) & 0x7fffffffu)); /* :143, copysign_ps :87-93 */
  return 0.5f * (1.0f + s);                        /* :145 */
}

/* Abramowitz & Stegun 26.2.17, rans_interface.cpp:154-186 */
static inline float fgo_phi_as(float z) {
  float az = fgo_bits2f(fgo_f2bits(z) & 0x7fffffffu);     /* :167 */
  float zx = 0.3989422804014327f * fgo_exp((z * z) * -0.5f); /* :169-171 */
  float t = 1.0f / fmaf(0.2316419f, az, 1.0f);            /* :173 (mul+add contracted) */
  float poly = fmaf(1.330274429f, t, -1.821255978f);      /* :175 */
  poly = fmaf(poly, t, 1.781477937f);
  poly = fmaf(poly, t, -0.356563782f);
  poly = fmaf(poly, t, 0.319381530f);
  poly = poly * t;                                        /* :179 */
  float res_pos = fmaf(-zx, poly, 1.0f);                  /* :181 (1 - zx*poly contracted) */
  float res_neg = 1.0f - res_pos;                         /* :182 */
  return (fgo_f2bits(z) & 0x80000000u) ? res_neg : res_pos; /* :184-185 blendv on the sign bit */
}

/* logistic, rans_interface.cpp:208-214 */
static inline float fgo_phi_logistic(float z) {
  float e = fgo_exp(-1.0f * (1.702f * z)); /* :211-212 */
  return 1.0f / (1.0f + e);                /* :213 */
}

static inline float fgo_phi(int mode, float z) { /* _fast_gaussian_cdf(__m256) :223-233 */
  switch (mode) {
  case FGO_MODE_AS: return fgo_phi_as(z);
  case FGO_MODE_LOGISTIC: return fgo_phi_logistic(z);
  default: return fgo_phi_polya(z);
  }
}

/* _fast_gmm_cdf<4>, SIMD branch, rans_interface.cpp:259-283: z = (x-mu)/sigma; p_k = pi_k*Phi(z_k);
 * two hadd's => (p0+p1)+(p2+p3). */
static inline float fgo_mix4(int mode, float x, const float *mu, const float *sg, const float *pi) {
  float p[4];
  for (int k = 0; k < 4; ++k) p[k] = pi[k] * fgo_phi(mode, (x - mu[k]) / sg[k]);
  return (p[0] + p[1]) + (p[2] + p[3]);
}

/* static_cast<uint16_t>(float) as x86-64 GCC emits it (cvttss2si r32 ; movzwl), :509-510 */
static inline uint32_t fgo_u16(float f) { return (uint32_t)fgo_cvtt(f) & 0xFFFFu; }

static inline void fgo_gather4(const float *base, int64_t i, int64_t sn, int64_t sk, float *out) {
  for (int k = 0; k < 4; ++k) out[k] = base[i * sn + k * sk]; /* accessor[i][k], :491-495 */
}

/* quantised edge pair of symbol value v: lo = E(v), hi = E(v+1)   (:498-510) */
static inline void fgo_edges(int mode, int32_t v, const float *mu, const float *sg, const float *pi,
                             uint32_t *lo, uint32_t *hi, float *c1, float *c2) {
  float x1 = (float)v - 0.5f;
  float x2 = (float)v - 0.5f + 1.0f;
  float a = fgo_mix4(mode, x1, mu, sg, pi);
  float b = fgo_mix4(mode, x2, mu, sg, pi);
  if (c1) *c1 = a;
  if (c2) *c2 = b;
  *lo = fgo_u16(a * 65535.0f);
  *hi = fgo_u16(b * 65535.0f);
}

/* ------------------------------------------------------------------------------------------------
 * exported probes: float CDF pairs and the quantised tables
 * params are (n,4) with element strides (sn, sk) exactly like the reference's accessor<float,2>.
 * ---------------------------------------------------------------------------------------------- */

void fgo_gmm_cdf(int mode, int64_t n, const int32_t *v, const float *scales, const float *means,
                 const float *weights, int64_t sn, int64_t sk, float *c1, float *c2) {
  for (int64_t i = 0; i < n; ++i) {
    float mu[4], sg[4], pi[4];
    uint32_t lo, hi;
    fgo_gather4(means, i, sn, sk, mu);
    fgo_gather4(scales, i, sn, sk, sg);
    fgo_gather4(weights, i, sn, sk, pi);
    fgo_edges(mode, v[i], mu, sg, pi, &lo, &hi, &c1[i], &c2[i]);
  }
}

void fgo_gmm_cdf_x(int mode, int64_t n, const float *x1, const float *x2, const float *scales,
                   const float *means, const float *weights, int64_t sn, int64_t sk, float *c1, float *c2) {
  for (int64_t i = 0; i < n; ++i) {
    float mu[4], sg[4], pi[4];
    fgo_gather4(means, i, sn, sk, mu);
    fgo_gather4(scales, i, sn, sk, sg);
    fgo_gather4(weights, i, sn, sk, pi);
    c1[i] = fgo_mix4(mode, x1[i], mu, sg, pi);
    c2[i] = fgo_mix4(mode, x2[i], mu, sg, pi);
  }
}

/* Encode-side symbol table: packed[i] = start | range<<16 with range = (uint16)(hi-lo).
 * range == 0  <=>  the reference takes the bypass escape for this symbol (:512-517); for those entries the
 * low half carries the low 16 bits of the symbol value (a convention of THIS repo's table format). */
void fgo_symtab(int mode, int64_t n, const int32_t *v, const float *scales, const float *means,
                const float *weights, int64_t sn, int64_t sk, uint32_t *packed) {
  for (int64_t i = 0; i < n; ++i) {
    float mu[4], sg[4], pi[4];
    uint32_t lo, hi;
    fgo_gather4(means, i, sn, sk, mu);
    fgo_gather4(scales, i, sn, sk, sg);
    fgo_gather4(weights, i, sn, sk, pi);
    fgo_edges(mode, v[i], mu, sg, pi, &lo, &hi, 0, 0);
    uint32_t pmf = (hi - lo) & 0xFFFFu;
    packed[i] = pmf ? (lo | (pmf << 16)) : ((uint32_t)v[i] & 0xFFFFu);
  }
}

/* Decode-side full edge table: tab[i*W + j] = E_i(v = -max_bs + j), j = 0 .. W-1, W = 2*max_bs + 2.
 * These are exactly the values the reference's bisection can ever look at (:826-862: mid in
 * [-max_bs, max_bs], probes E(mid) and E(mid+1)). */
void fgo_cdftab(int mode, int64_t n, const float *scales, const float *means, const float *weights,
                int64_t sn, int64_t sk, int32_t max_bs, uint16_t *tab) {
  const int64_t W = 2 * (int64_t)max_bs + 2;
  for (int64_t i = 0; i < n; ++i) {
    float mu[4], sg[4], pi[4];
    fgo_gather4(means, i, sn, sk, mu);
    fgo_gather4(scales, i, sn, sk, sg);
    fgo_gather4(weights, i, sn, sk, pi);
    for (int64_t j = 0; j < W; ++j) {
      /* E(v) is evaluated as the *lower* edge of v; the upper edge of v-1 is float(v-1)-0.5f+1.0f, the same
       * binary32 number for |v| < 2^22, so one table serves both probes. */
      float x = (float)(int32_t)(-max_bs + j) - 0.5f;
      tab[i * W + j] = (uint16_t)fgo_u16(fgo_mix4(mode, x, mu, sg, pi) * 65535.0f);
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * rANS core — rans64.h:65-142, bypass bits rans_interface.cpp:295-331
 * ---------------------------------------------------------------------------------------------- */

static inline void fgo_enc_put(uint64_t *r, uint32_t **pptr, uint32_t start, uint32_t freq) {
  uint64_t x = *r;
  uint64_t x_max = ((FGO_RANS64_L >> FGO_PRECISION) << 32) * freq; /* rans64.h:83 */
  if (x >= x_max) {
    *pptr -= 1;
    **pptr = (uint32_t)x;
    x >>= 32;
  }
  *r = ((x / freq) << FGO_PRECISION) + (x % freq) + start; /* rans64.h:92 */
}

static inline void fgo_enc_put_bits(uint64_t *r, uint32_t **pptr, uint32_t val, uint32_t nbits) {
  uint64_t x = *r;
  uint32_t freq = 1u << (16 - nbits);                        /* :302 */
  uint64_t x_max = ((FGO_RANS64_L >> 16) << 32) * freq;      /* :303 */
  if (x >= x_max) {
    *pptr -= 1;
    **pptr = (uint32_t)x;
    x >>= 32;
  }
  *r = (x << nbits) | val; /* :312 */
}

static inline uint32_t fgo_dec_get_bits(uint64_t *r, const uint32_t **pptr, uint32_t nbits) {
  uint64_t x = *r;
  uint32_t val = (uint32_t)(x & ((1u << nbits) - 1)); /* :318 */
  x >>= nbits;
  if (x < FGO_RANS64_L) {
    x = (x << 32) | **pptr;
    *pptr += 1;
  }
  *r = x;
  return val;
}

static inline void fgo_dec_advance(uint64_t *r, const uint32_t **pptr, uint32_t start, uint32_t freq) {
  uint64_t mask = (1ull << FGO_PRECISION) - 1;
  uint64_t x = *r;
  x = freq * (x >> FGO_PRECISION) + (x & mask) - start; /* rans64.h:132 */
  if (x < FGO_RANS64_L) {
    x = (x << 32) | **pptr;
    *pptr += 1;
  }
  *r = x;
}

typedef struct {
  uint16_t start, range;
  uint8_t bypass;
} fgo_sym; /* RansSymbol, rans_interface.hpp:48-52 */

typedef struct {
  fgo_sym *s;
  int64_t n, cap;
} fgo_symvec;

static int fgo_push(fgo_symvec *v, uint32_t start, uint32_t range, int bypass) {
  if (v->n == v->cap) {
    int64_t nc = v->cap ? v->cap * 2 : 1024;
    fgo_sym *p = (fgo_sym *)realloc(v->s, (size_t)nc * sizeof(fgo_sym));
    if (!p) return -1;
    v->s = p;
    v->cap = nc;
  }
  v->s[v->n].start = (uint16_t)start;
  v->s[v->n].range = (uint16_t)range;
  v->s[v->n].bypass = (uint8_t)bypass;
  v->n++;
  return 0;
}

/* one symbol with quantised edges (lo,hi) -> RansSymbol entries, rans_interface.cpp:509-552 */
static int fgo_push_symbol(fgo_symvec *sv, int32_t value, uint32_t lo, uint32_t hi) {
  int32_t cdf_value = (int32_t)lo, cdf_value_next = (int32_t)hi;
  uint16_t pmf = (uint16_t)(cdf_value_next - cdf_value); /* :512 */
  int bypass = 0;
  if (pmf == 0) { /* :513-517 */
    bypass = 1;
    cdf_value = FGO_MAX_CDF;
    cdf_value_next = FGO_MAX_CDF + 1;
  }
  if (fgo_push(sv, (uint16_t)cdf_value, (uint16_t)(cdf_value_next - cdf_value), 0)) return -1; /* :519-521 */
  if (bypass) {
    uint32_t raw_val = (uint32_t)value; /* :525 bit pattern of the int32 */
    int32_t n_bypass = 0;
    uint32_t t = raw_val;
    while (t != 0 && n_bypass * FGO_BYPASS_PRECISION < 32) { /* :530-533 */
      t >>= FGO_BYPASS_PRECISION;
      ++n_bypass;
    }
    int32_t val_n = n_bypass;
    while (val_n >= FGO_MAX_BYPASS_VAL) { /* :538-541 (never: n_bypass <= 8) */
      if (fgo_push(sv, FGO_MAX_BYPASS_VAL, FGO_MAX_BYPASS_VAL + 1, 1)) return -1;
      val_n -= FGO_MAX_BYPASS_VAL;
    }
    if (fgo_push(sv, (uint32_t)val_n, (uint32_t)val_n + 1, 1)) return -1; /* :542-543 */
    for (int32_t j = 0; j < n_bypass; ++j) {                                /* :546-551 */
      uint32_t nib = (raw_val >> (j * FGO_BYPASS_PRECISION)) & FGO_MAX_BYPASS_VAL;
      if (fgo_push(sv, nib, nib + 1, 1)) return -1;
    }
  }
  return 0;
}

/* BufferedRansEncoder::flush, rans_interface.cpp:557-585.  Returns malloc'ed bytes. */
static int fgo_flush(fgo_symvec *sv, uint8_t **out, size_t *out_len) {
  size_t nwords = (size_t)sv->n + 16; /* :563 */
  uint32_t *buf = (uint32_t *)malloc(nwords * sizeof(uint32_t));
  if (!buf) return -1;
  uint32_t *end = buf + nwords, *ptr = end;
  uint64_t rans = FGO_RANS64_L;              /* Rans64EncInit */
  for (int64_t i = sv->n - 1; i >= 0; --i) { /* std::reverse + forward walk, :569-577 */
    const fgo_sym *s = &sv->s[i];
    if (!s->bypass)
      fgo_enc_put(&rans, &ptr, s->start, s->range);
    else
      fgo_enc_put_bits(&rans, &ptr, s->start, FGO_BYPASS_PRECISION);
  }
  ptr -= 2; /* Rans64EncFlush, rans64.h:96-103 */
  ptr[0] = (uint32_t)(rans >> 0);
  ptr[1] = (uint32_t)(rans >> 32);
  size_t nbytes = (size_t)(end - ptr) * sizeof(uint32_t);
  uint8_t *o = (uint8_t *)malloc(nbytes ? nbytes : 1);
  if (!o) {
    free(buf);
    return -1;
  }
  memcpy(o, ptr, nbytes);
  free(buf);
  *out = o;
  *out_len = nbytes;
  return 0;
}

void fgo_free(void *p) { free(p); }

/* RansEncoder::encode_with_indexes_gmm<4>, rans_interface.cpp:609-617 (-> :458-554 -> :557-585).
 * n_bypass_out (optional) counts symbols that took the escape. */
int fgo_encode_gmm(int mode, int64_t n, const int32_t *symbols, const float *scales, const float *means,
                   const float *weights, int64_t sn, int64_t sk, uint8_t **out, size_t *out_len,
                   int64_t *n_bypass_out) {
  fgo_symvec sv = {0, 0, 0};
  int64_t nb = 0;
  for (int64_t i = 0; i < n; ++i) {
    float mu[4], sg[4], pi[4];
    uint32_t lo, hi;
    fgo_gather4(means, i, sn, sk, mu);
    fgo_gather4(scales, i, sn, sk, sg);
    fgo_gather4(weights, i, sn, sk, pi);
    fgo_edges(mode, symbols[i], mu, sg, pi, &lo, &hi, 0, 0);
    if (((hi - lo) & 0xFFFFu) == 0) nb++;
    if (fgo_push_symbol(&sv, symbols[i], lo, hi)) {
      free(sv.s);
      return -1;
    }
  }
  int rc = fgo_flush(&sv, out, out_len);
  free(sv.s);
  if (n_bypass_out) *n_bypass_out = nb;
  return rc;
}

/* Integer-only encode from a symbol table in this repo's packed format (see fgo_symtab).  `symbols` supplies
 * the raw values for bypass entries.  "Same tables => same bytes" surface. */
int fgo_rans_encode_symtab(int64_t n, const uint32_t *packed, const int32_t *symbols, uint8_t **out,
                           size_t *out_len) {
  fgo_symvec sv = {0, 0, 0};
  for (int64_t i = 0; i < n; ++i) {
    uint32_t start = packed[i] & 0xFFFFu, range = packed[i] >> 16;
    uint32_t lo = range ? start : 0, hi = range ? ((start + range) & 0xFFFFu) : 0;
    /* (lo,hi) with (hi-lo)&0xFFFF == range reproduces the entry; range==0 -> bypass */
    if (fgo_push_symbol(&sv, symbols[i], lo, hi)) {
      free(sv.s);
      return -1;
    }
  }
  int rc = fgo_flush(&sv, out, out_len);
  free(sv.s);
  return rc;
}

/* bypass read shared by both decoders, rans_interface.cpp:808-824 */
static inline int32_t fgo_dec_bypass(uint64_t *rans, const uint32_t **ptr) {
  fgo_dec_advance(rans, ptr, FGO_MAX_CDF, 1);                             /* :809 */
  int32_t val_bypass = (int32_t)fgo_dec_get_bits(rans, ptr, FGO_BYPASS_PRECISION); /* :810 */
  int32_t n_bypass = val_bypass;
  while (val_bypass == FGO_MAX_BYPASS_VAL) { /* :813-816 */
    val_bypass = (int32_t)fgo_dec_get_bits(rans, ptr, FGO_BYPASS_PRECISION);
    n_bypass += val_bypass;
  }
  uint32_t raw_val = 0;
  for (int j = 0; j < n_bypass; ++j) { /* :819-823 */
    val_bypass = (int32_t)fgo_dec_get_bits(rans, ptr, FGO_BYPASS_PRECISION);
    /* the reference shifts an int by j*4 (j can reach 31+ only on corrupt streams); mask the shift to stay
     * defined — identical for every stream the encoder can produce (n_bypass <= 8). */
    raw_val |= (uint32_t)val_bypass << ((j * FGO_BYPASS_PRECISION) & 31);
  }
  return (int32_t)raw_val; /* :824 */
}

/* RansDecoder::decode_with_indexes_gmm<4>, rans_interface.cpp:766-883: float bisection per symbol. */
int fgo_decode_gmm(int mode, const uint8_t *enc, size_t enc_len, int64_t n, const float *scales,
                   const float *means, const float *weights, int64_t sn, int64_t sk, int32_t max_bs,
                   int32_t *out) {
  if (enc_len < 8) return -2;
  uint32_t *words = (uint32_t *)malloc(enc_len + 64); /* private, padded copy (the reference reads past the end on desync) */
  if (!words) return -1;
  memset(words, 0, enc_len + 64);
  memcpy(words, enc, enc_len);
  const uint32_t *ptr = words;
  uint64_t rans = (uint64_t)ptr[0] | ((uint64_t)ptr[1] << 32); /* Rans64DecInit rans64.h:107-115 */
  ptr += 2;
  const uint32_t *limit = words + (enc_len + 64) / 4 - 12;
  for (int64_t i = 0; i < n; ++i) {
    if (ptr > limit) { /* corrupt stream guard (the reference would read out of bounds) */
      free(words);
      return -3;
    }
    float mu[4], sg[4], pi[4];
    fgo_gather4(means, i, sn, sk, mu);
    fgo_gather4(scales, i, sn, sk, sg);
    fgo_gather4(weights, i, sn, sk, pi);
    uint32_t cum_freq = (uint32_t)(rans & 0xFFFFu); /* :805 */
    int32_t value;
    if (cum_freq == FGO_MAX_CDF) { /* :808 */
      value = fgo_dec_bypass(&rans, &ptr);
    } else {
      int32_t s_bs = -max_bs, e_bs = max_bs, mid = 0; /* :826-828 */
      uint32_t c1 = 0, c2 = 0;
      while (s_bs <= e_bs) { /* :833-854 */
        mid = s_bs + (e_bs - s_bs) / 2;
        fgo_edges(mode, mid, mu, sg, pi, &c1, &c2, 0, 0);
        if (c1 <= cum_freq && c2 > cum_freq) break;
        else if (c1 > cum_freq) e_bs = mid - 1;
        else s_bs = mid + 1;
      }
      fgo_edges(mode, mid, mu, sg, pi, &c1, &c2, 0, 0); /* :856-862 */
      uint32_t pmf = (c2 - c1) & 0xFFFFu;               /* :865 */
      if (pmf == 0) {                                   /* :866-875 */
        pmf = 1;
        if (c1 + pmf > (1u << FGO_PRECISION)) c1 = (1u << FGO_PRECISION) - pmf; /* unreachable: c1 <= 65535 */
      }
      fgo_dec_advance(&rans, &ptr, c1, pmf); /* :877 */
      value = mid;
    }
    out[i] = value;
  }
  free(words);
  return 0;
}

/* Integer-only decode from the full edge table of fgo_cdftab: the reference's bisection (:826-877) with
 * every float evaluation replaced by a table look-up.  "Same tables => same symbols" surface. */
int fgo_rans_decode_cdftab(const uint8_t *enc, size_t enc_len, int64_t n, const uint16_t *tab, int32_t max_bs,
                           int32_t *out) {
  if (enc_len < 8) return -2;
  const int64_t W = 2 * (int64_t)max_bs + 2;
  uint32_t *words = (uint32_t *)malloc(enc_len + 64);
  if (!words) return -1;
  memset(words, 0, enc_len + 64);
  memcpy(words, enc, enc_len);
  const uint32_t *ptr = words;
  uint64_t rans = (uint64_t)ptr[0] | ((uint64_t)ptr[1] << 32);
  ptr += 2;
  const uint32_t *limit = words + (enc_len + 64) / 4 - 12;
  for (int64_t i = 0; i < n; ++i) {
    if (ptr > limit) {
      free(words);
      return -3;
    }
    const uint16_t *E = tab + i * W + max_bs; /* E[v], v in [-max_bs, max_bs+1] */
    uint32_t cum_freq = (uint32_t)(rans & 0xFFFFu);
    int32_t value;
    if (cum_freq == FGO_MAX_CDF) {
      value = fgo_dec_bypass(&rans, &ptr);
    } else {
      int32_t s_bs = -max_bs, e_bs = max_bs, mid = 0;
      uint32_t c1 = 0, c2 = 0;
      while (s_bs <= e_bs) {
        mid = s_bs + (e_bs - s_bs) / 2;
        c1 = E[mid];
        c2 = E[mid + 1];
        if (c1 <= cum_freq && c2 > cum_freq) break;
        else if (c1 > cum_freq) e_bs = mid - 1;
        else s_bs = mid + 1;
      }
      c1 = E[mid];
      c2 = E[mid + 1];
      uint32_t pmf = (c2 - c1) & 0xFFFFu;
      if (pmf == 0) pmf = 1;
      fgo_dec_advance(&rans, &ptr, c1, pmf);
      value = mid;
    }
    out[i] = value;
  }
  free(words);
  return 0;
}

/* ================================================================================================
 * Table path (the `z` hyper-latent coder) — CompressAI's original table rANS
 *   BufferedRansEncoder::encode_with_indexes   rans_interface.cpp:334-399
 *   RansDecoder::decode_with_indexes           rans_interface.cpp:619-688   (decode_stream :894-956 is the same loop)
 *   pmf_to_quantized_cdf                       compressai/cpp_exts/ops/ops.cpp:40-109
 * cdfs is a row-major [n_cdfs, cdf_stride] int32 matrix (the reference takes a list of lists).
 * ============================================================================================== */

static int fgo_push_table_symbol(fgo_symvec *sv, int32_t symbol, const int32_t *cdf, int32_t cdf_size, int32_t offset) {
  const int32_t max_value = cdf_size - 2; /* :350 */
  int32_t value = symbol - offset;        /* :354 */
  uint32_t raw_val = 0;
  if (value < 0) { /* :357-363 */
    raw_val = (uint32_t)(-2 * value - 1);
    value = max_value;
  } else if (value >= max_value) {
    raw_val = (uint32_t)(2 * (value - max_value));
    value = max_value;
  }
  if (fgo_push(sv, (uint16_t)cdf[value], (uint16_t)(cdf[value + 1] - cdf[value]), 0)) return -1; /* :368-370 */
  if (value == max_value) { /* :373-397 */
    int32_t n_bypass = 0;
    while ((raw_val >> (n_bypass * FGO_BYPASS_PRECISION)) != 0) ++n_bypass;
    int32_t val = n_bypass;
    while (val >= FGO_MAX_BYPASS_VAL) {
      if (fgo_push(sv, FGO_MAX_BYPASS_VAL, FGO_MAX_BYPASS_VAL + 1, 1)) return -1;
      val -= FGO_MAX_BYPASS_VAL;
    }
    if (fgo_push(sv, (uint32_t)val, (uint32_t)val + 1, 1)) return -1;
    for (int32_t j = 0; j < n_bypass; ++j) {
      const uint32_t nib = (raw_val >> (j * FGO_BYPASS_PRECISION)) & FGO_MAX_BYPASS_VAL;
      if (fgo_push(sv, nib, nib + 1, 1)) return -1;
    }
  }
  return 0;
}

int fgo_encode_table(int64_t n, const int32_t *symbols, const int32_t *indexes, const int32_t *cdfs,
                     int64_t cdf_stride, const int32_t *cdfs_sizes, const int32_t *offsets, uint8_t **out,
                     size_t *out_len) {
  fgo_symvec sv = {0, 0, 0};
  for (int64_t i = 0; i < n; ++i) {
    const int32_t k = indexes[i];
    if (fgo_push_table_symbol(&sv, symbols[i], cdfs + k * cdf_stride, cdfs_sizes[k], offsets[k])) {
      free(sv.s);
      return -1;
    }
  }
  int rc = fgo_flush(&sv, out, out_len);
  free(sv.s);
  return rc;
}

int fgo_decode_table(const uint8_t *enc, size_t enc_len, int64_t n, const int32_t *indexes, const int32_t *cdfs,
                     int64_t cdf_stride, const int32_t *cdfs_sizes, const int32_t *offsets, int32_t *out) {
  if (enc_len < 8) return -2;
  uint32_t *words = (uint32_t *)malloc(enc_len + 64);
  if (!words) return -1;
  memset(words, 0, enc_len + 64);
  memcpy(words, enc, enc_len);
  const uint32_t *ptr = words;
  uint64_t rans = (uint64_t)ptr[0] | ((uint64_t)ptr[1] << 32);
  ptr += 2;
  const uint32_t *limit = words + (enc_len + 64) / 4 - 12;
  for (int64_t i = 0; i < n; ++i) {
    if (ptr > limit) {
      free(words);
      return -3;
    }
    const int32_t k = indexes[i];
    const int32_t *cdf = cdfs + k * cdf_stride;
    const int32_t max_value = cdfs_sizes[k] - 2;
    const uint32_t cum_freq = (uint32_t)(rans & 0xFFFFu); /* :648 */
    /* std::lower_bound(cdf, cdf + size, cum_freq + 1) - 1  == last s with cdf[s] <= cum_freq   (:651-653) */
    int32_t lo = 0, hi = cdfs_sizes[k];
    while (lo < hi) {
      const int32_t mid = lo + (hi - lo) / 2;
      if ((uint32_t)cdf[mid] < cum_freq + 1) lo = mid + 1; else hi = mid;
    }
    const int32_t s = lo - 1;
    fgo_dec_advance(&rans, &ptr, (uint32_t)cdf[s], (uint32_t)(cdf[s + 1] - cdf[s])); /* :656 */
    int32_t value = s;
    if (value == max_value) { /* :660-682 */
      int32_t val = (int32_t)fgo_dec_get_bits(&rans, &ptr, FGO_BYPASS_PRECISION);
      int32_t n_bypass = val;
      while (val == FGO_MAX_BYPASS_VAL) {
        val = (int32_t)fgo_dec_get_bits(&rans, &ptr, FGO_BYPASS_PRECISION);
        n_bypass += val;
      }
      int32_t raw_val = 0;
      for (int j = 0; j < n_bypass; ++j) {
        val = (int32_t)fgo_dec_get_bits(&rans, &ptr, FGO_BYPASS_PRECISION);
        raw_val |= (int32_t)((uint32_t)val << ((j * FGO_BYPASS_PRECISION) & 31));
      }
      value = raw_val >> 1;
      if (raw_val & 1) value = -value - 1; else value += max_value;
    }
    out[i] = value + offsets[k]; /* :684 */
  }
  free(words);
  return 0;
}

/* ops.cpp:40-109.  Returns 0, or -1 for a negative / non-finite element, -2 for an all-zero pmf. */
int fgo_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf /* n+1 */) {
  for (int i = 0; i < n; ++i)
    if (pmf[i] < 0 || !isfinite(pmf[i])) return -1;
  cdf[0] = 0;
  for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)roundf(pmf[i] * (float)(1 << precision)); /* std::round(float) */
  uint32_t total = 0;
  for (int i = 0; i <= n; ++i) total += cdf[i];
  if (total == 0) return -2;
  for (int i = 0; i <= n; ++i) cdf[i] = (uint32_t)(((uint64_t)(1 << precision) * cdf[i]) / total);
  for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
  cdf[n] = 1u << precision;
  for (int i = 0; i < n; ++i) {
    if (cdf[i] == cdf[i + 1]) {
      uint32_t best_freq = ~0u;
      int best_steal = -1;
      for (int j = 0; j < n; ++j) {
        uint32_t freq = cdf[j + 1] - cdf[j];
        if (freq > 1 && freq < best_freq) {
          best_freq = freq;
          best_steal = j;
        }
      }
      if (best_steal < 0) return -3; /* the reference asserts */
      if (best_steal < i) {
        for (int j = best_steal + 1; j <= i; ++j) cdf[j]--;
      } else {
        for (int j = i + 1; j <= best_steal; ++j) cdf[j]++;
      }
    }
  }
  return 0;
}

/* ---- the parameter head's last layer --------------------------------------------------------------------
 * The final 1x1 convolution of `entropy_parameters` (compressai/models/ckbd_gmm.py:115-121: nn.Conv2d(N*10//3, 3*K*N, 1)) as the
 * product library defines its arithmetic (include/flashgmm_amd.h section 2b): per output channel o and position p ONE chain of
 * binary32 fused multiply-adds, k ascending, starting from the bias -
 *     acc = bias[o];  for k = 0 .. c_in-1:  acc = fmaf(w[o][k], x[k][p], acc)
 * which is what v_mfma_f32_32x32x2_f32 computes.  A convolution is a sum and torch / MIOpen fix no order for it, so this is pinned
 * two ways: bit for bit against this chain, and within 1e-5 (relative to sum |w x|) of torch.nn.functional.conv2d in fp32.
 * w [n_out, c_in] row-major, bias [n_out] or NULL, x [c_in, hw], out [n_out, hw]. */
void fgo_head_params(int n_out, int c_in, int64_t hw, const float *w, const float *bias, const float *x, float *out) {
  for (int o = 0; o < n_out; ++o)
    for (int64_t p = 0; p < hw; ++p) {
      float acc = bias ? bias[o] : 0.0f;
      for (int k = 0; k < c_in; ++k) acc = fmaf(w[(int64_t)o * c_in + k], x[(int64_t)k * hw + p], acc);
      out[(int64_t)o * hw + p] = acc;
    }
}

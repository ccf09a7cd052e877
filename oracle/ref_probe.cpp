// TEST INFRASTRUCTURE ONLY. Probe of the REAL reference's float functions.
//
// This translation unit #includes the reference's own rans_interface.cpp *where it lies*
// (path injected by oracle/Makefile as REF_RANS_CPP; nothing is copied into this repo) so that the
// functions in its anonymous namespace become callable, and exports two extern "C" probes:
//
//   ref_probe_gmm_cdf   -> _fast_gmm_cdf<4>           (rans_interface.cpp:250-292), the float CDF pair
//   ref_probe_mode      -> get_approx_mode()/use_simd_path()   (rans_interface.cpp:99-130)
//
// APPROX_MODE / USE_SIMD are function-statics read once per process (rans_interface.cpp:100,120), so a
// caller that wants another mode must start another process (tests/golden/make_golden.py does).
#include REF_RANS_CPP

extern "C" {

// rows: v[i], mu/sigma/pi as (n,4) row-major.  c1 = cdf(v-0.5), c2 = cdf(v-0.5+1.0)  — exactly the call
// made at rans_interface.cpp:498-501.
void ref_probe_gmm_cdf(long n, const int *v, const float *mu, const float *sigma, const float *pi,
                       float *c1, float *c2) {
  for (long i = 0; i < n; ++i) {
    std::array<float, 4> m, s, w;
    for (int k = 0; k < 4; ++k) {
      m[k] = mu[4 * i + k];
      s[k] = sigma[4 * i + k];
      w[k] = pi[4 * i + k];
    }
    float a, b;
    std::tie(a, b) = _fast_gmm_cdf<4>(static_cast<float>(v[i]) - offset,
                                      static_cast<float>(v[i]) - offset + 1.0f, m, s, w);
    c1[i] = a;
    c2[i] = b;
  }
}

// Same, but at arbitrary float abscissae (tail / clamp-edge studies).
void ref_probe_gmm_cdf_x(long n, const float *x1, const float *x2, const float *mu, const float *sigma,
                         const float *pi, float *c1, float *c2) {
  for (long i = 0; i < n; ++i) {
    std::array<float, 4> m, s, w;
    for (int k = 0; k < 4; ++k) {
      m[k] = mu[4 * i + k];
      s[k] = sigma[4 * i + k];
      w[k] = pi[4 * i + k];
    }
    float a, b;
    std::tie(a, b) = _fast_gmm_cdf<4>(x1[i], x2[i], m, s, w);
    c1[i] = a;
    c2[i] = b;
  }
}

int ref_probe_mode(void) { return get_approx_mode() | (use_simd_path() ? 0x100 : 0); }

} // extern "C"

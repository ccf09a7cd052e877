// Dev aid (CPU only): how much does interleaving K independent rANS encoders in one thread buy?  The state update is a
// ~11-cycle dependent chain per symbol; K chains can share the core's issue width.  Same update as fgmm_rans.cpp.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
struct Rcp { uint64_t rcp; uint32_t shift, bias_add; };
static Rcp g_rcp[65536];
static void init_rcp() {
  g_rcp[0] = {0, 0, 0}; g_rcp[1] = {~0ull, 0, 65535};
  for (uint32_t f = 2; f < 65536; ++f) { uint32_t s = 0; while (f > (1u << s)) s++;
    const unsigned __int128 num = ((unsigned __int128)1 << (s + 63)) + f - 1; g_rcp[f] = {(uint64_t)(num / f), s - 1, 0}; }
}
static inline uint64_t mulhi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
struct Enc { uint64_t x; uint32_t* ptr;
  inline void put(uint32_t start, uint32_t freq) {
    const uint64_t x_max = (uint64_t)freq << 47; uint64_t xx = x;
    if (xx >= x_max) { *--ptr = (uint32_t)xx; xx >>= 32; }
    const Rcp& r = g_rcp[freq]; const uint64_t q = mulhi(xx, r.rcp) >> r.shift;
    x = xx + start + r.bias_add + q * (65536u - freq); } };
struct EncB { uint64_t x; uint32_t* ptr;  // branch-free renormalisation: always store, advance conditionally
  inline void put(uint32_t start, uint32_t freq) {
    const uint64_t x_max = (uint64_t)freq << 47; uint64_t xx = x;
    const bool r = xx >= x_max; ptr[-1] = (uint32_t)xx; ptr -= r; xx = r ? (xx >> 32) : xx;
    const Rcp& rc = g_rcp[freq]; const uint64_t q = mulhi(xx, rc.rcp) >> rc.shift;
    x = xx + start + rc.bias_add + q * (65536u - freq); } };
template <int K> static uint64_t runb(const std::vector<std::vector<uint32_t>>& tabs, std::vector<std::vector<uint32_t>>& outs, int first) {
  EncB e[K]; const int64_t n = (int64_t)tabs[0].size();
  for (int k = 0; k < K; ++k) e[k] = {1ull << 31, outs[first + k].data() + outs[first + k].size()};
  for (int64_t i = n - 1; i >= 0; --i) {
#pragma GCC unroll 4
    for (int k = 0; k < K; ++k) { const uint32_t ent = tabs[first + k][i]; e[k].put(ent & 0xFFFF, ent >> 16); }
  }
  uint64_t h = 0; for (int k = 0; k < K; ++k) h ^= e[k].x + (uint64_t)(outs[first + k].data() + outs[first + k].size() - e[k].ptr);
  return h;
}
template <int K> static uint64_t run(const std::vector<std::vector<uint32_t>>& tabs, std::vector<std::vector<uint32_t>>& outs, int first) {
  Enc e[K]; const int64_t n = (int64_t)tabs[0].size();
  for (int k = 0; k < K; ++k) e[k] = {1ull << 31, outs[first + k].data() + outs[first + k].size()};
  for (int64_t i = n - 1; i >= 0; --i) {
#pragma GCC unroll 4
    for (int k = 0; k < K; ++k) { const uint32_t ent = tabs[first + k][i]; e[k].put(ent & 0xFFFF, ent >> 16); }
  }
  uint64_t h = 0; for (int k = 0; k < K; ++k) h ^= e[k].x + (uint64_t)(outs[first + k].data() + outs[first + k].size() - e[k].ptr);
  return h;
}
int main() {
  init_rcp();
  const int S = 12; const int64_t n = 129024;
  std::vector<std::vector<uint32_t>> tabs(S), outs(S);
  uint64_t seed = 88172645463325252ull;
  auto rnd = [&] { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return seed; };
  for (int s = 0; s < S; ++s) { tabs[s].resize(n); outs[s].resize(n + 16);
    for (int64_t i = 0; i < n; ++i) { uint32_t f = 1 + (uint32_t)(rnd() % 12000); if (rnd() % 7 == 0) f = 1 + (uint32_t)(rnd() % 40);
      uint32_t st = (uint32_t)(rnd() % (65536 - f)); tabs[s][i] = st | (f << 16); } }
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int rep = 0; rep < 3; ++rep) {
    double t0 = now(); uint64_t h1 = 0; for (int s = 0; s < S; ++s) h1 ^= run<1>(tabs, outs, s); double t1 = now();
    uint64_t h2 = 0; for (int s = 0; s < S; s += 2) h2 ^= run<2>(tabs, outs, s); double t2 = now();
    uint64_t h3 = 0; for (int s = 0; s < S; s += 3) h3 ^= run<3>(tabs, outs, s); double t3 = now();
    uint64_t h4 = 0; for (int s = 0; s < S; s += 4) h4 ^= run<4>(tabs, outs, s); double t4 = now();
    double u0 = now(); uint64_t g1 = 0; for (int s = 0; s < S; ++s) g1 ^= runb<1>(tabs, outs, s); double u1 = now();
    uint64_t g2 = 0; for (int s = 0; s < S; s += 2) g2 ^= runb<2>(tabs, outs, s); double u2 = now();
    uint64_t g3 = 0; for (int s = 0; s < S; s += 3) g3 ^= runb<3>(tabs, outs, s); double u3 = now();
    uint64_t g4 = 0; for (int s = 0; s < S; s += 4) g4 ^= runb<4>(tabs, outs, s); double u4 = now();
    printf("branch-free K=1 %.2f  K=2 %.2f  K=3 %.2f  K=4 %.2f   (checks %d %d %d %d)\n", 1e9 * (u1 - u0) / (S * n), 1e9 * (u2 - u1) / (S * n),
           1e9 * (u3 - u2) / (S * n), 1e9 * (u4 - u3) / (S * n), (int)(h1 == g1), (int)(h1 == g2), (int)(h1 == g3), (int)(h1 == g4));
    printf("ns/symbol  K=1 %.2f  K=2 %.2f  K=3 %.2f  K=4 %.2f   (checks %d %d %d)\n", 1e9 * (t1 - t0) / (S * n), 1e9 * (t2 - t1) / (S * n),
           1e9 * (t3 - t2) / (S * n), 1e9 * (t4 - t3) / (S * n), (int)(h1 == h2), (int)(h1 == h3), (int)(h1 == h4));
  }
  return 0;
}

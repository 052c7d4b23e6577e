// field52.hpp (eight field elements per AVX-512 IFMA register, the host's share of a layer's sumcheck) against the scalar H4 loop of prover.hpp host_cubic_rounds: the three sums
// of every round, the binds, the final heads — bit for bit, k circuits x m elements, both fields (-DLASSO_BN254).  Prints "OK ..." / "SKIP" (no IFMA on this CPU); exit 1 on a mismatch.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../lasso_amd/host/field52.hpp"
using namespace lasso;
static uint64_t rng_state = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static H4 rand_h4(int kind) {
  H4 x;
  if (kind == 0) { memset(&x, 0, sizeof(x)); return x; }
  if (kind == 1) return h4_from(fr_one());
  if (kind == 2) { for (int i = 0; i < 4; i++) x.l[i] = g_fr_p64.p[i]; x.l[0] -= 1; return x; }   // p - 1
  for (int i = 0; i < 4; i++) x.l[i] = rnd();
  x.l[3] &= (1ull << 60) - 1;   // < 2^252 < p on both fields
  return x;
}
int main() {
#ifndef LASSO_HOST_IFMA
  printf("SKIP\n"); return 0;
#else
  if (!field52_ok()) { printf("SKIP\n"); return 0; }
  size_t checked = 0; double ns_scalar = 0, ns_vec = 0;
  for (size_t k : {1, 2, 3, 9, 16}) for (size_t lg = 1; lg <= 8; lg++) for (int rep = 0; rep < 3; rep++) {
    const size_t m = (size_t)1 << lg;
    std::vector<std::vector<H4>> a(k, std::vector<H4>(m)), b(k, std::vector<H4>(m)); std::vector<H4> w(k), cw(m);
    for (size_t c = 0; c < k; c++) { w[c] = rand_h4(rep == 0 ? 3 : 1 + (int)(rnd() % 3)); /* never zero: the prover keeps the scalar loop for a zero coefficient */ for (size_t i = 0; i < m; i++) { a[c][i] = rand_h4(rep == 2 ? (int)(rnd() % 4) : 3); b[c][i] = rand_h4(rep == 2 ? (int)(rnd() % 4) : 3); } }
    for (size_t i = 0; i < m; i++) cw[i] = rand_h4(rep == 2 ? (int)(rnd() % 4) : 3);
    auto t0 = std::chrono::steady_clock::now();
    HostRounds52 V(a, b, w, cw);
    auto t1 = std::chrono::steady_clock::now();
    ns_vec += std::chrono::duration<double, std::nano>(t1 - t0).count();
    // the scalar loop (prover.hpp host_cubic_rounds)
    std::vector<H4> A(k * m), B(k * m), Aw(k * m), C(cw);
    t0 = std::chrono::steady_clock::now();
    for (size_t c = 0; c < k; c++) for (size_t i = 0; i < m; i++) { A[c * m + i] = a[c][i]; B[c * m + i] = b[c][i]; Aw[c * m + i] = h4_mul(a[c][i], w[c]); }
    ns_scalar += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
    size_t mm = m;
    while (mm > 1) {
      const size_t h = mm / 2;
      t0 = std::chrono::steady_clock::now();
      H4 e0 = h4_zero(), e2 = h4_zero(), e3 = h4_zero();
      for (size_t i = 0; i < h; i++) {
        H4 s0 = h4_zero(), s2 = h4_zero(), s3 = h4_zero();
        for (size_t c = 0; c < k; c++) {
          const H4 *pa = &Aw[c * m], *pb = &B[c * m];
          const H4 da = h4_sub(pa[i + h], pa[i]), db = h4_sub(pb[i + h], pb[i]), a2 = h4_add(pa[i + h], da), b2 = h4_add(pb[i + h], db), a3 = h4_add(a2, da), b3 = h4_add(b2, db);
          s0 = h4_add(s0, h4_mul(pa[i], pb[i])); s2 = h4_add(s2, h4_mul(a2, b2)); s3 = h4_add(s3, h4_mul(a3, b3));
        }
        const H4 dc = h4_sub(C[i + h], C[i]), c2 = h4_add(C[i + h], dc), c3 = h4_add(c2, dc);
        e0 = h4_add(e0, h4_mul(s0, C[i])); e2 = h4_add(e2, h4_mul(s2, c2)); e3 = h4_add(e3, h4_mul(s3, c3));
      }
      t1 = std::chrono::steady_clock::now();
      H4 v0, v2, v3; V.sums(h, v0, v2, v3);
      auto t2 = std::chrono::steady_clock::now();
      if (memcmp(&e0, &v0, 32) || memcmp(&e2, &v2, 32) || memcmp(&e3, &v3, 32)) { printf("FAIL sums k=%zu m=%zu h=%zu rep=%d\n", k, m, h, rep); return 1; }
      const H4 r = rand_h4(rep == 2 ? (int)(rnd() % 4) : 3);
      auto t3 = std::chrono::steady_clock::now();
      for (size_t c = 0; c < k; c++) for (H4* arr : {&A[c * m], &B[c * m], &Aw[c * m]}) for (size_t i = 0; i < h; i++) arr[i] = h4_add(arr[i], h4_mul(r, h4_sub(arr[i + h], arr[i])));
      for (size_t i = 0; i < h; i++) C[i] = h4_add(C[i], h4_mul(r, h4_sub(C[i + h], C[i])));
      auto t4 = std::chrono::steady_clock::now();
      V.bind(h, r);
      auto t5 = std::chrono::steady_clock::now();
      ns_scalar += std::chrono::duration<double, std::nano>((t1 - t0) + (t4 - t3)).count(); ns_vec += std::chrono::duration<double, std::nano>((t2 - t1) + (t5 - t4)).count();
      mm = h; checked++;
    }
    std::vector<H4> hh; V.heads(hh);
    for (size_t c = 0; c < k; c++) if (memcmp(&hh[c], &A[c * m], 32) || memcmp(&hh[k + c], &B[c * m], 32)) { printf("FAIL heads k=%zu m=%zu rep=%d\n", k, m, rep); return 1; }
  }
  // the case the prover hands over by default: k = 2, m = 64
  {
    const size_t k = 2, m = 64; std::vector<std::vector<H4>> a(k, std::vector<H4>(m)), b(k, std::vector<H4>(m)); std::vector<H4> w(k), cw(m);
    for (size_t c = 0; c < k; c++) { w[c] = rand_h4(3); for (size_t i = 0; i < m; i++) { a[c][i] = rand_h4(3); b[c][i] = rand_h4(3); } }
    for (size_t i = 0; i < m; i++) cw[i] = rand_h4(3);
    const int N = 2000; H4 sink = h4_zero();
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < N; it++) { HostRounds52 V(a, b, w, cw); H4 x, y, z; for (size_t h = m / 2; h >= 1; h /= 2) { V.sums(h, x, y, z); V.bind(h, x); sink = h4_add(sink, z); } }
    const double per = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    printf("OK %zu rounds checked; scalar %.0f ns, ifma %.0f ns over the same rounds; a whole k=2 m=64 layer (6 rounds) on ifma: %.2f us (%llu)\n", checked, ns_scalar, ns_vec, per, sink.l[0] & 1);
  }
  return 0;
#endif
}

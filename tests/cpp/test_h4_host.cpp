// lasso_amd/host/field_host.hpp: the host-only 64-bit additions (fr_add_host / fr_sub_host) and the plain 4 x u64 form of the field the host's finishing rounds compute in
// (H4: h4_add / h4_sub / h4_mul) against the shared arithmetic the device executes (fr_add / fr_sub / fr_mul, lasso_amd/csrc/fr.cuh, bn254_fr.cuh), on random and edge values.
#include <cstdio>
#include <cstring>
#include "../../lasso_amd/host/field_host.hpp"
using namespace lasso;
static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static fr_t canon_rand() { fr_t t; for (;;) { uint64_t l[4] = {rnd(), rnd(), rnd(), rnd() >> 3}; memcpy(t.v, l, 32); if (!fr_geq_p(t.v)) return t; } }
int main() {
  fr_t pm1; { uint64_t l[4]; for (int i = 0; i < 4; i++) l[i] = ((uint64_t)fr_p_limb(2 * i + 1) << 32) | fr_p_limb(2 * i); l[0] -= 1; memcpy(pm1.v, l, 32); }
  fr_t edge[6] = {fr_zero(), fr_one(), pm1, fr_from_u64(1), fr_sub(pm1, fr_one()), fr_r2()};
  long n = 0;
  for (int it = 0; it < 200000; it++) {
    const fr_t a = it < 36 ? edge[it / 6] : canon_rand(), b = it < 36 ? edge[it % 6] : canon_rand();
    const fr_t s0 = fr_add(a, b), d0 = fr_sub(a, b), m0 = fr_mul(a, b);
    if (!fr_eq(fr_add_host(a, b), s0) || !fr_eq(fr_sub_host(a, b), d0)) { printf("FAIL host add/sub at %d\n", it); return 1; }
    const H4 x = h4_from(a), y = h4_from(b);
    if (!fr_eq(h4_to(h4_add(x, y)), s0) || !fr_eq(h4_to(h4_sub(x, y)), d0) || !fr_eq(h4_to(h4_mul(x, y)), m0)) { printf("FAIL h4 at %d\n", it); return 1; }
    n += 5;
  }
  printf("OK %ld comparisons\n", n);
  return 0;
}

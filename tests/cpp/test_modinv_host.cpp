// Host safegcd inversion (lasso_amd/host/modinv.hpp) against the Fermat chains of fr.cuh / fq.cuh, for both moduli of the path
// (curve25519 Fr and Fq): random values, the small integers, and the values next to the modulus; inverse(0) = 0.
#include <cstdio>
#include <cstring>
#include "../../lasso_amd/host/field_host.hpp"
using namespace lasso;
static uint64_t st = 88172645463325252ull;
static uint64_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
int main() {
  int bad = 0;
  for (int i = 0; i < 60000; i++) {
    Sc x = Sc::from_u64(rnd()) * Sc::from_u64(rnd()) + Sc::from_u64(rnd());
    if (i < 40) x = Sc::from_u64(i);
    if (i >= 40 && i < 80) x = Sc::zero() - Sc::from_u64(i - 39);
    Sc a, b; a.v = fr_inv_host(x.v); b.v = fr_inv(x.v);
    if (!(a == b)) { bad++; if (bad < 5) printf("Fr mismatch at %d\n", i); }
    if (i && !x.is_zero() && !((a * x) == Sc::one())) { bad++; if (bad < 5) printf("Fr x * x^-1 != 1 at %d\n", i); }
  }
  for (int i = 0; i < 60000; i++) {
    fq_t x = fq_zero(); for (int k = 0; k < 8; k++) x.v[k] = (uint32_t)rnd();   // any 256-bit value (lazily reduced form)
    if (i < 40) { x = fq_zero(); x.v[0] = i; }
    if (i >= 40 && i < 80) { x = fq_zero(); x.v[0] = i - 39; x = fq_neg(x); }
    fq_t a = fq_canonical(fq_inv_host(x)), b = fq_canonical(fq_inv(x));
    if (memcmp(a.v, b.v, 32)) { bad++; if (bad < 5) printf("Fq mismatch at %d\n", i); }
  }
  printf("%s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}

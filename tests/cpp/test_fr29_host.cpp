// CPU check of lasso_amd/csrc/fr29.cuh (the 29-bit-limb Fr form the polynomial kernels compute in) against fr.cuh's Montgomery
// arithmetic, which tests/cpp/test_arith_host.cpp pins to the oracle.  Exercises the bounds the kernels rely on (loose operands,
// negative values, sums of thousands of terms through the column form).
#include "../../lasso_amd/csrc/fr29.cuh"
#include <random>
#include <cstdio>
#include <cstring>
#include <vector>

#ifdef LASSO_BN254
#define FR_TOP_SHAVE 2   // 254-bit modulus
#else
#define FR_TOP_SHAVE 3   // 253-bit modulus
#endif
static std::mt19937_64 rng(4242);
static fr_t rand_fr() { for (;;) { uint64_t l[4]; for (int i = 0; i < 4; i++) l[i] = rng(); l[3] &= (~0ull) >> FR_TOP_SHAVE; fr_t t; memcpy(t.v, l, 32); if (!fr_geq_p(t.v)) return t; } }
static bool same(const fr_t& a, const fr_t& b) { return memcmp(a.v, b.v, 32) == 0; }
#define CHECK(c) do { if (!(c)) { printf("FAIL %s line %d\n", #c, __LINE__); return 1; } } while (0)

int main() {
  std::vector<fr_t> xs{fr_zero(), fr_one(), fr_neg(fr_one()), fr_from_u64(2), fr_from_u64(~0ull)};
  { fr_t pm1; for (int i = 0; i < 8; i++) pm1.v[i] = fr_p_limb(i); pm1.v[0] -= 1u; xs.push_back(pm1); }   // memory integer p-1
  for (int i = 0; i < 400; i++) xs.push_back(rand_fr());
  const size_t N = xs.size();
  for (size_t i = 0; i < N; i++) {
    const fr_t &a = xs[i], &b = xs[(i * 7 + 3) % N], &c = xs[(i * 11 + 5) % N];
    fr29 au = fr29_unpack_u(a), bu = fr29_unpack_u(b), cu = fr29_unpack_u(c), as = fr29_unpack_s(a), bs = fr29_unpack_s(b);
    CHECK(same(fr29_pack(au), a));                                   // unpack/pack round trip
    CHECK(same(fr29_store(au), a));
    CHECK(same(fr29_store(fr29_mul(au, bs)), fr_mul(a, b)));         // mul(u, s) = u-form of the product
    CHECK(same(fr29_store(fr29_mul(as, bu)), fr_mul(a, b)));
    CHECK(same(fr29_store(fr29_mul(au, fr29_one_s())), a));          // ONE_S is the multiplicative identity of the radix
    // bind: lo + r*(hi - lo) with a negative difference
    CHECK(same(fr29_store(fr29_add(au, fr29_mul(fr29_sub(bu, au), fr29_unpack_s(c)))), fr_add(a, fr_mul(c, fr_sub(b, a)))));
    // cubic term at x = 3: (3hi - 2lo) products, deficient by 2^10, corrected once with K10
    {
      fr29 da = fr29_sub(bu, au), a2 = fr29_weak(fr29_add(bu, da)), a3 = fr29_add(a2, da);          // a3 loose
      fr29 dc = fr29_sub(au, cu), c2 = fr29_weak(fr29_add(au, dc)), c3 = fr29_weak(fr29_add(c2, dc));
      fr29 t = fr29_mul(fr29_mul(a3, c3), a3);
      fr_t a3r = fr_add(fr_add(b, fr_sub(b, a)), fr_sub(b, a)), c3r = fr_add(fr_add(a, fr_sub(a, c)), fr_sub(a, c));
      CHECK(same(fr29_store(fr29_mul(t, fr29_k10())), fr_mul(fr_mul(a3r, c3r), a3r)));
    }
    CHECK(same(fr29_store(fr29_mul(fr29_mul(fr29_mul(au, bs), cu), fr29_k5())), fr_mul(fr_mul(a, b), c)));   // one 2^5 short -> K5
    // canonical over the whole allowed range (-4p, 4p)
    CHECK(same(fr29_store(fr29_sub(fr29_sub(au, bu), fr29_add(cu, bu))), fr_sub(fr_sub(a, b), fr_add(c, b))));
    CHECK(same(fr29_store(fr29_add(fr29_add(au, bu), fr29_add(cu, au))), fr_add(fr_add(a, b), fr_add(c, a))));
    CHECK(same(fr29_store(fr29_sub(fr29_zero(), au)), fr_neg(a)));
    // from_u64 via R2S
    uint64_t x = rng() >> (i % 64);
    CHECK(same(fr29_store(fr29_mul(fr29_from_u64_int(x), fr29_r2s())), fr_from_u64(x)));
    // to canonical integer: mul by the integer 2^5 (as limbs) = x*2^256*2^5/2^261 = x
    { fr29 k32 = fr29_zero(); k32.v[0] = 32; CHECK(same(fr29_store(fr29_mul(au, k32)), fr_to_canonical(a))); }
  }
  // long accumulations through 64-bit columns: sum of T products, T up to 2^16, vs the reference sum
  for (int T : {1, 2, 255, 4096, 65536}) {
    int64_t col[9] = {0}; fr_t ref = fr_zero();
    for (int i = 0; i < T; i++) {
      const fr_t &a = xs[(size_t)i % N], &b = xs[((size_t)i * 13 + 1) % N];
      fr29 t = fr29_mul(fr29_unpack_u(a), fr29_unpack_s(b));
      for (int k = 0; k < 9; k++) col[k] += t.v[k];
      ref = fr_add(ref, fr_mul(a, b));
    }
    fr29 s = fr29_from_columns(col);
    CHECK(same(fr29_store(fr29_mul(s, fr29_one_s())), ref));
  }
  // per-thread style accumulation: weak after every add, fold with ONE_S every 128 terms
  {
    fr29 e = fr29_zero(); fr_t ref = fr_zero();
    for (int i = 0; i < 1000; i++) {
      const fr_t &a = xs[(size_t)i % N], &b = xs[((size_t)i * 5 + 2) % N];
      e = fr29_weak(fr29_add(e, fr29_mul(fr29_unpack_u(a), fr29_unpack_s(b))));
      if ((i & 127) == 127) e = fr29_mul(e, fr29_one_s());
      ref = fr_add(ref, fr_mul(a, b));
    }
    CHECK(same(fr29_store(fr29_mul(e, fr29_one_s())), ref));
  }
  // lazily reduced memory form (fr29_semi): same residue, digits in range, survives pack/unpack, accepted by fr29_canonical and as a product
  // operand; inputs over the whole allowed range |value| < 2^255 (sums and differences of up to eight elements, loose limbs)
  for (size_t i = 0; i < N; i++) {
    const fr_t &a = xs[i], &b = xs[(i * 7 + 3) % N], &c = xs[(i * 11 + 5) % N];
    const fr29 au = fr29_unpack_u(a), bu = fr29_unpack_u(b), cu = fr29_unpack_u(c);
    // eight-element sums: two four-element halves, each carried once so that every limb stays below 2^31 (the contract of the reductions)
    const fr29 h1 = fr29_weak(fr29_add(fr29_add(au, bu), fr29_add(cu, au))), h2 = fr29_weak(fr29_add(fr29_add(bu, bu), fr29_add(cu, cu)));
    const fr29 cases[6] = {au, fr29_add(au, bu), fr29_sub(au, bu), fr29_sub(fr29_sub(fr29_zero(), au), fr29_add(bu, cu)),
                           fr29_add(h1, h2), fr29_sub(fr29_zero(), fr29_add(h1, h2))};
    for (const fr29& x : cases) {
      const fr29 sm = fr29_semi(x);
      for (int k = 0; k < 8; k++) CHECK(sm.v[k] >= 0 && sm.v[k] < (1 << 29));
      CHECK(sm.v[8] >= 0 && sm.v[8] <= (1 << 22));
      CHECK(same(fr29_store(sm), fr29_store(x)));                         // same residue
      const fr_t mem = fr29_pack(sm); const fr29 back = fr29_unpack_u(mem);
      for (int k = 0; k < 9; k++) CHECK(back.v[k] == sm.v[k]);             // memory round trip keeps the digits
      CHECK(same(fr29_store(fr29_mul(sm, fr29_unpack_s(c))), fr_mul(fr29_store(x), c)));   // a valid product operand
      // a bind on lazily reduced inputs stays in range: lo + r (hi - lo), then semi again
      const fr29 sm2 = fr29_semi(fr29_add(sm, fr29_mul(fr29_sub(fr29_unpack_u(fr29_pack(fr29_semi(cu))), sm), fr29_unpack_s(b))));
      CHECK(sm2.v[8] >= 0 && sm2.v[8] <= (1 << 22));
      CHECK(same(fr29_store(sm2), fr_add(fr29_store(x), fr_mul(b, fr_sub(c, fr29_store(x))))));
    }
  }
  // block sums: column values times 2^shift reduced without a product (fr29_reduce_columns) == the old from_columns + product with 2^(261+shift)
  for (int T : {1, 2, 255, 4096, 65536, 1 << 20}) {
    int64_t col[9] = {0}, ncol[9] = {0};
    for (int i = 0; i < T; i++) {
      const fr_t &a = xs[(size_t)i % N], &b = xs[((size_t)i * 13 + 1) % N];
      fr29 t = fr29_mul(fr29_unpack_u(a), fr29_unpack_s(b));
      if (i % 5 == 4) t = fr29_add(t, fr29_add(t, t));                    // limbs up to 3 * 2^29, limb 8 grows
      for (int k = 0; k < 9; k++) { col[k] += t.v[k]; ncol[k] -= t.v[k]; }
    }
    const fr29 fixes[3] = {fr29_one_s(), fr29_k5(), fr29_k10()}; const int shifts[3] = {0, 5, 10};
    for (int v = 0; v < 3; v++) {
      if (T > 65536 && v) continue;                                          // the old path's bounds end there
      const fr_t want = T <= 65536 ? fr29_store(fr29_mul(fr29_from_columns(col), fixes[v])) : fr29_store(fr29_mul(fr29_mul(fr29_from_columns(col), fr29_one_s()), fixes[v]));
      CHECK(same(fr29_pack(fr29_reduce_columns(col, shifts[v])), want));
      CHECK(same(fr29_pack(fr29_reduce_columns(ncol, shifts[v])), fr_neg(want)));
    }
  }
  // sums of products through the double-width accumulator (fr29_mul_acc / fr29_acc_carry / fr29_acc_reduce): carry pass every third product,
  // signed differences as operands (the cubic rounds' leading-coefficient term), T up to 2^16, vs the reference sum of Montgomery products
  for (int T : {1, 2, 3, 4, 255, 4096, 65536}) {
    fr29_acc w = fr29_acc_zero(), wd = fr29_acc_zero(); fr_t ref = fr_zero(), refd = fr_zero(); int pend = 0;
    for (int i = 0; i < T; i++) {
      const fr_t &a = xs[(size_t)i % N], &b = xs[((size_t)i * 13 + 1) % N], &c = xs[((size_t)i * 5 + 2) % N], &d = xs[((size_t)i * 3 + 4) % N];
      const fr29 au = fr29_unpack_u(a), bu = fr29_unpack_u(b), cu = fr29_unpack_u(c), du = fr29_unpack_u(d);
      fr29_mul_acc(w, au, fr29_mul(bu, fr29_unpack_s(c)));                  // canonical x reduced product
      fr29_mul_acc(wd, fr29_sub(au, bu), fr29_sub(cu, du));                 // differences: |limb| < 2^29, both signs
      if (++pend == 3) { fr29_acc_carry(w); fr29_acc_carry(wd); pend = 0; }
      ref = fr_add(ref, fr_mul(a, fr_mul(b, c)));
      refd = fr_add(refd, fr_mul(fr_sub(a, b), fr_sub(c, d)));
    }
    fr29_acc_carry(w); fr29_acc_carry(wd);
    // u * u products are 2^5 short of memory form: K5 restores it (as the kernels do once per block)
    CHECK(same(fr29_store(fr29_mul(fr29_acc_reduce(w), fr29_k5())), ref));
    CHECK(same(fr29_store(fr29_mul(fr29_acc_reduce(wd), fr29_k5())), refd));
  }
  printf("OK\n");
  return 0;
}

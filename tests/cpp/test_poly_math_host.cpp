// CPU check (and, built with -fsanitize=undefined, a bounds check) of the per-index arithmetic of the sumcheck kernels in
// lasso_amd/csrc/poly_kernels.cuh.  The pure-arithmetic device functions and accumulation macros are EXTRACTED from the kernel source by
// tests/test_host_arith_cpp.py into poly_math_extract.hpp (so this test follows the product's text, it does not restate it); here they are driven
// the way the kernels' per-thread loops drive them — 256 "threads", strided indices, the block's column sums and the shift-corrected reduction at
// the end — and compared with fr.cuh's Montgomery arithmetic.  Arrays carry the lazily reduced memory form from one round to the next.
#include "../../lasso_amd/csrc/fr29.cuh"
#define __device__
#define __forceinline__ inline
#include "poly_math_extract.hpp"
#include <random>
#include <cstdio>
#include <cstring>
#include <vector>

#ifdef LASSO_BN254
#define FR_TOP_SHAVE 2
#else
#define FR_TOP_SHAVE 3
#endif
static std::mt19937_64 rng(99);
static fr_t rand_fr() { for (;;) { uint64_t l[4]; for (int i = 0; i < 4; i++) l[i] = rng(); l[3] &= (~0ull) >> FR_TOP_SHAVE; fr_t t; memcpy(t.v, l, 32); if (!fr_geq_p(t.v)) return t; } }
static fr_t edge(int k) { fr_t pm1; for (int i = 0; i < 8; i++) pm1.v[i] = fr_p_limb(i); pm1.v[0] -= 1u; return k % 3 == 0 ? pm1 : (k % 3 == 1 ? fr_zero() : fr_neg(fr_from_u64(2))); }
static bool same(const fr_t& a, const fr_t& b) { return memcmp(a.v, b.v, 32) == 0; }
#define CHECK(c) do { if (!(c)) { printf("FAIL %s line %d\n", #c, __LINE__); return 1; } } while (0)
static const int NTHREADS = 256;

// what cubic_epilogue does with the threads' accumulators: limb-wise column sums over the block, then the shift-corrected reduction
static fr_t block_sum(const std::vector<fr29>& acc, int shift) {
  int64_t col[9] = {0};
  for (const fr29& a : acc) for (int k = 0; k < 9; k++) col[k] += a.v[k];
  return fr29_pack(fr29_reduce_columns(col, shift));
}

int main() {
  for (int trial = 0; trial < 3; trial++) {
    // ---- k_cubic_round_lb: three polynomials, sums of a b c at x = 0, 2, 3 (u*u*u products, corrected by 2^10 at the end)
    {
      const size_t half = 1 << 12;
      std::vector<fr_t> A(2 * half), B(2 * half), Cc(2 * half);
      for (size_t i = 0; i < 2 * half; i++) { A[i] = trial == 2 && i % 5 == 0 ? edge((int)i) : rand_fr(); B[i] = trial == 2 && i % 7 == 0 ? edge((int)i + 1) : rand_fr(); Cc[i] = trial == 2 && i % 3 == 0 ? edge((int)i + 2) : rand_fr(); }
      std::vector<fr29> e0(NTHREADS, fr29_zero()), e2(NTHREADS, fr29_zero()), e3(NTHREADS, fr29_zero());
      for (int t = 0; t < NTHREADS; t++) {
        fr29 e[3] = {fr29_zero(), fr29_zero(), fr29_zero()}; uint32_t cnt = 0;
        for (size_t i = t; i < half; i += NTHREADS) {
          fr29 t0, t2, t3;
          cubic_terms(fr29_unpack_u(A[i]), fr29_unpack_u(A[i + half]), fr29_unpack_u(B[i]), fr29_unpack_u(B[i + half]), fr29_unpack_u(Cc[i]), fr29_unpack_u(Cc[i + half]), t0, t2, t3);
          CUBIC_ACCUMULATE(e, t0, t2, t3, cnt);
        }
        e0[t] = e[0]; e2[t] = e[1]; e3[t] = e[2];
      }
      fr_t r0 = fr_zero(), r2 = fr_zero(), r3 = fr_zero();
      for (size_t i = 0; i < half; i++) {
        auto at = [](const fr_t& lo, const fr_t& hi, int x) { fr_t d = fr_sub(hi, lo), v = lo; for (int k = 0; k < x; k++) v = fr_add(v, d); return v; };
        r0 = fr_add(r0, fr_mul(Cc[i], fr_mul(A[i], B[i])));
        r2 = fr_add(r2, fr_mul(at(Cc[i], Cc[i + half], 2), fr_mul(at(A[i], A[i + half], 2), at(B[i], B[i + half], 2))));
        r3 = fr_add(r3, fr_mul(at(Cc[i], Cc[i + half], 3), fr_mul(at(A[i], A[i + half], 3), at(B[i], B[i + half], 3))));
      }
      CHECK(same(block_sum(e0, 10), r0)); CHECK(same(block_sum(e2, 10), r2)); CHECK(same(block_sum(e3, 10), r3));
    }
    // ---- the eq-weighted fused rounds, three consecutive rounds on the same arrays: bind with the previous challenge (stored lazily reduced),
    // the two sums q(0) = sum E a0 b0 and q_inf = sum E (a1 - a0)(b1 - b0) through the double-width accumulators (k_cubic_eqw_fused<2, true>),
    // the three-sum form (k_cubic_eqw_fused<3>) and the narrow two-sum form (CUBIC_ACCUMULATE2) next to it
    {
      size_t q = 1 << 11;
      std::vector<fr_t> a(4 * q), b(4 * q), E(q), ra(4 * q), rb(4 * q);
      for (size_t i = 0; i < 4 * q; i++) { a[i] = trial == 2 && i % 4 == 0 ? edge((int)i) : rand_fr(); b[i] = trial == 2 && i % 6 == 0 ? edge((int)i + 1) : rand_fr(); ra[i] = a[i]; rb[i] = b[i]; }
      for (size_t i = 0; i < q; i++) E[i] = rand_fr();
      for (int round = 0; round < 3; round++, q /= 2) {
        const fr_t r = round == 1 ? edge(0) : rand_fr();   // one round with the challenge p - 1
        const fr29 rs = fr29_unpack_s(r);
        std::vector<fr29> w0s(NTHREADS), w1s(NTHREADS), n0(NTHREADS), n1(NTHREADS), t0s(NTHREADS), t2s(NTHREADS), t3s(NTHREADS);
        std::vector<fr_t> na(2 * q), nb(2 * q);
        for (int t = 0; t < NTHREADS; t++) {
          fr29_acc w0 = fr29_acc_zero(), w1 = fr29_acc_zero(); uint32_t cnt = 0, cn = 0, c3 = 0;
          fr29 en[2] = {fr29_zero(), fr29_zero()}, e3[3] = {fr29_zero(), fr29_zero(), fr29_zero()};
          for (size_t i = t; i < q; i += NTHREADS) {
            const fr29 a0 = bind29_semi(a[i], a[i + 2 * q], rs), a1 = bind29_semi(a[i + q], a[i + 3 * q], rs);
            const fr29 b0 = bind29_semi(b[i], b[i + 2 * q], rs), b1 = bind29_semi(b[i + q], b[i + 3 * q], rs);
            na[i] = fr29_pack(a0); na[i + q] = fr29_pack(a1); nb[i] = fr29_pack(b0); nb[i + q] = fr29_pack(b1);
            const fr29 es = fr29_unpack_s(E[i]);
            const fr29 g0 = fr29_mul(a0, es), g1 = fr29_mul(a1, es);
            fr29_mul_acc(w0, b0, g0); fr29_mul_acc(w1, fr29_sub(g1, g0), fr29_sub(b1, b0));
            if (++cnt == 3) { fr29_acc_carry(w0); fr29_acc_carry(w1); cnt = 0; }
            fr29 u0, uinf; cubic_eqw_terms2(a0, a1, b0, b1, es, u0, uinf); CUBIC_ACCUMULATE2(en, u0, uinf, cn);
            // the three-sum kernel binds to canonical values
            fr29 v0, v2, v3; cubic_eqw_terms(bind29(a[i], a[i + 2 * q], rs), bind29(a[i + q], a[i + 3 * q], rs), bind29(b[i], b[i + 2 * q], rs), bind29(b[i + q], b[i + 3 * q], rs), es, v0, v2, v3);
            CUBIC_ACCUMULATE(e3, v0, v2, v3, c3);
          }
          fr29_acc_carry(w0); fr29_acc_carry(w1); w0s[t] = fr29_acc_reduce(w0); w1s[t] = fr29_acc_reduce(w1);
          n0[t] = en[0]; n1[t] = en[1]; t0s[t] = e3[0]; t2s[t] = e3[1]; t3s[t] = e3[2];
        }
        // reference: canonical binds, then the sums
        std::vector<fr_t> ca(2 * q), cb(2 * q);
        for (size_t i = 0; i < 2 * q; i++) { ca[i] = fr_add(ra[i], fr_mul(r, fr_sub(ra[i + 2 * q], ra[i]))); cb[i] = fr_add(rb[i], fr_mul(r, fr_sub(rb[i + 2 * q], rb[i]))); }
        fr_t q0 = fr_zero(), qi = fr_zero(), q2 = fr_zero(), q3 = fr_zero();
        for (size_t i = 0; i < q; i++) {
          const fr_t da = fr_sub(ca[i + q], ca[i]), db = fr_sub(cb[i + q], cb[i]);
          q0 = fr_add(q0, fr_mul(E[i], fr_mul(ca[i], cb[i]))); qi = fr_add(qi, fr_mul(E[i], fr_mul(da, db)));
          const fr_t a2 = fr_add(ca[i + q], da), b2 = fr_add(cb[i + q], db);
          q2 = fr_add(q2, fr_mul(E[i], fr_mul(a2, b2))); q3 = fr_add(q3, fr_mul(E[i], fr_mul(fr_add(a2, da), fr_add(b2, db))));
        }
        CHECK(same(block_sum(w0s, 5), q0)); CHECK(same(block_sum(w1s, 5), qi));
        CHECK(same(block_sum(n0, 5), q0)); CHECK(same(block_sum(n1, 5), qi));
        CHECK(same(block_sum(t0s, 5), q0)); CHECK(same(block_sum(t2s, 5), q2)); CHECK(same(block_sum(t3s, 5), q3));
        // the stored (lazily reduced) arrays are the same residues, and they are the next round's inputs
        for (size_t i = 0; i < 2 * q; i++) { CHECK(same(fr29_store(fr29_unpack_u(na[i])), ca[i])); CHECK(same(fr29_store(fr29_unpack_u(nb[i])), cb[i])); }
        a = na; b = nb; ra = ca; rb = cb;
      }
    }
    // ---- LT combine (products of up to 16 s-form values) against the reference sum_i LT_i prod_{j<i} EQ_j
    for (uint32_t c : {1u, 2u, 4u, 8u}) {
      fr_t vals_m[16]; fr29 vals[16];
      for (int i = 0; i < 16; i++) { vals_m[i] = trial == 2 && i % 3 == 0 ? edge(i) : rand_fr(); vals[i] = fr29_unpack_s(vals_m[i]); }
      fr_t ref = fr_zero(), prod = fr_one();
      for (uint32_t i = 0; i < c; i++) { ref = fr_add(ref, fr_mul(vals_m[2 * i], prod)); prod = fr_mul(prod, vals_m[2 * i + 1]); }
      // s-form in (x 2^261), s-form out: a product with the integer 2^256 gives the memory form x 2^256 back
      fr29 k256 = fr29_zero(); k256.v[8] = 1 << 24;
      CHECK(same(fr29_store(fr29_mul(combine_lt<16>(vals, c), k256)), ref));
    }
    // ---- the LT ROUND in Horner form (k_combine_round_lt, round 3), driven as the kernel drives it: the LT arrays pre-multiplied by kappa_m = 32^-(C-1-m) (k_lt_prescale),
    // T lanes per index each walking PPG consecutive points from x0 = lane * PPG (lt_line_at), the memories from the last to the first with u-form lines stepped by addition,
    // one product per memory and point, then the eq weight; sums over the indices of a lane folded every LT_FOLD_EVERY, times 32^C, then the block's column sums.
    // Reference: sum_i e(x) sum_m LT_m(x) prod_{j<m} EQ_j(x), every line evaluated at x = 0..degree by fr.cuh.
    // Trial 2 is the magnitude worst case: every line runs from 0 to p - 1 or from p - 1 to 0, so the stepped values reach 18 (p - 1) with either sign.
    for (uint32_t c : {1u, 2u, 5u, 8u, 16u}) {
      const uint32_t degree = c + 1, T = degree <= 5 ? 1 : (degree <= 9 ? 2 : 3), D = degree <= 2 ? 2 : degree <= 3 ? 3 : degree <= 5 ? 5 : degree <= 9 ? 9 : 17, PPG = (D + 1 + T - 1) / T;
      const uint32_t SLOTS = NTHREADS / T;
      const size_t half = c == 16 ? (size_t)SLOTS * (LT_FOLD_EVERY() + 3) : 1024;   // C = 16: every lane runs through a whole fold period
      std::vector<std::vector<fr_t>> P(2 * c, std::vector<fr_t>(2 * half)); std::vector<fr_t> E(2 * half);
      for (size_t i = 0; i < 2 * half; i++) {
        const bool hi = i >= half;
        E[i] = trial == 2 ? (((i & 1) != 0) == hi ? edge(0) : fr_zero()) : rand_fr();
        for (uint32_t m = 0; m < 2 * c; m++) P[m][i] = trial == 2 ? ((((i + m) & 1) != 0) == hi ? edge(0) : fr_zero()) : (trial == 1 && (i + m) % 3 == 0 ? edge((int)(i + m)) : rand_fr());
      }
      const fr_t inv32 = fr_inv(fr_from_u64(32)); fr_t scale = fr_one(), kk = fr_one();
      std::vector<std::vector<fr_t>> Ps = P;     // what k_lt_prescale leaves in memory
      for (uint32_t m = c; m-- > 0;) { const fr29 ks = fr29_unpack_s(kk); for (size_t i = 0; i < 2 * half; i++) Ps[2 * m][i] = fr29_store(fr29_mul(fr29_unpack_u(P[2 * m][i]), ks)); kk = fr_mul(kk, inv32); }
      for (uint32_t m = 0; m < c; m++) scale = fr_mul(scale, fr_from_u64(32));
      std::vector<std::vector<fr29>> acc(degree + 1, std::vector<fr29>(NTHREADS, fr29_zero()));
      for (int th = 0; th < NTHREADS; th++) {
        const uint32_t slot = th / T, pg = th - slot * T, x0 = pg * PPG;
        if (slot >= SLOTS || x0 > degree) continue;
        fr29 sum[6], t[6]; for (auto& x : sum) x = fr29_zero();
        uint32_t cnt = 0;
        for (size_t i = slot; i < half; i += SLOTS) {
          {
            const fr29 lo = fr29_unpack_u(Ps[2 * (c - 1)][i]), dlt = fr29_sub(fr29_unpack_u(Ps[2 * (c - 1)][i + half]), lo);
            fr29 lt = lt_line_at(lo, dlt, x0);
            for (uint32_t k = 0; k < PPG; k++) if (x0 + k <= degree) { t[k] = lt; lt = lt_line_step(lt, dlt); }
          }
          for (uint32_t m = c - 1; m-- > 0;) {
            const fr29 lo = fr29_unpack_u(Ps[2 * m][i]), dlt = fr29_sub(fr29_unpack_u(Ps[2 * m][i + half]), lo);
            const fr29 eo = fr29_unpack_u(Ps[2 * m + 1][i]), deq = fr29_sub(fr29_unpack_u(Ps[2 * m + 1][i + half]), eo);
            fr29 lt = lt_line_at(lo, dlt, x0), eqv = lt_line_at(eo, deq, x0);
            for (uint32_t k = 0; k < PPG; k++) if (x0 + k <= degree) { t[k] = lt_horner_step(lt, eqv, t[k]); lt = lt_line_step(lt, dlt); eqv = lt_line_step(eqv, deq); }
          }
          {
            const fr29 e0 = fr29_unpack_u(E[i]), edif = fr29_sub(fr29_unpack_u(E[i + half]), e0);
            fr29 ecur = lt_line_at(e0, edif, x0);
            for (uint32_t k = 0; k < PPG; k++) if (x0 + k <= degree) { sum[k] = lt_weighted_acc(sum[k], ecur, t[k]); ecur = lt_line_step(ecur, edif); }
          }
          if (++cnt >= LT_FOLD_EVERY()) { cnt = 0; for (uint32_t k = 0; k < PPG; k++) sum[k] = fr29_mul(sum[k], fr29_one_s()); }
        }
        for (uint32_t k = 0; k < PPG; k++) if (x0 + k <= degree) acc[x0 + k][th] = fr29_mul(sum[k], fr29_unpack_s(scale));
      }
      auto at = [](const fr_t& lo, const fr_t& hi, uint32_t x) { fr_t d = fr_sub(hi, lo), v = lo; for (uint32_t k = 0; k < x; k++) v = fr_add(v, d); return v; };
      for (uint32_t k = 0; k <= degree; k++) {
        fr_t ref = fr_zero();
        for (size_t i = 0; i < half; i++) {
          fr_t g = fr_zero(), prod = fr_one();
          for (uint32_t m = 0; m < c; m++) { g = fr_add(g, fr_mul(at(P[2 * m][i], P[2 * m][i + half], k), prod)); prod = fr_mul(prod, at(P[2 * m + 1][i], P[2 * m + 1][i + half], k)); }
          ref = fr_add(ref, fr_mul(g, at(E[i], E[i + half], k)));
        }
        CHECK(same(block_sum(acc[k], 0), ref));
      }
    }
  }
  // ---- the FIRST LT round from bits (k_combine_round_lt_u32): exact 128-bit integer Horner walk per point, then (integer as three signed limbs) x (eq line x 2^517);
  // reference as above on the lifted field elements.  Patterns drive |t| to its extreme 17 (17^16 - 1) / 16 at C = 16.
  for (int pat = 0; pat < 3; pat++) for (uint32_t c : {1u, 2u, 5u, 16u}) {
    const uint32_t degree = c + 1; const size_t half = 256;
    std::vector<std::vector<uint32_t>> U(2 * c, std::vector<uint32_t>(2 * half)); std::vector<fr_t> E(2 * half);
    for (size_t i = 0; i < 2 * half; i++) { const bool hi = i >= half; E[i] = pat == 2 && i % 3 == 0 ? edge((int)i) : rand_fr();
      for (uint32_t m = 0; m < 2 * c; m++) U[m][i] = pat == 0 ? (uint32_t)(rng() & 1) : pat == 1 ? (hi ? 1u : 0u) : ((((m + 1) & 1) != 0) == !hi ? 1u : 0u); }
    std::vector<std::vector<fr29>> acc(degree + 1, std::vector<fr29>(NTHREADS, fr29_zero()));
    for (int th = 0; th < NTHREADS; th++) {
      fr29 sum[18]; for (auto& x : sum) x = fr29_zero();
      for (size_t i = th; i < half; i += NTHREADS) {
        __int128 t[18];
        { const int32_t lo = (int32_t)U[2 * (c - 1)][i], d = (int32_t)U[2 * (c - 1)][i + half] - lo; for (uint32_t k = 0; k <= degree; k++) t[k] = lo + (int32_t)k * d; }
        for (uint32_t m = c - 1; m-- > 0;) {
          const int32_t llo = (int32_t)U[2 * m][i], ld = (int32_t)U[2 * m][i + half] - llo, elo = (int32_t)U[2 * m + 1][i], ed = (int32_t)U[2 * m + 1][i + half] - elo;
          for (uint32_t k = 0; k <= degree; k++) { const int32_t x = (int32_t)k; t[k] = (__int128)(llo + x * ld) + (__int128)(elo + x * ed) * t[k]; }
        }
        const fr29 e0 = fr29_weak(lt_eq_times_r2(E[i])), edif = fr29_sub(lt_eq_times_r2(E[i + half]), e0);
        fr29 ecur = lt_line_at(e0, edif, 0);
        for (uint32_t k = 0; k <= degree; k++) { sum[k] = lt_weighted_acc(sum[k], lt_int_limbs(t[k]), ecur); ecur = lt_line_step(ecur, edif); }
      }
      for (uint32_t k = 0; k <= degree; k++) acc[k][th] = sum[k];
    }
    auto at = [](const fr_t& lo, const fr_t& hi, uint32_t x) { fr_t d = fr_sub(hi, lo), v = lo; for (uint32_t k = 0; k < x; k++) v = fr_add(v, d); return v; };
    for (uint32_t k = 0; k <= degree; k++) {
      fr_t ref = fr_zero();
      for (size_t i = 0; i < half; i++) {
        fr_t g = fr_zero(), prod = fr_one();
        for (uint32_t m = 0; m < c; m++) { g = fr_add(g, fr_mul(at(fr_from_u64(U[2 * m][i]), fr_from_u64(U[2 * m][i + half]), k), prod)); prod = fr_mul(prod, at(fr_from_u64(U[2 * m + 1][i]), fr_from_u64(U[2 * m + 1][i + half]), k)); }
        ref = fr_add(ref, fr_mul(g, at(E[i], E[i + half], k)));
      }
      CHECK(same(block_sum(acc[k], 0), ref));
    }
  }
  printf("OK\n");
  return 0;
}

// Eight field elements at a time for the rounds the HOST finishes (prover.hpp host_cubic_rounds): Fr in five 52-bit limbs, one element per 64-bit lane of a 512-bit register,
// products by AVX-512 IFMA (vpmadd52luq / vpmadd52huq: 52 x 52 -> low / high 52 bits, accumulated) — round 5.
//
// Why: when a grand-product layer's arrays are down to a few dozen elements the resident kernel hands them to the host (lasso_tail_handover_next), which finishes the layer with the
// literal loop of sumcheck.rs:49-124.  With the scalar Montgomery product (field_host.hpp h4_mul, 22 ns) four such rounds over 2 x 2 x 16 elements cost 18 us, three quarters of the
// time the device then idles between two layers, and taking the layer over any earlier costs more than the device turns it saves.  Here a product is ~150 vector instructions for
// eight elements (~2.5 ns each), so the host can take a layer over at 64 elements per array and prove the trees' tops up to 256 elements without the device.
//
// Representation: x is held as  x * 2^260 mod p  in limbs l0..l4 < 2^52 (canonical: the value is < p) — the Montgomery domain of the radix this arithmetic reduces in.  ark-ff's
// form (x * 2^256 mod p, field_host.hpp Sc / H4) goes in through one product with 2^264 mod p and comes out through one product with 2^256 mod p; both forms are canonical, so what
// comes out is bit for bit what the scalar loop computes (tests/cpp/test_field52_host.cpp: sums, binds and heads against the H4 loop on both fields).
// Every operation takes canonical operands and returns a canonical result: add / sub are a limb-wise operation, one carry chain and one conditional subtraction; mul is the
// word-serial Montgomery product (five steps of: accumulate a * b_i, m = t0 * (-p^-1) mod 2^52, accumulate m * p, drop the zero limb), result < p + p^2 / 2^260 < 2p.
// Compiled for avx512f + avx512ifma inside a library built for x86-64-v3 and selected at run time (field52_ok(); LASSO_HOST_IFMA=0 switches it off): CPUs without IFMA keep the
// scalar loop and its smaller take-over sizes.
#pragma once
#include "field_host.hpp"
#if defined(__x86_64__) && defined(__GNUC__) && defined(LASSO_HAVE_H4) && !defined(LASSO_NO_HOST_IFMA)
#include <immintrin.h>
#include <cstdlib>
#define LASSO_HOST_IFMA 1
#define F52_TARGET __attribute__((target("avx512f,avx512ifma"), always_inline)) inline
#define F52_FN __attribute__((target("avx512f,avx512ifma")))

namespace lasso {

__attribute__((target("arch=x86-64"))) inline bool field52_ok() {
  static const bool ok = [] { __builtin_cpu_init(); const char* v = getenv("LASSO_HOST_IFMA"); return !(v && v[0] == '0') && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma"); }();
  return ok;
}

struct F52Consts {
  uint64_t p[5], pinv, c_in[5], c_out[5];   // modulus, -p^-1 mod 2^52, 2^264 mod p, 2^256 mod p — 52-bit limbs
  static void split(const unsigned long long w[4], uint64_t l[5]) {
    const uint64_t M = (1ull << 52) - 1;
    l[0] = w[0] & M; l[1] = ((w[0] >> 52) | (w[1] << 12)) & M; l[2] = ((w[1] >> 40) | (w[2] << 24)) & M; l[3] = ((w[2] >> 28) | (w[3] << 36)) & M; l[4] = w[3] >> 16;
  }
  F52Consts() {
    split(g_fr_p64.p, p);
    uint64_t x = 1; for (int i = 0; i < 7; i++) x *= 2 - p[0] * x;   // p^-1 mod 2^64 (Newton; p is odd), then its negative mod 2^52
    pinv = (0ull - x) & ((1ull << 52) - 1);
    const H4 r256 = h4_from(fr_one()), r264 = h4_from(fr_from_u64(256));   // the ark-ff forms of 1 and 256 ARE the integers 2^256 mod p and 2^264 mod p
    split(r256.l, c_out); split(r264.l, c_in);
  }
};
inline const F52Consts& f52_consts() { static const F52Consts c; return c; }

struct F52 { __m512i l[5]; };

F52_TARGET F52 f52_bcast(const uint64_t l[5]) { F52 r; for (int j = 0; j < 5; j++) r.l[j] = _mm512_set1_epi64((long long)l[j]); return r; }
F52_TARGET F52 f52_zero() { F52 r; for (int j = 0; j < 5; j++) r.l[j] = _mm512_setzero_si512(); return r; }
// limbs may be any signed 64-bit values whose weighted sum is the (non-negative) value: afterwards l0..l3 < 2^52, l4 holds the rest (< 2^52 while the value is < 2^260)
F52_TARGET void f52_carry(F52& a) {
  const __m512i M = _mm512_set1_epi64((1ll << 52) - 1);
  for (int j = 0; j < 4; j++) { a.l[j + 1] = _mm512_add_epi64(a.l[j + 1], _mm512_srai_epi64(a.l[j], 52)); a.l[j] = _mm512_and_si512(a.l[j], M); }
}
// a (carried) -> a - p where a >= p (one step towards the canonical value; a < 2p gives a mod p)
F52_TARGET F52 f52_condsub(const F52& a, const F52& P) {
  F52 d; for (int j = 0; j < 5; j++) d.l[j] = _mm512_sub_epi64(a.l[j], P.l[j]);
  f52_carry(d);
  const __mmask8 neg = _mm512_cmplt_epi64_mask(d.l[4], _mm512_setzero_si512());   // borrowed: a < p
  F52 r; for (int j = 0; j < 5; j++) r.l[j] = _mm512_mask_blend_epi64(neg, d.l[j], a.l[j]);
  return r;
}
// LAZY forms: limb-wise, no carry, no reduction — the caller carries before a product (vpmadd52 reads the low 52 bits of a limb) and keeps the VALUE below 2^260
F52_TARGET F52 f52_addl(const F52& a, const F52& b) { F52 s; for (int j = 0; j < 5; j++) s.l[j] = _mm512_add_epi64(a.l[j], b.l[j]); return s; }
F52_TARGET F52 f52_subl(const F52& a, const F52& b, const F52& P) { F52 s; for (int j = 0; j < 5; j++) s.l[j] = _mm512_add_epi64(_mm512_sub_epi64(a.l[j], b.l[j]), P.l[j]); return s; }   // a - b + p > 0 for b <= p
// N independent Montgomery products side by side (R = 2^260; the word-serial form: five steps of accumulate a * b_i, m = t0 * (-p^-1) mod 2^52, accumulate m * p, drop the zero
// limb) — ONE product is a chain of ~14 dependent cycles per step, N of them keep the two IFMA pipes busy.  Operands carried, r = a * b * 2^-260 + (multiple of p) < p + a b / 2^260,
// carried, NOT reduced.
// (written out with named accumulators: as loops over t[n][j] the compiler kept the accumulators in memory)
#define F52_LO _mm512_madd52lo_epu64
#define F52_HI _mm512_madd52hi_epu64
#define F52_MUL_DECL(n) __m512i t##n##0 = Z, t##n##1 = Z, t##n##2 = Z, t##n##3 = Z, t##n##4 = Z, t##n##5 = Z;
#define F52_MUL_AB(n, i) { const __m512i bi = b[n].l[i]; \
    t##n##0 = F52_LO(t##n##0, a[n].l[0], bi); t##n##1 = F52_LO(t##n##1, a[n].l[1], bi); t##n##2 = F52_LO(t##n##2, a[n].l[2], bi); t##n##3 = F52_LO(t##n##3, a[n].l[3], bi); t##n##4 = F52_LO(t##n##4, a[n].l[4], bi); \
    t##n##1 = F52_HI(t##n##1, a[n].l[0], bi); t##n##2 = F52_HI(t##n##2, a[n].l[1], bi); t##n##3 = F52_HI(t##n##3, a[n].l[2], bi); t##n##4 = F52_HI(t##n##4, a[n].l[3], bi); t##n##5 = F52_HI(t##n##5, a[n].l[4], bi); }
#define F52_MUL_RED(n) { const __m512i m = F52_LO(Z, t##n##0, pinv);   /* low 52 bits of (t0 mod 2^52) * (-p^-1): t0 + m * p_0 = 0 mod 2^52 */ \
    t##n##0 = F52_LO(t##n##0, m, P.l[0]); t##n##1 = F52_LO(t##n##1, m, P.l[1]); t##n##2 = F52_LO(t##n##2, m, P.l[2]); t##n##3 = F52_LO(t##n##3, m, P.l[3]); t##n##4 = F52_LO(t##n##4, m, P.l[4]); \
    t##n##1 = F52_HI(t##n##1, m, P.l[0]); t##n##2 = F52_HI(t##n##2, m, P.l[1]); t##n##3 = F52_HI(t##n##3, m, P.l[2]); t##n##4 = F52_HI(t##n##4, m, P.l[3]); t##n##5 = F52_HI(t##n##5, m, P.l[4]); \
    t##n##0 = _mm512_add_epi64(t##n##1, _mm512_srli_epi64(t##n##0, 52)); t##n##1 = t##n##2; t##n##2 = t##n##3; t##n##3 = t##n##4; t##n##4 = t##n##5; t##n##5 = Z; }   /* / 2^52 */
#define F52_MUL_OUT(n) { r[n].l[0] = t##n##0; r[n].l[1] = t##n##1; r[n].l[2] = t##n##2; r[n].l[3] = t##n##3; r[n].l[4] = t##n##4; f52_carry(r[n]); }
template <int N> F52_TARGET void f52_mul_lazy(F52* r, const F52* a, const F52* b, const F52& P, const __m512i pinv);
template <> F52_TARGET void f52_mul_lazy<1>(F52* r, const F52* a, const F52* b, const F52& P, const __m512i pinv) {
  const __m512i Z = _mm512_setzero_si512();
  F52_MUL_DECL(0)
  F52_MUL_AB(0, 0) F52_MUL_RED(0) F52_MUL_AB(0, 1) F52_MUL_RED(0) F52_MUL_AB(0, 2) F52_MUL_RED(0) F52_MUL_AB(0, 3) F52_MUL_RED(0) F52_MUL_AB(0, 4) F52_MUL_RED(0)
  F52_MUL_OUT(0)
}
template <> F52_TARGET void f52_mul_lazy<3>(F52* r, const F52* a, const F52* b, const F52& P, const __m512i pinv) {
  const __m512i Z = _mm512_setzero_si512();
  F52_MUL_DECL(0) F52_MUL_DECL(1) F52_MUL_DECL(2)
#define F52_MUL_STEP3(i) F52_MUL_AB(0, i) F52_MUL_AB(1, i) F52_MUL_AB(2, i) F52_MUL_RED(0) F52_MUL_RED(1) F52_MUL_RED(2)
  F52_MUL_STEP3(0) F52_MUL_STEP3(1) F52_MUL_STEP3(2) F52_MUL_STEP3(3) F52_MUL_STEP3(4)
#undef F52_MUL_STEP3
  F52_MUL_OUT(0) F52_MUL_OUT(1) F52_MUL_OUT(2)
}
F52_TARGET F52 f52_mul(const F52& a, const F52& b, const F52& P, const __m512i pinv) { F52 r; f52_mul_lazy<1>(&r, &a, &b, P, pinv); return f52_condsub(r, P); }   // canonical operands -> canonical product

// ---- the rounds of sumcheck.rs:49-124 on k circuits' handed-over arrays (A_c, B_c of m elements each), eq weights C (m elements, the running factor folded in) and batching
// coefficients w_c != 0, folded into A'_c = w_c * A_c as in the scalar loop — and here A_c itself is not carried along: its final value, a claim (:126-133), is A'_c / w_c.
// Arrays lie as structure-of-limbs, `stride` elements apart, 8 elements of slack at the end; array elements are CANONICAL between calls, everything inside a call is lazy with
// the bounds written where they matter (k <= 16 circuits, m <= 256).
class HostRounds52 {
  size_t k, m, stride, narr;
  std::vector<uint64_t> buf;   // [limb][array * stride + i]
  size_t plane;                // elements per limb plane
  std::vector<H4> w_ark;
  uint64_t* L(int j) { return buf.data() + (size_t)j * plane; }
  // array order: A'_0.., B_0.., C
  size_t off_aw(size_t c) const { return c * stride; }
  size_t off_b(size_t c) const { return (k + c) * stride; }
  size_t off_c() const { return 2 * k * stride; }
  F52_TARGET F52 load(size_t at) { F52 r; for (int j = 0; j < 5; j++) r.l[j] = _mm512_loadu_si512((const void*)(L(j) + at)); return r; }
  F52_TARGET void store(size_t at, const F52& v) { for (int j = 0; j < 5; j++) _mm512_storeu_si512((void*)(L(j) + at), v.l[j]); }
  void put(size_t at, const H4& x) { uint64_t l[5]; F52Consts::split(x.l, l); for (int j = 0; j < 5; j++) L(j)[at] = l[j]; }
  static H4 pack(const uint64_t l[5]) { H4 r; r.l[0] = l[0] | (l[1] << 52); r.l[1] = (l[1] >> 12) | (l[2] << 40); r.l[2] = (l[2] >> 24) | (l[3] << 28); r.l[3] = (l[3] >> 36) | (l[4] << 16); return r; }
  F52_TARGET static H4 lane0(const F52& o) { alignas(64) uint64_t lanes[5][8]; for (int j = 0; j < 5; j++) _mm512_store_si512((void*)lanes[j], o.l[j]); const uint64_t l[5] = {lanes[0][0], lanes[1][0], lanes[2][0], lanes[3][0], lanes[4][0]}; return pack(l); }

 public:
  static bool fits(size_t k_, size_t m_) { return k_ >= 1 && k_ <= 16 && m_ >= 2 && m_ <= 256; }
  // a[c][i], b[c][i] (ark form, canonical), w[c] != 0, cw[i] = eq weight * running factor (ark form)
  F52_FN HostRounds52(const std::vector<std::vector<H4>>& a, const std::vector<std::vector<H4>>& b, const std::vector<H4>& w, const std::vector<H4>& cw) : w_ark(w) {
    k = a.size(); m = cw.size(); stride = (m + 7) & ~(size_t)7; if (stride < 8) stride = 8;
    narr = 2 * k + 1; plane = narr * stride + 8; buf.assign(5 * plane, 0);
    for (size_t c = 0; c < k; c++) for (size_t i = 0; i < m; i++) { put(off_aw(c) + i, a[c][i]); put(off_b(c) + i, b[c][i]); }
    for (size_t i = 0; i < m; i++) put(off_c() + i, cw[i]);
    const F52Consts& K = f52_consts(); const F52 P = f52_bcast(K.p), cin = f52_bcast(K.c_in); const __m512i pinv = _mm512_set1_epi64((long long)K.pinv);
    // B, C into the 2^260 domain (x 2^264 / 2^260); A' = A * (w * 2^264) / 2^260 = A w 2^260 directly.  Work items (chunk, factor), three products side by side
    std::vector<F52> wx(k);
    for (size_t c = 0; c < k; c++) { uint64_t wl[5]; F52Consts::split(w[c].l, wl); wx[c] = f52_mul(f52_mul(f52_bcast(wl), cin, P, pinv), cin, P, pinv); }   // w 2^264
    const size_t per = stride / 8, items = narr * per;
    for (size_t it = 0; it < items; it += 3) {
      F52 x[3], f[3], q[3]; size_t at[3];
      for (int u = 0; u < 3; u++) {
        const size_t id = it + u < items ? it + u : items - 1, arr = id / per;   // (the last group repeats its last item)
        at[u] = arr * stride + (id % per) * 8; x[u] = load(at[u]); f[u] = arr < k ? wx[arr] : cin;
      }
      f52_mul_lazy<3>(q, x, f, P, pinv);
      for (int u = 0; u < 3; u++) q[u] = f52_condsub(q[u], P);
      for (int u = 0; u < 3; u++) store(at[u], q[u]);
    }
  }
  // e(0), e(2), e(3) of the round over the live prefix of length 2h (sumcheck.rs:68-97), ark form
  F52_FN void sums(size_t h, H4& e0, H4& e2, H4& e3) {
    const F52Consts& K = f52_consts(); const F52 P = f52_bcast(K.p), cout = f52_bcast(K.c_out); const __m512i pinv = _mm512_set1_epi64((long long)K.pinv);
    F52 s[3] = {f52_zero(), f52_zero(), f52_zero()};
    for (size_t i0 = 0; i0 < h; i0 += 8) {
      F52 t[3] = {f52_zero(), f52_zero(), f52_zero()};
      for (size_t c = 0; c < k; c++) {
        const F52 al = load(off_aw(c) + i0), ah = load(off_aw(c) + i0 + h), bl = load(off_b(c) + i0), bh = load(off_b(c) + i0 + h);
        // x = 2, 3 by `prev + hi - lo` (:68-89): da in (0, 2p), a2 < 3p, a3 < 5p
        const F52 da = f52_subl(ah, al, P), db = f52_subl(bh, bl, P);
        F52 av[3], bv[3], q[3];
        av[0] = al; bv[0] = bl; av[1] = f52_addl(ah, da); bv[1] = f52_addl(bh, db); av[2] = f52_addl(av[1], da); bv[2] = f52_addl(bv[1], db);
        f52_carry(av[1]); f52_carry(bv[1]); f52_carry(av[2]); f52_carry(bv[2]);
        f52_mul_lazy<3>(q, av, bv, P, pinv);                     // < p + 25 p^2 / 2^260 < 1.3 p each
        for (int x = 0; x < 3; x++) t[x] = f52_addl(t[x], q[x]);   // < 1.3 k p <= 21 p
      }
      const F52 cl = load(off_c() + i0), ch = load(off_c() + i0 + h);
      const F52 dc = f52_subl(ch, cl, P);
      F52 cv[3], pr[3]; cv[0] = cl; cv[1] = f52_addl(ch, dc); cv[2] = f52_addl(cv[1], dc);
      f52_carry(cv[1]); f52_carry(cv[2]); f52_carry(t[0]); f52_carry(t[1]); f52_carry(t[2]);
      f52_mul_lazy<3>(pr, t, cv, P, pinv);                       // < p + 21 * 5 p^2 / 2^260 < 2.3 p
      const __mmask8 live = (h - i0 >= 8) ? (__mmask8)0xFF : (__mmask8)((1u << (h - i0)) - 1u);
      for (int x = 0; x < 3; x++) for (int j = 0; j < 5; j++) s[x].l[j] = _mm512_add_epi64(s[x].l[j], _mm512_maskz_mov_epi64(live, pr[x].l[j]));   // <= 16 chunks: < 37 p < 2^260
    }
    // back to ark form (x 2^256 / 2^260), canonical, then the eight lanes of each sum
    F52 co[3] = {cout, cout, cout}, o[3];
    f52_carry(s[0]); f52_carry(s[1]); f52_carry(s[2]);
    f52_mul_lazy<3>(o, s, co, P, pinv);                          // < p + 37 p^2 / 2^260 < 2 p
    H4* outs[3] = {&e0, &e2, &e3};
    for (int x = 0; x < 3; x++) {
      const F52 oc = f52_condsub(o[x], P);
      alignas(64) uint64_t lanes[5][8]; for (int j = 0; j < 5; j++) _mm512_store_si512((void*)lanes[j], oc.l[j]);
      H4 acc = h4_zero();
      for (int lane = 0; lane < 8; lane++) { const uint64_t l[5] = {lanes[0][lane], lanes[1][lane], lanes[2][lane], lanes[3][lane], lanes[4][lane]}; acc = h4_add(acc, pack(l)); }
      *outs[x] = acc;
    }
  }
  // every array: x[i] <- x[i] + r (x[i + h] - x[i]), i < h (sumcheck.rs:116-120); r in ark form
  F52_FN void bind(size_t h, const H4& r) {
    const F52Consts& K = f52_consts(); const F52 P = f52_bcast(K.p), cin = f52_bcast(K.c_in); const __m512i pinv = _mm512_set1_epi64((long long)K.pinv);
    uint64_t rl[5]; F52Consts::split(r.l, rl);
    const F52 rv = f52_mul(f52_bcast(rl), cin, P, pinv);
    // work items (array, chunk), three at a time
    const size_t chunks = (h + 7) / 8, items = narr * chunks;
    for (size_t it = 0; it < items; it += 3) {
      F52 lo[3], d[3], rr[3] = {rv, rv, rv}, q[3]; size_t at[3];
      for (int x = 0; x < 3; x++) {
        const size_t id = it + x < items ? it + x : items - 1;   // (the last group repeats its last item)
        at[x] = (id / chunks) * stride + (id % chunks) * 8;
        lo[x] = load(at[x]); d[x] = f52_subl(load(at[x] + h), lo[x], P); f52_carry(d[x]);   // in (0, 2p)
      }
      f52_mul_lazy<3>(q, rr, d, P, pinv);                                                    // < 1.03 p
      for (int x = 0; x < 3; x++) { F52 n = f52_addl(lo[x], q[x]); f52_carry(n); q[x] = f52_condsub(f52_condsub(n, P), P); }   // < 2.03 p -> canonical
      for (int x = 0; x < 3; x++) store(at[x], q[x]);   // lanes beyond h receive values nobody reads again (the old upper half is in registers, the next array starts at the next multiple of 8)
    }
  }
  // A_c[0] = A'_c[0] / w_c and B_c[0], ark form
  F52_FN void heads(std::vector<H4>& out) {
    const F52Consts& K = f52_consts(); const F52 P = f52_bcast(K.p), cout = f52_bcast(K.c_out); const __m512i pinv = _mm512_set1_epi64((long long)K.pinv);
    out.resize(2 * k);
    // 1 / w_c for all circuits with one inversion
    std::vector<H4> pre(k); H4 acc = h4_from(fr_one());
    for (size_t c = 0; c < k; c++) { pre[c] = acc; acc = h4_mul(acc, w_ark[c]); }
    H4 inv = h4_from(fr_inv_host(h4_to(acc)));
    for (size_t c = k; c-- > 0;) {
      const H4 wi = h4_mul(inv, pre[c]); inv = h4_mul(inv, w_ark[c]);
      out[c] = h4_mul(lane0(f52_mul(load(off_aw(c)), cout, P, pinv)), wi);
      out[k + c] = lane0(f52_mul(load(off_b(c)), cout, P, pinv));
    }
  }
};

}  // namespace lasso
#endif

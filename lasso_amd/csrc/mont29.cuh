// A 254-bit prime field in nine signed 29-bit limbs with Montgomery radix 2^261, for a GENERAL modulus (all nine limbs of p non-zero).
// fr29.cuh / fe29.cuh are the curve25519 instances, shaped around the sparse moduli 2^252 + c and 2^255 - 19; this header carries the same
// carry-free column arithmetic (81 back-to-back v_mad_i64_i32 per product, see fr29.cuh) to the BN254 fields (ark-bn254's Fr and Fq, the
// group BASELINE.json's configs[1] names), where the reduction is nine full rows (81 more multiply-adds) and the quotient estimates of the
// lazy reductions come from a reciprocal instead of a shift.
//
// value(a) = sum a.v[k] * 2^(29k), limbs signed, lazily reduced:
//   "reduced": limbs 0..7 in [0, 2^29), limb 8 small and signed     (outputs of mul / weak / unpack)
//   "loose":   |limb| <= 2^30
// m29_mul(a, b) = a*b / 2^261 (mod p); requires |a.v| <= 2^30, |b.v| <= 2^29.  |a*b| < X * 2^261  =>  result in (-X, p + X), reduced.
//   (column bound: nine products < 2^59 plus nine reduction terms < 2^58 plus a carry < 2^35: below 2^62.8.)
// M supplies: p(k) (29-bit limbs of p), PINV (-p^-1 mod 2^29), QC = floor(2^284 / p), and the limbs of ONE_S = 2^261 and K522 = 2^522 (mod p).
#pragma once
#include <stdint.h>

#ifndef LHD
#if defined(__HIPCC__)
#define LHD __host__ __device__ __forceinline__
#else
#define LHD inline
#endif
#endif

#define M29_MASK 0x1fffffff

template <class M> struct m29 { int32_t v[9]; };

template <class M> LHD m29<M> m29_zero() { m29<M> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = 0; return r; }
template <class M> LHD m29<M> m29_add(const m29<M>& a, const m29<M>& b) { m29<M> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
template <class M> LHD m29<M> m29_sub(const m29<M>& a, const m29<M>& b) { m29<M> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] - b.v[i]; return r; }
template <class M> LHD m29<M> m29_neg(const m29<M>& a) { m29<M> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = -a.v[i]; return r; }
// carry pass: any limbs with |.| < 2^31 -> reduced (value unchanged; limb 8 absorbs the top carry)
template <class M> LHD m29<M> m29_weak(const m29<M>& a) {
  m29<M> r; int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { int32_t x = a.v[i] + c; c = x >> 29; r.v[i] = x & M29_MASK; }
  r.v[8] = a.v[8] + c;
  return r;
}
// value * k for a small k >= 0 (k * 2^31 must fit 63 bits): reduced input -> reduced output
template <class M> LHD m29<M> m29_mul_small(const m29<M>& a, int32_t k) {
  m29<M> r; int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { int64_t x = (int64_t)a.v[i] * k + c; c = x >> 29; r.v[i] = (int32_t)x & M29_MASK; }
  r.v[8] = (int32_t)((int64_t)a.v[8] * k + c);
  return r;
}
// the nine Montgomery rows on a 17-column product: afterwards columns 9..16 hold (product + m p) / 2^261, not yet carried
template <class M> LHD void m29_rows(int64_t* h) {
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int32_t m = (int32_t)(((uint32_t)h[k] * (uint32_t)M::PINV) & M29_MASK);   // h[k] + m*p0 = 0 (mod 2^29)
#pragma unroll
    for (int j = 0; j < 9; j++) h[k + j] += (int64_t)m * M::p(j);
    h[k + 1] += h[k] >> 29;          // exact
  }
}
template <class M> LHD m29<M> m29_mul(const m29<M>& a, const m29<M>& b) {
  int64_t h[17];
#pragma unroll
  for (int k = 0; k < 17; k++) h[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++) h[i + j] += (int64_t)a.v[i] * b.v[j];
  m29_rows<M>(h);
  m29<M> r; int64_t c = 0;
#pragma unroll
  for (int k = 9; k < 17; k++) { int64_t x = h[k] + c; c = x >> 29; r.v[k - 9] = (int32_t)x & M29_MASK; }
  r.v[8] = (int32_t)c;
  return r;
}
// (a*b + sg*c*d) / 2^261 (mod p) with ONE reduction (nine rows instead of eighteen): a, b, c, d all REDUCED (|limb| < 2^29), sg = +1 / -1.
// Column bound: eighteen products < 2^58 plus nine reduction terms < 2^58 plus a carry: below 2^62.8.  |ab| + |cd| < X * 2^261  =>  result in (-X, p + X), reduced.
template <class M> LHD m29<M> m29_mul2(const m29<M>& a, const m29<M>& b, const m29<M>& c, const m29<M>& d, int32_t sg) {
  int64_t h[17];
#pragma unroll
  for (int k = 0; k < 17; k++) h[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++) h[i + j] += (int64_t)a.v[i] * b.v[j];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int32_t ci = sg * c.v[i];
#pragma unroll
    for (int j = 0; j < 9; j++) h[i + j] += (int64_t)ci * d.v[j];
  }
  m29_rows<M>(h);
  m29<M> r; int64_t cy = 0;
#pragma unroll
  for (int k = 9; k < 17; k++) { int64_t x = h[k] + cy; cy = x >> 29; r.v[k - 9] = (int32_t)x & M29_MASK; }
  r.v[8] = (int32_t)cy;
  return r;
}
// small signed 64-bit integer -> limbs (reduced; for products with a radix constant)
template <class M> LHD m29<M> m29_from_i64(int64_t x) {
  m29<M> r = m29_zero<M>();
  r.v[0] = (int32_t)(x & M29_MASK); r.v[1] = (int32_t)((x >> 29) & M29_MASK); r.v[2] = (int32_t)(x >> 58);
  return r;
}
template <class M> LHD m29<M> m29_const(const int32_t* c) { m29<M> r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = c[i]; return r; }

// Lazy reduction: any limbs with |.| < 2^31 and |value| < 2^258 -> the digits (limbs 0..7 in [0, 2^29), limb 8 in [0, 2^22)) of a
// representative in [0, p (1 + 2^-24)).  The quotient comes from the top 31 bits: U = floor(value / 2^227) up to one unit (the low seven limbs
// move it by < 2^-22), f = floor((U * QC - 2^32) / 2^57) with QC = floor(2^284 / p).  U * QC / 2^57 is within (-2^-25.3, +2^-26) of value / p;
// the bias 2^-25 makes it an under-estimate for either sign: f is floor(value / p) or one less, the latter only when value / p is within
// 2^-24 above an integer.
template <class M> LHD m29<M> m29_near(const m29<M>& a) {
  const int32_t U = a.v[8] * 32 + (a.v[7] >> 24);
  const int32_t f = (int32_t)(((int64_t)U * M::QC - ((int64_t)1 << 32)) >> 57);
  m29<M> r; int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { const int64_t x = (int64_t)a.v[k] - (int64_t)f * M::p(k) + c; if (k < 8) { r.v[k] = (int32_t)x & M29_MASK; c = x >> 29; } else r.v[8] = (int32_t)x; }
  return r;
}
// same input contract -> the canonical representative in [0, p)
template <class M> LHD m29<M> m29_canonical(const m29<M>& a) {
  const m29<M> r = m29_near<M>(a);
  m29<M> s; int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { const int64_t x = (int64_t)r.v[k] - M::p(k) + c; if (k < 8) { s.v[k] = (int32_t)x & M29_MASK; c = x >> 29; } else s.v[8] = (int32_t)x; }
  const bool keep_r = s.v[8] < 0;
#pragma unroll
  for (int k = 0; k < 9; k++) s.v[k] = keep_r ? r.v[k] : s.v[k];
  return s;
}

// ---- memory words (8 x u32, little endian, a 256-bit non-negative integer) <-> limbs
template <class M> LHD m29<M> m29_unpack_words(const uint32_t* w) {   // limbs of the same integer
  m29<M> r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int bit = 29 * k, i = bit >> 5, s = bit & 31;
    const uint64_t two = (uint64_t)w[i] | ((i + 1 < 8) ? ((uint64_t)w[i + 1] << 32) : 0);
    r.v[k] = (int32_t)((uint32_t)(two >> s) & M29_MASK);
  }
  return r;
}
template <class M> LHD m29<M> m29_unpack_words_shl5(const uint32_t* w) {   // limbs of (integer << 5); the integer is < 2^255, so limb 8 < 2^28
  m29<M> r;
  r.v[0] = (int32_t)((w[0] << 5) & M29_MASK);
#pragma unroll
  for (int k = 1; k < 9; k++) {
    const int bit = 29 * k - 5, i = bit >> 5, s = bit & 31;
    const uint64_t two = (uint64_t)w[i] | ((i + 1 < 8) ? ((uint64_t)w[i + 1] << 32) : 0);
    r.v[k] = (int32_t)((uint32_t)(two >> s) & M29_MASK);
  }
  return r;
}
template <class M> LHD void m29_pack_words(const m29<M>& a, uint32_t* out) {   // digits of a value in [0, 2^256) -> words
#pragma unroll
  for (int w = 0; w < 8; w++) {
    const int k0 = (32 * w) / 29, s = 32 * w - 29 * k0;
    uint64_t acc = (uint64_t)(uint32_t)a.v[k0] >> s;
    int have = 29 - s;
    if (k0 + 1 < 9) { acc |= (uint64_t)(uint32_t)a.v[k0 + 1] << have; have += 29; }
    if (have < 32 && k0 + 2 < 9) acc |= (uint64_t)(uint32_t)a.v[k0 + 2] << have;
    out[w] = (uint32_t)acc;
  }
}

// ---- sums of products without a reduction per product (see fr29.cuh: up to THREE products of operands with |limb| <= 2^29 between carries)
template <class M> struct m29_acc { int64_t h[17]; };
template <class M> LHD m29_acc<M> m29_acc_zero() { m29_acc<M> r;
#pragma unroll
  for (int k = 0; k < 17; k++) r.h[k] = 0; return r; }
template <class M> LHD void m29_mul_acc(m29_acc<M>& acc, const m29<M>& a, const m29<M>& b) {
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++) acc.h[i + j] += (int64_t)a.v[i] * b.v[j];
}
template <class M> LHD void m29_acc_carry(m29_acc<M>& acc) {
#pragma unroll
  for (int k = 0; k < 16; k++) { acc.h[k + 1] += acc.h[k] >> 29; acc.h[k] &= M29_MASK; }
}
// nine 64-bit columns (|col| < 2^62) -> a reduced value of the same residue, in (-2p, 2^261 + 2p): the part above 2^261 goes through one
// Montgomery product with 2^522 (K522).  Once per thread / block, never per element.
template <class M> LHD m29<M> m29_from_columns(const int64_t* col) {
  int64_t l[9]; int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { const int64_t x = col[k] + c; c = x >> 29; l[k] = x & M29_MASK; }
  m29<M> lo;
#pragma unroll
  for (int k = 0; k < 9; k++) lo.v[k] = (int32_t)l[k];
  const int32_t K522[9] = {M::K522_0, M::K522_1, M::K522_2, M::K522_3, M::K522_4, M::K522_5, M::K522_6, M::K522_7, M::K522_8};
  return m29_weak<M>(m29_add<M>(lo, m29_mul<M>(m29_from_i64<M>(c), m29_const<M>(K522))));
}
// the accumulated sum / 2^261 (mod p), reduced; valid after m29_acc_carry for sums of up to 2^20 products
template <class M> LHD m29<M> m29_acc_reduce(const m29_acc<M>& acc) {
  int64_t h[17];
#pragma unroll
  for (int k = 0; k < 17; k++) h[k] = acc.h[k];
  m29_rows<M>(h);
  int64_t col[9];
#pragma unroll
  for (int k = 0; k < 8; k++) col[k] = h[9 + k];
  col[8] = 0;
  return m29_from_columns<M>(col);
}
// Nine 64-bit column sums (|col| < 2^50), times 2^shift (shift <= 10) -> canonical limbs of the same residue: how a block / grid sum ends.
// value = L + H 2^261 with L the low nine digits and |H| < 2^32;  L -> L 2^261 / 2^261 (a product with ONE_S brings it below 2p),
// H 2^261 = H 2^522 / 2^261.
template <class M> LHD m29<M> m29_reduce_columns(const int64_t* col, int shift) {
  int64_t l[9]; int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { const int64_t x = col[k] * ((int64_t)1 << shift) + c; c = x >> 29; l[k] = x & M29_MASK; }
  m29<M> lo;
#pragma unroll
  for (int k = 0; k < 9; k++) lo.v[k] = (int32_t)l[k];
  const int32_t ONE_S[9] = {M::ONE_S_0, M::ONE_S_1, M::ONE_S_2, M::ONE_S_3, M::ONE_S_4, M::ONE_S_5, M::ONE_S_6, M::ONE_S_7, M::ONE_S_8};
  const int32_t K522[9] = {M::K522_0, M::K522_1, M::K522_2, M::K522_3, M::K522_4, M::K522_5, M::K522_6, M::K522_7, M::K522_8};
  const m29<M> a = m29_mul<M>(lo, m29_const<M>(ONE_S)), b = m29_mul<M>(m29_from_i64<M>(c), m29_const<M>(K522));
  return m29_canonical<M>(m29_add<M>(a, b));
}

// Fr (curve25519 scalar field, p = 2^252 + c) in nine signed 29-bit limbs — the form the polynomial kernels COMPUTE in.
//
// Memory keeps ark-ff's layout (fr_t: 8 x u32 = 4 x u64, x*2^256 mod p, canonical) because that is the ABI; a kernel unpacks on load,
// works in fr29, and canonicalises + packs on store.  Why another form: on gfx950 v_mad_i64_i32 issues every ~5 cycles and adds into a
// 64-bit column in place, so with 29-bit limbs a schoolbook product is 81 back-to-back multiply-adds with NO carry handling in between
// (nine 2^59 products fit a signed 64-bit column).  The 8 x 32-bit CIOS form (fr.cuh) spends 500 of its 600 instructions per product moving
// carries (v_mov / v_lshl_add_u64); it remains the host form and the reference the tests compare against.
//
// value(a) = sum a.v[k] * 2^(29k), limbs signed, lazily reduced: add/sub are limb-wise with no carries and no modular correction.
//   "reduced": limbs 0..7 in [0, 2^29), limb 8 small and signed  (outputs of fr29_mul, fr29_weak, the unpack functions)
//   "loose":   |limb| <= 2^30                                      (one add/sub of reduced values)
// fr29_mul(a, b) = a*b / 2^261 (mod p), Montgomery with radix 2^29 over the sparse modulus (limbs 5..7 of p are zero, limb 8 = 2^20).
//   requires |a.v[i]| <= 2^30, |b.v[j]| <= 2^29 (a loose, b reduced).  |a*b| < X * 2^261  =>  result in (-X, p + X), reduced.
//
// The radix is 2^261, memory is 2^256: a product of two "u-form" values (x*2^256) comes out 2^5 short.  Every kernel therefore loads
// ONE operand of each product in "s-form" (x*2^261 = the same bits shifted left by 5, free at unpack time) or corrects a whole sum
// once at the end with a constant (FR29_K5 / FR29_K10).  mul(u, s) = u-form; mul(s, s) = s-form; mul(u, u) = u-form / 2^5.
#pragma once
#ifdef LASSO_BN254
#include "bn254_fr29.cuh"   // the same interface over ark-bn254's Fr
#else
#include <stdint.h>
#include "fr.cuh"

struct fr29 { int32_t v[9]; };
#define FR29_MASK 0x1fffffff
#define FR29_PINV 307527195u   // -p^-1 mod 2^29
#define FR29_P0 485872621
#define FR29_P1 9640146
#define FR29_P2 501691798
#define FR29_P3 502512965
#define FR29_P4 333
// limbs 5..7 of p are zero; limb 8 = 2^20

LHD fr29 fr29_zero() { fr29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = 0; return r; }
LHD fr29 fr29_from_limbs(int32_t a0, int32_t a1, int32_t a2, int32_t a3, int32_t a4, int32_t a5, int32_t a6, int32_t a7, int32_t a8) {
  fr29 r; r.v[0] = a0; r.v[1] = a1; r.v[2] = a2; r.v[3] = a3; r.v[4] = a4; r.v[5] = a5; r.v[6] = a6; r.v[7] = a7; r.v[8] = a8; return r;
}
// 2^261 mod p: fr29_mul(a, ONE_S) = a (mod p) with the magnitude brought back to (-X, p + X)
LHD fr29 fr29_one_s() { return fr29_from_limbs(290322925, 442594051, 259787148, 377041255, 536700270, 536870911, 536870911, 536870911, 1048575); }
// 2^266 mod p: corrects a sum of mul(mul(u, s), u)-style terms that came out 2^5 short
LHD fr29 fr29_k5() { return fr29_from_limbs(133862381, 442392295, 276935791, 245514615, 531400038, 536870911, 536870911, 536870911, 1048575); }
// 2^271 mod p: corrects a sum of mul(mul(u, u), u) terms (2^10 short)
LHD fr29 fr29_k10() { return fr29_from_limbs(495834093, 435936093, 288821455, 331629432, 361792606, 536870911, 536870911, 536870911, 1048575); }
// 2^517 mod p: fr29_mul(integer x < 2^64 as limbs, R2S) = x * 2^256 = u-form of x
LHD fr29 fr29_r2s() { return fr29_from_limbs(147395749, 34354560, 457688582, 356494647, 483104506, 488734555, 518485561, 233882216, 206883); }

// the integer 2^10 as limbs: fr29_mul(fr29_mul(u, u), INT_FROM_UU) = the canonical integer x*y (mod p) of a product of two u-form values
LHD fr29 fr29_int_from_uu() { fr29 r = fr29_zero(); r.v[0] = 1 << 10; return r; }

LHD fr29 fr29_add(const fr29& a, const fr29& b) { fr29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
LHD fr29 fr29_sub(const fr29& a, const fr29& b) { fr29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] - b.v[i]; return r; }

// carry pass: any limbs with |.| < 2^31 -> reduced (value unchanged; limb 8 absorbs the top carry)
LHD fr29 fr29_weak(const fr29& a) {
  fr29 r; int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { int32_t x = a.v[i] + c; c = x >> 29; r.v[i] = x & FR29_MASK; }
  r.v[8] = a.v[8] + c;
  return r;
}

// memory (canonical x*2^256, 8 x u32) -> limbs of the same integer ("u-form")
LHD fr29 fr29_unpack_u(const fr_t& x) {
  fr29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int bit = 29 * k, w = bit >> 5, s = bit & 31;
    uint64_t two = (uint64_t)x.v[w] | ((w + 1 < 8) ? ((uint64_t)x.v[w + 1] << 32) : 0);
    r.v[k] = (int32_t)((uint32_t)(two >> s) & FR29_MASK);
  }
  return r;
}
// memory -> limbs of (integer << 5) = x*2^261 ("s-form"); the integer is < 2^253, so limb 8 < 2^26
LHD fr29 fr29_unpack_s(const fr_t& x) {
  fr29 r;
  r.v[0] = (int32_t)((x.v[0] << 5) & FR29_MASK);
#pragma unroll
  for (int k = 1; k < 9; k++) {
    const int bit = 29 * k - 5, w = bit >> 5, s = bit & 31;
    uint64_t two = (uint64_t)x.v[w] | ((w + 1 < 8) ? ((uint64_t)x.v[w + 1] << 32) : 0);
    r.v[k] = (int32_t)((uint32_t)(two >> s) & FR29_MASK);
  }
  return r;
}
// small non-negative integer -> limbs (for fr29_mul(x, R2S))
LHD fr29 fr29_from_u64_int(uint64_t x) {
  fr29 r = fr29_zero();
  r.v[0] = (int32_t)(x & FR29_MASK); r.v[1] = (int32_t)((x >> 29) & FR29_MASK); r.v[2] = (int32_t)(x >> 58);
  return r;
}

// a loose, b reduced -> a*b/2^261 mod p, reduced.  81 + 45 multiply-adds.
LHD fr29 fr29_mul(const fr29& a, const fr29& b) {
  int64_t h[17];
#pragma unroll
  for (int k = 0; k < 17; k++) h[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++) h[i + j] += (int64_t)a.v[i] * b.v[j];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int32_t m = (int32_t)(((uint32_t)h[k] * FR29_PINV) & FR29_MASK);   // h[k] + m*p0 = 0 (mod 2^29)
    h[k] += (int64_t)m * FR29_P0;
    h[k + 1] += (int64_t)m * FR29_P1;
    h[k + 2] += (int64_t)m * FR29_P2;
    h[k + 3] += (int64_t)m * FR29_P3;
    h[k + 4] += (int64_t)m * FR29_P4;
    h[k + 8] += (int64_t)m << 20;
    h[k + 1] += h[k] >> 29;          // exact
  }
  fr29 r; int64_t c = 0;
#pragma unroll
  for (int k = 9; k < 17; k++) { int64_t x = h[k] + c; c = x >> 29; r.v[k - 9] = (int32_t)x & FR29_MASK; }
  r.v[8] = (int32_t)c;
  return r;
}

// ---- sums of products without a reduction per product.  acc is the 17-column double-width value (29-bit columns, signed 64-bit);
// fr29_mul_acc adds a*b into it with the same 81 multiply-adds fr29_mul starts with and nothing else.  One product adds < 9 * 2^58 to a column
// when |a.v|, |b.v| <= 2^29, so up to THREE products may be added between two fr29_acc_carry passes (which bring columns 0..15 back to
// [0, 2^29) and let column 16 absorb the carries).  fr29_acc_reduce finishes with the Montgomery reduction: the sum / 2^261 (mod p), reduced,
// correct for sums of up to 2^20 products.
LHD fr29 fr29_from_columns(const int64_t* col);
struct fr29_acc { int64_t h[17]; };
LHD fr29_acc fr29_acc_zero() { fr29_acc r;
#pragma unroll
  for (int k = 0; k < 17; k++) r.h[k] = 0; return r; }
LHD void fr29_mul_acc(fr29_acc& acc, const fr29& a, const fr29& b) {
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++) acc.h[i + j] += (int64_t)a.v[i] * b.v[j];
}
LHD void fr29_acc_carry(fr29_acc& acc) {
#pragma unroll
  for (int k = 0; k < 16; k++) { acc.h[k + 1] += acc.h[k] >> 29; acc.h[k] &= FR29_MASK; }
}
LHD fr29 fr29_acc_reduce(const fr29_acc& acc) {
  int64_t h[17];
#pragma unroll
  for (int k = 0; k < 17; k++) h[k] = acc.h[k];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int32_t m = (int32_t)(((uint32_t)h[k] * FR29_PINV) & FR29_MASK);
    h[k] += (int64_t)m * FR29_P0;
    h[k + 1] += (int64_t)m * FR29_P1;
    h[k + 2] += (int64_t)m * FR29_P2;
    h[k + 3] += (int64_t)m * FR29_P3;
    h[k + 4] += (int64_t)m * FR29_P4;
    h[k + 8] += (int64_t)m << 20;
    h[k + 1] += h[k] >> 29;          // exact
  }
  // columns 9..16 hold the quotient as eight 64-bit columns (column 16 carries the accumulated magnitude): fold through 2^261 = ONE_S
  int64_t col[9];
#pragma unroll
  for (int k = 0; k < 8; k++) col[k] = h[9 + k];
  col[8] = 0;
  return fr29_from_columns(col);
}

// Lazily reduced memory form for arrays that only kernels read (the bound arrays of a sumcheck between two rounds): any limbs with
// |.| < 2^31 and |value| < 2^255 -> digits (limbs in [0, 2^29), limb 8 <= 2^22) of SOME representative in (0, 2^254 + 2^130) of the same
// residue.  One fused pass: floor(value / 2^252) is estimated from the two top limbs (off by at most 1 either way: the carries of the lower
// limbs it ignores), and with f = estimate - 2 the value - f p lies in [2^252, 4 * 2^252) up to |f| c (c = p - 2^252 ~ 2^125).  48 instructions
// against 115 for fr29_canonical; fr29_pack / fr29_unpack_u carry such a value through memory unchanged, and every reader (fr29_mul
// operands, fr29_canonical) accepts it.
LHD fr29 fr29_semi(const fr29& a) {
  const int32_t P[9] = {FR29_P0, FR29_P1, FR29_P2, FR29_P3, FR29_P4, 0, 0, 0, 1 << 20};
  const int32_t f = ((a.v[8] + (a.v[7] >> 29)) >> 20) - 2;
  fr29 r; int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { int64_t x = (int64_t)a.v[k] - (int64_t)f * P[k] + c; if (k < 8) { r.v[k] = (int32_t)x & FR29_MASK; c = x >> 29; } else r.v[8] = (int32_t)x; }
  return r;
}

// any limbs with |.| < 2^31 and |value| < 2^255 -> the canonical representative in [0, p), limbs in [0, 2^29)
// (f below is floor(value / 2^252), |f| <= 8: the remainder r stays within 8c of [0, 2^252), which the second stage absorbs)
LHD fr29 fr29_canonical(const fr29& a) {
  const int32_t P[9] = {FR29_P0, FR29_P1, FR29_P2, FR29_P3, FR29_P4, 0, 0, 0, 1 << 20};
  fr29 w = fr29_weak(a);
  // r = value - f*p with f = floor(value / 2^252): r in (-3c, 2^252 + 3c)
  const int32_t f = w.v[8] >> 20;
  fr29 r; int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { int64_t x = (int64_t)w.v[k] - (int64_t)f * P[k] + c; if (k < 8) { r.v[k] = (int32_t)x & FR29_MASK; c = x >> 29; } else r.v[8] = (int32_t)x; }
  // g = -1: r < 0, r + p is canonical.  g = 0: canonical.  g = 1: r in [2^252, 2^252 + 3c): r - p if that is non-negative, else r.
  const int32_t g = r.v[8] >> 20;
  fr29 s; c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { int64_t x = (int64_t)r.v[k] - (int64_t)g * P[k] + c; if (k < 8) { s.v[k] = (int32_t)x & FR29_MASK; c = x >> 29; } else s.v[8] = (int32_t)x; }
  const bool keep_r = s.v[8] < 0;
#pragma unroll
  for (int k = 0; k < 9; k++) s.v[k] = keep_r ? r.v[k] : s.v[k];
  return s;
}
// Nine 64-bit column sums, times 2^shift (shift <= 10), -> the canonical limbs of the same residue.  This is how a block / grid sum ends:
// the radix corrections the kernels used to apply as one more Montgomery product with 2^261 (ONE_S), 2^266 (K5) or 2^271 (K10) are the
// shifts 0, 5, 10 of the column values, and the reduction is one exact quotient estimate instead of a product: with l_8 the top column after a
// carry pass (everything above 2^232), f = l_8 >> 20 is floor(value / 2^252) exactly, and value - (f - 1) p lies in (0, 2^253 + 2^152).
// |col[k]| < 2^50 before the shift (sums of up to 2^20 reduced limbs).  ~220 instructions against ~460 for from_columns + product + canonical.
LHD fr29 fr29_reduce_columns(const int64_t* col, int shift) {
  const int32_t P[9] = {FR29_P0, FR29_P1, FR29_P2, FR29_P3, FR29_P4, 0, 0, 0, 1 << 20};
  int64_t l[9]; int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { const int64_t x = col[k] * ((int64_t)1 << shift) + c; if (k < 8) { l[k] = x & FR29_MASK; c = x >> 29; } else l[8] = x; }
  const int32_t f = (int32_t)(l[8] >> 20) - 1;   // |l_8| < 2^51: fits
  fr29 r; c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { const int64_t x = l[k] - (int64_t)f * P[k] + c; if (k < 8) { r.v[k] = (int32_t)x & FR29_MASK; c = x >> 29; } else r.v[8] = (int32_t)x; }
  return fr29_canonical(r);
}
// canonical limbs -> memory words
LHD fr_t fr29_pack(const fr29& a) {
  fr_t r;
#pragma unroll
  for (int w = 0; w < 8; w++) {
    // word w = bits [32w, 32w+32): limb k0 = floor(32w/29) from bit offset s, plus the next limb(s)
    const int k0 = (32 * w) / 29, s = 32 * w - 29 * k0;
    uint64_t acc = (uint64_t)(uint32_t)a.v[k0] >> s;
    int have = 29 - s;
    if (k0 + 1 < 9) { acc |= (uint64_t)(uint32_t)a.v[k0 + 1] << have; have += 29; }
    if (have < 32 && k0 + 2 < 9) acc |= (uint64_t)(uint32_t)a.v[k0 + 2] << have;
    r.v[w] = (uint32_t)acc;
  }
  return r;
}
// u-form value with |value| < 4p -> memory
LHD fr_t fr29_store(const fr29& a) { return fr29_pack(fr29_canonical(a)); }

// nine 64-bit column sums (e.g. of up to 2^20 reduced values) -> reduced fr29 of the same value mod p, with |value| < 2^262 + 2^29 p
LHD fr29 fr29_from_columns(const int64_t* col) {
  const int32_t ONE_S[9] = {290322925, 442594051, 259787148, 377041255, 536700270, 536870911, 536870911, 536870911, 1048575};
  int64_t l[9]; int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { int64_t x = col[k] + c; c = x >> 29; l[k] = x & FR29_MASK; }
  // value = l + c * 2^261 and 2^261 = ONE_S (mod p); |c| < 2^35 in any use here, c * ONE_S[k] < 2^64
  fr29 r; int64_t d = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { int64_t x = l[k] + c * ONE_S[k] + d; if (k < 8) { r.v[k] = (int32_t)x & FR29_MASK; d = x >> 29; } else r.v[8] = (int32_t)x; }
  return r;
}
#endif  // LASSO_BN254

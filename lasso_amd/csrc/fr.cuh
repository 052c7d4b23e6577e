// Fr = curve25519 scalar field, p = 2^252 + 27742317777372353535851937790883648493, as 8 x u32
// Montgomery limbs (R = 2^256).  The byte layout equals ark-ff's `Fp256<MontBackend<_,4>>`
// (4 x u64 little endian), so device buffers are bit-for-bit what the Rust host would hand over
// (SURVEY.md §8b "Data representation at the ABI").
//
// gfx950 notes: every limb product is written as u64 = u32*u32 + u32 so hipcc emits v_mad_u64_u32;
// the modulus limbs 4..6 are zero and limb 7 is 2^28, which the unrolled reduction folds into
// carries and a shift (4 mads + 1 mul per reduction row instead of 8).
// This header is __host__ __device__: the host prover uses the same arithmetic for its O(log n) tails.
#pragma once
#ifdef LASSO_BN254
#include "bn254_fr.cuh"   // the same interface over ark-bn254's Fr
#else
#include <stdint.h>

#if defined(__HIPCC__)
#define LHD __host__ __device__ __forceinline__
#else
#define LHD inline
#endif

struct alignas(16) fr_t {
  uint32_t v[8];
};

#define FR_P0 0x5cf5d3edu
#define FR_P1 0x5812631au
#define FR_P2 0xa2f79cd6u
#define FR_P3 0x14def9deu
#define FR_P7 0x10000000u
#define FR_INV32 0x12547e1bu  // -p^{-1} mod 2^32

LHD fr_t fr_zero() { fr_t r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
LHD fr_t fr_one() {  // R mod p
  fr_t r; r.v[0] = 0x8d98951du; r.v[1] = 0xd6ec3174u; r.v[2] = 0x737dcf70u; r.v[3] = 0xc6ef5bf4u;
  r.v[4] = 0xfffffffeu; r.v[5] = 0xffffffffu; r.v[6] = 0xffffffffu; r.v[7] = 0x0fffffffu; return r;
}
LHD fr_t fr_r2() {  // R^2 mod p
  fr_t r; r.v[0] = 0x449c0f01u; r.v[1] = 0xa40611e3u; r.v[2] = 0x68859347u; r.v[3] = 0xd00e1ba7u;
  r.v[4] = 0x17f5be65u; r.v[5] = 0xceec73d2u; r.v[6] = 0x7c309a3du; r.v[7] = 0x0399411bu; return r;
}
LHD uint32_t fr_p_limb(int i) { return i == 0 ? FR_P0 : i == 1 ? FR_P1 : i == 2 ? FR_P2 : i == 3 ? FR_P3 : i == 7 ? FR_P7 : 0u; }

LHD bool fr_is_zero(const fr_t& a) { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= a.v[i]; return o == 0; }
LHD bool fr_eq(const fr_t& a, const fr_t& b) { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i]; return o == 0; }

// a >= p ?
LHD bool fr_geq_p(const uint32_t* a) {
  // compare from the top; limbs 4..6 of p are zero
  if (a[7] != FR_P7) return a[7] > FR_P7;
  if (a[6] | a[5] | a[4]) return true;
  if (a[3] != FR_P3) return a[3] > FR_P3;
  if (a[2] != FR_P2) return a[2] > FR_P2;
  if (a[1] != FR_P1) return a[1] > FR_P1;
  return a[0] >= FR_P0;
}
// r = a - p if a >= p (a < 2p), branch-free
LHD void fr_cond_sub_p(uint32_t* a) {
  uint32_t t[8]; uint64_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a[i] - fr_p_limb(i) - bw; t[i] = (uint32_t)d; bw = (d >> 63); }
  // bw == 1  <=>  a < p  => keep a
  uint32_t keep = (uint32_t)0 - (uint32_t)bw;
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = (a[i] & keep) | (t[i] & ~keep);
}

LHD fr_t fr_add(const fr_t& a, const fr_t& b) {
  fr_t r; uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
  // a,b < p < 2^253 so no carry out of 256 bits
  fr_cond_sub_p(r.v);
  return r;
}
LHD fr_t fr_sub(const fr_t& a, const fr_t& b) {
  fr_t r; uint64_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.v[i] - b.v[i] - bw; r.v[i] = (uint32_t)d; bw = d >> 63; }
  uint32_t m = (uint32_t)0 - (uint32_t)bw;  // add p back when borrowed
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (uint64_t)r.v[i] + (fr_p_limb(i) & m); r.v[i] = (uint32_t)c; c >>= 32; }
  return r;
}
LHD fr_t fr_neg(const fr_t& a) { return fr_sub(fr_zero(), a); }
LHD fr_t fr_dbl(const fr_t& a) { return fr_add(a, a); }

// Montgomery product a*b*R^-1 mod p.
#if !defined(__HIPCC__) && defined(__SIZEOF_INT128__) && !defined(LASSO_HOST_LIMBS32)   /* LASSO_HOST_LIMBS32: tests force the device form on the host */
// Host build (g++, the O(log n) tails of the prover): CIOS over 64-bit limbs with 128-bit products — same function, ~6x faster on x86-64
// than the 32-bit form.  fr_t's bytes are the 4 x u64 little-endian limbs on a little-endian host.
inline fr_t fr_mul(const fr_t& a, const fr_t& b) {
  typedef unsigned __int128 u128;
  const uint64_t P0 = 0x5812631a5cf5d3edull, P1 = 0x14def9dea2f79cd6ull, P3 = 0x1000000000000000ull, INV = 0xd2b51da312547e1bull;  // -p^-1 mod 2^64
  uint64_t x[4], y[4]; __builtin_memcpy(x, a.v, 32); __builtin_memcpy(y, b.v, 32);
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#define FR64_ROW(yi)                                                                                   \
  {                                                                                                    \
    u128 c = (u128)x[0] * (yi) + t0; t0 = (uint64_t)c; c >>= 64;                                        \
    c += (u128)x[1] * (yi) + t1; t1 = (uint64_t)c; c >>= 64;                                            \
    c += (u128)x[2] * (yi) + t2; t2 = (uint64_t)c; c >>= 64;                                            \
    c += (u128)x[3] * (yi) + t3; t3 = (uint64_t)c; c >>= 64;                                            \
    c += t4; t4 = (uint64_t)c; const uint64_t t5 = (uint64_t)(c >> 64);                                 \
    const uint64_t m = t0 * INV;                                                                       \
    c = (u128)m * P0 + t0; c >>= 64;                                                                    \
    c += (u128)m * P1 + t1; t0 = (uint64_t)c; c >>= 64;                                                 \
    c += t2; t1 = (uint64_t)c; c >>= 64;                                                                \
    c += (u128)m * P3 + t3; t2 = (uint64_t)c; c >>= 64;                                                 \
    c += t4; t3 = (uint64_t)c; c >>= 64;                                                                \
    t4 = t5 + (uint64_t)c;                                                                              \
  }
  FR64_ROW(y[0]) FR64_ROW(y[1]) FR64_ROW(y[2]) FR64_ROW(y[3])
#undef FR64_ROW
  // result < 2p < 2^254: conditional subtraction of p on 64-bit limbs
  u128 d = (u128)t0 - P0; const uint64_t s0 = (uint64_t)d; uint64_t bw = (uint64_t)(d >> 64) & 1;
  d = (u128)t1 - P1 - bw; const uint64_t s1 = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1;
  d = (u128)t2 - bw; const uint64_t s2 = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1;
  d = (u128)t3 - P3 - bw; const uint64_t s3 = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1;
  const uint64_t keep = (uint64_t)0 - bw;   // borrow: the value was < p
  uint64_t r[4] = {(t0 & keep) | (s0 & ~keep), (t1 & keep) | (s1 & ~keep), (t2 & keep) | (s2 & ~keep), (t3 & keep) | (s3 & ~keep)};
  fr_t o; __builtin_memcpy(o.v, r, 32);
  return o;
}
#else
// CIOS over 32-bit limbs (device form; also the host form inside hipcc translation units).
LHD fr_t fr_mul(const fr_t& a, const fr_t& b) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
    const uint32_t bi = b.v[i];
#pragma unroll
    for (int j = 0; j < 8; j++) { c += (uint64_t)a.v[j] * bi + t[j]; t[j] = (uint32_t)c; c >>= 32; }
    c += t[8]; t[8] = (uint32_t)c; t[9] = (uint32_t)(c >> 32);
    const uint32_t m = t[0] * FR_INV32;
    c = (uint64_t)m * FR_P0 + t[0]; c >>= 32;
    c += (uint64_t)m * FR_P1 + t[1]; t[0] = (uint32_t)c; c >>= 32;
    c += (uint64_t)m * FR_P2 + t[2]; t[1] = (uint32_t)c; c >>= 32;
    c += (uint64_t)m * FR_P3 + t[3]; t[2] = (uint32_t)c; c >>= 32;
    c += t[4]; t[3] = (uint32_t)c; c >>= 32;
    c += t[5]; t[4] = (uint32_t)c; c >>= 32;
    c += t[6]; t[5] = (uint32_t)c; c >>= 32;
    c += ((uint64_t)m << 28) + t[7]; t[6] = (uint32_t)c; c >>= 32;
    c += t[8]; t[7] = (uint32_t)c; c >>= 32;
    t[8] = t[9] + (uint32_t)c;
  }
  // result < 2p and p < 2^253, so t[8] == 0
  fr_t r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  fr_cond_sub_p(r.v);
  return r;
}
#endif
LHD fr_t fr_sqr(const fr_t& a) { return fr_mul(a, a); }

// small integer -> Montgomery form (dense_mlpoly.rs:263-269 `F::from(Z[i] as u64)`)
LHD fr_t fr_from_u64(uint64_t x) {
  fr_t t = fr_zero(); t.v[0] = (uint32_t)x; t.v[1] = (uint32_t)(x >> 32);
  return fr_mul(t, fr_r2());
}
// Montgomery -> canonical integer limbs
LHD fr_t fr_to_canonical(const fr_t& a) { fr_t o = fr_zero(); o.v[0] = 1; return fr_mul(a, o); }
// canonical integer (< 2^256) -> Montgomery
LHD fr_t fr_from_canonical(const fr_t& c) {
  fr_t t = c;
  // c may be >= p (up to 2^256-1 < 16p): subtract while needed
  for (int k = 0; k < 16 && fr_geq_p(t.v); k++) {
    uint64_t bw = 0;
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)t.v[i] - fr_p_limb(i) - bw; t.v[i] = (uint32_t)d; bw = d >> 63; }
  }
  return fr_mul(t, fr_r2());
}
// a^e, e canonical 8-limb exponent (host-side tails only)
LHD fr_t fr_pow(const fr_t& a, const uint32_t* e) {
  fr_t r = fr_one();
  for (int i = 255; i >= 0; i--) { r = fr_sqr(r); if ((e[i / 32] >> (i % 32)) & 1) r = fr_mul(r, a); }
  return r;
}
LHD fr_t fr_inv(const fr_t& a) {  // Fermat; inverse(0) = 0
  uint32_t e[8] = {FR_P0 - 2u, FR_P1, FR_P2, FR_P3, 0, 0, 0, FR_P7};
  return fr_pow(a, e);
}
// number of significant bits of the canonical value
LHD int fr_canonical_bits(const fr_t& c) {
  for (int i = 7; i >= 0; i--) if (c.v[i]) { uint32_t x = c.v[i]; int n = 0; while (x) { n++; x >>= 1; } return 32 * i + n; }
  return 0;
}
#endif  // LASSO_BN254

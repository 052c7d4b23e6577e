// BN254 build (-DLASSO_BN254) of fr.cuh: Fr = the order of ark-bn254's G1,
// p = 21888242871839275222246405745257275088548364400416034343698204186575808495617 (254 bits), as 8 x u32 Montgomery limbs (R = 2^256) —
// byte-identical to ark-ff's `Fp256<MontBackend<FrConfig, 4>>`.  Same interface as the curve25519 header; the modulus has no structure to
// exploit, so the reduction rows are full.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LHD __host__ __device__ __forceinline__
#else
#define LHD inline
#endif

struct alignas(16) fr_t {
  uint32_t v[8];
};

#define FR_INV32 0xefffffffu  // -p^{-1} mod 2^32
LHD uint32_t fr_p_limb(int i) {
  const uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  return P[i];
}
LHD fr_t fr_zero() { fr_t r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
LHD fr_t fr_one() {  // R mod p
  fr_t r; r.v[0] = 0x4ffffffbu; r.v[1] = 0xac96341cu; r.v[2] = 0x9f60cd29u; r.v[3] = 0x36fc7695u;
  r.v[4] = 0x7879462eu; r.v[5] = 0x666ea36fu; r.v[6] = 0x9a07df2fu; r.v[7] = 0x0e0a77c1u; return r;
}
LHD fr_t fr_r2() {  // R^2 mod p
  fr_t r; r.v[0] = 0xae216da7u; r.v[1] = 0x1bb8e645u; r.v[2] = 0xe35c59e3u; r.v[3] = 0x53fe3ab1u;
  r.v[4] = 0x53bb8085u; r.v[5] = 0x8c49833du; r.v[6] = 0x7f4e44a5u; r.v[7] = 0x0216d0b1u; return r;
}
LHD bool fr_is_zero(const fr_t& a) { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= a.v[i]; return o == 0; }
LHD bool fr_eq(const fr_t& a, const fr_t& b) { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i]; return o == 0; }

LHD bool fr_geq_p(const uint32_t* a) {
  for (int i = 7; i >= 0; i--) { const uint32_t pi = fr_p_limb(i); if (a[i] != pi) return a[i] > pi; }
  return true;
}
// r = a - p if a >= p (a < 2p), branch-free
LHD void fr_cond_sub_p(uint32_t* a) {
  uint32_t t[8]; uint64_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a[i] - fr_p_limb(i) - bw; t[i] = (uint32_t)d; bw = (d >> 63); }
  uint32_t keep = (uint32_t)0 - (uint32_t)bw;
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = (a[i] & keep) | (t[i] & ~keep);
}
LHD fr_t fr_add(const fr_t& a, const fr_t& b) {
  fr_t r; uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
  fr_cond_sub_p(r.v);   // a, b < p < 2^254: no carry out of 256 bits
  return r;
}
LHD fr_t fr_sub(const fr_t& a, const fr_t& b) {
  fr_t r; uint64_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.v[i] - b.v[i] - bw; r.v[i] = (uint32_t)d; bw = d >> 63; }
  uint32_t m = (uint32_t)0 - (uint32_t)bw;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (uint64_t)r.v[i] + (fr_p_limb(i) & m); r.v[i] = (uint32_t)c; c >>= 32; }
  return r;
}
LHD fr_t fr_neg(const fr_t& a) { return fr_sub(fr_zero(), a); }
LHD fr_t fr_dbl(const fr_t& a) { return fr_add(a, a); }

#if !defined(__HIPCC__) && defined(__SIZEOF_INT128__) && !defined(LASSO_HOST_LIMBS32)
// Host build (the O(log n) tails of the prover): CIOS over 64-bit limbs.
inline fr_t fr_mul(const fr_t& a, const fr_t& b) {
  typedef unsigned __int128 u128;
  const uint64_t P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull}, INV = 0xc2e1f593efffffffull;
  uint64_t x[4], y[4]; __builtin_memcpy(x, a.v, 32); __builtin_memcpy(y, b.v, 32);
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)x[j] * y[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * INV;
    c = (u128)m * P[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; j++) { c += (u128)m * P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  // result < 2p < 2^255
  uint64_t s[4]; uint64_t bw = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - P[i] - bw; s[i] = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1; }
  const uint64_t keep = (uint64_t)0 - bw;
  uint64_t r[4]; for (int i = 0; i < 4; i++) r[i] = (t[i] & keep) | (s[i] & ~keep);
  fr_t o; __builtin_memcpy(o.v, r, 32);
  return o;
}
#else
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
    c = (uint64_t)m * fr_p_limb(0) + t[0]; c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; j++) { c += (uint64_t)m * fr_p_limb(j) + t[j]; t[j - 1] = (uint32_t)c; c >>= 32; }
    c += t[8]; t[7] = (uint32_t)c; c >>= 32;
    t[8] = t[9] + (uint32_t)c;
  }
  fr_t r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  fr_cond_sub_p(r.v);   // result < 2p < 2^255: t[8] == 0
  return r;
}
#endif
LHD fr_t fr_sqr(const fr_t& a) { return fr_mul(a, a); }

LHD fr_t fr_from_u64(uint64_t x) {
  fr_t t = fr_zero(); t.v[0] = (uint32_t)x; t.v[1] = (uint32_t)(x >> 32);
  return fr_mul(t, fr_r2());
}
LHD fr_t fr_to_canonical(const fr_t& a) { fr_t o = fr_zero(); o.v[0] = 1; return fr_mul(a, o); }
LHD fr_t fr_from_canonical(const fr_t& c) {
  fr_t t = c;
  for (int k = 0; k < 8 && fr_geq_p(t.v); k++) {   // c < 2^256 < 6p
    uint64_t bw = 0;
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)t.v[i] - fr_p_limb(i) - bw; t.v[i] = (uint32_t)d; bw = d >> 63; }
  }
  return fr_mul(t, fr_r2());
}
LHD fr_t fr_pow(const fr_t& a, const uint32_t* e) {
  fr_t r = fr_one();
  for (int i = 255; i >= 0; i--) { r = fr_sqr(r); if ((e[i / 32] >> (i % 32)) & 1) r = fr_mul(r, a); }
  return r;
}
LHD fr_t fr_inv(const fr_t& a) {  // Fermat; inverse(0) = 0
  uint32_t e[8]; for (int i = 0; i < 8; i++) e[i] = fr_p_limb(i);
  e[0] -= 2u;
  return fr_pow(a, e);
}
LHD int fr_canonical_bits(const fr_t& c) {
  for (int i = 7; i >= 0; i--) if (c.v[i]) { uint32_t x = c.v[i]; int n = 0; while (x) { n++; x >>= 1; } return 32 * i + n; }
  return 0;
}

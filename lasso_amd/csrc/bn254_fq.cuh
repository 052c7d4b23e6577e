// BN254 build (-DLASSO_BN254) of fq.cuh: Fq = the base field of ark-bn254's G1,
// q = 21888242871839275222246405745257275088696311157297823662689037894645226208583, and the group y^2 = x^3 + 3 over it.
// Unlike the curve25519 header (plain lazy limbs around 2^255 - 19), fq_t here IS ark-ff's in-memory Montgomery form (R = 2^256), always
// canonical; fq_from_mont / fq_to_mont are the identity and stay only so that shared code reads the same.  Points are homogeneous projective
// (X : Y : Z), x = X/Z, y = Y/Z, identity (0 : 1 : 0), under the COMPLETE formulas of Renes-Costello-Batina 2016 (a = 0: algorithms 7-9) — no
// exceptional cases, so bucket sums and trees need no branches.  ed_point keeps the four-coordinate layout of the ABI's lasso_point (t unused,
// zero): ark-ec's `short_weierstrass::Projective` is Jacobian, but the transcript only ever sees the compressed affine point
// (utils/transcript.rs:47-51), so any projective representative is equivalent downstream.
#pragma once
#include <stdint.h>
#include "fr.cuh"

struct alignas(16) fq_t {
  uint32_t v[8];
};

#define FQ_INV32 0xe4866389u  // -q^{-1} mod 2^32
LHD uint32_t fq_p_limb(int i) {
  const uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  return P[i];
}
LHD fq_t fq_zero() { fq_t r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
LHD fq_t fq_one() {  // R mod p
  fq_t r; r.v[0] = 0xc58f0d9du; r.v[1] = 0xd35d438du; r.v[2] = 0xf5c70b3du; r.v[3] = 0x0a78eb28u;
  r.v[4] = 0x7879462cu; r.v[5] = 0x666ea36fu; r.v[6] = 0x9a07df2fu; r.v[7] = 0x0e0a77c1u; return r;
}
LHD fq_t fq_r2() {  // R^2 mod p
  fq_t r; r.v[0] = 0x538afa89u; r.v[1] = 0xf32cfc5bu; r.v[2] = 0xd44501fbu; r.v[3] = 0xb5e71911u;
  r.v[4] = 0x0a417ff6u; r.v[5] = 0x47ab1effu; r.v[6] = 0xcab8351fu; r.v[7] = 0x06d89f71u; return r;
}
LHD bool fq_is_zero(const fq_t& a) { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= a.v[i]; return o == 0; }
LHD bool fq_eq(const fq_t& a, const fq_t& b) { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i]; return o == 0; }

LHD bool fq_geq_p(const uint32_t* a) {
  for (int i = 7; i >= 0; i--) { const uint32_t pi = fq_p_limb(i); if (a[i] != pi) return a[i] > pi; }
  return true;
}
// r = a - p if a >= p (a < 2p), branch-free
LHD void fq_cond_sub_p(uint32_t* a) {
  uint32_t t[8]; uint64_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a[i] - fq_p_limb(i) - bw; t[i] = (uint32_t)d; bw = (d >> 63); }
  uint32_t keep = (uint32_t)0 - (uint32_t)bw;
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = (a[i] & keep) | (t[i] & ~keep);
}
LHD fq_t fq_add(const fq_t& a, const fq_t& b) {
  fq_t r; uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
  fq_cond_sub_p(r.v);   // a, b < p < 2^254: no carry out of 256 bits
  return r;
}
LHD fq_t fq_sub(const fq_t& a, const fq_t& b) {
  fq_t r; uint64_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.v[i] - b.v[i] - bw; r.v[i] = (uint32_t)d; bw = d >> 63; }
  uint32_t m = (uint32_t)0 - (uint32_t)bw;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (uint64_t)r.v[i] + (fq_p_limb(i) & m); r.v[i] = (uint32_t)c; c >>= 32; }
  return r;
}
LHD fq_t fq_neg(const fq_t& a) { return fq_sub(fq_zero(), a); }
LHD fq_t fq_dbl(const fq_t& a) { return fq_add(a, a); }

#if !defined(__HIPCC__) && defined(__SIZEOF_INT128__) && !defined(LASSO_HOST_LIMBS32)
// Host build (the O(log n) tails of the prover): CIOS over 64-bit limbs.
inline fq_t fq_mul(const fq_t& a, const fq_t& b) {
  typedef unsigned __int128 u128;
  const uint64_t P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}, INV = 0x87d20782e4866389ull;
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
  fq_t o; __builtin_memcpy(o.v, r, 32);
  return o;
}
#else
LHD fq_t fq_mul(const fq_t& a, const fq_t& b) {
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
    const uint32_t m = t[0] * FQ_INV32;
    c = (uint64_t)m * fq_p_limb(0) + t[0]; c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; j++) { c += (uint64_t)m * fq_p_limb(j) + t[j]; t[j - 1] = (uint32_t)c; c >>= 32; }
    c += t[8]; t[7] = (uint32_t)c; c >>= 32;
    t[8] = t[9] + (uint32_t)c;
  }
  fq_t r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  fq_cond_sub_p(r.v);   // result < 2p < 2^255: t[8] == 0
  return r;
}
#endif
LHD fq_t fq_sqr(const fq_t& a) { return fq_mul(a, a); }

LHD fq_t fq_from_u64(uint64_t x) {
  fq_t t = fq_zero(); t.v[0] = (uint32_t)x; t.v[1] = (uint32_t)(x >> 32);
  return fq_mul(t, fq_r2());
}
LHD fq_t fq_to_canonical(const fq_t& a) { fq_t o = fq_zero(); o.v[0] = 1; return fq_mul(a, o); }
LHD fq_t fq_from_canonical(const fq_t& c) {
  fq_t t = c;
  for (int k = 0; k < 8 && fq_geq_p(t.v); k++) {   // c < 2^256 < 6p
    uint64_t bw = 0;
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)t.v[i] - fq_p_limb(i) - bw; t.v[i] = (uint32_t)d; bw = d >> 63; }
  }
  return fq_mul(t, fq_r2());
}
LHD fq_t fq_pow(const fq_t& a, const uint32_t* e) {
  fq_t r = fq_one();
  for (int i = 255; i >= 0; i--) { r = fq_sqr(r); if ((e[i / 32] >> (i % 32)) & 1) r = fq_mul(r, a); }
  return r;
}
LHD fq_t fq_inv(const fq_t& a) {  // Fermat; inverse(0) = 0
  uint32_t e[8]; for (int i = 0; i < 8; i++) e[i] = fq_p_limb(i);
  e[0] -= 2u;
  return fq_pow(a, e);
}
LHD int fq_canonical_bits(const fq_t& c) {
  for (int i = 7; i >= 0; i--) if (c.v[i]) { uint32_t x = c.v[i]; int n = 0; while (x) { n++; x >>= 1; } return 32 * i + n; }
  return 0;
}

LHD fq_t fq_from_limbs(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7) {
  fq_t r; r.v[0] = a0; r.v[1] = a1; r.v[2] = a2; r.v[3] = a3; r.v[4] = a4; r.v[5] = a5; r.v[6] = a6; r.v[7] = a7; return r;
}
LHD fq_t fq_from_mont(const fq_t& a) { return a; }
LHD fq_t fq_to_mont(const fq_t& a) { return a; }
LHD fq_t fq_canonical(const fq_t& a) { return a; }   // values are kept canonical
LHD fq_t fq_inv_chain(const fq_t& a) { return fq_inv(a); }
LHD fq_t fq_mul3(const fq_t& a) { return fq_add(fq_dbl(a), a); }
LHD fq_t fq_mul9(const fq_t& a) { const fq_t t = fq_dbl(fq_dbl(fq_dbl(a))); return fq_add(t, a); }   // b3 = 3 b = 9

struct ed_point { fq_t X, Y, T, Z; };   // (X : Y : Z), T unused (zero): the lasso_point layout
LHD ed_point ed_identity() { ed_point p; p.X = fq_zero(); p.Y = fq_one(); p.T = fq_zero(); p.Z = fq_zero(); return p; }
LHD ed_point ed_from_affine(const fq_t& x, const fq_t& y) { ed_point p; p.X = x; p.Y = y; p.T = fq_zero(); p.Z = fq_one(); return p; }
LHD ed_point ed_neg(const ed_point& p) { ed_point r = p; r.Y = fq_neg(p.Y); return r; }
LHD bool ed_eq(const ed_point& a, const ed_point& b) {
  return fq_eq(fq_mul(a.X, b.Z), fq_mul(b.X, a.Z)) && fq_eq(fq_mul(a.Y, b.Z), fq_mul(b.Y, a.Z)) && fq_eq(fq_mul(a.X, b.Y), fq_mul(b.X, a.Y));
}
// complete addition (RCB16 algorithm 7, a = 0, b3 = 9): 12 products
LHD ed_point ed_add(const ed_point& p, const ed_point& q) {
  fq_t t0 = fq_mul(p.X, q.X), t1 = fq_mul(p.Y, q.Y), t2 = fq_mul(p.Z, q.Z);
  fq_t t3 = fq_sub(fq_sub(fq_mul(fq_add(p.X, p.Y), fq_add(q.X, q.Y)), t0), t1);   // X1Y2 + X2Y1
  fq_t t4 = fq_sub(fq_sub(fq_mul(fq_add(p.Y, p.Z), fq_add(q.Y, q.Z)), t1), t2);   // Y1Z2 + Y2Z1
  fq_t y3 = fq_sub(fq_sub(fq_mul(fq_add(p.X, p.Z), fq_add(q.X, q.Z)), t0), t2);   // X1Z2 + X2Z1
  t0 = fq_mul3(t0); t2 = fq_mul9(t2);
  fq_t z3 = fq_add(t1, t2); t1 = fq_sub(t1, t2); y3 = fq_mul9(y3);
  ed_point r;
  r.X = fq_sub(fq_mul(t3, t1), fq_mul(t4, y3));
  r.Y = fq_add(fq_mul(t1, z3), fq_mul(y3, t0));
  r.Z = fq_add(fq_mul(z3, t4), fq_mul(t0, t3));
  r.T = fq_zero();
  return r;
}
// complete doubling (RCB16 algorithm 9, a = 0)
LHD ed_point ed_dbl(const ed_point& p) {
  fq_t t0 = fq_sqr(p.Y), z3 = fq_dbl(fq_dbl(fq_dbl(t0))), t1 = fq_mul(p.Y, p.Z), t2 = fq_mul9(fq_sqr(p.Z));
  fq_t x3 = fq_mul(t2, z3), y3 = fq_add(t0, t2);
  z3 = fq_mul(t1, z3);
  t0 = fq_sub(t0, fq_mul3(t2));
  ed_point r;
  r.Y = fq_add(fq_mul(t0, y3), x3);
  r.X = fq_dbl(fq_mul(t0, fq_mul(p.X, p.Y)));
  r.Z = z3; r.T = fq_zero();
  return r;
}
LHD ed_point ed_mul_limbs(const ed_point& p, const uint32_t* e) {   // e: canonical 8-limb scalar
  ed_point r = ed_identity();
  for (int i = 255; i >= 0; i--) { r = ed_dbl(r); if ((e[i / 32] >> (i % 32)) & 1) r = ed_add(r, p); }
  return r;
}

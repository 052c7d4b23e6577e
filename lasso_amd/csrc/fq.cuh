// Fq = 2^255 - 19 (curve25519 base field) and the twisted Edwards group -x^2+y^2 = 1+d x^2 y^2
// (`ark_curve25519::EdwardsProjective`, the group the reference instantiates everywhere:
// src/benches/bench.rs:6).  Internal form: plain (non-Montgomery) 8 x u32 limbs with LAZY reduction —
// any 256-bit value congruent to the element — because 2^256 = 38 (mod p) makes reduction a
// multiply-by-38 fold (8 mads) instead of a Montgomery row (32+ mads).  ark-ff's in-memory Montgomery
// form (R = 2^256 = 38) only exists at the ABI: mont = plain*38, plain = mont*38^-1.
// Points: extended (X:Y:T:Z) = ark-ec `twisted_edwards::Projective`; bases are kept as precomputed
// affine "Niels" triples (y+x, y-x, 2dxy) so bucket accumulation is a 7-multiplication mixed add.
#pragma once
#ifdef LASSO_BN254
#include "bn254_fq.cuh"   // the same interface over ark-bn254's Fq and G1
#else
#include <stdint.h>
#include "fr.cuh"

struct alignas(16) fq_t { uint32_t v[8]; };

LHD fq_t fq_zero() { fq_t r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
LHD fq_t fq_one() { fq_t r = fq_zero(); r.v[0] = 1; return r; }
LHD fq_t fq_from_limbs(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7) {
  fq_t r; r.v[0] = a0; r.v[1] = a1; r.v[2] = a2; r.v[3] = a3; r.v[4] = a4; r.v[5] = a5; r.v[6] = a6; r.v[7] = a7; return r;
}
LHD fq_t fq_d2() { return fq_from_limbs(0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u, 0x56dffce7u, 0x2406d9dcu); }
LHD fq_t fq_d() { return fq_from_limbs(0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu); }
LHD fq_t fq_inv38() { return fq_from_limbs(0x9435e50au, 0x435e50d7u, 0x35e50d79u, 0x5e50d794u, 0xe50d7943u, 0x50d79435u, 0x0d79435eu, 0x179435e5u); }

// fold a carry/borrow of weight 2^256 (= 38 mod p) back in; `k` is 0 or 1
LHD void fq_fold_add38(uint32_t* r, uint32_t k) {
  uint64_t c = (uint64_t)38 * k;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += r[i]; r[i] = (uint32_t)c; c >>= 32; }
  // a second wrap is only possible when r was within 38 of 2^256; then r is now < 38 and this cannot wrap again
  r[0] += 38u * (uint32_t)c;
}
LHD fq_t fq_add(const fq_t& a, const fq_t& b) {
  fq_t r; uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
  fq_fold_add38(r.v, (uint32_t)c);
  return r;
}
LHD fq_t fq_sub(const fq_t& a, const fq_t& b) {
  fq_t r; uint64_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.v[i] - b.v[i] - bw; r.v[i] = (uint32_t)d; bw = d >> 63; }
  // borrowed: r = a-b+2^256 = a-b+38 (mod p): subtract 38; a second borrow wraps once more
  uint64_t s = (uint64_t)38 * bw; uint64_t bw2 = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)r.v[i] - (i == 0 ? s : 0) - bw2; r.v[i] = (uint32_t)d; bw2 = d >> 63; }
  r.v[0] -= 38u * (uint32_t)bw2;  // r >= 2^256-38 here, no further borrow
  return r;
}
LHD fq_t fq_neg(const fq_t& a) { return fq_sub(fq_zero(), a); }
LHD fq_t fq_dbl(const fq_t& a) { return fq_add(a, a); }

#if !defined(__HIPCC__) && defined(__SIZEOF_INT128__) && !defined(LASSO_HOST_LIMBS32)   /* LASSO_HOST_LIMBS32: tests force the device form on the host */
// Host build (g++): 4 x 64-bit schoolbook with 128-bit products, then the same 2^256 = 38 fold.
inline fq_t fq_mul(const fq_t& a, const fq_t& b) {
  typedef unsigned __int128 u128;
  uint64_t x[4], y[4], t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; __builtin_memcpy(x, a.v, 32); __builtin_memcpy(y, b.v, 32);
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)x[j] * y[i] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
    t[i + 4] = (uint64_t)c;
  }
  uint64_t r[4]; u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)t[i + 4] * 38u + t[i]; r[i] = (uint64_t)c; c >>= 64; }
  c *= 38u;   // c <= 38: fold, then at most one more wrap
  for (int i = 0; i < 4; i++) { c += r[i]; r[i] = (uint64_t)c; c >>= 64; }
  r[0] += 38u * (uint64_t)c;
  fq_t o; __builtin_memcpy(o.v, r, 32); return o;
}
#else
LHD fq_t fq_mul(const fq_t& a, const fq_t& b) {
  uint32_t t[16];
#pragma unroll
  for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
    const uint32_t bi = b.v[i];
#pragma unroll
    for (int j = 0; j < 8; j++) { c += (uint64_t)a.v[j] * bi + t[i + j]; t[i + j] = (uint32_t)c; c >>= 32; }
    t[i + 8] = (uint32_t)c;
  }
  fq_t r; uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (uint64_t)t[i + 8] * 38u + t[i]; r.v[i] = (uint32_t)c; c >>= 32; }
  // c <= 38: fold c*38 (<= 1444), then at most one more wrap
  c *= 38u;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += r.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
  r.v[0] += 38u * (uint32_t)c;
  return r;
}
#endif
LHD fq_t fq_sqr(const fq_t& a) { return fq_mul(a, a); }

// unique representative in [0, p)
LHD fq_t fq_canonical(const fq_t& a) {
  fq_t r = a;
  uint32_t top = r.v[7] >> 31; r.v[7] &= 0x7fffffffu;
  uint64_t c = (uint64_t)19 * top;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += r.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
  // now r < 2^255 + 19 < 2p: subtract p = 2^255 - 19 if r >= p, i.e. if r + 19 >= 2^255
  uint32_t t[8]; c = 19;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += r.v[i]; t[i] = (uint32_t)c; c >>= 32; }
  uint32_t ge = t[7] >> 31;  // (r + 19) has bit 255 set  <=>  r >= p
  t[7] &= 0x7fffffffu;       // r + 19 - 2^255 = r - p
  uint32_t m = (uint32_t)0 - ge;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = (r.v[i] & ~m) | (t[i] & m);
  return r;
}
LHD bool fq_is_zero(const fq_t& a) { fq_t c = fq_canonical(a); uint32_t o = 0; for (int i = 0; i < 8; i++) o |= c.v[i]; return o == 0; }
LHD bool fq_eq(const fq_t& a, const fq_t& b) { return fq_is_zero(fq_sub(a, b)); }
LHD fq_t fq_pow(const fq_t& a, const uint32_t* e) {
  fq_t r = fq_one();
  for (int i = 255; i >= 0; i--) { r = fq_sqr(r); if ((e[i / 32] >> (i % 32)) & 1) r = fq_mul(r, a); }
  return r;
}
// curve25519 inversion chain: a^(p-2), 254 squarings + 11 multiplications
LHD fq_t fq_inv_chain(const fq_t& z) {
  fq_t z2 = fq_sqr(z);
  fq_t z8 = fq_sqr(fq_sqr(z2));
  fq_t z9 = fq_mul(z8, z);
  fq_t z11 = fq_mul(z9, z2);
  fq_t z22 = fq_sqr(z11);
  fq_t z_5_0 = fq_mul(z22, z9);                       // 2^5 - 1
  fq_t t = z_5_0; for (int i = 0; i < 5; i++) t = fq_sqr(t);
  fq_t z_10_0 = fq_mul(t, z_5_0);                     // 2^10 - 1
  t = z_10_0; for (int i = 0; i < 10; i++) t = fq_sqr(t);
  fq_t z_20_0 = fq_mul(t, z_10_0);
  t = z_20_0; for (int i = 0; i < 20; i++) t = fq_sqr(t);
  t = fq_mul(t, z_20_0);                              // 2^40 - 1
  for (int i = 0; i < 10; i++) t = fq_sqr(t);
  fq_t z_50_0 = fq_mul(t, z_10_0);
  t = z_50_0; for (int i = 0; i < 50; i++) t = fq_sqr(t);
  fq_t z_100_0 = fq_mul(t, z_50_0);
  t = z_100_0; for (int i = 0; i < 100; i++) t = fq_sqr(t);
  t = fq_mul(t, z_100_0);                             // 2^200 - 1
  for (int i = 0; i < 50; i++) t = fq_sqr(t);
  t = fq_mul(t, z_50_0);                              // 2^250 - 1
  for (int i = 0; i < 5; i++) t = fq_sqr(t);
  return fq_mul(t, z11);                              // 2^255 - 21
}

LHD fq_t fq_inv(const fq_t& a) { return fq_inv_chain(a); }   // inverse(0) = 0
// ABI conversions (ark-ff Montgomery, R = 2^256 = 38 mod p)
LHD fq_t fq_from_mont(const fq_t& m) { return fq_mul(m, fq_inv38()); }
LHD fq_t fq_to_mont(const fq_t& a) {  // canonical Montgomery limbs, as ark-ff stores them
  fq_t k = fq_zero(); k.v[0] = 38; return fq_canonical(fq_mul(a, k));
}

// ------------------------------------------------------------------ group
struct ed_point { fq_t X, Y, T, Z; };
struct ed_niels { fq_t ypx, ymx, t2d; };

LHD ed_point ed_identity() { ed_point p; p.X = fq_zero(); p.Y = fq_one(); p.T = fq_zero(); p.Z = fq_one(); return p; }
LHD ed_niels ed_to_niels_affine(const fq_t& x, const fq_t& y) {
  ed_niels n; n.ypx = fq_add(y, x); n.ymx = fq_sub(y, x); n.t2d = fq_mul(fq_mul(x, y), fq_d2()); return n;
}
LHD ed_point ed_from_affine(const fq_t& x, const fq_t& y) { ed_point p; p.X = x; p.Y = y; p.T = fq_mul(x, y); p.Z = fq_one(); return p; }
// add-2008-hwcd-3 (a = -1), unified and complete on this curve: 9M
LHD ed_point ed_add(const ed_point& p, const ed_point& q) {
  fq_t A = fq_mul(fq_sub(p.Y, p.X), fq_sub(q.Y, q.X));
  fq_t B = fq_mul(fq_add(p.Y, p.X), fq_add(q.Y, q.X));
  fq_t C = fq_mul(fq_mul(p.T, fq_d2()), q.T);
  fq_t D = fq_dbl(fq_mul(p.Z, q.Z));
  fq_t E = fq_sub(B, A), F = fq_sub(D, C), G = fq_add(D, C), H = fq_add(B, A);
  ed_point r; r.X = fq_mul(E, F); r.Y = fq_mul(G, H); r.T = fq_mul(E, H); r.Z = fq_mul(F, G); return r;
}
// mixed add with a precomputed affine base: 7M
LHD ed_point ed_madd(const ed_point& p, const ed_niels& n) {
  fq_t A = fq_mul(fq_sub(p.Y, p.X), n.ymx);
  fq_t B = fq_mul(fq_add(p.Y, p.X), n.ypx);
  fq_t C = fq_mul(p.T, n.t2d);
  fq_t D = fq_dbl(p.Z);
  fq_t E = fq_sub(B, A), F = fq_sub(D, C), G = fq_add(D, C), H = fq_add(B, A);
  ed_point r; r.X = fq_mul(E, F); r.Y = fq_mul(G, H); r.T = fq_mul(E, H); r.Z = fq_mul(F, G); return r;
}
LHD ed_point ed_msub(const ed_point& p, const ed_niels& n) {  // p - base
  fq_t A = fq_mul(fq_sub(p.Y, p.X), n.ypx);
  fq_t B = fq_mul(fq_add(p.Y, p.X), n.ymx);
  fq_t C = fq_mul(p.T, n.t2d);
  fq_t D = fq_dbl(p.Z);
  fq_t E = fq_sub(B, A), F = fq_add(D, C), G = fq_sub(D, C), H = fq_add(B, A);
  ed_point r; r.X = fq_mul(E, F); r.Y = fq_mul(G, H); r.T = fq_mul(E, H); r.Z = fq_mul(F, G); return r;
}
LHD ed_point ed_dbl(const ed_point& p) {  // dbl-2008-hwcd, a = -1
  fq_t A = fq_sqr(p.X), B = fq_sqr(p.Y), C = fq_dbl(fq_sqr(p.Z));
  fq_t D = fq_neg(A);
  fq_t E = fq_sub(fq_sub(fq_sqr(fq_add(p.X, p.Y)), A), B);
  fq_t G = fq_add(D, B), F = fq_sub(G, C), H = fq_sub(D, B);
  ed_point r; r.X = fq_mul(E, F); r.Y = fq_mul(G, H); r.T = fq_mul(E, H); r.Z = fq_mul(F, G); return r;
}
LHD ed_point ed_neg(const ed_point& p) { ed_point r = p; r.X = fq_neg(p.X); r.T = fq_neg(p.T); return r; }
LHD bool ed_eq(const ed_point& a, const ed_point& b) { return fq_eq(fq_mul(a.X, b.Z), fq_mul(b.X, a.Z)) && fq_eq(fq_mul(a.Y, b.Z), fq_mul(b.Y, a.Z)); }
// scalar given as canonical 8-limb integer (host tails: commitments of one or two scalars, window recombination)
LHD ed_point ed_mul_limbs(const ed_point& p, const uint32_t* e, int nbits = 256) {
  ed_point r = ed_identity();
  for (int i = nbits - 1; i >= 0; i--) { r = ed_dbl(r); if ((e[i / 32] >> (i % 32)) & 1) r = ed_add(r, p); }
  return r;
}
#endif  // LASSO_BN254

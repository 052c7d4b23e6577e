// Host-side scalar / point helpers for the O(log n) tails of the prover.  Same arithmetic headers as the device
// (lasso_amd/csrc/fr.cuh, fq.cuh are __host__ __device__), so there is exactly one field implementation in the product.
#pragma once
#include <cstring>
#include <vector>
#include <stdexcept>
#include "../csrc/fq.cuh"
#include "../../include/lasso_hip.h"
#include "modinv.hpp"

namespace lasso {

// host inversions: safegcd (modinv.hpp), ~10x faster than the Fermat chains the device code keeps (fr_inv / fq_inv remain the reference the
// tests compare against).  Montgomery form: (aR)^-1 as an integer is a^-1 R^-1, one Montgomery product with R^3 brings it back to a^-1 R.
#ifdef LASSO_BN254
static const uint32_t FR_R3_WORDS[8] = {0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu, 0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu};   // 2^768 mod p (BN254 Fr)
#else
static const uint32_t FR_R3_WORDS[8] = {0x7b83a2dbu, 0x2a9e4968u, 0xaef7f3ecu, 0x278324e6u, 0x04ec5b65u, 0x8065dc6cu, 0x3599cec7u, 0x0e530b77u};   // 2^768 mod p (curve25519 Fr)
#endif
inline fr_t fr_inv_host(const fr_t& a) {
  static const ModInv256 mi([] { static uint64_t P[4]; for (int i = 0; i < 4; i++) P[i] = ((uint64_t)fr_p_limb(2 * i + 1) << 32) | fr_p_limb(2 * i); return (const uint64_t*)P; }());
  uint64_t x[4], y[4]; memcpy(x, a.v, 32);
  if (!mi.inverse(x, y)) return fr_inv(a);
  fr_t t, r3; memcpy(t.v, y, 32);
  memcpy(r3.v, FR_R3_WORDS, 32);
  return fr_mul(t, r3);
}
#ifdef LASSO_BN254
inline fq_t fq_inv_host(const fq_t& a) {   // Montgomery Fq (canonical): (aR)^-1 as an integer is a^-1 R^-1; one product with R^3 gives a^-1 R
  static const uint64_t Q[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  static const ModInv256 mi(Q);
  uint64_t x[4], y[4]; memcpy(x, a.v, 32);
  if (!mi.inverse(x, y)) return fq_inv(a);
  fq_t t; memcpy(t.v, y, 32);
  return fq_mul(t, fq_from_limbs(0xda1530dfu, 0xb1cd6dafu, 0xa7283db6u, 0x62f210e6u, 0x0ada0afbu, 0xef7f0b0cu, 0x2d592544u, 0x20fd6e90u));   // 2^768 mod q
}
#else
inline fq_t fq_inv_host(const fq_t& a) {   // plain (non-Montgomery) Fq, lazily reduced input
  static const uint64_t Q[4] = {0xffffffffffffffedull, ~0ull, ~0ull, 0x7fffffffffffffffull};
  static const ModInv256 mi(Q);
  const fq_t c = fq_canonical(a);
  uint64_t x[4], y[4]; memcpy(x, c.v, 32);
  if (!mi.inverse(x, y)) return fq_inv(a);
  fq_t r; memcpy(r.v, y, 32); return r;
}
#endif

// Host additions / subtractions over four 64-bit limbs (the shared fr_add / fr_sub walk eight 32-bit limbs — the form the device executes — at 17 ns each on x86-64; these take
// 3-4 ns).  The rounds the host finishes (prover.hpp host_cubic_rounds) are as many additions as products.  Operands canonical (< p), as for fr_add / fr_sub; same results
// (tests/cpp/test_arith_host.cpp compares them).
#if defined(__x86_64__) && !defined(LASSO_HOST_LIMBS32)
}  // namespace lasso
#include <x86intrin.h>
namespace lasso {
struct FrModulus64 { unsigned long long p[4]; FrModulus64() { for (int i = 0; i < 4; i++) p[i] = ((unsigned long long)fr_p_limb(2 * i + 1) << 32) | fr_p_limb(2 * i); } };
static const FrModulus64 g_fr_p64;
inline fr_t fr_add_host(const fr_t& a, const fr_t& b) {   // adc / sbb chains (the 128-bit-integer form of the same loops compiled to 25 ns)
  unsigned long long x[4], y[4], s[4], d[4]; memcpy(x, a.v, 32); memcpy(y, b.v, 32);
  unsigned char c = 0;
  c = _addcarry_u64(c, x[0], y[0], &s[0]); c = _addcarry_u64(c, x[1], y[1], &s[1]); c = _addcarry_u64(c, x[2], y[2], &s[2]); c = _addcarry_u64(c, x[3], y[3], &s[3]);   // a, b < p < 2^254: no carry out
  unsigned char bw = 0;
  bw = _subborrow_u64(bw, s[0], g_fr_p64.p[0], &d[0]); bw = _subborrow_u64(bw, s[1], g_fr_p64.p[1], &d[1]); bw = _subborrow_u64(bw, s[2], g_fr_p64.p[2], &d[2]); bw = _subborrow_u64(bw, s[3], g_fr_p64.p[3], &d[3]);
  for (int i = 0; i < 4; i++) s[i] = bw ? s[i] : d[i];   // borrow: the sum is below p
  fr_t r; memcpy(r.v, s, 32); return r;
}
inline fr_t fr_sub_host(const fr_t& a, const fr_t& b) {
  unsigned long long x[4], y[4], d[4], r4[4]; memcpy(x, a.v, 32); memcpy(y, b.v, 32);
  unsigned char bw = 0;
  bw = _subborrow_u64(bw, x[0], y[0], &d[0]); bw = _subborrow_u64(bw, x[1], y[1], &d[1]); bw = _subborrow_u64(bw, x[2], y[2], &d[2]); bw = _subborrow_u64(bw, x[3], y[3], &d[3]);
  const unsigned long long m = 0ull - (unsigned long long)bw;   // borrowed: add p back
  unsigned char c = 0;
  c = _addcarry_u64(c, d[0], g_fr_p64.p[0] & m, &r4[0]); c = _addcarry_u64(c, d[1], g_fr_p64.p[1] & m, &r4[1]); c = _addcarry_u64(c, d[2], g_fr_p64.p[2] & m, &r4[2]); c = _addcarry_u64(c, d[3], g_fr_p64.p[3] & m, &r4[3]);
  fr_t r; memcpy(r.v, r4, 32); return r;
}
// The same field as plain 4 x u64 values that STAY in that form across operations (H4): going through fr_t (eight 32-bit words, 16-byte aligned) between two operations costs a
// store-forwarding stall each time — 13 ns per addition where the adc chain itself takes 3.  For the rounds the host finishes (prover.hpp host_cubic_rounds).
struct H4 { unsigned long long l[4]; };
struct FrMont64 { unsigned long long inv; FrMont64() { unsigned long long x = 1; for (int i = 0; i < 7; i++) x *= 2 - g_fr_p64.p[0] * x; inv = 0ull - x; } };   // -p^-1 mod 2^64 (Newton)
static const FrMont64 g_fr_mont64;
inline H4 h4_add(const H4& a, const H4& b) {
  H4 s, d; unsigned char c = 0, bw = 0;
  c = _addcarry_u64(c, a.l[0], b.l[0], &s.l[0]); c = _addcarry_u64(c, a.l[1], b.l[1], &s.l[1]); c = _addcarry_u64(c, a.l[2], b.l[2], &s.l[2]); c = _addcarry_u64(c, a.l[3], b.l[3], &s.l[3]);
  bw = _subborrow_u64(bw, s.l[0], g_fr_p64.p[0], &d.l[0]); bw = _subborrow_u64(bw, s.l[1], g_fr_p64.p[1], &d.l[1]); bw = _subborrow_u64(bw, s.l[2], g_fr_p64.p[2], &d.l[2]); bw = _subborrow_u64(bw, s.l[3], g_fr_p64.p[3], &d.l[3]);
  for (int i = 0; i < 4; i++) s.l[i] = bw ? s.l[i] : d.l[i];
  return s;
}
inline H4 h4_sub(const H4& a, const H4& b) {
  H4 d, r; unsigned char bw = 0, c = 0;
  bw = _subborrow_u64(bw, a.l[0], b.l[0], &d.l[0]); bw = _subborrow_u64(bw, a.l[1], b.l[1], &d.l[1]); bw = _subborrow_u64(bw, a.l[2], b.l[2], &d.l[2]); bw = _subborrow_u64(bw, a.l[3], b.l[3], &d.l[3]);
  const unsigned long long m = 0ull - (unsigned long long)bw;
  c = _addcarry_u64(c, d.l[0], g_fr_p64.p[0] & m, &r.l[0]); c = _addcarry_u64(c, d.l[1], g_fr_p64.p[1] & m, &r.l[1]); c = _addcarry_u64(c, d.l[2], g_fr_p64.p[2] & m, &r.l[2]); c = _addcarry_u64(c, d.l[3], g_fr_p64.p[3] & m, &r.l[3]);
  return r;
}
inline H4 h4_mul(const H4& a, const H4& b) {   // Montgomery product (CIOS, R = 2^256): the function fr_mul computes, operands and result canonical
  typedef unsigned __int128 u128;
  const unsigned long long* P = g_fr_p64.p; const unsigned long long INV = g_fr_mont64.inv;
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
  for (int i = 0; i < 4; i++) {
    const unsigned long long y = b.l[i];
    u128 c = (u128)a.l[0] * y + t0; t0 = (unsigned long long)c; c >>= 64;
    c += (u128)a.l[1] * y + t1; t1 = (unsigned long long)c; c >>= 64;
    c += (u128)a.l[2] * y + t2; t2 = (unsigned long long)c; c >>= 64;
    c += (u128)a.l[3] * y + t3; t3 = (unsigned long long)c; c >>= 64;
    c += t4; t4 = (unsigned long long)c; const unsigned long long t5 = (unsigned long long)(c >> 64);
    const unsigned long long m = t0 * INV;
    c = (u128)m * P[0] + t0; c >>= 64;
    c += (u128)m * P[1] + t1; t0 = (unsigned long long)c; c >>= 64;
    c += (u128)m * P[2] + t2; t1 = (unsigned long long)c; c >>= 64;
    c += (u128)m * P[3] + t3; t2 = (unsigned long long)c; c >>= 64;
    c += t4; t3 = (unsigned long long)c; t4 = t5 + (unsigned long long)(c >> 64);
  }
  H4 s = {{t0, t1, t2, t3}}, d; unsigned char bw = 0;   // < 2p < 2^255: t4 == 0
  bw = _subborrow_u64(bw, s.l[0], P[0], &d.l[0]); bw = _subborrow_u64(bw, s.l[1], P[1], &d.l[1]); bw = _subborrow_u64(bw, s.l[2], P[2], &d.l[2]); bw = _subborrow_u64(bw, s.l[3], P[3], &d.l[3]);
  for (int i = 0; i < 4; i++) s.l[i] = bw ? s.l[i] : d.l[i];
  return s;
}
#define LASSO_HAVE_H4 1
#else
inline fr_t fr_add_host(const fr_t& a, const fr_t& b) { return fr_add(a, b); }
inline fr_t fr_sub_host(const fr_t& a, const fr_t& b) { return fr_sub(a, b); }
struct H4 { fr_t v; };   // portable form: the shared arithmetic
inline H4 h4_add(const H4& a, const H4& b) { H4 r; r.v = fr_add(a.v, b.v); return r; }
inline H4 h4_sub(const H4& a, const H4& b) { H4 r; r.v = fr_sub(a.v, b.v); return r; }
inline H4 h4_mul(const H4& a, const H4& b) { H4 r; r.v = fr_mul(a.v, b.v); return r; }
#endif
inline H4 h4_from(const fr_t& a) { H4 r; memcpy(&r, a.v, 32); return r; }
inline fr_t h4_to(const H4& a) { fr_t r; memcpy(r.v, &a, 32); return r; }
inline H4 h4_zero() { H4 r; memset(&r, 0, sizeof(r)); return r; }

struct Sc {  // element of Fr in ark-ff's Montgomery form (bytes == lasso_fr)
  fr_t v;
  static Sc zero() { Sc s; s.v = fr_zero(); return s; }
  static Sc one() { Sc s; s.v = fr_one(); return s; }
  static Sc from_u64(uint64_t x) { Sc s; s.v = fr_from_u64(x); return s; }
  static Sc from_abi(const lasso_fr& f) { Sc s; memcpy(s.v.v, f.l, 32); return s; }
  lasso_fr abi() const { lasso_fr f; memcpy(f.l, v.v, 32); return f; }
  Sc operator+(const Sc& o) const { Sc s; s.v = fr_add_host(v, o.v); return s; }
  Sc operator-(const Sc& o) const { Sc s; s.v = fr_sub_host(v, o.v); return s; }
  Sc operator*(const Sc& o) const { Sc s; s.v = fr_mul(v, o.v); return s; }
  Sc operator-() const { Sc s; s.v = fr_neg(v); return s; }
  Sc& operator+=(const Sc& o) { v = fr_add_host(v, o.v); return *this; }
  Sc& operator-=(const Sc& o) { v = fr_sub_host(v, o.v); return *this; }
  Sc& operator*=(const Sc& o) { v = fr_mul(v, o.v); return *this; }
  bool operator==(const Sc& o) const { return fr_eq(v, o.v); }
  bool is_zero() const { return fr_is_zero(v); }
  Sc square() const { Sc s; s.v = fr_sqr(v); return s; }
  Sc inverse() const { Sc s; s.v = fr_inv_host(v); return s; }
  void to_bytes(uint8_t out[32]) const { fr_t c = fr_to_canonical(v); memcpy(out, c.v, 32); }  // ark-serialize: canonical, little endian
  void canonical_limbs(uint32_t out[8]) const { fr_t c = fr_to_canonical(v); memcpy(out, c.v, 32); }
  // ark-ff from_le_bytes_mod_order on 64 bytes (utils/transcript.rs:61-65): (lo + hi*2^256) mod p
  static Sc from_wide_bytes(const uint8_t b[64]) {
    fr_t lo, hi; memcpy(lo.v, b, 32); memcpy(hi.v, b + 32, 32);
    fr_t r3; memcpy(r3.v, FR_R3_WORDS, 32);   // 2^768 mod p
    Sc s; s.v = fr_add(fr_mul(lo, fr_r2()), fr_mul(hi, r3)); return s;  // Montgomery products accept any 256-bit left operand
  }
};
typedef std::vector<Sc> ScVec;

struct Pt {  // group element, extended coordinates over plain (non-Montgomery) Fq
  ed_point p;
  static Pt identity() { Pt r; r.p = ed_identity(); return r; }
  static Pt from_abi(const lasso_point& q) {
    Pt r; fq_t t;
    memcpy(t.v, q.x, 32); r.p.X = fq_from_mont(t); memcpy(t.v, q.y, 32); r.p.Y = fq_from_mont(t);
    memcpy(t.v, q.t, 32); r.p.T = fq_from_mont(t); memcpy(t.v, q.z, 32); r.p.Z = fq_from_mont(t); return r;
  }
  static Pt from_affine_plain(const fq_t& x, const fq_t& y) { Pt r; r.p = ed_from_affine(x, y); return r; }
  Pt operator+(const Pt& o) const { Pt r; r.p = ed_add(p, o.p); return r; }
  Pt dbl() const { Pt r; r.p = ed_dbl(p); return r; }
  Pt operator*(const Sc& s) const {  // 4-bit fixed window
    uint32_t e[8]; s.canonical_limbs(e);
    ed_point tbl[16]; tbl[0] = ed_identity(); tbl[1] = p;
    for (int i = 2; i < 16; i++) tbl[i] = ed_add(tbl[i - 1], p);
    ed_point acc = ed_identity();
    for (int nib = 63; nib >= 0; nib--) {
      for (int k = 0; k < 4; k++) acc = ed_dbl(acc);
      uint32_t d = (e[nib / 8] >> (4 * (nib % 8))) & 15u;
      if (d) acc = ed_add(acc, tbl[d]);
    }
    Pt r; r.p = acc; return r;
  }
};

// Fixed-base scalar multiplication for the two generators every opening multiplies by fresh scalars (gens_1.G[0] = Q and h:
// dot_product.rs:198-231 computes Cy, delta, beta from them): 64 windows x 15 multiples built once per generator set, then 64 additions.
struct FixedBase {
  std::vector<ed_point> tbl;   // tbl[w*16 + d] = d * 16^w * P
  FixedBase() {}
  explicit FixedBase(const Pt& P) : tbl(64 * 16) {
    ed_point base = P.p;
    for (int w = 0; w < 64; w++) {
      tbl[w * 16] = ed_identity(); tbl[w * 16 + 1] = base;
      for (int d = 2; d < 16; d++) tbl[w * 16 + d] = ed_add(tbl[w * 16 + d - 1], base);
      base = ed_dbl(ed_dbl(ed_dbl(ed_dbl(base))));
    }
  }
  Pt mul(const Sc& s) const {
    uint32_t e[8]; s.canonical_limbs(e);
    ed_point acc = ed_identity();
    for (int w = 0; w < 64; w++) { uint32_t d = (e[w / 8] >> (4 * (w % 8))) & 15u; if (d) acc = ed_add(acc, tbl[w * 16 + d]); }
    Pt r; r.p = acc; return r;
  }
};

#ifdef LASSO_BN254
// ark-serialize compressed short-Weierstrass point (ark-ec SWFlags): canonical x (LE); bit 7 of the last byte = "y is negative" (y > -y as
// canonical integers); bit 6 = point at infinity.  x, y: Montgomery form.
inline void compress_affine(const fq_t& x, const fq_t& y, uint8_t out[32]) {
  const fq_t xc = fq_to_canonical(x), yc = fq_to_canonical(y), ny = fq_to_canonical(fq_neg(y));
  memcpy(out, xc.v, 32);
  bool neg = false;
  for (int i = 7; i >= 0; i--) if (yc.v[i] != ny.v[i]) { neg = yc.v[i] > ny.v[i]; break; }
  if (neg) out[31] |= 0x80;
}
inline void compress_infinity(uint8_t out[32]) { memset(out, 0, 32); out[31] = 0x40; }
#else
// ark-serialize compressed TE point: y (LE) with bit 7 of the last byte = "x is negative" (x > -x as canonical integers)
inline void compress_affine(const fq_t& x, const fq_t& y, uint8_t out[32]) {
  fq_t yc = fq_canonical(y), xc = fq_canonical(x), nx = fq_canonical(fq_neg(x));
  memcpy(out, yc.v, 32);
  bool neg = false;
  for (int i = 7; i >= 0; i--) if (xc.v[i] != nx.v[i]) { neg = xc.v[i] > nx.v[i]; break; }
  if (neg) out[31] |= 0x80;
}
#endif
// normalize_batch + serialize_compressed for many points with ONE inversion (Montgomery's trick)
inline void compress_batch(const std::vector<Pt>& pts, std::vector<uint8_t>& out) {
  size_t n = pts.size(); out.resize(32 * n);
  if (!n) return;
  std::vector<fq_t> pre(n);
  fq_t acc = fq_one();
#ifdef LASSO_BN254
  // a projective point at infinity has Z = 0 (an all-zero row of a commitment): it stays out of the product
  for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!fq_is_zero(pts[i].p.Z)) acc = fq_mul(acc, pts[i].p.Z); }
  fq_t inv = fq_inv_host(acc);
  for (size_t i = n; i-- > 0;) {
    if (fq_is_zero(pts[i].p.Z)) { compress_infinity(&out[32 * i]); continue; }
    fq_t zi = fq_mul(inv, pre[i]); inv = fq_mul(inv, pts[i].p.Z);
    compress_affine(fq_mul(pts[i].p.X, zi), fq_mul(pts[i].p.Y, zi), &out[32 * i]);
  }
#else
  for (size_t i = 0; i < n; i++) { pre[i] = acc; acc = fq_mul(acc, pts[i].p.Z); }
  fq_t inv = fq_inv_host(acc);
  for (size_t i = n; i-- > 0;) {
    fq_t zi = fq_mul(inv, pre[i]); inv = fq_mul(inv, pts[i].p.Z);
    compress_affine(fq_mul(pts[i].p.X, zi), fq_mul(pts[i].p.Y, zi), &out[32 * i]);
  }
#endif
}
inline void compress_one(const Pt& p, uint8_t out[32]) {
#ifdef LASSO_BN254
  if (fq_is_zero(p.p.Z)) { compress_infinity(out); return; }
#endif
  fq_t zi = fq_inv_host(p.p.Z); compress_affine(fq_mul(p.p.X, zi), fq_mul(p.p.Y, zi), out);
}

}  // namespace lasso

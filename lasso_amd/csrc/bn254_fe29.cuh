// BN254 build (-DLASSO_BN254) of fe29.cuh: Fq of ark-bn254 in nine signed 29-bit limbs (mont29.cuh, Montgomery radix 2^261: a coordinate x is
// held as x * 2^261 mod q, lazily reduced) and G1 in homogeneous projective coordinates under the complete formulas of Renes-Costello-Batina
// 2016 (a = 0, b3 = 9) — the representation the MSM kernels keep points and tables in.  Same names as the curve25519 header so the kernels are
// shared: pt29 keeps its four-coordinate layout (T unused), a table entry ("niels29") is the affine point (x, y), the mixed addition costs 11
// products (7 on the Edwards curve), the full addition 12 (9), a doubling 8.  Negating a table entry negates y.
// Bounds: "reduced" / "loose" as in mont29.cuh.  Every coordinate a function returns is reduced; products renormalise magnitudes to
// (-X, q + X) with X < 3 q, sums of up to three such values and their small multiples (3, 8, 9) stay far below the 2^30 limb bound.
#pragma once
#include <stdint.h>
#include "fq.cuh"
#include "mont29.cuh"

struct Bn254FqM {
  static LHD int32_t p(int k) { const int32_t P[9] = {410844487, 17064118, 477274959, 47522512, 361093496, 47923392, 10936641, 240920116, 3171406}; return P[k]; }
  static constexpr uint32_t PINV = 75916169u;
  static constexpr int32_t QC = 1420063842;
  static constexpr int32_t ONE_S_0 = 360500257, ONE_S_1 = 337389400, ONE_S_2 = 408039635, ONE_S_3 = 21759001, ONE_S_4 = 178483129, ONE_S_5 = 490881230, ONE_S_6 = 299191303,
                           ONE_S_7 = 86689704, ONE_S_8 = 903222;
  static constexpr int32_t K522_0 = 94088208, K522_1 = 219480995, K522_2 = 25171640, K522_3 = 279645352, K522_4 = 40052281, K522_5 = 46143135, K522_6 = 379321683,
                           K522_7 = 294034764, K522_8 = 2757030;
};
typedef m29<Bn254FqM> fe29;
#define FE29_MASK M29_MASK

LHD fe29 fe_zero() { return m29_zero<Bn254FqM>(); }
LHD fe29 fe_one() { fe29 r; r.v[0] = 360500257; r.v[1] = 337389400; r.v[2] = 408039635; r.v[3] = 21759001; r.v[4] = 178483129; r.v[5] = 490881230; r.v[6] = 299191303; r.v[7] = 86689704; r.v[8] = 903222; return r; }   // 2^261 mod q
LHD fe29 fe_add(const fe29& a, const fe29& b) { return m29_add(a, b); }
LHD fe29 fe_sub(const fe29& a, const fe29& b) { return m29_sub(a, b); }
LHD fe29 fe_neg(const fe29& a) { return m29_neg(a); }
LHD fe29 fe_dbl(const fe29& a) { return m29_add(a, a); }
LHD fe29 fe_weak(const fe29& a) { return m29_weak(a); }
LHD fe29 fe_small(const fe29& a, int32_t k) { return m29_mul_small(a, k); }   // value * k, any limbs with |.| <= 2^30 -> reduced
LHD fe29 fe_mul(const fe29& a, const fe29& b) { return m29_mul(a, b); }       // a loose, b reduced -> reduced
LHD fe29 fe_sqr(const fe29& a) { return m29_mul(a, a); }                       // a reduced
// a^(q-2), a reduced
LHD fe29 fe_inv_chain(const fe29& z) {
  const uint32_t e[8] = {0xd87cfd45u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};   // q - 2
  fe29 r = fe_one();
  for (int i = 253; i >= 0; i--) { r = fe_sqr(r); if ((e[i >> 5] >> (i & 31)) & 1u) r = fe_mul(r, z); }
  return r;
}

// ------------------------------------------------------------------ conversions (fq_t = ark's Montgomery words x * 2^256, canonical)
// fq_t words (x 2^256, canonical) -> x 2^261 mod q with reduced MAGNITUDE: one product with 2^266 (the bare shift-by-5 form would be up to 32 q,
// outside what m29_canonical accepts and larger than the group law's bounds assume)
LHD fe29 fe_from_fq(const fq_t& x) {
  const int32_t K5[9] = {322215073, 442336424, 171859116, 268585440, 314135016, 244503300, 348886451, 68918589, 360451};   // 2^266 mod q
  return fe_mul(m29_unpack_words<Bn254FqM>(x.v), m29_const<Bn254FqM>(K5));
}
LHD fq_t fe_to_fq(const fe29& a) {   // any reduced / loose a
  fe29 c256 = fe_zero(); c256.v[8] = 1 << 24;   // the integer 2^256: (x 2^261) 2^256 / 2^261 = x 2^256
  fq_t r; m29_pack_words(m29_canonical(fe_mul(a, c256)), r.v); return r;
}
LHD void fe_to_plain_words(const fe29& a, uint32_t* out) {   // the canonical integer x itself
  fe29 one = fe_zero(); one.v[0] = 1;
  m29_pack_words(m29_canonical(fe_mul(a, one)), out);
}

// ------------------------------------------------------------------ group law
struct pt29 { fe29 X, Y, T, Z; };                                  // (X : Y : Z); T unused (kept so that layouts match the Edwards build)
#ifndef MSM_NIELS_ALIGN
#define MSM_NIELS_ALIGN 128   // one table entry = one cache line (fe29.cuh says why)
#endif
struct alignas(MSM_NIELS_ALIGN) niels29 { fe29 x, y; int32_t pad[10]; };        // affine table entry: 72 bytes of payload in a 128-byte line

// The identity as a starting value of an accumulation loop.  On the device its limbs are made OPAQUE to the optimizer (an empty asm per limb): with the constants visible,
// hipcc (ROCm 7.2) derives value ranges for the loop-carried limbs of `B = pt_identity(); for (..) B = pt_madd(B, entry)` under which the 32 x 32 -> 64 products of
// fe_mul no longer match v_mad_i64_i32 and are expanded into 64 x 32 multiplies: 782 multiply-adds and 365 moves per mixed addition instead of 638 and 45 (1700 instructions
// instead of 1220; round 6, measured with hipcc -S on a four-line loop: tools/README.md "pt_identity").  Every commitment / opening kernel starts its sums this way.
LHD pt29 pt_identity() {
  pt29 p; p.X = fe_zero(); p.Y = fe_one(); p.T = fe_zero(); p.Z = fe_zero();
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LASSO_VISIBLE_IDENTITY)
  fe29* c = reinterpret_cast<fe29*>(&p);
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int k = 0; k < 9; k++) asm volatile("" : "+v"(c[i].v[k]));
#endif
  return p;
}
LHD fe29 fe_d2() { return fe_zero(); }   // the Edwards build passes 2d to pt_add; nothing to pass here
LHD fe29 fe_x3(const fe29& a) { return fe_weak(fe_add(fe_add(a, a), a)); }   // reduced -> reduced
// shared tail of the complete formulas:  X3 = t3 t1 - t4 y3,  Y3 = t1 z3 + y3 t0,  Z3 = z3 t4 + t0 t3
//   t0 = 3 X1X2, t1 = Y1Y2 - 9 Z1Z2, z3 = Y1Y2 + 9 Z1Z2, t3 = X1Y2 + X2Y1, t4 = Y1Z2 + Y2Z1, y3 = 9 (X1Z2 + X2Z1)
LHD pt29 pt_finish(const fe29& t0r, const fe29& t1l, const fe29& z3l, const fe29& t3l, const fe29& t4l, const fe29& y3r) {
  const fe29 t1r = fe_weak(t1l), z3r = fe_weak(z3l), t3r = fe_weak(t3l), t4r = fe_weak(t4l);
  pt29 r;
#ifdef LASSO_BN254_SPLIT_FINISH   // round 3's form (A/B): six products, six reductions
  r.X = fe_weak(fe_sub(fe_mul(t3l, t1r), fe_mul(t4l, y3r)));
  r.Y = fe_weak(fe_add(fe_mul(t1l, z3r), fe_mul(y3r, t0r)));
  r.Z = fe_weak(fe_add(fe_mul(z3l, t4r), fe_mul(t0r, t3r)));
#else                             // each coordinate is a sum of two products: accumulated in the 64-bit columns, ONE Montgomery reduction per coordinate (3 x 81 multiply-adds fewer)
  r.X = m29_mul2(t3r, t1r, t4r, y3r, -1);
  r.Y = m29_mul2(t1r, z3r, y3r, t0r, 1);
  r.Z = m29_mul2(z3r, t4r, t0r, t3r, 1);
#endif
  r.T = fe_zero();
  return r;
}
// mixed addition p + (x, y): 11 products
LHD pt29 pt_madd(const pt29& p, const niels29& n) {
  const fe29 t0 = fe_mul(p.X, n.x), t1 = fe_mul(p.Y, n.y);
  const fe29 t3 = fe_sub(fe_sub(fe_mul(fe_add(p.X, p.Y), fe_weak(fe_add(n.x, n.y))), t0), t1);
  const fe29 t4 = fe_add(fe_mul(p.Z, n.y), p.Y);
  const fe29 y3 = fe_add(fe_mul(p.Z, n.x), p.X);
  const fe29 t2 = fe_small(p.Z, 9);
  return pt_finish(fe_x3(t0), fe_sub(t1, t2), fe_add(t1, t2), t3, t4, fe_small(y3, 9));
}
// full addition: 12 products.  The second argument of the Edwards build (2d) is ignored.
LHD pt29 pt_add(const pt29& p, const pt29& q, const fe29&) {
  const fe29 t0 = fe_mul(p.X, q.X), t1 = fe_mul(p.Y, q.Y), t2 = fe_mul(p.Z, q.Z);
  const fe29 t3 = fe_sub(fe_sub(fe_mul(fe_add(p.X, p.Y), fe_weak(fe_add(q.X, q.Y))), t0), t1);
  const fe29 t4 = fe_sub(fe_sub(fe_mul(fe_add(p.Y, p.Z), fe_weak(fe_add(q.Y, q.Z))), t1), t2);
  const fe29 y3 = fe_sub(fe_sub(fe_mul(fe_add(p.X, p.Z), fe_weak(fe_add(q.X, q.Z))), t0), t2);
  const fe29 t29 = fe_small(t2, 9);
  return pt_finish(fe_x3(t0), fe_sub(t1, t29), fe_add(t1, t29), t3, t4, fe_small(y3, 9));
}
LHD pt29 pt_dbl(const pt29& p) {
  const fe29 t0 = fe_sqr(p.Y), z8 = fe_small(t0, 8), t1 = fe_mul(p.Y, p.Z), t2 = fe_small(fe_sqr(p.Z), 9);
  const fe29 x3 = fe_mul(t2, z8), y3 = fe_weak(fe_add(t0, t2));
  const fe29 u = fe_sub(t0, fe_x3(t2));                       // Y^2 - 27 Z^2
  pt29 r;
  r.Y = fe_weak(fe_add(fe_mul(u, y3), x3));
  r.X = fe_mul(u, fe_weak(fe_dbl(fe_mul(p.X, p.Y))));
  r.Z = fe_mul(t1, z8);
  r.T = fe_zero();
  return r;
}
// ---- the full addition split over SIX lanes (msm_coop_tree): its twelve products form two dependent layers of six, so lane role c of a
// sextet computes product c of each layer and a tree level costs two product times instead of twelve.  Roles are selected by data (coordinate
// indices, 0/1 masks, per-limb selects), never by branch: the six lanes share a wave.
//   layer 1: m0 = X1X2, m1 = Y1Y2, m2 = Z1Z2, m3 = (X1+Y1)(X2+Y2), m4 = (Y1+Z1)(Y2+Z2), m5 = (X1+Z1)(X2+Z2)
//   layer 2: p0 = t3 t1, p1 = t4 y3, p2 = t1 z3, p3 = y3 t0, p4 = z3 t4, p5 = t0 t3   (names as in pt_finish)
//   output:  X3 = p0 - p1, Y3 = p2 + p3, Z3 = p4 + p5
LHD fe29 pt_coop_layer1(const pt29& p, const pt29& q, uint32_t c) {
  const fe29* pc = reinterpret_cast<const fe29*>(&p);   // {X, Y, T, Z} = coordinates 0, 1, 2, 3
  const fe29* qc = reinterpret_cast<const fe29*>(&q);
  const uint32_t i0 = c == 0 ? 0u : (c == 1 ? 1u : (c == 2 ? 3u : (c == 3 ? 0u : (c == 4 ? 1u : 0u))));
  const uint32_t i1 = c == 3 ? 1u : 3u;                  // second summand (roles 3..5): Y, Z, Z
  const int32_t m = c >= 3 ? 1 : 0;
  const fe29 p0 = pc[i0], p1 = pc[i1], q0 = qc[i0], q1 = qc[i1];
  fe29 a, b;
#pragma unroll
  for (int k = 0; k < 9; k++) { a.v[k] = p0.v[k] + m * p1.v[k]; b.v[k] = q0.v[k] + m * q1.v[k]; }
  return fe_mul(a, fe_weak(b));
}
LHD fe29 pt_coop_layer2(const fe29* m, uint32_t c) {   // m[0..5] = the six layer-1 products
  const fe29 t3 = fe_sub(fe_sub(m[3], m[0]), m[1]), t4 = fe_sub(fe_sub(m[4], m[1]), m[2]), y3 = fe_small(fe_sub(fe_sub(m[5], m[0]), m[2]), 9);
  const fe29 t0 = fe_x3(m[0]), t29 = fe_small(m[2], 9), t1 = fe_sub(m[1], t29), z3 = fe_add(m[1], t29);
  fe29 u, w;   // u loose, w weakened below
#pragma unroll
  for (int k = 0; k < 9; k++) {
    u.v[k] = c == 0 ? t3.v[k] : (c == 1 ? t4.v[k] : (c == 2 ? t1.v[k] : (c == 3 ? y3.v[k] : (c == 4 ? z3.v[k] : t0.v[k]))));
    w.v[k] = c == 0 ? t1.v[k] : (c == 1 ? y3.v[k] : (c == 2 ? z3.v[k] : (c == 3 ? t0.v[k] : (c == 4 ? t4.v[k] : t3.v[k]))));
  }
  return fe_mul(u, fe_weak(w));
}
// Round 4: the linear step between the layers as its OWN lane step.  pt_coop_layer2 above has every lane form all six combinations (two 9-limb multiplications by 9, a
// tripling, six subtractions) and pick two of them through 90 per-limb selects; here lane role c forms ONE combination f_c from at most three of the products — read at
// role-dependent addresses, so nothing is selected per limb — and the second layer's product reads its two operands the same way:
//   f0 = t0 = 3 m0,  f1 = t1 = m1 - 9 m2,  f2 = z3 = m1 + 9 m2,  f3 = t3 = m3 - m0 - m1,  f4 = t4 = m4 - m1 - m2,  f5 = y3 = 9 (m5 - m0 - m2)
//   p0 = f3 f1, p1 = f4 f5, p2 = f1 f2, p3 = f5 f0, p4 = f2 f4, p5 = f0 f3      (the same integers as pt_coop_layer2 / pt_finish: same limbs out)
// m: the six layer-1 products of the group (reduced).  Result reduced (limbs 0..7 in [0, 2^29), limb 8 small and signed), |value| < 27 (q + X).
LHD fe29 pt_coop_form(const fe29* m, uint32_t c) {
  // (ix, iy, iz) and the coefficients (a, b, cz) of f_c = a m[ix] + b m[iy] + cz m[iz]
  const uint32_t ix = c == 0 ? 0u : (c <= 2 ? 1u : c);
  const uint32_t iy = c == 0 ? 0u : (c <= 2 ? 2u : (c == 3 ? 0u : (c == 4 ? 1u : 0u)));
  const uint32_t iz = c <= 2 ? 0u : (c == 3 ? 1u : 2u);
  const int32_t a = c == 0 ? 3 : (c == 5 ? 9 : 1);
  const int32_t b = c == 0 ? 0 : (c == 1 ? -9 : (c == 2 ? 9 : (c == 5 ? -9 : -1)));
  const int32_t cz = c <= 2 ? 0 : (c == 5 ? -9 : -1);
  const fe29 X = m[ix], Y = m[iy], Z = m[iz];
  fe29 r; int64_t carry = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { const int64_t x = (int64_t)a * X.v[k] + (int64_t)b * Y.v[k] + (int64_t)cz * Z.v[k] + carry; r.v[k] = (int32_t)x & FE29_MASK; carry = x >> 29; }
  r.v[8] = (int32_t)((int64_t)a * X.v[8] + (int64_t)b * Y.v[8] + (int64_t)cz * Z.v[8] + carry);
  return r;
}
LHD fe29 pt_coop_prod2(const fe29* f, uint32_t c) {   // f[0..5] = the six forms; product c of the second layer
  const uint32_t iu = c == 0 ? 3u : (c == 1 ? 4u : (c == 2 ? 1u : (c == 3 ? 5u : (c == 4 ? 2u : 0u))));
  const uint32_t iw = c == 0 ? 1u : (c == 1 ? 5u : (c == 2 ? 2u : (c == 3 ? 0u : (c == 4 ? 4u : 3u))));
  return fe_mul(f[iu], f[iw]);
}
LHD fe29 pt_coop_out(const fe29& pe, const fe29& po, uint32_t j) {   // coordinate j (0: X, 1: Y, 2: Z) from the products 2j and 2j + 1
  fe29 r;
  const int32_t sg = j == 0 ? -1 : 1;
#pragma unroll
  for (int k = 0; k < 9; k++) r.v[k] = pe.v[k] + sg * po.v[k];
  return fe_weak(r);
}
LHD niels29 niels_from_xy29(const fe29& x, const fe29& y, const fe29&) {   // x, y reduced (canonicalised here so that table entries are digits)
  niels29 n; n.x = m29_canonical(x); n.y = m29_canonical(y);
#pragma unroll
  for (int i = 0; i < 10; i++) n.pad[i] = 0;
  return n;
}
LHD niels29 niels_from_affine(const fq_t& x, const fq_t& y) { return niels_from_xy29(fe_from_fq(x), fe_from_fq(y), fe_zero()); }
LHD niels29 niels_cond_neg(const niels29& n, bool neg) {
  niels29 r = n;
#pragma unroll
  for (int k = 0; k < 9; k++) r.y.v[k] = neg ? -n.y.v[k] : n.y.v[k];
  return r;
}
LHD pt29 pt_from_ed(const ed_point& e) { pt29 p; p.X = fe_from_fq(e.X); p.Y = fe_from_fq(e.Y); p.T = fe_zero(); p.Z = fe_from_fq(e.Z); return p; }
LHD ed_point pt_to_ed(const pt29& p) { ed_point e; e.X = fe_to_fq(p.X); e.Y = fe_to_fq(p.Y); e.T = fq_zero(); e.Z = fe_to_fq(p.Z); return e; }
LHD ed_point pt_to_abi(const pt29& p) { return pt_to_ed(p); }
LHD fq_t pt_coord_abi(const pt29& p, uint32_t c) { return c == 2u ? fq_zero() : fe_to_fq(reinterpret_cast<const fe29*>(&p)[c]); }   // coordinate c of pt_to_abi(p): {X, Y, 0, Z}   // fq_t already is ark's Montgomery form
// ark-serialize's compressed short-Weierstrass point (ark-ec SWFlags; serialize_compressed of the normalised point, utils/transcript.rs:47-51):
// canonical x, little endian; bit 7 of the last byte set iff y > -y as canonical integers; bit 6 = point at infinity (x = 0).
LHD void pt_compress(const pt29& p, uint32_t* out) {
  uint32_t zw[8]; fe_to_plain_words(p.Z, zw);
  uint32_t any = 0; for (int i = 0; i < 8; i++) any |= zw[i];
  if (!any) { for (int i = 0; i < 8; i++) out[i] = 0; out[7] = 0x40000000u; return; }
  const fe29 zi = fe_inv_chain(fe_weak(p.Z));
  const fe29 y = fe_mul(p.Y, zi);
  uint32_t yw[8], nyw[8];
  fe_to_plain_words(fe_mul(p.X, zi), out); fe_to_plain_words(y, yw); fe_to_plain_words(fe_neg(y), nyw);
  bool neg = false;
  for (int i = 7; i >= 0; i--) if (yw[i] != nyw[i]) { neg = yw[i] > nyw[i]; break; }
  if (neg) out[7] |= 0x80000000u;
}

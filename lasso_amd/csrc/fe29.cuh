// Fq = 2^255 - 19 in nine signed 29-bit limbs (value = sum v[k] * 2^(29k), lazily reduced mod p), the representation
// the MSM kernels keep points and tables in.  Why: on gfx950 `v_mad_i64_i32` issues at ~5 cycles per wave and accumulates a
// 64-bit column in place, so a schoolbook product in a radix with headroom needs NO carry handling between partial
// products — the 8x32-bit version in fq.cuh spends three quarters of its instructions on `v_mov`/`v_lshl_add_u64` moving
// carries around (profiles/r01_microbench_initial.txt).  2^261 = 64 * 2^255 = 1216 (mod p), so high columns fold back with a
// small constant.  Signed limbs make subtraction a plain limb-wise `v_sub`.
//
// Bounds (|.| per limb):  "reduced"  <= 2^29 + 2^15   (outputs of fe_mul / fe_weak)
//                         "loose"    <= 2^30 + 2^16   (one add/sub of reduced values)
// fe_mul(a, b) requires a loose, b reduced: 9 products of <= 2^59.01 stay below 2^63.
#pragma once
#ifdef LASSO_BN254
#include "bn254_fe29.cuh"   // the same interface over ark-bn254's Fq and G1
#else
#include <stdint.h>
#include "fq.cuh"

struct fe29 { int32_t v[9]; };
#define FE29_MASK 0x1fffffff

LHD fe29 fe_zero() { fe29 r; for (int i = 0; i < 9; i++) r.v[i] = 0; return r; }
LHD fe29 fe_one() { fe29 r = fe_zero(); r.v[0] = 1; return r; }
LHD fe29 fe_add(const fe29& a, const fe29& b) { fe29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
LHD fe29 fe_sub(const fe29& a, const fe29& b) { fe29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] - b.v[i]; return r; }
LHD fe29 fe_neg(const fe29& a) { fe29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = -a.v[i]; return r; }
LHD fe29 fe_dbl(const fe29& a) { fe29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] * 2; return r; }

// carry pass on 32-bit limbs: |in| < 2^31  ->  reduced
LHD fe29 fe_weak(const fe29& a) {
  fe29 r; int32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) { int32_t x = a.v[i] + c; c = x >> 29; r.v[i] = x & FE29_MASK; }
  r.v[0] += c * 1216;   // |c| <= 4: limb 0 leaves [0, 2^29) by at most 4864
  return r;
}

// a loose, b reduced -> reduced.  81 + 9 multiply-accumulates.
LHD fe29 fe_mul(const fe29& a, const fe29& b) {
  int64_t h[18];
#pragma unroll
  for (int k = 0; k < 18; k++) h[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++) h[i + j] += (int64_t)a.v[i] * b.v[j];
  // normalise columns 9..16 to 29 bits (carry into 17), then fold 2^(29(k+9)) = 1216 * 2^(29k)
  int32_t hi[9];
#pragma unroll
  for (int k = 9; k < 17; k++) { int64_t c = h[k] >> 29; hi[k - 9] = (int32_t)h[k] & FE29_MASK; h[k + 1] += c; }
  hi[8] = (int32_t)h[17];   // |h17| < 2^31
#pragma unroll
  for (int k = 0; k < 9; k++) h[k] += (int64_t)hi[k] * 1216;
  fe29 r; int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { int64_t x = h[k] + c; c = x >> 29; r.v[k] = (int32_t)x & FE29_MASK; }
  // carry out of limb 8 (|c| < 2^34) has weight 2^261 = 1216
  int64_t x0 = (int64_t)r.v[0] + c * 1216;
  r.v[0] = (int32_t)x0 & FE29_MASK;
  r.v[1] += (int32_t)(x0 >> 29);   // < 2^16 in magnitude
  return r;
}
LHD fe29 fe_sqr(const fe29& a) { return fe_mul(a, a); }   // a must be reduced
// a reduced, 0 < k < 2^18 -> reduced: nine multiply-adds and one carry pass (a third of a product)
LHD fe29 fe_mul_small(const fe29& a, int32_t k) {
  fe29 r; int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) { const int64_t x = (int64_t)a.v[i] * k + c; c = x >> 29; r.v[i] = (int32_t)x & FE29_MASK; }
  const int64_t x0 = (int64_t)r.v[0] + c * 1216;   // the carry out of limb 8 (< 2^19) has weight 2^261 = 1216
  r.v[0] = (int32_t)x0 & FE29_MASK;
  r.v[1] += (int32_t)(x0 >> 29);
  return r;
}

// a^(p-2) by the curve25519 addition chain (254 squarings + 11 multiplications); a reduced
LHD fe29 fe_inv_chain(const fe29& z) {
  fe29 z2 = fe_sqr(z);
  fe29 z8 = fe_sqr(fe_sqr(z2));
  fe29 z9 = fe_mul(z8, z);
  fe29 z11 = fe_mul(z9, z2);
  fe29 z22 = fe_sqr(z11);
  fe29 z_5_0 = fe_mul(z22, z9);
  fe29 t = z_5_0; for (int i = 0; i < 5; i++) t = fe_sqr(t);
  fe29 z_10_0 = fe_mul(t, z_5_0);
  t = z_10_0; for (int i = 0; i < 10; i++) t = fe_sqr(t);
  fe29 z_20_0 = fe_mul(t, z_10_0);
  t = z_20_0; for (int i = 0; i < 20; i++) t = fe_sqr(t);
  t = fe_mul(t, z_20_0);
  for (int i = 0; i < 10; i++) t = fe_sqr(t);
  fe29 z_50_0 = fe_mul(t, z_10_0);
  t = z_50_0; for (int i = 0; i < 50; i++) t = fe_sqr(t);
  fe29 z_100_0 = fe_mul(t, z_50_0);
  t = z_100_0; for (int i = 0; i < 100; i++) t = fe_sqr(t);
  t = fe_mul(t, z_100_0);
  for (int i = 0; i < 50; i++) t = fe_sqr(t);
  t = fe_mul(t, z_50_0);
  for (int i = 0; i < 5; i++) t = fe_sqr(t);
  return fe_mul(t, z11);
}

// ------------------------------------------------------------------ conversions (table build / result hand-back only)
LHD fe29 fe_from_fq(const fq_t& x) {   // any lazy fq_t (< 2^256)
  fq_t c = fq_canonical(x);
  fe29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    int bit = 29 * k, w = bit >> 5, s = bit & 31;
    uint64_t two = (uint64_t)c.v[w] | ((w + 1 < 8) ? ((uint64_t)c.v[w + 1] << 32) : 0);
    r.v[k] = (int32_t)((two >> s) & FE29_MASK);
  }
  return r;
}
LHD fq_t fe_to_fq(const fe29& a) {   // loose input ok; returns a lazy fq_t congruent to a
  fe29 t = fe_weak(a);               // limbs 1..8 in [0, 2^29), limb 0 in [-4864, 2^29 + 4864]
  // V = value(t) + (2^261 - 1216) = value(t) + 64p  >= 0: limb 0 -= 1216, limb 9 = 1, then an exact carry pass
  int64_t l[10];
#pragma unroll
  for (int k = 0; k < 9; k++) l[k] = t.v[k];
  l[0] -= 1216; l[9] = 1;
  int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 10; k++) { int64_t x = l[k] + c; c = x >> 29; l[k] = x & FE29_MASK; }
  // pack the ten 29-bit limbs (V < 2^262) into nine 32-bit words
  uint32_t w[10];
#pragma unroll
  for (int i = 0; i < 10; i++) w[i] = 0;
#pragma unroll
  for (int k = 0; k < 10; k++) {
    const int bit = 29 * k, wi = bit >> 5, sh = bit & 31;
    const uint64_t v = (uint64_t)l[k] << sh;
    w[wi] |= (uint32_t)v;
    if (wi + 1 < 10) w[wi + 1] |= (uint32_t)(v >> 32);
  }
  // V = lo256 + w[8] * 2^256 and 2^256 = 38 (mod p); w[8] < 2^6, w[9] == 0
  fq_t lo; for (int i = 0; i < 8; i++) lo.v[i] = w[i];
  fq_t hi = fq_zero(); hi.v[0] = 38u * w[8];
  return fq_add(lo, hi);
}

// ------------------------------------------------------------------ group law on fe29 coordinates
struct pt29 { fe29 X, Y, T, Z; };            // extended coordinates, all reduced
// One table entry = ONE 128-byte cache line (round 6).  The payload is 108 bytes (7 x dwordx4 loads); at the 112-byte stride of rounds 1-5 seven entries in eight straddled two
// lines, so every mixed addition pulled 256 bytes through the fabric: the row-parallel commitment ran at 18 G additions/s x 256 B = 4.6 TB/s — at the memory system's rate, not
// the VALUs' (tools/madd_bench.hip section C against section A's 31 G/s).  -DMSM_NIELS_ALIGN=16 restores the packed layout (A/B).
#ifndef MSM_NIELS_ALIGN
#define MSM_NIELS_ALIGN 128
#endif
struct alignas(MSM_NIELS_ALIGN) niels29 { fe29 ypx, ymx, t2d; int32_t pad; };   // 108 bytes of payload: 7 x dwordx4

// The identity as a starting value of an accumulation loop.  On the device its limbs are made OPAQUE to the optimizer (an empty asm per limb): with the constants visible,
// hipcc (ROCm 7.2) derives value ranges for the loop-carried limbs of `B = pt_identity(); for (..) B = pt_madd(B, entry)` under which the 32 x 32 -> 64 products of
// fe_mul no longer match v_mad_i64_i32 and are expanded into 64 x 32 multiplies: 782 multiply-adds and 365 moves per mixed addition instead of 638 and 45 (1700 instructions
// instead of 1220; round 6, measured with hipcc -S on a four-line loop: tools/README.md "pt_identity").  Every commitment / opening kernel starts its sums this way.
LHD pt29 pt_identity() {
  pt29 p; p.X = fe_zero(); p.Y = fe_one(); p.T = fe_zero(); p.Z = fe_one();
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LASSO_VISIBLE_IDENTITY)
  fe29* c = reinterpret_cast<fe29*>(&p);
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int k = 0; k < 9; k++) asm volatile("" : "+v"(c[i].v[k]));
#endif
  return p;
}
LHD fe29 fe_d2() {   // 2d mod p
  fe29 r; r.v[0] = 112390489; r.v[1] = 515169441; r.v[2] = 15488442; r.v[3] = 2700549; r.v[4] = 487784462; r.v[5] = 7960441; r.v[6] = 329016890; r.v[7] = 462085119; r.v[8] = 2361049; return r;
}
// mixed add (7 multiplications): p + n
LHD pt29 pt_madd(const pt29& p, const niels29& n) {
  fe29 A = fe_mul(fe_sub(p.Y, p.X), n.ymx);
  fe29 B = fe_mul(fe_add(p.Y, p.X), n.ypx);
  fe29 C = fe_mul(p.T, n.t2d);
  fe29 D = fe_dbl(p.Z);
  fe29 E = fe_sub(B, A), F = fe_weak(fe_sub(D, C)), G = fe_weak(fe_add(D, C)), H = fe_weak(fe_add(B, A));
  pt29 r; r.X = fe_mul(E, F); r.Y = fe_mul(H, G); r.T = fe_mul(E, H); r.Z = fe_mul(F, G); return r;
}
// full add (9 multiplications), unified / complete
LHD pt29 pt_add(const pt29& p, const pt29& q, const fe29& d2) {
  fe29 A = fe_mul(fe_sub(p.Y, p.X), fe_weak(fe_sub(q.Y, q.X)));
  fe29 B = fe_mul(fe_add(p.Y, p.X), fe_weak(fe_add(q.Y, q.X)));
  fe29 C = fe_mul(fe_mul(p.T, d2), q.T);
  fe29 D = fe_dbl(fe_mul(p.Z, q.Z));
  fe29 E = fe_sub(B, A), F = fe_weak(fe_sub(D, C)), G = fe_weak(fe_add(D, C)), H = fe_weak(fe_add(B, A));
  pt29 r; r.X = fe_mul(E, F); r.Y = fe_mul(H, G); r.T = fe_mul(E, H); r.Z = fe_mul(F, G); return r;
}
// The curve constant is d = -121665 / 121666 (RFC 7748 / 8032: edwards25519; ark-curve25519's COEFF_D).  Scaling all four of A, B, C = 2d T1 T2, D = 2 Z1 Z2 by
// lambda = 121666 scales E, F, G, H by lambda and hence X3 = E F, Y3 = G H, T3 = E H, Z3 = F G all by lambda^2: the same point, a consistent extended representation.  And
//   lambda A = 121666 A,   lambda B = 121666 B,   lambda C = -243330 T1 T2,   lambda D = 243332 Z1 Z2
// are products with SMALL constants (fe_mul_small: a third of a product) — no full-width multiplication by 2d is left.  For one lane that is no gain (four small products
// for one large), but in the four-lane tree (msm_coop_tree) every role has exactly one of them: a level is two product times deep instead of three.
#define ED_K_AB 121666  // lambda
#define ED_K_C 243330   // 2 * 121665:  Cs = ED_K_C T1 T2 = -lambda C
#define ED_K_D 243332   // 2 * 121666:  Ds = ED_K_D Z1 Z2 =  lambda D
// The addition shared by four lanes (msm_coop_tree): lane role c computes one of lambda A, lambda B, Cs, Ds (stage 1: ONE product + the small constant, selected by data so
// that the instruction stream is uniform), the four exchange them, and role c computes coordinate c of the sum (stage 2: one product).
// Round 6 (profiles/r06_bullet_phase_curve25519_before.txt: a tree level cost 1.28-1.36 us = 551 VALU instructions, of which the two products are 284): the small constant now rides
// on the SECOND operand before the product — fe_mul_small(b, k) is the carry pass fe_weak(b) was, with a multiply-add per limb in it — so the separate pass after the
// product is gone (-30 instructions), and stage 2 below neither selects per limb nor weakens (-114).  Same integers mod p as before; other representatives, same points.
LHD fe29 pt_coop4_stage1(const pt29& p, const pt29& q, uint32_t c) {
  const fe29* pc = reinterpret_cast<const fe29*>(&p); const fe29* qc = reinterpret_cast<const fe29*>(&q);   // {X, Y, T, Z}
  const uint32_t ci = c < 2 ? 1u : c;                               // roles 0 / 1 use Y -/+ X, role 2 T, role 3 Z
  const int32_t sg = c == 0 ? -1 : (c == 1 ? 1 : 0);
  fe29 a, b;
#pragma unroll
  for (int k = 0; k < 9; k++) { a.v[k] = pc[ci].v[k] + sg * pc[0].v[k]; b.v[k] = qc[ci].v[k] + sg * qc[0].v[k]; }
  // a: a sum or difference of two reduced values = loose.  b: likewise (|limb| < 2^30 + 2^17), times k < 2^18 stays far inside 64 bits; fe_mul_small's carry pass leaves it reduced.
  return fe_mul(a, fe_mul_small(b, c == 2 ? ED_K_C : (c == 3 ? ED_K_D : ED_K_AB)));
}
// Stage 2 by ADDRESS instead of by select: P = {lambda A, lambda B, Cs, Ds} as the quad left them (reduced: outputs of fe_mul).  Every coordinate of the sum is (a sum) x (a difference):
//   X3 = E F = (B - A)(Ds + Cs),  Y3 = G H = (Ds - Cs)(B + A),  T3 = E H = (B - A)(B + A),  Z3 = F G = (Ds + Cs)(Ds - Cs)
// so role c reads the pair its sum is made of and the pair its difference is made of — sum of two reduced values = loose (fe_mul's first operand), difference of two
// reduced values = reduced in magnitude (its second) — and multiplies: 18 additions and one product, no per-limb selects, no carry pass.
LHD fe29 pt_coop4_stage2p(const fe29* P, uint32_t c) {
  const uint32_t sl = (c == 0u || c == 3u) ? 2u : 0u, sr = (c == 1u || c == 3u) ? 2u : 0u;   // the sum is F (from Cs, Ds) for X3 and Z3, H (from A, B) otherwise; the difference is G for Y3 and Z3, E otherwise
  const fe29 l0 = P[sl], l1 = P[sl + 1], r0 = P[sr], r1 = P[sr + 1];
  fe29 u, w;
#pragma unroll
  for (int k = 0; k < 9; k++) { u.v[k] = l1.v[k] + l0.v[k]; w.v[k] = r1.v[k] - r0.v[k]; }
  return fe_mul(u, w);
}
LHD fe29 pt_coop4_stage2(const fe29& A, const fe29& Bv, const fe29& Cs, const fe29& Ds, uint32_t c) {
  // role 0: E*F, role 1: H*G, role 2: E*H, role 3: F*G   with (all times lambda) E = B - A, F = D - C = Ds + Cs, G = D + C = Ds - Cs, H = B + A
  fe29 u, w;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int32_t E = Bv.v[k] - A.v[k], F = Ds.v[k] + Cs.v[k], G = Ds.v[k] - Cs.v[k], H = Bv.v[k] + A.v[k];
    u.v[k] = c == 0 ? E : (c == 1 ? H : (c == 2 ? E : F));
    w.v[k] = c == 0 ? F : (c == 1 ? G : (c == 2 ? H : G));
  }
  return fe_mul(fe_weak(u), fe_weak(w));
}
// The MIXED addition shared by four lanes (round 6: the last, partly filled pass of a latency-shaped MSM's accumulation — msm_coop_leftover — costs a tree level's time
// instead of a lone lane's seven products in a row): role c computes P[c] of {A, B, -C, D} = {(Y - X) ymx, (Y + X) ypx, T (-t2d), Z 2} — one product each, D's by the constant
// 2 so that the instruction stream stays uniform and every P is a reduced product output — and pt_coop4_stage2p finishes: E = B - A, F = D - C = P3 + P2, G = D + C = P3 - P2,
// H = B + A, exactly pt_madd's X3 = E F, Y3 = H G, T3 = E H, Z3 = F G.  n: {ypx, ymx, t2d} as three reduced values (a table entry, sign applied).
LHD fe29 pt_coop4_madd_stage1(const pt29& p, const fe29* n, uint32_t c) {
  const fe29* pc = reinterpret_cast<const fe29*>(&p);   // {X, Y, T, Z}
  const uint32_t ci = c < 2 ? 1u : c;
  const int32_t sg = c == 0 ? -1 : (c == 1 ? 1 : 0), sb = c == 2 ? -1 : 1;
  const fe29 nb = n[c == 0 ? 1u : (c == 1 ? 0u : 2u)];   // role 0: ymx, role 1: ypx, role 2: t2d (negated below); role 3 reads t2d too and replaces it by the constant
  fe29 a, b;
#pragma unroll
  for (int k = 0; k < 9; k++) { a.v[k] = pc[ci].v[k] + sg * pc[0].v[k]; b.v[k] = c == 3 ? (k == 0 ? 2 : 0) : sb * nb.v[k]; }
  return fe_mul(a, b);
}
LHD pt29 pt_dbl(const pt29& p) {
  fe29 A = fe_sqr(p.X), B = fe_sqr(p.Y), C = fe_dbl(fe_sqr(p.Z));
  fe29 E = fe_dbl(fe_mul(p.X, p.Y));                       // (X+Y)^2 - A - B = 2XY
  fe29 G = fe_sub(B, A), F = fe_weak(fe_sub(G, C)), H = fe_weak(fe_neg(fe_add(A, B)));   // D = -A
  pt29 r; r.X = fe_mul(E, F); r.Y = fe_mul(G, H); r.T = fe_mul(E, H); r.Z = fe_mul(G, F); return r;
}
LHD niels29 niels_from_affine(const fq_t& x, const fq_t& y) {
  niels29 n; n.ypx = fe_from_fq(fq_add(y, x)); n.ymx = fe_from_fq(fq_sub(y, x)); n.t2d = fe_from_fq(fq_mul(fq_mul(x, y), fq_d2())); n.pad = 0; return n;
}
LHD pt29 pt_from_ed(const ed_point& e) { pt29 p; p.X = fe_from_fq(e.X); p.Y = fe_from_fq(e.Y); p.T = fe_from_fq(e.T); p.Z = fe_from_fq(e.Z); return p; }
// ark-serialize's compressed twisted-Edwards point (serialize_compressed of the normalised point, utils/transcript.rs:47-51): canonical y,
// little endian, with bit 7 of the last byte set iff x is "negative", i.e. x > -x as canonical integers.  out = 8 little-endian words.
LHD void pt_compress(const pt29& p, uint32_t* out) {
  const fe29 zi = fe_inv_chain(p.Z);
  const fq_t x = fq_canonical(fe_to_fq(fe_mul(p.X, zi))), y = fq_canonical(fe_to_fq(fe_mul(p.Y, zi))), nx = fq_canonical(fq_neg(x));
  bool neg = false;
  for (int i = 7; i >= 0; i--) if (x.v[i] != nx.v[i]) { neg = x.v[i] > nx.v[i]; break; }
  for (int i = 0; i < 8; i++) out[i] = y.v[i];
  if (neg) out[7] |= 0x80000000u;
}
LHD ed_point pt_to_ed(const pt29& p) { ed_point e; e.X = fe_to_fq(p.X); e.Y = fe_to_fq(p.Y); e.T = fe_to_fq(p.T); e.Z = fe_to_fq(p.Z); return e; }

// helpers the kernels share with the BN254 build (bn254_fe29.cuh)
LHD niels29 niels_from_xy29(const fe29& x, const fe29& y, const fe29& d2) { niels29 e; e.ypx = fe_weak(fe_add(y, x)); e.ymx = fe_weak(fe_sub(y, x)); e.t2d = fe_mul(fe_mul(x, y), d2); e.pad = 0; return e; }
LHD niels29 niels_cond_neg(const niels29& n, bool neg) {   // -(x, y) = (-x, y): swap y+x and y-x, negate 2dxy
  niels29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) { r.ypx.v[k] = neg ? n.ymx.v[k] : n.ypx.v[k]; r.ymx.v[k] = neg ? n.ypx.v[k] : n.ymx.v[k]; r.t2d.v[k] = neg ? -n.t2d.v[k] : n.t2d.v[k]; }
  r.pad = 0;
  return r;
}
// coordinate c (0 X, 1 Y, 2 T, 3 Z) of pt_to_abi(p): four lanes convert one point side by side (msm_direct_finish's tagged hand-over)
LHD fq_t pt_coord_abi(const pt29& p, uint32_t c) { return fq_to_mont(fe_to_fq(reinterpret_cast<const fe29*>(&p)[c])); }
LHD ed_point pt_to_abi(const pt29& p) { ed_point e = pt_to_ed(p), o; o.X = fq_to_mont(e.X); o.Y = fq_to_mont(e.Y); o.T = fq_to_mont(e.T); o.Z = fq_to_mont(e.Z); return o; }   // ark's Montgomery limbs
#endif  // LASSO_BN254

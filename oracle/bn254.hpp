// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// BN254 G1 (`ark_bn254::G1Projective`, the group BASELINE.json's configs[1] names): short Weierstrass y^2 = x^3 + 3 over
// Fq = 21888242871839275222246405745257275088696311157297823662689037894645226208583, prime order
// r  = 21888242871839275222246405745257275088548364400416034343698204186575808495617 (cofactor 1), generator (1, 2).
// The reference itself never instantiates it (its harness and tests use curve25519: src/benches/bench.rs:6, src/e2e_test.rs:1); the prover
// is generic over `G: CurveGroup` (surge.rs:108-118), so this is the same path with another ark-ec model.  Group law, `normalize_batch`,
// (de)serialisation and `G::rand` live in ark-ec / ark-serialize ^0.4.2 (Cargo.toml:40,43; NOT in /root/reference) and are restated from their
// published algorithms.  Built with -DORC_BN254; exposes the same Point interface as ed25519.hpp.
// Coordinates: Jacobian (X:Y:Z), x = X/Z^2, y = Y/Z^3, identity Z = 0 (ark-ec `short_weierstrass::Projective`).
// Parity status: field and curve constants pinned by the EIP-196 known answer 2*(1,2) and by r*(1,2) = identity (tests/test_oracle_bn254.py);
// group law pinned against Python big-int affine formulas; serialisation flags and G::rand restated from memory of the crates, UNPINNED.
#pragma once
#include "ff.hpp"
#include <vector>

namespace orc {

struct Point {
  Fq X, Y, Z;
  static Fq b() { static const Fq v = Fq::from_u64(3); return v; }
  static Point identity() { Point p; p.X = Fq::one(); p.Y = Fq::one(); p.Z = Fq::zero(); return p; }
  static Point from_affine(const Fq& x, const Fq& y) { Point p; p.X = x; p.Y = y; p.Z = Fq::one(); return p; }
  static Point generator() { return from_affine(Fq::from_u64(1), Fq::from_u64(2)); }
  bool is_identity() const { return Z.is_zero(); }
  Point dbl() const {  // dbl-2009-l (a = 0)
    if (is_identity()) return *this;
    Fq A = X.square(), B = Y.square(), C = B.square();
    Fq D = ((X + B).square() - A - C).dbl();
    Fq E = A + A + A, F = E.square();
    Point r;
    r.X = F - D.dbl();
    r.Y = E * (D - r.X) - C.dbl().dbl().dbl();
    r.Z = (Y * Z).dbl();
    return r;
  }
  Point operator+(const Point& o) const {  // add-2007-bl with the exceptional cases handled
    if (is_identity()) return o;
    if (o.is_identity()) return *this;
    Fq Z1Z1 = Z.square(), Z2Z2 = o.Z.square();
    Fq U1 = X * Z2Z2, U2 = o.X * Z1Z1;
    Fq S1 = Y * o.Z * Z2Z2, S2 = o.Y * Z * Z1Z1;
    if (U1 == U2) return S1 == S2 ? dbl() : identity();
    Fq H = U2 - U1, I = H.dbl().square(), J = H * I;
    Fq rr = (S2 - S1).dbl(), V = U1 * I;
    Point r;
    r.X = rr.square() - J - V.dbl();
    r.Y = rr * (V - r.X) - (S1 * J).dbl();
    r.Z = ((Z + o.Z).square() - Z1Z1 - Z2Z2) * H;
    return r;
  }
  Point neg() const { Point r = *this; r.Y = -Y; return r; }
  Point operator-(const Point& o) const { return *this + o.neg(); }
  Point& operator+=(const Point& o) { *this = *this + o; return *this; }
  bool operator==(const Point& o) const {
    if (is_identity() || o.is_identity()) return is_identity() && o.is_identity();
    Fq Z1Z1 = Z.square(), Z2Z2 = o.Z.square();
    return X * Z2Z2 == o.X * Z1Z1 && Y * o.Z * Z2Z2 == o.Y * Z * Z1Z1;
  }
  Point mul_limbs(const u64* e) const {
    Point r = identity();
    for (int i = 255; i >= 0; i--) { r = r.dbl(); if ((e[i / 64] >> (i % 64)) & 1) r = r + *this; }
    return r;
  }
  Point operator*(const Fr& s) const { u64 e[4]; s.to_canonical(e); return mul_limbs(e); }
  // affine form; the identity has no affine coordinates: (0, 0) by ark-ec's convention for `Affine::identity()` (x = y = 0, infinity flag)
  void to_affine(Fq& x, Fq& y) const {
    if (is_identity()) { x = Fq::zero(); y = Fq::zero(); return; }
    Fq zi = Z.inverse(), zi2 = zi.square(); x = X * zi2; y = Y * zi2 * zi;
  }
  // ark-serialize compressed form of a SW affine point (ark-ec `SWFlags`): x as 32 bytes LE; bit 7 of the last byte = "y is negative"
  // (y > -y as canonical integers), bit 6 = point at infinity (x = 0).  Both bits are free: Fq has 254 bits.
  void compress(uint8_t* out) const {
    if (is_identity()) { memset(out, 0, 32); out[31] = 0x40; return; }
    Fq x, y; to_affine(x, y);
    x.to_bytes_le(out);
    Fq ny = -y;
    if (!y.le_canonical(ny)) out[31] |= 0x80;
  }
};

inline bool fq_sqrt(const Fq& a, Fq& out) {  // q = 3 mod 4: a^((q+1)/4)
  static const u64 e[4] = {0x4f082305b61f3f52ULL, 0x65e05aa45a1c72a3ULL, 0x6e14116da0605617ULL, 0x0c19139cb84c680aULL};
  Fq r = a.pow(e);
  if (r.square() == a) { out = r; return true; }
  return false;
}

// ark-ec `Affine::get_ys_from_x_unchecked`: y^2 = x^3 + 3; returns (smaller, larger) as canonical integers
inline bool sw_ys_from_x(const Fq& x, Fq& y_small, Fq& y_large) {
  Fq y;
  if (!fq_sqrt(x.square() * x + Point::b(), y)) return false;
  Fq ny = -y;
  if (y.le_canonical(ny)) { y_small = y; y_large = ny; } else { y_small = ny; y_large = y; }
  return true;
}

inline bool curve_decompress(const uint8_t* in, Point& out) {
  uint8_t b[32]; memcpy(b, in, 32);
  const bool neg = (b[31] & 0x80) != 0, inf = (b[31] & 0x40) != 0; b[31] &= 0x3f;
  u64 c[4] = {0, 0, 0, 0};
  for (int i = 0; i < 32; i++) c[i / 8] |= (u64)b[i] << (8 * (i % 8));
  // ark-ec 0.4 SWCurveConfig::deserialize_with_mode: (negative, infinity) both set is not a flag value; x is read as a canonical Fq first; with the
  // infinity flag the result is Affine::identity() whatever x holds
  if (neg && inf) return false;
  if (Fq::geq_p(c)) return false;
  if (inf) { out = Point::identity(); return true; }
  Fq x = Fq::from_canonical(c), ys, yl;
  if (!sw_ys_from_x(x, ys, yl)) return false;
  out = Point::from_affine(x, neg ? yl : ys);
  return true;
}

}  // namespace orc

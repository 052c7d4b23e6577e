// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// Twisted Edwards curve -x^2 + y^2 = 1 + d x^2 y^2 over Fq = 2^255-19 ("curve25519" in arkworks:
// `ark_curve25519::EdwardsProjective`, the only group the reference instantiates —
// src/benches/bench.rs:6, src/e2e_test.rs:1).  The group law, `normalize_batch`, point
// (de)serialisation and `G::rand` live in ark-ec / ark-serialize ^0.4.2 (Cargo.toml:40,43), which
// are NOT in /root/reference; they are restated here from their published algorithms.
// Coordinates: extended (X:Y:T:Z), x = X/Z, y = Y/Z, T = XY/Z (ark-ec `twisted_edwards::Projective`).
// Parity status: group law pinned by algebraic identities and by Python big-int affine formulas
// (tests/test_oracle_curve.py); serialisation sign convention and G::rand are restated from memory of
// the crates and are UNPINNED (no reference fixture exists — SURVEY.md §8c).
#pragma once
#include "ff.hpp"
#include <vector>

namespace orc {

struct EdConsts {
  static Fq d() {
    static const u64 c[4] = {0x75eb4dca135978a3ULL, 0x00700a4d4141d8abULL, 0x8cc740797779e898ULL, 0x52036cee2b6ffe73ULL};
    static const Fq v = Fq::from_canonical(c); return v;
  }
  static Fq d2() { static const Fq v = d() + d(); return v; }
  static Fq sqrtm1() {
    // 2^((p-1)/4) mod p
    static const u64 c[4] = {0xc4ee1b274a0ea0b0ULL, 0x2f431806ad2fe478ULL, 0x2b4d00993dfbd7a7ULL, 0x2b8324804fc1df0bULL};
    static const Fq v = Fq::from_canonical(c); return v;
  }
};

struct Point {
  Fq X, Y, T, Z;
  static Point identity() { Point p; p.X = Fq::zero(); p.Y = Fq::one(); p.T = Fq::zero(); p.Z = Fq::one(); return p; }
  static Point from_affine(const Fq& x, const Fq& y) { Point p; p.X = x; p.Y = y; p.T = x * y; p.Z = Fq::one(); return p; }
  static Point generator() {
    static const u64 gx[4] = {0xc9562d608f25d51aULL, 0x692cc7609525a7b2ULL, 0xc0a4e231fdd6dc5cULL, 0x216936d3cd6e53feULL};
    static const u64 gy[4] = {0x6666666666666658ULL, 0x6666666666666666ULL, 0x6666666666666666ULL, 0x6666666666666666ULL};
    return from_affine(Fq::from_canonical(gx), Fq::from_canonical(gy));
  }
  // add-2008-hwcd-3 (a = -1), unified and complete on this curve
  Point operator+(const Point& o) const {
    Fq A = (Y - X) * (o.Y - o.X);
    Fq B = (Y + X) * (o.Y + o.X);
    Fq C = T * EdConsts::d2() * o.T;
    Fq D = (Z * o.Z).dbl();
    Fq E = B - A, F = D - C, G = D + C, H = B + A;
    Point r; r.X = E * F; r.Y = G * H; r.T = E * H; r.Z = F * G; return r;
  }
  Point dbl() const {  // dbl-2008-hwcd, a = -1
    Fq A = X.square(), B = Y.square(), C = Z.square().dbl();
    Fq D = -A;
    Fq E = (X + Y).square() - A - B;
    Fq G = D + B, F = G - C, H = D - B;
    Point r; r.X = E * F; r.Y = G * H; r.T = E * H; r.Z = F * G; return r;
  }
  Point neg() const { Point r = *this; r.X = -X; r.T = -T; return r; }
  Point operator-(const Point& o) const { return *this + o.neg(); }
  Point& operator+=(const Point& o) { *this = *this + o; return *this; }
  bool operator==(const Point& o) const { return X * o.Z == o.X * Z && Y * o.Z == o.Y * Z; }
  bool is_identity() const { return X.is_zero() && Y == Z; }
  // scalar given as canonical 4-limb integer
  Point mul_limbs(const u64* e) const {
    Point r = identity();
    for (int i = 255; i >= 0; i--) { r = r.dbl(); if ((e[i / 64] >> (i % 64)) & 1) r = r + *this; }
    return r;
  }
  Point operator*(const Fr& s) const { u64 e[4]; s.to_canonical(e); return mul_limbs(e); }
  void to_affine(Fq& x, Fq& y) const { Fq zi = Z.inverse(); x = X * zi; y = Y * zi; }
  // ark-serialize compressed form of a TE affine point: y (32 bytes LE) with the x-sign flag in bit 7
  // of the last byte; "negative" means x > -x as canonical integers (ark-ec TEFlags::from_x_coordinate).
  void compress(uint8_t* out) const {
    Fq x, y; to_affine(x, y);
    y.to_bytes_le(out);
    Fq nx = -x;
    if (!x.le_canonical(nx)) out[31] |= 0x80;
  }
};

inline bool fq_sqrt(const Fq& a, Fq& out) {  // p = 5 mod 8
  static const u64 e[4] = {0xfffffffffffffffeULL, 0xffffffffffffffffULL, 0xffffffffffffffffULL, 0x0fffffffffffffffULL};  // (p+3)/8
  Fq r = a.pow(e);
  if (r.square() == a) { out = r; return true; }
  r = r * EdConsts::sqrtm1();
  if (r.square() == a) { out = r; return true; }
  return false;
}

// ark-ec `Affine::get_xs_from_y_unchecked`: x^2 = (1 - y^2) / (a - d y^2), a = -1; returns (smaller, larger)
inline bool ed_xs_from_y(const Fq& y, Fq& x_small, Fq& x_large) {
  Fq y2 = y.square();
  Fq num = Fq::one() - y2;
  Fq den = -Fq::one() - EdConsts::d() * y2;
  if (den.is_zero()) return false;
  Fq x2 = num * den.inverse();
  Fq x;
  if (!fq_sqrt(x2, x)) return false;
  Fq nx = -x;
  if (x.le_canonical(nx)) { x_small = x; x_large = nx; } else { x_small = nx; x_large = x; }
  return true;
}

inline bool ed_decompress(const uint8_t* in, Point& out) {
  uint8_t b[32]; memcpy(b, in, 32);
  bool neg = (b[31] & 0x80) != 0; b[31] &= 0x7f;
  u64 c[4] = {0, 0, 0, 0};
  for (int i = 0; i < 32; i++) c[i / 8] |= (u64)b[i] << (8 * (i % 8));
  if (Fq::geq_p(c)) return false;
  Fq y = Fq::from_canonical(c), xs, xl;
  if (!ed_xs_from_y(y, xs, xl)) return false;
  out = Point::from_affine(neg ? xl : xs, y);
  return true;
}

inline bool curve_decompress(const uint8_t* in, Point& out) { return ed_decompress(in, out); }   // the name both curve headers share

}  // namespace orc

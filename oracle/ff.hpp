// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path
// (lasso_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
//
// 256-bit Montgomery prime-field arithmetic, 4 x u64 limbs, R = 2^256 — a restatement of what
// ark-ff ^0.4.2 `Fp256<MontBackend<_, 4>>` does (third-party crate, NOT in /root/reference:
// Cargo.toml:38 `ark-ff`).  The in-memory form is identical to ark-ff's: little-endian u64 limbs
// holding a*R mod p.  Two instances are used by the reference path:
//   Fr = curve25519 scalar field, p = 2^252 + 27742317777372353535851937790883648493
//        (`ark_curve25519::Fr`, src/benches/bench.rs:6, src/e2e_test.rs:1)
//   Fq = curve25519 base field,   p = 2^255 - 19
// With -DORC_BN254 the same two names stand for ark-bn254's Fr / Fq (configs[1] of BASELINE.json names G = BN254).
// Parity status: arithmetic is pinned against Python big integers (tests/test_oracle_field.py)
// and against the reference's own small-integer KATs.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstdio>

namespace orc {
typedef uint64_t u64;
typedef unsigned __int128 u128;

template <class P>
struct Fp {
  u64 v[4];  // Montgomery form

  static Fp zero() { Fp r; r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0; return r; }
  static Fp one() { Fp r; memcpy(r.v, P::R1, 32); return r; }
  static Fp from_raw(const u64* limbs) { Fp r; memcpy(r.v, limbs, 32); return r; }  // already Montgomery
  static Fp from_u64(u64 x) {
    Fp t; t.v[0] = x; t.v[1] = t.v[2] = t.v[3] = 0;
    Fp r2 = from_raw(P::R2);
    return t * r2;  // x * R^2 / R = xR
  }
  // canonical integer (4 limbs, little endian) -> field element; input must be < 2^256, reduced here
  static Fp from_canonical(const u64* limbs) {
    Fp t; memcpy(t.v, limbs, 32);
    while (geq_p(t.v)) sub_p(t.v);
    return t * from_raw(P::R2);
  }
  void to_canonical(u64* out) const {
    Fp o; o.v[0] = 1; o.v[1] = o.v[2] = o.v[3] = 0;
    Fp t = (*this) * o;  // aR * 1 / R = a
    memcpy(out, t.v, 32);
  }
  // ark-ff `from_le_bytes_mod_order` (utils/transcript.rs:61-65 call site): integer from LE bytes, mod p
  static Fp from_le_bytes_mod_order(const uint8_t* bytes, size_t n) {
    // Horner over bytes from the most significant end: acc = acc*256 + b
    Fp acc = zero();
    Fp c256 = from_u64(256);
    for (size_t i = n; i-- > 0;) acc = acc * c256 + from_u64(bytes[i]);
    return acc;
  }
  void to_bytes_le(uint8_t* out) const {  // ark-serialize: 32 bytes, canonical, little endian
    u64 c[4]; to_canonical(c);
    for (int i = 0; i < 32; i++) out[i] = (uint8_t)(c[i / 8] >> (8 * (i % 8)));
  }

  static bool geq_p(const u64* a) {
    for (int i = 3; i >= 0; i--) {
      if (a[i] > P::P[i]) return true;
      if (a[i] < P::P[i]) return false;
    }
    return true;
  }
  static void sub_p(u64* a) {
    u64 borrow = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)a[i] - P::P[i] - borrow;
      a[i] = (u64)d; borrow = (u64)(d >> 64) & 1;
    }
  }

  Fp operator+(const Fp& o) const {
    Fp r; u64 carry = 0;
    for (int i = 0; i < 4; i++) { u128 s = (u128)v[i] + o.v[i] + carry; r.v[i] = (u64)s; carry = (u64)(s >> 64); }
    if (carry || geq_p(r.v)) sub_p(r.v);
    return r;
  }
  Fp operator-(const Fp& o) const {
    Fp r; u64 borrow = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)v[i] - o.v[i] - borrow; r.v[i] = (u64)d; borrow = (u64)(d >> 64) & 1; }
    if (borrow) { u64 carry = 0; for (int i = 0; i < 4; i++) { u128 s = (u128)r.v[i] + P::P[i] + carry; r.v[i] = (u64)s; carry = (u64)(s >> 64); } }
    return r;
  }
  Fp operator-() const { return zero() - *this; }
  // CIOS Montgomery multiplication
  Fp operator*(const Fp& o) const {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u64 carry = 0;
      for (int j = 0; j < 4; j++) { u128 s = (u128)v[j] * o.v[i] + t[j] + carry; t[j] = (u64)s; carry = (u64)(s >> 64); }
      u128 s = (u128)t[4] + carry; t[4] = (u64)s; t[5] = (u64)(s >> 64);
      u64 m = t[0] * P::INV;
      u128 s0 = (u128)m * P::P[0] + t[0]; carry = (u64)(s0 >> 64);
      for (int j = 1; j < 4; j++) { u128 s1 = (u128)m * P::P[j] + t[j] + carry; t[j - 1] = (u64)s1; carry = (u64)(s1 >> 64); }
      u128 s2 = (u128)t[4] + carry; t[3] = (u64)s2; t[4] = t[5] + (u64)(s2 >> 64);
    }
    Fp r; memcpy(r.v, t, 32);
    if (t[4] || geq_p(r.v)) sub_p(r.v);
    return r;
  }
  Fp& operator+=(const Fp& o) { *this = *this + o; return *this; }
  Fp& operator-=(const Fp& o) { *this = *this - o; return *this; }
  Fp& operator*=(const Fp& o) { *this = *this * o; return *this; }
  bool operator==(const Fp& o) const { return memcmp(v, o.v, 32) == 0; }
  bool operator!=(const Fp& o) const { return !(*this == o); }
  bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
  Fp square() const { return (*this) * (*this); }
  Fp dbl() const { return *this + *this; }
  // exponent given as canonical 4-limb integer
  Fp pow(const u64* e) const {
    Fp r = one();
    for (int i = 255; i >= 0; i--) { r = r.square(); if ((e[i / 64] >> (i % 64)) & 1) r = r * (*this); }
    return r;
  }
  Fp inverse() const {  // Fermat: a^(p-2); inverse(0) = 0 (callers never divide by zero on the path)
    u64 e[4]; memcpy(e, P::P, 32);
    // p - 2 (p is odd and its low limb is >= 2 for both fields)
    e[0] -= 2;
    return pow(e);
  }
  Fp operator/(const Fp& o) const { return (*this) * o.inverse(); }
  // number of significant bits of the canonical integer (ark-ff BigInteger::num_bits)
  int num_bits() const {
    u64 c[4]; to_canonical(c);
    for (int i = 3; i >= 0; i--) if (c[i]) return 64 * i + (64 - __builtin_clzll(c[i]));
    return 0;
  }
  // canonical-integer comparison this <= o (ark-ff Ord on Fp compares into_bigint())
  bool le_canonical(const Fp& o) const {
    u64 a[4], b[4]; to_canonical(a); o.to_canonical(b);
    for (int i = 3; i >= 0; i--) { if (a[i] < b[i]) return true; if (a[i] > b[i]) return false; }
    return true;
  }
};

#ifdef ORC_BN254
// BN254 (ark-bn254 ^0.4): Fr = the order of G1, Fq = its base field; both 254 bits (see bn254.hpp)
struct FrParams {
  static constexpr u64 P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr u64 R1[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
  static constexpr u64 R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
  static constexpr u64 INV = 0xc2e1f593efffffffULL;
  static constexpr int MODULUS_BITS = 254;
};
struct FqParams {
  static constexpr u64 P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr u64 R1[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
  static constexpr u64 R2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL};
  static constexpr u64 INV = 0x87d20782e4866389ULL;
  static constexpr int MODULUS_BITS = 254;
};
#else
struct FrParams {
  static constexpr u64 P[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0x0ULL, 0x1000000000000000ULL};
  static constexpr u64 R1[4] = {0xd6ec31748d98951dULL, 0xc6ef5bf4737dcf70ULL, 0xfffffffffffffffeULL, 0x0fffffffffffffffULL};
  static constexpr u64 R2[4] = {0xa40611e3449c0f01ULL, 0xd00e1ba768859347ULL, 0xceec73d217f5be65ULL, 0x0399411b7c309a3dULL};
  static constexpr u64 INV = 0xd2b51da312547e1bULL;
  static constexpr int MODULUS_BITS = 253;
};
struct FqParams {
  static constexpr u64 P[4] = {0xffffffffffffffedULL, 0xffffffffffffffffULL, 0xffffffffffffffffULL, 0x7fffffffffffffffULL};
  static constexpr u64 R1[4] = {0x26ULL, 0, 0, 0};
  static constexpr u64 R2[4] = {0x5a4ULL, 0, 0, 0};
  static constexpr u64 INV = 0x86bca1af286bca1bULL;
  static constexpr int MODULUS_BITS = 255;
};
#endif
typedef Fp<FrParams> Fr;
typedef Fp<FqParams> Fq;

}  // namespace orc

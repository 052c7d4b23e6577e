// Modular inversion for the host tails of the prover: Bernstein-Yang "safegcd" divsteps on 62-bit signed limbs (variable time — the prover's
// challenges and blinds are not secrets on this path).  Every bullet round (bullet.rs:98-132) needs u^-1 and one point normalisation before
// the next launch can go out; with Fermat exponentiation those two inversions were 21 us of the ~30 us the host spends per round.
//
// State: f = modulus, g = x, and d, e with d*x = f, e*x = g (mod modulus).  A batch of 62 divsteps is decided from the low 64 bits of f, g
// and summarised by a 2x2 integer matrix t (entries < 2^62 in magnitude) with  2^62 * (f', g') = t * (f, g);  (d, e) follow the same
// matrix modulo the modulus, where the division by 2^62 is made exact by adding a multiple of the modulus.  When g reaches 0, f = +-1 and
// d = +-x^-1.  (divstep: delta > 0 and g odd -> (1 - delta, g, (g - f)/2);  g odd -> (1 + delta, f, (g + f)/2);  else (1 + delta, f, g/2).)
#pragma once
#include <cstdint>
#include <cstring>

namespace lasso {

struct ModInv256 {
  int64_t m[5];       // modulus in 62-bit limbs
  uint64_t m_inv62;   // modulus^-1 mod 2^62

  static constexpr uint64_t M62 = (1ull << 62) - 1;

  static void to_s62(const uint64_t w[4], int64_t r[5]) {
    r[0] = (int64_t)(w[0] & M62);
    r[1] = (int64_t)(((w[0] >> 62) | (w[1] << 2)) & M62);
    r[2] = (int64_t)(((w[1] >> 60) | (w[2] << 4)) & M62);
    r[3] = (int64_t)(((w[2] >> 58) | (w[3] << 6)) & M62);
    r[4] = (int64_t)(w[3] >> 56);
  }
  static void from_s62(const int64_t r[5], uint64_t w[4]) {   // limbs 0..3 in [0, 2^62), limb 4 in [0, 2^8)
    w[0] = (uint64_t)r[0] | ((uint64_t)r[1] << 62);
    w[1] = ((uint64_t)r[1] >> 2) | ((uint64_t)r[2] << 60);
    w[2] = ((uint64_t)r[2] >> 4) | ((uint64_t)r[3] << 58);
    w[3] = ((uint64_t)r[3] >> 6) | ((uint64_t)r[4] << 56);
  }
  explicit ModInv256(const uint64_t modulus[4]) {
    to_s62(modulus, m);
    uint64_t x = (uint64_t)m[0];   // odd; Newton: x <- x (2 - m x) doubles the number of correct low bits (x = m is correct to 3 bits)
    for (int i = 0; i < 6; i++) x *= 2 - (uint64_t)m[0] * x;
    m_inv62 = x & M62;
  }

  struct T2 { int64_t u, v, q, r; };

  // 62 divsteps on the low words; eta = -delta.  Returns the new eta.
  static int64_t divsteps_62(int64_t eta, uint64_t f, uint64_t g, T2& t) {
    uint64_t u = 1, v = 0, q = 0, r = 1;
    int i = 62;
    for (;;) {
      // strip the zeros of g (each is one "g even" divstep): g /= 2 is tracked as doubling the f-row of the matrix
      int zeros = __builtin_ctzll(g | (~0ull << i));
      g >>= zeros; u <<= zeros; v <<= zeros; eta -= zeros; i -= zeros;
      if (i == 0) break;
      // g is odd here
      if (eta < 0) {   // delta > 0: swap, (f, g) <- (g, -f)
        eta = -eta;
        uint64_t tmp = f; f = g; g = 0 - tmp;
        tmp = u; u = q; q = 0 - tmp;
        tmp = v; v = r; r = 0 - tmp;
      }
      // g <- g + f (then at least one zero bit appears, stripped at the top of the loop); the same row operation on the matrix
      g += f; q += u; r += v;
    }
    t.u = (int64_t)u; t.v = (int64_t)v; t.q = (int64_t)q; t.r = (int64_t)r;
    return eta;
  }
  // (f, g) <- t * (f, g) / 2^62, exact
  static void update_fg(int64_t f[5], int64_t g[5], const T2& t) {
    typedef __int128 i128;
    i128 cf = (i128)t.u * f[0] + (i128)t.v * g[0];
    i128 cg = (i128)t.q * f[0] + (i128)t.r * g[0];
    cf >>= 62; cg >>= 62;
    for (int i = 1; i < 5; i++) {
      cf += (i128)t.u * f[i] + (i128)t.v * g[i];
      cg += (i128)t.q * f[i] + (i128)t.r * g[i];
      f[i - 1] = (int64_t)((uint64_t)cf & M62); cf >>= 62;
      g[i - 1] = (int64_t)((uint64_t)cg & M62); cg >>= 62;
    }
    f[4] = (int64_t)cf; g[4] = (int64_t)cg;
  }
  // (d, e) <- t * (d, e) / 2^62 mod modulus; d, e stay in (-2 modulus, modulus)
  void update_de(int64_t d[5], int64_t e[5], const T2& t) const {
    typedef __int128 i128;
    const int64_t sd = d[4] >> 63, se = e[4] >> 63;
    int64_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);   // negative d / e: start from d + modulus, e + modulus
    i128 cd = (i128)t.u * d[0] + (i128)t.v * e[0];
    i128 ce = (i128)t.q * d[0] + (i128)t.r * e[0];
    md -= (int64_t)((m_inv62 * (uint64_t)cd + (uint64_t)md) & M62);   // make the low 62 bits of t*(d,e) + modulus*(md,me) vanish
    me -= (int64_t)((m_inv62 * (uint64_t)ce + (uint64_t)me) & M62);
    cd += (i128)m[0] * md; ce += (i128)m[0] * me;
    cd >>= 62; ce >>= 62;
    for (int i = 1; i < 5; i++) {
      cd += (i128)t.u * d[i] + (i128)t.v * e[i] + (i128)m[i] * md;
      ce += (i128)t.q * d[i] + (i128)t.r * e[i] + (i128)m[i] * me;
      d[i - 1] = (int64_t)((uint64_t)cd & M62); cd >>= 62;
      e[i - 1] = (int64_t)((uint64_t)ce & M62); ce >>= 62;
    }
    d[4] = (int64_t)cd; e[4] = (int64_t)ce;
  }
  static bool is_zero(const int64_t a[5]) { return (a[0] | a[1] | a[2] | a[3] | a[4]) == 0; }
  static bool is_negative(const int64_t a[5]) { return a[4] < 0; }
  // a <- a + s*modulus (s = +-1) with carry propagation; top limb keeps the sign
  void add_mod(int64_t a[5], int64_t s) const {
    int64_t c = 0;
    for (int i = 0; i < 4; i++) { int64_t x = a[i] + s * m[i] + c; a[i] = x & (int64_t)M62; c = x >> 62; }
    a[4] = a[4] + s * m[4] + c;
  }
  static void negate(int64_t a[5]) {
    int64_t c = 0;
    for (int i = 0; i < 4; i++) { int64_t x = c - a[i]; a[i] = x & (int64_t)M62; c = x >> 62; }
    a[4] = c - a[4];
  }
  bool geq_mod(const int64_t a[5]) const {   // a >= modulus, a non-negative and carry-normalised
    for (int i = 4; i >= 0; i--) if (a[i] != m[i]) return a[i] > m[i];
    return true;
  }
  // x^-1 mod modulus for x in [0, modulus) given as four 64-bit words (plain integer, not Montgomery); inverse(0) = 0.
  // Returns false (out untouched) only if the iteration guard trips, which a 256-bit input cannot do; callers then use Fermat.
  bool inverse(const uint64_t x[4], uint64_t out[4]) const {
    int64_t f[5], g[5], d[5] = {0, 0, 0, 0, 0}, e[5] = {1, 0, 0, 0, 0};
    memcpy(f, m, sizeof(f)); to_s62(x, g);
    if (is_zero(g)) { out[0] = out[1] = out[2] = out[3] = 0; return true; }
    int64_t eta = -1;
    for (int it = 0; it < 16 && !is_zero(g); it++) {   // 590 divsteps suffice for 256-bit inputs: 10 batches
      T2 t; eta = divsteps_62(eta, (uint64_t)f[0], (uint64_t)g[0], t);
      update_de(d, e, t);
      update_fg(f, g, t);
    }
    if (!is_zero(g)) return false;
    // f = +-1; d = sign(f) * x^-1 in (-2 modulus, modulus)
    if (is_negative(f)) negate(d);
    while (is_negative(d)) add_mod(d, 1);
    while (geq_mod(d)) add_mod(d, -1);
    from_s62(d, out);
    return true;
  }
};

}  // namespace lasso

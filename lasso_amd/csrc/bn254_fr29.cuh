// BN254 build (-DLASSO_BN254) of fr29.cuh: the same interface and the same contracts (u-form / s-form, "reduced" / "loose", what the
// reductions accept) over ark-bn254's Fr, through the general-modulus arithmetic of mont29.cuh.  What changes against the curve25519 header:
// every reduction row is full (162 multiply-adds per product instead of 126), the lazy reductions take their quotient from a reciprocal
// (m29_near), and they accept MORE than the curve25519 ones (|value| < 2^258 ~ 21 p against 2^255 ~ 8 p) and return LESS (semi: below
// p (1 + 2^-24) against 4 p), so every kernel whose magnitudes were argued for the curve25519 header in units of p holds here as well:
// products come out in (-X, p + X) with X = |a| |b| / 2^261 — three times larger relative to p (p / 2^261 = 2^-7.4 against 2^-9), still below p
// for every operand pair the kernels form (a < 8 p times an s-form challenge < 32 p gives X < 1.5 p only for the widest sums; binds multiply a
// difference of two semi values, |a| < 1.01 p, X < 0.2 p).
#pragma once
#include <stdint.h>
#include "fr.cuh"
#include "mont29.cuh"

struct Bn254FrM {
  static LHD int32_t p(int k) { const int32_t P[9] = {268435457, 521120927, 240919632, 131109107, 361091715, 47923392, 10936641, 240920116, 3171406}; return P[k]; }
  static constexpr uint32_t PINV = 268435455u;
  static constexpr int32_t QC = 1420063842;
  static constexpr int32_t ONE_S_0 = 268435287, ONE_S_1 = 514263732, ONE_S_2 = 86771339, ONE_S_3 = 391139145, ONE_S_4 = 178784091, ONE_S_5 = 490881230, ONE_S_6 = 299191303,
                           ONE_S_7 = 86689704, ONE_S_8 = 903222;
  static constexpr int32_t K522_0 = 95853524, K522_1 = 102173274, K522_2 = 34397646, K522_3 = 498479371, K522_4 = 240439551, K522_5 = 486036963, K522_6 = 471195907,
                           K522_7 = 131109217, K522_8 = 656714;
};
typedef m29<Bn254FrM> fr29;
typedef m29_acc<Bn254FrM> fr29_acc;
#define FR29_MASK M29_MASK

LHD fr29 fr29_zero() { return m29_zero<Bn254FrM>(); }
LHD fr29 fr29_from_limbs(int32_t a0, int32_t a1, int32_t a2, int32_t a3, int32_t a4, int32_t a5, int32_t a6, int32_t a7, int32_t a8) {
  fr29 r; r.v[0] = a0; r.v[1] = a1; r.v[2] = a2; r.v[3] = a3; r.v[4] = a4; r.v[5] = a5; r.v[6] = a6; r.v[7] = a7; r.v[8] = a8; return r;
}
LHD fr29 fr29_one_s() { return fr29_from_limbs(268435287, 514263732, 86771339, 391139145, 178784091, 490881230, 299191303, 86689704, 903222); }   // 2^261 mod p
LHD fr29 fr29_k5() { return fr29_from_limbs(268430039, 492061940, 71535269, 62181526, 323781850, 244503300, 348886451, 68918589, 360451); }          // 2^266 mod p
LHD fr29 fr29_k10() { return fr29_from_limbs(268262109, 223975601, 492627914, 522739689, 150938553, 164142673, 394138283, 408892696, 2020216); }       // 2^271 mod p
LHD fr29 fr29_r2s() { return fr29_from_limbs(338539743, 433494286, 343078028, 115075043, 193254777, 284818167, 304038784, 396432094, 1209799); }       // 2^517 mod p
LHD fr29 fr29_int_from_uu() { fr29 r = fr29_zero(); r.v[0] = 1 << 10; return r; }

LHD fr29 fr29_add(const fr29& a, const fr29& b) { return m29_add(a, b); }
LHD fr29 fr29_sub(const fr29& a, const fr29& b) { return m29_sub(a, b); }
LHD fr29 fr29_weak(const fr29& a) { return m29_weak(a); }
LHD fr29 fr29_unpack_u(const fr_t& x) { return m29_unpack_words<Bn254FrM>(x.v); }
LHD fr29 fr29_unpack_s(const fr_t& x) { return m29_unpack_words_shl5<Bn254FrM>(x.v); }
LHD fr29 fr29_from_u64_int(uint64_t x) { return fr29_from_limbs((int32_t)(x & FR29_MASK), (int32_t)((x >> 29) & FR29_MASK), (int32_t)(x >> 58), 0, 0, 0, 0, 0, 0); }
LHD fr29 fr29_mul(const fr29& a, const fr29& b) { return m29_mul(a, b); }
LHD fr29_acc fr29_acc_zero() { return m29_acc_zero<Bn254FrM>(); }
LHD void fr29_mul_acc(fr29_acc& acc, const fr29& a, const fr29& b) { m29_mul_acc(acc, a, b); }
LHD void fr29_acc_carry(fr29_acc& acc) { m29_acc_carry(acc); }
LHD fr29 fr29_acc_reduce(const fr29_acc& acc) { return m29_acc_reduce(acc); }
LHD fr29 fr29_semi(const fr29& a) { return m29_near(a); }
LHD fr29 fr29_canonical(const fr29& a) { return m29_canonical(a); }
LHD fr29 fr29_reduce_columns(const int64_t* col, int shift) { return m29_reduce_columns<Bn254FrM>(col, shift); }
LHD fr_t fr29_pack(const fr29& a) { fr_t r; m29_pack_words(a, r.v); return r; }
LHD fr_t fr29_store(const fr29& a) { return fr29_pack(fr29_canonical(a)); }
LHD fr29 fr29_from_columns(const int64_t* col) { return m29_from_columns<Bn254FrM>(col); }

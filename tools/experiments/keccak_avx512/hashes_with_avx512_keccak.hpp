// Host-side Fiat–Shamir plumbing of the product: Keccak-f[1600], SHAKE256, STROBE-128/Merlin, ChaCha RNG.
// The reference gets these from crates (merlin 3.0.0, sha3 0.8.2, rand_chacha 0.3, ark-std test_rng —
// Cargo.toml:29-33); call sites: src/utils/transcript.rs:20-72, src/poly/commitments.rs:22-44,
// src/utils/random.rs:15-30.  Sequential, O(rounds) work: it stays on the host by design.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>
#include <stdexcept>
#include <cstdlib>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace lasso {


// ---- Keccak-f[1600] on five 512-bit registers (x86-64 with AVX-512F: the EPYC hosts gfx950 ships in).  Register y holds plane y of the state (lanes x = 0..4 = A[x + 5y]).
// A round: theta = one vertical five-way XOR, two lane rotations of the parity vector, one three-way XOR per plane; rho = a per-lane variable rotate per plane; pi + chi:
// lane (X, Y) of the new state is old (x = X + 3Y mod 5, y = X), so permuting plane X by Y -> (X + 3Y) mod 5 gives COLUMN X of pi's output indexed by Y, chi is then a vertical
// three-operand logic op over columns X, X+1, X+2, and a 5 x 5 transposition (four unpacks, five two-source permutes, five masked inserts) returns to planes; iota.
// ~40 vector instructions per round instead of ~130 scalar ones; the scalar permutation (~650 cycles, every compiler and flag tried) is what the openings' a-vectors —
// 16 640 scalars per headline proof, appended one by one (dot_product.rs:196) — spend 1.08 ms per proof in.
// The algorithm is written ONCE (LASSO_KECCAK_ROUNDS below) over a small set of vector operations and instantiated twice: with the AVX-512 intrinsics, and with a plain C++
// emulation of exactly those operations (KeccakEmu) that the CPU tests run against the scalar permutation on any host.  Keccak1600::permute picks the AVX-512 form when the CPU
// has it AND a start-up self-test against the scalar form passes (LASSO_KECCAK_AVX512=0 forces the scalar form).
namespace keccak_vec {
alignas(64) static const uint64_t IDX_PREV[8] = {4, 0, 1, 2, 3, 5, 6, 7}, IDX_NEXT[8] = {1, 2, 3, 4, 0, 5, 6, 7};
alignas(64) static const uint64_t RHO[5][8] = {{0, 1, 62, 28, 27, 0, 0, 0}, {36, 44, 6, 55, 20, 0, 0, 0}, {3, 10, 43, 25, 39, 0, 0, 0}, {41, 45, 15, 21, 8, 0, 0, 0}, {18, 2, 61, 56, 14, 0, 0, 0}};
alignas(64) static const uint64_t PI1[5][8] = {{0, 3, 1, 4, 2, 5, 6, 7}, {1, 4, 2, 0, 3, 5, 6, 7}, {2, 0, 3, 1, 4, 5, 6, 7}, {3, 1, 4, 2, 0, 5, 6, 7}, {4, 2, 0, 3, 1, 5, 6, 7}};
alignas(64) static const uint64_t T_A[8] = {0, 1, 8, 9, 4, 5, 6, 7}, T_B[8] = {2, 3, 10, 11, 4, 5, 6, 7}, T_C[8] = {4, 5, 12, 13, 4, 5, 6, 7};
alignas(64) static const uint64_t T_4[5][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {1, 1, 1, 1, 1, 1, 1, 1}, {2, 2, 2, 2, 2, 2, 2, 2}, {3, 3, 3, 3, 3, 3, 3, 3}, {4, 4, 4, 4, 4, 4, 4, 4}};
static const uint64_t RC[24] = {0x1ULL, 0x8082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x808bULL, 0x80000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x8aULL, 0x88ULL,
                                0x80008009ULL, 0x8000000aULL, 0x8000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL,
                                0x8000000000000080ULL, 0x800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x80000001ULL, 0x8000000080008008ULL};
// the 24 rounds over the operations V_LOAD5 / V_STORE5 / V_IDX / V_XOR3 / V_CHI / V_PERM / V_PERM2 / V_MPERM4 / V_ROL1 / V_ROLV / V_UNPACKLO / V_UNPACKHI / V_XOR_LANE0 and the type V_T
#define LASSO_KECCAK_ROUNDS(A_)                                                                                                                                          \
  V_T r0 = V_LOAD5((A_)), r1 = V_LOAD5((A_) + 5), r2 = V_LOAD5((A_) + 10), r3 = V_LOAD5((A_) + 15), r4 = V_LOAD5((A_) + 20);                                              \
  const V_T ip = V_IDX(keccak_vec::IDX_PREV), in = V_IDX(keccak_vec::IDX_NEXT);                                                                                           \
  const V_T h0 = V_IDX(keccak_vec::RHO[0]), h1 = V_IDX(keccak_vec::RHO[1]), h2 = V_IDX(keccak_vec::RHO[2]), h3 = V_IDX(keccak_vec::RHO[3]), h4 = V_IDX(keccak_vec::RHO[4]);  \
  const V_T p0 = V_IDX(keccak_vec::PI1[0]), p1 = V_IDX(keccak_vec::PI1[1]), p2 = V_IDX(keccak_vec::PI1[2]), p3 = V_IDX(keccak_vec::PI1[3]), p4 = V_IDX(keccak_vec::PI1[4]);  \
  const V_T ta = V_IDX(keccak_vec::T_A), tb = V_IDX(keccak_vec::T_B), tc = V_IDX(keccak_vec::T_C);                                                                         \
  const V_T q0 = V_IDX(keccak_vec::T_4[0]), q1 = V_IDX(keccak_vec::T_4[1]), q2 = V_IDX(keccak_vec::T_4[2]), q3 = V_IDX(keccak_vec::T_4[3]), q4 = V_IDX(keccak_vec::T_4[4]);  \
  for (int rnd = 0; rnd < 24; rnd++) {                                                                                                                                    \
    const V_T c = V_XOR3(V_XOR3(r0, r1, r2), r3, r4);                                                                                                                     \
    const V_T cp = V_PERM(ip, c), cn = V_ROL1(V_PERM(in, c));                                                                                                             \
    r0 = V_ROLV(V_XOR3(r0, cp, cn), h0); r1 = V_ROLV(V_XOR3(r1, cp, cn), h1); r2 = V_ROLV(V_XOR3(r2, cp, cn), h2);                                                         \
    r3 = V_ROLV(V_XOR3(r3, cp, cn), h3); r4 = V_ROLV(V_XOR3(r4, cp, cn), h4);                                                                                             \
    const V_T t0 = V_PERM(p0, r0), t1 = V_PERM(p1, r1), t2 = V_PERM(p2, r2), t3 = V_PERM(p3, r3), t4 = V_PERM(p4, r4);                                                      \
    const V_T u0 = V_CHI(t0, t1, t2), u1 = V_CHI(t1, t2, t3), u2 = V_CHI(t2, t3, t4), u3 = V_CHI(t3, t4, t0), u4 = V_CHI(t4, t0, t1);                                       \
    const V_T lo01 = V_UNPACKLO(u0, u1), hi01 = V_UNPACKHI(u0, u1), lo23 = V_UNPACKLO(u2, u3), hi23 = V_UNPACKHI(u2, u3);                                                   \
    r0 = V_XOR_LANE0(V_MPERM4(V_PERM2(lo01, ta, lo23), q0, u4), keccak_vec::RC[rnd]);                                                                                     \
    r1 = V_MPERM4(V_PERM2(hi01, ta, hi23), q1, u4);                                                                                                                       \
    r2 = V_MPERM4(V_PERM2(lo01, tb, lo23), q2, u4);                                                                                                                       \
    r3 = V_MPERM4(V_PERM2(hi01, tb, hi23), q3, u4);                                                                                                                       \
    r4 = V_MPERM4(V_PERM2(lo01, tc, lo23), q4, u4);                                                                                                                       \
  }                                                                                                                                                                       \
  V_STORE5((A_), r0); V_STORE5((A_) + 5, r1); V_STORE5((A_) + 10, r2); V_STORE5((A_) + 15, r3); V_STORE5((A_) + 20, r4);

// plain C++ statement of the vector operations above (Intel SDM semantics of VPERMQ / VPERMT2Q / VPUNPCK{L,H}QDQ / VPROLVQ / VPTERNLOGQ 0x96, 0xD2 / masked VPERMQ)
struct Emu {
  uint64_t l[8];
  static Emu load5(const uint64_t* a) { Emu r; for (int i = 0; i < 8; i++) r.l[i] = i < 5 ? a[i] : 0; return r; }
  static void store5(uint64_t* a, const Emu& v) { for (int i = 0; i < 5; i++) a[i] = v.l[i]; }
  static Emu idx(const uint64_t* a) { Emu r; for (int i = 0; i < 8; i++) r.l[i] = a[i]; return r; }
  static Emu xor3(const Emu& a, const Emu& b, const Emu& c) { Emu r; for (int i = 0; i < 8; i++) r.l[i] = a.l[i] ^ b.l[i] ^ c.l[i]; return r; }
  static Emu chi(const Emu& a, const Emu& b, const Emu& c) { Emu r; for (int i = 0; i < 8; i++) r.l[i] = a.l[i] ^ (~b.l[i] & c.l[i]); return r; }
  static Emu perm(const Emu& ix, const Emu& a) { Emu r; for (int i = 0; i < 8; i++) r.l[i] = a.l[ix.l[i] & 7]; return r; }
  static Emu perm2(const Emu& a, const Emu& ix, const Emu& b) { Emu r; for (int i = 0; i < 8; i++) r.l[i] = (ix.l[i] & 8) ? b.l[ix.l[i] & 7] : a.l[ix.l[i] & 7]; return r; }
  static Emu mperm4(const Emu& src, const Emu& ix, const Emu& a) { Emu r = src; r.l[4] = a.l[ix.l[4] & 7]; return r; }   // mask 0x10: lane 4 only
  static Emu rol1(const Emu& a) { Emu r; for (int i = 0; i < 8; i++) r.l[i] = (a.l[i] << 1) | (a.l[i] >> 63); return r; }
  static Emu rolv(const Emu& a, const Emu& n) { Emu r; for (int i = 0; i < 8; i++) { const unsigned k = (unsigned)(n.l[i] & 63); r.l[i] = k ? (a.l[i] << k) | (a.l[i] >> (64 - k)) : a.l[i]; } return r; }
  static Emu unpacklo(const Emu& a, const Emu& b) { Emu r; for (int j = 0; j < 4; j++) { r.l[2 * j] = a.l[2 * j]; r.l[2 * j + 1] = b.l[2 * j]; } return r; }
  static Emu unpackhi(const Emu& a, const Emu& b) { Emu r; for (int j = 0; j < 4; j++) { r.l[2 * j] = a.l[2 * j + 1]; r.l[2 * j + 1] = b.l[2 * j + 1]; } return r; }
  static Emu xor_lane0(const Emu& a, uint64_t k) { Emu r = a; r.l[0] ^= k; return r; }
};
inline void permute_emulated(uint64_t* A) {
#define V_T keccak_vec::Emu
#define V_LOAD5 keccak_vec::Emu::load5
#define V_STORE5 keccak_vec::Emu::store5
#define V_IDX keccak_vec::Emu::idx
#define V_XOR3 keccak_vec::Emu::xor3
#define V_CHI keccak_vec::Emu::chi
#define V_PERM keccak_vec::Emu::perm
#define V_PERM2 keccak_vec::Emu::perm2
#define V_MPERM4 keccak_vec::Emu::mperm4
#define V_ROL1 keccak_vec::Emu::rol1
#define V_ROLV keccak_vec::Emu::rolv
#define V_UNPACKLO keccak_vec::Emu::unpacklo
#define V_UNPACKHI keccak_vec::Emu::unpackhi
#define V_XOR_LANE0 keccak_vec::Emu::xor_lane0
  LASSO_KECCAK_ROUNDS(A)
#undef V_T
#undef V_LOAD5
#undef V_STORE5
#undef V_IDX
#undef V_XOR3
#undef V_CHI
#undef V_PERM
#undef V_PERM2
#undef V_MPERM4
#undef V_ROL1
#undef V_ROLV
#undef V_UNPACKLO
#undef V_UNPACKHI
#undef V_XOR_LANE0
}
#if defined(__x86_64__) && !defined(LASSO_NO_AVX512_KECCAK)
#define LASSO_HAVE_AVX512_KECCAK 1
__attribute__((target("avx512f"))) inline void permute_avx512(uint64_t* A) {
#define V_T __m512i
#define V_LOAD5(p_) _mm512_maskz_loadu_epi64((__mmask8)0x1f, (p_))
#define V_STORE5(p_, v_) _mm512_mask_storeu_epi64((p_), (__mmask8)0x1f, (v_))
#define V_IDX(p_) _mm512_load_si512((const void*)(p_))
#define V_XOR3(a_, b_, c_) _mm512_ternarylogic_epi64((a_), (b_), (c_), 0x96)
#define V_CHI(a_, b_, c_) _mm512_ternarylogic_epi64((a_), (b_), (c_), 0xD2)
#define V_PERM(ix_, a_) _mm512_permutexvar_epi64((ix_), (a_))
#define V_PERM2(a_, ix_, b_) _mm512_permutex2var_epi64((a_), (ix_), (b_))
#define V_MPERM4(src_, ix_, a_) _mm512_mask_permutexvar_epi64((src_), (__mmask8)0x10, (ix_), (a_))
#define V_ROL1(a_) _mm512_rol_epi64((a_), 1)
#define V_ROLV(a_, n_) _mm512_rolv_epi64((a_), (n_))
#define V_UNPACKLO(a_, b_) _mm512_unpacklo_epi64((a_), (b_))
#define V_UNPACKHI(a_, b_) _mm512_unpackhi_epi64((a_), (b_))
#define V_XOR_LANE0(a_, k_) _mm512_xor_si512((a_), _mm512_maskz_set1_epi64((__mmask8)1, (long long)(k_)))
  LASSO_KECCAK_ROUNDS(A)
#undef V_T
#undef V_LOAD5
#undef V_STORE5
#undef V_IDX
#undef V_XOR3
#undef V_CHI
#undef V_PERM
#undef V_PERM2
#undef V_MPERM4
#undef V_ROL1
#undef V_ROLV
#undef V_UNPACKLO
#undef V_UNPACKHI
#undef V_XOR_LANE0
}
#endif
}  // namespace keccak_vec

// State kept as 25 little-endian 64-bit lanes (x86-64 host: the byte view aliases them); the round function is fully unrolled —
// the 4096-scalar vector appends of the opening proofs (dot_product.rs:196) push ~1 MB through STROBE per proof.
class Keccak1600 {
  static inline uint64_t rol(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }

 public:
  union { uint64_t A[25]; uint8_t bytes[200]; };
  Keccak1600() { memset(bytes, 0, sizeof(bytes)); }
  // 0 = scalar, 1 = AVX-512 (the CPU has it, LASSO_KECCAK_AVX512 != 0, and 64 chained permutations of a test state equal the scalar form's)
  static int vector_mode() {
    static const int mode = [] {
#ifdef LASSO_HAVE_AVX512_KECCAK
      const char* e = getenv("LASSO_KECCAK_AVX512"); if (e && e[0] == '0') return 0;
      __builtin_cpu_init(); if (!__builtin_cpu_supports("avx512f")) return 0;
      Keccak1600 a, b; for (int i = 0; i < 25; i++) a.A[i] = b.A[i] = 0x9e3779b97f4a7c15ull * (uint64_t)(i + 1) + (uint64_t)i;
      for (int i = 0; i < 64; i++) { a.permute_scalar(); keccak_vec::permute_avx512(b.A); }
      return memcmp(a.bytes, b.bytes, 200) == 0 ? 1 : 0;
#else
      return 0;
#endif
    }();
    return mode;
  }
  void permute() {
#ifdef LASSO_HAVE_AVX512_KECCAK
    if (vector_mode() == 1) { keccak_vec::permute_avx512(A); return; }
#endif
    permute_scalar();
  }
  void permute_scalar() {
    static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "lane/byte aliasing assumes a little-endian host");
    static const uint64_t RC[24] = {0x1ULL, 0x8082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x808bULL, 0x80000001ULL, 0x8000000080008081ULL,
                                    0x8000000000008009ULL, 0x8aULL, 0x88ULL, 0x80008009ULL, 0x8000000aULL, 0x8000808bULL, 0x800000000000008bULL,
                                    0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x800aULL,
                                    0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x80000001ULL, 0x8000000080008008ULL};
    for (int rnd = 0; rnd < 24; rnd++) {
      // theta, rho + pi (B[y][2x+3y] = rot(A[x][y] ^ D[x], r[x][y]), index = x + 5y), chi, iota
      const uint64_t C0 = A[0] ^ A[5] ^ A[10] ^ A[15] ^ A[20], C1 = A[1] ^ A[6] ^ A[11] ^ A[16] ^ A[21], C2 = A[2] ^ A[7] ^ A[12] ^ A[17] ^ A[22], C3 = A[3] ^ A[8] ^ A[13] ^ A[18] ^ A[23], C4 = A[4] ^ A[9] ^ A[14] ^ A[19] ^ A[24];
      const uint64_t D0 = C4 ^ rol(C1, 1), D1 = C0 ^ rol(C2, 1), D2 = C1 ^ rol(C3, 1), D3 = C2 ^ rol(C4, 1), D4 = C3 ^ rol(C0, 1);
      const uint64_t B0 = (A[0] ^ D0), B1 = rol((A[6] ^ D1), 44), B2 = rol((A[12] ^ D2), 43), B3 = rol((A[18] ^ D3), 21), B4 = rol((A[24] ^ D4), 14), B5 = rol((A[3] ^ D3), 28), B6 = rol((A[9] ^ D4), 20), B7 = rol((A[10] ^ D0), 3), B8 = rol((A[16] ^ D1), 45), B9 = rol((A[22] ^ D2), 61), B10 = rol((A[1] ^ D1), 1), B11 = rol((A[7] ^ D2), 6), B12 = rol((A[13] ^ D3), 25), B13 = rol((A[19] ^ D4), 8), B14 = rol((A[20] ^ D0), 18), B15 = rol((A[4] ^ D4), 27), B16 = rol((A[5] ^ D0), 36), B17 = rol((A[11] ^ D1), 10), B18 = rol((A[17] ^ D2), 15), B19 = rol((A[23] ^ D3), 56), B20 = rol((A[2] ^ D2), 62), B21 = rol((A[8] ^ D3), 55), B22 = rol((A[14] ^ D4), 39), B23 = rol((A[15] ^ D0), 41), B24 = rol((A[21] ^ D1), 2);
      A[0] = B0 ^ (~B1 & B2); A[1] = B1 ^ (~B2 & B3); A[2] = B2 ^ (~B3 & B4); A[3] = B3 ^ (~B4 & B0); A[4] = B4 ^ (~B0 & B1);
      A[5] = B5 ^ (~B6 & B7); A[6] = B6 ^ (~B7 & B8); A[7] = B7 ^ (~B8 & B9); A[8] = B8 ^ (~B9 & B5); A[9] = B9 ^ (~B5 & B6);
      A[10] = B10 ^ (~B11 & B12); A[11] = B11 ^ (~B12 & B13); A[12] = B12 ^ (~B13 & B14); A[13] = B13 ^ (~B14 & B10); A[14] = B14 ^ (~B10 & B11);
      A[15] = B15 ^ (~B16 & B17); A[16] = B16 ^ (~B17 & B18); A[17] = B17 ^ (~B18 & B19); A[18] = B18 ^ (~B19 & B15); A[19] = B19 ^ (~B15 & B16);
      A[20] = B20 ^ (~B21 & B22); A[21] = B21 ^ (~B22 & B23); A[22] = B22 ^ (~B23 & B24); A[23] = B23 ^ (~B24 & B20); A[24] = B24 ^ (~B20 & B21);
      A[0] ^= RC[rnd];
    }
  }
};

class Shake256 {
  Keccak1600 k; size_t at = 0; bool out_mode = false;
  static constexpr size_t RATE = 136;

 public:
  void update(const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data;
    for (size_t i = 0; i < n; i++) { k.bytes[at++] ^= p[i]; if (at == RATE) { k.permute(); at = 0; } }
  }
  void read(uint8_t* out, size_t n) {
    if (!out_mode) { k.bytes[at] ^= 0x1f; k.bytes[RATE - 1] ^= 0x80; k.permute(); at = 0; out_mode = true; }
    for (size_t i = 0; i < n; i++) { if (at == RATE) { k.permute(); at = 0; } out[i] = k.bytes[at++]; }
  }
};

// STROBE-128 as specialised by Merlin (only AD / meta-AD / PRF are used)
class Strobe {
  Keccak1600 k; uint8_t pos = 0, pos_begin = 0, cur = 0;
  static constexpr uint8_t RATE = 166;
  enum : uint8_t { I = 1, A = 2, C = 4, T = 8, M = 16, K = 32 };
  void run_f() { k.bytes[pos] ^= pos_begin; k.bytes[pos + 1] ^= 0x04; k.bytes[RATE + 1] ^= 0x80; k.permute(); pos = 0; pos_begin = 0; }
  // bulk forms: whole runs up to the end of the rate block at a time (the openings push ~1 MB of scalars through here per proof)
  void absorb(const uint8_t* d, size_t n) {
    while (n) {
      size_t chunk = RATE - pos; if (chunk > n) chunk = n;
      uint8_t* dst = k.bytes + pos;
      size_t i = 0;
      for (; i + 8 <= chunk; i += 8) { uint64_t a, b; memcpy(&a, dst + i, 8); memcpy(&b, d + i, 8); a ^= b; memcpy(dst + i, &a, 8); }
      for (; i < chunk; i++) dst[i] ^= d[i];
      pos = (uint8_t)(pos + chunk); d += chunk; n -= chunk;
      if (pos == RATE) run_f();
    }
  }
  void squeeze(uint8_t* d, size_t n) {
    while (n) {
      size_t chunk = RATE - pos; if (chunk > n) chunk = n;
      memcpy(d, k.bytes + pos, chunk); memset(k.bytes + pos, 0, chunk);
      pos = (uint8_t)(pos + chunk); d += chunk; n -= chunk;
      if (pos == RATE) run_f();
    }
  }
  void begin(uint8_t flags, bool more) {
    if (more) { if (cur != flags) throw std::logic_error("strobe: continued op with different flags"); return; }
    uint8_t hdr[2] = {pos_begin, flags};
    pos_begin = pos + 1; cur = flags;
    absorb(hdr, 2);
    if ((flags & (C | K)) && pos != 0) run_f();
  }

 public:
  explicit Strobe(const char* proto) {
    const uint8_t head[18] = {1, RATE + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    memcpy(k.bytes, head, 18); k.permute();
    meta_ad(proto, strlen(proto), false);
  }
  void meta_ad(const void* d, size_t n, bool more) { begin(M | A, more); absorb((const uint8_t*)d, n); }
  void ad(const void* d, size_t n, bool more) { begin(A, more); absorb((const uint8_t*)d, n); }
  void prf(uint8_t* d, size_t n, bool more) { begin(I | A | C, more); squeeze(d, n); }
};

// merlin::Transcript
class Merlin {
  Strobe s;
  static void le32(uint32_t v, uint8_t* o) { o[0] = (uint8_t)v; o[1] = (uint8_t)(v >> 8); o[2] = (uint8_t)(v >> 16); o[3] = (uint8_t)(v >> 24); }

 public:
  explicit Merlin(const char* label) : s("Merlin v1.0") { append_message("dom-sep", label, strlen(label)); }
  void append_message(const char* label, const void* msg, size_t n) { append_message_l(label, strlen(label), msg, n); }
  // the same with the label as (pointer, length): labels that arrive through the C ABI are not NUL-terminated (include/lasso_prover.h lasso_transcript_vtbl)
  void append_message_l(const void* label, size_t label_len, const void* msg, size_t n) {
    uint8_t len[4]; le32((uint32_t)n, len);
    s.meta_ad(label, label_len, false); s.meta_ad(len, 4, true); s.ad(msg, n, false);
  }
  void append_str(const char* label, const char* msg) { append_message(label, msg, strlen(msg)); }
  void append_u64(const char* label, uint64_t x) { uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i)); append_message(label, b, 8); }
  void challenge_bytes(const char* label, uint8_t* out, size_t n) { challenge_bytes_l(label, strlen(label), out, n); }
  void challenge_bytes_l(const void* label, size_t label_len, uint8_t* out, size_t n) {
    uint8_t len[4]; le32((uint32_t)n, len);
    s.meta_ad(label, label_len, false); s.meta_ad(len, 4, true); s.prf(out, n, false);
  }
};

// rand_chacha ChaCha{12,20}Rng: key = seed, 64-bit counter in words 12-13, stream id 0; 4-block output buffer with
// rand_core::BlockRng's word-index semantics for next_u32/next_u64.
class ChaChaRng {
  uint32_t key[8], buf[64]; uint64_t ctr = 0; int idx = 64, rounds;
  static uint32_t rl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
  void refill() {
    for (int blk = 0; blk < 4; blk++) {
      uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u}, x[16];
      memcpy(in + 4, key, 32); in[12] = (uint32_t)(ctr + blk); in[13] = (uint32_t)((ctr + blk) >> 32); in[14] = in[15] = 0;
      memcpy(x, in, 64);
      auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rl(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rl(x[d] ^ x[a], 8); x[c] += x[d]; x[b] = rl(x[b] ^ x[c], 7);
      };
      for (int r = 0; r < rounds; r += 2) { qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15); qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14); }
      for (int i = 0; i < 16; i++) buf[16 * blk + i] = x[i] + in[i];
    }
    ctr += 4;
  }

 public:
  ChaChaRng(const uint8_t seed[32], int rounds_) : rounds(rounds_) { for (int i = 0; i < 8; i++) key[i] = (uint32_t)seed[4 * i] | (uint32_t)seed[4 * i + 1] << 8 | (uint32_t)seed[4 * i + 2] << 16 | (uint32_t)seed[4 * i + 3] << 24; }
  uint32_t next_u32() { if (idx >= 64) { refill(); idx = 0; } return buf[idx++]; }
  uint64_t next_u64() {
    if (idx < 63) { uint64_t v = buf[idx] | (uint64_t)buf[idx + 1] << 32; idx += 2; return v; }
    if (idx >= 64) { refill(); idx = 2; return buf[0] | (uint64_t)buf[1] << 32; }
    uint64_t lo = buf[63]; refill(); idx = 1; return lo | (uint64_t)buf[0] << 32;
  }
  // ark_std::test_rng(): rand's StdRng (ChaCha12) from the fixed seed
  static ChaChaRng test_rng() {
    const uint8_t seed[32] = {1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0};
    return ChaChaRng(seed, 12);
  }
};

}  // namespace lasso

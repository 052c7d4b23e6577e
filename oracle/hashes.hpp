// ORACLE — TEST INFRASTRUCTURE ONLY (see ff.hpp header).
//
// Keccak-f[1600], SHAKE256, STROBE-128 and the Merlin transcript; ChaCha block RNG.
// These live in third-party crates absent from /root/reference: merlin ^3.0.0 (Cargo.toml:29),
// sha3 ^0.8.2 (Cargo.toml:33), rand_chacha ^0.3.0 (Cargo.toml:32), rand/ark-std (test_rng).
// Restated from the published specifications (FIPS 202; STROBE v1.0.2; merlin.cool transcript
// framing; RFC 7539 block function with rand_chacha's 64-bit counter / 4-block buffer).
// Call sites anchored on: src/utils/transcript.rs:20-72 (Merlin use), src/poly/commitments.rs:22-44
// (SHAKE256 -> ChaCha20Rng generator derivation), src/utils/random.rs:15-30 (RandomTape).
// Pins: SHAKE256 vs Python hashlib; Merlin vs merlin's published "test protocol" vector
// (tests/test_oracle_hashes.py); ChaCha20 vs RFC 7539 §2.3.2 block vector.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include <string>
#include <cassert>

namespace orc {

inline uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

inline void keccak_f1600(uint64_t st[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
      0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
      0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  static const int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  static const int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  for (int round = 0; round < 24; round++) {
    uint64_t bc[5];
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      uint64_t t = bc[(i + 4) % 5] ^ rotl64(bc[(i + 1) % 5], 1);
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    uint64_t t = st[1];
    for (int i = 0; i < 24; i++) { int j = PILN[i]; uint64_t b = st[j]; st[j] = rotl64(t, ROTC[i]); t = b; }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= RC[round];
  }
}

// byte-addressed view of the little-endian lane state
struct KeccakState {
  uint64_t lanes[25];
  KeccakState() { memset(lanes, 0, sizeof(lanes)); }
  uint8_t get(int i) const { return (uint8_t)(lanes[i / 8] >> (8 * (i % 8))); }
  void xor_byte(int i, uint8_t b) { lanes[i / 8] ^= (uint64_t)b << (8 * (i % 8)); }
  void set(int i, uint8_t b) { lanes[i / 8] &= ~((uint64_t)0xff << (8 * (i % 8))); lanes[i / 8] |= (uint64_t)b << (8 * (i % 8)); }
  void permute() { keccak_f1600(lanes); }
};

struct Shake256 {
  KeccakState st; int pos = 0; bool squeezing = false;
  static const int RATE = 136;
  void absorb(const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) { st.xor_byte(pos++, d[i]); if (pos == RATE) { st.permute(); pos = 0; } }
  }
  void squeeze(uint8_t* out, size_t n) {
    if (!squeezing) { st.xor_byte(pos, 0x1f); st.xor_byte(RATE - 1, 0x80); st.permute(); pos = 0; squeezing = true; }
    for (size_t i = 0; i < n; i++) { if (pos == RATE) { st.permute(); pos = 0; } out[i] = st.get(pos++); }
  }
};

// STROBE-128 subset used by Merlin (strobe.rs in merlin 3.0.0)
struct Strobe128 {
  KeccakState st; uint8_t pos = 0, pos_begin = 0, cur_flags = 0;
  static const int R = 166;
  enum { FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };
  explicit Strobe128(const char* protocol_label) {
    const uint8_t init[6] = {1, R + 2, 1, 0, 1, 96};
    for (int i = 0; i < 6; i++) st.set(i, init[i]);
    const char* s = "STROBEv1.0.2";
    for (int i = 0; i < 12; i++) st.set(6 + i, (uint8_t)s[i]);
    st.permute();
    meta_ad((const uint8_t*)protocol_label, strlen(protocol_label), false);
  }
  void run_f() { st.xor_byte(pos, pos_begin); st.xor_byte(pos + 1, 0x04); st.xor_byte(R + 1, 0x80); st.permute(); pos = 0; pos_begin = 0; }
  void absorb(const uint8_t* d, size_t n) { for (size_t i = 0; i < n; i++) { st.xor_byte(pos, d[i]); pos++; if (pos == R) run_f(); } }
  void squeeze(uint8_t* d, size_t n) { for (size_t i = 0; i < n; i++) { d[i] = st.get(pos); st.set(pos, 0); pos++; if (pos == R) run_f(); } }
  void begin_op(uint8_t flags, bool more) {
    if (more) { assert(cur_flags == flags); return; }
    assert((flags & FLAG_T) == 0);
    uint8_t old_begin = pos_begin;
    pos_begin = pos + 1; cur_flags = flags;
    uint8_t hdr[2] = {old_begin, flags};
    absorb(hdr, 2);
    bool force_f = (flags & (FLAG_C | FLAG_K)) != 0;
    if (force_f && pos != 0) run_f();
  }
  void meta_ad(const uint8_t* d, size_t n, bool more) { begin_op(FLAG_M | FLAG_A, more); absorb(d, n); }
  void ad(const uint8_t* d, size_t n, bool more) { begin_op(FLAG_A, more); absorb(d, n); }
  void prf(uint8_t* d, size_t n, bool more) { begin_op(FLAG_I | FLAG_A | FLAG_C, more); squeeze(d, n); }
};

// merlin::Transcript (transcript.rs in merlin 3.0.0)
struct Transcript {
  Strobe128 strobe;
  explicit Transcript(const char* label) : strobe("Merlin v1.0") { append_message("dom-sep", (const uint8_t*)label, strlen(label)); }
  void append_message(const char* label, const uint8_t* msg, size_t n) {
    uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe.meta_ad((const uint8_t*)label, strlen(label), false);
    strobe.meta_ad(len, 4, true);
    strobe.ad(msg, n, false);
  }
  void append_message(const char* label, const char* msg) { append_message(label, (const uint8_t*)msg, strlen(msg)); }
  void append_u64(const char* label, uint64_t x) { uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i)); append_message(label, b, 8); }
  void challenge_bytes(const char* label, uint8_t* dest, size_t n) {
    uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe.meta_ad((const uint8_t*)label, strlen(label), false);
    strobe.meta_ad(len, 4, true);
    strobe.prf(dest, n, false);
  }
};

// rand_chacha 0.3 ChaChaXRng: 32-byte key, 64-bit block counter (words 12,13), 64-bit stream id = 0
// (words 14,15); results buffered 4 blocks (64 words) at a time; rand_core BlockRng index semantics.
struct ChaChaRng {
  uint32_t key[8]; uint64_t counter = 0; int rounds;
  uint32_t buf[64]; int index = 64;
  ChaChaRng(const uint8_t seed[32], int rounds_) : rounds(rounds_) {
    for (int i = 0; i < 8; i++) key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
  }
  static inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
  static void block(const uint32_t key[8], uint64_t ctr, int rounds, uint32_t out[16]) {
    uint32_t in[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    for (int i = 0; i < 8; i++) in[4 + i] = key[i];
    in[12] = (uint32_t)ctr; in[13] = (uint32_t)(ctr >> 32); in[14] = 0; in[15] = 0;
    uint32_t x[16]; memcpy(x, in, 64);
#define QR(a, b, c, d) x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 16); x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 12); \
  x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 8); x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 7);
    for (int r = 0; r < rounds; r += 2) {
      QR(0, 4, 8, 12) QR(1, 5, 9, 13) QR(2, 6, 10, 14) QR(3, 7, 11, 15)
      QR(0, 5, 10, 15) QR(1, 6, 11, 12) QR(2, 7, 8, 13) QR(3, 4, 9, 14)
    }
#undef QR
    for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
  }
  void generate() { for (int b = 0; b < 4; b++) block(key, counter + b, rounds, buf + 16 * b); counter += 4; }
  uint32_t next_u32() { if (index >= 64) { generate(); index = 0; } return buf[index++]; }
  uint64_t next_u64() {
    if (index < 63) { uint64_t v = (uint64_t)buf[index] | ((uint64_t)buf[index + 1] << 32); index += 2; return v; }
    if (index >= 64) { generate(); index = 2; return (uint64_t)buf[0] | ((uint64_t)buf[1] << 32); }
    uint64_t x = buf[63]; generate(); index = 1; uint64_t y = buf[0]; return (y << 32) | x;
  }
};

// ark_std::test_rng(): StdRng (= ChaCha12 in rand 0.8) from a fixed seed
inline ChaChaRng test_rng() {
  const uint8_t seed[32] = {1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  return ChaChaRng(seed, 12);
}

}  // namespace orc

// Host-side Fiat–Shamir plumbing of the product: Keccak-f[1600], SHAKE256, STROBE-128/Merlin, ChaCha RNG.
// The reference gets these from crates (merlin 3.0.0, sha3 0.8.2, rand_chacha 0.3, ark-std test_rng —
// Cargo.toml:29-33); call sites: src/utils/transcript.rs:20-72, src/poly/commitments.rs:22-44,
// src/utils/random.rs:15-30.  Sequential, O(rounds) work: it stays on the host by design.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>
#include <stdexcept>

namespace lasso {

class Keccak1600 {
 public:
  uint8_t bytes[200];
  Keccak1600() { memset(bytes, 0, sizeof(bytes)); }
  void permute() {
    uint64_t A[25];
    for (int i = 0; i < 25; i++) { uint64_t w = 0; for (int b = 7; b >= 0; b--) w = (w << 8) | bytes[8 * i + b]; A[i] = w; }
    static const uint64_t RC[24] = {0x1ULL, 0x8082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x808bULL, 0x80000001ULL, 0x8000000080008081ULL,
                                    0x8000000000008009ULL, 0x8aULL, 0x88ULL, 0x80008009ULL, 0x8000000aULL, 0x8000808bULL, 0x800000000000008bULL,
                                    0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x800aULL,
                                    0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x80000001ULL, 0x8000000080008008ULL};
    static const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int rnd = 0; rnd < 24; rnd++) {
      uint64_t Cc[5], D[5], Bm[25];
      for (int x = 0; x < 5; x++) Cc[x] = A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20];
      for (int x = 0; x < 5; x++) { uint64_t n = Cc[(x + 1) % 5]; D[x] = Cc[(x + 4) % 5] ^ ((n << 1) | (n >> 63)); }
      for (int i = 0; i < 25; i++) A[i] ^= D[i % 5];
      for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++) {  // rho + pi: B[y][2x+3y] = rot(A[x][y], r[x][y]); index = x + 5y
          int src = x + 5 * y, dst = y + 5 * ((2 * x + 3 * y) % 5), r = RHO[src];
          Bm[dst] = r ? ((A[src] << r) | (A[src] >> (64 - r))) : A[src];
        }
      for (int y = 0; y < 5; y++)
        for (int x = 0; x < 5; x++) A[x + 5 * y] = Bm[x + 5 * y] ^ (~Bm[(x + 1) % 5 + 5 * y] & Bm[(x + 2) % 5 + 5 * y]);
      A[0] ^= RC[rnd];
    }
    for (int i = 0; i < 25; i++) for (int b = 0; b < 8; b++) bytes[8 * i + b] = (uint8_t)(A[i] >> (8 * b));
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
  void absorb(const uint8_t* d, size_t n) { for (size_t i = 0; i < n; i++) { k.bytes[pos++] ^= d[i]; if (pos == RATE) run_f(); } }
  void squeeze(uint8_t* d, size_t n) { for (size_t i = 0; i < n; i++) { d[i] = k.bytes[pos]; k.bytes[pos++] = 0; if (pos == RATE) run_f(); } }
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
  void append_message(const char* label, const void* msg, size_t n) {
    uint8_t len[4]; le32((uint32_t)n, len);
    s.meta_ad(label, strlen(label), false); s.meta_ad(len, 4, true); s.ad(msg, n, false);
  }
  void append_str(const char* label, const char* msg) { append_message(label, msg, strlen(msg)); }
  void append_u64(const char* label, uint64_t x) { uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i)); append_message(label, b, 8); }
  void challenge_bytes(const char* label, uint8_t* out, size_t n) {
    uint8_t len[4]; le32((uint32_t)n, len);
    s.meta_ad(label, strlen(label), false); s.meta_ad(len, 4, true); s.prf(out, n, false);
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

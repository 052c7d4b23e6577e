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

// State kept as 25 little-endian 64-bit lanes (x86-64 host: the byte view aliases them); the round function is fully unrolled —
// the 4096-scalar vector appends of the opening proofs (dot_product.rs:196) push ~1 MB through STROBE per proof.
class Keccak1600 {
  static inline uint64_t rol(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }

 public:
  union { uint64_t A[25]; uint8_t bytes[200]; };
  Keccak1600() { memset(bytes, 0, sizeof(bytes)); }
  void permute() {
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
  // meta_ad(label, false); meta_ad(len4, true); ad(msg, false) — merlin's append_message — in ONE pass over the rate block when the framed record (2 + label + 4 + 2 + msg
  // bytes) ends before the block does; false = a block boundary falls inside it and the caller takes the three calls.  The openings append their 16 640 a-vector scalars one
  // by one (dot_product.rs:196): three out of four of those records take this path.
  bool append_framed(const void* label, size_t ll, const uint8_t len4[4], const void* msg, size_t n) {
    const size_t total = 2 + ll + 4 + 2 + n;
    if ((size_t)pos + total >= RATE) return false;
    uint8_t* d = k.bytes + pos;
    d[0] ^= pos_begin; d[1] ^= (uint8_t)(M | A);                       // begin(M | A): header = [previous pos_begin, flags], pos_begin = pos + 1
    const uint8_t* l = (const uint8_t*)label; for (size_t i = 0; i < ll; i++) d[2 + i] ^= l[i];
    uint8_t* e = d + 2 + ll;
    e[0] ^= len4[0]; e[1] ^= len4[1]; e[2] ^= len4[2]; e[3] ^= len4[3];
    e[4] ^= (uint8_t)(pos + 1); e[5] ^= (uint8_t)A;                    // begin(A): header = [pos_begin of the meta op, flags]
    const uint8_t* m = (const uint8_t*)msg; uint8_t* f = e + 6; size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t a, b; memcpy(&a, f + i, 8); memcpy(&b, m + i, 8); a ^= b; memcpy(f + i, &a, 8); }
    for (; i < n; i++) f[i] ^= m[i];
    pos_begin = (uint8_t)(pos + 2 + ll + 4 + 1); cur = A; pos = (uint8_t)(pos + total);
    return true;
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
    if (s.append_framed(label, label_len, len, msg, n)) return;
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

// EXPERIMENT (tools/, not part of the product): Fiat-Shamir on the device — Keccak-f[1600], STROBE-128 and the Merlin framing (merlin ^3.0.0 —
// Cargo.toml:29; call sites src/utils/transcript.rs:20-72) with the 200-byte state spread over 25 LANES, one 64-bit word each.
// The idea: the proof is latency-bound by ~470 sequential transcript rounds, each a device -> host -> device turn of ~12 us; with
// the transcript next to the data the last workgroup of a round could derive the next challenge itself.  MEASURED on MI355X and dropped
// (profiles/r02_transcript_bench_device_fiat_shamir.txt): the lane-distributed transcript matches the host's bytes, but one sumcheck-shaped round
// (3 appends + 1 challenge = 4-5 permutations) costs 19.3 us on one wave — each Keccak round is four dependent cross-lane exchanges — against ~2 us of host
// Keccak inside the 12 us turn it would replace.  The host keeps the transcript.  Host-checked by tests/cpp/test_transcript_dev_host.cpp against the host
// transcript (lasso_amd/host/hashes.hpp): the permutation, the STROBE operations Merlin uses (meta-AD, AD, PRF), append_message /
// challenge_bytes, and the 64-byte -> Fr reduction of challenge_scalar (utils/transcript.rs:61-65).
//
// One body, two lane backends.  Everything is written against a backend L that owns "a value per lane" (L::V) and can
//   gather(v, src)   — every lane l receives the value lane src(l) holds      (device: two __shfl per 64-bit word; host: an index loop)
//   map(f, ...)      — lane-wise arithmetic with the lane index available
// so the device and the host emulation execute the SAME index arithmetic (which lane reads which, the rho offsets, where a byte of the rate
// block lives); only the exchange primitive differs.  Lanes 25..63 of the wave idle; a permutation is 24 x (9 gathers + ~20 lane operations).
#pragma once
#include <stdint.h>
#include "fr.cuh"

#define STROBE_RATE 166u
enum : uint8_t { STROBE_I = 1, STROBE_A = 2, STROBE_C = 4, STROBE_T = 8, STROBE_M = 16, STROBE_K = 32 };

LHD uint64_t keccak_rc(int rnd) {
  const uint64_t RC[24] = {0x1ULL, 0x8082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x808bULL, 0x80000001ULL, 0x8000000080008081ULL,
                           0x8000000000008009ULL, 0x8aULL, 0x88ULL, 0x80008009ULL, 0x8000000aULL, 0x8000808bULL, 0x800000000000008bULL,
                           0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x800aULL,
                           0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x80000001ULL, 0x8000000080008008ULL};
  return RC[rnd];
}
// rho offset of lane l = x + 5y
LHD uint32_t keccak_rho(uint32_t l) {
  const uint8_t R[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  return R[l];
}
LHD uint64_t rol64(uint64_t v, uint32_t r) { return r ? (v << r) | (v >> (64 - r)) : v; }
// lane index helpers (l = x + 5y)
LHD uint32_t kl_row_shift(uint32_t l, uint32_t dx) { const uint32_t x = l % 5u; return l - x + (x + dx) % 5u; }   // same row, column x + dx
LHD uint32_t kl_col_shift(uint32_t l, uint32_t dy) { return (l + 5u * dy) % 25u; }                                 // same column, row y + dy
LHD uint32_t kl_pi_src(uint32_t l) { const uint32_t X = l % 5u, Y = l / 5u; return (X + 3u * Y) % 5u + 5u * X; }   // B[X][Y] = rot(A[x][y]) with X = y, Y = 2x + 3y

template <class L>
LHD void keccak_f1600_lanes(L& ln, typename L::V& w) {
  typedef typename L::V V;
  for (int rnd = 0; rnd < 24; rnd++) {
    // theta: column parity (every lane of a column computes its own copy), D = C[x-1] ^ rot(C[x+1], 1)
    V c = w;
    for (uint32_t dy = 1; dy < 5; dy++) { const V t = ln.gather(w, [dy](uint32_t l) { return kl_col_shift(l, dy); }); c = ln.map2(c, t, [](uint32_t, uint64_t a, uint64_t b) { return a ^ b; }); }
    const V cp = ln.gather(c, [](uint32_t l) { return kl_row_shift(l, 4); }), cn = ln.gather(c, [](uint32_t l) { return kl_row_shift(l, 1); });
    w = ln.map3(w, cp, cn, [](uint32_t, uint64_t a, uint64_t p, uint64_t n) { return a ^ p ^ rol64(n, 1); });
    // rho (own offset), pi (one gather)
    const V r = ln.map1(w, [](uint32_t l, uint64_t a) { return rol64(a, keccak_rho(l)); });
    const V b = ln.gather(r, [](uint32_t l) { return kl_pi_src(l); });
    // chi along the row, iota on lane 0
    const V b1 = ln.gather(b, [](uint32_t l) { return kl_row_shift(l, 1); }), b2 = ln.gather(b, [](uint32_t l) { return kl_row_shift(l, 2); });
    const uint64_t rc = keccak_rc(rnd);
    w = ln.map3(b, b1, b2, [rc](uint32_t l, uint64_t a, uint64_t n1, uint64_t n2) { return a ^ (~n1 & n2) ^ (l == 0 ? rc : 0ull); });
  }
}

// ---- STROBE-128 as Merlin uses it, on the lane-distributed state.  Byte p of the state is byte p % 8 of lane p / 8.
// The control variables (pos, pos_begin, cur) are uniform across the lanes.
LHD uint64_t strobe_lane_mask(uint32_t lane, uint32_t pos, const uint8_t* data, uint32_t n) {   // the bytes of data[0..n) that land in this lane when absorbed at pos
  uint64_t m = 0;
  for (uint32_t k = 0; k < 8; k++) { const uint32_t g = 8u * lane + k; if (g >= pos && g < pos + n) m |= (uint64_t)data[g - pos] << (8u * k); }
  return m;
}
LHD uint64_t strobe_lane_keep(uint32_t lane, uint32_t pos, uint32_t n) {   // all-ones bytes OUTSIDE [pos, pos + n)
  uint64_t m = 0;
  for (uint32_t k = 0; k < 8; k++) { const uint32_t g = 8u * lane + k; if (!(g >= pos && g < pos + n)) m |= 0xffull << (8u * k); }
  return m;
}

template <class L>
struct strobe_lanes {
  typename L::V w;
  uint32_t pos, pos_begin, cur;

  LHD void xor_byte(L& ln, uint32_t idx, uint8_t v) { const uint8_t d[1] = {v}; w = ln.map1(w, [idx, &d](uint32_t l, uint64_t a) { return a ^ strobe_lane_mask(l, idx, d, 1); }); }
  LHD void run_f(L& ln) {
    xor_byte(ln, pos, (uint8_t)pos_begin); xor_byte(ln, pos + 1, 0x04); xor_byte(ln, STROBE_RATE + 1, 0x80);
    keccak_f1600_lanes(ln, w);
    pos = 0; pos_begin = 0;
  }
  LHD void absorb(L& ln, const uint8_t* d, uint32_t n) {
    while (n) {
      uint32_t chunk = STROBE_RATE - pos; if (chunk > n) chunk = n;
      const uint32_t p = pos;
      w = ln.map1(w, [p, d, chunk](uint32_t l, uint64_t a) { return a ^ strobe_lane_mask(l, p, d, chunk); });
      pos += chunk; d += chunk; n -= chunk;
      if (pos == STROBE_RATE) run_f(ln);
    }
  }
  // squeeze: the bytes come out of the lanes into `out` (shared / local memory every lane can write) and are zeroed in the state
  LHD void squeeze(L& ln, uint8_t* out, uint32_t n) {
    while (n) {
      uint32_t chunk = STROBE_RATE - pos; if (chunk > n) chunk = n;
      const uint32_t p = pos;
      ln.scatter_bytes(w, p, out, chunk);
      w = ln.map1(w, [p, chunk](uint32_t l, uint64_t a) { return a & strobe_lane_keep(l, p, chunk); });
      pos += chunk; out += chunk; n -= chunk;
      if (pos == STROBE_RATE) run_f(ln);
    }
  }
  LHD void begin(L& ln, uint8_t flags, bool more) {
    if (more) return;   // a continued operation keeps its flags (the host form checks cur == flags)
    const uint8_t hdr[2] = {(uint8_t)pos_begin, flags};
    pos_begin = pos + 1; cur = flags;
    absorb(ln, hdr, 2);
    if ((flags & (STROBE_C | STROBE_K)) && pos != 0) run_f(ln);
  }
  LHD void meta_ad(L& ln, const uint8_t* d, uint32_t n, bool more) { begin(ln, STROBE_M | STROBE_A, more); absorb(ln, d, n); }
  LHD void ad(L& ln, const uint8_t* d, uint32_t n, bool more) { begin(ln, STROBE_A, more); absorb(ln, d, n); }
  LHD void prf(L& ln, uint8_t* d, uint32_t n, bool more) { begin(ln, STROBE_I | STROBE_A | STROBE_C, more); squeeze(ln, d, n); }

  // Strobe128::new(b"Merlin v1.0") followed by Transcript::new(label)'s dom-sep message
  LHD void init_merlin(L& ln, const uint8_t* label, uint32_t label_len) {
    const uint8_t head[18] = {1, (uint8_t)(STROBE_RATE + 2), 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    w = ln.zero();
    w = ln.map1(w, [&head](uint32_t l, uint64_t a) { return a ^ strobe_lane_mask(l, 0, head, 18); });
    keccak_f1600_lanes(ln, w);
    pos = 0; pos_begin = 0; cur = 0;
    const uint8_t proto[11] = {'M', 'e', 'r', 'l', 'i', 'n', ' ', 'v', '1', '.', '0'};
    meta_ad(ln, proto, 11, false);
    const uint8_t ds[7] = {'d', 'o', 'm', '-', 's', 'e', 'p'};
    append_message(ln, ds, 7, label, label_len);
  }
  LHD void append_message(L& ln, const uint8_t* label, uint32_t label_len, const uint8_t* msg, uint32_t n) {
    const uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    meta_ad(ln, label, label_len, false); meta_ad(ln, len, 4, true); ad(ln, msg, n, false);
  }
  LHD void challenge_bytes(L& ln, const uint8_t* label, uint32_t label_len, uint8_t* out, uint32_t n) {
    const uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    meta_ad(ln, label, label_len, false); meta_ad(ln, len, 4, true); prf(ln, out, n, false);
  }
};

// utils/transcript.rs:61-65 challenge_scalar: 64 challenge bytes -> F::from_le_bytes_mod_order = (lo + hi 2^256) mod p, in Montgomery form.
// Montgomery products accept any 256-bit left operand: lo * R^2 / R = lo R, hi * R^3 / R = hi 2^256 R.
LHD fr_t fr_from_wide_bytes(const uint8_t* b) {
  fr_t lo, hi, r3;
  for (int i = 0; i < 8; i++) {
    lo.v[i] = (uint32_t)b[4 * i] | (uint32_t)b[4 * i + 1] << 8 | (uint32_t)b[4 * i + 2] << 16 | (uint32_t)b[4 * i + 3] << 24;
    hi.v[i] = (uint32_t)b[32 + 4 * i] | (uint32_t)b[32 + 4 * i + 1] << 8 | (uint32_t)b[32 + 4 * i + 2] << 16 | (uint32_t)b[32 + 4 * i + 3] << 24;
  }
#ifdef LASSO_BN254
  const uint32_t R3[8] = {0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu, 0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu};   // 2^768 mod p
#else
  const uint32_t R3[8] = {0x7b83a2dbu, 0x2a9e4968u, 0xaef7f3ecu, 0x278324e6u, 0x04ec5b65u, 0x8065dc6cu, 0x3599cec7u, 0x0e530b77u};   // 2^768 mod p
#endif
  for (int i = 0; i < 8; i++) r3.v[i] = R3[i];
  return fr_add(fr_mul(lo, fr_r2()), fr_mul(hi, r3));
}

#if defined(__HIPCC__)
// Device backend: the 25 lanes are lanes 0..24 of a wave; a 64-bit gather is two 32-bit shuffles.  Lanes >= 25 carry dead values.
struct wave_lanes {
  typedef uint64_t V;
  uint32_t lane;
  __device__ explicit wave_lanes() : lane(threadIdx.x & 63u) {}
  __device__ V zero() const { return 0; }
  template <class F> __device__ V gather(const V& v, F src) const {
    const int s = (int)src(lane < 25u ? lane : 0u);
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, s, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), s, 64);
    return (uint64_t)hi << 32 | lo;
  }
  template <class F> __device__ V map1(const V& a, F f) const { return f(lane < 25u ? lane : 0u, a); }
  template <class F> __device__ V map2(const V& a, const V& b, F f) const { return f(lane < 25u ? lane : 0u, a, b); }
  template <class F> __device__ V map3(const V& a, const V& b, const V& c, F f) const { return f(lane < 25u ? lane : 0u, a, b, c); }
  __device__ void scatter_bytes(const V& v, uint32_t pos, uint8_t* out, uint32_t n) const {   // out: LDS or global memory visible to the wave
    if (lane < 25u) for (uint32_t k = 0; k < 8; k++) { const uint32_t g = 8u * lane + k; if (g >= pos && g < pos + n) out[g - pos] = (uint8_t)(v >> (8u * k)); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
};
#endif

// HIP kernels for the polynomial half of the Lasso hot path (gfx950, wave64).
// Memory holds 32-byte Fr elements exactly as a Rust Vec<Fr> (two dwordx4 per lane); arithmetic runs in the carry-free 29-bit-limb
// form of fr29.cuh (unpack on load, canonicalise + pack on store).  Streaming kernels are HBM-bound by design; the sumcheck round
// evaluators are integer-ALU-bound.  Sums are exact field sums, so any reduction order is bit-identical to the reference's loops.
#pragma once
#define LASSO_MAIL_POISON 0xFFFFFFFFu   // mailbox tag written by lasso_abort: resident kernels waiting for a challenge leave at once (sequence tags never reach it)
#include <hip/hip_runtime.h>
#include "fr.cuh"
#include "fr29.cuh"

#ifndef LASSO_MAX_PTRS
#define LASSO_MAX_PTRS 136   // 2 * (2 * 33 memories) + slack; pointer tables travel by value in the kernarg segment
#endif
#define LASSO_BLOCK 256
#define LASSO_MAX_ALPHA 32

struct PtrTable { const fr_t* p[LASSO_MAX_PTRS]; };
struct MutPtrTable { fr_t* p[LASSO_MAX_PTRS]; };
// the same for <= 8 arrays: 64 bytes of kernel arguments instead of 1088.  The cubic round kernels are templated on the table type — a launch's arguments are written through the
// BAR before the dispatch, and two full tables plus an unused EqInline were 2.7 KB per launch (18.54 -> 18.34 ms per proof with 16-entry tables, profiles/r03_ab_kernarg_tables.txt)
struct PtrTable8 { const fr_t* p[8]; };
struct MutPtrTable8 { fr_t* p[8]; };
struct StrategyDev { int kind; uint32_t c, log_m, log_r, alpha; };
struct WeightTable { fr_t w[LASSO_MAX_ALPHA]; };

// ------------------------------------------------------------------ reductions
// Block-wide sum of up to 3 field values per thread WITHOUT carries: every thread parks its 27 limbs in LDS, 216 threads add 32-row
// strips of one limb column each into 64-bit sums, 27 threads finish the columns.  (256 limbs of < 2^30 fit a 64-bit column with room to
// spare — the point of the 29-bit form.)  cols[v*9 + k] is valid for all threads after the call; fr29_from_columns turns nine columns
// back into a field value.  Input limbs must be reduced (fr29_weak) with |limb 8| < 2^30.  Requires blockDim.x == 256.
struct RedScratch { int32_t rows[LASSO_BLOCK * 27]; int64_t strips[8 * 27]; int64_t cols[27]; };
template <int NV>
__device__ __forceinline__ void block_columns(const fr29* vals, RedScratch& S) {
  const uint32_t t = threadIdx.x;
  __syncthreads();   // any previous use of S is over
#pragma unroll
  for (int v = 0; v < NV; v++)
#pragma unroll
    for (int k = 0; k < 9; k++) S.rows[t * 27 + v * 9 + k] = vals[v].v[k];
  __syncthreads();
  if (t < 8 * 27) {
    const uint32_t j = t % 27, g = t / 27;
    int64_t s = 0;
    if (j < NV * 9) {
#pragma unroll 8
      for (uint32_t r = 0; r < 32; r++) s += S.rows[(g * 32 + r) * 27 + j];
    }
    S.strips[g * 27 + j] = s;
  }
  __syncthreads();
  if (t < 27) {
    int64_t s = 0;
#pragma unroll
    for (int g = 0; g < 8; g++) s += S.strips[g * 27 + t];
    S.cols[t] = s;
  }
  __syncthreads();
}
// value of column group v after block_columns, times 2^shift (0 to only reduce; 5 / 10 also correct the radix of sums of u*u / u*u*u products:
// what a Montgomery product with K5 = 2^266 / K10 = 2^271 would do), as memory words
__device__ __forceinline__ fr_t columns_to_fr(const RedScratch& S, int v, int shift) {
  int64_t c[9];
#pragma unroll
  for (int k = 0; k < 9; k++) c[k] = S.cols[v * 9 + k];
  return fr29_pack(fr29_reduce_columns(c, shift));
}
// per-thread accumulator step: acc += t, kept reduced; every 128 terms the magnitude is folded back (limb 8 must stay below 2^30)
__device__ __forceinline__ void acc_add(fr29& acc, const fr29& t, uint32_t& count) {
  acc = fr29_weak(fr29_add(acc, t));
  if ((++count & 127u) == 0) acc = fr29_mul(acc, fr29_one_s());
}
// ------------------------------------------------------------------ results for the host
// Two ways a launch hands its few field elements to the host, chosen by the `flag` argument every round kernel takes:
//  * flag = a word of host-mapped memory: elements stored to out[slot] (host-mapped), then row_done: system-scope fence, an agent-scope ticket over the
//    grid rows, and the row that arrives last stores the sequence number to the flag.  4.1 us from the host's word to the host seeing the answer for a
//    resident workgroup (tools/handoff_bench.hip);
//  * flag = LASSO_TAGGED: `out` is an area of SELF-VALIDATING 16-byte chunks, three per element: [seq, w0, w1, w2] [seq, w3, w4, w5] [seq, w6, w7, check].
//    Every row stores its own chunks (one aligned dwordx4 each = one PCIe write) and releases them with ONE system-scope fence: no ticket, no flag store, no
//    cross-row ordering.  The host accepts an element once its three chunks carry the hand-off's sequence number (unique for the life of the context) and the
//    check word matches.  2.1-2.3 us for the same turn.
#define LASSO_TAGGED (reinterpret_cast<uint32_t*>(uintptr_t(16)))
// the same area, and EVERY workgroup of a row publishes its own block sums under slot (row * nx + bx) * K + k: the host adds the nx of them (lasso_hip.hip wait_flag) — for launches
// of a few workgroups per row, where the in-launch second stage (agent-scope release, ticket, acquire, re-read, second block reduction) is a third of the kernel's time
#define LASSO_TAGGED_DIRECT (reinterpret_cast<uint32_t*>(uintptr_t(32)))
typedef uint32_t lasso_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t result_check(const fr_t& v, uint32_t seq) { return (v.v[0] ^ v.v[1] ^ v.v[2] ^ v.v[3] ^ v.v[4] ^ v.v[5] ^ v.v[6] ^ v.v[7]) + seq * 0x9E3779B9u; }
__device__ __forceinline__ void result_store(fr_t* __restrict__ out, size_t slot, const fr_t& v, uint32_t* flag, uint32_t seq) {
  if (flag == LASSO_TAGGED || flag == LASSO_TAGGED_DIRECT) {
    lasso_u32x4* o = reinterpret_cast<lasso_u32x4*>(out) + 3 * slot;
    const lasso_u32x4 c0 = {seq, v.v[0], v.v[1], v.v[2]}, c1 = {seq, v.v[3], v.v[4], v.v[5]}, c2 = {seq, v.v[6], v.v[7], result_check(v, seq)};
    o[0] = c0; o[1] = c1; o[2] = c2;
  } else {
    // A value another workgroup of THIS launch will read (block partials on their way to last_block_reduce): write-through (sc1) 8-byte stores, so that what publishes them
    // is the storing wave's `s_waitcnt vmcnt(0)` and not an L2 write-back (round 6; cdna_hip_programming.md §6 G16, the sc1 form).  The release fence this replaces wrote back
    // EVERY dirty line of the XCD's L2 — and every workgroup of a fused round has just dirtied 32 KB of bound values there: 512 such fences per launch.
    // (flag != nullptr: the flag protocol's result area in host-mapped memory, LASSO_TAGGED_RESULTS=0 — plain stores, published by row_done's system-scope fence as before)
#ifndef LASSO_PLAIN_PARTIALS
    if (flag == nullptr) {
      uint64_t* o = reinterpret_cast<uint64_t*>(out + slot);
#pragma unroll
      for (int k = 0; k < 4; k++) __hip_atomic_store(o + k, (uint64_t)v.v[2 * k] | ((uint64_t)v.v[2 * k + 1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
#endif
    out[slot] = v;
  }
}
// the host -> device direction of a resident kernel (lasso_hip.hip post_mail): the three mailbox chunks carry this tag and the check word of their eight challenge words
__device__ __forceinline__ bool mail_valid(const lasso_u32x4& c0, const lasso_u32x4& c1, const lasso_u32x4& c2, uint32_t tag) {
  return c0.x == tag && c1.x == tag && c2.x == tag && c2.w == (c0.y ^ c0.z ^ c0.w ^ c1.y ^ c1.z ^ c1.w ^ c2.y ^ c2.z) + tag * 0x9E3779B9u;
}
// ------------------------------------------------------------------ challenges that arrive while the kernel is already running
// One poll loop of the host-mapped mailbox (lasso_hip.hip post_mail: three self-validating 16-byte chunks [tag, w, w, w] [tag, w, w, w] [tag, w, w, check]; the host writes each with one
// aligned 16-byte store and a PCIe read of an aligned 16 bytes is one transaction, so when all three carry the expected tag and the check word agrees the eight words are that message's).
// Returns false on lasso_abort's poison tag or when the wall clock passes t_end (a host that never answers cannot hang the device).  ONE lane calls it.
__device__ __forceinline__ bool mail_wait(const uint32_t* mailbox, uint32_t tag, uint64_t t_end, fr_t& chal) {   // no __restrict__ on the mailbox: the host writes it while the kernel polls
  const lasso_u32x4* m4 = reinterpret_cast<const lasso_u32x4*>(mailbox);
  lasso_u32x4 c0, c1, c2; uint32_t spins = 0;
  for (;;) {
#ifndef LASSO_NO_POLL_FENCE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // nothing cached from the previous poll
#else
    asm volatile("" ::: "memory");
#endif
    c0 = __builtin_nontemporal_load(m4); c1 = __builtin_nontemporal_load(m4 + 1); c2 = __builtin_nontemporal_load(m4 + 2);   // three 16-byte reads in flight together
    if (mail_valid(c0, c1, c2, tag)) break;
    if (c0.x == LASSO_MAIL_POISON || ((++spins & 63u) == 0 && wall_clock64() > t_end)) return false;
  }
  chal.v[0] = c0.y; chal.v[1] = c0.z; chal.v[2] = c0.w; chal.v[3] = c1.y; chal.v[4] = c1.z; chal.v[5] = c1.w; chal.v[6] = c2.y; chal.v[7] = c2.z;
  return true;
}
// A round LAUNCHED AHEAD of its challenge (round 5): the host enqueues round j + 1 behind round j BEFORE it has round j's sums, so that neither the launch nor its dispatch sits
// between two rounds (12 us of host turn per launched round, of which ~1.5 us are the Fiat-Shamir step), and posts the challenge into the host-mapped mailbox under the launch's own
// sequence number when it has it.  The wait is a GATE: a one-wave kernel (k_gate) enqueued in front of the round polls the mailbox and leaves the scalar(s) in device memory —
// gmail[0..8) (and [8..16) for the openings' pair u, u^-1), then gmail[16] = seq; gmail[17] = seq instead means "no challenge" (lasso_abort's poison tag / 5 s without a post) —
// and the round's kernel, which the stream starts when the gate ends, reads them there (gated_challenge).  Measured on the way (profiles/r05_ahead_wait_forms.txt): with the wait
// INSIDE the round's kernel (workgroup 0 polling, the others spinning on gmail[16]) a launch lost 40 ns per waiting workgroup — 22 us at 512 workgroups, more than the launch it saved —
// to 512 lanes hammering one cache line; queued kernels follow each other without a measurable gap, so the gate costs nothing of the kind and occupies one wave, not the chip.
__global__ void __launch_bounds__(64) k_gate(const uint32_t* mail, uint32_t* gmail, uint32_t seq, uint32_t pair) {   // no __restrict__: the host writes the mailbox while this polls
  if (threadIdx.x != 0) return;
  const uint64_t t_end = wall_clock64() + 500000000ull;   // 5 s at 100 MHz
  fr_t c0, c1;
  bool ok = mail_wait(mail, seq, t_end, c0);
  if (ok && pair) ok = mail_wait(mail + 12, seq, t_end, c1);
  if (ok) {
#pragma unroll
    for (int k = 0; k < 8; k++) { gmail[k] = c0.v[k]; if (pair) gmail[8 + k] = c1.v[k]; }
    gmail[16] = seq;
  } else gmail[17] = seq;
}
// The same wait INSIDE the round's kernel, for launches of a few workgroups (<= 32: profiles/r05_ahead_wait_forms.txt — there the spinning costs less than the gate's second launch
// costs the host, 14.8 against 18.2 us per round at 8 workgroups; from 64 workgroups on the gate wins): workgroup 0 — dispatched first — polls the host-mapped mailbox and
// republishes the scalar in gmail[0..8) behind the tag gmail[16]; the others wait on that tag.  gmail[17] = tag: no challenge (poison / 5 s), everybody leaves.  s_mail = 9 words.
__device__ __forceinline__ bool ahead_challenge(const uint32_t* mail, uint32_t* gmail, uint32_t seq, fr_t& r, uint32_t* s_mail) {
  if (threadIdx.x == 0) {
    const uint64_t t_end = wall_clock64() + 500000000ull;   // 5 s at 100 MHz
    uint32_t ok = 1, spins = 0;
    if (blockIdx.x == 0) {
      fr_t c;
      if (mail_wait(mail, seq, t_end, c)) {
#pragma unroll
        for (int k = 0; k < 8; k++) gmail[k] = c.v[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(gmail + 16, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else __hip_atomic_store(gmail + 17, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (;;) {
      if (__hip_atomic_load(gmail + 16, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == seq) break;
      if (__hip_atomic_load(gmail + 17, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == seq || ((++spins & 63u) == 0 && wall_clock64() > t_end + 100000000ull)) { ok = 0; break; }
      __builtin_amdgcn_s_sleep(8);   // ~0.2 us between looks (the waiting workgroups must not crowd the poller's L2 traffic)
    }
    if (ok) {
#pragma unroll
      for (int k = 0; k < 8; k++) s_mail[k] = __hip_atomic_load(gmail + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_mail[8] = ok;
  }
  __syncthreads();
  if (!s_mail[8]) return false;
#pragma unroll
  for (int k = 0; k < 8; k++) r.v[k] = s_mail[k];
  return true;
}
// every workgroup of the gated launch: the scalar the gate left (false: this launch gets no challenge and must end without a result).  s_mail = 17 words of LDS.
__device__ __forceinline__ bool gated_challenge(const uint32_t* gmail, uint32_t seq, uint32_t words, uint32_t* s_mail) {
  if (threadIdx.x < 17) {
    const uint32_t t = threadIdx.x;
    if (t < words) s_mail[t] = gmail[t];
    if (t == 16) s_mail[16] = (gmail[16] == seq && gmail[17] != seq) ? 1u : 0u;
  }
  __syncthreads();
  return s_mail[16] != 0;
}
// A LAYER's first launch enqueued ahead of the layer's eq point (round 5): the host enqueues the next grand-product layer's round 0 (or its resident tail) while the current layer's
// last kernel is still running, and posts the point — the current layer's challenges and 1 - r_layer, known only after the layer's last Fiat-Shamir step — when it has it.  The
// message is ell + 2 field elements (point[0..ell), scale, control: word 0 != 0 cancels) in a host-mapped area of one three-chunk mailbox entry each (mail_wait's format), all
// under the launch's sequence number; ONE wave waits for it, lane j for entry j, and leaves the elements in device memory: gpoint[8 j ..), then gpoint[LASSO_GP_TAG] = seq —
// or gpoint[LASSO_GP_TAG + 1] = seq: cancelled, poisoned (lasso_abort) or 5 s without a post, and the kernels behind the gate end without a result.
#define LASSO_POINT_MAX 40
#define LASSO_GP_TAG (8 * (LASSO_POINT_MAX + 2))
#define LASSO_PMAIL_BYTES 2048    // (LASSO_POINT_MAX + 2) entries of 48 bytes
#define LASSO_GPOINT_BYTES 2048   // (LASSO_GP_TAG + 2) words
#define LASSO_PMAIL_ACK_WORD 504   // the first spare word behind the 42 entries: the gate leaves its sequence number there when it ENDS (message consumed, cancelled or timed out), so the
                                  // host knows the one mailbox area is free for the next gate's message (cancel -> begin -> post could overwrite a message no gate had read yet)
__global__ void __launch_bounds__(64) k_gate_point(const uint32_t* pmail, uint32_t* gpoint, uint32_t seq, uint32_t nfr, uint32_t* ack) {   // no __restrict__: the host writes pmail while this polls
  const uint32_t t = threadIdx.x;
  const uint64_t t_end = wall_clock64() + 500000000ull;   // 5 s at 100 MHz
  bool ok = true; fr_t v = fr_zero();
  if (t < nfr) {
    ok = mail_wait(pmail + 12 * t, seq, t_end, v);
    if (ok) {
#pragma unroll
      for (int k = 0; k < 8; k++) gpoint[8 * t + k] = v.v[k];
      if (t == nfr - 1 && v.v[0] != 0) ok = false;   // the control entry: cancelled by the host
    }
  }
  const bool all_ok = __ballot(!ok) == 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if (t == 0) { gpoint[LASSO_GP_TAG + (all_ok ? 0 : 1)] = seq; if (ack) __hip_atomic_store(ack, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
__device__ __forceinline__ bool gate_point_ok(const uint32_t* __restrict__ gp, uint32_t seq) { return gp[LASSO_GP_TAG] == seq && gp[LASSO_GP_TAG + 1] != seq; }
__device__ __forceinline__ fr_t gate_point_fr(const uint32_t* __restrict__ gp, uint32_t j) {
  fr_t r;
#pragma unroll
  for (int k = 0; k < 8; k++) r.v[k] = gp[8 * j + k];
  return r;
}
// block partials of up to KMAX accumulators (groups of 3) -> dst[k], k < K; `shift` also corrects the radix of the accumulated products.
// With `flag` the destination is the launch's result area (result_store at slot0 + k).
template <int KMAX>
__device__ __forceinline__ void store_block_partials(const fr29* acc, uint32_t K, fr_t* __restrict__ dst, int shift, RedScratch& S, uint32_t* flag = nullptr, uint32_t seq = 0, size_t slot0 = 0) {
#pragma unroll
  for (int k0 = 0; k0 < KMAX; k0 += 3) if ((uint32_t)k0 < K) {
    fr29 grp[3];
#pragma unroll
    for (int v = 0; v < 3; v++) grp[v] = (k0 + v < KMAX) ? acc[(k0 + v < KMAX) ? k0 + v : 0] : fr29_zero();
    block_columns<3>(grp, S);
    if (threadIdx.x < 3 && k0 + threadIdx.x < K) result_store(dst, slot0 + k0 + threadIdx.x, columns_to_fr(S, threadIdx.x, shift), flag, seq);
  }
}

// second stage: out[y*K + k] = sum_x partials[(y*nx + x)*K + k]   (partials are canonical memory words)
// K = row stride of partials/out, Kv <= K = number of valid values in this row
__device__ __forceinline__ void reduce_partials_row(const fr_t* __restrict__ partials, uint32_t nx, uint32_t K, uint32_t y, fr_t* __restrict__ out, RedScratch& S, uint32_t Kv = 0xffffffffu,
                                                    uint32_t* flag = nullptr, uint32_t seq = 0) {
  if (Kv > K) Kv = K;
  for (uint32_t k0 = 0; k0 < Kv; k0 += 3) {
    fr29 acc[3];
#pragma unroll
    for (int v = 0; v < 3; v++) {
      acc[v] = fr29_zero();
      if (k0 + v < Kv) for (uint32_t x = threadIdx.x; x < nx; x += blockDim.x) acc[v] = fr29_weak(fr29_add(acc[v], fr29_unpack_u(partials[((size_t)y * nx + x) * K + k0 + v])));
    }
    block_columns<3>(acc, S);
    if (threadIdx.x < 3 && k0 + threadIdx.x < Kv) result_store(out, (size_t)y * K + k0 + threadIdx.x, columns_to_fr(S, threadIdx.x, 0), flag, seq);
  }
}
__global__ void __launch_bounds__(LASSO_BLOCK) k_reduce_partials(const fr_t* __restrict__ partials, uint32_t nx, uint32_t K, fr_t* __restrict__ out) {
  __shared__ RedScratch S;
  reduce_partials_row(partials, nx, K, blockIdx.x, out, S);
}

// In-launch second stage (no extra kernel per round).  Block `bx` of row `y` has written its K partial sums (threads 0..2, one wave).
// The last block to arrive for that row sums all nx partials and writes out[y*K + k].  Hand-off follows cdna_hip_programming.md §6 G16:
// plain stores -> agent-scope release -> drained vmcnt -> relaxed agent atomic ticket; the last arriver does ONE agent-scope acquire,
// then reads plain.  `out` is host-mapped memory.  The row that finishes last raises the host's sequence flag: every row's stores are
// released at system scope before its ticket, so the flag store (system-scope release) is ordered after all of them.
__device__ __forceinline__ void row_done(uint32_t nrows, uint32_t* counters, uint32_t* flag, uint32_t seq);
__device__ __forceinline__ void last_block_reduce(const fr_t* partials, uint32_t nx, uint32_t K, uint32_t y, uint32_t nrows, uint32_t* counters, fr_t* __restrict__ out, RedScratch& S,
                                                  uint32_t* flag, uint32_t seq, uint32_t Kv = 0xffffffffu) {
  __shared__ uint32_t is_last;
  if (threadIdx.x == 0) {   // wave 0 holds the partial stores (sc1: result_store); vmcnt is wave-wide
#ifdef LASSO_PLAIN_PARTIALS
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t ticket = __hip_atomic_fetch_add(&counters[y], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t last = (ticket == nx - 1) ? 1u : 0u;
    if (last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); counters[y] = 0; }   // reset for the next launch (ordered by the kernel boundary)
    is_last = last;
  }
  __syncthreads();
  if (!is_last) return;
  reduce_partials_row(partials, nx, K, y, out, S, Kv, flag, seq);
  row_done(nrows, counters, flag, seq);   // the <= 3 result stores were issued by wave 0
}

// ------------------------------------------------------------------ K1: bound_poly_var_top (dense_mlpoly.rs:209-216)
// grid = (blocks over i, polys).  Z[i] <- Z[i] + r*(Z[i+half] - Z[i]);  r in s-form makes the product come out in memory form.
__global__ void __launch_bounds__(LASSO_BLOCK) k_bind_top(MutPtrTable polys, size_t half, fr_t r) {
  fr_t* __restrict__ z = polys.p[blockIdx.y];
  const fr29 rs = fr29_unpack_s(r);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    const fr29 lo = fr29_unpack_u(z[i]), hi = fr29_unpack_u(z[i + half]);
    z[i] = fr29_store(fr29_add(lo, fr29_mul(fr29_sub(hi, lo), rs)));
  }
}

// ------------------------------------------------------------------ K4: cubic round (sumcheck.rs:49-93)
// One index of one circuit: the products A*B*C at x = 0, 2, 3 from the values at x = 0 (a0..) and x = 1 (a1..), all reduced u-form.
// Each product of three u-form values comes out 2^10 short; the block result is corrected once with K10.
__device__ __forceinline__ void cubic_terms(const fr29& a0, const fr29& a1, const fr29& b0, const fr29& b1, const fr29& c0, const fr29& c1, fr29& t0, fr29& t2, fr29& t3) {
  t0 = fr29_mul(c0, fr29_mul(a0, b0));
  const fr29 da = fr29_sub(a1, a0), db = fr29_sub(b1, b0), dc = fr29_sub(c1, c0);
  const fr29 a2 = fr29_weak(fr29_add(a1, da)), b2 = fr29_weak(fr29_add(b1, db)), c2 = fr29_weak(fr29_add(c1, dc));   // 2*hi - lo
  t2 = fr29_mul(c2, fr29_mul(a2, b2));
  const fr29 a3 = fr29_add(a2, da), b3 = fr29_weak(fr29_add(b2, db)), c3 = fr29_add(c2, dc);                          // 3*hi - 2*lo
  t3 = fr29_mul(c3, fr29_mul(a3, b3));
}
// Grid shape of the cubic kernels: one workgroup row per circuit (the kernels are VALU-bound: two circuits per thread cost a third of the
// occupancy and ran 5% slower), laid out so that the workgroups of the SAME index range and different circuits land on the same XCD a few
// dispatch slots apart — they read the same slice of the shared eq polynomial C, and the second read should hit that XCD's L2 instead of
// HBM (the first PMC pass showed 1.2x the algorithmic read traffic at k = 2 with a plain (x, circuit) grid).  Workgroup ids go round-robin
// over the 8 XCDs, so ids b and b + 8 share an XCD: id = (x / 8) * 8 * ny + circuit * 8 + (x % 8).
struct CubicGrid { uint32_t bx, by, nx, ny; };
__device__ __forceinline__ CubicGrid cubic_grid(uint32_t nx, uint32_t ny) {
  CubicGrid g; g.nx = nx; g.ny = ny;
  const uint32_t b = blockIdx.x;
  if ((nx & 7u) == 0) { const uint32_t grp = b / (8 * ny), r = b - grp * 8 * ny; g.by = r >> 3; g.bx = grp * 8 + (r & 7u); }
  else { g.bx = b % nx; g.by = b / nx; }
  return g;
}
// raise the host flag once every grid row has stored its results: called by the workgroup that finished row y, stores issued by wave 0
__device__ __forceinline__ void row_done(uint32_t nrows, uint32_t* counters, uint32_t* flag, uint32_t seq) {
  if (flag == LASSO_TAGGED || flag == LASSO_TAGGED_DIRECT) {   // this row's chunks leave the device; nothing to agree on with the other rows
    // (the fence is REQUIRED: the mapped result area is cached in the device's L2 — without it the host never sees the chunks of a kernel that stays resident.  Round 6 tried:
    // tools/tail_phase_bench built with -DLASSO_NO_PUBLISH_FENCE stops at turn 0, with -DLASSO_NO_POLL_FENCE (no acquire in the mailbox poll) at turn 1 on a stale line.)
#ifndef LASSO_NO_PUBLISH_FENCE
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
#endif
    return;
  }
  if (threadIdx.x == 0 && flag) {
    __threadfence_system();
    uint32_t t2 = __hip_atomic_fetch_add(&counters[LASSO_MAX_PTRS], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t2 == nrows - 1) { counters[LASSO_MAX_PTRS] = 0; __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
}
// block partials -> memory (shift 10 / 5 also supplies the missing 2^10 / 2^5), then the in-launch second stage; a single-block row writes its result directly.
__device__ __forceinline__ void cubic_epilogue(const fr29* e, const CubicGrid& g, fr_t* __restrict__ partials, uint32_t* counters, fr_t* __restrict__ out, uint32_t* flag, uint32_t seq, RedScratch& S,
                                               int shift, uint32_t K = 3) {   // K <= 3 results per row
  if (g.nx == 1) {
    store_block_partials<3>(e, K, out, shift, S, flag, seq, (size_t)g.by * K);
    row_done(g.ny, counters, flag, seq);
    return;
  }
  if (flag == LASSO_TAGGED_DIRECT) {
    store_block_partials<3>(e, K, out, shift, S, flag, seq, ((size_t)g.by * g.nx + g.bx) * K);
    row_done(g.ny, counters, flag, seq);
    return;
  }
  store_block_partials<3>(e, K, partials + ((size_t)g.by * g.nx + g.bx) * K, shift, S);
  last_block_reduce(partials, g.nx, K, g.by, g.ny, counters, out, S, flag, seq);
}
#define CUBIC_ACCUMULATE(e, t0, t2, t3, cnt)                                                                                          \
  do {                                                                                                                                \
    e[0] = fr29_weak(fr29_add(e[0], t0)); e[1] = fr29_weak(fr29_add(e[1], t2)); e[2] = fr29_weak(fr29_add(e[2], t3));                    \
    if ((++cnt & 127u) == 0) { e[0] = fr29_mul(e[0], fr29_one_s()); e[1] = fr29_mul(e[1], fr29_one_s()); e[2] = fr29_mul(e[2], fr29_one_s()); } \
  } while (0)
// out[c*3 + {0,1,2}] = evals at x = 0, 2, 3.  First round of a layer: evaluation only.  1-D grid of nx*ny workgroups (cubic_grid).
__global__ void __launch_bounds__(LASSO_BLOCK) k_cubic_round_lb(PtrTable A, PtrTable B, uint32_t nx, uint32_t ny, const fr_t* __restrict__ C, size_t half, fr_t* __restrict__ partials, uint32_t* counters,
                                                                 fr_t* __restrict__ out, uint32_t* flag, uint32_t seq) {
  __shared__ RedScratch S;
  const CubicGrid g = cubic_grid(nx, ny);
  const fr_t* __restrict__ a = A.p[g.by];
  const fr_t* __restrict__ b = B.p[g.by];
  fr29 e[3] = {fr29_zero(), fr29_zero(), fr29_zero()}; uint32_t cnt = 0;
  for (size_t i = g.bx * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)nx * blockDim.x) {
    fr29 t0, t2, t3;
    cubic_terms(fr29_unpack_u(a[i]), fr29_unpack_u(a[i + half]), fr29_unpack_u(b[i]), fr29_unpack_u(b[i + half]), fr29_unpack_u(C[i]), fr29_unpack_u(C[i + half]), t0, t2, t3);
    CUBIC_ACCUMULATE(e, t0, t2, t3, cnt);
  }
  cubic_epilogue(e, g, partials, counters, out, flag, seq, S, 10);
}
// bind for arrays only kernels read until the layer's final bind: the stored value is lazily reduced (fr29_semi: same residue, < 2^254 + 2^130,
// 48 instructions against 115) — the next round's loads, the resident tail kernel and k_bind_top all accept it, and what leaves the device
// (the heads after the last bind) is canonical again.  Used by the two-sum fused rounds (the dominant kernel of a proof).
__device__ __forceinline__ fr29 bind29_semi(const fr_t& lo, const fr_t& hi, const fr29& rs) {
  const fr29 l = fr29_unpack_u(lo);
  return fr29_semi(fr29_add(l, fr29_mul(fr29_sub(fr29_unpack_u(hi), l), rs)));
}
__device__ __forceinline__ fr29 bind29(const fr_t& lo, const fr_t& hi, const fr29& rs) {
  const fr29 l = fr29_unpack_u(lo);
  return fr29_canonical(fr29_add(l, fr29_mul(fr29_sub(fr29_unpack_u(hi), l), rs)));   // canonical: stored as is, and a reduced operand below
}

// ------------------------------------------------------------------ K4 in eq-weighted form (the form the prover uses)
// In prove_cubic_batched (grand_product.rs:126-128 -> sumcheck.rs:49-124) the third polynomial C is ALWAYS EqPolynomial(rand).evals().  After
// binding its top j variables with challenges rho it is  C_j[i] = s_j * eq1(rand_j, x_top) * T_j[i_low]  with T_j = eq(rand[j+1..]) and
// s_j = prod_{t<j} eq1(rand_t, rho_t); and T_j is the PREFIX of the original table up to a scalar: E[i] = prod_{t<=j}(1 - rand_t) * T_j[i]
// for i < 2^(l-1-j).  So the round's sums are  e(x) = [s_j * eq1(rand_j, x) / prod_{t<=j}(1 - rand_t)] * sum_i a(x) b(x) E[i]:
// the device never binds or stores C — it reads the prefix of the one table built for the layer — and the bracket is three host scalars.
// Per index and circuit (three-sum form): 2 products for aE(x) = a(x) E[i] (linear in x), 3 for b(x) aE(x): 5 instead of 6, and the fused kernel
// binds 4 values instead of 6 (9 products per index against 12).  Exact field arithmetic: the round polynomials are the same field elements.
// The prover's path is the two-sum form below (cubic_eqw_terms2: 8 products, two of them accumulated double-width); the three-sum kernels serve
// the rounds whose eq coordinate is 0 (no claim-derived evaluation) and the literal API.
__device__ __forceinline__ void cubic_eqw_terms(const fr29& a0, const fr29& a1, const fr29& b0, const fr29& b1, const fr29& es, fr29& t0, fr29& t2, fr29& t3) {
  const fr29 g0 = fr29_mul(a0, es), g1 = fr29_mul(a1, es);          // u * s = u-form, reduced
  t0 = fr29_mul(b0, g0);
  const fr29 dg = fr29_sub(g1, g0), db = fr29_sub(b1, b0);
  const fr29 g2 = fr29_weak(fr29_add(g1, dg)), b2 = fr29_weak(fr29_add(b1, db));   // 2*hi - lo
  t2 = fr29_mul(b2, g2);
  t3 = fr29_mul(fr29_add(b2, db), fr29_weak(fr29_add(g2, dg)));                     // 3*hi - 2*lo; u * u: 2^5 short, fixed once per block with K5
}
// Two sums instead of three.  q(x) = sum_i a(x) b(x) E[i] is QUADRATIC in x, and the round's claim already fixes one linear condition on it
// (e = cubic(0) + cubic(1), sumcheck.rs:99-104 uses it to derive the evaluation at 1), so two numbers per circuit determine the round polynomial:
// q(0) = sum a0 b0 E  and the leading coefficient  q_inf = sum (a1 - a0)(b1 - b0) E.   The host recovers q(1) from the claim and the evaluations at
// 2, 3 by extrapolation (prover.hpp cubic_rounds) — the same field elements.  8 products per index and circuit in the fused kernel instead of 9,
// two accumulators instead of three.   NT = 3: sums at x = 0, 2, 3;  NT = 2: (q(0), q_inf).
__device__ __forceinline__ void cubic_eqw_terms2(const fr29& a0, const fr29& a1, const fr29& b0, const fr29& b1, const fr29& es, fr29& t0, fr29& tinf) {
  const fr29 g0 = fr29_mul(a0, es), g1 = fr29_mul(a1, es);
  t0 = fr29_mul(b0, g0);
  tinf = fr29_mul(fr29_sub(g1, g0), fr29_sub(b1, b0));   // b0, b1 canonical: the difference keeps |limb| < 2^29
}
#define CUBIC_ACCUMULATE2(e, t0, t1, cnt)                                                                         \
  do {                                                                                                            \
    e[0] = fr29_weak(fr29_add(e[0], t0)); e[1] = fr29_weak(fr29_add(e[1], t1));                                     \
    if ((++cnt & 127u) == 0) { e[0] = fr29_mul(e[0], fr29_one_s()); e[1] = fr29_mul(e[1], fr29_one_s()); }          \
  } while (0)
// The layer's eq table built INSIDE the first round's launch (round 3).  Every grand-product layer starts with  E = scale * EqPolynomial(point).evals()  and only then its
// round 0; as launches of their own the table kernels sit on the proof's critical path 40 times (0.9 ms at the headline: launch-bound, profiles/r03_kernel_trace_one_proof_2p24.csv).
// For tables of up to 2^14 entries every workgroup of the round's launch rebuilds the two FACTOR tables of eq_poly.rs:44-52's split — hi over the first ceil(ell/2)
// coordinates (scale folded in), lo over the rest, at most 128 entries each: one lane per entry, a chain of <= 7 products, ~3 us — in LDS, and an entry of E costs one product
// E[x] = hi[x >> lo_bits] * lo[x & mask] where it is used.  hi and lo are kept in s-form (their product is the s-form operand the round needs); lo also in u-form, so that the
// workgroups of circuit 0 can write E[x] = hi * lo_u to memory in the canonical bytes k_eq_small / k_eq_outer produce — the later rounds of the layer read its prefix.
struct EqInline { fr_t r[14]; fr_t scale; uint32_t ell; };
struct EqNone { uint32_t ell = 0; const uint32_t* gp = nullptr; uint32_t seq = 0; };   // stands in for EqInline in the launches that carry no point (16 bytes of kernel arguments instead of 484); gp: the launch sits behind a point gate that may have failed
struct EqInlineTables { fr29 hi_s[128], lo_s[128], lo_u[128]; };
// the same point, not an argument: left in device memory by the gate in front of the launch (k_gate_point), for a layer enqueued before its point was known
struct EqInlineMem { const uint32_t* gp; uint32_t seq; uint32_t ell; };
__device__ __forceinline__ fr_t eq_point_r(const EqInline& Q, uint32_t j) { return Q.r[j]; }
__device__ __forceinline__ fr_t eq_point_scale(const EqInline& Q) { return Q.scale; }
__device__ __forceinline__ bool eq_point_ok(const EqInline&) { return true; }
__device__ __forceinline__ fr_t eq_point_r(const EqInlineMem& Q, uint32_t j) { return gate_point_fr(Q.gp, j); }
__device__ __forceinline__ fr_t eq_point_scale(const EqInlineMem& Q) { return gate_point_fr(Q.gp, Q.ell); }
__device__ __forceinline__ bool eq_point_ok(const EqInlineMem& Q) { return gate_point_ok(Q.gp, Q.seq); }
// all threads of the workgroup; ends with a barrier.  false (block-uniform): the point never came, the launch ends without a result
template <class QT>
__device__ __forceinline__ bool eq_inline_build_t(const QT& Q, EqInlineTables& T) {
  if (!eq_point_ok(Q)) return false;
  const uint32_t t = threadIdx.x, lb = Q.ell / 2, hb = Q.ell - lb;
  const fr29 one_s = fr29_one_s();
  if (t < (1u << hb)) {
    fr29 p = fr29_unpack_s(eq_point_scale(Q));
    for (uint32_t j = 0; j < hb; j++) { const bool bit = (t >> (hb - 1 - j)) & 1u; const fr29 rs = fr29_unpack_s(eq_point_r(Q, j)); p = fr29_mul(bit ? rs : fr29_sub(one_s, rs), p); }
    T.hi_s[t] = p;
  } else if (t >= 128 && t - 128 < (1u << lb)) {
    const uint32_t x = t - 128;
    fr29 p = one_s;
    for (uint32_t j = 0; j < lb; j++) { const bool bit = (x >> (lb - 1 - j)) & 1u; const fr29 rs = fr29_unpack_s(eq_point_r(Q, hb + j)); p = fr29_mul(bit ? rs : fr29_sub(one_s, rs), p); }
    T.lo_s[x] = p;
    T.lo_u[x] = fr29_mul(p, fr29_unpack_u(fr_one()));   // s-form times the integer 2^256: the same value in u-form
  }
  __syncthreads();
  return true;
}
__device__ __forceinline__ bool eq_inline_build(const EqInline& Q, EqInlineTables& T) { return eq_inline_build_t(Q, T); }
__device__ __forceinline__ bool eq_inline_build(const EqInlineMem& Q, EqInlineTables& T) { return eq_inline_build_t(Q, T); }
__device__ __forceinline__ fr29 eq_inline_s(const EqInlineTables& T, uint32_t lb, size_t x) { return fr29_mul(T.hi_s[x >> lb], T.lo_s[x & ((1u << lb) - 1u)]); }
// Tables of MORE than 2^14 entries, optional form (LASSO_EQ_INLINE_BIG=1; round 5 — measured and NOT the default: it takes k_eq_outer's 63 us per proof out of the stream but costs
// round 0 of the eight largest layers 120 us, +18 %, in a kernel that is already VALU co-limited, and those launches are a third of the roofline set:
// profiles/r05_ab_layer_ahead_eq_global.txt; with the layers enqueued ahead of their point the separate kernels cost no launch gap any more): the two factor tables (<= 2^11 entries each) do not fit LDS, so they stay where k_eq_small2 wrote them — global memory, memory form,
// L2-resident — and round 0 forms  E[x] = hi[x >> lo_bits] * lo[x & mask]  where it uses it: k_eq_outer's product, bit for bit, written to E_out by circuit 0's workgroups for the
// later rounds and never read back in this launch.  Takes k_eq_outer (a 32-byte write per entry, then the same bytes read again by round 0: 5-200 us in front of round 0 of the
// nine largest layers of a 2^24 proof) off the critical path for one product per index in a launch that is HBM-bound.
struct EqGlobal { const fr_t* hi; const fr_t* lo; uint32_t lo_bits; uint32_t ell; const uint32_t* gp; uint32_t seq; };   // gp != nullptr: the factor tables were built behind a point gate (k_eq_small2_mem), which may have failed
__device__ __forceinline__ bool eq_inline_build(const EqGlobal& Q, EqInlineTables&) { return Q.gp == nullptr || gate_point_ok(Q.gp, Q.seq); }
__device__ __forceinline__ bool eq_inline_build(const EqNone&, EqInlineTables&) { return true; }
// launches that READ their table (EQI = false) behind a point gate: the table was built from the gated point by the kernels in front (k_eq_small2_mem, k_eq_outer)
template <class TE> __device__ __forceinline__ bool eq_gate_ok(const TE&) { return true; }
template <> __device__ __forceinline__ bool eq_gate_ok<EqNone>(const EqNone& Q) { return Q.gp == nullptr || gate_point_ok(Q.gp, Q.seq); }
// the eq weight of index i as the s-form operand of the round, and (write) the table entry in memory form
__device__ __forceinline__ fr29 eq_inline_value(const EqInline&, const EqInlineTables& T, uint32_t lb, size_t i, bool write, fr_t* __restrict__ E_out) {
  if (write) E_out[i] = fr29_store(fr29_mul(T.hi_s[i >> lb], T.lo_u[i & ((1u << lb) - 1u)]));
  return eq_inline_s(T, lb, i);
}
__device__ __forceinline__ fr29 eq_inline_value(const EqInlineMem&, const EqInlineTables& T, uint32_t lb, size_t i, bool write, fr_t* __restrict__ E_out) {
  if (write) E_out[i] = fr29_store(fr29_mul(T.hi_s[i >> lb], T.lo_u[i & ((1u << lb) - 1u)]));
  return eq_inline_s(T, lb, i);
}
__device__ __forceinline__ fr29 eq_inline_value(const EqGlobal& Q, const EqInlineTables&, uint32_t, size_t i, bool write, fr_t* __restrict__ E_out) {
  const fr_t mem = fr29_store(fr29_mul(fr29_unpack_u(Q.hi[i >> Q.lo_bits]), fr29_unpack_s(Q.lo[i & (((size_t)1 << Q.lo_bits) - 1)])));
  if (write) E_out[i] = mem;
  return fr29_unpack_s(mem);
}
__device__ __forceinline__ fr29 eq_inline_value(const EqNone&, const EqInlineTables&, uint32_t, size_t, bool, fr_t* __restrict__) { return fr29_zero(); }
// out[c*NT + ..] = the NT sums over i < half of circuit c.  1-D grid of nx*ny workgroups (cubic_grid).
// EQI: E is built on the way (eq_inline_build): E_out (half entries) is written by the workgroups of circuit 0, nothing is read from it.
// one 32-byte element with the non-temporal hint (two dwordx4): data that is read exactly once per launch (the layer's A and B in a round that only evaluates)
__device__ __forceinline__ fr_t fr_load_nt(const fr_t* p) {
  const lasso_u32x4* q = reinterpret_cast<const lasso_u32x4*>(p);
  const lasso_u32x4 x = __builtin_nontemporal_load(q), y = __builtin_nontemporal_load(q + 1);
  fr_t r; r.v[0] = x.x; r.v[1] = x.y; r.v[2] = x.z; r.v[3] = x.w; r.v[4] = y.x; r.v[5] = y.y; r.v[6] = y.z; r.v[7] = y.w; return r;
}
template <int NT, bool EQI = false, class TP = PtrTable, class TE = EqInline, bool NTL = false>
__global__ void __launch_bounds__(LASSO_BLOCK) k_cubic_eqw_lb(TP A, TP B, uint32_t nx, uint32_t ny, const fr_t* __restrict__ E, size_t half, fr_t* __restrict__ partials, uint32_t* counters,
                                                               fr_t* __restrict__ out, uint32_t* flag, uint32_t seq, uint32_t pipeline, TE EQ = TE(), fr_t* __restrict__ E_out = nullptr) {
  __shared__ RedScratch S;
  __shared__ EqInlineTables ET;
  const CubicGrid g = cubic_grid(nx, ny);
  const uint32_t elb = EQ.ell / 2;
  if constexpr (EQI) { if (!eq_inline_build(EQ, ET)) return; } else { if (!eq_gate_ok(EQ)) return; }
  const fr_t* __restrict__ a = A.p[g.by];
  const fr_t* __restrict__ b = B.p[g.by];
  fr29 e[3] = {fr29_zero(), fr29_zero(), fr29_zero()}; uint32_t cnt = 0;
  fr29_acc w0 = fr29_acc_zero(), w1 = fr29_acc_zero();
  const size_t stride = (size_t)nx * blockDim.x;
  size_t i = g.bx * (size_t)blockDim.x + threadIdx.x;
  if (NT == 3) {
    for (; i < half; i += stride) {
      fr29 t0, t2, t3;
      cubic_eqw_terms(fr29_unpack_u(a[i]), fr29_unpack_u(a[i + half]), fr29_unpack_u(b[i]), fr29_unpack_u(b[i + half]), fr29_unpack_s(E[i]), t0, t2, t3); CUBIC_ACCUMULATE(e, t0, t2, t3, cnt);
    }
  } else if (!pipeline) {   // A/B switch (LASSO_LB_PIPELINE=0): the plain loop of round 2
    for (; i < half; i += stride) {
      const fr29 es = fr29_unpack_s(E[i]), b0 = fr29_unpack_u(b[i]), b1 = fr29_unpack_u(b[i + half]);
      const fr29 g0 = fr29_mul(fr29_unpack_u(a[i]), es), g1 = fr29_mul(fr29_unpack_u(a[i + half]), es);
      fr29_mul_acc(w0, b0, g0); fr29_mul_acc(w1, fr29_sub(g1, g0), fr29_sub(b1, b0));
      if (++cnt == 3) { fr29_acc_carry(w0); fr29_acc_carry(w1); cnt = 0; }
    }
  } else if (i < half) {
    // software-pipelined: the five 32-byte loads of the NEXT index are in flight while this one's four products issue.  The round reads and never writes, its
    // ~900 instructions per index are too few to hide a ~2 us HBM access behind two waves per SIMD, and the compiler keeps the loads at the head of the loop
    // body: measured 3.6 TB/s of reads (1.07 GB in 316 us at the 2^24 top layer) where a read stream reaches 6.
    // NTL (LASSO_LB_NT=1): A and B with the non-temporal hint — they are read once, and a read-only stream reaches 6.7-7.0 TB/s that way against 6.0-6.3 plain
    // (DESIGN_HISTORY 6.1); E is shared by the circuits' workgroups on one XCD and stays cached
#define LB_LD(ptr) (NTL ? fr_load_nt(ptr) : *(ptr))
    fr_t a0 = LB_LD(a + i), a1 = LB_LD(a + i + half), b0m = LB_LD(b + i), b1m = LB_LD(b + i + half), em = EQI ? fr_zero() : E[i];
    for (;;) {
      const size_t in = i + stride; const bool more = in < half;
      const size_t ip = more ? in : i;     // clamp: the last iteration re-reads its own (cached) lines instead of branching around the loads
      const fr_t na0 = LB_LD(a + ip), na1 = LB_LD(a + ip + half), nb0 = LB_LD(b + ip), nb1 = LB_LD(b + ip + half), ne = EQI ? fr_zero() : E[ip];
#undef LB_LD
      fr29 es;
      if (EQI) es = eq_inline_value(EQ, ET, elb, i, g.by == 0, E_out);
      else es = fr29_unpack_s(em);
      const fr29 b0 = fr29_unpack_u(b0m), b1 = fr29_unpack_u(b1m);
      const fr29 g0 = fr29_mul(fr29_unpack_u(a0), es), g1 = fr29_mul(fr29_unpack_u(a1), es);
      fr29_mul_acc(w0, b0, g0); fr29_mul_acc(w1, fr29_sub(g1, g0), fr29_sub(b1, b0));
      if (++cnt == 3) { fr29_acc_carry(w0); fr29_acc_carry(w1); cnt = 0; }
      if (!more) break;
      a0 = na0; a1 = na1; b0m = nb0; b1m = nb1; em = ne; i = in;
    }
  }
  if (NT == 2) { fr29_acc_carry(w0); fr29_acc_carry(w1); e[0] = fr29_acc_reduce(w0); e[1] = fr29_acc_reduce(w1); }
  cubic_epilogue(e, g, partials, counters, out, flag, seq, S, 5, NT);
}
// fused with K1: bind A and B with r (length n = 4q -> 2q, in place: each element is owned by exactly one thread), then the sums of the NEXT round
// on the bound values while they are still in registers — one launch per round, 48 bytes per element of A and B plus 32 per index of E.
// AHEAD: the challenge is not an argument — the launch was enqueued before the host had it, behind a gate kernel that waits for it (k_gate above) and leaves it in gmail.
template <int NT, bool WIDE = false, class TM = MutPtrTable, bool AHEAD = false>
#ifdef LASSO_FUSED_WAVES   // experiment switch: force the register budget of the fused round (waves per SIMD); default = the compiler's choice (157 VGPRs, 3 waves)
__attribute__((amdgpu_waves_per_eu(LASSO_FUSED_WAVES, LASSO_FUSED_WAVES)))
#endif
__global__ void __launch_bounds__(LASSO_BLOCK) k_cubic_eqw_fused(TM A, TM B, uint32_t nx, uint32_t ny, const fr_t* __restrict__ E, size_t q, fr_t r,
                                                                  fr_t* __restrict__ partials, uint32_t* counters, fr_t* __restrict__ out, uint32_t* flag, uint32_t seq,
                                                                  const uint32_t* gmail = nullptr, const uint32_t* mail = nullptr) {
  __shared__ RedScratch S;
  const CubicGrid g = cubic_grid(nx, ny);
  fr_t* __restrict__ a = A.p[g.by];
  fr_t* __restrict__ b = B.p[g.by];
  if constexpr (AHEAD) {
    __shared__ uint32_t s_mail[17];
    if (mail) {   // few workgroups: they wait themselves (gmail is written by workgroup 0 of this launch)
      if (!ahead_challenge(mail, const_cast<uint32_t*>(gmail), seq, r, s_mail)) return;
    } else {      // many: a gate kernel in front of this launch did the waiting
      if (!gated_challenge(gmail, seq, 8, s_mail)) return;
#pragma unroll
      for (int k = 0; k < 8; k++) r.v[k] = s_mail[k];
    }
  }
  const fr29 rs = fr29_unpack_s(r);
  fr29 e[3] = {fr29_zero(), fr29_zero(), fr29_zero()}; uint32_t cnt = 0;
  fr29_acc w0 = fr29_acc_zero(), w1 = fr29_acc_zero();
  for (size_t i = g.bx * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)nx * blockDim.x) {
    const fr29 a0 = NT == 2 ? bind29_semi(a[i], a[i + 2 * q], rs) : bind29(a[i], a[i + 2 * q], rs), a1 = NT == 2 ? bind29_semi(a[i + q], a[i + 3 * q], rs) : bind29(a[i + q], a[i + 3 * q], rs);
    a[i] = fr29_pack(a0); a[i + q] = fr29_pack(a1);
    const fr29 b0 = NT == 2 ? bind29_semi(b[i], b[i + 2 * q], rs) : bind29(b[i], b[i + 2 * q], rs), b1 = NT == 2 ? bind29_semi(b[i + q], b[i + 3 * q], rs) : bind29(b[i + q], b[i + 3 * q], rs);
    b[i] = fr29_pack(b0); b[i + q] = fr29_pack(b1);
    fr29 t0, t2, t3;
    if (NT == 3) { cubic_eqw_terms(a0, a1, b0, b1, fr29_unpack_s(E[i]), t0, t2, t3); CUBIC_ACCUMULATE(e, t0, t2, t3, cnt); }
    else if (!WIDE) { cubic_eqw_terms2(a0, a1, b0, b1, fr29_unpack_s(E[i]), t0, t2); CUBIC_ACCUMULATE2(e, t0, t2, cnt); }
    else {
      // the two sums are sums of PRODUCTS: add the double-width products and reduce once per thread (fr29_mul_acc)
      const fr29 es = fr29_unpack_s(E[i]);
      const fr29 g0 = fr29_mul(a0, es), g1 = fr29_mul(a1, es);
      fr29_mul_acc(w0, b0, g0); fr29_mul_acc(w1, fr29_sub(g1, g0), fr29_sub(b1, b0));
      if (++cnt == 3) { fr29_acc_carry(w0); fr29_acc_carry(w1); cnt = 0; }
    }
  }
  if (WIDE && NT == 2) { fr29_acc_carry(w0); fr29_acc_carry(w1); e[0] = fr29_acc_reduce(w0); e[1] = fr29_acc_reduce(w1); }
  cubic_epilogue(e, g, partials, counters, out, flag, seq, S, 5, NT);
}
// Late rounds (q <= 64 indices per circuit): the same round, laid out for LATENCY instead of throughput.  One workgroup per circuit;
// phase 1 gives every bind its own lane (4q products side by side instead of 4 in a row per thread), phase 2 every weighted value a'[i] E[i mod q],
// phase 3 every (index, evaluation point), one wave per point; the three sums are 64-row column sums.  BIND = false is the first round of a
// layer (no challenge yet): phase 1 only unpacks.  Lengths: A, B hold 4q elements when BIND, 2q otherwise.
template <bool BIND, int NT, class TM = MutPtrTable>
__global__ void __launch_bounds__(LASSO_BLOCK) k_cubic_eqw_small(TM A, TM B, const fr_t* __restrict__ E, uint32_t q, fr_t r, uint32_t* counters, fr_t* __restrict__ out,
                                                                  uint32_t* flag, uint32_t seq) {
  __shared__ fr29 bound[2][128];   // A', B' (2q values each)
  __shared__ fr29 ge[128];         // A'[i] * E[i mod q]
  __shared__ int32_t rows[192 * 9];
  __shared__ int64_t cols[27];
  const uint32_t t = threadIdx.x, y = blockIdx.x, m = 2 * q;
  const fr29 rs = fr29_unpack_s(r);
  for (uint32_t item = t; item < 2 * m; item += LASSO_BLOCK) {
    const uint32_t p = item / m, i = item - p * m;
    fr_t* dst = p == 0 ? A.p[y] : B.p[y];
    fr29 v;
    if (BIND) { v = bind29(dst[i], dst[i + m], rs); dst[i] = fr29_pack(v); } else v = fr29_unpack_u(dst[i]);
    bound[p][i] = v;
  }
  __syncthreads();
  if (t < m) ge[t] = fr29_mul(bound[0][t], fr29_unpack_s(E[t < q ? t : t - q]));
  __syncthreads();
  const uint32_t x = t >> 6, i = t & 63;   // wave x evaluates point {0, 2, 3}[x]  (NT = 2: q(0) and the leading coefficient)
  if (x < NT) {
    fr29 term = fr29_zero();
    if (i < q) {
      const fr29 g0 = ge[i], g1 = ge[i + q], b0 = bound[1][i], b1 = bound[1][i + q];
      if (x == 0) term = fr29_mul(b0, g0);
      else if (NT == 2) term = fr29_mul(fr29_sub(g1, g0), fr29_sub(b1, b0));
      else {
        const fr29 dg = fr29_sub(g1, g0), db = fr29_sub(b1, b0);
        const fr29 g2 = fr29_weak(fr29_add(g1, dg)), b2 = fr29_weak(fr29_add(b1, db));
        term = x == 1 ? fr29_mul(b2, g2) : fr29_mul(fr29_add(b2, db), fr29_weak(fr29_add(g2, dg)));
      }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) rows[t * 9 + k] = term.v[k];
  }
  __syncthreads();
  if (t < 9 * NT) {
    const uint32_t v = t / 9, k = t - v * 9;
    int64_t sum = 0;
    for (uint32_t j = 0; j < 64; j++) sum += rows[(v * 64 + j) * 9 + k];
    cols[t] = sum;
  }
  __syncthreads();
  if (t < NT) {
    int64_t c[9];
#pragma unroll
    for (int k = 0; k < 9; k++) c[k] = cols[t * 9 + k];
    result_store(out, (size_t)y * NT + t, fr29_pack(fr29_reduce_columns(c, 5)), flag, seq);
  }
  row_done(gridDim.x, counters, flag, seq);
}

// The whole TAIL of a layer's sumcheck in one resident kernel.  Once a layer is down to q <= CUBIC_TAIL_Q indices per circuit its remaining log2(q) + 1
// rounds move a few KB each; a launch per round costs 6.5 us of round trip (tools/latency_bench.hip) against 2.5 us for a resident kernel that
// waits for the host's next challenge in a host-mapped mailbox (tools/pingpong_bench.hip).  So: bind (or load) into LDS once, then per round
// publish the two sums (q(0), q_inf) through the mapped result buffer + flag, spin on the mailbox, bind in LDS, ... and finally publish the
// bound heads A[0], B[0] (sumcheck.rs:126-133) — the arrays never go back to HBM (nothing reads a layer's bound arrays after its sumcheck).
// One workgroup per circuit.  mailbox (host-mapped memory, or device memory the host writes through the BAR: lasso_hip.hip device_mailbox): three 16-byte chunks
// [tag, w0, w1, w2] [tag, w3, w4, w5] [tag, w6, w7, check], w = the challenge, tag = seq0 + turn + 1, check = result_check's formula;
// turn k's results carry sequence number seq0 + k.  Every spin has a wall-clock bail-out: a host that never answers cannot hang the device.
// out (2 * ncirc elements per turn): sums turns: out[2c], out[2c+1];  final turn: out[c] = A_c head, out[ncirc + c] = B_c head.
// Q = capacity in indices per circuit = threads of the workgroup: 256 (74 KB of LDS) or, since round 3, 512 (147 KB of the CU's 160 KB: one streaming round fewer per layer —
// a resident turn costs ~10 us where a launch-per-round costs ~25 us at these sizes, profiles/r03_kernel_trace_one_proof_2p24.csv)
#define CUBIC_TAIL_Q 512   // the resident kernels take over at <= this many indices per circuit
#ifdef TAIL_PHASE_CLOCK   // tools/tail_phase_bench.hip: workgroup 0 stamps the 100 MHz wall clock at the phase boundaries of each of its first 16 turns
__device__ uint64_t tail_phase_clock[16 * 8];
#define TAIL_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && turn < 16) tail_phase_clock[turn * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TAIL_STAMP(k) do { } while (0)
#endif
// EQI (first round of a layer only): the eq table is never materialised — the two factor tables are built in LDS (eq_inline_build, <= 32 entries each at q <= 512) and
// every use of E[i] is one product.
// m_stop (round 5): the kernel stops when the arrays are down to m_stop elements each and its LAST publication carries the arrays — out[(p * ncirc + c) * m_stop + i] = (p ? B : A)_c[i] —
// instead of the heads (m_stop = 1: the heads, as before): the host finishes the last log2(m_stop) rounds itself (a few dozen field products against ~7 us per resident turn).
// wait_r0 (BIND only): the launch was enqueued AHEAD of its first challenge; r0 arrives through the mailbox under tag seq0 (the publication it enables) while the loads are in flight.
template <bool BIND, int Q, bool EQI = false, class TM = MutPtrTable, class TE = EqInline>
__global__ void __launch_bounds__(Q) k_cubic_tail(TM A, TM B, const fr_t* __restrict__ E, uint32_t q, fr_t r0, const uint32_t* mailbox, uint32_t* counters,
                                                             fr_t* __restrict__ out, uint32_t* flag, uint32_t seq0, TE EQ = TE(), uint32_t m_stop = 1, uint32_t wait_r0 = 0) {
  __shared__ fr29 eq_hi[EQI ? 32 : 1], eq_lo[EQI ? 32 : 1];
  const uint32_t elb = EQ.ell / 2;
  if constexpr (EQI) {   // ell = log2 q <= 9: hi over the first ceil(ell/2) coordinates (scale folded in), lo over the rest, both s-form
    if (!eq_point_ok(EQ)) return;   // (EqInlineMem: the layer was enqueued ahead of its point and the point never came)
    const uint32_t tt = threadIdx.x, hb = EQ.ell - elb; const fr29 one_s = fr29_one_s();
    if (tt < (1u << hb)) { fr29 p = fr29_unpack_s(eq_point_scale(EQ)); for (uint32_t j = 0; j < hb; j++) { const bool bit = (tt >> (hb - 1 - j)) & 1u; const fr29 rs = fr29_unpack_s(eq_point_r(EQ, j)); p = fr29_mul(bit ? rs : fr29_sub(one_s, rs), p); } eq_hi[tt] = p; }
    else if (tt >= 64 && tt - 64 < (1u << elb)) { const uint32_t x = tt - 64; fr29 p = one_s; for (uint32_t j = 0; j < elb; j++) { const bool bit = (x >> (elb - 1 - j)) & 1u; const fr29 rs = fr29_unpack_s(eq_point_r(EQ, hb + j)); p = fr29_mul(bit ? rs : fr29_sub(one_s, rs), p); } eq_lo[x] = p; }
    __syncthreads();
  }
#define TAIL_EQ_S(idx) (EQI ? fr29_mul(eq_hi[(idx) >> elb], eq_lo[(idx) & ((1u << elb) - 1u)]) : fr29_unpack_s(E[(idx)]))
  __shared__ fr29 bound[2][2 * Q];   // A', B' (m values each)
  __shared__ fr29 ge[2 * Q];         // A'[i] * E[i mod h]
  __shared__ int32_t rows[2 * Q * 9];
  __shared__ int64_t strips[8 * 18];
  __shared__ int64_t cols[18];
  __shared__ fr_t chal;
  __shared__ uint32_t alive;
  const uint32_t t = threadIdx.x, y = blockIdx.x, ncirc = gridDim.x;
  const uint64_t t_end = wall_clock64() + 500000000ull;   // 5 s at 100 MHz
  uint32_t m = 2 * q;
  {
    // every lane task (array p, index i) loads / binds its element and, for A, weights it with the eq table right away (no separate pass).  2 m <= 4 Q lane tasks: all of a
    // thread's loads are issued first (and, launched ahead, travel while the first challenge does)
    // (four named pairs, not arrays: the compiler does not unroll the second loop and would index arrays in scratch — 272 bytes per lane and a round trip through memory)
    fr_t lo0, lo1, lo2, lo3, hi0, hi1, hi2, hi3;
#define TAIL_LD(LO, HI, IT) do { const uint32_t item = t + (IT) * Q; if (item < 2 * m) { const uint32_t p = item / m, i = item - p * m; const fr_t* src = p == 0 ? A.p[y] : B.p[y]; LO = src[i]; if (BIND) HI = src[i + m]; } } while (0)
    TAIL_LD(lo0, hi0, 0); TAIL_LD(lo1, hi1, 1); TAIL_LD(lo2, hi2, 2); TAIL_LD(lo3, hi3, 3);
#undef TAIL_LD
    if (BIND && wait_r0) {
      if (t == 0) alive = mail_wait(mailbox, seq0, t_end, chal) ? 1u : 0u;
      __syncthreads();
      if (!alive) return;
      r0 = chal;
      __syncthreads();   // chal / alive are written again at the first turn's poll
    }
    const fr29 rs = fr29_unpack_s(r0);
#define TAIL_BIND(LO, HI, IT) do { const uint32_t item = t + (IT) * Q; if (item < 2 * m) { const uint32_t p = item / m, i = item - p * m; \
      const fr29 v = BIND ? bind29(LO, HI, rs) : fr29_unpack_u(LO); bound[p][i] = v; if (p == 0) ge[i] = fr29_mul(v, TAIL_EQ_S(i < q ? i : i - q)); } } while (0)
    TAIL_BIND(lo0, hi0, 0); TAIL_BIND(lo1, hi1, 1); TAIL_BIND(lo2, hi2, 2); TAIL_BIND(lo3, hi3, 3);
#undef TAIL_BIND
  }
  __syncthreads();
  for (uint32_t turn = 0;; turn++) {
    const uint32_t h = m / 2;      // pairs this round
    TAIL_STAMP(0);
    // 2h terms: u < h the q(0) terms, u >= h the leading-coefficient terms; row u of `rows`
    for (uint32_t u = t; u < 2 * h; u += Q) {
      const uint32_t v = u >= h ? 1u : 0u, i = u - v * h;
      // operands selected, ONE product: at h <= 32 both kinds of term sit in the same wave, and a branch per kind would run the product twice
      const fr29 g0 = ge[i], g1 = ge[i + h], b0 = bound[1][i], b1 = bound[1][i + h];
      const fr29 dg = fr29_sub(g1, g0), db = fr29_sub(b1, b0);
      fr29 fa, fb;
#pragma unroll
      for (int k = 0; k < 9; k++) { fa.v[k] = v ? dg.v[k] : b0.v[k]; fb.v[k] = v ? db.v[k] : g0.v[k]; }
      const fr29 term = fr29_mul(fa, fb);
#pragma unroll
      for (int k = 0; k < 9; k++) rows[u * 9 + k] = term.v[k];
    }
    __syncthreads();
    TAIL_STAMP(1);
    if (h > 16) {   // eight strips of rows per (sum, limb) column, then the strips
      if (t < 8 * 18) {
        const uint32_t col = t % 18, strip = t / 18, v = col / 9, k = col - v * 9;
        const uint32_t per = (h + 7) / 8, i0 = strip * per, i1 = i0 + per < h ? i0 + per : h;
        int64_t sum = 0;
        for (uint32_t i = i0; i < i1; i++) sum += rows[(v * h + i) * 9 + k];
        strips[strip * 18 + col] = sum;
      }
      __syncthreads();
      if (t < 18) { int64_t sum = 0; for (int g = 0; g < 8; g++) sum += strips[g * 18 + t]; cols[t] = sum; }
    } else if (t < 18) {   // a handful of rows: one pass
      const uint32_t v = t / 9, k = t - v * 9;
      int64_t sum = 0;
      for (uint32_t i = 0; i < h; i++) sum += rows[(v * h + i) * 9 + k];
      cols[t] = sum;
    }
    __syncthreads();
    TAIL_STAMP(2);
    if (t < 2) {
      int64_t c[9];
#pragma unroll
      for (int k = 0; k < 9; k++) c[k] = cols[t * 9 + k];
      result_store(out, (size_t)y * 2 + t, fr29_pack(fr29_reduce_columns(c, 5)), flag, seq0 + turn);
    }
    row_done(ncirc, counters, flag, seq0 + turn);
    TAIL_STAMP(3);
    // everything of the coming bind that does not depend on the challenge, while it travels: the pair (lo, hi - lo) of every lane task and the eq weight of the A tasks
    const uint32_t hn = h / 2;     // pairs of the NEXT round
    fr29 blo[2], bdf[2], bw[2];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const uint32_t u = t + pass * Q;
      if (u < 2 * h) {
        const uint32_t p = u >= h ? 1u : 0u, jx = u - p * h;
        blo[pass] = bound[p][jx]; bdf[pass] = fr29_sub(bound[p][jx + h], blo[pass]);
        if (p == 0 && h > 1) bw[pass] = TAIL_EQ_S(jx < hn ? jx : jx - hn);
      }
    }
    // the host's answer: the round's challenge
    if (t == 0) {
      // three self-validating 16-byte chunks [turn, w, w, w]: the host writes each with one aligned 16-byte store and a PCIe read of an aligned
      // 16 bytes is one transaction, so when all three carry this turn's number the eight challenge words are this turn's — ONE read round trip
      // per poll (eight dependent 4-byte reads of host memory cost 12 us)
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const u32x4* m4 = reinterpret_cast<const u32x4*>(mailbox);
      uint32_t ok = 1; u32x4 c0, c1, c2; uint32_t spins = 0;
      for (;;) {
#ifndef LASSO_NO_POLL_FENCE
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // nothing cached from the previous poll
#else
        asm volatile("" ::: "memory");
#endif
        c0 = __builtin_nontemporal_load(m4); c1 = __builtin_nontemporal_load(m4 + 1); c2 = __builtin_nontemporal_load(m4 + 2);   // three 16-byte reads in flight together
        if (mail_valid(c0, c1, c2, seq0 + turn + 1)) break;   // tagged with the sequence number of the publication it enables: unique per context, never reset
        if (c0.x == LASSO_MAIL_POISON || ((++spins & 63u) == 0 && wall_clock64() > t_end)) { ok = 0; break; }   // lasso_abort's tag, or the host stopped answering
      }
      if (ok) { chal.v[0] = c0.y; chal.v[1] = c0.z; chal.v[2] = c0.w; chal.v[3] = c1.y; chal.v[4] = c1.z; chal.v[5] = c1.w; chal.v[6] = c2.y; chal.v[7] = c2.z; }
      alive = ok;
    }
    __syncthreads();
    if (!alive) return;
    TAIL_STAMP(4);
    const fr29 rs = fr29_unpack_s(chal);
    // bind in LDS: 2h lane tasks (array p, index j); the A tasks also produce the next round's weighted value.  Every task read its operands before the
    // barrier above, so the results can be stored at once
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const uint32_t u = t + pass * Q;
      if (u < 2 * h) {
        const uint32_t p = u >= h ? 1u : 0u, jx = u - p * h;
        const fr29 nb = fr29_canonical(fr29_add(blo[pass], fr29_mul(bdf[pass], rs)));
        bound[p][jx] = nb;
        if (p == 0 && h > 1) ge[jx] = fr29_mul(nb, bw[pass]);
      }
    }
    __syncthreads();
    TAIL_STAMP(5);
    m = h;
    if (m <= m_stop) {   // m_stop = 1: the heads A_c[0], B_c[0]; otherwise the bound arrays, for the host to finish the layer
      for (uint32_t u = t; u < 2 * m; u += Q) { const uint32_t p = u / m, i = u - p * m; result_store(out, ((size_t)p * ncirc + y) * m + i, fr29_pack(bound[p][i]), flag, seq0 + turn + 1); }
      if (2 * m > 64) { if (t < 2 * m) __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); __syncthreads(); }   // more than one wave stored: each releases its own stores before wave 0 signs off in row_done
      row_done(ncirc, counters, flag, seq0 + turn + 1);
      return;
    }
  }
}


#undef TAIL_EQ_S
// ------------------------------------------------------------------ K3 in eq-weighted form for the linear strategies (AND / OR / XOR / RangeCheck)
// prove_arbitrary's comb_func is g(E_1..E_alpha) * eq with g = sum_k w_k E_k LINEAR (and.rs:45-53, range_check.rs:78-86) and eq = EqPolynomial(r).evals()
// (surge.rs:156-172).  With the eq polynomial factored as in the cubic rounds (prefix of the one table + host scalars) a round needs, per polynomial k,
// only two dot products against the table prefix:  S0_k = sum_{i<half} E_k[i] eq[i],  S1_k = sum_{i<half} E_k[i+half] eq[i];  the host forms
// G(x) = sum_k w_k (S0_k + x (S1_k - S0_k)) and e(x) = s_j * eq1(r_j, x) / prod_{t<=j}(1 - r_t) * G(x) at x = 0, 1, 2.  No weight products, no eq
// binding, and the bind of the previous challenge rides in the same launch.  out[3k + {0,1}] = S0_k, S1_k (slot 2 unused).  Grid = cubic_grid(nx, alpha).
__global__ void __launch_bounds__(LASSO_BLOCK) k_dot_eqw_lb(PtrTable polys, uint32_t nx, uint32_t ny, const fr_t* __restrict__ E, size_t half, fr_t* __restrict__ partials, uint32_t* counters,
                                                             fr_t* __restrict__ out, uint32_t* flag, uint32_t seq) {
  __shared__ RedScratch S;
  const CubicGrid g = cubic_grid(nx, ny);
  const fr_t* __restrict__ z = polys.p[g.by];
  fr29_acc w0 = fr29_acc_zero(), w1 = fr29_acc_zero(); uint32_t cnt = 0;   // two dot products: double-width accumulation (fr29_mul_acc)
  for (size_t i = g.bx * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)nx * blockDim.x) {
    const fr29 es = fr29_unpack_s(E[i]);
    fr29_mul_acc(w0, fr29_unpack_u(z[i]), es);
    fr29_mul_acc(w1, fr29_unpack_u(z[i + half]), es);
    if (++cnt == 3) { fr29_acc_carry(w0); fr29_acc_carry(w1); cnt = 0; }
  }
  fr29_acc_carry(w0); fr29_acc_carry(w1);
  fr29 e[3] = {fr29_acc_reduce(w0), fr29_acc_reduce(w1), fr29_zero()};
  cubic_epilogue(e, g, partials, counters, out, flag, seq, S, 0);
}
// src != polys: the bind reads src (length 4q) and writes the bound halves to polys — the first bind of the primary sumcheck takes E itself as
// src, so surge.rs:151's clone of the lookup polynomials never happens (the sumcheck must not modify E, which the later openings read)
// AHEAD: launched before the host had the challenge, behind a gate kernel (k_gate) that leaves it in gmail
template <bool AHEAD = false>
__global__ void __launch_bounds__(LASSO_BLOCK) k_dot_eqw_fused(PtrTable src, MutPtrTable polys, uint32_t nx, uint32_t ny, const fr_t* __restrict__ E, size_t q, fr_t r, fr_t* __restrict__ partials, uint32_t* counters,
                                                                fr_t* __restrict__ out, uint32_t* flag, uint32_t seq, const uint32_t* gmail = nullptr) {
  __shared__ RedScratch S;
  const CubicGrid g = cubic_grid(nx, ny);
  fr_t* zd = polys.p[g.by];          // may alias z (in-place call): every thread reads its four elements before it writes its two
  const fr_t* z = src.p[g.by];
  if constexpr (AHEAD) {
    __shared__ uint32_t s_mail[17];
    if (!gated_challenge(gmail, seq, 8, s_mail)) return;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = s_mail[k];
  }
  const fr29 rs = fr29_unpack_s(r);
  fr29_acc w0 = fr29_acc_zero(), w1 = fr29_acc_zero(); uint32_t cnt = 0;
  for (size_t i = g.bx * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)nx * blockDim.x) {
    const fr29 z0 = bind29(z[i], z[i + 2 * q], rs), z1 = bind29(z[i + q], z[i + 3 * q], rs);
    zd[i] = fr29_pack(z0); zd[i + q] = fr29_pack(z1);
    const fr29 es = fr29_unpack_s(E[i]);
    fr29_mul_acc(w0, z0, es);
    fr29_mul_acc(w1, z1, es);
    if (++cnt == 3) { fr29_acc_carry(w0); fr29_acc_carry(w1); cnt = 0; }
  }
  fr29_acc_carry(w0); fr29_acc_carry(w1);
  fr29 e[3] = {fr29_acc_reduce(w0), fr29_acc_reduce(w1), fr29_zero()};
  cubic_epilogue(e, g, partials, counters, out, flag, seq, S, 0);
}
// The same two rounds from the lookup polynomials' INTEGER values (round 3).  E_k = T[dim] holds table entries — small integers the prover has as u32 anyway (the commitment
// of E is made from them) — so the primary sumcheck's first round and first bind need not stream the 32-byte field form: 4 bytes per element instead of 32, and a product with a
// two-limb operand is 18 multiply-adds instead of 81.  Same field elements: sum_i F(x_i) eq_i and F(lo) + r (F(hi) - F(lo)), F = Fr::from.
struct PtrTableU32 { const uint32_t* p[LASSO_MAX_PTRS]; };
__global__ void __launch_bounds__(LASSO_BLOCK) k_dot_eqw_lb_u32(PtrTableU32 polys, uint32_t nx, uint32_t ny, const fr_t* __restrict__ E, size_t half, fr_t* __restrict__ partials, uint32_t* counters,
                                                                 fr_t* __restrict__ out, uint32_t* flag, uint32_t seq) {
  __shared__ RedScratch S;
  const CubicGrid g = cubic_grid(nx, ny);
  const uint32_t* __restrict__ z = polys.p[g.by];
  fr29_acc w0 = fr29_acc_zero(), w1 = fr29_acc_zero(); uint32_t cnt = 0;
  for (size_t i = g.bx * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)nx * blockDim.x) {
    const fr29 es = fr29_unpack_s(E[i]);
    fr29_mul_acc(w0, fr29_from_u64_int(z[i]), es);          // integer * s-form: after the reduction the plain residue sum x_i eq_i
    fr29_mul_acc(w1, fr29_from_u64_int(z[i + half]), es);
    if (++cnt == 3) { fr29_acc_carry(w0); fr29_acc_carry(w1); cnt = 0; }
  }
  fr29_acc_carry(w0); fr29_acc_carry(w1);
  const fr29 r2s = fr29_r2s();   // 2^517: brings the thread's plain sum into memory (u-) form, once per thread
  fr29 e[3] = {fr29_mul(fr29_acc_reduce(w0), r2s), fr29_mul(fr29_acc_reduce(w1), r2s), fr29_zero()};
  cubic_epilogue(e, g, partials, counters, out, flag, seq, S, 0);
}
// F(lo) + r (F(hi) - F(lo)) for 32-bit integers lo, hi: two products with a two-limb operand; rx = r * 2^517 (so that integer * rx is the u-form of integer * r), canonical result
__device__ __forceinline__ fr29 bind29_u32(uint32_t lo, uint32_t hi, const fr29& rx, const fr29& r2s) {
  const bool neg = hi < lo; const uint32_t ad = neg ? lo - hi : hi - lo;
  const fr29 t = fr29_mul(fr29_from_u64_int(ad), rx), l = fr29_mul(fr29_from_u64_int(lo), r2s);
  return fr29_canonical(neg ? fr29_sub(l, t) : fr29_add(l, t));
}
__global__ void __launch_bounds__(LASSO_BLOCK) k_dot_eqw_fused_from_u32(PtrTableU32 src, MutPtrTable polys, uint32_t nx, uint32_t ny, const fr_t* __restrict__ E, size_t q, fr_t r, fr_t* __restrict__ partials,
                                                                         uint32_t* counters, fr_t* __restrict__ out, uint32_t* flag, uint32_t seq) {
  __shared__ RedScratch S;
  const CubicGrid g = cubic_grid(nx, ny);
  fr_t* __restrict__ zd = polys.p[g.by];
  const uint32_t* __restrict__ z = src.p[g.by];
  const fr29 r2s = fr29_r2s(), rx = fr29_mul(fr29_unpack_s(r), r2s);
  fr29_acc w0 = fr29_acc_zero(), w1 = fr29_acc_zero(); uint32_t cnt = 0;
  for (size_t i = g.bx * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)nx * blockDim.x) {
    const fr29 z0 = bind29_u32(z[i], z[i + 2 * q], rx, r2s), z1 = bind29_u32(z[i + q], z[i + 3 * q], rx, r2s);
    zd[i] = fr29_pack(z0); zd[i + q] = fr29_pack(z1);
    const fr29 es = fr29_unpack_s(E[i]);
    fr29_mul_acc(w0, z0, es);
    fr29_mul_acc(w1, z1, es);
    if (++cnt == 3) { fr29_acc_carry(w0); fr29_acc_carry(w1); cnt = 0; }
  }
  fr29_acc_carry(w0); fr29_acc_carry(w1);
  fr29 e[3] = {fr29_acc_reduce(w0), fr29_acc_reduce(w1), fr29_zero()};
  cubic_epilogue(e, g, partials, counters, out, flag, seq, S, 0);
}
// The tail of the primary sumcheck for a linear strategy, resident like k_cubic_tail: from q <= CUBIC_TAIL_Q indices per polynomial on, the
// remaining rounds' two dot products per polynomial (S0_k = sum_{i<h} z[i] E[i], S1_k = sum_{i<h} z[i+h] E[i]) and the binds run out of LDS,
// challenges arrive through the host mailbox, and the last publication is the heads z_k[0] = E_k(r_z).  One workgroup per polynomial; src is
// only read.  out: sums turns out[2k], out[2k+1]; final turn out[k].
template <bool BIND, int Q>
__global__ void __launch_bounds__(Q) k_linear_tail(PtrTable src, const fr_t* __restrict__ E, uint32_t q, fr_t r0, const uint32_t* mailbox, uint32_t* counters,
                                                              fr_t* __restrict__ out, uint32_t* flag, uint32_t seq0) {
  __shared__ fr29 z[2 * Q];
  __shared__ int32_t rows[2 * Q * 9];
  __shared__ int64_t strips[8 * 18];
  __shared__ int64_t cols[18];
  __shared__ fr_t chal;
  __shared__ uint32_t alive;
  const uint32_t t = threadIdx.x, y = blockIdx.x, npoly = gridDim.x;
  const uint64_t t_end = wall_clock64() + 500000000ull;   // 5 s at 100 MHz
  uint32_t m = 2 * q;
  {
    const fr29 rs = fr29_unpack_s(r0);
    const fr_t* p = src.p[y];
    for (uint32_t i = t; i < m; i += Q) z[i] = BIND ? bind29(p[i], p[i + m], rs) : fr29_unpack_u(p[i]);
  }
  __syncthreads();
  for (uint32_t turn = 0;; turn++) {
    const uint32_t h = m / 2;
    for (uint32_t u = t; u < m; u += Q) {   // rows 0..h-1: S0 terms, h..2h-1: S1 terms
      const fr29 term = fr29_mul(z[u], fr29_unpack_s(E[u < h ? u : u - h]));
#pragma unroll
      for (int k = 0; k < 9; k++) rows[u * 9 + k] = term.v[k];
    }
    __syncthreads();
    if (h > 16) {
      if (t < 8 * 18) {
        const uint32_t col = t % 18, strip = t / 18, v = col / 9, k = col - v * 9;
        const uint32_t per = (h + 7) / 8, i0 = strip * per, i1 = i0 + per < h ? i0 + per : h;
        int64_t sum = 0;
        for (uint32_t i = i0; i < i1; i++) sum += rows[(v * h + i) * 9 + k];
        strips[strip * 18 + col] = sum;
      }
      __syncthreads();
      if (t < 18) { int64_t sum = 0; for (int g = 0; g < 8; g++) sum += strips[g * 18 + t]; cols[t] = sum; }
    } else if (t < 18) {
      const uint32_t v = t / 9, k = t - v * 9;
      int64_t sum = 0;
      for (uint32_t i = 0; i < h; i++) sum += rows[(v * h + i) * 9 + k];
      cols[t] = sum;
    }
    __syncthreads();
    if (t < 2) {
      int64_t c[9];
#pragma unroll
      for (int k = 0; k < 9; k++) c[k] = cols[t * 9 + k];
      result_store(out, (size_t)y * 2 + t, fr29_pack(fr29_reduce_columns(c, 0)), flag, seq0 + turn);   // u * s products: memory form already
    }
    row_done(npoly, counters, flag, seq0 + turn);
    if (t == 0) {   // the round's challenge (three self-validating 16-byte chunks, see k_cubic_tail)
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const u32x4* m4 = reinterpret_cast<const u32x4*>(mailbox);
      uint32_t ok = 1; u32x4 c0, c1, c2; uint32_t spins = 0;
      for (;;) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        c0 = __builtin_nontemporal_load(m4); c1 = __builtin_nontemporal_load(m4 + 1); c2 = __builtin_nontemporal_load(m4 + 2);
        if (mail_valid(c0, c1, c2, seq0 + turn + 1)) break;   // tagged with the sequence number of the publication it enables: unique per context, never reset
        if (c0.x == LASSO_MAIL_POISON || ((++spins & 63u) == 0 && wall_clock64() > t_end)) { ok = 0; break; }   // lasso_abort's tag, or the host stopped answering
      }
      if (ok) { chal.v[0] = c0.y; chal.v[1] = c0.z; chal.v[2] = c0.w; chal.v[3] = c1.y; chal.v[4] = c1.z; chal.v[5] = c1.w; chal.v[6] = c2.y; chal.v[7] = c2.z; }
      alive = ok;
    }
    __syncthreads();
    if (!alive) return;
    const fr29 rs = fr29_unpack_s(chal);
    fr29 nb;
    if (t < h) nb = fr29_canonical(fr29_add(z[t], fr29_mul(fr29_sub(z[t + h], z[t]), rs)));
    __syncthreads();
    if (t < h) z[t] = nb;
    __syncthreads();
    m = h;
    if (m == 1) {
      if (t == 0) result_store(out, y, fr29_pack(z[0]), flag, seq0 + turn + 1);
      row_done(npoly, counters, flag, seq0 + turn + 1);
      return;
    }
  }
}


// ------------------------------------------------------------------ g = S::combine_lookups (subtables/*.rs)
// AND/OR/XOR (and.rs:45-53) and RangeCheck (range_check.rs:78-86): g = sum_i 2^(i*inc) * vals[i] is LINEAR, so along the line
// lo + x*(hi - lo) it is g(lo) + x*(g(hi) - g(lo)): two weighted sums per index, whatever the number of evaluation points.
// ws[i] = weights in s-form (LDS).  Returns a reduced u-form value.
__device__ __forceinline__ fr29 weighted_sum(const PtrTable& polys, size_t idx, uint32_t alpha, const fr29* ws) {
  fr29 s = fr29_zero();
  for (uint32_t j = 0; j < alpha; j++) s = fr29_weak(fr29_add(s, fr29_mul(fr29_unpack_u(polys.p[j][idx]), ws[j])));
  return s;
}
// LT (lt.rs:62-71): sum_i LT[i] * prod_{j<i} EQ[j], values in s-form so that products of any degree stay in s-form
template <int A>
__device__ __forceinline__ fr29 combine_lt(const fr29* vals, uint32_t c) {
  fr29 sum = fr29_zero(), eq_prod = fr29_one_s();
#pragma unroll
  for (int i = 0; i < A / 2; i++) if ((uint32_t)i < c) { sum = fr29_weak(fr29_add(sum, fr29_mul(vals[2 * i], eq_prod))); eq_prod = fr29_mul(vals[2 * i + 1], eq_prod); }
  return sum;
}
__device__ __forceinline__ void load_weights(const StrategyDev& S, const WeightTable& W, fr29* ws) {
  if (threadIdx.x < S.alpha) ws[threadIdx.x] = fr29_unpack_s(W.w[threadIdx.x]);
  __syncthreads();
}

// K3: prove_arbitrary round (sumcheck.rs:165-237), linear strategies: partials[bx*3 + x], x = 0, 1, 2 (sumcheck degree 2)
__global__ void __launch_bounds__(LASSO_BLOCK) k_combine_round_linear(StrategyDev S, PtrTable polys, const fr_t* __restrict__ eq, WeightTable W, size_t half, fr_t* __restrict__ partials) {
  __shared__ RedScratch R;
  __shared__ fr29 ws[LASSO_MAX_ALPHA];
  load_weights(S, W, ws);
  fr29 acc[3] = {fr29_zero(), fr29_zero(), fr29_zero()}; uint32_t cnt = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    const fr29 g0 = weighted_sum(polys, i, S.alpha, ws), g1 = weighted_sum(polys, i + half, S.alpha, ws);
    const fr29 e0 = fr29_unpack_u(eq[i]), e1 = fr29_unpack_u(eq[i + half]);
    const fr29 g2 = fr29_weak(fr29_add(g1, fr29_sub(g1, g0))), e2 = fr29_weak(fr29_add(e1, fr29_sub(e1, e0)));
    acc[0] = fr29_weak(fr29_add(acc[0], fr29_mul(g0, e0)));
    acc[1] = fr29_weak(fr29_add(acc[1], fr29_mul(g1, e1)));
    acc[2] = fr29_weak(fr29_add(acc[2], fr29_mul(g2, e2)));
    if ((++cnt & 63u) == 0) { acc[0] = fr29_mul(acc[0], fr29_one_s()); acc[1] = fr29_mul(acc[1], fr29_one_s()); acc[2] = fr29_mul(acc[2], fr29_one_s()); }
  }
  store_block_partials<3>(acc, 3, partials + (size_t)blockIdx.x * 3, 5, R);   // (u * s) * u: 2^5 short
}
// K3 for LT: degree = C + 1.  A = compile-time bound on NUM_MEMORIES = 2C (dispatch only), D = bound on the degree, T = lanes that share one index.
// g(x) eq(x) = e(x) * sum_i LT_i(x) * prod_{j<i} EQ_j(x) at the points x = 0..degree (lt.rs:62-71 inside sumcheck.rs:179-218), every factor linear in x.
// HORNER FORM (round 3): sum_i LT_i prod_{j<i} EQ_j = LT_0 + EQ_0 (LT_1 + EQ_1 (LT_2 + ... + EQ_{C-2} LT_{C-1})), so walking the memories from the LAST to the
// first needs ONE product per memory and point — t <- LT_m + EQ_m t — instead of the forward form's two (term = LT_i * run, run = EQ_i * run), and one more per
// point for the eq weight: C (D + 1) products per index instead of 2 C (D + 1) — 288 instead of 576 at C = 16 — for the same field elements (EQ_{C-1} enters no
// term of the sum and is not even loaded).  The memories are STREAMED: per point a lane keeps only t and the running sum over indices, loading one (LT_m, EQ_m)
// pair of lines at a time and stepping them from point to point by addition.
// Radix and magnitudes.  Horner feeds t back through a product at every step, so the multiplier must be SMALL or t grows geometrically: a line stepped to x = 17
// is up to 18 values of < p each, and in s-form (x 32) that is 2^261.2 for curve25519 and 2^262.8 for BN254 — above the Montgomery radix 2^261, a growth factor
// above 1.  Everything therefore stays in u-form (factor 18 p / 2^261 <= 0.11: |t| <= 24 p throughout), and the 2^5 each u * u product comes out short is
// carried by the DATA: the caller has multiplied LT_m by kappa_m = 32^-(C-1-m) once, before the first round (k_lt_prescale; binding is linear, so the arrays stay
// scaled from round to round), the recursion  t_m = kappa_m LT_m + (EQ_m * t_{m+1}) / 32  then yields t_0 = T_0 / 32^(C-1), the weighted sum comes out as
// sum e T_0 / 32^C, and the block partials are multiplied by 32^C (`scale`).  tests/cpp/test_poly_math_host.cpp drives this arithmetic under UBSan with extreme inputs.
// Registers.  With all D + 1 points in one lane the state is 2 (D + 1) field elements = 324 VGPRs at D = 17: the compiler parked half of it in AGPRs (a copy in and out
// around every use) and spilled the rest, and the kernel ran at 11 cycles per instruction.  So T lanes share an index (adjacent lanes: their loads of the same
// lines coalesce into one request), each walking PPG = ceil((D + 1) / T) <= 6 consecutive points from its own start x0 = lane * PPG (lt_line_at): 12 state elements,
// two waves per SIMD, nothing in AGPRs or scratch.
// indices between two folds of the running sums: a term e * t is below p + |e| |t| / 2^261 <= 3.6 p (18 p * 24 p; BN254's p / 2^261 = 2^-7.4), so 32 of them stay
// below 2^261 and limb 8 inside the loose bound 2^30 that the fold's product requires
#define LT_FOLD_EVERY() 32u
#define LT_REP(M) M(0) M(1) M(2) M(3) M(4) M(5)
#define LT_DECL(k) fr29 sum##k = fr29_zero(), t##k = fr29_zero();
#define LT_TOP(k) if constexpr (k < PPG) { if (x0 + k <= degree) { t##k = lt; lt = lt_line_step(lt, dlt); } }
#define LT_STEP(k) if constexpr (k < PPG) { if (x0 + k <= degree) { t##k = lt_horner_step(lt, eqv, t##k); lt = lt_line_step(lt, dlt); eqv = lt_line_step(eqv, deq); } }
#define LT_STEP_PROD(k) if constexpr (k < PPG) { if (x0 + k <= degree) { t##k = fr29_weak(fr29_mul(eqv, t##k)); eqv = lt_line_step(eqv, deq); } }
#define LT_ACC(k) if constexpr (k < PPG) { if (x0 + k <= degree) { sum##k = lt_weighted_acc(sum##k, ecur, t##k); ecur = lt_line_step(ecur, edif); } }
#define LT_FOLD(k) if constexpr (k < PPG) sum##k = fr29_mul(sum##k, fr29_one_s());
#define LT_OUT(k) if constexpr (k < PPG) mine[k] = fr29_mul(sum##k, sc);
// the per-point operations of the LT round (extracted verbatim by tests/test_host_arith_cpp.py and driven under UBSan): a line stepped to the next evaluation
// point; a line started at point x0 (lo reduced, d a difference of reduced values, x0 <= 17); one Horner step t <- LT + (EQ * t) (u-form operands: the product is
// 2^5 short, see above); the eq-weighted accumulation sum += e * t
__device__ __forceinline__ fr29 lt_line_step(const fr29& v, const fr29& d) {
  return fr29_weak(fr29_add(v, d));
}
__device__ __forceinline__ fr29 lt_line_at(const fr29& lo, const fr29& d, uint32_t x0) {
  fr29 r; int64_t c = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { const int64_t x = (int64_t)lo.v[k] + (int64_t)x0 * d.v[k] + c; r.v[k] = (int32_t)(x & FR29_MASK); c = x >> 29; }
  r.v[8] = (int32_t)((int64_t)lo.v[8] + (int64_t)x0 * d.v[8] + c);
  return r;
}
__device__ __forceinline__ fr29 lt_horner_step(const fr29& lt, const fr29& eqv, const fr29& t) {
  return fr29_weak(fr29_add(lt, fr29_mul(eqv, t)));
}
__device__ __forceinline__ fr29 lt_weighted_acc(const fr29& sum, const fr29& e, const fr29& t) {
  return fr29_weak(fr29_add(sum, fr29_mul(e, t)));
}
// LT_m <- 32^-(C-1-m) LT_m for the memories 2m of an LT strategy (kappa in memory form, s-form at use: mul(u, s) = u-form); grid = (blocks over i, C - 1): m = C - 1 has kappa = 1
struct LtKappa { fr_t k[LASSO_MAX_ALPHA / 2]; };
__global__ void __launch_bounds__(LASSO_BLOCK) k_lt_prescale(PtrTable src, MutPtrTable polys, LtKappa K, size_t n) {   // src[2m] may be polys[2m] (in place)
  const uint32_t m = blockIdx.y; fr_t* z = polys.p[2 * m]; const fr_t* x = src.p[2 * m];
  const fr29 ks = fr29_unpack_s(K.k[m]);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) z[i] = fr29_store(fr29_mul(fr29_unpack_u(x[i]), ks));
}
// PROD (the unconfirmed Spark strategy: g = prod_m E_m, degree C): the same walk without the LT terms — t starts as the last memory's line and every other memory multiplies it
// once, t <- (E_m * t) / 32; memory m is polys[m] (no LT / EQ pairs), nothing is pre-scaled, and the block sums come out as sum e prod_m E_m / 32^C like LT's (same `scale`).
template <int A, int D, int T, bool PROD = false>
__global__ void __launch_bounds__(LASSO_BLOCK) k_combine_round_lt(StrategyDev S, PtrTable polys, const fr_t* __restrict__ eq, fr_t scale, size_t half, uint32_t degree, fr_t* __restrict__ partials) {
  constexpr int PPG = (D + 1 + T - 1) / T;
  constexpr uint32_t SLOTS = LASSO_BLOCK / T;
  static_assert(PPG <= 6, "LT_REP lists 6 points per lane");
  __shared__ RedScratch R;
  const uint32_t slot = threadIdx.x / T, pg = threadIdx.x - slot * T, x0 = pg * PPG;
  LT_REP(LT_DECL)
  uint32_t cnt = 0;
  if (slot < SLOTS && x0 <= degree)
  for (size_t i = blockIdx.x * (size_t)SLOTS + slot; i < half; i += (size_t)gridDim.x * SLOTS) {
    {   // innermost term: t(x) = LT_{C-1}(x)   (PROD: the last memory's line)
      const fr_t* __restrict__ pl = polys.p[PROD ? S.c - 1 : 2 * (S.c - 1)];
      const fr29 lo = fr29_unpack_u(pl[i]), dlt = fr29_sub(fr29_unpack_u(pl[i + half]), lo);
      fr29 lt = lt_line_at(lo, dlt, x0);
      LT_REP(LT_TOP)
    }
    if constexpr (PROD) {
      for (uint32_t m = S.c - 1; m-- > 0;) {   // t <- (E_m * t) / 32
        const fr_t* __restrict__ pe = polys.p[m];
        const fr29 eo = fr29_unpack_u(pe[i]), deq = fr29_sub(fr29_unpack_u(pe[i + half]), eo);
        fr29 eqv = lt_line_at(eo, deq, x0);
        LT_REP(LT_STEP_PROD)
      }
    } else
    for (uint32_t m = S.c - 1; m-- > 0;) {   // t <- LT_m + (EQ_m * t) / 32   (LT_m pre-scaled)
      const fr_t* __restrict__ pl = polys.p[2 * m]; const fr_t* __restrict__ pe = polys.p[2 * m + 1];
      const fr29 lo = fr29_unpack_u(pl[i]), dlt = fr29_sub(fr29_unpack_u(pl[i + half]), lo);
      const fr29 eo = fr29_unpack_u(pe[i]), deq = fr29_sub(fr29_unpack_u(pe[i + half]), eo);
      fr29 lt = lt_line_at(lo, dlt, x0), eqv = lt_line_at(eo, deq, x0);
      LT_REP(LT_STEP)
    }
    {   // weight by the eq polynomial's line and accumulate over the indices
      const fr29 e0 = fr29_unpack_u(eq[i]), edif = fr29_sub(fr29_unpack_u(eq[i + half]), e0);
      fr29 ecur = lt_line_at(e0, edif, x0);
      LT_REP(LT_ACC)
    }
    if (++cnt >= LT_FOLD_EVERY()) { cnt = 0; LT_REP(LT_FOLD) }
  }
  const fr29 sc = fr29_unpack_s(scale);   // u-form sums of (value / 32^C) times the s-form of 32^C: u-form of the value
  fr29 mine[6];
#pragma unroll
  for (int k = 0; k < 6; k++) mine[k] = fr29_zero();
  LT_REP(LT_OUT)
  // block sums, one point group at a time: lanes of group g contribute their PPG sums, everybody else zeros (no register array is indexed by a run-time value)
#pragma unroll
  for (int g = 0; g < T; g++) {
#pragma unroll
    for (int k0 = 0; k0 < PPG; k0 += 3) {
      fr29 grp[3];
#pragma unroll
      for (int v = 0; v < 3; v++) {
        grp[v] = fr29_zero();
        if (k0 + v < PPG) {
#pragma unroll
          for (int l = 0; l < 9; l++) grp[v].v[l] = pg == (uint32_t)g ? mine[k0 + v].v[l] : 0;
        }
      }
      block_columns<3>(grp, R);
      const uint32_t pt = (uint32_t)(g * PPG + k0) + threadIdx.x;
      if (threadIdx.x < 3 && k0 + (int)threadIdx.x < PPG && pt <= degree) partials[(size_t)blockIdx.x * (degree + 1) + pt] = columns_to_fr(R, threadIdx.x, 0);
    }
  }
}
#undef LT_REP
#undef LT_DECL
#undef LT_TOP
#undef LT_STEP
#undef LT_STEP_PROD
#undef LT_ACC
#undef LT_FOLD
#undef LT_OUT
// The FIRST round of the LT sumcheck from the lookup polynomials' integer values.  Before any bind E_k = T[dim_k] holds subtable entries, and the LT / EQ subtables hold
// bits (lt.rs:17-44): every line LT_m(x), EQ_m(x) = lo + x (hi - lo) is an integer in [-16, 17] for x <= 17, and the whole Horner walk t <- LT_m + EQ_m t is exact INTEGER
// arithmetic — |t| <= 17 (17^16 - 1) / 16 < 2^67, a 128-bit multiply-add by a small number per memory and point instead of a 256-bit field product.  Only the eq weight is
// a field product (integer t as three signed 29-bit limbs times the eq line pre-multiplied by 2^517, so that the Montgomery product lands in memory form).  Round 0 is half of the sumcheck's
// work; it drops from 288 field products per index at C = 16 to 18 cheap ones, reading 4 bytes per element.  Requires every entry <= 1 (the caller checks its tables).
// a signed integer of magnitude < 2^87 as three signed 29-bit limbs (limbs 3..8 literal zeros: a product with it is 27 multiply-adds)
__device__ __forceinline__ fr29 lt_int_limbs(__int128 t) {
  const bool neg = t < 0; const unsigned __int128 mag = neg ? (unsigned __int128)(-t) : (unsigned __int128)t;
  const int32_t l0 = (int32_t)((uint64_t)mag & FR29_MASK), l1 = (int32_t)((uint64_t)(mag >> 29) & FR29_MASK), l2 = (int32_t)((uint64_t)(mag >> 58) & FR29_MASK);
  fr29 r = fr29_zero(); r.v[0] = neg ? -l0 : l0; r.v[1] = neg ? -l1 : l1; r.v[2] = neg ? -l2 : l2;
  return r;
}
// the eq line's end point times 2^517: an INTEGER times it, Montgomery-reduced by 2^261, is (integer * e) 2^256 — the memory form, no correction needed
__device__ __forceinline__ fr29 lt_eq_times_r2(const fr_t& e) {
  return fr29_mul(fr29_unpack_s(e), fr29_r2s());
}
template <int A, int D, int T>
__global__ void __launch_bounds__(LASSO_BLOCK) k_combine_round_lt_u32(StrategyDev S, PtrTableU32 polys, const fr_t* __restrict__ eq, size_t half, uint32_t degree, fr_t* __restrict__ partials,
                                                                      uint32_t* __restrict__ bad) {
  constexpr int PPG = (D + 1 + T - 1) / T;
  uint32_t seen = 0;   // OR of every entry read: the integer walk is only exact (and only fits its containers) for entries 0 / 1 — anything else is reported, not computed
  constexpr uint32_t SLOTS = LASSO_BLOCK / T;
  static_assert(PPG <= 6, "at most 6 points per lane");
  __shared__ RedScratch R;
  const uint32_t slot = threadIdx.x / T, pg = threadIdx.x - slot * T, x0 = pg * PPG;
  fr29 sum[PPG];
#pragma unroll
  for (int k = 0; k < PPG; k++) sum[k] = fr29_zero();
  uint32_t cnt = 0;
  if (slot < SLOTS && x0 <= degree)
  for (size_t i = blockIdx.x * (size_t)SLOTS + slot; i < half; i += (size_t)gridDim.x * SLOTS) {
    __int128 t[PPG];
    {
      const uint32_t* __restrict__ pl = polys.p[2 * (S.c - 1)];
      const uint32_t u0 = pl[i], u1 = pl[i + half]; seen |= u0 | u1 | polys.p[2 * (S.c - 1) + 1][i] | polys.p[2 * (S.c - 1) + 1][i + half];   // the last EQ memory never enters the walk: checked all the same
      const int32_t lo = (int32_t)u0, d = (int32_t)u1 - lo;
#pragma unroll
      for (int k = 0; k < PPG; k++) t[k] = lo + (int32_t)(x0 + k) * d;
    }
    for (uint32_t m = S.c - 1; m-- > 0;) {
      const uint32_t* __restrict__ pl = polys.p[2 * m]; const uint32_t* __restrict__ pe = polys.p[2 * m + 1];
      const uint32_t a0 = pl[i], a1 = pl[i + half], b0 = pe[i], b1 = pe[i + half]; seen |= a0 | a1 | b0 | b1;
      const int32_t llo = (int32_t)a0, ld = (int32_t)a1 - llo, elo = (int32_t)b0, ed = (int32_t)b1 - elo;
#pragma unroll
      for (int k = 0; k < PPG; k++) { const int32_t x = (int32_t)(x0 + k); t[k] = (__int128)(llo + x * ld) + (__int128)(elo + x * ed) * t[k]; }
    }
    {
      // the eq line times 2^517 (two products per index): an INTEGER t times e 2^517, Montgomery-reduced by 2^261, is t e 2^256 — the memory form of t e, no correction needed
      const fr29 e0 = fr29_weak(lt_eq_times_r2(eq[i])), edif = fr29_sub(lt_eq_times_r2(eq[i + half]), e0);
      fr29 ecur = lt_line_at(e0, edif, x0);
#pragma unroll
      for (int k = 0; k < PPG; k++) if (x0 + k <= degree) {
        sum[k] = lt_weighted_acc(sum[k], lt_int_limbs(t[k]), ecur);
        ecur = lt_line_step(ecur, edif);
      }
    }
    if (++cnt >= LT_FOLD_EVERY()) {
      cnt = 0;
#pragma unroll
      for (int k = 0; k < PPG; k++) sum[k] = fr29_mul(sum[k], fr29_one_s());
    }
  }
  if (seen > 1u) *bad = 1u;   // host-mapped word, read by the host behind the result's flag (lasso_sumcheck_combine_round_lt_u32)
#pragma unroll
  for (int g = 0; g < T; g++) {
#pragma unroll
    for (int k0 = 0; k0 < PPG; k0 += 3) {
      fr29 grp[3];
#pragma unroll
      for (int v = 0; v < 3; v++) {
        grp[v] = fr29_zero();
        if (k0 + v < PPG) {
#pragma unroll
          for (int l = 0; l < 9; l++) grp[v].v[l] = pg == (uint32_t)g ? sum[k0 + v < PPG ? k0 + v : 0].v[l] : 0;
        }
      }
      block_columns<3>(grp, R);
      const uint32_t pt = (uint32_t)(g * PPG + k0) + threadIdx.x;
      if (threadIdx.x < 3 && k0 + (int)threadIdx.x < PPG && pt <= degree) partials[(size_t)blockIdx.x * (degree + 1) + pt] = columns_to_fr(R, threadIdx.x, 0);
    }
  }
}
// K10: claim = sum_k eq[k] * g(E(k))  (subtables/mod.rs:187-216)
template <int A>
__global__ void __launch_bounds__(LASSO_BLOCK) k_combine_claim(StrategyDev S, PtrTable polys, const fr_t* __restrict__ eq, WeightTable W, size_t n, fr_t* __restrict__ partials) {
  __shared__ RedScratch R;
  __shared__ fr29 ws[LASSO_MAX_ALPHA];
  load_weights(S, W, ws);
  const bool lt = S.kind == 3, prod = S.kind == 5;   // LASSO_LT, LASSO_SPARK_UNCONFIRMED
  fr29 acc[1] = {fr29_zero()}; uint32_t cnt = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    fr29 g;
    if (prod) {   // g = prod_m E_m: s-form values, so the running product stays in s-form whatever the degree
      g = fr29_unpack_s(polys.p[S.c - 1][i]);
      for (uint32_t m = S.c - 1; m-- > 0;) g = fr29_mul(fr29_unpack_s(polys.p[m][i]), g);
    } else if (lt) {   // lt.rs:62-71 streamed over the memories: running sum and running product, no array of values (which ended up in scratch memory at A >= 16)
      // Horner from the last memory: g = LT_0 + EQ_0 (LT_1 + EQ_1 (...  + EQ_{C-2} LT_{C-1})): one product per memory instead of two
      g = fr29_unpack_s(polys.p[2 * (S.c - 1)][i]);
      for (uint32_t m = S.c - 1; m-- > 0;) g = fr29_weak(fr29_add(fr29_unpack_s(polys.p[2 * m][i]), fr29_mul(fr29_unpack_s(polys.p[2 * m + 1][i]), g)));
    } else g = weighted_sum(polys, i, S.alpha, ws);
    acc_add(acc[0], fr29_mul(g, fr29_unpack_u(eq[i])), cnt);
  }
  store_block_partials<1>(acc, 1, partials + blockIdx.x, (lt || prod) ? 0 : 5, R);
}

// K12: out[p] = sum_i polys[p][i] * w[i]; 1-D grid of nx*ny workgroups in cubic_grid order (the workgroups of one index range and different polynomials
// share an XCD, so the weight vector w is fetched from HBM once: the (x, polynomial) grid re-read it once per polynomial, PMC traffic 1.31x);
// partials[p*nx + bx]
__global__ void __launch_bounds__(LASSO_BLOCK) k_multi_dot(PtrTable polys, uint32_t nx, uint32_t ny, const fr_t* __restrict__ w, size_t n, fr_t* __restrict__ partials) {
  __shared__ RedScratch R;
  const CubicGrid g = cubic_grid(nx, ny);
  const fr_t* __restrict__ z = polys.p[g.by];
  // a sum of products: double-width accumulation, one Montgomery reduction per thread (fr29_mul_acc)
  fr29_acc wa = fr29_acc_zero(); uint32_t cnt = 0;
  for (size_t i = g.bx * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)nx * blockDim.x) {
    fr29_mul_acc(wa, fr29_unpack_u(z[i]), fr29_unpack_s(w[i]));
    if (++cnt == 3) { fr29_acc_carry(wa); cnt = 0; }
  }
  fr29_acc_carry(wa);
  fr29 acc[1] = {fr29_acc_reduce(wa)};
  store_block_partials<1>(acc, 1, partials + (size_t)g.by * nx + g.bx, 0, R);
}

// ------------------------------------------------------------------ K6: eq evals (eq_poly.rs:22-38)
// small table: out[x] = prod_j (bit_j(x) ? r[j] : 1 - r[j]), bit 0 of the product order = most significant bit of x
struct RTable { fr_t r[32]; };
__global__ void k_eq_small(RTable R, uint32_t ell, fr_t scale, fr_t* __restrict__ out) {
  size_t x = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (x >= ((size_t)1 << ell)) return;
  fr29 p = fr29_unpack_u(scale);   // 1 normally; in slab mode the eq factor of the rank's low index bits
  const fr29 one_s = fr29_one_s();
  for (uint32_t j = 0; j < ell; j++) { const bool bit = (x >> (ell - 1 - j)) & 1; const fr29 rs = fr29_unpack_s(R.r[j]); p = fr29_mul(bit ? rs : fr29_sub(one_s, rs), p); }
  out[x] = fr29_store(p);
}
// both factor tables of a large eq table in ONE launch (the first hi_blocks workgroups build hi, scale folded in; the rest lo): a launch of a few hundred threads costs 8-15 us
// whatever it computes, and every streaming layer of the grand-product argument paid two of them in front of k_eq_outer
struct RTable16 { fr_t r[16]; };
__global__ void k_eq_small2(RTable16 Rh, uint32_t hi_bits, fr_t scale, fr_t* __restrict__ hi, uint32_t hi_blocks, RTable16 Rl, uint32_t lo_bits, fr_t* __restrict__ lo) {
  const bool is_lo = blockIdx.x >= hi_blocks;
  const uint32_t ell = is_lo ? lo_bits : hi_bits;
  const size_t x = (blockIdx.x - (is_lo ? hi_blocks : 0u)) * (size_t)blockDim.x + threadIdx.x;
  if (x >= ((size_t)1 << ell)) return;
  fr29 p = is_lo ? fr29_unpack_u(fr_one()) : fr29_unpack_u(scale);
  const fr29 one_s = fr29_one_s();
  // one loop per table: `is_lo ? Rl.r[j] : Rh.r[j]` made the compiler copy a 512-byte argument struct into scratch (1040 bytes per lane) and index it there — ~2.5 us per step of
  // the chain, 18-34 us for a kernel that computes two tables of <= 2^11 entries in front of round 0 of every large layer (profiles/r05_kernel_trace_one_proof_2p24.csv); with
  // the block-uniform branch outside, r[j] is a scalar load from the argument segment
  if (is_lo) { for (uint32_t j = 0; j < ell; j++) { const bool bit = (x >> (ell - 1 - j)) & 1; const fr29 rs = fr29_unpack_s(Rl.r[j]); p = fr29_mul(bit ? rs : fr29_sub(one_s, rs), p); } lo[x] = fr29_store(p); }
  else { for (uint32_t j = 0; j < ell; j++) { const bool bit = (x >> (ell - 1 - j)) & 1; const fr29 rs = fr29_unpack_s(Rh.r[j]); p = fr29_mul(bit ? rs : fr29_sub(one_s, rs), p); } hi[x] = fr29_store(p); }
}
// the same two tables from a point the gate left in device memory (k_gate_point): r_j = gp[8 j ..), scale = entry hi_bits + lo_bits
__global__ void k_eq_small2_mem(const uint32_t* __restrict__ gp, uint32_t seq, uint32_t hi_bits, fr_t* __restrict__ hi, uint32_t hi_blocks, uint32_t lo_bits, fr_t* __restrict__ lo) {
  if (!gate_point_ok(gp, seq)) return;
  const bool is_lo = blockIdx.x >= hi_blocks;
  const uint32_t ell = is_lo ? lo_bits : hi_bits, j0 = is_lo ? hi_bits : 0u;
  const size_t x = (blockIdx.x - (is_lo ? hi_blocks : 0u)) * (size_t)blockDim.x + threadIdx.x;
  if (x >= ((size_t)1 << ell)) return;
  fr29 p = is_lo ? fr29_unpack_u(fr_one()) : fr29_unpack_u(gate_point_fr(gp, hi_bits + lo_bits));
  const fr29 one_s = fr29_one_s();
  for (uint32_t j = 0; j < ell; j++) { const bool bit = (x >> (ell - 1 - j)) & 1; const fr29 rs = fr29_unpack_s(gate_point_fr(gp, j0 + j)); p = fr29_mul(bit ? rs : fr29_sub(one_s, rs), p); }
  (is_lo ? lo : hi)[x] = fr29_store(p);
}
// out[x] = hi[x >> lo_bits] * lo[x & mask]
__global__ void __launch_bounds__(LASSO_BLOCK) k_eq_outer(const fr_t* __restrict__ hi, const fr_t* __restrict__ lo, uint32_t lo_bits, size_t n, fr_t* __restrict__ out, const uint32_t* __restrict__ gp = nullptr, uint32_t seq = 0) {
  if (gp != nullptr && !gate_point_ok(gp, seq)) return;   // behind a point gate that failed: nothing is touched
  const size_t mask = ((size_t)1 << lo_bits) - 1;
  for (size_t x = blockIdx.x * (size_t)blockDim.x + threadIdx.x; x < n; x += (size_t)gridDim.x * blockDim.x) out[x] = fr29_store(fr29_mul(fr29_unpack_u(hi[x >> lo_bits]), fr29_unpack_s(lo[x & mask])));
}

// ------------------------------------------------------------------ K7: product tree layer (grand_product.rs:20-36)
__global__ void __launch_bounds__(LASSO_BLOCK) k_gp_layer(const fr_t* __restrict__ in, size_t half, fr_t* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) out[i] = fr29_store(fr29_mul(fr29_unpack_u(in[i]), fr29_unpack_s(in[i + half])));
}
// two layers per launch: thread i < q = len/4 holds in[i], in[i+q], in[i+2q], in[i+3q], writes the next layer's o1[i] = in[i]*in[i+2q], o1[i+q] = in[i+q]*in[i+3q]
// (the pairs (j, j + len/2) of grand_product.rs:25-30) and the layer after it, o2[i] = o1[i]*o1[i+q], without reading o1 back: 4 reads + 3 writes instead of
// 6 + 3 for the same two layers.  Operands go through the memory form exactly as k_gp_layer reads them.
__global__ void __launch_bounds__(LASSO_BLOCK) k_gp_layer2(const fr_t* __restrict__ in, size_t q, fr_t* __restrict__ o1, fr_t* __restrict__ o2) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
    const fr_t a = fr29_store(fr29_mul(fr29_unpack_u(in[i]), fr29_unpack_s(in[i + 2 * q])));
    const fr_t b = fr29_store(fr29_mul(fr29_unpack_u(in[i + q]), fr29_unpack_s(in[i + 3 * q])));
    o1[i] = a; o1[i + q] = b;
    o2[i] = fr29_store(fr29_mul(fr29_unpack_u(a), fr29_unpack_s(b)));
  }
}
// the remaining small layers in one workgroup: in has `len` elements (len <= 2*blockDim.x), layers are laid out back to back
__global__ void k_gp_tail(fr_t* __restrict__ tree, size_t len) {
  fr_t* in = tree;
  while (len > 2) {
    size_t half = len / 2; fr_t* out = in + len;
    for (size_t i = threadIdx.x; i < half; i += blockDim.x) out[i] = fr29_store(fr29_mul(fr29_unpack_u(in[i]), fr29_unpack_s(in[i + half])));
    __threadfence_block();
    __syncthreads();
    in = out; len = half;
  }
}

// ------------------------------------------------------------------ K8: Reed-Solomon fingerprints (memory_checking.rs:236-310)
// h(a, v, t) = t*gamma^2 + v*gamma + a - tau
__global__ void __launch_bounds__(LASSO_BLOCK) k_fingerprint_ops(const fr_t* __restrict__ table, const uint32_t* __restrict__ dim, const fr_t* __restrict__ read, size_t s,
                                                                  fr_t gamma, fr_t gamma2, fr_t tau, fr_t* __restrict__ out_r, fr_t* __restrict__ out_w) {
  const fr29 gs = fr29_unpack_s(gamma), g2s = fr29_unpack_s(gamma2), g2u = fr29_unpack_u(gamma2), tu = fr29_unpack_u(tau), r2s = fr29_r2s();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < s; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t a = dim[i];
    fr29 h = fr29_add(fr29_mul(fr29_unpack_u(read[i]), g2s), fr29_mul(fr29_unpack_u(table[a]), gs));
    h = fr29_canonical(fr29_sub(fr29_add(h, fr29_mul(fr29_from_u64_int(a), r2s)), tu));
    out_r[i] = fr29_pack(h);
    out_w[i] = fr29_store(fr29_add(h, g2u));   // ts+1: (t+1)*gamma^2 = t*gamma^2 + gamma^2
  }
}
// The same fingerprints TOGETHER WITH the first product layer of the two trees (grand_product.rs:20-36 on the leaves just computed): thread i < s/2 makes the
// leaves i and i + s/2 of both circuits, stores them (the bottom layer's sumcheck reads them) and multiplies the pair while it still holds them — the
// 2 x 32 s bytes that k_gp_layer would read straight back never leave the chip.  l1_r / l1_w: s/2 elements each.  Operands go through the memory form exactly
// as k_gp_layer reads them, so the tree is bit-identical.
// RU32: the read timestamps as 32-bit integers (capacity mode keeps dim / read compact): t * gamma^2 = mul(integer t, gamma^2 * 2^517) lands in the same u-form — same canonical leaf
template <bool RU32>
__global__ void __launch_bounds__(LASSO_BLOCK) k_fingerprint_ops_l1(const fr_t* __restrict__ table, const uint32_t* __restrict__ dim, const void* __restrict__ read_any, size_t s,
                                                                     fr_t gamma, fr_t gamma2, fr_t tau, fr_t* __restrict__ out_r, fr_t* __restrict__ out_w, fr_t* __restrict__ l1_r, fr_t* __restrict__ l1_w,
                                                                     uint32_t store_leaves) {
  const fr29 gs = fr29_unpack_s(gamma), g2s = fr29_unpack_s(gamma2), g2u = fr29_unpack_u(gamma2), tu = fr29_unpack_u(tau), r2s = fr29_r2s();
  const fr29 g2r = fr29_mul(g2s, r2s);   // gamma^2 * 2^517: an INTEGER times it is (integer * gamma^2) in u-form
  const fr_t* __restrict__ read = reinterpret_cast<const fr_t*>(read_any); const uint32_t* __restrict__ read32 = reinterpret_cast<const uint32_t*>(read_any);
  const size_t half = s / 2;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    fr_t lr[2], lw[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const size_t k = i + e * half;
      const uint32_t a = dim[k];
      fr29 h = fr29_add(RU32 ? fr29_mul(fr29_from_u64_int(read32[k]), g2r) : fr29_mul(fr29_unpack_u(read[k]), g2s), fr29_mul(fr29_unpack_u(table[a]), gs));
      h = fr29_canonical(fr29_sub(fr29_add(h, fr29_mul(fr29_from_u64_int(a), r2s)), tu));
      lr[e] = fr29_pack(h); lw[e] = fr29_store(fr29_add(h, g2u));
      if (store_leaves) { out_r[k] = lr[e]; out_w[k] = lw[e]; }   // 0: capacity mode's leafless trees (lasso_fingerprint_ops_gp_upper; out_r / out_w are then NULL), and the timing experiment LASSO_EXP_NO_LEAF_STORE
    }
    l1_r[i] = fr29_store(fr29_mul(fr29_unpack_u(lr[0]), fr29_unpack_s(lr[1])));
    l1_w[i] = fr29_store(fr29_mul(fr29_unpack_u(lw[0]), fr29_unpack_s(lw[1])));
  }
}
// Capacity mode (trees kept without their leaf layer): the fingerprints of ONE strip set of the bottom layer, recomputed where the bottom layer's two streaming rounds need them.
// The layer's A = leaves[0 .. s/2), B = leaves[s/2 .. s); a round on an index range [i0, i0 + cs) reads, of each array, `nstrips` strips `stride` apart (2 strips s/4 apart for
// the first round, 4 strips s/8 apart for the bind-fused second round).  out_r / out_w (2 * nstrips * cs elements each) receive the mini-layer [A strips..., B strips...]:
// element (arr * nstrips + t) * cs + i = leaf[arr * s/2 + t * stride + i0 + i] — exactly the arrays lasso_sumcheck_cubic_eqw2_begin takes with n = nstrips * cs.
// Same arithmetic, same canonical bytes as k_fingerprint_ops.
template <bool RU32>
__global__ void __launch_bounds__(LASSO_BLOCK) k_fingerprint_ops_strips(const fr_t* __restrict__ table, const uint32_t* __restrict__ dim, const void* __restrict__ read_any, size_t s,
                                                                         fr_t gamma, fr_t gamma2, fr_t tau, uint32_t nstrips, size_t stride, size_t i0, size_t cs,
                                                                         fr_t* __restrict__ out_r, fr_t* __restrict__ out_w) {
  const fr29 gs = fr29_unpack_s(gamma), g2s = fr29_unpack_s(gamma2), g2u = fr29_unpack_u(gamma2), tu = fr29_unpack_u(tau), r2s = fr29_r2s();
  const fr29 g2r = fr29_mul(g2s, r2s);
  const fr_t* __restrict__ read = reinterpret_cast<const fr_t*>(read_any); const uint32_t* __restrict__ read32 = reinterpret_cast<const uint32_t*>(read_any);
  const size_t total = 2 * (size_t)nstrips * cs;
  for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < total; j += (size_t)gridDim.x * blockDim.x) {
    const size_t strip = j / cs, i = j - strip * cs, arr = strip / nstrips, t = strip - arr * nstrips;
    const size_t k = arr * (s / 2) + t * stride + i0 + i;
    const uint32_t a = dim[k];
    fr29 h = fr29_add(RU32 ? fr29_mul(fr29_from_u64_int(read32[k]), g2r) : fr29_mul(fr29_unpack_u(read[k]), g2s), fr29_mul(fr29_unpack_u(table[a]), gs));
    h = fr29_canonical(fr29_sub(fr29_add(h, fr29_mul(fr29_from_u64_int(a), r2s)), tu));
    out_r[j] = fr29_pack(h);
    out_w[j] = fr29_store(fr29_add(h, g2u));
  }
}
// slab mode: local index i stands for global address a = i*world + rank; `table` is the whole subtable, `fin` and the outputs are local (m = local length)
__global__ void __launch_bounds__(LASSO_BLOCK) k_fingerprint_mem(const fr_t* __restrict__ table, const fr_t* __restrict__ fin, size_t m, uint32_t world, uint32_t rank, fr_t gamma, fr_t gamma2, fr_t tau,
                                                                  fr_t* __restrict__ out_i, fr_t* __restrict__ out_f) {
  const fr29 gs = fr29_unpack_s(gamma), g2s = fr29_unpack_s(gamma2), tu = fr29_unpack_u(tau), r2s = fr29_r2s();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
    const size_t a = i * world + rank;
    const fr29 h = fr29_canonical(fr29_sub(fr29_add(fr29_mul(fr29_unpack_u(table[a]), gs), fr29_mul(fr29_from_u64_int(a), r2s)), tu));
    out_i[i] = fr29_pack(h);
    out_f[i] = fr29_store(fr29_add(h, fr29_mul(fr29_unpack_u(fin[i]), g2s)));
  }
}

// ------------------------------------------------------------------ small conversions / gathers
__global__ void __launch_bounds__(LASSO_BLOCK) k_from_u32(const uint32_t* __restrict__ src, size_t n, fr_t* __restrict__ dst) {
  const fr29 r2s = fr29_r2s();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = fr29_store(fr29_mul(fr29_from_u64_int(src[i]), r2s));
}
__global__ void __launch_bounds__(LASSO_BLOCK) k_gather(const fr_t* __restrict__ table, const uint32_t* __restrict__ idx, size_t n, fr_t* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = table[idx[i]];
}
__global__ void k_read_heads(PtrTable polys, uint32_t k, fr_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) out[i] = polys.p[i][0];
}
// out[i * count + j] = polys_i[j], j < count: short runs of several arrays for the host (the tops of the product trees)
__global__ void k_read_runs(PtrTable polys, uint32_t k, uint32_t count, fr_t* __restrict__ out) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < k * count) out[e] = polys.p[e / count][e % count];
}
// Montgomery memory form -> the canonical integer (ark-ff into_bigint): x*2^256 * 2^5 / 2^261 = x
__device__ __forceinline__ fr_t fr29_to_integer(const fr29& u) { fr29 k32 = fr29_zero(); k32.v[0] = 32; return fr29_store(fr29_mul(u, k32)); }

// ------------------------------------------------------------------ K11: L*Z mat-vec (dense_mlpoly.rs:184-207)
// grid = (column blocks, row chunks); partials[chunk*R + col] = sum_{j in chunk} L[j] * Z[j*R + col]
__global__ void __launch_bounds__(LASSO_BLOCK) k_matvec_left(const fr_t* __restrict__ Z, const fr_t* __restrict__ Lv, size_t l_size, size_t r_size, size_t rows_per_chunk,
                                                              fr_t* __restrict__ partials) {
  size_t col = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (col >= r_size) return;
  size_t j0 = (size_t)blockIdx.y * rows_per_chunk, j1 = j0 + rows_per_chunk; if (j1 > l_size) j1 = l_size;
  fr29_acc wa = fr29_acc_zero(); uint32_t cnt = 0;   // sum of products: reduce once (fr29_mul_acc)
  for (size_t j = j0; j < j1; j++) {
    fr29_mul_acc(wa, fr29_unpack_u(Z[j * r_size + col]), fr29_unpack_s(Lv[j]));
    if (++cnt == 3) { fr29_acc_carry(wa); cnt = 0; }
  }
  fr29_acc_carry(wa);
  partials[(size_t)blockIdx.y * r_size + col] = fr29_store(fr29_mul(fr29_acc_reduce(wa), fr29_one_s()));
}
__global__ void __launch_bounds__(LASSO_BLOCK) k_matvec_reduce(const fr_t* __restrict__ partials, size_t nchunks, size_t r_size, fr_t* __restrict__ out) {
  size_t col = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (col >= r_size) return;
  fr29 acc = fr29_zero(); uint32_t cnt = 0;
  for (size_t c = 0; c < nchunks; c++) acc_add(acc, fr29_unpack_u(partials[c * r_size + col]), cnt);
  out[col] = fr29_store(fr29_mul(acc, fr29_one_s()));
}

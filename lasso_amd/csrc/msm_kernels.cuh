// Pippenger-style MSM kernels for the Hyrax commitment (gfx950).
//
// Design (MI355X-first, not the reference's serial window loop src/msm/mod.rs:91-164):
//  * the generators are fixed for the lifetime of a gens object, so every window multiple 2^(4w)*G_j is
//    precomputed once (table[w][j], affine "Niels" form, 96 B) — all windows of a scalar then fall into ONE
//    bucket set, with no per-window bucket reduction and no doubling chain at all;
//  * 4-bit unsigned digits = the nibbles of the canonical little-endian scalar; zero digits are skipped, so
//    the reference's small-scalar shortcut (msm/mod.rs:95-106) is automatic: a 16-bit scalar costs <= 4 adds;
//  * one 256-thread workgroup owns one bucket set (15 non-zero digits) for a chunk of columns; its threads are shared out over the
//    digits in proportion to the digits' pair counts (see k_msm_buckets), so skewed scalars do not serialise on the busiest digit;
//  * (digit, table-index) pairs are counting-sorted in LDS in batches of 8192, accumulated in registers with 7-multiplication mixed adds;
//  * reduction: segmented LDS tree per digit, four bit-plane sums, one Horner chain of doublings.
// Results are group elements, so any accumulation order is bit-identical after compression.
#pragma once
#include <hip/hip_runtime.h>
#include "fq.cuh"
#include "fe29.cuh"
#include "poly_kernels.cuh"   // block_reduce_fr

#define MSM_THREADS 256
#define MSM_BATCH 8192   // (bin, index) pairs sorted per pass: 32 KB of the 36 KB LDS buffer the reduction tree reuses
#define MSM_WINDOWS 64   // 4-bit windows over 256-bit scalars

// table[w*n + j] = Niels(2^(4w) * G_j) in 29-bit-limb form (fe29.cuh).  One thread per generator; built once per gens object
// with the 8x32 arithmetic (needs an inversion per entry), then converted.  `aff` = ark Affine {x,y} Montgomery limbs.
__global__ void k_precompute_table(const fq_t* __restrict__ aff, size_t n, niels29* __restrict__ table) {
  size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (j >= n) return;
  fq_t x = fq_from_mont(aff[2 * j]), y = fq_from_mont(aff[2 * j + 1]);
  table[j] = niels_from_affine(x, y);
  ed_point P = ed_from_affine(x, y);
  for (int w = 1; w < MSM_WINDOWS; w++) {
    for (int k = 0; k < 4; k++) P = ed_dbl(P);
    fq_t zi = fq_inv_chain(P.Z);
    table[(size_t)w * n + j] = niels_from_affine(fq_mul(P.X, zi), fq_mul(P.Y, zi));
  }
}

// Montgomery Fr -> low 32 bits of the canonical value; flags[0] = max low word seen, flags[1] |= 1 if any value >= 2^32
__global__ void __launch_bounds__(256) k_fr_to_u32(const fr_t* __restrict__ src, size_t n, uint32_t* __restrict__ dst, uint32_t* __restrict__ flags) {
  uint32_t mx = 0, big = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    fr_t c = fr29_to_integer(fr29_unpack_u(src[i]));
    dst[i] = c.v[0];
    mx = max(mx, c.v[0]);
    big |= c.v[1] | c.v[2] | c.v[3] | c.v[4] | c.v[5] | c.v[6] | c.v[7];
  }
  for (int off = 32; off > 0; off >>= 1) { mx = max(mx, (uint32_t)__shfl_down(mx, off, 64)); big |= (uint32_t)__shfl_down(big, off, 64); }
  if ((threadIdx.x & 63) == 0) { atomicMax(&flags[0], mx); if (big) atomicOr(&flags[1], 1u); }
}
// Montgomery Fr -> canonical 32-byte little-endian integers
__global__ void __launch_bounds__(256) k_fr_to_canonical(const fr_t* __restrict__ src, size_t n, fr_t* __restrict__ dst) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = fr29_to_integer(fr29_unpack_u(src[i]));
}

// grid = (chunks per row K, rows).  scal: canonical little-endian scalars, `bps` bytes each (4 or 32), row r at
// scal + r*row_stride (bytes).  Windows 0..W-1 = nibble w of each scalar.  out[row*K + chunk] = partial sum (extended, fe29 limbs).
//
// Work split inside the workgroup: (digit, base) pairs are counting-sorted by digit in LDS (16 column-slice sub-bins per digit keep the
// LDS atomics apart), then the 256 threads are shared out over the 15 digits IN PROPORTION TO THEIR PAIR COUNTS — real Lasso scalars are
// heavily skewed (timestamp high nibbles, popcount-skewed AND values: a one-thread-per-bin layout leaves the busiest bin 2-3x the
// average) — every thread accumulates a strided share of its digit's pairs in registers (7-multiplication mixed adds, next table entry
// fetched while the current one is added), a segmented LDS tree joins the threads of each digit, and sum_d d*B_d is taken through the four
// bit planes S_b = sum_{d: bit b} B_d (three tree levels on 32 lanes) and one Horner chain 2(2(2 S_3 + S_2) + S_1) + S_0.
__device__ __forceinline__ uint32_t msm_nibble(const uint8_t* s, uint32_t w) { return (reinterpret_cast<const uint32_t*>(s)[w >> 3] >> (4 * (w & 7))) & 15u; }
__global__ void __launch_bounds__(MSM_THREADS) k_msm_buckets(const uint8_t* __restrict__ scal, uint32_t bps, uint32_t W, size_t row_stride, size_t n_cols, size_t cols_per_chunk,
                                                              const niels29* __restrict__ table, size_t table_stride, pt29* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint8_t raw[MSM_THREADS * sizeof(pt29)];  // sorted[] (32 KB) during accumulation, points (36 KB) during the trees
  __shared__ uint32_t counts[MSM_THREADS], start[MSM_THREADS], cursor[MSM_THREADS];
  __shared__ uint32_t toff[17], tree_top;
  uint32_t* sorted = reinterpret_cast<uint32_t*>(raw);
  pt29* pts = reinterpret_cast<pt29*>(raw);
  const fe29 d2 = fe_d2();
  const uint32_t t = threadIdx.x;
  const uint8_t* row = scal + (size_t)blockIdx.y * row_stride;
  const size_t c0 = (size_t)blockIdx.x * cols_per_chunk;
  size_t c1 = c0 + cols_per_chunk; if (c1 > n_cols) c1 = n_cols;
  const uint32_t scalars_per_batch = MSM_BATCH / W;
  pt29 B = pt_identity();
  uint32_t my_d = 0, my_j = 0, my_T = 0;   // this thread's digit, its rank among the digit's threads, and how many threads share the digit
  for (size_t b0 = c0; b0 < c1; b0 += scalars_per_batch) {
    size_t b1 = b0 + scalars_per_batch; if (b1 > c1) b1 = c1;
    const uint32_t items = (uint32_t)(b1 - b0) * W;   // one item = one (scalar, window)
    counts[t] = 0;
    __syncthreads();
    for (uint32_t it = t; it < items; it += MSM_THREADS) {
      const uint32_t ci = it / W, w = it - ci * W; const size_t c = b0 + ci;
      const uint32_t d = msm_nibble(row + c * bps, w);
      if (d) atomicAdd(&counts[(d << 4) | ((uint32_t)c & 15u)], 1u);
    }
    __syncthreads();
    start[t] = counts[t];
    __syncthreads();
    for (uint32_t off = 1; off < MSM_THREADS; off <<= 1) { uint32_t v = t >= off ? start[t - off] : 0; __syncthreads(); start[t] += v; __syncthreads(); }   // inclusive scan
    cursor[t] = start[t] - counts[t];
    __syncthreads();
    for (uint32_t it = t; it < items; it += MSM_THREADS) {
      const uint32_t ci = it / W, w = it - ci * W; const size_t c = b0 + ci;
      const uint32_t d = msm_nibble(row + c * bps, w);
      if (d) sorted[atomicAdd(&cursor[(d << 4) | ((uint32_t)c & 15u)], 1u)] = (uint32_t)(w * table_stride + c);
    }
    if (b0 == c0) {
      // share the threads out over digits 1..15 from the first batch's histogram (later batches of the same row have the same statistics);
      // every digit keeps at least one thread, so no pair of a later batch can be orphaned
      if (t == 0) {
        const uint32_t total = start[MSM_THREADS - 1];
        uint32_t acc = 0, top = 1; toff[0] = 0; toff[1] = 0;
        for (uint32_t d = 1; d < 16; d++) {
          const uint32_t cnt = start[d * 16 + 15] - (start[d * 16] - counts[d * 16]);
          const uint32_t T = 1 + (total ? (uint32_t)(((uint64_t)cnt * (MSM_THREADS - 15)) / total) : 0);
          acc += T; toff[d + 1] = acc; if (T > top) top = T;
        }
        uint32_t p2 = 1; while (p2 < top) p2 <<= 1;
        tree_top = p2 >> 1;
      }
      __syncthreads();
      for (uint32_t d = 1; d < 16; d++) if (t >= toff[d] && t < toff[d + 1]) { my_d = d; my_j = t - toff[d]; my_T = toff[d + 1] - toff[d]; }
    }
    __syncthreads();
    if (my_T) {
      const uint32_t lo = start[my_d * 16] - counts[my_d * 16], hi = start[my_d * 16 + 15];
      uint32_t pos = lo + my_j;
      if (pos < hi) {
        niels29 cur = table[sorted[pos]];
        for (pos += my_T; pos < hi; pos += my_T) { const niels29 nxt = table[sorted[pos]]; B = pt_madd(B, cur); cur = nxt; }
        B = pt_madd(B, cur);
      }
    }
    __syncthreads();
  }
  // segmented tree: the threads of one digit are contiguous; pts[toff[d]] ends up holding B_d
  pts[t] = B;
  __syncthreads();
  for (uint32_t s = tree_top; s > 0; s >>= 1) {
    pt29 sum;
    const bool act = my_j < s && my_j + s < my_T;
    if (act) sum = pt_add(pts[t], pts[t + s], d2);
    __syncthreads();
    if (act) pts[t] = sum;
    __syncthreads();
  }
  // bit planes: lane (b, i), i < 8, takes the i-th digit that has bit b set
  pt29 P;
  const uint32_t b = t >> 3, i = t & 7;
  if (t < 32) { const uint32_t d = ((i >> b) << (b + 1)) | (1u << b) | (i & ((1u << b) - 1)); P = pts[toff[d]]; }
  __syncthreads();
  if (t < 32) pts[t] = P;
  __syncthreads();
  for (uint32_t s = 4; s > 0; s >>= 1) {
    if (t < 32 && i < s) P = pt_add(P, pts[t + s], d2);
    __syncthreads();
    if (t < 32 && i < s) pts[t] = P;
    __syncthreads();
  }
  if (t == 0) {
    pt29 acc = pts[24];                                   // S_3
    acc = pt_add(pt_dbl(acc), pts[16], d2);               // 2 S_3 + S_2
    acc = pt_add(pt_dbl(acc), pts[8], d2);
    acc = pt_add(pt_dbl(acc), pts[0], d2);
    out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
  }
}

// out[row] = sum_k partial[row*K + k], converted to ark's Montgomery limbs (out_mont) or, when out_compressed is given, to the 32-byte wire form.  One workgroup per row; thread t first adds partials
// t, t+256, ... serially, then an LDS tree.  `out_mont` may be host-mapped memory: when `flag` is set, the row that finishes last
// raises the host's sequence flag (same hand-off as last_block_reduce in poly_kernels.cuh).
__global__ void __launch_bounds__(MSM_THREADS) k_points_sum(const pt29* __restrict__ partial, uint32_t K, ed_point* __restrict__ out_mont, uint32_t* __restrict__ out_compressed, uint32_t* counters,
                                                             uint32_t* flag, uint32_t seq) {
  __shared__ pt29 pts[MSM_THREADS];
  const fe29 d2 = fe_d2();
  const uint32_t t = threadIdx.x;
  pt29 acc = t < K ? partial[(size_t)blockIdx.x * K + t] : pt_identity();
  for (uint32_t k = t + MSM_THREADS; k < K; k += MSM_THREADS) acc = pt_add(acc, partial[(size_t)blockIdx.x * K + k], d2);
  pts[t] = acc;
  __syncthreads();
  const uint32_t live = K < MSM_THREADS ? K : MSM_THREADS;
  for (uint32_t s = MSM_THREADS / 2; s > 0; s >>= 1) { if (t < s && t + s < live) pts[t] = pt_add(pts[t], pts[t + s], d2); __syncthreads(); }
  if (t == 0 && out_compressed) reinterpret_cast<pt29*>(out_compressed)[blockIdx.x] = pts[0];   // compressed mode: hand the row sum to k_points_compress (one lane per row)
  if (t == 0 && !out_compressed) {
    ed_point p = pt_to_ed(pts[0]), o; o.X = fq_to_mont(p.X); o.Y = fq_to_mont(p.Y); o.T = fq_to_mont(p.T); o.Z = fq_to_mont(p.Z); out_mont[blockIdx.x] = o;
    if (flag) {
      __threadfence_system();
      uint32_t t2 = __hip_atomic_fetch_add(counters, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (t2 == gridDim.x - 1) { *counters = 0; __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
  }
}

// wire form of many points at once: one lane per point (normalize_batch + serialize_compressed on the device; the inversion chain is 265 products)
__global__ void __launch_bounds__(256) k_points_compress(const pt29* __restrict__ pts, size_t n, uint32_t* __restrict__ out32) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) pt_compress(pts[i], out32 + 8 * i);
}

// ------------------------------------------------------------------ Hyrax opening tail (bullet.rs:40-154), vectors resident on the device
// partials[bx*2 + {0,1}] = partial <a_L, b_R>, <a_R, b_L>
__global__ void __launch_bounds__(256) k_inner_lr(const fr_t* __restrict__ a, const fr_t* __restrict__ b, size_t half, fr_t* __restrict__ partials);
// canonical scalars of the two MSMs of one round over the ORIGINAL generators: rows SL, SR of n+2 entries each
__global__ void __launch_bounds__(256) k_bullet_expand(const fr_t* __restrict__ a, size_t nk, const fr_t* __restrict__ w, size_t n, fr_t cL, fr_t bL, fr_t cR, fr_t bR, fr_t* __restrict__ SL,
                                                        fr_t* __restrict__ SR) {
  const size_t half = nk / 2;
  for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
    const size_t blk = j / nk, pos = j % nk;
    const fr29 ws = fr29_unpack_s(w[blk]);
    if (pos >= half) { SL[j] = fr29_to_integer(fr29_mul(fr29_unpack_u(a[pos - half]), ws)); SR[j] = fr_zero(); }
    else { SL[j] = fr_zero(); SR[j] = fr29_to_integer(fr29_mul(fr29_unpack_u(a[pos + half]), ws)); }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    SL[n] = fr29_to_integer(fr29_unpack_u(cL)); SL[n + 1] = fr29_to_integer(fr29_unpack_u(bL)); SR[n] = fr29_to_integer(fr29_unpack_u(cR)); SR[n + 1] = fr29_to_integer(fr29_unpack_u(bR));
  }
}
__global__ void __launch_bounds__(256) k_bullet_fold(fr_t* __restrict__ a, fr_t* __restrict__ b, size_t half, const fr_t* __restrict__ w, size_t nw, fr_t* __restrict__ w_out, fr_t u, fr_t u_inv) {
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  const fr29 us = fr29_unpack_s(u), uis = fr29_unpack_s(u_inv);
  for (size_t i = tid; i < half; i += stride) {
    const fr29 al = fr29_unpack_u(a[i]), ar = fr29_unpack_u(a[i + half]), bl = fr29_unpack_u(b[i]), br = fr29_unpack_u(b[i + half]);
    a[i] = fr29_store(fr29_add(fr29_mul(al, us), fr29_mul(ar, uis)));
    b[i] = fr29_store(fr29_add(fr29_mul(bl, uis), fr29_mul(br, us)));
  }
  for (size_t k = tid; k < nw; k += stride) { const fr29 x = fr29_unpack_u(w[k]); w_out[2 * k] = fr29_store(fr29_mul(x, uis)); w_out[2 * k + 1] = fr29_store(fr29_mul(x, us)); }
}

// ------------------------------------------------------------------ one bullet round in one pass (bullet.rs:66-132)
// FOLD: apply the previous challenge first (a' = a_L*u + u_inv*a_R, b' = b_L*u_inv + u*b_R, w'[2k] = w[k]*u_inv, w'[2k+1] = w[k]*u), reading the
// ping-pong inputs of length 2*nk and writing the outputs of length nk; then, on the state of length nk: c_L = <a_L, b_R>, c_R = <a_R, b_L>
// (last-block reduction) and the canonical scalar rows SL, SR (n + 2 entries each) of the two MSMs over the ORIGINAL generators.
// One thread per (i < nk/2, blk < n/nk): n/2 threads whatever the round, so late rounds are as parallel as early ones.
template <bool FOLD>
__global__ void __launch_bounds__(256) k_bullet_step(const fr_t* __restrict__ a_in, const fr_t* __restrict__ b_in, const fr_t* __restrict__ w_in, fr_t* __restrict__ a_out, fr_t* __restrict__ b_out,
                                                      fr_t* __restrict__ w_out, size_t nk, size_t n, fr_t u, fr_t u_inv, fr_t blind_l, fr_t blind_r, fr_t* __restrict__ SL, fr_t* __restrict__ SR,
                                                      fr_t* __restrict__ partials, uint32_t* counters) {
  __shared__ RedScratch S;
  __shared__ uint32_t is_last;
  const size_t half = nk / 2, total = n / 2;
  const fr29 us = fr29_unpack_s(u), uis = fr29_unpack_s(u_inv);
  fr29 acc[3] = {fr29_zero(), fr29_zero(), fr29_zero()}; uint32_t cnt = 0;   // acc[0] = c_L, acc[1] = c_R partial sums
  for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
    const size_t i = g % half, blk = g / half;
    fr29 a0, a1, wv;   // canonical u-form
    if (FOLD) {
      a0 = fr29_canonical(fr29_add(fr29_mul(fr29_unpack_u(a_in[i]), us), fr29_mul(fr29_unpack_u(a_in[i + nk]), uis)));
      a1 = fr29_canonical(fr29_add(fr29_mul(fr29_unpack_u(a_in[i + half]), us), fr29_mul(fr29_unpack_u(a_in[i + half + nk]), uis)));
      wv = fr29_canonical(fr29_mul(fr29_unpack_u(w_in[blk >> 1]), (blk & 1) ? us : uis));
      if (i == 0) w_out[blk] = fr29_pack(wv);
    } else { a0 = fr29_unpack_u(a_in[i]); a1 = fr29_unpack_u(a_in[i + half]); wv = fr29_unpack_u(w_in[blk]); }
    if (blk == 0) {
      fr29 b0, b1;
      if (FOLD) {
        b0 = fr29_canonical(fr29_add(fr29_mul(fr29_unpack_u(b_in[i]), uis), fr29_mul(fr29_unpack_u(b_in[i + nk]), us)));
        b1 = fr29_canonical(fr29_add(fr29_mul(fr29_unpack_u(b_in[i + half]), uis), fr29_mul(fr29_unpack_u(b_in[i + half + nk]), us)));
        a_out[i] = fr29_pack(a0); a_out[i + half] = fr29_pack(a1); b_out[i] = fr29_pack(b0); b_out[i + half] = fr29_pack(b1);
      } else { b0 = fr29_unpack_u(b_in[i]); b1 = fr29_unpack_u(b_in[i + half]); }
      // u * u products are 2^5 short: corrected with K5 when the block partial is written
      acc[0] = fr29_weak(fr29_add(acc[0], fr29_mul(a0, b1)));
      acc[1] = fr29_weak(fr29_add(acc[1], fr29_mul(a1, b0)));
      if ((++cnt & 127u) == 0) { acc[0] = fr29_mul(acc[0], fr29_one_s()); acc[1] = fr29_mul(acc[1], fr29_one_s()); }
    }
    // MSM scalars are canonical integers: mul(u, u) = wv*a*2^251, one more Montgomery step with the integer 2^10 gives wv*a itself
    const size_t base = blk * nk + i;
    SL[base] = fr_zero(); SL[base + half] = fr29_store(fr29_mul(fr29_mul(wv, a0), fr29_int_from_uu()));
    SR[base] = fr29_store(fr29_mul(fr29_mul(wv, a1), fr29_int_from_uu())); SR[base + half] = fr_zero();
  }
  store_block_partials<3>(acc, 2, partials + 2 * (size_t)blockIdx.x, fr29_k5(), S);
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t ticket = __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t last = (ticket == gridDim.x - 1) ? 1u : 0u;
    if (last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); *counters = 0; }
    is_last = last;
  }
  __syncthreads();
  if (!is_last) return;
  fr29 tot[3] = {fr29_zero(), fr29_zero(), fr29_zero()};
  for (uint32_t x = threadIdx.x; x < gridDim.x; x += blockDim.x) { tot[0] = fr29_weak(fr29_add(tot[0], fr29_unpack_u(partials[2 * (size_t)x]))); tot[1] = fr29_weak(fr29_add(tot[1], fr29_unpack_u(partials[2 * (size_t)x + 1]))); }
  block_columns<3>(tot, S);
  if (threadIdx.x < 2) {
    int64_t c[9];
    for (int k = 0; k < 9; k++) c[k] = S.cols[threadIdx.x * 9 + k];
    fr29 k32 = fr29_zero(); k32.v[0] = 32;
    const fr_t v = fr29_store(fr29_mul(fr29_from_columns(c), k32));   // u-form sum -> canonical integer
    if (threadIdx.x == 0) { SL[n] = v; SL[n + 1] = fr29_to_integer(fr29_unpack_u(blind_l)); } else { SR[n] = v; SR[n + 1] = fr29_to_integer(fr29_unpack_u(blind_r)); }
  }
}

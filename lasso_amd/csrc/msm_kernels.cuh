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
// Layout of the digit- / byte-multiple tables: entry (window w, multiple m1 = m - 1, generator j) of a table with NW windows of NM multiples over tn generators.
//   window-major (rounds 2-5):  ((w NM + m1) tn + j)   — a wave's 64 lanes (consecutive items = consecutive windows of a column, or consecutive columns with random bytes) touch 64 pages
//   column-major (MSM_TABLE_COLMAJOR): ((j NW + w) NM + m1) — everything a column can ever ask for is one contiguous block (28 KB per byte window), a wave's lanes stay inside 1-2 MB
#ifdef MSM_TABLE_COLMAJOR
#define MSM_IDX(w, m1, j, tn, NW, NM) ((((size_t)(j)) * (NW) + (w)) * (NM) + (m1))
#define MSM_COL_BASE(tab, j0, NW, NM) ((tab) + (size_t)(j0) * (NW) * (NM))
#else
#define MSM_IDX(w, m1, j, tn, NW, NM) ((((size_t)(w)) * (NM) + (m1)) * (tn) + (j))
#define MSM_COL_BASE(tab, j0, NW, NM) ((tab) + (j0))
#endif

// table[w*n + j] = Niels(2^(4w) * G_j) in 29-bit-limb form (fe29.cuh).  One thread per generator; built once per gens object
// with the 8x32 arithmetic (needs an inversion per entry), then converted.  `aff` = ark Affine {x,y} Montgomery limbs.
__global__ void k_precompute_table(const fq_t* __restrict__ aff, size_t n, niels29* __restrict__ table) {
  size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (j >= n) return;
#ifdef LASSO_BN254
  // BN254 build: the window multiples in the 29-bit form directly (four complete doublings, one inversion per entry)
  niels29 e = niels_from_affine(aff[2 * j], aff[2 * j + 1]);
  table[j] = e;
  for (int w = 1; w < MSM_WINDOWS; w++) {
    pt29 P = pt_madd(pt_identity(), e);
    for (int k = 0; k < 4; k++) P = pt_dbl(P);
    const fe29 zi = fe_inv_chain(P.Z);
    e = niels_from_xy29(fe_mul(P.X, zi), fe_mul(P.Y, zi), fe_d2());
    table[(size_t)w * n + j] = e;
  }
#else
  fq_t x = fq_from_mont(aff[2 * j]), y = fq_from_mont(aff[2 * j + 1]);
  table[j] = niels_from_affine(x, y);
  ed_point P = ed_from_affine(x, y);
  for (int w = 1; w < MSM_WINDOWS; w++) {
    for (int k = 0; k < 4; k++) P = ed_dbl(P);
    fq_t zi = fq_inv_chain(P.Z);
    table[(size_t)w * n + j] = niels_from_affine(fq_mul(P.X, zi), fq_mul(P.Y, zi));
  }
#endif
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
// phase timing for tools/msm_phase_bench.hip only (compiled out of the library): workgroup (0,0) stamps the 100 MHz wall clock at phase boundaries
// MSM_PHASE_LOG (tools/bullet_phase_bench.hip): EVERY workgroup stamps, 32 slots each, so that the launch's critical path — which runs through whichever workgroup arrives last —
// can be read off afterwards; slots 16.. / 24.. are the levels of the workgroup tree / the cross-workgroup tree (MSM_TREE_STAMP).
#if defined(MSM_PHASE_LOG)
#define MSM_PHASE_LOG_WGS 1024
__device__ uint64_t msm_phase_log[MSM_PHASE_LOG_WGS * 32];
#define MSM_STAMP(k) do { if (threadIdx.x == 0) msm_phase_log[((blockIdx.y * gridDim.x + blockIdx.x) & (MSM_PHASE_LOG_WGS - 1)) * 32 + (k)] = wall_clock64(); } while (0)
#define MSM_TREE_STAMP(base, lvl) do { if ((base) >= 0 && (lvl) < 8) MSM_STAMP((base) + (lvl)); } while (0)
#elif defined(MSM_PHASE_CLOCK)
__device__ uint64_t msm_phase_clock[32];
#define MSM_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) msm_phase_clock[k] = wall_clock64(); } while (0)
#define MSM_TREE_STAMP(base, lvl) do { } while (0)
#else
#define MSM_STAMP(k) do { } while (0)
#define MSM_TREE_STAMP(base, lvl) do { } while (0)
#endif
// exact count of the mixed additions a launch executes (bench.py roofline_msm): only when the host passes a counter (the untimed fully-profiled step)
__device__ __forceinline__ void msm_count_adds(uint32_t* digit_count, uint32_t mine) {
  if (!digit_count) return;
  for (int off = 32; off > 0; off >>= 1) mine += (uint32_t)__shfl_down(mine, off, 64);
  // one atomic per wave, spread over the 64 slots of the launch's counter: a thousand waves adding to ONE address cost the opening MSM 40 us
  if ((threadIdx.x & 63u) == 0 && mine) atomicAdd(digit_count + ((blockIdx.x * 4u + (threadIdx.x >> 6) + blockIdx.y * 17u) & 63u), mine);
}
__device__ __forceinline__ uint32_t msm_nibble(const uint8_t* s, uint32_t w) { return (reinterpret_cast<const uint32_t*>(s)[w >> 3] >> (4 * (w & 7))) & 15u; }
__global__ void __launch_bounds__(MSM_THREADS) k_msm_buckets(const uint8_t* __restrict__ scal, uint32_t bps, uint32_t W, size_t row_stride, size_t n_cols, size_t cols_per_chunk,
                                                              const niels29* __restrict__ table, size_t table_stride, pt29* __restrict__ out, uint32_t* digit_count) {
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
  MSM_STAMP(0);
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
    if (digit_count && t == 0 && start[MSM_THREADS - 1]) atomicAdd(digit_count + ((blockIdx.x + blockIdx.y * 17u) & 63u), start[MSM_THREADS - 1]);   // non-zero digits of this batch = additions issued for it
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
    MSM_STAMP(1);
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
  MSM_STAMP(2);
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
  MSM_STAMP(3);
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
  MSM_STAMP(4);
  if (t == 0) {
    pt29 acc = pts[24];                                   // S_3
    acc = pt_add(pt_dbl(acc), pts[16], d2);               // 2 S_3 + S_2
    acc = pt_add(pt_dbl(acc), pts[8], d2);
    acc = pt_add(pt_dbl(acc), pts[0], d2);
    out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
  }
  MSM_STAMP(5);
}

// ------------------------------------------------------------------ latency-shaped MSM for a few rows of full-width scalars (the opening tail)
// A bullet round (bullet.rs:98-121) is two MSMs of n/2 + 2 full-width scalars and the host waits for the result before it can draw the next
// challenge, so what matters is the LENGTH OF THE DEPENDENT CHAIN, not the operation count.  With buckets that chain is: digit sort,
// accumulate, segmented tree, three bit-plane levels and a six-operation Horner tail run by ONE lane (36 us on MI355X, tools/msm_phase_bench.hip),
// then the cross-chunk tree.  Here the generators' table holds every signed digit multiple m * 16^w * G_j, m = 1..8 (affine Niels form,
// 8 x 64 entries per generator), so the MSM is a plain sum of one table entry per non-zero digit: no sort, no buckets, no bit planes, no
// Horner — per thread a handful of mixed adds, then one binary tree (in the workgroup, then across workgroups in the last one to arrive).
//   signed digits: s + 0x88..8 has nibbles e_w, and s = sum (e_w - 8) 16^w with e_w - 8 in [-8, 7]; s < 2^253 so the top nibble cannot overflow.
//   negation of a Niels entry is free (swap y+x and y-x, negate 2dxy).
#define MSM_MULTS 8
// mult[(w*8 + m-1)*n + j] = m * 16^w * G_j from table[w*n + j] = 16^w * G_j.  One thread per (w, j): 4 doublings + 3 mixed adds, then the
// seven results share ONE inversion (Montgomery's trick) on their way to affine Niels form.
__global__ void __launch_bounds__(64) k_precompute_multiples(const niels29* __restrict__ table, size_t n, niels29* __restrict__ mult) {
  const size_t id = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (id >= n * MSM_WINDOWS) return;
  const size_t w = id / n, j = id - w * n;
  const niels29 b = table[id];
  const fe29 d2 = fe_d2();
  pt29 m[7];   // 2P .. 8P
  const pt29 P1 = pt_madd(pt_identity(), b);
  m[0] = pt_dbl(P1); m[1] = pt_madd(m[0], b); m[2] = pt_dbl(m[0]); m[3] = pt_madd(m[2], b); m[4] = pt_dbl(m[1]); m[5] = pt_madd(m[4], b); m[6] = pt_dbl(m[2]);
  fe29 pre[7]; pre[0] = m[0].Z;
  for (int k = 1; k < 7; k++) pre[k] = fe_mul(pre[k - 1], m[k].Z);
  fe29 inv = fe_inv_chain(pre[6]);
  mult[MSM_IDX(w, 0, j, n, MSM_WINDOWS, MSM_MULTS)] = b;
  for (int k = 6; k >= 0; k--) {
    const fe29 zi = k ? fe_mul(inv, pre[k - 1]) : inv;
    if (k) inv = fe_mul(inv, m[k].Z);
    const fe29 x = fe_mul(m[k].X, zi), y = fe_mul(m[k].Y, zi);
#ifdef LASSO_BN254
    const niels29 e = niels_from_xy29(x, y, d2);
#else
    niels29 e; e.ypx = fe_weak(fe_add(y, x)); e.ymx = fe_weak(fe_sub(y, x)); e.t2d = fe_mul(fe_mul(x, y), d2); e.pad = 0;
#endif
    mult[MSM_IDX(w, k + 1, j, n, MSM_WINDOWS, MSM_MULTS)] = e;
  }
}
// Logical column j of row `row` -> generator index.  nk = 0: identity.  Bullet rows (nk > 0) are stored compactly: the n/2 non-zero scalars
// of L sit on the RIGHT halves of the nk-blocks of the generator vector, those of R on the LEFT halves (bullet.rs:98-121 with the folds kept
// as weights on the original generators); the two trailing columns (Q and the blinding base) follow the n main generators.
struct MsmColMap { uint32_t nk, half, n_main, phys_main; };
__device__ __forceinline__ uint32_t msm_phys_col(const MsmColMap& m, uint32_t row, uint32_t j) {
  if (!m.nk) return j;
  if (j >= m.n_main) return m.phys_main + (j - m.n_main);
  const uint32_t blk = j / m.half, i = j - blk * m.half;
  return blk * m.nk + i + (row == 0 ? m.half : 0u);
}
// Binary tree over points in LDS with FOUR lanes per addition.  A lone lane runs the nine field products of a unified add back to back
// (3.1 us per tree level on MI355X: one wave per SIMD issues a product's ~150 instructions at the multiply-add rate, and a tree level has
// nothing else to overlap).  Here lane role c of a quad computes ONE product per stage — stage 1: A, B, C, D of add-2008-hwcd-3 (role 2's
// C = (T1*2d)*T2 takes two, so every role issues two: the instruction stream stays uniform), stage 2: X3, Y3, T3, Z3 — through an LDS
// exchange buffer, so a level costs three product times instead of nine.  Role selection is by data (sign / coordinate index), never by branch.
// pts[0..live) -> pts[0].  st: exchange buffer for 64 additions.  All MSM_THREADS threads must call.
#ifdef LASSO_BN254
#define MSM_ST_ROWS (MSM_THREADS / 2)   // two exchange buffers (products, linear forms)
#else
#define MSM_ST_ROWS (MSM_THREADS / 4)
#endif
#ifdef LASSO_BN254
// BN254 build: the complete projective addition's twelve products form two dependent layers of six (bn254_fe29.cuh pt_coop_*), so SIX lanes share
// one addition: 42 additions per pass of the workgroup, a tree level costs two product times (plus the linear step) instead of twelve.  The exchange
// buffer is the Edwards build's (64 x 4 values >= 42 x 6), used twice per pass.  Same interface.
#ifndef MSM_COOP_PLAIN_FROM
#define MSM_COOP_PLAIN_FROM 128u   // tree levels with at least this many additions run one addition per lane (-DMSM_COOP_PLAIN_FROM=1024: the six-lane form at every level, as in round 2)
#endif
__device__ __forceinline__ void tree_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#ifdef MSM_COOP6_V1
// round 3's pass: 42 sextets over the workgroup, every lane forms all six linear combinations, four workgroup barriers (kept for A/B: -DMSM_COOP6_V1)
__device__ __forceinline__ void msm_coop_tree(pt29* pts, fe29 (*st4)[4], uint32_t live, const fe29& d2_unused, int stamp_base = -1) {
  fe29* st = &st4[0][0];
  const uint32_t t = threadIdx.x, g = t / 6u, c = t - g * 6u;
  constexpr uint32_t GROUPS = MSM_THREADS / 6;   // 42
  const bool lane_ok = g < GROUPS;
  uint32_t p2 = 1; while (p2 < live) p2 <<= 1;
  for (uint32_t s = p2 >> 1; s > 0; s >>= 1) {
    if (s >= MSM_COOP_PLAIN_FROM) {
      if (t < s && t + s < live) pts[t] = pt_add(pts[t], pts[t + s], d2_unused);
      __syncthreads();
      continue;
    }
    for (uint32_t i0 = 0; i0 < s; i0 += GROUPS) {
      const uint32_t i = i0 + g;
      const bool act = lane_ok && i < s && i + s < live;
      if (act) st[g * 6 + c] = pt_coop_layer1(pts[i], pts[i + s], c);
      __syncthreads();
      fe29 m[6];
      if (act) {
#pragma unroll
        for (int k = 0; k < 6; k++) m[k] = st[g * 6 + k];
      }
      __syncthreads();
      if (act) st[g * 6 + c] = pt_coop_layer2(m, c);
      __syncthreads();
      if (act && c < 3) {
        const fe29 v = pt_coop_out(st[g * 6 + 2 * c], st[g * 6 + 2 * c + 1], c);
        reinterpret_cast<fe29*>(&pts[i])[c == 2 ? 3 : c] = v;   // pt29 = {X, Y, T, Z}
      }
      __syncthreads();
    }
  }
}
#else
// Round 4's pass.  A sextet is six adjacent lanes of ONE wave (ten sextets per wave, lanes 60..63 idle: 40 additions per pass), so the exchanges inside a pass need the wave's
// own LDS operations kept in order, not a workgroup barrier; the linear step between the two product layers is a lane step of its own (pt_coop_form: one combination per lane
// instead of all six in every lane) through a second exchange buffer.  Measured (tools/msm_phase_bench.hip, BN254 build): a pass 4.8 -> 2.5 us, DESIGN 6.11.  Workgroup barriers
// remain after the levels whose results the next level reads from another wave (more than 10 additions) and after the plain levels.
__device__ __forceinline__ void msm_coop_tree(pt29* pts, fe29 (*st4)[4], uint32_t live, const fe29& d2_unused, int stamp_base = -1) {
  fe29* st = &st4[0][0]; fe29* sf = st + MSM_THREADS;   // the exchange buffer is two buffers of MSM_THREADS values in this build (MSM_ST_ROWS)
  const uint32_t t = threadIdx.x, wv = t >> 6, ln = t & 63u, q = ln / 6u, c = ln - q * 6u, g = wv * 10u + q;
  constexpr uint32_t GROUPS = (MSM_THREADS / 64) * 10;   // 40
  const bool lane_ok = q < 10u;
  uint32_t p2 = 1; while (p2 < live) p2 <<= 1;
  for (uint32_t s = p2 >> 1; s > 0; s >>= 1) {
    if (s >= MSM_COOP_PLAIN_FROM) {
      // the widest levels: one whole addition per lane (128 lanes running the 12 products of an addition on their own take ~7 us; four passes of sextets more).  Same group
      // elements, other projective representatives than the sextets' only in the order of the field operations: the wire bytes do not change
      if (t < s && t + s < live) pts[t] = pt_add(pts[t], pts[t + s], d2_unused);
      __syncthreads();
      MSM_TREE_STAMP(stamp_base, 31 - __clz(s));
      continue;
    }
    for (uint32_t i0 = 0; i0 < s; i0 += GROUPS) {
      const uint32_t i = i0 + g;
      const bool act = lane_ok && i < s && i + s < live;
      if (act) st[g * 6 + c] = pt_coop_layer1(pts[i], pts[i + s], c);
      tree_wave_sync();
      if (act) sf[g * 6 + c] = pt_coop_form(&st[g * 6], c);
      tree_wave_sync();
      if (act) st[g * 6 + c] = pt_coop_prod2(&sf[g * 6], c);
      tree_wave_sync();
      if (act && c < 3) {
        const fe29 v = pt_coop_out(st[g * 6 + 2 * c], st[g * 6 + 2 * c + 1], c);
        reinterpret_cast<fe29*>(&pts[i])[c == 2 ? 3 : c] = v;   // pt29 = {X, Y, T, Z}
      }
      if (s > 10u || i0 + GROUPS < s) __syncthreads(); else tree_wave_sync();   // uniform over the workgroup
    }
    MSM_TREE_STAMP(stamp_base, 31 - __clz(s));
  }
}
#endif  // MSM_COOP6_V1
#else
// Barriers: a quad is four adjacent lanes of ONE wave, so the exchange between its two stages only needs the wave's own LDS operations kept in order (tree_wave_sync), not a
// workgroup barrier; and from the level of 16 additions down every active quad AND every point it reads (written one level up by quads 0..31 -> for s <= 16 by quads 0..15)
// belongs to wave 0, so those levels need no workgroup barrier either.  What stays: one workgroup barrier after each level of more than 16 additions.  LASSO_TREE_BARRIERS=1
// at compile time restores the two workgroup barriers per pass (A/B).
__device__ __forceinline__ void tree_wave_sync() {
#ifdef LASSO_TREE_BARRIERS
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
__device__ __forceinline__ void msm_coop_tree(pt29* pts, fe29 (*st)[4], uint32_t live, const fe29& d2, int stamp_base = -1) {
  (void)d2; (void)stamp_base;   // the curve constant enters as two small multipliers (fe29.cuh pt_coop4_stage1)
  const uint32_t t = threadIdx.x, c = t & 3u, g = t >> 2;
  uint32_t p2 = 1; while (p2 < live) p2 <<= 1;
  for (uint32_t s = p2 >> 1; s > 0; s >>= 1) {
    for (uint32_t i0 = 0; i0 < s; i0 += MSM_THREADS / 4) {
      const uint32_t i = i0 + g;
      const bool act = i < s && i + s < live;
      if (act) st[g][c] = pt_coop4_stage1(pts[i], pts[i + s], c);
      tree_wave_sync();
      if (act) reinterpret_cast<fe29*>(&pts[i])[c] = pt_coop4_stage2p(&st[g][0], c);   // pt29 = {X, Y, T, Z}: role c owns coordinate c
#ifdef LASSO_TREE_BARRIERS
      __syncthreads();
#else
      if (s > 16) __syncthreads(); else tree_wave_sync();   // s is uniform over the workgroup: every thread takes the same path
#endif
    }
    MSM_TREE_STAMP(stamp_base, 31 - __clz(s));   // slot = log2 of the level's addition count
  }
}
#endif  // LASSO_BN254
// ------------------------------------------------------------------ many rows of FULL-WIDTH scalars: 12-bit signed windows, 2048 buckets per row (round 6)
// k_msm_buckets spends one mixed addition per non-zero NIBBLE: 60 per full-width scalar.  The nibble-window table it reads already holds 2^(4w) G_j for every w, so a
// 12-bit window w12 finds 2^(12 w12) G_j at nibble window 3 w12 — no new table — and a scalar costs 22 additions instead of 60 if the row's (scalar, window) pairs
// are first sorted by the magnitude of their signed digit: sum_j s_j G_j = sum_{d=1..2048} d * B_d, B_d = sum of +-(table entries) whose digit is +-d
// (msm/mod.rs:91-164's bucket method with the window sums folded into ONE bucket set per row, because the table carries the window's power of two).  Three launches:
//   k_msm_pip_sort        one workgroup per row: signed digits (carry into the next window), histogram and offsets of the 2048 buckets in LDS, the row's pairs scattered
//                         into bucket order (global scratch), and the buckets RANKED BY SIZE (bitonic sort of 2048 keys in LDS);
//   k_msm_pip_accumulate  eight workgroups per row: in workgroup k lane t sums the bucket of rank 256 k + t — the lanes of a wave run buckets of (nearly) equal
//                         length, so the lockstep loop wastes no lane (by bucket NUMBER the lengths scatter ~ +-25 % around 84 at the Spark shape: a fifth of the
//                         issue slots); next pair and next table entry in flight during the current addition;
//   k_msm_pip_reduce      one workgroup per row: thread t owns buckets 8t .. 8t+7: running sums give T_t = sum_k (k+1) B_{8t+k} and S_t = sum_k B_{8t+k};
//                         the row is sum_t T_t + 8 sum_t t S_t, the second sum through a suffix scan of the S_t in LDS (sum_t t S_t = sum_{j>=1} sum_{t>=j} S_t).
// Same group element per row as k_msm_buckets (another projective representative; the order of additions inside a bucket depends on the scatter's atomics: wire bytes do
// not change).  Additions per 8192-column row: 172 K + 4 K (the values above bit 252) + 8 K (bucket sums) instead of 491 K.
#define MSM_PIP_WINDOWS 21            // bucket windows: bits 0 .. 251; what is left above them (bits 252 .. 255 plus the last carry: 0 .. 4 for a canonical scalar) is NOT a bucket digit:
                                      // half the scalars of a row have 1 there (the carry), and a bucket of 4096 pairs in a row of 88-pair buckets is one lane working alone for 35 ms (measured)
#define MSM_PIP_BUCKETS 2048
#define MSM_PIP_PER_THREAD (MSM_PIP_BUCKETS / MSM_THREADS)
__device__ __forceinline__ uint32_t msm_pip_bits(const uint32_t* s, uint32_t w) {   // bits [12 w, 12 w + 12) of a 256-bit little-endian integer held in eight words
  const uint32_t bit = 12u * w, word = bit >> 5, sh = bit & 31u;
  uint64_t v = s[word]; if (word < 7u) v |= (uint64_t)s[word + 1] << 32;
  return (uint32_t)(v >> sh) & 4095u;
}
// signed digit of window w given the carry of the windows below: returns the magnitude (0 .. 2048), sets neg and the carry out
__device__ __forceinline__ uint32_t msm_pip_digit(const uint32_t* s, uint32_t w, uint32_t& carry, bool& neg) {
  const uint32_t raw = msm_pip_bits(s, w) + carry;
  carry = raw > 2048u; neg = carry != 0;
  return carry ? 4096u - raw : raw;
}
// grid = rows.  scal: canonical 32-byte scalars, row r at scal + r * row_stride.  sorted + r * items_stride: the row's pairs ((3 w) * table_stride + column) | sign << 31 in
// bucket order; offs + r * 2049: exclusive offsets of the buckets (+ the total); perm + r * 2048: bucket numbers by ascending size.
__global__ void __launch_bounds__(MSM_THREADS) k_msm_pip_sort(const uint8_t* __restrict__ scal, size_t row_stride, uint32_t n_cols, uint32_t table_stride, uint32_t* __restrict__ sorted,
                                                               size_t items_stride, uint32_t* __restrict__ offs, uint16_t* __restrict__ perm, uint8_t* __restrict__ vtop, uint32_t* digit_count) {
  __shared__ uint32_t cnt[MSM_PIP_BUCKETS], key[MSM_PIP_BUCKETS], tsum[MSM_THREADS], topsum;
  const uint32_t t = threadIdx.x; const size_t r = blockIdx.x;
  const uint32_t* row = reinterpret_cast<const uint32_t*>(scal + r * row_stride);
  for (uint32_t i = t; i < MSM_PIP_BUCKETS; i += MSM_THREADS) cnt[i] = 0;
  if (t == 0) topsum = 0;
  __syncthreads();
  uint32_t my_top = 0;
  for (uint32_t c = t; c < n_cols; c += MSM_THREADS) {
    uint32_t s[8]; const lasso_u32x4 lo = reinterpret_cast<const lasso_u32x4*>(row + 8 * (size_t)c)[0], hi = reinterpret_cast<const lasso_u32x4*>(row + 8 * (size_t)c)[1];
    s[0] = lo.x; s[1] = lo.y; s[2] = lo.z; s[3] = lo.w; s[4] = hi.x; s[5] = hi.y; s[6] = hi.z; s[7] = hi.w;
    uint32_t carry = 0; bool neg;
#pragma unroll
    for (uint32_t w = 0; w < MSM_PIP_WINDOWS; w++) { const uint32_t d = msm_pip_digit(s, w, carry, neg); if (d) atomicAdd(&cnt[d - 1], 1u); }
    const uint32_t v = (s[7] >> 28) + carry;          // the value above the bucket windows: times 2^252 G_c, added by k_msm_pip_reduce
    vtop[r * n_cols + c] = (uint8_t)v; my_top += v;
  }
  if (my_top) atomicAdd(&topsum, my_top);
  __syncthreads();
  // exclusive offsets: eight consecutive buckets per thread, a scan over the threads' sums
  uint32_t mine[MSM_PIP_PER_THREAD], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < MSM_PIP_PER_THREAD; k++) { mine[k] = cnt[MSM_PIP_PER_THREAD * t + k]; sum += mine[k]; }
  tsum[t] = sum;
  __syncthreads();
  for (uint32_t off = 1; off < MSM_THREADS; off <<= 1) { const uint32_t v = t >= off ? tsum[t - off] : 0; __syncthreads(); tsum[t] += v; __syncthreads(); }
  uint32_t run = tsum[t] - sum;
  uint32_t* of = offs + r * (MSM_PIP_BUCKETS + 1);
#pragma unroll
  for (uint32_t k = 0; k < MSM_PIP_PER_THREAD; k++) {
    const uint32_t b = MSM_PIP_PER_THREAD * t + k;
    of[b] = run; cnt[b] = run;                                               // cnt: now the scatter's cursors
    key[b] = ((mine[k] < (1u << 21) ? mine[k] : (1u << 21) - 1u) << 11) | b;   // size (clipped: it only orders the work) above the bucket number
    run += mine[k];
  }
  if (t == MSM_THREADS - 1) { of[MSM_PIP_BUCKETS] = run; if (digit_count && run + topsum) atomicAdd(digit_count + (blockIdx.x & 63u), run + topsum); }
  __syncthreads();
  // buckets by ascending size (bitonic, 2048 keys, eight per thread and stage)
  for (uint32_t k = 2; k <= MSM_PIP_BUCKETS; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = t; i < MSM_PIP_BUCKETS; i += MSM_THREADS) {
        const uint32_t x = i ^ j;
        if (x > i) { const uint32_t a = key[i], b = key[x]; if ((a > b) == ((i & k) == 0)) { key[i] = b; key[x] = a; } }
      }
      __syncthreads();
    }
  for (uint32_t i = t; i < MSM_PIP_BUCKETS; i += MSM_THREADS) perm[r * MSM_PIP_BUCKETS + i] = (uint16_t)(key[i] & (MSM_PIP_BUCKETS - 1u));
  // the scatter
  uint32_t* so = sorted + r * items_stride;
  for (uint32_t c = t; c < n_cols; c += MSM_THREADS) {
    uint32_t s[8]; const lasso_u32x4 lo = reinterpret_cast<const lasso_u32x4*>(row + 8 * (size_t)c)[0], hi = reinterpret_cast<const lasso_u32x4*>(row + 8 * (size_t)c)[1];
    s[0] = lo.x; s[1] = lo.y; s[2] = lo.z; s[3] = lo.w; s[4] = hi.x; s[5] = hi.y; s[6] = hi.z; s[7] = hi.w;
    uint32_t carry = 0; bool neg;
#pragma unroll
    for (uint32_t w = 0; w < MSM_PIP_WINDOWS; w++) {
      const uint32_t d = msm_pip_digit(s, w, carry, neg);
      if (d) so[atomicAdd(&cnt[d - 1], 1u)] = (3u * w * table_stride + c) | (neg ? 0x80000000u : 0u);
    }
  }
}
// grid = (rows, 8 steps): workgroup (r, k) sums the 256 buckets of ranks 256 k .. 256 k + 255 of row r, one per lane.  bk + (r * 2048 + b): the sum of bucket b of row r.
// (Rows in x: with the steps in x every step-k workgroup lands on XCD k, and the step of the largest buckets becomes one XCD's work.)  Measured on the way here
// (profiles/r06_full_width_commit_ab.txt): the loop is VALU time — folding the table into 1024 entries and dropping the pair loads altogether changed nothing — so the pairs
// are read one at a time, two ahead of the addition that uses them; fetching them eight at a time (32 bytes per lane) bought 0.5 %.
__global__ void __launch_bounds__(MSM_THREADS) k_msm_pip_accumulate(const uint32_t* __restrict__ sorted, size_t items_stride, const uint32_t* __restrict__ offs, const uint16_t* __restrict__ perm,
                                                                     const niels29* __restrict__ table, pt29* __restrict__ bk) {
  const uint32_t t = threadIdx.x, k = blockIdx.y; const size_t r = blockIdx.x;
  const uint32_t* so = sorted + r * items_stride; const uint32_t* of = offs + r * (MSM_PIP_BUCKETS + 1);
  const uint32_t b = perm[r * MSM_PIP_BUCKETS + k * MSM_THREADS + t];
  const uint32_t lo = of[b], hi = of[b + 1];
  pt29 B = pt_identity();
  if (lo < hi) {
    uint32_t p_cur = so[lo], p_nxt = lo + 1 < hi ? so[lo + 1] : 0u;
    niels29 cur = table[p_cur & 0x7fffffffu];
    for (uint32_t pos = lo; pos < hi; pos++) {
      const uint32_t p_nn = pos + 2 < hi ? so[pos + 2] : 0u;          // two pairs ahead: its table address must exist one addition before its entry is needed
      const niels29 nxt = table[p_nxt & 0x7fffffffu];                  // in flight during the addition below (entry 0 when the bucket ends)
      B = pt_madd(B, niels_cond_neg(cur, (p_cur >> 31) != 0));
      cur = nxt; p_cur = p_nxt; p_nxt = p_nn;
    }
  }
  bk[r * MSM_PIP_BUCKETS + b] = B;
}
// grid = rows.  out[r] = sum_b (b + 1) * bk[r][b]  (K = 1 layout of k_points_sum)
__global__ void __launch_bounds__(MSM_THREADS) k_msm_pip_reduce(const pt29* __restrict__ bk, const uint8_t* __restrict__ vtop, uint32_t n_cols, const niels29* __restrict__ top_table,
                                                                 pt29* __restrict__ out) {
  __shared__ pt29 pts[MSM_THREADS];
  __shared__ fe29 st[MSM_ST_ROWS][4];
  const fe29 d2 = fe_d2();
  const uint32_t t = threadIdx.x; const size_t r = blockIdx.x;
  const pt29* B = bk + r * MSM_PIP_BUCKETS + MSM_PIP_PER_THREAD * (size_t)t;
  pt29 R = pt_identity(), T = pt_identity();
  for (uint32_t k = MSM_PIP_PER_THREAD; k-- > 0;) { R = pt_add(R, B[k], d2); T = pt_add(T, R, d2); }   // R = S_t, T = sum_k (k + 1) B_{8t+k}
  pts[t] = R;
  __syncthreads();
  for (uint32_t off = 1; off < MSM_THREADS; off <<= 1) {   // inclusive suffix sums of the S_t
    pt29 v; const bool a = t + off < MSM_THREADS;
    if (a) v = pts[t + off];
    __syncthreads();
    if (a) pts[t] = pt_add(pts[t], v, d2);
    __syncthreads();
  }
  if (t == 0) pts[0] = pt_identity();                      // sum_t t S_t = sum_{j >= 1} suffix_j
  __syncthreads();
  msm_coop_tree(pts, st, MSM_THREADS, d2);
  pt29 W = pt_identity();
  if (t == 0) W = pts[0];
  __syncthreads();
  pts[t] = T;
  __syncthreads();
  msm_coop_tree(pts, st, MSM_THREADS, d2);
  pt29 V = pt_identity();
  if (t == 0) V = pt_add(pts[0], pt_dbl(pt_dbl(pt_dbl(W))), d2);
  __syncthreads();
  // the values above the bucket windows: sum_c vtop[c] * (2^252 G_c), vtop <= 4 for canonical scalars (any value up to 16 is handled: the entry is added that many times)
  pt29 top = pt_identity();
  for (uint32_t c = t; c < n_cols; c += MSM_THREADS) {
    const uint32_t v = vtop[r * n_cols + c];
    if (v) { const niels29 e = top_table[c]; for (uint32_t i = 0; i < v; i++) top = pt_madd(top, e); }
  }
  pts[t] = top;
  __syncthreads();
  msm_coop_tree(pts, st, MSM_THREADS, d2);
  if (t == 0) out[r] = pt_add(V, pts[0], d2);
}

// ------------------------------------------------------------------ row-parallel commitment of SMALL scalars: one mixed addition per byte
// The commitments of the path are L rows x R columns of small integers over shared generators — E = T[dim] holds table values (< 2^8 for AND / OR / XOR over
// 16-bit indices, bits for LT), dim / read / final hold indices and counters (<= 16 bits at the benchmark's sizes).  The bucket kernel above spends one mixed
// addition per non-zero NIBBLE plus a fixed tail per workgroup (digit sort, segmented tree, bit planes, Horner chain: ~45% of a 4096-column row).  With every
// byte multiple m * 256^w * G_j, m = 1..255, tabulated once per generator set (tab8: 255 x n affine Niels entries per byte window, 117 MB for n = 4096 — the
// generators are fixed for the life of a gens object and HBM is 288 GB), a row is a plain sum of ONE table entry per non-zero byte: no sort, no buckets, no
// bit planes, then one cooperative tree.  2^24 8-bit scalars: 2^24 mixed additions instead of ~1.9 * 2^24 + tails.
// tab8[(m-1)*n + j] = m * B_j with B_j = table[(2*w8)*n + j] = 256^w8 * G_j.  One thread per generator; multiples by repeated mixed addition, brought to
// affine Niels form in runs of 16 that share one inversion (Montgomery's trick).
#define MSM8_MULTS 255
// nm: multiples per generator (255 for the commitments' unsigned bytes; 128 for the signed bytes of the latency-shaped MSMs, mult8)
// grid.y > 1: one launch for gridDim.y consecutive byte windows w8 + blockIdx.y, each window's nm * n entries after the previous one's (mult8)
__global__ void __launch_bounds__(64) k_precompute_tab8(const niels29* __restrict__ table, size_t n, uint32_t w8, niels29* __restrict__ tab8, uint32_t nm = MSM8_MULTS) {
  const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (j >= n) return;
  w8 += blockIdx.y;
  const uint32_t wy = blockIdx.y, nwin = gridDim.y;
  const niels29 b = table[(size_t)(2 * w8) * n + j];
  const fe29 d2 = fe_d2();
  tab8[MSM_IDX(wy, 0, j, n, nwin, nm)] = b;
  pt29 P = pt_madd(pt_identity(), b);
  for (uint32_t m0 = 2; m0 <= nm; m0 += 16) {
    pt29 q[16]; fe29 pre[16];
    const uint32_t cnt = nm + 1 - m0 < 16 ? nm + 1 - m0 : 16;
    for (uint32_t c = 0; c < cnt; c++) { P = pt_madd(P, b); q[c] = P; pre[c] = c ? fe_mul(pre[c - 1], P.Z) : P.Z; }
    fe29 inv = fe_inv_chain(pre[cnt - 1]);
    for (uint32_t c = cnt; c-- > 0;) {
      const fe29 zi = c ? fe_mul(inv, pre[c - 1]) : inv;
      if (c) inv = fe_mul(inv, q[c].Z);
      tab8[MSM_IDX(wy, m0 + c - 1, j, n, nwin, nm)] = niels_from_xy29(fe_mul(q[c].X, zi), fe_mul(q[c].Y, zi), d2);
    }
  }
}
// grid = (K chunks, rows).  scal: u32 scalars, row r at scal + r*row_words; W8 = bytes per scalar that can be non-zero (1 or 2).  out[row*K + chunk] = the chunk's
// partial sum in the kernels' point form (k_points_sum finishes the row exactly as it does for k_msm_buckets).
__global__ void __launch_bounds__(MSM_THREADS) k_msm_rows8(const uint32_t* __restrict__ scal, size_t row_words, uint32_t n_cols, uint32_t cols_per_chunk, uint32_t W8,
                                                            const niels29* __restrict__ tab8_0, const niels29* __restrict__ tab8_1, size_t tn, pt29* __restrict__ out, uint32_t* digit_count) {
  __shared__ pt29 pts[MSM_THREADS];
  __shared__ fe29 st[MSM_ST_ROWS][4];
  const fe29 d2 = fe_d2();
  const uint32_t t = threadIdx.x;
  const uint32_t* row = scal + (size_t)blockIdx.y * row_words;
  const uint32_t c0 = blockIdx.x * cols_per_chunk;
  uint32_t c1 = c0 + cols_per_chunk; if (c1 > n_cols) c1 = n_cols;
  pt29 B = pt_identity();
  niels29 cur; bool have = false; uint32_t nadds = 0;
  for (uint32_t c = c0 + t; c < c1; c += MSM_THREADS) {
    const uint32_t v = row[c];
    for (uint32_t w = 0; w < W8; w++) {
      const uint32_t d = (v >> (8 * w)) & 255u;
      nadds += d != 0;
      // the fetch is unconditional (entry 0 for a zero byte) so that it is issued BEFORE the mixed addition below and waited for after it
      const niels29 nxt = (w ? tab8_1 : tab8_0)[d ? MSM_IDX(0, d - 1, c, tn, 1, MSM8_MULTS) : 0];
      if (have) B = pt_madd(B, cur);
      cur = nxt; have = d != 0;
    }
  }
  if (have) B = pt_madd(B, cur);
  msm_count_adds(digit_count, nadds);
  pts[t] = B;
  __syncthreads();
  msm_coop_tree(pts, st, MSM_THREADS, d2);
  if (t == 0) out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = pts[0];
}

// The same commitment with ONE WAVE PER ROW (four rows per workgroup), for launches of many rows (>= 1024).  k_msm_rows8 gives a row 256 threads: at the headline's E (4096
// columns of 8-bit values) that is 16 mixed additions per thread followed by a 256-point tree of nine passes — a quarter of the workgroup's time is the tree.  With 64 lanes
// per row a lane runs 64 additions and the row's partial sums are 64: six plain addition levels inside the wave, no workgroup barrier anywhere, the tree a tenth of the chain.
// Same table entries, same group element per row (other projective representative: the wire bytes do not change).  out[row] = the row's sum (K = 1 layout of k_points_sum).
__global__ void __launch_bounds__(MSM_THREADS) k_msm_rows8w(const uint32_t* __restrict__ scal, size_t row_words, uint32_t n_cols, uint32_t W8, const niels29* __restrict__ tab8_0,
                                                             const niels29* __restrict__ tab8_1, size_t tn, pt29* __restrict__ out, uint32_t rows, uint32_t* digit_count, uint32_t rpw) {
  // Round 6 (profiles/r06_madd_bench_curve25519.txt: this kernel runs at 18.3 G additions/s where a chain of additions sustains 31).  Tried: (i) `rpw` rows per wave so that a
  // launch is two waves per SIMD, all resident at once, instead of 4096 waves on 3072 slots (a round of three per SIMD, then a round of one) — SLOWER (0.966 against 0.896 ms: two
  // waves hide less latency than three), kept as a parameter, default 1; (ii) the row's scalar for the next column loaded one iteration ahead, like the table entry (the compiler's
  // vmcnt(0) at the top of the loop body waited for a load issued two instructions earlier) — kept, worth 1-2 %.
  __shared__ pt29 pts[MSM_THREADS];
  const fe29 d2 = fe_d2();
  const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63u;
  pt29* mine = pts + 64 * wave;
  uint32_t nadds = 0;
  for (uint32_t k = 0; k < rpw; k++) {
    const uint32_t r = (blockIdx.x * (MSM_THREADS / 64) + wave) * rpw + k;   // wave-uniform
    if (r >= rows) break;
    pt29 B = pt_identity();
    const uint32_t* row = scal + (size_t)r * row_words;
    niels29 cur; bool have = false;
    uint32_t v_next = lane < n_cols ? row[lane] : 0u;
    for (uint32_t c = lane; c < n_cols; c += 64) {
      const uint32_t v = v_next;
      v_next = c + 64 < n_cols ? row[c + 64] : 0u;   // in flight during this column's additions
      for (uint32_t w = 0; w < W8; w++) {
        const uint32_t d = (v >> (8 * w)) & 255u;
        nadds += d != 0;
        const niels29 nxt = (w ? tab8_1 : tab8_0)[d ? MSM_IDX(0, d - 1, c, tn, 1, MSM8_MULTS) : 0];   // issued before the addition below, waited for after it
        if (have) B = pt_madd(B, cur);
        cur = nxt; have = d != 0;
      }
    }
    if (have) B = pt_madd(B, cur);
    mine[lane] = B;
    // the wave's 64 partial sums -> one: plain additions, 32 + 16 + .. + 1; only this wave touches `mine`, so its own LDS operations kept in order are all the synchronisation there is
    for (uint32_t s2 = 32; s2 > 0; s2 >>= 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane < s2) { const pt29 x = pt_add(mine[lane], mine[lane + s2], d2); mine[lane] = x; }
    }
    if (lane == 0) out[r] = mine[0];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // lane 0's read of mine[0] before the next row overwrites it
  }
  msm_count_adds(digit_count, nadds);
}

#define MSM_DIRECT_MAX_COLS 160   // columns a workgroup may touch (items_per_chunk <= 64 * (MSM_DIRECT_MAX_COLS - 1))
// Window width of the digit-multiple table a latency-shaped launch reads: WB = 4 (mult: m * 16^w * G, m = 1..8, 64 windows — every generator set has it) or WB = 8
// (mult8: m * 256^w * G, m = 1..128, 32 windows — 8x the bytes, half the mixed additions per scalar; generator sets up to LASSO_MSM_DIRECT8_MAX_N).
template <int WB> struct MsmD {
  static constexpr uint32_t WINDOWS = 256u / WB, LOGW = WB == 4 ? 6u : 5u, MULTS = 1u << (WB - 1), DMASK = (1u << WB) - 1u, PER_WORD = 32u / WB;
  static constexpr uint32_t BIAS = WB == 4 ? 0x88888888u : 0x80808080u;
};
// signed-digit recoding of one canonical scalar into LDS: e = s + 0x88..8 (0x8080..80): digit w = (field w of e) - 2^(WB-1), in [-2^(WB-1), 2^(WB-1) - 1]; s < 2^254, so the top field cannot overflow
template <int WB>
__device__ __forceinline__ void msm_recode(const uint32_t* s, uint32_t* dst) {
  uint64_t carry = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { const uint64_t x = (uint64_t)s[k] + MsmD<WB>::BIAS + carry; dst[k] = (uint32_t)x; carry = x >> 32; }
}
// items [it0, it1) of a row (one item = one (column, window); sb holds the recoded scalars of columns col0..): one mixed addition per non-zero digit
// phys (optional, LDS): table index of every staged column (phys[c - col0]); otherwise the column map decides
// The last, partly filled pass of a chunk (round 6).  A chunk of 544 items is two full passes of the workgroup and 32 items more; as a third pass of the loop below those 32
// items cost wave 0 a third mixed addition in a row — 2.65 us on the launch's critical path for an eighth of a wave's worth of work (profiles/r06_bullet_phase_curve25519_before.txt:
// accumulate 12.4 us).  With `left` the loop stops at the last full pass; the <= 64 entries that remain are fetched FIRST (their latency hides under the full passes) and handed
// back, and msm_coop_leftover adds entry j to the partial sum of lane j with FOUR lanes per addition after the sums are in LDS: one tree level's time (1.2 us).  Edwards build only.
struct MsmLeft { niels29 e; uint32_t valid = 0, rem = 0; };
#ifndef LASSO_BN254
#define MSM_LEFT_MAX 64u
__device__ __forceinline__ void msm_coop_leftover(pt29* pts, fe29 (*st)[4], fe29 (*ln)[3], uint32_t* lval, const MsmLeft& L) {   // all threads; pts[t] already holds the lanes' sums (barrier passed)
  if (!L.rem) return;   // block-uniform
  const uint32_t t = threadIdx.x, c = t & 3u, g = t >> 2;
  if (t < L.rem) { ln[t][0] = L.e.ypx; ln[t][1] = L.e.ymx; ln[t][2] = L.e.t2d; lval[t] = L.valid; }
  __syncthreads();
  const bool act = g < L.rem && lval[g] != 0;
  if (act) st[g][c] = pt_coop4_madd_stage1(pts[g], &ln[g][0], c);
  tree_wave_sync();
  if (act) reinterpret_cast<fe29*>(&pts[g])[c] = pt_coop4_stage2p(&st[g][0], c);
  __syncthreads();
}
#endif
template <int WB>
__device__ __forceinline__ pt29 msm_direct_accumulate(const uint32_t* sb, uint32_t col0, uint32_t it0, uint32_t it1, const MsmColMap& cm, uint32_t row, const niels29* __restrict__ mult, size_t tn,
                                                      uint32_t* digit_count = nullptr, const uint32_t* phys = nullptr, MsmLeft* left = nullptr) {
  typedef MsmD<WB> D;
  const uint32_t t = threadIdx.x;
  pt29 B = pt_identity();
  niels29 cur; bool have = false; uint32_t nadds = 0;
#ifndef LASSO_BN254
  if (left) {
    const uint32_t span = it1 - it0, rem = span % MSM_THREADS;
    if (rem != 0 && rem <= MSM_LEFT_MAX && span > MSM_THREADS) {   // block-uniform
      left->rem = rem; it1 -= rem;
      const uint32_t it = it1 + t;
      bool valid = t < rem; int32_t d = 0; uint32_t c = 0, w = 0;
      if (valid) { c = it >> D::LOGW; w = it & (D::WINDOWS - 1u); d = (int32_t)((sb[(c - col0) * 8 + w / D::PER_WORD] >> (WB * (w % D::PER_WORD))) & D::DMASK) - (int32_t)D::MULTS; valid = d != 0; }
      const uint32_t m = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
      const size_t idx = valid ? MSM_IDX(w, m - 1, (phys ? phys[c - col0] : msm_phys_col(cm, row, c)), tn, D::WINDOWS, D::MULTS) : 0;
      if (t < rem) left->e = niels_cond_neg(mult[idx], d < 0);   // only the first rem lanes fetch (<= one wave's worth): in flight while the full passes run
      left->valid = valid ? 1u : 0u; nadds += valid;
    }
  }
#endif
  for (uint32_t base = it0; base < it1; base += MSM_THREADS) {
    const uint32_t it = base + t;
    bool valid = it < it1; int32_t d = 0; uint32_t c = 0, w = 0;
    if (valid) { c = it >> D::LOGW; w = it & (D::WINDOWS - 1u); d = (int32_t)((sb[(c - col0) * 8 + w / D::PER_WORD] >> (WB * (w % D::PER_WORD))) & D::DMASK) - (int32_t)D::MULTS; valid = d != 0; }
    // the fetch is unconditional (entry 0 for a skipped item) so that it is issued BEFORE the mixed add below and waited for after it
    const uint32_t m = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
    const size_t idx = valid ? MSM_IDX(w, m - 1, (phys ? phys[c - col0] : msm_phys_col(cm, row, c)), tn, D::WINDOWS, D::MULTS) : 0;
    const niels29 nxt = mult[idx];
    if (have) B = pt_madd(B, cur);
    const bool neg = d < 0;
#ifdef LASSO_BN254
    cur = niels_cond_neg(nxt, neg);
#else
#pragma unroll
    for (int k = 0; k < 9; k++) { cur.ypx.v[k] = neg ? nxt.ymx.v[k] : nxt.ypx.v[k]; cur.ymx.v[k] = neg ? nxt.ypx.v[k] : nxt.ymx.v[k]; cur.t2d.v[k] = neg ? -nxt.t2d.v[k] : nxt.t2d.v[k]; }
#endif
    have = valid; nadds += valid;
  }
  if (have) B = pt_madd(B, cur);
  msm_count_adds(digit_count, nadds);
  return B;
}
// the workgroup's 256 partial sums -> one point (cooperative tree), then across the K workgroups of the row in the last one to arrive (ticket),
// then the row's sum to the host-mapped result buffer in ark's Montgomery limbs; the workgroup that completes the last row raises the flag.
// counters[0..rows) = per-row arrival tickets, counters[16] = finished rows.
__device__ __forceinline__ void msm_direct_finish(pt29* pts, fe29 (*st)[4], uint32_t* is_last, const pt29& B, uint32_t K, uint32_t row, pt29* __restrict__ partial, ed_point* __restrict__ out_mont,
                                                  uint32_t* counters, uint32_t* flag, uint32_t seq, const MsmLeft* left = nullptr, fe29 (*ln)[3] = nullptr, uint32_t* lval = nullptr,
                                                  const fr_t* head0 = nullptr, const fr_t* head1 = nullptr) {
  const fe29 d2 = fe_d2();
  const uint32_t t = threadIdx.x;
  pts[t] = B;
  __syncthreads();
#ifndef LASSO_BN254
  if (left) msm_coop_leftover(pts, st, ln, lval, *left);
#endif
  MSM_STAMP(13);
  msm_coop_tree(pts, st, MSM_THREADS, d2, 16);
  MSM_STAMP(3);
  if (K > 1) {
    // round 6: the partial leaves as 18 write-through (sc1) 8-byte stores by 18 lanes of wave 0 — one store instruction instead of 36 by one lane — and what orders it before the
    // ticket is the wave's own `s_waitcnt vmcnt(0)`, not an L2 write-back (cdna_hip_programming.md §6 G16, the sc1 form: 0.3-1.0 us cheaper per episode than release + plain stores;
    // the reader keeps its agent-scope acquire + plain loads).  LASSO_MSM_PLAIN_PARTIALS (compile time): the round-5 form.
#ifndef LASSO_MSM_PLAIN_PARTIALS
    static_assert(sizeof(pt29) == 144, "pt29 = 18 x 8 bytes");
    if (t < 18) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(&pts[0]);
      const uint64_t v = (uint64_t)src[2 * t] | ((uint64_t)src[2 * t + 1] << 32);
      __hip_atomic_store(reinterpret_cast<uint64_t*>(&partial[(size_t)row * K + blockIdx.x]) + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // all 18 lanes are wave 0's: its stores have been acknowledged before lane 0 takes the ticket
#endif
    if (t == 0) {
#ifdef LASSO_MSM_PLAIN_PARTIALS
      partial[(size_t)row * K + blockIdx.x] = pts[0];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      const uint32_t ticket = __hip_atomic_fetch_add(&counters[row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t last = ticket == K - 1 ? 1u : 0u;
      if (last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); counters[row] = 0; }
      *is_last = last;
    }
    MSM_STAMP(11);
    __syncthreads();
    if (!*is_last) return;
    pt29 acc = t < K ? partial[(size_t)row * K + t] : pt_identity();
    for (uint32_t k = t + MSM_THREADS; k < K; k += MSM_THREADS) acc = pt_add(acc, partial[(size_t)row * K + k], d2);
    __syncthreads();
    pts[t] = acc;
    __syncthreads();
    MSM_STAMP(12);
    msm_coop_tree(pts, st, K < MSM_THREADS ? K : MSM_THREADS, d2, 24);
  }
  MSM_STAMP(4);
  if (flag == LASSO_TAGGED) {
    // round 6 (profiles/r06_bullet_phase_*_before.txt: conversion by ONE lane + system fence + ticket + flag = 2.5-3.4 us at the very end of every launch's critical path): the row's
    // point leaves as four self-validating elements (poly_kernels.cuh result_store: three 16-byte chunks each, tagged with the launch's sequence number), one coordinate per lane —
    // no ticket between the rows, no flag; the host reads the 2 x 4 elements of the launch like any tagged result.  out_mont = the context's tagged area.
    if (t < 4) {
      const fq_t q = pt_coord_abi(pts[0], t);
      fr_t v;
#pragma unroll
      for (int k = 0; k < 8; k++) v.v[k] = q.v[k];
      result_store(reinterpret_cast<fr_t*>(out_mont), (size_t)row * 4 + t, v, flag, seq);
    }
    // the opening's tail chain: the two heads a[0], b[0] the fold in front of this launch left ride behind the point(s) — elements 4 * rows and 4 * rows + 1 — instead of a
    // launch and a hand-off of their own (lasso_read_heads)
    if (head0 != nullptr && row == 0 && t >= 4 && t < 6) result_store(reinterpret_cast<fr_t*>(out_mont), (size_t)gridDim.y * 4 + (t - 4), (t == 4 ? head0 : head1)[0], flag, seq);
    if (t == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // the wave's stores leave the device (row_done's form)
    MSM_STAMP(5);
    return;
  }
  if (t == 0) {
#ifdef LASSO_BN254
    out_mont[row] = pt_to_abi(pts[0]);
#else
    ed_point p = pt_to_ed(pts[0]), o; o.X = fq_to_mont(p.X); o.Y = fq_to_mont(p.Y); o.T = fq_to_mont(p.T); o.Z = fq_to_mont(p.Z); out_mont[row] = o;
#endif
    if (flag) {
      __threadfence_system();
      const uint32_t t2 = __hip_atomic_fetch_add(&counters[16], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (t2 == gridDim.y - 1) { counters[16] = 0; __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
  }
  MSM_STAMP(5);
}
// grid = (K chunks, rows).  scal: 8 words per scalar, row r at scal + r*row_words.  One item = one (column, window); chunk k owns items
// [k*items_per_chunk, ...).  out_mont[row] (host-mapped) = the row's sum in ark's Montgomery limbs.
// MODE 0: scal holds canonical little-endian integers.  MODE 1: scal holds field elements in memory (Montgomery) form, converted here — the
// k_fr_to_canonical pass in front of the opening's Cx = <x, G> saved.  MODE 2: as 1, but columns below n_cols - 2 are multiplied by `scale` first and the
// last two columns are the scalars tail0, tail1 (delta = d * g_hat + r_delta * h over the resident fold weights, dot_product.rs:219-224).
// sstride / soffset: column j takes scalar j * sstride + soffset (1, 0 = plain; world, rank = slab mode's share of a whole vector).
template <int MODE, int WB>
__global__ void __launch_bounds__(MSM_THREADS) k_msm_direct(const uint32_t* __restrict__ scal, size_t row_words, uint32_t n_cols, uint32_t items_per_chunk, MsmColMap cm,
                                                             const niels29* __restrict__ mult, size_t tn, pt29* __restrict__ partial, ed_point* __restrict__ out_mont, uint32_t* counters,
                                                             uint32_t* flag, uint32_t seq, fr_t scale, fr_t tail0, fr_t tail1, uint32_t* digit_count, uint32_t sstride, uint32_t soffset,
                                                             const fr_t* head0 = nullptr, const fr_t* head1 = nullptr, const uint32_t* gate_gmail = nullptr) {
  // behind a gate that may have ended without a challenge (lasso_bullet_tail_ahead): then the fold in front did nothing and this launch must publish nothing
  if (gate_gmail != nullptr && __hip_atomic_load(gate_gmail + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) return;
  __shared__ pt29 pts[MSM_THREADS];
  __shared__ fe29 st[MSM_ST_ROWS][4];
  __shared__ uint32_t sb[MSM_DIRECT_MAX_COLS * 8];
  __shared__ uint32_t is_last;
#ifndef LASSO_BN254
  __shared__ fe29 ln[MSM_LEFT_MAX][3];
  __shared__ uint32_t lval[MSM_LEFT_MAX];
#else
  fe29 (*ln)[3] = nullptr; uint32_t* lval = nullptr;
#endif
  const uint32_t t = threadIdx.x, row = blockIdx.y, K = gridDim.x;
  const uint32_t total = n_cols * MsmD<WB>::WINDOWS;
  const uint32_t it0 = blockIdx.x * items_per_chunk;
  uint32_t it1 = it0 + items_per_chunk; if (it1 > total) it1 = total;
  const uint32_t col0 = it0 >> MsmD<WB>::LOGW, col1 = (it1 + MsmD<WB>::WINDOWS - 1u) >> MsmD<WB>::LOGW;
  MSM_STAMP(0);
  for (uint32_t c = t; c < col1 - col0; c += MSM_THREADS) {
    // column j reads scalar j * sstride + soffset (slab mode: this rank's generators are every P-th one, the scalar vector is the whole one)
    const uint32_t* s = scal + (size_t)row * row_words + ((size_t)(col0 + c) * sstride + soffset) * 8;
    if (MODE == 0) msm_recode<WB>(s, &sb[c * 8]);
    else {
      fr_t v;
      if (MODE == 2 && col0 + c + 2 >= n_cols) v = fr29_to_integer(fr29_unpack_u(col0 + c + 2 == n_cols ? tail0 : tail1));
      else {
        fr_t x;
#pragma unroll
        for (int k = 0; k < 8; k++) x.v[k] = s[k];
        fr29 k32 = fr29_zero(); k32.v[0] = 32;
        v = MODE == 2 ? fr29_store(fr29_mul(fr29_mul(fr29_unpack_u(x), fr29_unpack_s(scale)), k32)) : fr29_to_integer(fr29_unpack_u(x));
      }
      msm_recode<WB>(v.v, &sb[c * 8]);
    }
  }
  __syncthreads();
  MSM_STAMP(1);
  MsmLeft left;
  const pt29 B = msm_direct_accumulate<WB>(sb, col0, it0, it1, cm, row, mult, tn, digit_count, nullptr, &left);
  MSM_STAMP(2);
  msm_direct_finish(pts, st, &is_last, B, K, row, partial, out_mont, counters, flag, seq, &left, ln, lval, head0, head1);
}

// ------------------------------------------------------------------ row-parallel commitment of FULL-WIDTH scalars over the signed byte-multiple table (round 6)
// MEASURED, NOT THE DEFAULT (lasso_hip.hip run_msm says why: half the additions, but random line reads of a multi-GB table at HBM's random-access rate).
// The bucket kernel spends, per full-width scalar, one mixed addition per non-zero NIBBLE (~60) plus per workgroup a digit sort, a segmented tree, bit planes and a Horner chain.
// Generator sets of up to 2^14 points already hold mult8 — every signed byte multiple m 256^w G_j, m = 1..128, built for the openings' latency-shaped MSMs — and with it a scalar is
// a plain sum of ONE table entry per non-zero byte: 32 additions, no sort, no buckets, no doublings.  This is k_msm_rows8's schedule with k_msm_direct's signed recoding: a
// workgroup owns (a chunk of) a row, stages 128 columns' recoded scalars in LDS at a time, every thread runs its share of the (column, window) items with the next entry in
// flight, one cooperative tree at the end.  scal: canonical little-endian scalars, 8 words each, row r at scal + r * row_words.  out[row * K + chunk] (k_points_sum finishes).
#define MSM_FULL8_COLS 128u
template <int WB>
__global__ void __launch_bounds__(MSM_THREADS) k_msm_rows_full(const uint32_t* __restrict__ scal, size_t row_words, uint32_t n_cols, uint32_t cols_per_chunk, const niels29* __restrict__ mult, size_t tn,
                                                                pt29* __restrict__ out, uint32_t* digit_count) {
  typedef MsmD<WB> D;
  __shared__ pt29 pts[MSM_THREADS];
  __shared__ fe29 st[MSM_ST_ROWS][4];
  __shared__ uint32_t sb[MSM_FULL8_COLS * 8];
  const fe29 d2 = fe_d2();
  const uint32_t t = threadIdx.x;
  const uint32_t* row = scal + (size_t)blockIdx.y * row_words;
  const uint32_t c0 = blockIdx.x * cols_per_chunk;
  uint32_t c1 = c0 + cols_per_chunk; if (c1 > n_cols) c1 = n_cols;
  pt29 B = pt_identity();
  niels29 cur; bool have = false; uint32_t nadds = 0;
  for (uint32_t b0 = c0; b0 < c1; b0 += MSM_FULL8_COLS) {
    const uint32_t nb = c1 - b0 < MSM_FULL8_COLS ? c1 - b0 : MSM_FULL8_COLS;
    __syncthreads();   // the previous batch's digits have been read
    for (uint32_t c = t; c < nb; c += MSM_THREADS) msm_recode<WB>(row + (size_t)(b0 + c) * 8, &sb[c * 8]);
    __syncthreads();
    const uint32_t items = nb * D::WINDOWS;
    for (uint32_t it = t; it < items; it += MSM_THREADS) {
      const uint32_t c = it >> D::LOGW, w = it & (D::WINDOWS - 1u);
      const int32_t d = (int32_t)((sb[c * 8 + w / D::PER_WORD] >> (WB * (w % D::PER_WORD))) & D::DMASK) - (int32_t)D::MULTS;
      const bool valid = d != 0;
      const uint32_t m = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
      const niels29 nxt = mult[valid ? MSM_IDX(w, m - 1, b0 + c, tn, D::WINDOWS, D::MULTS) : 0];   // issued before the addition below, waited for after it
      if (have) B = pt_madd(B, cur);
      cur = niels_cond_neg(nxt, d < 0); have = valid; nadds += valid;
    }
  }
  if (have) B = pt_madd(B, cur);
  msm_count_adds(digit_count, nadds);
  pts[t] = B;
  __syncthreads();
  msm_coop_tree(pts, st, MSM_THREADS, d2);
  if (t == 0) out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = pts[0];
}

// out[row] = sum_k partial[row*K + k], converted to ark's Montgomery limbs (out_mont) or, when out_compressed is given, to the 32-byte wire form.  One workgroup per row; thread t first adds partials
// t, t+256, ... serially, then an LDS tree.  `out_mont` may be host-mapped memory: when `flag` is set, the row that finishes last
// raises the host's sequence flag (same hand-off as last_block_reduce in poly_kernels.cuh).
__global__ void __launch_bounds__(MSM_THREADS) k_points_sum(const pt29* __restrict__ partial, uint32_t K, ed_point* __restrict__ out_mont, uint32_t* __restrict__ out_compressed, uint32_t* counters,
                                                             uint32_t* flag, uint32_t seq) {
  __shared__ pt29 pts[MSM_THREADS];
  const fe29 d2 = fe_d2();
  const uint32_t t = threadIdx.x;
  MSM_STAMP(8);
  pt29 acc = t < K ? partial[(size_t)blockIdx.x * K + t] : pt_identity();
  for (uint32_t k = t + MSM_THREADS; k < K; k += MSM_THREADS) acc = pt_add(acc, partial[(size_t)blockIdx.x * K + k], d2);
  pts[t] = acc;
  __syncthreads();
  const uint32_t live = K < MSM_THREADS ? K : MSM_THREADS;
  for (uint32_t s = MSM_THREADS / 2; s > 0; s >>= 1) { if (t < s && t + s < live) pts[t] = pt_add(pts[t], pts[t + s], d2); __syncthreads(); }
  MSM_STAMP(9);
  if (t == 0 && out_compressed) reinterpret_cast<pt29*>(out_compressed)[blockIdx.x] = pts[0];   // compressed mode: hand the row sum to k_points_compress (one lane per row)
  if (t == 0 && !out_compressed) {
    #ifdef LASSO_BN254
    out_mont[blockIdx.x] = pt_to_abi(pts[0]);
#else
    ed_point p = pt_to_ed(pts[0]), o; o.X = fq_to_mont(p.X); o.Y = fq_to_mont(p.Y); o.T = fq_to_mont(p.T); o.Z = fq_to_mont(p.Z); out_mont[blockIdx.x] = o;
#endif
    if (flag) {
      __threadfence_system();
      uint32_t t2 = __hip_atomic_fetch_add(counters, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (t2 == gridDim.x - 1) { *counters = 0; __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
  }
  MSM_STAMP(10);
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
// gmail != nullptr (the opening's tail chain, lasso_bullet_tail_ahead): enqueued before the host had the last challenge, behind k_gate — u, u^-1 are read where the gate left them
__global__ void __launch_bounds__(256) k_bullet_fold(fr_t* __restrict__ a, fr_t* __restrict__ b, size_t half, const fr_t* __restrict__ w, size_t nw, fr_t* __restrict__ w_out, fr_t u, fr_t u_inv,
                                                      const uint32_t* gmail = nullptr, uint32_t seq = 0) {
  if (gmail != nullptr) {
    __shared__ uint32_t s_mail[17];
    if (!gated_challenge(gmail, seq, 16, s_mail)) return;
#pragma unroll
    for (int k = 0; k < 8; k++) { u.v[k] = s_mail[k]; u_inv.v[k] = s_mail[8 + k]; }
  }
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
// (last-block reduction) and the canonical scalar rows SL, SR of the two MSMs over the ORIGINAL generators (n + 2 entries each, half of them
// zero; or, `compact`, only the n/2 + 2 non-zero ones).
// One thread per (i < nk/2, blk < n/nk): n/2 threads whatever the round, so late rounds are as parallel as early ones.
template <bool FOLD>
__global__ void __launch_bounds__(256) k_bullet_step(const fr_t* __restrict__ a_in, const fr_t* __restrict__ b_in, const fr_t* __restrict__ w_in, fr_t* __restrict__ a_out, fr_t* __restrict__ b_out,
                                                      fr_t* __restrict__ w_out, size_t nk, size_t n, fr_t u, fr_t u_inv, fr_t blind_l, fr_t blind_r, fr_t* __restrict__ SL, fr_t* __restrict__ SR,
                                                      fr_t* __restrict__ partials, uint32_t* counters, uint32_t compact) {
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
    // compact rows (k_msm_direct with the bullet column map): entry g of L belongs to generator blk*nk + half + i, entry g of R to blk*nk + i
    const fr_t sl = fr29_store(fr29_mul(fr29_mul(wv, a0), fr29_int_from_uu())), sr = fr29_store(fr29_mul(fr29_mul(wv, a1), fr29_int_from_uu()));
    if (compact) { SL[g] = sl; SR[g] = sr; }
    else { const size_t base = blk * nk + i; SL[base] = fr_zero(); SL[base + half] = sl; SR[base] = sr; SR[base + half] = fr_zero(); }
  }
  store_block_partials<3>(acc, 2, partials + 2 * (size_t)blockIdx.x, 5, S);
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
    const size_t tail = compact ? n / 2 : n;
    if (threadIdx.x == 0) { SL[tail] = v; SL[tail + 1] = fr29_to_integer(fr29_unpack_u(blind_l)); } else { SR[tail] = v; SR[tail + 1] = fr29_to_integer(fr29_unpack_u(blind_r)); }
  }
}

// ------------------------------------------------------------------ one bullet round in ONE launch (bullet.rs:66-132): k_bullet_step + k_msm_direct fused
// A round is on the proof's critical path (the host cannot draw u_{k+1} before it has L_k, R_k; 44-50 rounds per proof), and as two launches it paid
// the fold kernel (15-22 us: a', b', w', the inner products through a last-block reduction, the scalar rows written to HBM), the gap to the next
// launch, and the MSM kernel reading those rows back.  Here every MSM workgroup derives the scalars of ITS columns itself — column g = (i, blk) of
// row L is w'_blk * a'_L[i] over generator blk*nk + half + i, of row R w'_blk * a'_R[i] over blk*nk + i; 3 to 5 field products per column, <= 33 columns
// per workgroup — straight into the LDS digit buffer, and one extra workgroup per row (blockIdx.x == 0) folds a and b, takes the row's inner
// product (c_L = <a'_L, b'_R> for row 0, c_R = <a'_R, b'_L> for row 1) and contributes c * Q + blind * H as its partial sum.  The state of the next round
// (a', b' by the two extra workgroups, w' by the row-0 workgroups that own a column with i = 0) is written on the way.
// grid = (1 + K, 2): the extra workgroup, then K chunks over the n/2 columns x 64 windows of a row; 2 (K + 1) <= the CU count, so that every workgroup has a
// CU to itself (at 258 workgroups the two that had to wait for a free CU — the extra ones — put 18 us on every round).  FOLD = false: first round, the inputs are the state.
// SLAB MODE (P > 1: one proof over P GPUs, the generator vector split by residue class like every other array — `mult` is the table of the generators
// G_{jl * P + rank}, jl < n / P, then Q, H): the rank adds up only ITS generators' share of L and R (the host sums the P partial points), so the per-round MSM —
// the openings' critical path — shrinks by P.  a, b, w are small and stay replicated: every rank folds them in full, and rank 0 alone adds c * Q + blind * H.
// Which of a rank's generators feed which row: generator j = blk * nk + pos belongs to L if pos >= half (scalar w_blk a'_L[pos - half]), to R otherwise
// (w_blk a'_R[pos]).  While half >= P the low bits of pos are the rank, both rows get n / (2 P) local columns and the local picture is the global one with nk / P,
// half / P; in the last log2 P rounds (half < P) pos = rank mod nk is fixed, so ALL n / P of the rank's generators feed one row and none the other.
template <bool FOLD, int WB>
__global__ void __launch_bounds__(MSM_THREADS) k_bullet_msm(const fr_t* __restrict__ a_in, const fr_t* __restrict__ b_in, const fr_t* __restrict__ w_in, fr_t* __restrict__ a_out,
                                                             fr_t* __restrict__ b_out, fr_t* __restrict__ w_out, uint32_t nk, uint32_t n, fr_t u, fr_t u_inv, fr_t blind_l, fr_t blind_r,
                                                             uint32_t items_per_chunk, const niels29* __restrict__ mult, size_t tn, pt29* __restrict__ partial, ed_point* __restrict__ out_mont,
                                                             uint32_t* counters, uint32_t* flag, uint32_t seq, uint32_t* digit_count, uint32_t P, uint32_t rank,
                                                             const uint32_t* mail, uint32_t* gmail) {   // no __restrict__: the host writes the mailbox while the kernel polls it (with it the compiler may keep the first read)
  __shared__ pt29 pts[MSM_THREADS];   // the extra workgroup's reduction scratch (RedScratch, 29.6 KB) lives here before the tree needs it
  __shared__ fe29 st[MSM_ST_ROWS][4];
  __shared__ uint32_t sb[MSM_DIRECT_MAX_COLS * 8];
  __shared__ uint32_t sphys[MSM_DIRECT_MAX_COLS];
  __shared__ uint32_t is_last;
#ifndef LASSO_BN254
  __shared__ fe29 ln[MSM_LEFT_MAX][3];
  __shared__ uint32_t lval[MSM_LEFT_MAX];
#else
  fe29 (*ln)[3] = nullptr; uint32_t* lval = nullptr;
#endif
  MsmLeft left;
  static_assert(sizeof(RedScratch) <= sizeof(pt29) * MSM_THREADS, "RedScratch must fit the point buffer");
  const uint32_t t = threadIdx.x, row = blockIdx.y, K = gridDim.x - 1;
  const uint32_t half = nk / 2, n_loc = n / P;
  // LAUNCHED AHEAD (mail != nullptr): the host enqueued this round behind the previous one, before it knew the challenge — so that neither the launch nor the dispatch sits between
  // two rounds (27 of the 31 us between two bullet launches were launch and completion latency, 4 us host work).  Since round 5 the wait is a GATE KERNEL in front of this one
  // (poly_kernels.cuh k_gate: one wave polls the two host-mapped mailboxes and leaves u, u^-1 in gmail[0..16) behind a tag); this kernel starts when the gate ends and reads them
  // there.  Round 4's form — workgroup (0, 0) polling inside this kernel, the other ~170 spinning on the tag — cost 40 ns per waiting workgroup on the one cache line they shared.
  if (FOLD && mail != nullptr) {
    __shared__ uint32_t s_mail[17];
    if (!gated_challenge(gmail, seq, 16, s_mail)) return;
#pragma unroll
    for (int k = 0; k < 8; k++) { u.v[k] = s_mail[k]; u_inv.v[k] = s_mail[8 + k]; }
  }
  const fr29 us = fr29_unpack_s(u), uis = fr29_unpack_s(u_inv);
  pt29 B;
  MSM_STAMP(0);
  // the extra workgroup is blockIdx.x == 0: dispatched first, because in the early rounds (long a, b) it is the longest of the launch
  if (blockIdx.x > 0) {
    const bool wide = half >= P;                       // always when P == 1
    const uint32_t hl = wide ? half / P : 1u, pos0 = rank & (nk - 1u);
    const uint32_t ncols = wide ? n_loc / 2 : (((pos0 >= half) == (row == 0)) ? n_loc : 0u);   // this row's local columns
    const uint32_t total = ncols * MsmD<WB>::WINDOWS;
    uint32_t it0 = (blockIdx.x - 1) * items_per_chunk; if (it0 > total) it0 = total;
    uint32_t it1 = it0 + items_per_chunk; if (it1 > total) it1 = total;
    const uint32_t col0 = it0 >> MsmD<WB>::LOGW, col1 = (it1 + MsmD<WB>::WINDOWS - 1u) >> MsmD<WB>::LOGW;
    for (uint32_t c = t; c < col1 - col0; c += MSM_THREADS) {
      const uint32_t g = col0 + c;
      uint32_t blk, i, jl;   // fold-weight block, index into a'_L / a'_R, index into the (local) generator table
      if (wide) { blk = g / hl; const uint32_t il = g - blk * hl; i = il * P + rank; jl = blk * (nk / P) + il + (row == 0 ? hl : 0u); }
      else { jl = g; blk = (g * P + rank) / nk; i = row == 0 ? pos0 - half : pos0; }
      const uint32_t ia = i + (row ? half : 0u);   // row 0 (L) takes a'_L[i], row 1 (R) a'_R[i]
      fr29 av, wv;   // canonical u-form
      if (FOLD) {
        av = fr29_canonical(fr29_add(fr29_mul(fr29_unpack_u(a_in[ia]), us), fr29_mul(fr29_unpack_u(a_in[ia + nk]), uis)));
        wv = fr29_canonical(fr29_mul(fr29_unpack_u(w_in[blk >> 1]), (blk & 1) ? us : uis));
        if (P == 1 && row == 0 && i == 0) w_out[blk] = fr29_pack(wv);
      } else { av = fr29_unpack_u(a_in[ia]); wv = fr29_unpack_u(w_in[blk]); }
      // mul(u, u) = wv * a * 2^251; one more Montgomery step with the integer 2^10 gives the canonical integer wv * a
      const fr_t s = fr29_store(fr29_mul(fr29_mul(wv, av), fr29_int_from_uu()));
      msm_recode<WB>(s.v, &sb[c * 8]);
      sphys[c] = jl;
    }
    __syncthreads();
    MSM_STAMP(1);
    const MsmColMap id = {0, 0, 0, 0};
    B = msm_direct_accumulate<WB>(sb, col0, it0, it1, id, row, mult, tn, digit_count, sphys, &left);
  } else {
    RedScratch& S = *reinterpret_cast<RedScratch*>(pts);
    fr29 acc[3] = {fr29_zero(), fr29_zero(), fr29_zero()}; uint32_t cnt = 0;
    for (uint32_t i = t; i < half; i += MSM_THREADS) {
      const uint32_t ia = i + (row ? half : 0u), ib = i + (row ? 0u : half);   // c_L = <a_L, b_R>, c_R = <a_R, b_L>  (bullet.rs:79-80)
      fr29 x, y;
      if (FOLD) {
        x = fr29_canonical(fr29_add(fr29_mul(fr29_unpack_u(a_in[ia]), us), fr29_mul(fr29_unpack_u(a_in[ia + nk]), uis)));
        y = fr29_canonical(fr29_add(fr29_mul(fr29_unpack_u(b_in[ib]), uis), fr29_mul(fr29_unpack_u(b_in[ib + nk]), us)));
        a_out[ia] = fr29_pack(x); b_out[ib] = fr29_pack(y);
      } else { x = fr29_unpack_u(a_in[ia]); y = fr29_unpack_u(b_in[ib]); }
      acc[0] = fr29_weak(fr29_add(acc[0], fr29_mul(x, y)));   // u * u products are 2^5 short: the 2^10 below covers it
      if ((++cnt & 127u) == 0) acc[0] = fr29_mul(acc[0], fr29_one_s());
    }
    if (FOLD && P > 1) {   // slab mode: no rank owns a column of every block, so the fold weights w' (n / nk of them, replicated) are written here — even blocks by row 0, odd by row 1
      for (uint32_t blk = 2 * t + row; blk < n / nk; blk += 2 * MSM_THREADS) w_out[blk] = fr29_store(fr29_mul(fr29_unpack_u(w_in[blk >> 1]), (blk & 1) ? us : uis));
    }
    block_columns<3>(acc, S);
    if (t == 0) {
      int64_t c[9];
#pragma unroll
      for (int k = 0; k < 9; k++) c[k] = S.cols[k];
      const fr_t ci = fr29_store(fr29_mul(fr29_from_columns(c), fr29_int_from_uu()));   // sum of (u*u) products -> canonical integer
      const fr_t bi = fr29_to_integer(fr29_unpack_u(row ? blind_r : blind_l));
      msm_recode<WB>(ci.v, &sb[0]); msm_recode<WB>(bi.v, &sb[8]);
    }
    __syncthreads();   // S (aliasing pts) is dead from here on
    MSM_STAMP(1);
    const MsmColMap id = {0, 0, 0, 0};
    // columns n_loc (Q) and n_loc + 1 (H) of the table; in slab mode rank 0 alone adds them
    B = msm_direct_accumulate<WB>(sb, 0, 0, rank == 0 ? 2 * MsmD<WB>::WINDOWS : 0u, id, row, MSM_COL_BASE(mult, n_loc, MsmD<WB>::WINDOWS, MsmD<WB>::MULTS), tn, digit_count);
  }
  MSM_STAMP(2);
  msm_direct_finish(pts, st, &is_last, B, K + 1, row, partial, out_mont, counters, flag, seq, &left, ln, lval);
}

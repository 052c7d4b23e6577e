// HIP kernels for the polynomial half of the Lasso hot path (gfx950, wave64).
// Each kernel streams 32-byte Fr elements (two dwordx4 per lane) and is HBM-bound by design except the
// sumcheck round evaluators, which are integer-ALU-bound (6 Montgomery products per 128 B).  Reductions are
// exact field sums, so any tree order gives bit-identical results to the reference's serial/rayon loops.
#pragma once
#include <hip/hip_runtime.h>
#include "fr.cuh"

#define LASSO_MAX_PTRS 136   // 2 * (2 * 33 memories) + slack; pointer tables travel by value in the kernarg segment
#define LASSO_BLOCK 256

struct PtrTable { const fr_t* p[LASSO_MAX_PTRS]; };
struct MutPtrTable { fr_t* p[LASSO_MAX_PTRS]; };

struct StrategyDev { int kind; uint32_t c, log_m, log_r, alpha; };

// ------------------------------------------------------------------ reductions
__device__ __forceinline__ fr_t shfl_down_fr(const fr_t& a, int off) {
  fr_t r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = __shfl_down(a.v[i], off, 64);
  return r;
}
__device__ __forceinline__ fr_t wave_reduce_fr(fr_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fr_add(v, shfl_down_fr(v, off));
  return v;  // lane 0 holds the wave total
}
// sum over the 256-thread block; result valid in thread 0.  smem: 4 fr_t
__device__ __forceinline__ fr_t block_reduce_fr(fr_t v, fr_t* smem) {
  v = wave_reduce_fr(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) { v = smem[0]; for (int w = 1; w < (int)(blockDim.x >> 6); w++) v = fr_add(v, smem[w]); }
  return v;
}
// second stage: out[y*K + k] = sum_x partials[(y*nx + x)*K + k]
__global__ void k_reduce_partials(const fr_t* __restrict__ partials, uint32_t nx, uint32_t K, fr_t* __restrict__ out) {
  __shared__ fr_t smem[4];
  const uint32_t y = blockIdx.x;
  for (uint32_t k = 0; k < K; k++) {
    fr_t acc = fr_zero();
    for (uint32_t x = threadIdx.x; x < nx; x += blockDim.x) acc = fr_add(acc, partials[((size_t)y * nx + x) * K + k]);
    acc = block_reduce_fr(acc, smem);
    if (threadIdx.x == 0) out[(size_t)y * K + k] = acc;
  }
}

// ------------------------------------------------------------------ K1: bound_poly_var_top (dense_mlpoly.rs:209-216)
// grid = (blocks over i, polys).  Z[i] <- Z[i] + r*(Z[i+half] - Z[i])
__global__ void __launch_bounds__(LASSO_BLOCK) k_bind_top(MutPtrTable polys, size_t half, fr_t r) {
  fr_t* __restrict__ z = polys.p[blockIdx.y];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    fr_t lo = z[i], hi = z[i + half];
    z[i] = fr_add(lo, fr_mul(r, fr_sub(hi, lo)));
  }
}

// ------------------------------------------------------------------ K4: cubic round (sumcheck.rs:49-93)
// grid = (blocks over i, circuits); partials[(c*nx + bx)*3 + {0,1,2}] = evals at x = 0, 2, 3
__global__ void __launch_bounds__(LASSO_BLOCK) k_cubic_round(PtrTable A, PtrTable B, const fr_t* __restrict__ C, size_t half, fr_t* __restrict__ partials) {
  __shared__ fr_t smem[4];
  const fr_t* __restrict__ a = A.p[blockIdx.y];
  const fr_t* __restrict__ b = B.p[blockIdx.y];
  fr_t e0 = fr_zero(), e2 = fr_zero(), e3 = fr_zero();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    fr_t a0 = a[i], a1 = a[i + half], b0 = b[i], b1 = b[i + half], c0 = C[i], c1 = C[i + half];
    e0 = fr_add(e0, fr_mul(fr_mul(a0, b0), c0));
    fr_t da = fr_sub(a1, a0), db = fr_sub(b1, b0), dc = fr_sub(c1, c0);
    fr_t a2 = fr_add(a1, da), b2 = fr_add(b1, db), c2 = fr_add(c1, dc);   // 2*hi - lo
    e2 = fr_add(e2, fr_mul(fr_mul(a2, b2), c2));
    fr_t a3 = fr_add(a2, da), b3 = fr_add(b2, db), c3 = fr_add(c2, dc);   // 3*hi - 2*lo
    e3 = fr_add(e3, fr_mul(fr_mul(a3, b3), c3));
  }
  e0 = block_reduce_fr(e0, smem); e2 = block_reduce_fr(e2, smem); e3 = block_reduce_fr(e3, smem);
  if (threadIdx.x == 0) {
    fr_t* o = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
    o[0] = e0; o[1] = e2; o[2] = e3;
  }
}

// ------------------------------------------------------------------ in-launch second-stage reduction (no extra kernel per round)
// Block `bx` of row `y` has written its K partial sums (thread 0).  The last block to arrive for that row sums all nx partials
// and writes out[y*K + k].  Hand-off follows cdna_hip_programming.md §6 G16: plain stores -> agent-scope release -> drained
// vmcnt -> relaxed agent atomic ticket; the last arriver does ONE agent-scope acquire, then the workgroup reads plain.
__device__ __forceinline__ void last_block_reduce(const fr_t* partials, uint32_t nx, uint32_t K, uint32_t y, uint32_t nrows, uint32_t* counters, fr_t* __restrict__ out, fr_t* smem,
                                                  uint32_t* flag, uint32_t seq) {
  __shared__ uint32_t is_last;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t ticket = __hip_atomic_fetch_add(&counters[y], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t last = (ticket == nx - 1) ? 1u : 0u;
    if (last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); counters[y] = 0; }   // reset for the next launch (ordered by the kernel boundary)
    is_last = last;
  }
  __syncthreads();
  if (!is_last) return;
  for (uint32_t k = 0; k < K; k++) {
    fr_t acc = fr_zero();
    for (uint32_t x = threadIdx.x; x < nx; x += blockDim.x) acc = fr_add(acc, partials[((size_t)y * nx + x) * K + k]);
    acc = block_reduce_fr(acc, smem);
    if (threadIdx.x == 0) out[(size_t)y * K + k] = acc;
  }
  // `out` is host-mapped memory.  The row that finishes last raises the host's sequence flag: every row's stores are released at
  // system scope before its ticket, so the flag store (system-scope release) is ordered after all of them.
  if (threadIdx.x == 0 && flag) {
    __threadfence_system();
    uint32_t t2 = __hip_atomic_fetch_add(&counters[LASSO_MAX_PTRS], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t2 == nrows - 1) { counters[LASSO_MAX_PTRS] = 0; __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
}

// K4 fused with K1: bind every polynomial of the round with r (length n = 4q -> 2q), then evaluate the NEXT round on the bound
// values while they are still in registers (80 -> 48 bytes per element per round, one launch per round).
// A, B are bound in place (each element is owned by exactly one thread); the shared eq polynomial C is read from C_in and written
// to C_out by the row-0 workgroups only, because every circuit row re-reads it.
__global__ void __launch_bounds__(LASSO_BLOCK) k_cubic_fused(MutPtrTable A, MutPtrTable B, const fr_t* __restrict__ C_in, fr_t* __restrict__ C_out, size_t q, fr_t r,
                                                              fr_t* __restrict__ partials, uint32_t* counters, fr_t* __restrict__ out, uint32_t* flag, uint32_t seq) {
  __shared__ fr_t smem[4];
  fr_t* __restrict__ a = A.p[blockIdx.y];
  fr_t* __restrict__ b = B.p[blockIdx.y];
  fr_t e0 = fr_zero(), e2 = fr_zero(), e3 = fr_zero();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
    fr_t x0 = a[i], x1 = a[i + q], x2 = a[i + 2 * q], x3 = a[i + 3 * q];
    fr_t a0 = fr_add(x0, fr_mul(r, fr_sub(x2, x0))), a1 = fr_add(x1, fr_mul(r, fr_sub(x3, x1)));
    a[i] = a0; a[i + q] = a1;
    x0 = b[i]; x1 = b[i + q]; x2 = b[i + 2 * q]; x3 = b[i + 3 * q];
    fr_t b0 = fr_add(x0, fr_mul(r, fr_sub(x2, x0))), b1 = fr_add(x1, fr_mul(r, fr_sub(x3, x1)));
    b[i] = b0; b[i + q] = b1;
    x0 = C_in[i]; x1 = C_in[i + q]; x2 = C_in[i + 2 * q]; x3 = C_in[i + 3 * q];
    fr_t c0 = fr_add(x0, fr_mul(r, fr_sub(x2, x0))), c1 = fr_add(x1, fr_mul(r, fr_sub(x3, x1)));
    if (blockIdx.y == 0) { C_out[i] = c0; C_out[i + q] = c1; }
    e0 = fr_add(e0, fr_mul(fr_mul(a0, b0), c0));
    fr_t da = fr_sub(a1, a0), db = fr_sub(b1, b0), dc = fr_sub(c1, c0);
    fr_t a2 = fr_add(a1, da), b2 = fr_add(b1, db), c2 = fr_add(c1, dc);
    e2 = fr_add(e2, fr_mul(fr_mul(a2, b2), c2));
    fr_t a3 = fr_add(a2, da), b3 = fr_add(b2, db), c3 = fr_add(c2, dc);
    e3 = fr_add(e3, fr_mul(fr_mul(a3, b3), c3));
  }
  e0 = block_reduce_fr(e0, smem); e2 = block_reduce_fr(e2, smem); e3 = block_reduce_fr(e3, smem);
  if (threadIdx.x == 0) { fr_t* o = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3; o[0] = e0; o[1] = e2; o[2] = e3; }
  last_block_reduce(partials, gridDim.x, 3, blockIdx.y, gridDim.y, counters, out, smem, flag, seq);
}
// first round of a layer: evaluation only, with the in-launch second stage
__global__ void __launch_bounds__(LASSO_BLOCK) k_cubic_round_lb(PtrTable A, PtrTable B, const fr_t* __restrict__ C, size_t half, fr_t* __restrict__ partials, uint32_t* counters,
                                                                 fr_t* __restrict__ out, uint32_t* flag, uint32_t seq) {
  __shared__ fr_t smem[4];
  const fr_t* __restrict__ a = A.p[blockIdx.y];
  const fr_t* __restrict__ b = B.p[blockIdx.y];
  fr_t e0 = fr_zero(), e2 = fr_zero(), e3 = fr_zero();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    fr_t a0 = a[i], a1 = a[i + half], b0 = b[i], b1 = b[i + half], c0 = C[i], c1 = C[i + half];
    e0 = fr_add(e0, fr_mul(fr_mul(a0, b0), c0));
    fr_t da = fr_sub(a1, a0), db = fr_sub(b1, b0), dc = fr_sub(c1, c0);
    fr_t a2 = fr_add(a1, da), b2 = fr_add(b1, db), c2 = fr_add(c1, dc);
    e2 = fr_add(e2, fr_mul(fr_mul(a2, b2), c2));
    fr_t a3 = fr_add(a2, da), b3 = fr_add(b2, db), c3 = fr_add(c2, dc);
    e3 = fr_add(e3, fr_mul(fr_mul(a3, b3), c3));
  }
  e0 = block_reduce_fr(e0, smem); e2 = block_reduce_fr(e2, smem); e3 = block_reduce_fr(e3, smem);
  if (threadIdx.x == 0) { fr_t* o = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3; o[0] = e0; o[1] = e2; o[2] = e3; }
  last_block_reduce(partials, gridDim.x, 3, blockIdx.y, gridDim.y, counters, out, smem, flag, seq);
}

// ------------------------------------------------------------------ g = S::combine_lookups (subtables/*.rs)
#define LASSO_MAX_ALPHA 32
// A = compile-time bound on NUM_MEMORIES so `vals` stays in registers (all indexing static after unrolling)
template <int A>
__device__ __forceinline__ fr_t combine_lookups_dev(const StrategyDev& S, const fr_t* vals, const fr_t* weights) {
  if (S.kind == 3) {  // LT: lt.rs:62-71  sum_i LT[i] * prod_{j<i} EQ[j]
    fr_t sum = fr_zero(), eq_prod = fr_one();
#pragma unroll
    for (int i = 0; i < A / 2; i++) if ((uint32_t)i < S.c) { sum = fr_add(sum, fr_mul(vals[2 * i], eq_prod)); eq_prod = fr_mul(eq_prod, vals[2 * i + 1]); }
    return sum;
  }
  // AND/OR/XOR (and.rs:45-53) and RangeCheck (range_check.rs:78-86): sum_i 2^(i*inc) * vals[i]; weights precomputed in Montgomery form
  fr_t sum = fr_zero();
#pragma unroll
  for (int i = 0; i < A; i++) if ((uint32_t)i < S.alpha) sum = fr_add(sum, fr_mul(weights[i], vals[i]));
  return sum;
}
struct WeightTable { fr_t w[LASSO_MAX_ALPHA]; };

// K3: prove_arbitrary round (sumcheck.rs:165-237).  partials[bx*(degree+1) + x];  D = compile-time bound on degree
template <int A, int D>
__global__ void __launch_bounds__(LASSO_BLOCK) k_combine_round(StrategyDev S, PtrTable polys, const fr_t* __restrict__ eq, WeightTable W, size_t half, uint32_t degree,
                                                                fr_t* __restrict__ partials) {
  __shared__ fr_t smem[4];
  __shared__ fr_t wsm[LASSO_MAX_ALPHA];
  if (threadIdx.x < S.alpha) wsm[threadIdx.x] = W.w[threadIdx.x];
  __syncthreads();
  fr_t acc[D + 1];
#pragma unroll
  for (int x = 0; x <= D; x++) acc[x] = fr_zero();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    fr_t cur[A], dif[A];
#pragma unroll
    for (int j = 0; j < A; j++) if ((uint32_t)j < S.alpha) { fr_t lo = polys.p[j][i], hi = polys.p[j][i + half]; cur[j] = lo; dif[j] = fr_sub(hi, lo); }
    fr_t elo = eq[i], ehi = eq[i + half], ecur = elo, edif = fr_sub(ehi, elo);
#pragma unroll
    for (int x = 0; x <= D; x++) if ((uint32_t)x <= degree) {
      acc[x] = fr_add(acc[x], fr_mul(combine_lookups_dev<A>(S, cur, wsm), ecur));
#pragma unroll
      for (int j = 0; j < A; j++) if ((uint32_t)j < S.alpha) cur[j] = fr_add(cur[j], dif[j]);
      ecur = fr_add(ecur, edif);
    }
  }
#pragma unroll
  for (int x = 0; x <= D; x++) if ((uint32_t)x <= degree) {
    fr_t v = block_reduce_fr(acc[x], smem);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * (degree + 1) + x] = v;
  }
}
// K10: claim = sum_k eq[k] * g(E(k))  (subtables/mod.rs:187-216)
template <int A>
__global__ void __launch_bounds__(LASSO_BLOCK) k_combine_claim(StrategyDev S, PtrTable polys, const fr_t* __restrict__ eq, WeightTable W, size_t n, fr_t* __restrict__ partials) {
  __shared__ fr_t smem[4];
  __shared__ fr_t wsm[LASSO_MAX_ALPHA];
  if (threadIdx.x < S.alpha) wsm[threadIdx.x] = W.w[threadIdx.x];
  __syncthreads();
  fr_t acc = fr_zero();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    fr_t vals[A];
#pragma unroll
    for (int j = 0; j < A; j++) if ((uint32_t)j < S.alpha) vals[j] = polys.p[j][i];
    acc = fr_add(acc, fr_mul(combine_lookups_dev<A>(S, vals, wsm), eq[i]));
  }
  acc = block_reduce_fr(acc, smem);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// K12: out[p] = sum_i polys[p][i] * w[i]; grid = (blocks, polys); partials[p*nx + bx]
__global__ void __launch_bounds__(LASSO_BLOCK) k_multi_dot(PtrTable polys, const fr_t* __restrict__ w, size_t n, fr_t* __restrict__ partials) {
  __shared__ fr_t smem[4];
  const fr_t* __restrict__ z = polys.p[blockIdx.y];
  fr_t acc = fr_zero();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc = fr_add(acc, fr_mul(z[i], w[i]));
  acc = block_reduce_fr(acc, smem);
  if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
}

// ------------------------------------------------------------------ K6: eq evals (eq_poly.rs:22-38)
// small table: out[x] = prod_j (bit_j(x) ? r[j] : 1 - r[j]), bit 0 of the product order = most significant bit of x
struct RTable { fr_t r[32]; };
__global__ void k_eq_small(RTable R, uint32_t ell, fr_t* __restrict__ out) {
  size_t x = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (x >= ((size_t)1 << ell)) return;
  fr_t p = fr_one(), one = fr_one();
  for (uint32_t j = 0; j < ell; j++) { bool bit = (x >> (ell - 1 - j)) & 1; p = fr_mul(p, bit ? R.r[j] : fr_sub(one, R.r[j])); }
  out[x] = p;
}
// out[x] = hi[x >> lo_bits] * lo[x & mask]
__global__ void __launch_bounds__(LASSO_BLOCK) k_eq_outer(const fr_t* __restrict__ hi, const fr_t* __restrict__ lo, uint32_t lo_bits, size_t n, fr_t* __restrict__ out) {
  const size_t mask = ((size_t)1 << lo_bits) - 1;
  for (size_t x = blockIdx.x * (size_t)blockDim.x + threadIdx.x; x < n; x += (size_t)gridDim.x * blockDim.x) out[x] = fr_mul(hi[x >> lo_bits], lo[x & mask]);
}

// ------------------------------------------------------------------ K7: product tree layer (grand_product.rs:20-36)
__global__ void __launch_bounds__(LASSO_BLOCK) k_gp_layer(const fr_t* __restrict__ in, size_t half, fr_t* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) out[i] = fr_mul(in[i], in[i + half]);
}
// the remaining small layers in one workgroup: in has `len` elements (len <= 2*blockDim.x), layers are laid out back to back
__global__ void k_gp_tail(fr_t* __restrict__ tree, size_t len) {
  fr_t* in = tree;
  while (len > 2) {
    size_t half = len / 2; fr_t* out = in + len;
    for (size_t i = threadIdx.x; i < half; i += blockDim.x) out[i] = fr_mul(in[i], in[i + half]);
    __threadfence_block();
    __syncthreads();
    in = out; len = half;
  }
}

// ------------------------------------------------------------------ K8: Reed-Solomon fingerprints (memory_checking.rs:236-310)
__global__ void __launch_bounds__(LASSO_BLOCK) k_fingerprint_ops(const fr_t* __restrict__ table, const uint32_t* __restrict__ dim, const fr_t* __restrict__ read, size_t s,
                                                                  fr_t gamma, fr_t gamma2, fr_t tau, fr_t* __restrict__ out_r, fr_t* __restrict__ out_w) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < s; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t a = dim[i];
    fr_t h = fr_add(fr_mul(read[i], gamma2), fr_mul(table[a], gamma));
    h = fr_sub(fr_add(h, fr_from_u64(a)), tau);
    out_r[i] = h;
    out_w[i] = fr_add(h, gamma2);   // ts+1: (t+1)*gamma^2 = t*gamma^2 + gamma^2
  }
}
__global__ void __launch_bounds__(LASSO_BLOCK) k_fingerprint_mem(const fr_t* __restrict__ table, const fr_t* __restrict__ fin, size_t m, fr_t gamma, fr_t gamma2, fr_t tau,
                                                                  fr_t* __restrict__ out_i, fr_t* __restrict__ out_f) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
    fr_t h = fr_sub(fr_add(fr_mul(table[i], gamma), fr_from_u64(i)), tau);
    out_i[i] = h;
    out_f[i] = fr_add(h, fr_mul(fin[i], gamma2));
  }
}

// ------------------------------------------------------------------ small conversions / gathers
__global__ void __launch_bounds__(LASSO_BLOCK) k_from_u32(const uint32_t* __restrict__ src, size_t n, fr_t* __restrict__ dst) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = fr_from_u64(src[i]);
}
__global__ void __launch_bounds__(LASSO_BLOCK) k_gather(const fr_t* __restrict__ table, const uint32_t* __restrict__ idx, size_t n, fr_t* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = table[idx[i]];
}

__global__ void k_read_heads(PtrTable polys, uint32_t k, fr_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) out[i] = polys.p[i][0];
}

// ------------------------------------------------------------------ K11: L*Z mat-vec (dense_mlpoly.rs:184-207)
// grid = (column blocks, row chunks); partials[chunk*R + col] = sum_{j in chunk} L[j] * Z[j*R + col]
__global__ void __launch_bounds__(LASSO_BLOCK) k_matvec_left(const fr_t* __restrict__ Z, const fr_t* __restrict__ Lv, size_t l_size, size_t r_size, size_t rows_per_chunk,
                                                              fr_t* __restrict__ partials) {
  size_t col = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (col >= r_size) return;
  size_t j0 = (size_t)blockIdx.y * rows_per_chunk, j1 = j0 + rows_per_chunk; if (j1 > l_size) j1 = l_size;
  fr_t acc = fr_zero();
  for (size_t j = j0; j < j1; j++) acc = fr_add(acc, fr_mul(Lv[j], Z[j * r_size + col]));
  partials[(size_t)blockIdx.y * r_size + col] = acc;
}
__global__ void __launch_bounds__(LASSO_BLOCK) k_matvec_reduce(const fr_t* __restrict__ partials, size_t nchunks, size_t r_size, fr_t* __restrict__ out) {
  size_t col = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (col >= r_size) return;
  fr_t acc = fr_zero();
  for (size_t c = 0; c < nchunks; c++) acc = fr_add(acc, partials[c * r_size + col]);
  out[col] = acc;
}

// DensifiedRepresentation::from_lookup_indices on the device (src/lasso/densified.rs:22-75; serial in the reference, TODO(#29)).
//
// For one dimension: access[k] = indices[k][dim] (0 for the padded tail k >= n_lookups, which the reference's loop also counts),
// read_ts[k] = number of earlier positions with the same address, final_ts[a] = number of positions with address a.
// Data-parallel form: a STABLE LSD radix sort of (address, position) by address (8 bits per pass, ceil(log_m / 8) passes) puts the
// positions of every address next to each other in sequence order, so read_ts[pos_i] = i - start_of_run(address_i) and
// final_ts[a] = run length.  The three polynomials are written straight in Montgomery Fr form (DensePolynomial::from_usize).
//
// Kernels (all hand-written; u32 keys/values, 4096 elements per workgroup):
//   k_densify_extract   indices -> keys (+ identity values, + dim polynomial, + range check)
//   k_radix_hist        per-tile digit histogram            hist[digit][tile]
//   k_scan_*            exclusive scan of hist (three-step, any length)
//   k_radix_scatter     stable scatter: in-tile rank by wave ballots (8 ballots give the lanes holding the same digit), waves in order
//   k_densify_runs      run boundaries of the sorted addresses
//   k_densify_read / k_densify_final   timestamps -> Fr polynomials
#pragma once
#include <hip/hip_runtime.h>
#include "fr29.cuh"

#define RADIX_TILE 4096
#define RADIX_THREADS 256

// Slab mode (world = P > 1, one proof sharded over P GPUs): every rank sorts the WHOLE sequence (timestamps are a global property of it)
// but materialises only its residue class: global index k with k mod P == rank lands at local index k / P.
__global__ void __launch_bounds__(256) k_densify_extract(const uint64_t* __restrict__ idx, size_t n_lookups, size_t C, size_t col, size_t s, uint64_t m, uint32_t world, uint32_t rank,
                                                          uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ dim_u32, fr_t* __restrict__ dim_fr, uint32_t* __restrict__ bad) {
  const fr29 r2s = fr29_r2s();
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < s; k += (size_t)gridDim.x * blockDim.x) {
    uint64_t a = k < n_lookups ? idx[k * C + col] : 0;         // access_sequence.resize(s, 0)  densified.rs:38
    if (a >= m) { atomicOr(bad, 1u); a = 0; }                    // debug_assert!(memory_address < m)  :46 — flagged (the call returns LASSO_ERR_INVALID) and
                                                                 // clamped, so the sort / run kernels that index 2*m words of scratch by key stay in bounds
    keys[k] = (uint32_t)a; vals[k] = (uint32_t)k;
    if (k % world == rank) { dim_u32[k / world] = (uint32_t)a; dim_fr[k / world] = fr29_store(fr29_mul(fr29_from_u64_int(a), r2s)); }
  }
}

__global__ void __launch_bounds__(RADIX_THREADS) k_radix_hist(const uint32_t* __restrict__ keys, size_t n, uint32_t shift, uint32_t* __restrict__ hist, uint32_t ntiles) {
  __shared__ uint32_t h[256];
  const uint32_t t = threadIdx.x, tile = blockIdx.x;
  h[t] = 0;
  __syncthreads();
  const size_t base = (size_t)tile * RADIX_TILE;
  for (uint32_t r = 0; r < RADIX_TILE / RADIX_THREADS; r++) { const size_t i = base + r * RADIX_THREADS + t; if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u); }
  __syncthreads();
  hist[(size_t)t * ntiles + tile] = h[t];
}

// exclusive scan of `data` (length n) in three steps over blocks of 4096 entries
__device__ __forceinline__ uint32_t block_exscan_256(uint32_t v, uint32_t* sm /*[256]*/, uint32_t& total) {   // exclusive scan across the 256 threads of the block
  const uint32_t t = threadIdx.x;
  sm[t] = v;
  __syncthreads();
  for (uint32_t off = 1; off < 256; off <<= 1) { uint32_t x = t >= off ? sm[t - off] : 0; __syncthreads(); sm[t] += x; __syncthreads(); }
  total = sm[255];
  const uint32_t ex = sm[t] - v;
  __syncthreads();
  return ex;
}
__global__ void __launch_bounds__(256) k_scan_block_sums(const uint32_t* __restrict__ data, size_t n, uint32_t* __restrict__ sums) {
  __shared__ uint32_t sm[256];
  const size_t base = (size_t)blockIdx.x * 4096 + (size_t)threadIdx.x * 16;
  uint32_t s = 0;
  for (int j = 0; j < 16; j++) if (base + j < n) s += data[base + j];
  uint32_t total; (void)block_exscan_256(s, sm, total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// single workgroup: exclusive scan of sums[0..nb) in place
__global__ void __launch_bounds__(256) k_scan_sums(uint32_t* __restrict__ sums, size_t nb) {
  __shared__ uint32_t sm[256];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (size_t b0 = 0; b0 < nb; b0 += 256) {
    const size_t i = b0 + threadIdx.x;
    const uint32_t v = i < nb ? sums[i] : 0;
    uint32_t total; const uint32_t ex = block_exscan_256(v, sm, total);
    if (i < nb) sums[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry += total;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_scan_apply(uint32_t* __restrict__ data, size_t n, const uint32_t* __restrict__ sums) {
  __shared__ uint32_t sm[256];
  const size_t base = (size_t)blockIdx.x * 4096 + (size_t)threadIdx.x * 16;
  uint32_t v[16], s = 0;
  for (int j = 0; j < 16; j++) { v[j] = base + j < n ? data[base + j] : 0; s += v[j]; }
  uint32_t total; uint32_t run = sums[blockIdx.x] + block_exscan_256(s, sm, total);
  for (int j = 0; j < 16; j++) if (base + j < n) { data[base + j] = run; run += v[j]; }
}

// stable scatter of one radix pass.  offs[digit*ntiles + tile] = first output slot of this tile's elements with that digit.
__global__ void __launch_bounds__(RADIX_THREADS) k_radix_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, size_t n, uint32_t shift, const uint32_t* __restrict__ offs,
                                                                  uint32_t ntiles, uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
  __shared__ uint32_t next[256];   // next free output slot per digit
  const uint32_t t = threadIdx.x, tile = blockIdx.x, lane = t & 63, wave = t >> 6;
  next[t] = offs[(size_t)t * ntiles + tile];
  __syncthreads();
  const size_t base = (size_t)tile * RADIX_TILE;
  for (uint32_t r = 0; r < RADIX_TILE / RADIX_THREADS; r++) {
    const size_t i = base + r * RADIX_THREADS + t;
    const bool valid = i < n;
    const uint32_t key = valid ? keys_in[i] : 0, val = valid ? vals_in[i] : 0, d = (key >> shift) & 255u;
    // lanes of this wave holding the same digit (8 ballots), in lane = sequence order
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) { const uint64_t bal = __ballot((d >> b) & 1u); peers &= ((d >> b) & 1u) ? bal : ~bal; }
    const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull)), cnt = (uint32_t)__popcll(peers);
    uint32_t pos = 0;
    for (uint32_t w = 0; w < RADIX_THREADS / 64; w++) {   // waves take their slots in order: wave w holds elements before wave w+1
      if (wave == w && valid) pos = next[d] + rank;
      __syncthreads();
      if (wave == w && valid && rank == 0) next[d] += cnt;
      __syncthreads();
    }
    if (valid) { keys_out[pos] = key; vals_out[pos] = val; }
  }
}

// run boundaries of the sorted addresses: run_start[a] = first sorted slot with address a, run_end[a] = one past the last (both stay 0 for absent addresses)
__global__ void __launch_bounds__(256) k_densify_runs(const uint32_t* __restrict__ skeys, size_t s, uint32_t* __restrict__ run_start, uint32_t* __restrict__ run_end) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < s; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t a = skeys[i];
    if (i == 0 || skeys[i - 1] != a) run_start[a] = (uint32_t)i;
    if (i + 1 == s || skeys[i + 1] != a) run_end[a] = (uint32_t)(i + 1);
  }
}
// read_ts[pos] = rank of pos among the positions of its address (densified.rs:44-50), written as Fr
__global__ void __launch_bounds__(256) k_densify_read(const uint32_t* __restrict__ skeys, const uint32_t* __restrict__ svals, size_t s, const uint32_t* __restrict__ run_start, uint32_t world,
                                                       uint32_t rank, fr_t* __restrict__ read_fr) {
  const fr29 r2s = fr29_r2s();
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < s; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t pos = svals[i];
    if (pos % world != rank) continue;
    const uint32_t ts = (uint32_t)i - run_start[skeys[i]];
    read_fr[pos / world] = fr29_store(fr29_mul(fr29_from_u64_int(ts), r2s));
  }
}
__global__ void __launch_bounds__(256) k_densify_final(const uint32_t* __restrict__ run_start, const uint32_t* __restrict__ run_end, size_t m, uint32_t world, uint32_t rank, fr_t* __restrict__ final_fr) {
  const fr29 r2s = fr29_r2s();
  for (size_t a = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * world + rank; a < m; a += (size_t)gridDim.x * blockDim.x * world)
    final_fr[a / world] = fr29_store(fr29_mul(fr29_from_u64_int(run_end[a] - run_start[a]), r2s));
}

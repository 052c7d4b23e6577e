// Where does one MSM launch of a bullet round spend its time?  Workgroup (0,0) stamps the 100 MHz wall clock at the phase boundaries of
// k_msm_buckets / k_points_sum (MSM_PHASE_CLOCK); the launch shapes are run_msm's for 2 rows of n + 2 full-width scalars, half of them zero
// (the L / R rows of bullet.rs:98-121 over the original generators).  Arithmetic is data-independent, so the table holds arbitrary limbs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/msm_phase_bench tools/msm_phase_bench.hip
#define MSM_PHASE_CLOCK 1
#ifndef MSM_BENCH_WB
#define MSM_BENCH_WB 4   // -DMSM_BENCH_WB=8: the byte-multiple table (32 windows x 128 multiples)
#endif
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../lasso_amd/csrc/poly_kernels.cuh"
#include "../lasso_amd/csrc/msm_kernels.cuh"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static uint64_t sm_state = 0x4C4153534Full;
static uint64_t splitmix() { uint64_t z = (sm_state += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 4096;
  const size_t rows = 2, n_cols = n + 2, W = MSM_WINDOWS;
  for (size_t K : {(size_t)128, (size_t)64, (size_t)32, (size_t)256}) {
    const size_t cols_per_chunk = (n_cols + K - 1) / K, Kc = (n_cols + cols_per_chunk - 1) / cols_per_chunk;
    std::vector<uint64_t> sc(rows * n_cols * 4, 0);
    for (size_t r = 0; r < rows; r++) for (size_t j = 0; j < n_cols; j++) {
      const bool zero = j < n && (((j % n) >= n / 2) == (r == 0));   // half of every row is zero (row 0: the second half, so workgroup (0,0) has work)
      if (!zero) { for (int k = 0; k < 4; k++) sc[(r * n_cols + j) * 4 + k] = splitmix(); sc[(r * n_cols + j) * 4 + 3] &= 0x0fffffffffffffffull; }
    }
    std::vector<uint32_t> tab(n_cols * W * sizeof(niels29) / 4);
    for (auto& x : tab) x = (uint32_t)splitmix() & 0x0fffffff;
    uint8_t* d_sc; niels29* d_tab; pt29* d_part; ed_point* d_out; uint32_t* d_cnt;
    CK(hipMalloc(&d_sc, sc.size() * 8)); CK(hipMalloc(&d_tab, tab.size() * 4)); CK(hipMalloc(&d_part, rows * Kc * sizeof(pt29))); CK(hipMalloc(&d_out, rows * sizeof(ed_point))); CK(hipMalloc(&d_cnt, 64)); CK(hipMemset(d_cnt, 0, 64));
    CK(hipMemcpy(d_sc, sc.data(), sc.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    float best1 = 1e9f, best2 = 1e9f; uint64_t clk[32] = {0};
    for (int it = 0; it < 6; it++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_msm_buckets, dim3((unsigned)Kc, (unsigned)rows), dim3(MSM_THREADS), 0, 0, (const uint8_t*)d_sc, 32u, (uint32_t)W, n_cols * 32, n_cols, cols_per_chunk, (const niels29*)d_tab, n_cols, d_part, (uint32_t*)nullptr);
      CK(hipEventRecord(e1));
      hipLaunchKernelGGL(k_points_sum, dim3((unsigned)rows), dim3(MSM_THREADS), 0, 0, (const pt29*)d_part, (uint32_t)Kc, d_out, (uint32_t*)nullptr, d_cnt, (uint32_t*)nullptr, 0u);
      CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
      float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
      if (it && a < best1) { best1 = a; CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(msm_phase_clock), sizeof(clk))); }
      if (it && b < best2) best2 = b;
    }
    auto us = [&](int a, int b) { return (double)(clk[b] - clk[a]) * 0.01; };
    printf("n=%zu K=%zu (cols/chunk %zu): k_msm_buckets %.1f us, k_points_sum %.1f us (events)\n", n, Kc, cols_per_chunk, best1 * 1e3, best2 * 1e3);
    printf("   workgroup (0,0): sort %.1f | accumulate %.1f | segmented tree %.1f | bit planes %.1f | horner %.1f us;  points_sum: tree %.1f | convert %.1f us;  end of buckets -> start of points_sum %.1f us\n",
           us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(8, 9), us(9, 10), us(5, 8));
    CK(hipFree(d_sc)); CK(hipFree(d_tab)); CK(hipFree(d_part)); CK(hipFree(d_out)); CK(hipFree(d_cnt));
  }
  // ---- the latency-shaped kernel on the same round: compact rows of n/2 + 2 scalars, signed digit multiples table
  {
    const size_t row = n / 2 + 2, tn = n + 2;
    std::vector<uint64_t> sc(rows * row * 4);
    for (size_t i = 0; i < sc.size(); i += 4) { for (int k = 0; k < 4; k++) sc[i + k] = splitmix(); sc[i + 3] &= 0x0fffffffffffffffull; }
    std::vector<uint32_t> tab(tn * (size_t)MsmD<MSM_BENCH_WB>::WINDOWS * MsmD<MSM_BENCH_WB>::MULTS * sizeof(niels29) / 4);
    for (auto& x : tab) x = (uint32_t)splitmix() & 0x0fffffff;
    uint32_t* d_sc; niels29* d_tab; pt29* d_part; ed_point* d_out; uint32_t* d_cnt;
    CK(hipMalloc(&d_sc, sc.size() * 8)); CK(hipMalloc(&d_tab, tab.size() * 4)); CK(hipMalloc(&d_part, rows * 4096 * sizeof(pt29))); CK(hipMalloc(&d_out, rows * sizeof(ed_point))); CK(hipMalloc(&d_cnt, 256)); CK(hipMemset(d_cnt, 0, 256));
    CK(hipMemcpy(d_sc, sc.data(), sc.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    const MsmColMap cm = {(uint32_t)n, (uint32_t)(n / 2), (uint32_t)(n / 2), (uint32_t)n};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t total = (uint32_t)(row * MsmD<MSM_BENCH_WB>::WINDOWS);
    for (uint32_t ipc : {256u, 512u, 768u, 1024u, 1280u, 2048u}) {
      const uint32_t K = (total + ipc - 1) / ipc;
      float best = 1e9f; uint64_t clk[32] = {0};
      for (int it = 0; it < 8; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_msm_direct<0, MSM_BENCH_WB>), dim3(K, (unsigned)rows), dim3(MSM_THREADS), 0, 0, (const uint32_t*)d_sc, row * 8, (uint32_t)row, ipc, cm, (const niels29*)d_tab, tn, d_part, d_out, d_cnt, (uint32_t*)nullptr, 0u, fr_zero(), fr_zero(), fr_zero(), (uint32_t*)nullptr, 1u, 0u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float a; CK(hipEventElapsedTime(&a, e0, e1));
        if (it && a < best) { best = a; CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(msm_phase_clock), sizeof(clk))); }
      }
      CK(hipGetLastError());
      auto us = [&](int a, int b) { return (double)(clk[b] - clk[a]) * 0.01; };
      printf("n=%zu k_msm_direct items/chunk=%u K=%u x %zu rows: %.1f us (events);  workgroup (0,0): stage scalars %.1f | accumulate %.1f | workgroup tree %.1f us\n", n, ipc, K, rows, best * 1e3, us(0, 1), us(1, 2), us(2, 3));
    }
    CK(hipFree(d_sc)); CK(hipFree(d_tab)); CK(hipFree(d_part)); CK(hipFree(d_out)); CK(hipFree(d_cnt));
  }
  return 0;
}

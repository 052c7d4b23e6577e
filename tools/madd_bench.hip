// The mixed addition's real ceiling (`frac_microbench` > 1 — the round-2 microbenchmark, tools/microbench.hip k_tp_ptmadd, has no __launch_bounds__, so the
// compiler caps it at 128 VGPRs and SPILLS the point to scratch: 52 scratch instructions per addition in its loop; the product kernels, bounded at 256 threads, do not spill).
// Section A: a chain of pt_madd with the table entry in registers, compiled for 1 / 2 / 3 / 4 waves per SIMD (VGPR budgets 512 / 256 / 168 / 128), run at that occupancy and below.
// Section B: field products alone (fe_mul chains) at the same occupancies — the issue rate of the product's instruction mix.
// Section C: the commitment kernels themselves (k_msm_rows8w / k_msm_rows8) on random one-byte scalars at the headline's E shape (4096 x 4096) and at configs[2]'s.
// Section D: the one-wave-per-row commitment with the table fetches made cheap (does the Infinity Cache bound it, or the instruction stream?).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DLASSO_BN254] -o tools/madd_bench tools/madd_bench.hip     Run: tools/madd_bench [sections, e.g. A,B,C]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../lasso_amd/csrc/poly_kernels.cuh"
#include "../lasso_amd/csrc/msm_kernels.cuh"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int MINW>
__global__ void __launch_bounds__(256, MINW) k_chain_madd(pt29* io, const niels29* nb, int iters) {
  pt29 p = io[threadIdx.x & 63]; const niels29 n0 = nb[threadIdx.x & 63];
  for (int i = 0; i < iters; i++) p = pt_madd(p, n0);
  if (p.X.v[0] == 0x12345678 && p.Y.v[1] == 0x1abcdef0) io[blockIdx.x] = p;
}
// the same chain with a FRESH table entry per addition (64 entries, L1-resident, next entry in flight during the addition): what a commitment kernel's loop looks like to the compiler
template <int MINW>
__global__ void __launch_bounds__(256, MINW) k_chain_madd_var(pt29* io, const niels29* nb, int iters) {
  pt29 p = io[threadIdx.x & 63]; niels29 cur = nb[threadIdx.x & 63];
  for (int i = 0; i < iters; i++) { const niels29 nxt = nb[(threadIdx.x + 7 * i + 7) & 63]; p = pt_madd(p, cur); cur = nxt; }
  if (p.X.v[0] == 0x12345678 && p.Y.v[1] == 0x1abcdef0) io[blockIdx.x] = p;
}
template <int MINW>
__global__ void __launch_bounds__(256, MINW) k_chain_femul(fe29* io, int iters) {
  fe29 a = io[threadIdx.x & 63], b = io[64 + (threadIdx.x & 63)], c = fe_weak(fe_add(a, b)), d = fe_weak(fe_sub(a, b));
  for (int i = 0; i < iters; i++) { a = fe_mul(a, b); b = fe_mul(b, c); c = fe_mul(c, d); d = fe_mul(d, a); }
  const fe29 r = fe_add(fe_add(a, b), fe_add(c, d));
  if (r.v[0] == 0x12345678 && r.v[1] == 0x1abcdef0) io[blockIdx.x] = r;
}
// one dependent chain of products per lane (the latency of ONE product when nothing else is in flight: what a tree level pays)
template <int MINW>
__global__ void __launch_bounds__(256, MINW) k_chain_femul1(fe29* io, int iters) {
  fe29 a = io[threadIdx.x & 63]; const fe29 b = io[64 + (threadIdx.x & 63)];
  for (int i = 0; i < iters; i++) a = fe_mul(a, b);
  if (a.v[0] == 0x12345678 && a.v[1] == 0x1abcdef0) io[blockIdx.x] = a;
}

template <class K>
static double time_kernel(K launch, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int i = 0; i < reps; i++) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
  CK(hipGetLastError());
  return best;
}
static uint64_t sm_state = 0x4C4153534Full;
static uint64_t splitmix() { uint64_t z = (sm_state += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static bool want(int argc, char** argv, char sec) { if (argc < 2) return true; return strchr(argv[1], sec) != nullptr; }

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int CU = prop.multiProcessorCount;
#ifdef LASSO_BN254
  const char* curve = "bn254";
#else
  const char* curve = "curve25519";
#endif
  printf("device: %s  CUs=%d  curve=%s\n", prop.name, CU, curve);
  std::vector<uint32_t> raw(65536 * 40);
  for (auto& x : raw) x = (uint32_t)splitmix() & 0x0fffffff;
  uint32_t* d; CK(hipMalloc(&d, raw.size() * 4)); CK(hipMemcpy(d, raw.data(), raw.size() * 4, hipMemcpyHostToDevice));
  uint32_t* d2; CK(hipMalloc(&d2, raw.size() * 4)); CK(hipMemcpy(d2, raw.data(), raw.size() * 4, hipMemcpyHostToDevice));
  double best_madd = 0; int best_cfg[2] = {0, 0};
  if (want(argc, argv, 'A')) {
    printf("\n== A. pt_madd chain, table entry in registers: compiled for MINW waves per SIMD, run at `wpc` workgroups of 256 threads per CU (= waves per SIMD)\n");
    const int iters = 256;
#define RUN_A(MINW) for (int wpc = 1; wpc <= MINW; wpc++) { const int blocks = CU * wpc; \
      const double ms = time_kernel([&] { hipLaunchKernelGGL((k_chain_madd<MINW>), dim3(blocks), dim3(256), 0, 0, (pt29*)d, (const niels29*)d2, iters); }); \
      const double g = (double)blocks * 256 * iters / (ms * 1e-3) * 1e-9; \
      printf("  compiled for %d waves/SIMD, run at %d: %8.3f ms  %7.2f G madd/s   (%.2f us per wave-addition per SIMD)\n", MINW, wpc, ms, g, ms * 1e3 / iters / wpc); \
      if (g > best_madd) { best_madd = g; best_cfg[0] = MINW; best_cfg[1] = wpc; } }
    RUN_A(1) RUN_A(2) RUN_A(3) RUN_A(4)
    printf("   -- the same with a fresh (L1-resident) table entry per addition:\n");
#define RUN_AV(MINW) for (int wpc = 1; wpc <= MINW; wpc++) { const int blocks = CU * wpc; \
      const double ms = time_kernel([&] { hipLaunchKernelGGL((k_chain_madd_var<MINW>), dim3(blocks), dim3(256), 0, 0, (pt29*)d, (const niels29*)d2, iters); }); \
      printf("  fresh entry, compiled for %d waves/SIMD, run at %d: %8.3f ms  %7.2f G madd/s\n", MINW, wpc, ms, (double)blocks * 256 * iters / (ms * 1e-3) * 1e-9); }
    RUN_AV(1) RUN_AV(2) RUN_AV(3)
    printf("MADD_CEILING {\"curve\": \"%s\", \"G_madd_per_s\": %.2f, \"compiled_for_waves_per_simd\": %d, \"run_at_waves_per_simd\": %d, \"device\": \"%s\", \"CUs\": %d}\n", curve, best_madd, best_cfg[0], best_cfg[1], prop.name, CU);
  }
  if (want(argc, argv, 'B')) {
    printf("\n== B. fe_mul: four independent chains per lane (throughput) and ONE chain per lane (latency of a product)\n");
    const int iters = 512;
#define RUN_B(MINW) for (int wpc = 1; wpc <= MINW; wpc++) { const int blocks = CU * wpc; \
      const double ms = time_kernel([&] { hipLaunchKernelGGL((k_chain_femul<MINW>), dim3(blocks), dim3(256), 0, 0, (fe29*)d, iters); }); \
      const double ms1 = time_kernel([&] { hipLaunchKernelGGL((k_chain_femul1<MINW>), dim3(blocks), dim3(256), 0, 0, (fe29*)d, iters); }); \
      printf("  compiled for %d waves/SIMD, run at %d: 4 chains %7.1f G products/s (%.3f us per wave-product per SIMD) | 1 chain %7.1f G/s (%.3f us per dependent product)\n", MINW, wpc, \
             (double)blocks * 256 * iters * 4 / (ms * 1e-3) * 1e-9, ms * 1e3 / (iters * 4) / wpc, (double)blocks * 256 * iters / (ms1 * 1e-3) * 1e-9, ms1 * 1e3 / iters); }
    RUN_B(1) RUN_B(2) RUN_B(4)
  }
  if (want(argc, argv, 'C')) {
    printf("\n== C. the commitment kernels on random one-byte scalars (every byte non-zero with probability 255/256): G mixed additions per second\n");
    struct Shape { size_t rows, cols; const char* what; };
    for (const Shape& sh : {Shape{4096, 4096, "headline E (AND C=1 2^24)"}, Shape{8192, 16384, "configs[2] E (XOR C=8 2^24: 2^27 values)"}, Shape{1024, 1024, "configs[1] shape (2^20)"}}) {
      const size_t n = sh.cols;
      std::vector<uint32_t> tab((size_t)MSM8_MULTS * n * (sizeof(niels29) / 4));
      for (auto& x : tab) x = (uint32_t)splitmix() & 0x0fffffff;
      niels29* d_tab; CK(hipMalloc(&d_tab, tab.size() * 4)); CK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
      std::vector<uint32_t> sc(sh.rows * sh.cols);
      for (auto& x : sc) x = (uint32_t)splitmix() & 255u;
      uint32_t* d_sc; CK(hipMalloc(&d_sc, sc.size() * 4)); CK(hipMemcpy(d_sc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
      pt29* d_out; CK(hipMalloc(&d_out, sh.rows * sizeof(pt29)));
      double nz = 0; for (auto x : sc) nz += x != 0;
      double msw = 0, msw1 = 0;   // rows per wave as the library chooses them (<= 2048 waves), and one row per wave (round 5)
      if (sh.cols <= 8192) {
        const uint32_t rpw = (uint32_t)((sh.rows + 2047) / 2048), waves = (uint32_t)((sh.rows + rpw - 1) / rpw);
        msw = time_kernel([&] { hipLaunchKernelGGL(k_msm_rows8w, dim3((waves + 3) / 4), dim3(MSM_THREADS), 0, 0, (const uint32_t*)d_sc, sh.cols, (uint32_t)sh.cols, 1u, (const niels29*)d_tab, (const niels29*)d_tab, n, d_out, (uint32_t)sh.rows, (uint32_t*)nullptr, rpw); }, 3);
        msw1 = time_kernel([&] { hipLaunchKernelGGL(k_msm_rows8w, dim3((unsigned)(sh.rows / 4)), dim3(MSM_THREADS), 0, 0, (const uint32_t*)d_sc, sh.cols, (uint32_t)sh.cols, 1u, (const niels29*)d_tab, (const niels29*)d_tab, n, d_out, (uint32_t)sh.rows, (uint32_t*)nullptr, 1u); }, 3);
      }
      const double ms8 = time_kernel([&] { hipLaunchKernelGGL(k_msm_rows8, dim3(1, (unsigned)sh.rows), dim3(MSM_THREADS), 0, 0, (const uint32_t*)d_sc, sh.cols, (uint32_t)sh.cols, (uint32_t)sh.cols, 1u, (const niels29*)d_tab, (const niels29*)d_tab, n, d_out, (uint32_t*)nullptr); }, 3);
      printf("  %-44s %5zu x %5zu: k_msm_rows8w %8.3f ms %6.2f G/s (one row per wave %8.3f ms %6.2f G/s) | k_msm_rows8 %8.3f ms %6.2f G/s\n", sh.what, sh.rows, sh.cols, msw, msw > 0 ? nz / (msw * 1e-3) * 1e-9 : 0.0,
             msw1, msw1 > 0 ? nz / (msw1 * 1e-3) * 1e-9 : 0.0, ms8, nz / (ms8 * 1e-3) * 1e-9);
      CK(hipFree(d_tab)); CK(hipFree(d_sc)); CK(hipFree(d_out));
    }
  }
  if (want(argc, argv, 'D')) {
    printf("\n== D. k_msm_rows8w at the headline's E shape with the table fetches made cheap: every scalar the SAME byte (one 512 KB run of the table, L2-resident) against random bytes (134 MB, Infinity Cache)\n");
    const size_t rows = 4096, cols = 4096, n = cols;
    std::vector<uint32_t> tab((size_t)MSM8_MULTS * n * (sizeof(niels29) / 4));
    for (auto& x : tab) x = (uint32_t)splitmix() & 0x0fffffff;
    niels29* d_tab; CK(hipMalloc(&d_tab, tab.size() * 4)); CK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    uint32_t* d_sc; CK(hipMalloc(&d_sc, rows * cols * 4));
    pt29* d_out; CK(hipMalloc(&d_out, rows * sizeof(pt29)));
    for (int mode = 0; mode < 3; mode++) {
      std::vector<uint32_t> sc(rows * cols);
      for (auto& x : sc) x = mode == 0 ? ((uint32_t)splitmix() % 255u) + 1u : (mode == 1 ? 7u : ((uint32_t)splitmix() % 8u) + 1u);
      CK(hipMemcpy(d_sc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
      const double ms = time_kernel([&] { hipLaunchKernelGGL(k_msm_rows8w, dim3((unsigned)(rows / 4)), dim3(MSM_THREADS), 0, 0, (const uint32_t*)d_sc, cols, (uint32_t)cols, 1u, (const niels29*)d_tab, (const niels29*)d_tab, n, d_out, (uint32_t)rows, (uint32_t*)nullptr, 1u); }, 5);
      printf("  %-62s %8.3f ms  %6.2f G additions/s\n", mode == 0 ? "random non-zero bytes (255 x 4096 entries = 134 MB)" : (mode == 1 ? "every byte = 7 (4096 entries = 512 KB)" : "bytes 1..8 (8 x 4096 entries = 4 MB)"), ms, (double)rows * cols / (ms * 1e-3) * 1e-9);
    }
  }
  return 0;
}

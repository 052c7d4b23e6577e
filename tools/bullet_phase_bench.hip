// Phase-stamped timeline of ONE k_bullet_msm launch: every workgroup stamps the 100 MHz wall clock at its phase boundaries (MSM_PHASE_LOG in
// msm_kernels.cuh), so the launch's critical path — start -> scalars staged -> accumulated -> workgroup tree (per level) -> ticket -> [last workgroup of the row] partials
// loaded -> cross-workgroup tree (per level) -> converted + flag — is read off the workgroup that arrived last.  Shapes are bullet_round_fused's (lasso_hip.hip) for a
// generator vector of n points: first folding round (nk = n / 2) and a late one (nk = 2).  Arithmetic is data-independent; the table holds arbitrary limbs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DLASSO_BN254] -o tools/bullet_phase_bench tools/bullet_phase_bench.hip        Run: tools/bullet_phase_bench [n] [wgs]
#define MSM_PHASE_LOG 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../lasso_amd/csrc/poly_kernels.cuh"
#include "../lasso_amd/csrc/msm_kernels.cuh"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
#ifndef BENCH_WB
#define BENCH_WB 8
#endif

static uint64_t sm_state = 0x4C4153534Full;
static uint64_t splitmix() { uint64_t z = (sm_state += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 4096;
  const size_t wgs = argc > 2 ? (size_t)atol(argv[2]) : 256;
  const bool tagged = !(argc > 3 && atoi(argv[3]) == 0);   // third argument 0: the flag protocol of round 5 (conversion by one lane, system fence, ticket, flag)
  typedef MsmD<BENCH_WB> D;
  const size_t tn = n + 2, windows = D::WINDOWS;
  // table: (n + 2) x windows x multiples entries of arbitrary small limbs
  const size_t tab_entries = tn * windows * D::MULTS;
  std::vector<uint32_t> tab(tab_entries * (sizeof(niels29) / 4));
  for (auto& x : tab) x = (uint32_t)splitmix() & 0x0fffffff;
  niels29* d_tab; CK(hipMalloc(&d_tab, tab.size() * 4)); CK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  std::vector<uint64_t> fr(4 * (2 * n + n + 8));
  for (size_t i = 0; i < fr.size(); i += 4) { for (int k = 0; k < 4; k++) fr[i + k] = splitmix(); fr[i + 3] &= 0x0fffffffffffffffull; }
  fr_t *d_a, *d_b, *d_w, *d_ao, *d_bo, *d_wo; pt29* d_part; ed_point* d_out; uint32_t *d_cnt, *d_flag, *d_gmail;
  CK(hipMalloc(&d_a, n * 32)); CK(hipMalloc(&d_b, n * 32)); CK(hipMalloc(&d_w, n * 32)); CK(hipMalloc(&d_ao, n * 32)); CK(hipMalloc(&d_bo, n * 32)); CK(hipMalloc(&d_wo, n * 32));
  CK(hipMemcpy(d_a, fr.data(), n * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(d_b, fr.data() + 4 * n, n * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, fr.data() + 8 * n, n * 32, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_part, 2 * 1024 * sizeof(pt29))); CK(hipMalloc(&d_out, 2 * sizeof(ed_point) + 8 * 48)); /* + room for the tagged form: 8 elements of 48 bytes */ CK(hipMalloc(&d_cnt, 256)); CK(hipMemset(d_cnt, 0, 256)); CK(hipMalloc(&d_flag, 64)); CK(hipMalloc(&d_gmail, 256));
  fr_t u, ui, bl, br; memcpy(u.v, fr.data() + 12 * n, 32); memcpy(ui.v, fr.data() + 12 * n + 4, 32); memcpy(bl.v, fr.data() + 12 * n + 8, 32); memcpy(br.v, fr.data() + 12 * n + 12, 32);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  void* d_log; CK(hipGetSymbolAddress(&d_log, HIP_SYMBOL(msm_phase_log)));
#ifdef LASSO_BN254
  const char* curve = "bn254";
#else
  const char* curve = "curve25519";
#endif
  for (size_t nk : {n / 2, (size_t)64, (size_t)2}) {
    if (2 * nk > n) continue;
    // bullet_round_fused's launch shape
    const size_t cols = n / 2, total = cols * windows, kmax = (wgs - 2) / 2;
    size_t ipc_ = (total + kmax - 1) / kmax; ipc_ = (ipc_ + windows - 1) / windows * windows; if (ipc_ < 256) ipc_ = 256; if (ipc_ > windows * 128) ipc_ = windows * 128;
    const uint32_t ipc = (uint32_t)ipc_; const size_t K = (total + ipc_ - 1) / ipc_;
    const size_t nwg = 2 * (K + 1);
    float best = 1e9f; std::vector<uint64_t> log(nwg * 32), cur(nwg * 32);
    for (int it = 0; it < 8; it++) {
      CK(hipMemsetAsync(d_log, 0, sizeof(uint64_t) * nwg * 32, 0));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_bullet_msm<true, BENCH_WB>), dim3((unsigned)K + 1, 2), dim3(MSM_THREADS), 0, 0, (const fr_t*)d_a, (const fr_t*)d_b, (const fr_t*)d_w, d_ao, d_bo, d_wo, (uint32_t)nk, (uint32_t)n, u, ui, bl, br,
                         ipc, (const niels29*)d_tab, tn, d_part, d_out, d_cnt, tagged ? LASSO_TAGGED : d_flag, (uint32_t)(it + 1), (uint32_t*)nullptr, 1u, 0u, (const uint32_t*)nullptr, d_gmail);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpyFromSymbol(cur.data(), HIP_SYMBOL(msm_phase_log), sizeof(uint64_t) * nwg * 32));
      if (it && ms < best) { best = ms; log = cur; }
    }
    CK(hipGetLastError());
    // ---- read the timeline
    uint64_t t0 = ~0ull, t_end = 0;
    for (size_t w = 0; w < nwg; w++) { if (log[w * 32 + 0] && log[w * 32 + 0] < t0) t0 = log[w * 32 + 0]; for (int k = 0; k < 32; k++) if (log[w * 32 + k] > t_end) t_end = log[w * 32 + k]; }
    auto us = [&](uint64_t x) { return x ? (double)(x - t0) * 0.01 : -1.0; };
    printf("\n#### %s  n=%zu nk=%zu  WB=%d  grid=(%zu+1) x 2 = %zu workgroups, %u items per chunk  — launch %.1f us (events, best of 7); first stamp -> last stamp %.1f us\n", curve, n, nk, BENCH_WB, K, nwg, ipc,
           best * 1e3, us(t_end));
    auto stat = [&](int k, const char* name) {
      std::vector<double> v; for (size_t w = 0; w < nwg; w++) if (log[w * 32 + k]) v.push_back(us(log[w * 32 + k]));
      if (v.empty()) return; std::sort(v.begin(), v.end());
      printf("  all workgroups: %-34s min %6.2f  median %6.2f  max %6.2f us   (%zu workgroups)\n", name, v.front(), v[v.size() / 2], v.back(), v.size());
    };
    stat(0, "start (after the challenge read)"); stat(1, "scalars staged"); stat(2, "accumulated"); stat(3, "workgroup tree done"); stat(11, "partial written + ticket");
    for (int row = 0; row < 2; row++) {
      size_t last = nwg; for (size_t w = row * (K + 1); w < (row + 1) * (K + 1); w++) if (log[w * 32 + 12]) last = w;
      if (last == nwg) { printf("  row %d: no last workgroup found\n", row); continue; }
      const uint64_t* L = &log[last * 32];
      printf("  row %d critical path = workgroup x=%zu (the last to take its ticket):\n", row, last - row * (K + 1));
      double prev = us(L[0]);
      auto step = [&](int k, const char* name) { if (!L[k]) return; const double t = us(L[k]); printf("     %-52s at %6.2f us   (+%5.2f)\n", name, t, t - prev); prev = t; };
      printf("     %-52s at %6.2f us\n", "start", prev);
      step(1, "scalars staged (fold + recode into LDS)"); step(2, "accumulated (mixed additions, table fetches)"); step(13, "sums -> LDS, last partly filled pass four lanes per addition");
      for (int l = 7; l >= 0; l--) { char nm[64]; snprintf(nm, sizeof nm, "workgroup tree: level of %d additions", 1 << l); step(16 + l, nm); }
      step(3, "workgroup tree done"); step(11, "partial written, release fence, ticket"); step(12, "last of the row: acquire, partials -> LDS");
      for (int l = 7; l >= 0; l--) { char nm[64]; snprintf(nm, sizeof nm, "cross-workgroup tree: level of %d additions", 1 << l); step(24 + l, nm); }
      step(4, "cross-workgroup tree done"); step(5, tagged ? "four lanes convert, tagged chunks to the host-visible area" : "converted to ark limbs, system fence, flag");
    }
  }
  return 0;
}

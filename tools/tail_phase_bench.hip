// Where does one TURN of the resident sumcheck tail (k_cubic_tail) spend its time?  Workgroup 0 stamps the 100 MHz wall clock at the phase boundaries of each turn
// (TAIL_PHASE_CLOCK); the host side here answers every result at once with a fixed challenge (no transcript), so "wait" is the hand-off's round trip alone.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/tail_phase_bench tools/tail_phase_bench.hip
#define TAIL_PHASE_CLOCK 1
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <emmintrin.h>
#include "../lasso_amd/csrc/poly_kernels.cuh"
#include "experiments/cubic_tail_ahead.cuh"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool element_there(const uint32_t* e, uint32_t seq) {
  uint32_t c[12];
  for (int k = 0; k < 3; k++) _mm_storeu_si128((__m128i*)(c + 4 * k), _mm_load_si128((const __m128i*)(e + 4 * k)));
  return c[0] == seq && c[4] == seq && c[8] == seq;
}
int main(int argc, char** argv) {
  const uint32_t q = argc > 1 ? (uint32_t)atol(argv[1]) : 512, ncirc = 2; const bool ahead = argc > 2 && atol(argv[2]) != 0;
  const uint32_t m = 2 * q;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<fr_t> host(m); for (uint32_t i = 0; i < m; i++) { memset(&host[i], 0, sizeof(fr_t)); host[i].v[0] = 1000 + i; host[i].v[3] = 77 * i + 5; }
  MutPtrTable A, B; fr_t* E;
  for (uint32_t c = 0; c < ncirc; c++) { CK(hipMalloc(&A.p[c], m * sizeof(fr_t))); CK(hipMalloc(&B.p[c], m * sizeof(fr_t))); }
  CK(hipMalloc(&E, m * sizeof(fr_t))); CK(hipMemcpy(E, host.data(), m * sizeof(fr_t), hipMemcpyHostToDevice));
  uint32_t *h_tag, *d_tag, *h_mail, *d_mail, *d_cnt;
  CK(hipHostMalloc(&h_tag, 4096, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostGetDevicePointer((void**)&d_tag, h_tag, 0)); memset(h_tag, 0, 4096);
  CK(hipHostMalloc(&h_mail, 256, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostGetDevicePointer((void**)&d_mail, h_mail, 0)); memset(h_mail, 0, 256);
  CK(hipMalloc(&d_cnt, (LASSO_MAX_PTRS + 40) * 4)); CK(hipMemset(d_cnt, 0, (LASSO_MAX_PTRS + 40) * 4));
  uint32_t turns = 0; for (uint32_t x = q; x; x >>= 1) turns++;   // log2(q) + 1 sums turns, then the heads
  uint32_t seq = 0; double best = 1e18; uint64_t clk[16 * 8] = {0}; std::vector<double> host_turn(turns + 1), best_turn(turns + 1);
  for (int rep = 0; rep < 12; rep++) {
    for (uint32_t c = 0; c < ncirc; c++) { CK(hipMemcpy(A.p[c], host.data(), m * sizeof(fr_t), hipMemcpyHostToDevice)); CK(hipMemcpy(B.p[c], host.data(), m * sizeof(fr_t), hipMemcpyHostToDevice)); }
    CK(hipDeviceSynchronize());
    const uint32_t seq0 = seq + 1; seq += turns + 1;
    const double t0 = now();
    if (ahead) hipLaunchKernelGGL((k_cubic_tail_ahead<false, 512, false>), dim3(ncirc), dim3(512), 0, s, A, B, (const fr_t*)E, q, fr_zero(), (const uint32_t*)d_mail, d_cnt, (fr_t*)d_tag, LASSO_TAGGED, seq0, EqInline());
    else hipLaunchKernelGGL((k_cubic_tail<false, 512, false>), dim3(ncirc), dim3(512), 0, s, A, B, (const fr_t*)E, q, fr_zero(), (const uint32_t*)d_mail, d_cnt, (fr_t*)d_tag, LASSO_TAGGED, seq0, EqInline());
    double tp = t0;
    for (uint32_t turn = 0; turn <= turns; turn++) {
      const uint32_t want = seq0 + turn; const uint32_t cnt = 2 * ncirc;
      for (uint32_t e = 0; e < cnt; e++) { uint64_t spins = 0; while (!element_there(h_tag + 12 * e, want)) { if ((++spins & 0xfffff) == 0 && now() - t0 > 3e6) { printf("timeout at turn %u\n", turn); return 2; } } }
      const double tn = now(); host_turn[turn] = tn - tp; tp = tn;
      if (turn < turns) {
        const uint32_t tag = want + 1;
        const uint32_t chk = (1u ^ 2u ^ 3u ^ 4u ^ 5u ^ 6u ^ 7u ^ 8u) + tag * 0x9E3779B9u;
        const __m128i c0 = _mm_set_epi32(3, 2, 1, (int)tag), c1 = _mm_set_epi32(6, 5, 4, (int)tag), c2 = _mm_set_epi32((int)chk, 8, 7, (int)tag);
        _mm_store_si128((__m128i*)(h_mail + 0), c0); _mm_store_si128((__m128i*)(h_mail + 4), c1); _mm_store_si128((__m128i*)(h_mail + 8), c2);
      }
    }
    const double total = now() - t0;
    CK(hipStreamSynchronize(s));
    if (rep >= 2 && total < best) { best = total; best_turn = host_turn; CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(tail_phase_clock), sizeof(clk))); }
  }
  printf("%s: q = %u, %u circuits: %.1f us for %u turns + heads (%.2f us per hand-off, launch included)\n", ahead ? "k_cubic_tail_ahead (phases: - | G + terms | column sums + coefficients | wait | evaluate + publish | bind)" : "k_cubic_tail", q, ncirc, best, turns, best / (turns + 1));
  printf("turn  pairs | terms  reduce  pack+publish  wait-for-host  bind | device turn   host-side turn (us)\n");
  for (uint32_t turn = 0; turn < turns && turn < 16; turn++) {
    const uint64_t* c = clk + turn * 8; auto us = [&](int a, int b) { return (double)(c[b] - c[a]) * 0.01; };
    printf("%4u  %5u | %5.2f  %6.2f  %12.2f  %13.2f  %4.2f | %11.2f   %8.2f\n", turn, q >> turn, us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(0, 5), best_turn[turn + 1]);
  }
  return 0;
}

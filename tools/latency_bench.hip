// Round-trip latency options for one sumcheck round (kernel -> few hundred bytes -> host), MI355X.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/latency_bench tools/latency_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_small(uint32_t* out, uint32_t v) { if (threadIdx.x < 24) out[threadIdx.x] = v + threadIdx.x; }
__global__ void k_flag(volatile uint32_t* out, volatile uint32_t* flag, uint32_t v) {
  if (threadIdx.x < 24) out[threadIdx.x] = v + threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence_system(); *flag = v; }
}
struct Big { const void* p[272]; };   // 2176 bytes of kernel arguments: the two 136-pointer tables of a cubic round
__global__ void k_flag_big(Big b, volatile uint32_t* out, volatile uint32_t* flag, uint32_t v) {
  if (threadIdx.x < 24) out[threadIdx.x] = v + threadIdx.x + (uint32_t)(uintptr_t)b.p[threadIdx.x & 1];
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence_system(); *flag = v; }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint32_t *d, *h, *hm, *hf; CK(hipMalloc(&d, 256)); CK(hipHostMalloc(&h, 256)); CK(hipHostMalloc(&hm, 256, hipHostMallocMapped)); CK(hipHostMalloc(&hf, 64, hipHostMallocMapped));
  uint32_t *dm, *df; CK(hipHostGetDevicePointer((void**)&dm, hm, 0)); CK(hipHostGetDevicePointer((void**)&df, hf, 0));
  const int N = 2000;
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, d, i); CK(hipStreamSynchronize(s)); }
    double t1 = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, d, i); CK(hipMemcpyAsync(h, d, 96, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
    double t2 = now();
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, dm, i); CK(hipStreamSynchronize(s)); if (hm[5] != (uint32_t)i + 5) { printf("mapped mismatch\n"); return 1; } }
    double t3 = now();
    *hf = 0xffffffffu;
    for (int i = 0; i < N; i++) {
      hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, dm, df, (uint32_t)i);
      long spins = 0; while (*(volatile uint32_t*)hf != (uint32_t)i) { if (++spins > 200000000L) { printf("flag timeout\n"); return 2; } }
      if (hm[5] != (uint32_t)i + 5) { printf("flag data mismatch\n"); return 1; }
    }
    CK(hipStreamSynchronize(s));
    double t4 = now();
    // two dependent launches then sync (eval kernel + reduce kernel)
    for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, d, i); hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, d, i); CK(hipMemcpyAsync(h, d, 96, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
    double t5 = now();
    Big big; for (int i = 0; i < 272; i++) big.p[i] = nullptr;
    *hf = 0xffffffffu;
    double t6 = now();
    for (int i = 0; i < N; i++) {
      hipLaunchKernelGGL(k_flag_big, dim3(1), dim3(64), 0, s, big, dm, df, (uint32_t)i);
      long spins = 0; while (*(volatile uint32_t*)hf != (uint32_t)i) { if (++spins > 200000000L) { printf("flag timeout\n"); return 2; } }
    }
    CK(hipStreamSynchronize(s));
    double t7 = now();
    if (rep) printf("launch with 2176 B of kernel arguments + host-flag spin %.1f us\n", (t7 - t6) / N);
    if (rep) printf("per round trip (us): launch+sync %.1f | launch+memcpyD2H+sync %.1f | launch(mapped host store)+sync %.1f | launch+host-flag spin %.1f | 2 launches+memcpy+sync %.1f\n",
                    (t1 - t0) / N, (t2 - t1) / N, (t3 - t2) / N, (t4 - t3) / N, (t5 - t4) / N);
  }
  // host-side cost of an enqueue (the launch API call itself, nothing waited for): what every one of a proof's ~700 launches costs the host thread
  {
    Big big; for (int i = 0; i < 272; i++) big.p[i] = nullptr;
    for (int rep = 0; rep < 2; rep++) {
      CK(hipStreamSynchronize(s));
      double a0 = now();
      for (int i = 0; i < 400; i++) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, d, i);
      double a1 = now(); CK(hipStreamSynchronize(s)); double a2 = now();
      for (int i = 0; i < 400; i++) hipLaunchKernelGGL(k_flag_big, dim3(1), dim3(64), 0, s, big, dm, df, (uint32_t)i);
      double a3 = now(); CK(hipStreamSynchronize(s)); double a4 = now();
      if (rep) printf("enqueue only (host time per hipLaunchKernelGGL, 400 back-to-back): 16 B of arguments %.2f us (drain %.2f us per kernel) | 2176 B of arguments %.2f us (drain %.2f)\n",
                      (a1 - a0) / 400, (a2 - a0) / 400, (a3 - a2) / 400, (a4 - a2) / 400);
    }
  }
  return 0;
}

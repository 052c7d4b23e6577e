// What does the device -> host half of a hand-off cost?  A resident workgroup answers the host's word either the way row_done does it (result stores to
// host-mapped memory, system-scope fence, agent-scope ticket, system-scope flag store) or with SELF-VALIDATING 16-byte chunks [tag, w, w, w] and nothing else
// (no fence, no ticket, no flag: the host accepts a result when every chunk carries the turn's tag).  Every device spin has a wall-clock bail-out.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/handoff_bench tools/handoff_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <emmintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE 5: 16-byte stores with sc0 sc1 (system-scope write-through), nothing else;  MODE 4: as 2 + s_waitcnt vmcnt(0) (the stores have left the wave);  MODE 2: as 1 with plain stores; MODE 3: as 2 followed by a system-scope release fence
// MODE 0: flag protocol (as row_done with `wgs` workgroups);  MODE 1: tagged chunks, each workgroup writes its own 6 chunks (two field elements)
template <int MODE>
__global__ void k_resident(const uint32_t* mailbox, uint32_t* out, uint32_t* flag, uint32_t* counters, uint32_t turns) {
  const uint64_t t_end = wall_clock64() + 300000000ull;
  __shared__ uint32_t go;
  for (uint32_t i = 1; i <= turns; i++) {
    if (threadIdx.x == 0) {
      uint32_t ok = 1;
      while (__hip_atomic_load(mailbox, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != i) { if (wall_clock64() > t_end) { ok = 0; break; } }
      go = ok;
    }
    __syncthreads();
    if (!go) return;
    if (MODE == 0) {
      if (threadIdx.x < 2) { u32x4 v = {i, i + 1, i + 2, i + 3}; u32x4* o = reinterpret_cast<u32x4*>(out + (blockIdx.x * 2 + threadIdx.x) * 8); o[0] = v; o[1] = v; }   // two 32-byte elements
      if (threadIdx.x == 0) {
        __threadfence_system();
        uint32_t t2 = __hip_atomic_fetch_add(counters, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t2 == gridDim.x - 1) { *counters = 0; __hip_atomic_store(flag, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
      }
    } else {
      if (threadIdx.x < 6) {
        u32x4 v = {i, i + 1, i + 2, i + 3}; u32x4* o = reinterpret_cast<u32x4*>(out) + blockIdx.x * 6 + threadIdx.x;
        if (MODE == 1) __builtin_nontemporal_store(v, o); else if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(o), "v"(v) : "memory"); else *o = v;
        if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        if (MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
  }
}
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint32_t *h_mail, *h_out, *d_mail, *d_out, *d_cnt;
  CK(hipHostMalloc(&h_mail, 64, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostMalloc(&h_out, 4096, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer((void**)&d_mail, h_mail, 0)); CK(hipHostGetDevicePointer((void**)&d_out, h_out, 0));
  CK(hipMalloc(&d_cnt, 64)); CK(hipMemset(d_cnt, 0, 64));
  const uint32_t N = 3000;
  hipLaunchKernelGGL(k_resident<0>, dim3(1), dim3(256), 0, s, d_mail, d_out, d_out + 1000, d_cnt, 0u); hipLaunchKernelGGL(k_resident<1>, dim3(1), dim3(256), 0, s, d_mail, d_out, d_out + 1000, d_cnt, 0u); CK(hipStreamSynchronize(s));
  for (int rep = 0; rep < 2; rep++) for (uint32_t wgs : {1u, 2u, 16u}) for (int mode : {0, 3, 5}) {
    for (int k = 0; k < 1024; k++) h_out[k] = 0;
    *h_mail = 0;
    uint32_t* h_flag = h_out + 1000; uint32_t* d_flag = d_out + 1000;
    if (mode == 0) hipLaunchKernelGGL(k_resident<0>, dim3(wgs), dim3(256), 0, s, d_mail, d_out, d_flag, d_cnt, N);
    else if (mode == 1) hipLaunchKernelGGL(k_resident<1>, dim3(wgs), dim3(256), 0, s, d_mail, d_out, d_flag, d_cnt, N);
    else if (mode == 2) hipLaunchKernelGGL(k_resident<2>, dim3(wgs), dim3(256), 0, s, d_mail, d_out, d_flag, d_cnt, N);
    else if (mode == 5) hipLaunchKernelGGL(k_resident<5>, dim3(wgs), dim3(256), 0, s, d_mail, d_out, d_flag, d_cnt, N);
    else if (mode == 4) hipLaunchKernelGGL(k_resident<4>, dim3(wgs), dim3(256), 0, s, d_mail, d_out, d_flag, d_cnt, N);
    else hipLaunchKernelGGL(k_resident<3>, dim3(wgs), dim3(256), 0, s, d_mail, d_out, d_flag, d_cnt, N);
    double t0 = now(); uint32_t bad = 0; bool failed = false;
    for (uint32_t i = 1; i <= N && !failed; i++) {
      __atomic_store_n(h_mail, i, __ATOMIC_RELEASE);
      long spins = 0;
      if (mode == 0) { while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != i) { if ((++spins & 0xfffff) == 0 && now() - t0 > 5e6) { printf("timeout at turn %u (mode %d, %u workgroups)\n", i, mode, wgs); failed = true; break; } } }
      else {
        for (uint32_t c = 0; c < wgs * 6 && !failed; c++) {
          for (;;) {
            __m128i v = _mm_load_si128((const __m128i*)(h_out + 4 * c)); uint32_t w[4]; _mm_storeu_si128((__m128i*)w, v);
            if (w[0] == i) { if (w[1] != i + 1 || w[2] != i + 2 || w[3] != i + 3) bad++; break; }
            if ((++spins & 0xfffff) == 0 && now() - t0 > 5e6) { printf("timeout at turn %u (mode %d, %u workgroups)\n", i, mode, wgs); failed = true; break; }
          }
        }
      }
    }
    double t1 = now();
    CK(hipStreamSynchronize(s));
    if (rep || failed) printf("%2u workgroups, %s: %.2f us per turn%s\n", wgs, failed ? "FAILED" : mode == 0 ? "stores + fence + ticket + flag " : mode == 1 ? "tagged chunks, nontemporal     " : mode == 2 ? "tagged 16-byte chunks, plain   " : mode == 3 ? "tagged chunks + release fence  " : mode == 4 ? "tagged chunks + s_waitcnt      " : "tagged chunks, sc0 sc1 stores  ", (t1 - t0) / N, bad ? "  TORN CHUNKS SEEN" : "");
    if (bad) printf("   torn chunks: %u\n", bad);
  }
  return 0;
}

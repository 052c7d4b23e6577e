// How fast can host and device take turns WITHOUT a launch per turn?  A resident kernel waits for the host's word in a mailbox, answers
// through host-mapped memory, waits for the next word ...  (what a persistent sumcheck-tail kernel would do per round).  Every device
// spin has a wall-clock bail-out so that a dead host can never leave a kernel spinning.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/pingpong_bench tools/pingpong_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <utility>
#include <emmintrin.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// mailbox: where the host writes turn i; answer: host-mapped word the device writes i to
__global__ void k_resident(volatile uint32_t* mailbox, volatile uint32_t* answer, uint32_t turns, uint32_t work) {
  const uint64_t t_end = wall_clock64() + 200000000ull;   // 2 s at 100 MHz: bail out whatever happens
  uint32_t acc = 0;
  for (uint32_t i = 1; i <= turns; i++) {
    if (threadIdx.x == 0) {
      while (__hip_atomic_load(mailbox, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != i) { if (wall_clock64() > t_end) return; }
    }
    __syncthreads();
    for (uint32_t k = 0; k < work; k++) acc = acc * 1664525u + 1013904223u + threadIdx.x;   // stand-in for the round's arithmetic
    if (threadIdx.x == 0) { __hip_atomic_store(answer, i + (acc & 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
}
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint32_t *h_mail, *h_ans, *d_mail, *d_ans;
  CK(hipHostMalloc(&h_mail, 64, hipHostMallocMapped)); CK(hipHostMalloc(&h_ans, 64, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&d_mail, h_mail, 0)); CK(hipHostGetDevicePointer((void**)&d_ans, h_ans, 0));
  const uint32_t N = 2000;
  for (uint32_t work : {0u, 200u}) {
    *h_mail = 0; *h_ans = 0;
    hipLaunchKernelGGL(k_resident, dim3(1), dim3(256), 0, s, d_mail, d_ans, N, work);
    double t0 = now();
    for (uint32_t i = 1; i <= N; i++) {
      __atomic_store_n(h_mail, i, __ATOMIC_RELEASE);
      long spins = 0; while (__atomic_load_n(h_ans, __ATOMIC_ACQUIRE) != i) { if (++spins > 400000000L) { printf("timeout at turn %u\n", i); return 2; } }
    }
    double t1 = now();
    CK(hipStreamSynchronize(s));
    printf("resident kernel, mailbox in host-mapped memory, %u dependent ops of work per turn: %.2f us per turn\n", work, (t1 - t0) / N);
  }
  // mailbox in fine-grained DEVICE memory written by the host through the PCIe aperture, if this platform exposes it
  {
    uint32_t* dm = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&dm, 4096, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { printf("fine-grained device allocation unavailable: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); return 0; }
    hipPointerAttribute_t at; e = hipPointerGetAttributes(&at, dm);
    printf("fine-grained device memory: hostPointer=%p devicePointer=%p type=%d\n", e == hipSuccess ? at.hostPointer : nullptr, e == hipSuccess ? at.devicePointer : nullptr, e == hipSuccess ? (int)at.type : -1);
    CK(hipMemset(dm, 0, 4096));
    if (e == hipSuccess && at.hostPointer) {
      volatile uint32_t* hp = (volatile uint32_t*)at.hostPointer;
      *h_ans = 0;
      hipLaunchKernelGGL(k_resident, dim3(1), dim3(256), 0, s, dm, d_ans, N, 0u);
      double t0 = now();
      for (uint32_t i = 1; i <= N; i++) {
        *hp = i; __atomic_thread_fence(__ATOMIC_SEQ_CST);
        long spins = 0; while (__atomic_load_n(h_ans, __ATOMIC_ACQUIRE) != i) { if (++spins > 400000000L) { printf("timeout at turn %u\n", i); return 2; } }
      }
      double t1 = now();
      CK(hipStreamSynchronize(s));
      printf("resident kernel, mailbox in fine-grained device memory written by the host: %.2f us per turn\n", (t1 - t0) / N);
    } else {
      // no host pointer through HIP: ask the HSA runtime to let the CPU agent reach the allocation (large-BAR systems map VRAM at the same address in the process)
      hsa_agent_t cpu{}; bool have = false;
      auto cb = [](hsa_agent_t a, void* d) -> hsa_status_t { hsa_device_type_t ty; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &ty); auto* p = (std::pair<hsa_agent_t*, bool*>*)d; if (ty == HSA_DEVICE_TYPE_CPU && !*p->second) { *p->first = a; *p->second = true; } return HSA_STATUS_SUCCESS; };
      std::pair<hsa_agent_t*, bool*> arg{&cpu, &have};
      hsa_status_t st = hsa_init(); if (st == HSA_STATUS_SUCCESS) st = hsa_iterate_agents(cb, &arg);
      if (st == HSA_STATUS_SUCCESS && have) st = hsa_amd_agents_allow_access(1, &cpu, nullptr, dm);
      printf("hsa_amd_agents_allow_access(cpu, device allocation): status %d\n", (int)st);
      if (st == HSA_STATUS_SUCCESS && have) {
        volatile uint32_t* hp = (volatile uint32_t*)dm;
        *h_ans = 0;
        hipLaunchKernelGGL(k_resident, dim3(1), dim3(256), 0, s, dm, d_ans, N, 0u);
        double t0 = now(); bool bad = false;
        for (uint32_t i = 1; i <= N && !bad; i++) {
          *hp = i; _mm_sfence();
          long spins = 0; while (__atomic_load_n(h_ans, __ATOMIC_ACQUIRE) != i) { if (++spins > 2000000000L) { printf("timeout at turn %u\n", i); bad = true; break; } }
        }
        double t1 = now();
        CK(hipStreamSynchronize(s));
        if (!bad) printf("resident kernel, mailbox in DEVICE memory written by the host through the BAR: %.2f us per turn\n", (t1 - t0) / N);
      }
    }
  }
  return 0;
}

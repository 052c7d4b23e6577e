// Device-resident Fiat-Shamir microbenchmark (tools/transcript_dev.cuh): one wave runs a sumcheck-shaped transcript schedule — per round three
// 32-byte scalars appended, one 64-byte challenge drawn and reduced to Fr — and the host checks the final challenge against its own Merlin transcript
// (lasso_amd/host/hashes.hpp) and prints the time per round.  The number to compare with: a device -> host -> device turn costs >= 12 us per round today.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilasso_amd/csrc -Iinclude -o tools/transcript_bench tools/transcript_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "transcript_dev.cuh"
#include "../lasso_amd/host/hashes.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(64) k_transcript(const uint8_t* __restrict__ scalars, uint32_t rounds, uint8_t* __restrict__ out64, fr_t* __restrict__ out_fr, uint64_t* __restrict__ cycles) {
  __shared__ uint8_t msg[96];
  __shared__ uint8_t chal[64];
  wave_lanes ln;
  strobe_lanes<wave_lanes> tr;
  const uint8_t proto[7] = {'e', 'x', 'a', 'm', 'p', 'l', 'e'};
  tr.init_merlin(ln, proto, 7);
  const uint8_t lab_a[19] = {'c', 'o', 'm', 'b', 'i', 'n', 'e', '_', 's', 'c', 'a', 'l', 'a', 'r', '_', 'p', 'o', 'l', 'y'};
  const uint8_t lab_c[19] = {'c', 'h', 'a', 'l', 'l', 'e', 'n', 'g', 'e', '_', 'n', 'e', 'x', 't', 'r', 'o', 'u', 'n', 'd'};
  const uint64_t t0 = wall_clock64();
  for (uint32_t r = 0; r < rounds; r++) {
    for (uint32_t i = threadIdx.x; i < 96; i += 64) msg[i] = scalars[(size_t)r * 96 + i];
    __syncthreads();
    for (int k = 0; k < 3; k++) tr.append_message(ln, lab_a, 19, msg + 32 * k, 32);
    tr.challenge_bytes(ln, lab_c, 19, chal, 64);
    __syncthreads();
  }
  const uint64_t t1 = wall_clock64();
  if (threadIdx.x < 64) out64[threadIdx.x] = chal[threadIdx.x];
  if (threadIdx.x == 0) { *out_fr = fr_from_wide_bytes(chal); *cycles = t1 - t0; }
}

int main() {
  const uint32_t rounds = 512;
  std::vector<uint8_t> sc((size_t)rounds * 96); for (size_t i = 0; i < sc.size(); i++) sc[i] = (uint8_t)(i * 131u + 7u);
  uint8_t *d_sc, *d_out; fr_t* d_fr; uint64_t* d_cy;
  CK(hipMalloc(&d_sc, sc.size())); CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_fr, sizeof(fr_t))); CK(hipMalloc(&d_cy, 8));
  CK(hipMemcpy(d_sc, sc.data(), sc.size(), hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_transcript, dim3(1), dim3(64), 0, 0, d_sc, rounds, d_out, d_fr, d_cy);
  CK(hipDeviceSynchronize());
  uint8_t got[64]; uint64_t cy; CK(hipMemcpy(got, d_out, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(&cy, d_cy, 8, hipMemcpyDeviceToHost));
  lasso::Merlin ref("example"); uint8_t want[64];
  for (uint32_t r = 0; r < rounds; r++) { for (int k = 0; k < 3; k++) ref.append_message("combine_scalar_poly", &sc[(size_t)r * 96 + 32 * k], 32); ref.challenge_bytes("challenge_nextround", want, 64); }
  printf("device transcript %s the host's; %.2f us per round (3 appends + 1 challenge), %u rounds\n", memcmp(got, want, 64) == 0 ? "MATCHES" : "DIFFERS FROM", cy / 100.0 / rounds, rounds);
  return memcmp(got, want, 64) == 0 ? 0 : 1;
}

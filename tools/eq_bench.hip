// k_eq_outer (out[x] = hi[x >> lo_bits] * lo[x & mask], the table of a grand-product layer above 2^14 entries) against variants, and k_gp_layer (same shape: two reads,
// one product, one write) as the yardstick.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/eq_bench tools/eq_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../lasso_amd/csrc/poly_kernels.cuh"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
// variant: one workgroup per run of 2^lo_bits entries... each thread keeps hi in registers (unpacked once) and walks lo with a fixed stride
__global__ void __launch_bounds__(256) k_eq_outer_v2(const fr_t* __restrict__ hi, const fr_t* __restrict__ lo, uint32_t lo_bits, size_t n, fr_t* __restrict__ out) {
  const size_t per = ((size_t)1 << lo_bits);           // entries that share one hi
  const size_t runs = n >> lo_bits;
  for (size_t run = blockIdx.x; run < runs; run += gridDim.x) {
    const fr29 h = fr29_unpack_u(hi[run]);
    fr_t* __restrict__ o = out + run * per;
#pragma unroll 2
    for (size_t j = threadIdx.x; j < per; j += 256) o[j] = fr29_store(fr29_mul(h, fr29_unpack_s(lo[j])));
  }
}
template <class F> static double time_it(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); double best = 1e9;
  for (int i = 0; i < 6; i++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (i && ms < best) best = ms; }
  return best * 1e3;
}
int main() {
  for (uint32_t ell : {18u, 20u, 21u, 22u, 24u}) {
    const size_t n = (size_t)1 << ell; const uint32_t lo_bits = ell / 2, hi_bits = ell - lo_bits;
    fr_t *hi, *lo, *out, *in;
    CK(hipMalloc(&hi, sizeof(fr_t) << hi_bits)); CK(hipMalloc(&lo, sizeof(fr_t) << lo_bits)); CK(hipMalloc(&out, n * sizeof(fr_t))); CK(hipMalloc(&in, n * sizeof(fr_t)));
    CK(hipMemset(hi, 1, sizeof(fr_t) << hi_bits)); CK(hipMemset(lo, 2, sizeof(fr_t) << lo_bits)); CK(hipMemset(in, 3, n * sizeof(fr_t)));
    auto grid = [&](size_t cap) { size_t g = (n + 255) / 256; return (unsigned)(g > cap ? cap : g); };
    const double t1 = time_it([&] { hipLaunchKernelGGL(k_eq_outer, dim3(grid(4096)), dim3(256), 0, 0, (const fr_t*)hi, (const fr_t*)lo, lo_bits, n, out); });
    const double t1b = time_it([&] { hipLaunchKernelGGL(k_eq_outer, dim3(grid(16384)), dim3(256), 0, 0, (const fr_t*)hi, (const fr_t*)lo, lo_bits, n, out); });
    const double t2 = time_it([&] { hipLaunchKernelGGL(k_eq_outer_v2, dim3((unsigned)((n >> lo_bits) > 4096 ? 4096 : (n >> lo_bits))), dim3(256), 0, 0, (const fr_t*)hi, (const fr_t*)lo, lo_bits, n, out); });
    const double t3 = time_it([&] { hipLaunchKernelGGL(k_gp_layer, dim3(grid(2048)), dim3(256), 0, 0, (const fr_t*)in, n / 2, out); });
    printf("2^%u entries (%.0f MB written): k_eq_outer %.1f us (%.2f TB/s) | 16384 workgroups %.1f us | v2 (hi in registers) %.1f us (%.2f TB/s) | k_gp_layer over the same bytes/2: %.1f us\n",
           ell, n * 32e-6, t1, n * 32e-6 / t1, t1b, t2, n * 32e-6 / t2, t3);
    CK(hipFree(hi)); CK(hipFree(lo)); CK(hipFree(out)); CK(hipFree(in));
  }
  return 0;
}

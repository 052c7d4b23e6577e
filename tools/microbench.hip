// Micro-benchmarks that decide the kernel design on MI355X (run on the GPU box via gpurun):
//   1. issue rate of the integer-multiply instructions the 256-bit field arithmetic can be built from
//   2. Montgomery / pseudo-Mersenne multiplier throughput as compiled from fr.cuh / fq.cuh
//   3. bind_top achieved bandwidth next to pure-copy ceilings with the same access pattern
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <cstring>
#include "../lasso_amd/csrc/poly_kernels.cuh"
#include "../lasso_amd/csrc/msm_kernels.cuh"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// ---- 1. instruction issue rates: 8 independent chains per lane, ITER iterations
#define ITER 4096
#define DEF_RATE_KERNEL(NAME, BODY)                                                         \
  __global__ void NAME(uint32_t* out, uint32_t seed) {                                       \
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;     \
    uint32_t a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;         \
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7;         \
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;           \
    uint32_t m = seed | 1u;                                                                  \
    for (int i = 0; i < ITER; i++) { BODY }                                                  \
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7); \
    if (r == 0x12345678u) out[0] = r;                                                        \
  }
#define X8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define OP_MAD64(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##k) : "v"(a##k), "v"(m) : "vcc");
#define OP_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##k) : "v"(m));
#define OP_MULHI(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##k) : "v"(m));
#define OP_MAD24(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a##k) : "v"(m));
#define OP_MULHI24(k) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a##k) : "v"(m));
#define OP_ADD32(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##k) : "v"(m));
#define OP_ADD3(k) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a##k) : "v"(m));
#define OP_LSHLADD64(k) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(q##k));
#define OP_FMA64(k) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d##k));
#define OP_ADDC(k) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a##k) : "v"(m) : "vcc");
#define OP_MOV(k) asm volatile("v_mov_b32 %0, %1" : "=v"(a##k) : "v"(a##k));
DEF_RATE_KERNEL(k_rate_mad64, X8(OP_MAD64))
DEF_RATE_KERNEL(k_rate_mullo, X8(OP_MULLO))
DEF_RATE_KERNEL(k_rate_mulhi, X8(OP_MULHI))
DEF_RATE_KERNEL(k_rate_mad24, X8(OP_MAD24))
DEF_RATE_KERNEL(k_rate_mulhi24, X8(OP_MULHI24))
DEF_RATE_KERNEL(k_rate_add32, X8(OP_ADD32))
DEF_RATE_KERNEL(k_rate_add3, X8(OP_ADD3))
DEF_RATE_KERNEL(k_rate_lshladd64, X8(OP_LSHLADD64))
DEF_RATE_KERNEL(k_rate_fma64, X8(OP_FMA64))
DEF_RATE_KERNEL(k_rate_addc, X8(OP_ADDC))
DEF_RATE_KERNEL(k_rate_mov, X8(OP_MOV))

template <class K>
static double time_kernel(K launch, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int i = 0; i < reps; i++) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
  CK(hipGetLastError());
  return best;
}

// ---- 2. multiplier throughput: 4 independent chains per lane
__global__ void k_tp_frmul(fr_t* io, int iters) {
  fr_t a = io[threadIdx.x & 63], b = io[64 + (threadIdx.x & 63)], c = fr_add(a, b), d = fr_sub(a, b);
  for (int i = 0; i < iters; i++) { a = fr_mul(a, b); b = fr_mul(b, c); c = fr_mul(c, d); d = fr_mul(d, a); }
  fr_t r = fr_add(fr_add(a, b), fr_add(c, d));
  if (r.v[0] == 0x12345678u && r.v[1] == 0x9abcdef0u) io[blockIdx.x] = r;
}
__global__ void k_tp_fr29mul(fr_t* io, int iters) {
  fr29 a = fr29_unpack_u(io[threadIdx.x & 63]), b = fr29_unpack_s(io[64 + (threadIdx.x & 63)]), c = fr29_weak(fr29_add(a, b)), d = fr29_weak(fr29_sub(a, b));
  for (int i = 0; i < iters; i++) { a = fr29_mul(a, b); b = fr29_mul(b, c); c = fr29_mul(c, d); d = fr29_mul(d, a); }
  fr_t r = fr29_store(fr29_mul(fr29_add(fr29_add(a, b), fr29_add(c, d)), fr29_one_s()));
  if (r.v[0] == 0x12345678u && r.v[1] == 0x9abcdef0u) io[blockIdx.x] = r;
}
__global__ void k_tp_fqmul(fq_t* io, int iters) {
  fq_t a = io[threadIdx.x & 63], b = io[64 + (threadIdx.x & 63)], c = fq_add(a, b), d = fq_sub(a, b);
  for (int i = 0; i < iters; i++) { a = fq_mul(a, b); b = fq_mul(b, c); c = fq_mul(c, d); d = fq_mul(d, a); }
  fq_t r = fq_add(fq_add(a, b), fq_add(c, d));
  if (r.v[0] == 0x12345678u && r.v[1] == 0x9abcdef0u) io[blockIdx.x] = r;
}
#ifndef LASSO_BN254
__global__ void k_tp_madd(ed_point* io, const ed_niels* nb, int iters) {
  ed_point p = io[threadIdx.x & 63]; ed_niels n0 = nb[threadIdx.x & 63];
  for (int i = 0; i < iters; i++) p = ed_madd(p, n0);
  if (p.X.v[0] == 0x12345678u && p.Y.v[1] == 0x9abcdef0u) io[blockIdx.x] = p;
}
#endif

// the MSM kernels' own mixed addition (29-bit limbs: fe29.cuh, or bn254_fe29.cuh under -DLASSO_BN254): the ceiling roofline_msm is priced against
__global__ void k_tp_ptmadd(pt29* io, const niels29* nb, int iters) {
  pt29 p = io[threadIdx.x & 63]; const niels29 n0 = nb[threadIdx.x & 63];
  for (int i = 0; i < iters; i++) p = pt_madd(p, n0);
  if (p.X.v[0] == 0x12345678 && p.Y.v[1] == 0x1abcdef0) io[blockIdx.x] = p;
}

// ---- 5. streaming-ceiling exploration: cache policy (nt = non-temporal), loads in flight per lane, grid size
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int LDNT, int STNT, int UNROLL>
__global__ void __launch_bounds__(256) k_copy_v(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = LDNT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) { if (STNT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
template <int LDNT>
__global__ void __launch_bounds__(256) k_read_v(const u32x4* __restrict__ src, size_t n16, uint32_t* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += stride) { const u32x4 v = LDNT ? __builtin_nontemporal_load(src + i) : src[i]; acc ^= v; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) out[0] = 1;
}
template <int STNT>
__global__ void __launch_bounds__(256) k_write_v(u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const u32x4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += stride) { if (STNT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
}
// k_bind_top with selectable cache policy on its two streams (in place, like the product kernel)
template <int LDNT, int STNT>
__global__ void __launch_bounds__(256) k_bind_pol(fr_t* __restrict__ z, size_t half, fr_t r) {
  const fr29 rs = fr29_unpack_s(r);
  u32x4* z4 = reinterpret_cast<u32x4*>(z);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    u32x4 l0, l1, h0, h1;
    if (LDNT) { l0 = __builtin_nontemporal_load(z4 + 2 * i); l1 = __builtin_nontemporal_load(z4 + 2 * i + 1); h0 = __builtin_nontemporal_load(z4 + 2 * (i + half)); h1 = __builtin_nontemporal_load(z4 + 2 * (i + half) + 1); }
    else { l0 = z4[2 * i]; l1 = z4[2 * i + 1]; h0 = z4[2 * (i + half)]; h1 = z4[2 * (i + half) + 1]; }
    fr_t lo, hi;
    lo.v[0] = l0.x; lo.v[1] = l0.y; lo.v[2] = l0.z; lo.v[3] = l0.w; lo.v[4] = l1.x; lo.v[5] = l1.y; lo.v[6] = l1.z; lo.v[7] = l1.w;
    hi.v[0] = h0.x; hi.v[1] = h0.y; hi.v[2] = h0.z; hi.v[3] = h0.w; hi.v[4] = h1.x; hi.v[5] = h1.y; hi.v[6] = h1.z; hi.v[7] = h1.w;
    const fr29 a = fr29_unpack_u(lo), b = fr29_unpack_u(hi);
    const fr_t o = fr29_store(fr29_add(a, fr29_mul(fr29_sub(b, a), rs)));
    const u32x4 o0 = {o.v[0], o.v[1], o.v[2], o.v[3]}, o1 = {o.v[4], o.v[5], o.v[6], o.v[7]};
    if (STNT) { __builtin_nontemporal_store(o0, z4 + 2 * i); __builtin_nontemporal_store(o1, z4 + 2 * i + 1); } else { z4[2 * i] = o0; z4[2 * i + 1] = o1; }
  }
}

// ---- 3. bandwidth kernels
__global__ void __launch_bounds__(256) k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// same traffic as bind_top (read lo+hi, write lo) without arithmetic
__global__ void __launch_bounds__(256) k_bind_traffic(fr_t* __restrict__ z, size_t half) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    fr_t lo = z[i], hi = z[i + half];
#pragma unroll
    for (int k = 0; k < 8; k++) lo.v[k] ^= hi.v[k];
    z[i] = lo;
  }
}
// bind with a lane pair per two elements: every 16-byte load is part of a fully contiguous 1 KiB wave access
__global__ void __launch_bounds__(256) k_bind_pairs(fr_t* __restrict__ z, size_t half, fr_t r) {
  // thread t handles element e = 2*(t/2) + (t&1) of its wave's 64-element tile, but loads halves so that lane l reads 16 B at byte 16*l
  uint4* z4 = reinterpret_cast<uint4*>(z);
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t base = (tid & ~(size_t)63); base < half; base += stride) {
    const int lane = threadIdx.x & 63;
    // tile of 64 elements = 128 uint4: pass 0 covers elements base..base+31, pass 1 covers base+32..base+63
    uint4 l0 = z4[2 * base + lane], l1 = z4[2 * base + 64 + lane];
    uint4 h0 = z4[2 * (base + half) + lane], h1 = z4[2 * (base + half) + 64 + lane];
    // lane pair (2j, 2j+1): lane 2j takes element j of pass 0, lane 2j+1 takes element j of pass 1
    uint4 pl0, pl1, ph0, ph1;
#define SWZ(dst, src) dst.x = __shfl_xor((int)src.x, 1, 64); dst.y = __shfl_xor((int)src.y, 1, 64); dst.z = __shfl_xor((int)src.z, 1, 64); dst.w = __shfl_xor((int)src.w, 1, 64);
    SWZ(pl0, l0) SWZ(pl1, l1) SWZ(ph0, h0) SWZ(ph1, h1)
    const bool odd = lane & 1;
    fr_t lo, hi;
    uint4 loA = odd ? pl1 : l0, loB = odd ? l1 : pl0, hiA = odd ? ph1 : h0, hiB = odd ? h1 : ph0;
    lo.v[0] = loA.x; lo.v[1] = loA.y; lo.v[2] = loA.z; lo.v[3] = loA.w; lo.v[4] = loB.x; lo.v[5] = loB.y; lo.v[6] = loB.z; lo.v[7] = loB.w;
    hi.v[0] = hiA.x; hi.v[1] = hiA.y; hi.v[2] = hiA.z; hi.v[3] = hiA.w; hi.v[4] = hiB.x; hi.v[5] = hiB.y; hi.v[6] = hiB.z; hi.v[7] = hiB.w;
    fr_t o = fr_add(lo, fr_mul(r, fr_sub(hi, lo)));
    // write back: lane 2j holds element j (pass 0), lane 2j+1 holds element 32+j (pass 1)
    uint4 oA = make_uint4(o.v[0], o.v[1], o.v[2], o.v[3]), oB = make_uint4(o.v[4], o.v[5], o.v[6], o.v[7]), xA, xB;
    SWZ(xA, oA) SWZ(xB, oB)
    uint4 w0 = odd ? xB : oA;   // pass 0, position lane: even lane -> low half of its own element; odd lane -> high half of partner's (even lane's) element
    uint4 w1 = odd ? oB : xA;   // pass 1: even lane -> low half of partner's element; odd lane -> high half of own
    z4[2 * base + lane] = w0; z4[2 * base + 64 + lane] = w1;
  }
}

static bool want(int argc, char** argv, int sec) {   // microbench [sections]: e.g. `microbench 2,4,5`; no argument = 1..4 (section 5 only on request: it allocates up to 8 GiB)
  if (argc < 2) return sec <= 4;
  const std::string a = std::string(",") + argv[1] + ",";
  return a.find("," + std::to_string(sec) + ",") != std::string::npos;
}
int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d MHz  memclk=%d MHz  L2=%d MB\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.memoryClockRate / 1000, prop.l2CacheSize >> 20);
  const int CU = prop.multiProcessorCount;
  uint32_t* d_out; CK(hipMalloc(&d_out, 4096));

  if (want(argc, argv, 1)) {
  printf("\n== 1. instruction issue (wave-instructions per cycle per CU at 2.4 GHz nominal; 8 waves/SIMD resident)\n");
  struct { const char* name; void (*k)(uint32_t*, uint32_t); int per_op; } rates[] = {
      {"v_mad_u64_u32", k_rate_mad64, 1}, {"v_mul_lo_u32", k_rate_mullo, 1}, {"v_mul_hi_u32", k_rate_mulhi, 1}, {"v_mad_u32_u24", k_rate_mad24, 1},
      {"v_mul_hi_u32_u24", k_rate_mulhi24, 1}, {"v_add_u32", k_rate_add32, 1}, {"v_add3_u32", k_rate_add3, 1}, {"v_lshl_add_u64", k_rate_lshladd64, 1},
      {"v_fma_f64", k_rate_fma64, 1}, {"v_add_co+v_addc_co", k_rate_addc, 2}, {"v_mov_b32", k_rate_mov, 1}};
  for (auto& r : rates) {
    const int blocks = CU * 8, threads = 256;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    double ms = time_kernel([&] { hipLaunchKernelGGL(r.k, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u); });
    double wave_instr = (double)blocks * (threads / 64) * ITER * 8 * r.per_op;
    double per_s = wave_instr / (ms * 1e-3);
    printf("  %-22s %8.3f ms  %8.2f G wave-instr/s  => %.2f cycles per wave-instr per SIMD (@2.4GHz)  lane-ops %.1f T/s\n", r.name, ms, per_s * 1e-9, (double)CU * 4 * 2.4e9 / per_s, per_s * 64e-12);
  }

  }
  if (want(argc, argv, 2)) {
  printf("\n== 2. multiplier throughput (G mult/s, whole chip)\n");
  {
    std::vector<fr_t> h(256); for (size_t i = 0; i < h.size(); i++) { h[i] = fr_from_u64(0x9e3779b97f4a7c15ull * (i + 1)); }
    fr_t* d; CK(hipMalloc(&d, 65536 * sizeof(fr_t))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(fr_t), hipMemcpyHostToDevice));
    for (int wpb : {4, 8}) {
      const int blocks = CU * wpb * 4 / 4, threads = 256, iters = 256;
      double ms = time_kernel([&] { hipLaunchKernelGGL(k_tp_frmul, dim3(blocks), dim3(threads), 0, 0, d, iters); });
      printf("  fr_mul (Montgomery 8x32 CIOS)   blocks/CU=%d: %8.3f ms  %7.1f G/s\n", wpb, ms, (double)blocks * threads * iters * 4 / (ms * 1e-3) * 1e-9);
      ms = time_kernel([&] { hipLaunchKernelGGL(k_tp_fr29mul, dim3(blocks), dim3(threads), 0, 0, d, iters); });
      printf("  fr29_mul (Montgomery 9x29, radix 2^261) blocks/CU=%d: %8.3f ms  %7.1f G/s\n", wpb, ms, (double)blocks * threads * iters * 4 / (ms * 1e-3) * 1e-9);
      ms = time_kernel([&] { hipLaunchKernelGGL(k_tp_fqmul, dim3(blocks), dim3(threads), 0, 0, (fq_t*)d, iters); });
      printf("  fq_mul (2^255-19 fold)          blocks/CU=%d: %8.3f ms  %7.1f G/s\n", wpb, ms, (double)blocks * threads * iters * 4 / (ms * 1e-3) * 1e-9);
    }
#ifndef LASSO_BN254
    std::vector<ed_point> hp(64); std::vector<ed_niels> hn(64);
    for (int i = 0; i < 64; i++) { fq_t x = fq_zero(), y = fq_one(); hp[i] = ed_from_affine(x, y); fq_t a = fq_zero(); a.v[0] = 5 + i; hn[i].ypx = a; hn[i].ymx = fq_one(); hn[i].t2d = a; }
    ed_point* dp; ed_niels* dn; CK(hipMalloc(&dp, 65536 * sizeof(ed_point))); CK(hipMalloc(&dn, 64 * sizeof(ed_niels)));
    CK(hipMemcpy(dp, hp.data(), 64 * sizeof(ed_point), hipMemcpyHostToDevice)); CK(hipMemcpy(dn, hn.data(), 64 * sizeof(ed_niels), hipMemcpyHostToDevice));
    const int blocks = CU * 4, threads = 256, iters = 256;
    double ms = time_kernel([&] { hipLaunchKernelGGL(k_tp_madd, dim3(blocks), dim3(threads), 0, 0, dp, dn, iters); });
    printf("  ed_madd (7 fq_mul)              blocks/CU=4: %8.3f ms  %7.1f G madd/s\n", ms, (double)blocks * threads * iters / (ms * 1e-3) * 1e-9);
    CK(hipFree(dp)); CK(hipFree(dn));
#endif
    CK(hipFree(d));
    // the kernels' own mixed addition in 29-bit limbs, at 1 / 2 / 4 workgroups of 256 threads per CU (the MSM kernels run 1 per CU: VGPR-heavy)
    {
      std::vector<pt29> hq(64); std::vector<niels29> hm(64);
      const pt29 id = pt_identity();
      for (int i = 0; i < 64; i++) { hq[i] = id; memset(&hm[i], 0, sizeof(niels29)); int32_t* w = reinterpret_cast<int32_t*>(&hm[i]); for (size_t k = 0; k < sizeof(niels29) / 4; k++) w[k] = (int32_t)((k * 2654435761u + i * 40503u) & 0x0fffffff); }
      pt29* dq; niels29* dm; CK(hipMalloc(&dq, 65536 * sizeof(pt29))); CK(hipMalloc(&dm, 64 * sizeof(niels29)));
      CK(hipMemcpy(dq, hq.data(), 64 * sizeof(pt29), hipMemcpyHostToDevice)); CK(hipMemcpy(dm, hm.data(), 64 * sizeof(niels29), hipMemcpyHostToDevice));
      double best = 0; int best_w = 0;
      for (int wpc : {1, 2, 4}) {
        const int blocks2 = CU * wpc, iters2 = 256;
        double ms2 = time_kernel([&] { hipLaunchKernelGGL(k_tp_ptmadd, dim3(blocks2), dim3(256), 0, 0, dq, dm, iters2); });
        const double g = (double)blocks2 * 256 * iters2 / (ms2 * 1e-3) * 1e-9;
        printf("  pt_madd (29-bit limbs, the MSM kernels' mixed addition) workgroups/CU=%d: %8.3f ms  %7.2f G madd/s\n", wpc, ms2, g);
        if (g > best) { best = g; best_w = wpc; }
      }
#ifdef LASSO_BN254
      const char* curve = "bn254";
#else
      const char* curve = "curve25519";
#endif
      printf("MADD_CEILING {\"curve\": \"%s\", \"G_madd_per_s\": %.2f, \"workgroups_per_cu\": %d, \"device\": \"%s\", \"CUs\": %d}\n", curve, best, best_w, prop.name, CU);
      CK(hipFree(dq)); CK(hipFree(dm));
    }
  }
  }

  if (want(argc, argv, 3)) {
  printf("\n== 3. bandwidth (GB/s); bind_top algorithmic bytes = 48*n (read 32n + write 16n)\n");
  for (int logn : {22, 24, 26}) {
    const size_t n = (size_t)1 << logn, half = n / 2;
    fr_t* z; CK(hipMalloc(&z, n * sizeof(fr_t))); CK(hipMemset(z, 0x11, n * sizeof(fr_t)));
    uint4* dst; CK(hipMalloc(&dst, n * sizeof(fr_t)));
    fr_t r = fr_from_u64(0x123456789abcdefull);
    MutPtrTable T; T.p[0] = z;
    double ms = time_kernel([&] { hipLaunchKernelGGL(k_copy16, dim3(CU * 8), dim3(256), 0, 0, (const uint4*)z, dst, n * 2); });
    printf("  n=2^%d copy16 (r+w %zu MB): %7.3f ms  %7.1f GB/s\n", logn, (n * 64) >> 20, ms, n * 64.0 / (ms * 1e-3) * 1e-9);
    for (int cap : {CU * 4, CU * 8, CU * 16, CU * 32}) {
      size_t g = (half + 255) / 256; if (g > (size_t)cap) g = cap;
      ms = time_kernel([&] { hipLaunchKernelGGL(k_bind_traffic, dim3((unsigned)g), dim3(256), 0, 0, z, half); });
      double t1 = n * 48.0 / (ms * 1e-3) * 1e-9;
      ms = time_kernel([&] { hipLaunchKernelGGL(k_bind_top, dim3((unsigned)g, 1), dim3(256), 0, 0, T, half, r); });
      double t2 = n * 48.0 / (ms * 1e-3) * 1e-9;
      ms = time_kernel([&] { hipLaunchKernelGGL(k_bind_pairs, dim3((unsigned)g), dim3(256), 0, 0, z, half, r); });
      double t3 = n * 48.0 / (ms * 1e-3) * 1e-9;
      printf("  n=2^%d grid=%5zu: traffic-only %7.1f GB/s | bind_top %7.1f GB/s (%.3f ms) | bind_pairs %7.1f GB/s\n", logn, g, t1, t2, n * 48.0 / t2 * 1e-6, t3);
    }
    // fused bind + eq-weighted cubic round with 2 circuits (the dominant streaming kernel of a proof)
    if (logn <= 24) {
      fr_t *a0, *b0, *a1, *b1, *part, *small; uint32_t* counters;
      CK(hipMalloc(&a0, n * 32)); CK(hipMalloc(&b0, n * 32)); CK(hipMalloc(&a1, n * 32)); CK(hipMalloc(&b1, n * 32)); CK(hipMalloc(&part, 4096 * 6 * 32)); CK(hipMalloc(&small, 4096)); CK(hipMalloc(&counters, 4096)); CK(hipMemset(counters, 0, 4096));
      CK(hipMemset(a0, 0x07, n * 32)); CK(hipMemset(b0, 0x05, n * 32)); CK(hipMemset(a1, 0x03, n * 32)); CK(hipMemset(b1, 0x02, n * 32));
      MutPtrTable A, B; A.p[0] = a0; A.p[1] = a1; B.p[0] = b0; B.p[1] = b1;
      for (unsigned nx : {128u, 256u, 512u}) {
        ms = time_kernel([&] { hipLaunchKernelGGL((k_cubic_eqw_fused<3, false>), dim3(nx * 2), dim3(256), 0, 0, A, B, nx, 2u, (const fr_t*)z, n / 4, r, part, counters, small, (uint32_t*)nullptr, 0u); });
        printf("  n=2^%d fused bind + eq-weighted cubic round (3 sums), k=2, nx=%u: %7.3f ms  alg (48 n (2k+1)) %7.1f GB/s\n", logn, nx, ms, n * 48.0 * 5 / (ms * 1e-3) * 1e-9);
        ms = time_kernel([&] { hipLaunchKernelGGL((k_cubic_eqw_fused<2, false>), dim3(nx * 2), dim3(256), 0, 0, A, B, nx, 2u, (const fr_t*)z, n / 4, r, part, counters, small, (uint32_t*)nullptr, 0u); });
        printf("  n=2^%d fused bind + eq-weighted cubic round (2 sums), k=2, nx=%u: %7.3f ms  alg (48 n (2k+1)) %7.1f GB/s\n", logn, nx, ms, n * 48.0 * 5 / (ms * 1e-3) * 1e-9);
        ms = time_kernel([&] { hipLaunchKernelGGL((k_cubic_eqw_fused<2, true>), dim3(nx * 2), dim3(256), 0, 0, A, B, nx, 2u, (const fr_t*)z, n / 4, r, part, counters, small, (uint32_t*)nullptr, 0u); });
        printf("  n=2^%d fused bind + eq-weighted cubic round (2 sums, double-width accumulators), k=2, nx=%u: %7.3f ms  alg (48 n (2k+1)) %7.1f GB/s\n", logn, nx, ms, n * 48.0 * 5 / (ms * 1e-3) * 1e-9);
      }
      CK(hipFree(a0)); CK(hipFree(b0)); CK(hipFree(a1)); CK(hipFree(b1)); CK(hipFree(part)); CK(hipFree(small)); CK(hipFree(counters));
    }
    CK(hipFree(z)); CK(hipFree(dst));
  }
  // SURVEY.md 8(d)'s kernel-level sweep as written: bind_top on p polynomials of n in {2^20, 2^24, 2^26, 2^28} field elements (p = 1 and the batched
  // alpha + 1 in {2, 5, 9, 17}), 3 warm-ups and 20 timed launches on buffers that are not bound in place repeatedly (a bind halves the live length, so every
  // launch binds the SAME full-length arrays: the upper halves are only read, the lower halves are rewritten — the traffic of a first bind every time).
  // Skipped when p * n * 32 bytes exceeds 160 GiB.
  }
  if (want(argc, argv, 4)) {
  printf("\n== 4. bind_top sweep (SURVEY 8d): algorithmic bytes 48 n p; 3 warm-ups + 20 timed launches\n");
  for (int logn : {20, 24, 26, 28}) {
    const size_t n = (size_t)1 << logn, half = n / 2;
    for (int p : {1, 2, 5, 9, 17}) {
      if ((double)p * n * 32.0 > 72.0 * 1073741824.0) { printf("  n=2^%d p=%2d: skipped (%.0f GiB)\n", logn, p, p * n * 32.0 / 1073741824.0); continue; }
      MutPtrTable T; bool ok = true;
      for (int k = 0; k < p; k++) { fr_t* z = nullptr; if (hipMalloc(&z, n * sizeof(fr_t)) != hipSuccess) { ok = false; T.p[k] = nullptr; break; } CK(hipMemset(z, 0x11 + k, n * sizeof(fr_t))); T.p[k] = z; }
      if (ok) {
        const fr_t r = fr_from_u64(0x123456789abcdefull);
        size_t g = (half + 255) / 256; if (g > (size_t)CU * 32) g = (size_t)CU * 32;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_bind_top, dim3((unsigned)g, (unsigned)p), dim3(256), 0, 0, T, half, r);
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < 20; it++) hipLaunchKernelGGL(k_bind_top, dim3((unsigned)g, (unsigned)p), dim3(256), 0, 0, T, half, r);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        printf("  n=2^%d p=%2d: %8.3f ms per launch  %7.1f GB/s algorithmic (%.3f of 8000)\n", logn, p, ms, 48.0 * n * p / (ms * 1e-3) * 1e-9, 48.0 * n * p / (ms * 1e-3) * 1e-9 / 8000.0);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
      } else printf("  n=2^%d p=%2d: allocation failed, skipped\n", logn, p);
      for (int k = 0; k < p; k++) if (T.p[k]) CK(hipFree(T.p[k]));
    }
  }
  }
  if (want(argc, argv, 5)) {
  // Where is the streaming ceiling?  (copy16 tops out at 4.9-5.3 TB/s, the guide measures 6.29 TB/s for a float4 copy.)
  printf("\n== 5. streaming ceiling: policy x loads in flight x grid x size (GB/s = bytes read + bytes written per second)\n");
  for (size_t mib : {(size_t)128, (size_t)1024, (size_t)4096}) {
    const size_t bytes = mib << 20, n16 = bytes / 16;
    u32x4 *src, *dst; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMemset(src, 0x5a, bytes)); CK(hipMemset(dst, 0, bytes));
    for (int gm : {4, 8, 16, 32, 64}) {
      const unsigned g = (unsigned)(CU * gm);
      double a = time_kernel([&] { hipLaunchKernelGGL((k_copy_v<0, 0, 1>), dim3(g), dim3(256), 0, 0, (const u32x4*)src, dst, n16); });
      double b = time_kernel([&] { hipLaunchKernelGGL((k_copy_v<1, 1, 1>), dim3(g), dim3(256), 0, 0, (const u32x4*)src, dst, n16); });
      double c = time_kernel([&] { hipLaunchKernelGGL((k_copy_v<0, 0, 4>), dim3(g), dim3(256), 0, 0, (const u32x4*)src, dst, n16); });
      double d = time_kernel([&] { hipLaunchKernelGGL((k_copy_v<1, 1, 4>), dim3(g), dim3(256), 0, 0, (const u32x4*)src, dst, n16); });
      double e = time_kernel([&] { hipLaunchKernelGGL((k_copy_v<0, 1, 4>), dim3(g), dim3(256), 0, 0, (const u32x4*)src, dst, n16); });
      double f = time_kernel([&] { hipLaunchKernelGGL((k_copy_v<1, 0, 4>), dim3(g), dim3(256), 0, 0, (const u32x4*)src, dst, n16); });
      double h8 = time_kernel([&] { hipLaunchKernelGGL((k_copy_v<1, 1, 8>), dim3(g), dim3(256), 0, 0, (const u32x4*)src, dst, n16); });
#define GBS(ms_) (2.0 * bytes / ((ms_) * 1e-3) * 1e-9)
      printf("  copy %4zu MiB grid=CUx%-2d: plain u1 %6.0f | nt/nt u1 %6.0f | plain u4 %6.0f | nt/nt u4 %6.0f | ld plain st nt u4 %6.0f | ld nt st plain u4 %6.0f | nt/nt u8 %6.0f\n", mib, gm, GBS(a), GBS(b), GBS(c), GBS(d), GBS(e), GBS(f), GBS(h8));
    }
    for (int gm : {8, 32}) {
      const unsigned g = (unsigned)(CU * gm);
      double r0 = time_kernel([&] { hipLaunchKernelGGL((k_read_v<0>), dim3(g), dim3(256), 0, 0, (const u32x4*)src, n16, d_out); });
      double r1 = time_kernel([&] { hipLaunchKernelGGL((k_read_v<1>), dim3(g), dim3(256), 0, 0, (const u32x4*)src, n16, d_out); });
      double w0 = time_kernel([&] { hipLaunchKernelGGL((k_write_v<0>), dim3(g), dim3(256), 0, 0, dst, n16); });
      double w1 = time_kernel([&] { hipLaunchKernelGGL((k_write_v<1>), dim3(g), dim3(256), 0, 0, dst, n16); });
      printf("  %4zu MiB grid=CUx%-2d: read-only plain %6.0f nt %6.0f | write-only plain %6.0f nt %6.0f GB/s\n", mib, gm, bytes / (r0 * 1e-3) * 1e-9, bytes / (r1 * 1e-3) * 1e-9, bytes / (w0 * 1e-3) * 1e-9, bytes / (w1 * 1e-3) * 1e-9);
    }
    CK(hipFree(src)); CK(hipFree(dst));
  }
  for (int logn : {24, 26}) {
    const size_t n = (size_t)1 << logn, half = n / 2;
    fr_t* z; CK(hipMalloc(&z, n * sizeof(fr_t))); CK(hipMemset(z, 0x11, n * sizeof(fr_t)));
    const fr_t r = fr_from_u64(0x123456789abcdefull);
    MutPtrTable T; T.p[0] = z;
    for (int gm : {8, 16, 32, 64}) {
      size_t g = (half + 255) / 256; if (g > (size_t)CU * gm) g = (size_t)CU * gm;
      double m0 = time_kernel([&] { hipLaunchKernelGGL(k_bind_top, dim3((unsigned)g, 1), dim3(256), 0, 0, T, half, r); });
      double m1 = time_kernel([&] { hipLaunchKernelGGL((k_bind_pol<0, 0>), dim3((unsigned)g), dim3(256), 0, 0, z, half, r); });
      double m2 = time_kernel([&] { hipLaunchKernelGGL((k_bind_pol<1, 0>), dim3((unsigned)g), dim3(256), 0, 0, z, half, r); });
      double m3 = time_kernel([&] { hipLaunchKernelGGL((k_bind_pol<0, 1>), dim3((unsigned)g), dim3(256), 0, 0, z, half, r); });
      double m4 = time_kernel([&] { hipLaunchKernelGGL((k_bind_pol<1, 1>), dim3((unsigned)g), dim3(256), 0, 0, z, half, r); });
#define BGB(ms_) (48.0 * n / ((ms_) * 1e-3) * 1e-9)
      printf("  bind n=2^%d grid=CUx%-2d: k_bind_top %6.0f | same, explicit 16B plain %6.0f | ld nt %6.0f | st nt %6.0f | ld nt + st nt %6.0f GB/s algorithmic\n", logn, gm, BGB(m0), BGB(m1), BGB(m2), BGB(m3), BGB(m4));
    }
    CK(hipFree(z));
  }
  }
  printf("\ndone\n");
  return 0;
}

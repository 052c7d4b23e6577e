// CPU check of the two hand-off encodings between the kernels and the host: the DEVICE side's result_store / result_check / mail_valid are extracted from
// lasso_amd/csrc/poly_kernels.cuh and the HOST side's tagged_element / mail_chunks from lasso_amd/csrc/lasso_hip.hip by tests/test_host_arith_cpp.py (handoff_extract.hpp:
// this test follows the product's text), compiled for the host with clang (ext_vector_type) and played against each other: what one side writes the other accepts, and a
// stale, torn or corrupted chunk is refused.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <emmintrin.h>
#include "../../lasso_amd/csrc/fr.cuh"
#define __device__
#define __forceinline__ inline
#define __restrict__
typedef uint32_t lasso_u32x4 __attribute__((ext_vector_type(4)));
// round 6: result_store's partials path (flag == nullptr) uses write-through 8-byte stores on the device; on the host they are plain stores of the same bytes
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_store(ptr, val, order, scope) (*(ptr) = (val))
#include "handoff_extract.hpp"
#define CHECK(c) do { if (!(c)) { printf("FAIL %s line %d\n", #c, __LINE__); return 1; } } while (0)
int main() {
  std::mt19937_64 rng(5);
  alignas(16) uint32_t area[12 * 8]; alignas(16) fr_t plain[8];
  for (int trial = 0; trial < 2000; trial++) {
    const uint32_t seq = (uint32_t)rng() | 1u, slot = (uint32_t)(rng() % 8);
    fr_t v; for (int k = 0; k < 8; k++) v.v[k] = (uint32_t)rng();
    if (trial % 7 == 0) memset(v.v, 0, 32);
    if (trial % 11 == 0) memset(v.v, 0xff, 32);
    memset(area, 0, sizeof(area));
    // device -> device, block partials (flag == nullptr): the element's 32 bytes as they are, at 32 * slot
    result_store(plain, slot, v, (uint32_t*)nullptr, seq);
    CHECK(memcmp(plain[slot].v, v.v, 32) == 0);
    // device -> host, tagged: three chunks at 48 * slot, accepted with exactly the stored words
    result_store(reinterpret_cast<fr_t*>(area), slot, v, LASSO_TAGGED, seq);
    uint32_t w[8];
    CHECK(tagged_element(area + 12 * slot, seq, w) && memcmp(w, v.v, 32) == 0);
    CHECK(!tagged_element(area + 12 * slot, seq + 1, w));                        // another hand-off's number
    CHECK(!tagged_element(area + 12 * ((slot + 1) % 8), seq, w));                // an untouched slot
    for (int c = 0; c < 3; c++) {                                                // one chunk still from an older hand-off
      uint32_t save[4]; memcpy(save, area + 12 * slot + 4 * c, 16);
      area[12 * slot + 4 * c] = seq - 1; CHECK(!tagged_element(area + 12 * slot, seq, w));
      memcpy(area + 12 * slot + 4 * c, save, 16);
    }
    for (int k = 0; k < 12; k++) if (k % 4) {                                    // a flipped bit in any word (the check word included)
      area[12 * slot + k] ^= 1u << (rng() % 32); CHECK(!tagged_element(area + 12 * slot, seq, w)); area[12 * slot + k] = 0; result_store(reinterpret_cast<fr_t*>(area), slot, v, LASSO_TAGGED, seq);
    }
    // the same through the direct-publication sentinel, and the plain path (a flag pointer): out[slot] = v
    memset(area, 0, sizeof(area)); result_store(reinterpret_cast<fr_t*>(area), slot, v, LASSO_TAGGED_DIRECT, seq); CHECK(tagged_element(area + 12 * slot, seq, w) && memcmp(w, v.v, 32) == 0);
    uint32_t flagword = 0; memset(plain, 0, sizeof(plain)); result_store(plain, slot, v, &flagword, seq); CHECK(memcmp(plain[slot].v, v.v, 32) == 0);
    // host -> device: the mailbox
    alignas(16) uint32_t mail[12]; mail_chunks(mail, seq, v.v);
    lasso_u32x4 c0, c1, c2; memcpy(&c0, mail, 16); memcpy(&c1, mail + 4, 16); memcpy(&c2, mail + 8, 16);
    CHECK(mail_valid(c0, c1, c2, seq) && !mail_valid(c0, c1, c2, seq + 1));
    CHECK(c0.y == v.v[0] && c0.z == v.v[1] && c0.w == v.v[2] && c1.y == v.v[3] && c1.z == v.v[4] && c1.w == v.v[5] && c2.y == v.v[6] && c2.z == v.v[7]);   // the words the kernels take the challenge from
    lasso_u32x4 t = c1; t.z ^= 4u; CHECK(!mail_valid(c0, t, c2, seq));
    t = c1; t.x = seq - 1; CHECK(!mail_valid(c0, t, c2, seq));
  }
  printf("OK\n");
  return 0;
}

// CPU check of tools/transcript_dev.cuh (Keccak-f[1600] / STROBE-128 / Merlin over 25 lanes, one state word each) against the host
// transcript of the product (lasso_amd/host/hashes.hpp, itself pinned to merlin's published vector through the oracle tests).  The lane backend here
// executes the same index arithmetic the device backend does; only the exchange primitive (array lookup instead of __shfl) differs.
#include "../../tools/transcript_dev.cuh"
#include "../../lasso_amd/host/field_host.hpp"
#include "../../lasso_amd/host/hashes.hpp"
#include <array>
#include <random>
#include <cstdio>
#include <string>
#include <vector>

struct host_lanes {
  typedef std::array<uint64_t, 25> V;
  V zero() const { V z; z.fill(0); return z; }
  template <class F> V gather(const V& v, F src) const { V o; for (uint32_t l = 0; l < 25; l++) o[l] = v[src(l)]; return o; }
  template <class F> V map1(const V& a, F f) const { V o; for (uint32_t l = 0; l < 25; l++) o[l] = f(l, a[l]); return o; }
  template <class F> V map2(const V& a, const V& b, F f) const { V o; for (uint32_t l = 0; l < 25; l++) o[l] = f(l, a[l], b[l]); return o; }
  template <class F> V map3(const V& a, const V& b, const V& c, F f) const { V o; for (uint32_t l = 0; l < 25; l++) o[l] = f(l, a[l], b[l], c[l]); return o; }
  void scatter_bytes(const V& v, uint32_t pos, uint8_t* out, uint32_t n) const {
    for (uint32_t l = 0; l < 25; l++) for (uint32_t k = 0; k < 8; k++) { const uint32_t g = 8 * l + k; if (g >= pos && g < pos + n) out[g - pos] = (uint8_t)(v[l] >> (8 * k)); }
  }
};
#define CHECK(c) do { if (!(c)) { printf("FAIL %s line %d\n", #c, __LINE__); return 1; } } while (0)

int main() {
  std::mt19937_64 rng(2718);
  host_lanes ln;
  // the permutation
  for (int t = 0; t < 50; t++) {
    lasso::Keccak1600 k; host_lanes::V w;
    for (int i = 0; i < 25; i++) { k.A[i] = t == 0 ? 0 : rng(); w[i] = k.A[i]; }
    k.permute(); keccak_f1600_lanes(ln, w);
    for (int i = 0; i < 25; i++) CHECK(w[i] == k.A[i]);
  }
  // transcripts: the same random schedule of appends (labels and messages of every length class, incl. messages longer than the rate block) and
  // challenges through both implementations
  for (int t = 0; t < 40; t++) {
    const std::string proto = t % 2 ? "example" : "proof";
    lasso::Merlin ref(proto.c_str());
    strobe_lanes<host_lanes> dev; dev.init_merlin(ln, (const uint8_t*)proto.data(), (uint32_t)proto.size());
    for (int step = 0; step < 60; step++) {
      const char* labels[6] = {"a", "comm_poly_row_col_ops_val", "claim_eval_scalar_product", "challenge_nextround", "begin_append_vector", ""};
      const std::string label = labels[rng() % 6];
      const int kind = (int)(rng() % 4);
      if (kind < 3) {
        const size_t lens[8] = {0, 1, 32, 32, 33, 165, 166, 700};
        std::vector<uint8_t> msg(lens[rng() % 8]); for (auto& b : msg) b = (uint8_t)rng();
        ref.append_message(label.c_str(), msg.data(), msg.size());
        dev.append_message(ln, (const uint8_t*)label.data(), (uint32_t)label.size(), msg.data(), (uint32_t)msg.size());
      } else {
        const uint32_t n = (rng() % 3) ? 64 : (uint32_t)(1 + rng() % 400);
        std::vector<uint8_t> a(n), b(n);
        ref.challenge_bytes(label.c_str(), a.data(), n);
        dev.challenge_bytes(ln, (const uint8_t*)label.data(), (uint32_t)label.size(), b.data(), n);
        CHECK(a == b);
        if (n == 64) { const lasso::Sc want = lasso::Sc::from_wide_bytes(a.data()); const fr_t got = fr_from_wide_bytes(b.data()); CHECK(fr_eq(want.v, got)); }
      }
    }
    uint8_t a[32], b[32]; ref.challenge_bytes("end", a, 32); dev.challenge_bytes(ln, (const uint8_t*)"end", 3, b, 32);
    CHECK(memcmp(a, b, 32) == 0);
  }
  printf("OK\n");
  return 0;
}

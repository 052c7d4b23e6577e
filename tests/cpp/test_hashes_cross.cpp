// The product's Merlin (lasso_amd/host/hashes.hpp: unrolled Keccak-f, STROBE with bulk absorb / squeeze and the one-pass append of a framed record, Strobe::append_framed)
// against the oracle's independently written one (oracle/hashes.hpp: table-driven), on random schedules of appends and challenges: messages of 0..400 bytes, labels of
// 1..40 bytes, so that framed records start at every position of the 166-byte rate block, end before it, on it and after it.
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>
#include "../../lasso_amd/host/hashes.hpp"
#include "../../oracle/hashes.hpp"

static uint64_t st = 0x243F6A8885A308D3ull;
static uint64_t rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }

int main() {
  int checks = 0;
  for (int run = 0; run < 200; run++) {
    lasso::Merlin a("example"); orc::Transcript b("example");
    const int ops = 20 + (int)(rnd() % 200);
    for (int i = 0; i < ops; i++) {
      const size_t ll = 1 + rnd() % ((run & 1) ? 4 : 40);
      std::string label(ll, 'a'); for (auto& ch : label) ch = (char)('a' + rnd() % 26);
      const uint64_t kind = rnd() % 8;
      if (kind < 6) {
        const size_t n = (run % 3 == 0) ? 32 : (size_t)(rnd() % ((kind & 1) ? 64 : 400));   // every third run: the scalar-sized messages of the prover
        std::vector<uint8_t> msg(n); for (auto& x : msg) x = (uint8_t)rnd();
        a.append_message(label.c_str(), msg.data(), n); b.append_message(label.c_str(), msg.data(), n);
      } else {
        const size_t n = 1 + rnd() % 100; std::vector<uint8_t> x(n), y(n);
        a.challenge_bytes(label.c_str(), x.data(), n); b.challenge_bytes(label.c_str(), y.data(), n);
        if (memcmp(x.data(), y.data(), n)) { printf("FAIL run %d op %d: challenge differs\n", run, i); return 1; }
        checks++;
      }
    }
    uint8_t x[64], y[64]; a.challenge_bytes("end", x, 64); b.challenge_bytes("end", y, 64);
    if (memcmp(x, y, 64)) { printf("FAIL run %d: final challenge differs\n", run); return 1; }
    checks++;
  }
  printf("OK %d challenges compared\n", checks);
  return 0;
}

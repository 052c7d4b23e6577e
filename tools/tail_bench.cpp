// Per-round latency of the last rounds of a grand-product layer through the C ABI: one launch per round (lasso_sumcheck_cubic_eqw2_begin)
// against the resident tail kernel (lasso_sumcheck_cubic_tail_*).  Build:
//   g++ -O2 -std=c++17 -Iinclude -o tools/tail_bench tools/tail_bench.cpp -Llasso_amd -llasso_hip -Wl,-rpath,'$ORIGIN/../lasso_amd'
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "lasso_hip.h"
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHK(x) do { int32_t rc_ = (x); if (rc_) { printf("error %d at line %d: %s\n", rc_, __LINE__, lasso_last_error(ctx)); return 1; } } while (0)
int main(int argc, char** argv) {
  lasso_ctx* ctx = nullptr; if (lasso_ctx_create(0, &ctx)) { printf("no context\n"); return 1; }
  const uint32_t k = 2; const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 256;   // default: q = 128 pairs, 8 rounds + heads
  int rounds = 0; while (((size_t)1 << rounds) < n) rounds++;
  std::vector<lasso_fr> host(n); for (size_t i = 0; i < n; i++) { memset(&host[i], 0, sizeof(lasso_fr)); host[i].l[0] = 1000 + i; }
  lasso_fr *A[2], *B[2], *E;
  for (uint32_t c = 0; c < k; c++) { CHK(lasso_alloc(ctx, n * sizeof(lasso_fr), (void**)&A[c])); CHK(lasso_alloc(ctx, n * sizeof(lasso_fr), (void**)&B[c])); }
  CHK(lasso_alloc(ctx, n * sizeof(lasso_fr), (void**)&E));
  CHK(lasso_upload(ctx, E, host.data(), n * sizeof(lasso_fr)));
  lasso_fr r = host[3], out[8];
  const int REP = 300;
  for (int mode = 0; mode < 2; mode++) for (int pass = 0; pass < 2; pass++) {
    if (mode == 1 && n / 2 > lasso_sumcheck_tail_capacity()) continue;   // the resident kernel holds at most that many pairs per circuit
    double total = 0;
    for (int rep = 0; rep < REP; rep++) {
      for (uint32_t c = 0; c < k; c++) { CHK(lasso_upload(ctx, A[c], host.data(), n * sizeof(lasso_fr))); CHK(lasso_upload(ctx, B[c], host.data(), n * sizeof(lasso_fr))); }
      CHK(lasso_sync(ctx));
      double t0 = now();
      if (mode == 0) {
        size_t len = n;
        CHK(lasso_sumcheck_cubic_eqw2_begin(ctx, A, B, k, E, len, nullptr)); CHK(lasso_result_wait(ctx, out, 2 * k));
        for (int j = 1; j < rounds; j++) { CHK(lasso_sumcheck_cubic_eqw2_begin(ctx, A, B, k, E, len, &r)); CHK(lasso_result_wait(ctx, out, 2 * k)); len /= 2; }
        lasso_fr* ab[4] = {A[0], A[1], B[0], B[1]};
        CHK(lasso_bind_top(ctx, ab, 4, len, &r));
        CHK(lasso_read_heads(ctx, (const lasso_fr* const*)ab, 4, out));
      } else {
        CHK(lasso_sumcheck_cubic_tail_begin(ctx, A, B, k, E, n, nullptr)); CHK(lasso_result_wait(ctx, out, 2 * k));
        for (int j = 0; j < rounds; j++) { CHK(lasso_sumcheck_cubic_tail_next(ctx, &r)); CHK(lasso_result_wait(ctx, out, 2 * k)); }
      }
      total += now() - t0;
    }
    if (pass) printf("n = %zu: %s: %.1f us per layer (%d rounds + heads), %.2f us per hand-off\n", n, mode ? "resident tail kernel" : "one launch per round ", total / REP, rounds, total / REP / (rounds + 1));
  }
  lasso_ctx_destroy(ctx);
  return 0;
}

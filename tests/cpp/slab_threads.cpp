// Test-only harness: ONE proof sharded over `world` ranks (slab mode, include/lasso_prover.h lasso_host_set_comm) with the ranks run as threads of
// this process and the all-gather done through shared memory.  Linked with lasso_amd/host/prover_capi.cpp and the oracle's mock of the device ABI,
// so the whole sharded host logic (slabs, tails, row-commitment exchange) runs on the CPU.  Returns rank 0's commitment and proof after checking
// that every rank produced the same bytes.
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/lasso_prover.h"

namespace {
struct Shared {
  int world; std::mutex mu; std::condition_variable cv;
  std::vector<const void*> ptrs; int arrived = 0; long gen = 0; int left = 0; long gen2 = 0;
  size_t calls = 0, bytes = 0;
};
struct RankCtx { Shared* sh; int rank; };
int32_t allgather(void* user, const void* send, void* recv, size_t bytes) {
  RankCtx* u = (RankCtx*)user; Shared& S = *u->sh;
  std::unique_lock<std::mutex> lk(S.mu);
  S.ptrs[u->rank] = send;
  long g = S.gen;
  if (++S.arrived == S.world) { S.arrived = 0; S.gen++; S.calls++; S.bytes += bytes * S.world; S.cv.notify_all(); } else S.cv.wait(lk, [&] { return S.gen != g; });
  std::vector<const void*> ptrs = S.ptrs;
  lk.unlock();
  for (int r = 0; r < S.world; r++) memcpy((uint8_t*)recv + (size_t)r * bytes, ptrs[r], bytes);
  lk.lock();   // nobody may reuse its send buffer before everyone has copied
  long g2 = S.gen2;
  if (++S.left == S.world) { S.left = 0; S.gen2++; S.cv.notify_all(); } else S.cv.wait(lk, [&] { return S.gen2 != g2; });
  return 0;
}
}  // namespace

// shm_name != NULL: the ranks meet in the library's own shared-memory exchange (lasso_host_set_comm_shm, the transport bench.py's slab leg and an N-GPU run use) instead of
// the callback above; capacity: lasso_host_set_capacity; steps >= 1 proofs are timed after the first (whose bytes are returned).  peak_bytes / prover_peak_bytes (world entries
// each, may be NULL): every rank's device high-water mark over densify + commit + the proofs (lasso_host_mem_stats) and the most its prover had in use.
extern "C" int slab_prove_threads_ex(int world, const lasso_strategy* st, size_t num_memories, const uint64_t* idx, size_t n_lookups, const lasso_fr* r, size_t r_len, const char* shm_name,
                                     int capacity, int steps, uint8_t* comm_out, size_t comm_cap, size_t* comm_len, uint8_t* proof_out, size_t proof_cap, size_t* proof_len,
                                     size_t* n_collectives, size_t* collective_bytes, uint64_t* peak_bytes, uint64_t* prover_peak_bytes, double* ms_per_proof, char* err, size_t err_cap) {
  Shared sh; sh.world = world; sh.ptrs.assign(world, nullptr);
  std::vector<RankCtx> ctx(world);
  std::vector<std::vector<uint8_t>> comms(world), proofs(world);
  std::vector<std::string> errs(world);
  std::vector<double> ms(world, 0.0);
  const size_t s = [&] { size_t p = 1; while (p < n_lookups) p <<= 1; return p; }();
  auto worker = [&](int rk) {
    lasso_host* h = nullptr; lasso_host_gens* g = nullptr; lasso_host_dense* d = nullptr;
    auto fail = [&](const char* what) { errs[rk] = std::string(what) + ": " + lasso_host_last_error(); };
    ctx[rk] = RankCtx{&sh, rk};
    do {
      if (lasso_host_create(0, &h)) { fail("create"); break; }
      if (capacity && lasso_host_set_capacity(h, 1)) { fail("set_capacity"); break; }
      if (shm_name ? lasso_host_set_comm_shm(h, rk, world, shm_name) : lasso_host_set_comm(h, rk, world, allgather, &ctx[rk])) { fail("set_comm"); break; }
      if (lasso_host_gens_new(h, "gens_sparse_poly", st->c, s, num_memories, st->log_m, &g)) { fail("gens"); break; }
      if (lasso_host_densify(h, idx, n_lookups, st->c, st->log_m, &d)) { fail("densify"); break; }
      std::vector<uint8_t> buf(1 << 22); size_t len = 0;
      if (lasso_host_commit(d, g, buf.data(), buf.size(), &len)) { fail("commit"); break; }
      comms[rk].assign(buf.begin(), buf.begin() + len);
      if (lasso_host_prove(h, d, g, st, r, r_len, "example", "proof", buf.data(), buf.size(), &len)) { fail("prove"); break; }
      proofs[rk].assign(buf.begin(), buf.begin() + len);
      bool bad = false;
      const auto t0 = std::chrono::steady_clock::now();
      for (int it = 1; it < steps && !bad; it++) {
        if (lasso_host_prove(h, d, g, st, r, r_len, "example", "proof", buf.data(), buf.size(), &len)) { fail("prove (timed)"); bad = true; break; }
        if (len != proofs[rk].size() || memcmp(buf.data(), proofs[rk].data(), len) != 0) { errs[rk] = "a repeated proof differs from the first"; bad = true; }
      }
      if (bad) break;
      if (steps > 1) ms[rk] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (steps - 1);
      if (peak_bytes || prover_peak_bytes) {
        uint64_t live = 0, peak = 0, used = 0;
        if (lasso_host_mem_stats(h, &live, &peak, &used, 0)) { fail("mem_stats"); break; }
        if (peak_bytes) peak_bytes[rk] = peak; if (prover_peak_bytes) prover_peak_bytes[rk] = used;
      }
    } while (0);
    if (d) lasso_host_dense_free(d);
    if (g) lasso_host_gens_free(g);
    if (h) lasso_host_destroy(h);
  };
  std::vector<std::thread> th;
  for (int rk = 0; rk < world; rk++) th.emplace_back(worker, rk);
  for (auto& t : th) t.join();
  for (int rk = 0; rk < world; rk++) if (!errs[rk].empty()) { snprintf(err, err_cap, "rank %d: %s", rk, errs[rk].c_str()); return -1; }
  for (int rk = 1; rk < world; rk++) if (comms[rk] != comms[0] || proofs[rk] != proofs[0]) { snprintf(err, err_cap, "rank %d produced different bytes than rank 0", rk); return -2; }
  if (comms[0].size() > comm_cap || proofs[0].size() > proof_cap) { snprintf(err, err_cap, "output buffers too small"); return -3; }
  memcpy(comm_out, comms[0].data(), comms[0].size()); *comm_len = comms[0].size();
  memcpy(proof_out, proofs[0].data(), proofs[0].size()); *proof_len = proofs[0].size();
  if (n_collectives) *n_collectives = sh.calls;
  if (collective_bytes) *collective_bytes = sh.bytes;
  if (ms_per_proof) { double m = 0; for (double x : ms) m = x > m ? x : m; *ms_per_proof = m; }
  return 0;
}
extern "C" int slab_prove_threads(int world, const lasso_strategy* st, size_t num_memories, const uint64_t* idx, size_t n_lookups, const lasso_fr* r, size_t r_len, uint8_t* comm_out,
                                  size_t comm_cap, size_t* comm_len, uint8_t* proof_out, size_t proof_cap, size_t* proof_len, size_t* n_collectives, size_t* collective_bytes, char* err, size_t err_cap) {
  return slab_prove_threads_ex(world, st, num_memories, idx, n_lookups, r, r_len, nullptr, 0, 1, comm_out, comm_cap, comm_len, proof_out, proof_cap, proof_len, n_collectives, collective_bytes,
                               nullptr, nullptr, nullptr, err, err_cap);
}

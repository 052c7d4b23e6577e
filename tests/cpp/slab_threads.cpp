// Test-only harness: ONE proof sharded over `world` ranks (slab mode, include/lasso_prover.h lasso_host_set_comm) with the ranks run as threads of
// this process and the all-gather done through shared memory.  Linked with lasso_amd/host/prover_capi.cpp and the oracle's mock of the device ABI,
// so the whole sharded host logic (slabs, tails, row-commitment exchange) runs on the CPU.  Returns rank 0's commitment and proof after checking
// that every rank produced the same bytes.
#include <condition_variable>
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

extern "C" int slab_prove_threads(int world, const lasso_strategy* st, size_t num_memories, const uint64_t* idx, size_t n_lookups, const lasso_fr* r, size_t r_len, uint8_t* comm_out,
                                  size_t comm_cap, size_t* comm_len, uint8_t* proof_out, size_t proof_cap, size_t* proof_len, size_t* n_collectives, size_t* collective_bytes, char* err, size_t err_cap) {
  Shared sh; sh.world = world; sh.ptrs.assign(world, nullptr);
  std::vector<RankCtx> ctx(world);
  std::vector<std::vector<uint8_t>> comms(world), proofs(world);
  std::vector<std::string> errs(world);
  const size_t s = [&] { size_t p = 1; while (p < n_lookups) p <<= 1; return p; }();
  auto worker = [&](int rk) {
    lasso_host* h = nullptr; lasso_host_gens* g = nullptr; lasso_host_dense* d = nullptr;
    auto fail = [&](const char* what) { errs[rk] = std::string(what) + ": " + lasso_host_last_error(); };
    ctx[rk] = RankCtx{&sh, rk};
    do {
      if (lasso_host_create(0, &h)) { fail("create"); break; }
      if (lasso_host_set_comm(h, rk, world, allgather, &ctx[rk])) { fail("set_comm"); break; }
      if (lasso_host_gens_new(h, "gens_sparse_poly", st->c, s, num_memories, st->log_m, &g)) { fail("gens"); break; }
      if (lasso_host_densify(h, idx, n_lookups, st->c, st->log_m, &d)) { fail("densify"); break; }
      std::vector<uint8_t> buf(1 << 22); size_t len = 0;
      if (lasso_host_commit(d, g, buf.data(), buf.size(), &len)) { fail("commit"); break; }
      comms[rk].assign(buf.begin(), buf.begin() + len);
      if (lasso_host_prove(h, d, g, st, r, r_len, "example", "proof", buf.data(), buf.size(), &len)) { fail("prove"); break; }
      proofs[rk].assign(buf.begin(), buf.begin() + len);
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
  return 0;
}

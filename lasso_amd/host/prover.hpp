// C++ mirror of the reference's Rust host for the north-star path: the protocol, the Merlin transcript and every
// O(log n) scalar step stay here; every O(n) loop is a call into the device C ABI (include/lasso_hip.h).
// Names, argument meaning and transcript schedule follow the reference so the parity tests read like its own:
//   DensifiedRepresentation::{from_lookup_indices, commit}        src/lasso/densified.rs:22-96
//   SparsePolyCommitmentGens::new                                  src/lasso/surge.rs:32-58
//   SparsePolynomialEvaluationProof::prove                         src/lasso/surge.rs:119-211
//   MemoryCheckingProof / ProductLayerProof / HashLayerProof       src/lasso/memory_checking.rs:56-83, :674-731, :338-460
//   BatchedGrandProductArgument::prove, prove_cubic_batched        src/subprotocols/grand_product.rs:101-201, sumcheck.rs:27-135
//   prove_arbitrary                                                src/subprotocols/sumcheck.rs:150-260
//   CombinedTableEvalProof / PolyEvalProof / DotProductProofLog / BulletReductionProof
//                                                                  src/subtables/mod.rs:230-313, src/poly/dense_mlpoly.rs:302-359,
//                                                                  src/subprotocols/dot_product.rs:167-249, bullet.rs:40-154
// There is no CPU fallback: a failing device call throws (the reference panics).
#pragma once
#include <array>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "field_host.hpp"
#include "field52.hpp"
#include "hashes.hpp"
#include "../../include/lasso_prover.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace lasso {

// Wall-clock spans with the reference's `tracing` span names, printed when LASSO_TRACE=1 — so a run can be
// laid next to src/benches/*.log line by line.  Device work is asynchronous: a span closes after a stream sync only when tracing.
struct Trace {
  static bool on() { static const bool v = [] { const char* e = getenv("LASSO_TRACE"); return e && e[0] == '1'; }(); return v; }
  // LASSO_TRACE=3: device bytes per span instead (live at entry, the span's own high-water mark; lasso_mem_stats with reset, so an enclosing span reports what followed its last child)
  static bool mem() { static const bool v = [] { const char* e = getenv("LASSO_TRACE"); return e && e[0] == '3'; }(); return v; }
  const char* name; lasso_ctx* ctx; std::chrono::steady_clock::time_point t0; static int& depth() { static thread_local int d = 0; return d; }
  uint64_t w0 = 0; double wus0 = 0; uint64_t live0 = 0;
  Trace(const char* n, lasso_ctx* c) : name(n), ctx(c) {
    if (on()) { t0 = std::chrono::steady_clock::now(); depth()++; if (ctx) lasso_wait_stats(ctx, &w0, &wus0, 0); }
    if (mem() && ctx) { uint64_t pk = 0; lasso_mem_stats(ctx, &live0, &pk, 1); depth()++; }
  }
  ~Trace() {
    if (mem() && ctx) {
      uint64_t live = 0, pk = 0; lasso_mem_stats(ctx, &live, &pk, 1); depth()--;
      fprintf(stderr, "[mem] %*s%s: live at entry %.2f GB, peak inside %.2f GB, live at exit %.2f GB\n", 2 * depth(), "", name, live0 / 1e9, pk / 1e9, live / 1e9);
    }
    if (!on()) return;
    if (ctx) lasso_sync(ctx);
    depth()--;
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    uint64_t w1 = 0; double wus1 = 0; if (ctx) lasso_wait_stats(ctx, &w1, &wus1, 0);
    fprintf(stderr, "[trace] %*s%s: time.busy=%.3fms (device hand-offs %llu, host spinning %.3fms)\n", 2 * depth(), "", name, ms, (unsigned long long)(w1 - w0), (wus1 - wus0) * 1e-3);
  }
};

// LASSO_TRACE=2: host-side time buckets (where the host spends its share of a proof), printed with the SparsePoly.prove span
struct HostClock {
  static bool on() { static const bool v = [] { const char* e = getenv("LASSO_TRACE"); return e && e[0] == '2'; }(); return v; }
  static std::map<std::string, double>& buckets() { static thread_local std::map<std::string, double> b; return b; }
  const char* name; std::chrono::steady_clock::time_point t0;
  explicit HostClock(const char* n) : name(n) { if (on()) t0 = std::chrono::steady_clock::now(); }
  ~HostClock() { if (on()) buckets()[name] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
  static void dump() { if (!on()) return; for (auto& kv : buckets()) fprintf(stderr, "[host] %-28s %.3f ms\n", kv.first.c_str(), kv.second); buckets().clear(); }
};
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
#define LASSO_REQUIRE(c) do { if (!(c)) throw Error(std::string("lasso prover: requirement failed: ") + #c); } while (0)

#define LASSO_HOST_TOPS_MAX_PTRS 128   // lasso_read_runs takes up to LASSO_MAX_PTRS (136) arrays per call
inline size_t next_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }
inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
inline size_t ceil_log2(size_t n) { size_t k = 0; while (((size_t)1 << k) < n) k++; return k; }  // == Math::log_2 (utils/math.rs:27-35) and ark_std::log2

// ------------------------------------------------------------------ slab sharding of ONE proof over P GPUs
// Every length-n array is split by LOW index bits: rank g of P holds {i : i mod P == g} as a contiguous local array (local j <-> global j*P + g).
// bound_poly_var_top pairs i with i + n/2 and product-tree layers pair i with i + len/2; n/2 is a multiple of P while n/2 >= P, so both members of
// every pair live on the same rank and every kernel runs UNCHANGED on the local array (it is the polynomial restricted to the rank's low bits, which
// are the LAST variables bound).  What crosses ranks: per-round partial sums (a few field elements: all-gather + local sum), the partial row
// commitments of the Hyrax matrices (every rank holds the columns = g mod P of every row: all-gather of points, summed per row), the L*Z vectors of
// the openings, and P-element "tails" when a local array is down to one element and the remaining log2 P variables are bound on replicated copies.
// The transcript is replicated: every rank sees the same sums, derives the same challenges and emits the same proof bytes.
// The collective is a callback (host buffers, `bytes` per rank, rank order): torch.distributed all_gather (RCCL = "nccl", or gloo in CPU tests).
typedef int32_t (*lasso_allgather_fn)(void* user, const void* send, void* recv, size_t bytes);
struct Comm {
  size_t rank = 0, world = 1; lasso_allgather_fn fn = nullptr; void* user = nullptr;
  bool shared_device = false;   // another rank of this proof runs on the SAME physical device (found out once, Dev::note_shared_devices): nothing may then wait on the device for a peer's result
  bool sharded() const { return world > 1; }
  size_t log_world() const { size_t k = 0; while (((size_t)1 << k) < world) k++; return k; }
  void allgather(const void* send, void* recv, size_t bytes) const {
    if (world == 1) { memcpy(recv, send, bytes); return; }
    if (!fn || fn(user, send, recv, bytes) != 0) throw std::runtime_error("lasso prover: all-gather failed");
  }
  // every rank contributes v.size() partial sums; all ranks get the element-wise totals (added in rank order everywhere)
  void sum(std::vector<lasso_fr>& v) const {
    if (world == 1 || v.empty()) return;
    std::vector<lasso_fr> all(v.size() * world);
    allgather(v.data(), all.data(), v.size() * sizeof(lasso_fr));
    for (size_t i = 0; i < v.size(); i++) { Sc acc = Sc::zero(); for (size_t g = 0; g < world; g++) acc += Sc::from_abi(all[g * v.size() + i]); v[i] = acc.abi(); }
  }
  // eq factor of this rank's low index bits: the last log2(world) variables of `point` (r[0] is the top bit, eq_poly.rs:22-38)
  Sc eq_low(const ScVec& point) const {
    const size_t p = log_world(); Sc f = Sc::one();
    for (size_t b = 0; b < p; b++) { const Sc& rv = point[point.size() - 1 - b]; f *= ((rank >> b) & 1) ? rv : (Sc::one() - rv); }
    return f;
  }
};

// ------------------------------------------------------------------ device handle (RAII over the C ABI)
// Device buffers are recycled through a size-keyed pool: a proof allocates the same ~40 buffers every time and hipFree is a
// device-wide synchronisation (the reference pays Vec allocations inside prove as well; the pool only removes the driver calls).
class Dev {
  mutable std::multimap<size_t, void*> pool_;
  mutable std::map<void*, size_t> live_;
  mutable size_t in_use_ = 0, in_use_peak_ = 0;   // bytes the prover holds (live_), and their high-water mark: what a pool-free allocator would need
  int device_ = 0;
  mutable lasso_ctx* side_ = nullptr;

 public:
  lasso_ctx* ctx = nullptr;
  Comm comm;
  // Capacity mode (slab mode's purpose): the prover trades time for resident bytes — the read / write product trees are kept WITHOUT their leaf layers
  // (half of every tree: the fingerprints are recomputed strip by strip when the bottom layer's two streaming rounds need them, Prover::leaf_rounds).
  // Round 4 first tried the obvious thing, handing every released buffer straight back to the driver: measured on configs[3] it bought nothing at P <= 4 (the
  // high-water mark is the live set at the tree phase, not the pool) and cost 10x the proof time in hipFree / hipMalloc (profiles/r04_slab_peak_bytes_first_cut.json).
  // LASSO_CAPACITY=1 turns it on for every host; lasso_host_set_capacity sets it per host; default: off.
  bool capacity = [] { const char* e = getenv("LASSO_CAPACITY"); return e && e[0] == '1'; }();
  bool throughput = false;   // lasso_host_set_throughput_mode: several hosts prove concurrently on this GPU — nothing of this host is launched ahead of its challenge (LASSO_THROUGHPUT_AHEAD=1: launched ahead all the same — since round 5 the wait is one gate wave, not a kernel's worth of compute units; A/B switch)
  bool no_ahead() const { static const bool keep = [] { const char* e = getenv("LASSO_THROUGHPUT_AHEAD"); return e && e[0] == '1'; }(); return throughput && !keep; }
  explicit Dev(int device) : device_(device) {
    if (lasso_ctx_create(device, &ctx) != 0) throw Error(std::string("lasso_ctx_create: ") + lasso_last_error(nullptr));
    const char* e = getenv("LASSO_SIDE_STREAM");
    if (!(e && e[0] == '0')) (void)side();   // created up front: a context costs ~9 ms, which must not land inside the first proof
  }
  ~Dev() { if (side_) lasso_ctx_destroy(side_); if (ctx) { for (auto& kv : pool_) lasso_free(ctx, kv.second); for (auto& kv : live_) lasso_free(ctx, kv.first); lasso_ctx_destroy(ctx); } }
  // A second context on the same device (own stream, scratch and result buffer): streaming work that does not depend on the transcript is
  // issued there while the main context is inside a latency-bound phase (Prover::prep_open).  Buffers it touches must outlive its work:
  // the pool's stream-ordered reuse argument below holds per context, so lasso_sync(side()) precedes their release.
  lasso_ctx* side() const { if (!side_ && lasso_ctx_create_background(device_, 1, &side_) != 0) throw Error(std::string("lasso_ctx_create (side): ") + lasso_last_error(nullptr)); return side_; }
  void chk_side(int32_t rc, const char* what) const { if (rc != 0) throw Error(std::string(what) + " failed on the side context (" + std::to_string(rc) + "): " + lasso_last_error(side_)); }
  // slab mode: do two ranks of the proof share a physical device?  One all-gather of the 16-byte device identifiers, the first time the question is asked after the
  // communicator was set (collective: every rank asks at the same point — the first proof's first slab sumcheck).
  bool ranks_share_a_device() const {
    if (!comm.sharded()) return false;
    if (!shared_known_) {
      std::vector<uint8_t> mine(16, 0), all(16 * comm.world, 0);
      chk(lasso_ctx_device_uuid(ctx, mine.data()), "lasso_ctx_device_uuid");
      comm.allgather(mine.data(), all.data(), 16);
      bool shared = false;
      for (size_t a = 0; a < comm.world; a++) for (size_t b = a + 1; b < comm.world; b++) if (memcmp(&all[16 * a], &all[16 * b], 16) == 0) shared = true;
      const_cast<Comm&>(comm).shared_device = shared; shared_known_ = true;
    }
    return comm.shared_device;
  }
  mutable bool shared_known_ = false;
  Dev(const Dev&) = delete; Dev& operator=(const Dev&) = delete;
  void chk(int32_t rc, const char* what) const { if (rc != 0) throw Error(std::string(what) + " failed (" + std::to_string(rc) + "): " + lasso_last_error(ctx)); }
  // error unwinding: leave both contexts usable (a resident kernel may be waiting for a challenge, a deferred result may be uncollected)
  void abort_all() const noexcept { if (ctx) (void)lasso_abort(ctx); if (side_) (void)lasso_abort(side_); }
  void* alloc_bytes(size_t bytes) const {
    if (!bytes) bytes = 1;
    auto it = pool_.find(bytes);
    void* p = nullptr;
    if (it != pool_.end()) { p = it->second; pool_.erase(it); }
    else {
      // capacity mode: a buffer the pool cannot serve must not land ON TOP of what is parked there — parked buffers go back to the driver first, largest first, until
      // they make up the request (live + parked bytes then do not grow past the larger of the two proofs' own needs)
      if (capacity && bytes >= ((size_t)64 << 20)) {   // small requests never evict (a 2 MB miss that returned a parked gigabyte started a miss / evict cycle that repeated every proof)
        auto fit = pool_.lower_bound(bytes);      // the smallest parked buffer that covers the request, else the largest ones until they add up
        // (a buffer the driver refuses to take back — lasso_free returns LASSO_ERR_INVALID while a launch waits for its challenge — stays parked: ADVICE r5)
        if (fit != pool_.end()) { if (lasso_free(ctx, fit->second) == 0) pool_.erase(fit); }
        else { size_t freed = 0; while (freed < bytes && !pool_.empty()) { auto big = std::prev(pool_.end()); if (lasso_free(ctx, big->second) != 0) break; freed += big->first; pool_.erase(big); } }
      }
      int32_t rc = lasso_alloc(ctx, bytes, &p);
      if (rc == LASSO_ERR_OOM && !pool_.empty()) { trim(); rc = lasso_alloc(ctx, bytes, &p); }   // the pool holds memory nobody uses: give it back before giving up
      chk(rc, "lasso_alloc");
    }
    live_[p] = bytes; in_use_ += bytes; if (in_use_ > in_use_peak_) in_use_peak_ = in_use_;
    return p;
  }
  lasso_fr* alloc_fr(size_t n) const { return (lasso_fr*)alloc_bytes(n * sizeof(lasso_fr)); }
  uint32_t* alloc_u32(size_t n) const { return (uint32_t*)alloc_bytes(n * 4); }
  // stream-ordered reuse: every kernel of this context runs on one stream, so a recycled buffer cannot be overtaken
  void free(void* p) const {
    if (!p) return;
    auto it = live_.find(p); if (it == live_.end()) return;
    const size_t bytes = it->second; live_.erase(it); in_use_ -= bytes;
    pool_.emplace(bytes, p);
  }
  // a one-off buffer (the uploaded index array of densify: 8 C s bytes that nothing of that size will ever want again) goes back to the driver, not into the pool
  void release(void* p) const {
    if (!p) return; auto it = live_.find(p); if (it == live_.end()) return;
    const size_t bytes = it->second; in_use_ -= bytes; live_.erase(it);
    if (lasso_free(ctx, p) != 0) pool_.emplace(bytes, p);   // refused inside a launched-ahead window: parked instead of dropped (the pointer must not be lost; the next trim returns it)
  }
  // LASSO_TRACE=3: what the prover holds right now, by size
  void dump_live(const char* tag) const {
    std::map<size_t, size_t> by; for (auto& kv : live_) by[kv.second]++;
    size_t pooled = 0; for (auto& kv : pool_) pooled += kv.first;
    fprintf(stderr, "[mem] %s: in use %.2f GB, parked in the pool %.2f GB;", tag, in_use_ / 1e9, pooled / 1e9);
    for (auto it = by.rbegin(); it != by.rend() && it->first >= ((size_t)1 << 20); ++it) fprintf(stderr, " %zu x %.3f GB", it->second, it->first / 1e9);
    fprintf(stderr, "\n");
  }
  // hand every pooled buffer back to the driver
  void trim() const { for (auto it = pool_.begin(); it != pool_.end();) { if (lasso_free(ctx, it->second) == 0) it = pool_.erase(it); else ++it; } (void)lasso_trim(ctx); }   // ... and the context's grown scratch buffer; what the driver refuses stays parked
  // the same, except up to `count` parked buffers of exactly `bytes` (what the caller is about to allocate)
  // ... and smaller ones (they add up to little; hipFree / hipMalloc of gigabytes per proof cost more than the proof: measured 600 ms at configs[3])
  void trim_keep(size_t bytes, size_t count) const {
    for (auto it = pool_.begin(); it != pool_.end();) {
      if (it->first < bytes || (it->first == bytes && count)) { if (it->first == bytes) count--; ++it; continue; }
      if (lasso_free(ctx, it->second) == 0) it = pool_.erase(it); else ++it;
    }
  }
  // device bytes held through this host's contexts now / at most (lasso_mem_stats of the main and the side context), and what the prover itself held at most
  void mem_stats(uint64_t* live, uint64_t* peak, uint64_t* in_use_peak, bool reset) const {
    uint64_t l = 0, p = 0, l2 = 0, p2 = 0;
    chk(lasso_mem_stats(ctx, &l, &p, reset ? 1 : 0), "lasso_mem_stats");
    if (side_) chk_side(lasso_mem_stats(side_, &l2, &p2, reset ? 1 : 0), "lasso_mem_stats");
    if (live) *live = l + l2;
    if (peak) *peak = p + p2;
    if (in_use_peak) *in_use_peak = in_use_peak_;
    if (reset) in_use_peak_ = in_use_;
  }
};
// owning device buffer of field elements
struct DBuf {
  const Dev* dev = nullptr; lasso_fr* p = nullptr; size_t n = 0;
  DBuf() {}
  DBuf(const Dev& d, size_t n_) : dev(&d), p(d.alloc_fr(n_)), n(n_) {}
  DBuf(DBuf&& o) noexcept : dev(o.dev), p(o.p), n(o.n) { o.p = nullptr; }
  DBuf& operator=(DBuf&& o) noexcept { if (this != &o) { reset(); dev = o.dev; p = o.p; n = o.n; o.p = nullptr; } return *this; }
  DBuf(const DBuf&) = delete; DBuf& operator=(const DBuf&) = delete;
  void reset() { if (p && dev) dev->free(p); p = nullptr; n = 0; }
  void release() { if (p && dev) dev->release(p); p = nullptr; n = 0; }   // straight back to the driver (Dev::release)
  ~DBuf() { try { reset(); } catch (...) {} }
};
struct DBufU64 {   // read-only upload of a host u64 array
  const Dev* dev = nullptr; uint64_t* p = nullptr; size_t n = 0;
  DBufU64(const Dev& d, const uint64_t* h, size_t n_) : dev(&d), p((uint64_t*)d.alloc_bytes((n_ ? n_ : 1) * 8)), n(n_) { if (n) d.chk(lasso_upload(d.ctx, p, h, n * 8), "lasso_upload"); }
  DBufU64(const DBufU64&) = delete; DBufU64& operator=(const DBufU64&) = delete;
  ~DBufU64() { try { if (p && dev) dev->release(p); } catch (...) {} }   // only densify's index upload uses this type: not worth keeping
};
struct DBufU32 {
  const Dev* dev = nullptr; uint32_t* p = nullptr; size_t n = 0;
  DBufU32() {}
  DBufU32(const Dev& d, size_t n_) : dev(&d), p(d.alloc_u32(n_ ? n_ : 1)), n(n_) {}
  DBufU32(const Dev& d, const std::vector<uint32_t>& h) : dev(&d), p(d.alloc_u32(h.size() ? h.size() : 1)), n(h.size()) { if (n) d.chk(lasso_upload(d.ctx, p, h.data(), n * 4), "lasso_upload"); }
  DBufU32(DBufU32&& o) noexcept : dev(o.dev), p(o.p), n(o.n) { o.p = nullptr; }
  DBufU32& operator=(DBufU32&& o) noexcept { if (this != &o) { if (p && dev) dev->free(p); dev = o.dev; p = o.p; n = o.n; o.p = nullptr; } return *this; }
  DBufU32(const DBufU32&) = delete; DBufU32& operator=(const DBufU32&) = delete;
  ~DBufU32() { try { if (p && dev) dev->free(p); } catch (...) {} }
};

// ------------------------------------------------------------------ transcript (utils/transcript.rs:6-72)
// Everything the reference does to a merlin::Transcript is one of two calls: append_message (append_u64 = its 8 little-endian bytes, the scalar / point / vector forms
// = messages built from ark-serialize's bytes, :20-62) and challenge_bytes (:64-72).  The transcript is therefore either the library's own Merlin (built from a label: the
// harness's `Transcript::new(b"example")`) or a caller's LIVE transcript behind those two callbacks (include/lasso_prover.h lasso_transcript_vtbl: surge.rs:119-125 takes
// `&mut Transcript`, whatever state it already holds) — the protocol code below cannot tell the difference.
class ProofTranscript {
  Merlin m; const lasso_transcript_vtbl* vt = nullptr; void* user = nullptr;
  void app(const char* label, const void* msg, size_t n) { if (vt) vt->append_message(user, (const uint8_t*)label, strlen(label), (const uint8_t*)msg, n); else m.append_message(label, msg, n); }

 public:
  explicit ProofTranscript(const char* label) : m(label) {}
  ProofTranscript(const lasso_transcript_vtbl* v, void* u) : m("(external transcript)"), vt(v), user(u) { if (!v || !v->append_message || !v->challenge_bytes) throw std::runtime_error("lasso prover: transcript callbacks missing"); }
  void append_message(const char* label, const char* msg) { app(label, msg, strlen(msg)); }
  void append_protocol_name(const char* name) { app("protocol-name", name, strlen(name)); }
  void append_u64(const char* label, uint64_t x) { uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i)); app(label, b, 8); }   // merlin::Transcript::append_u64 = encode_u64 little-endian
  void append_scalar(const char* label, const Sc& s) { uint8_t b[32]; s.to_bytes(b); app(label, b, 32); }
  // the same with the scalars already serialised (32 canonical bytes each, e.g. by lasso_fr_to_bytes)
  void append_scalars_bytes(const char* label, const std::vector<uint8_t>& b) { append_message(label, "begin_append_vector"); for (size_t i = 0; i + 32 <= b.size(); i += 32) app(label, &b[i], 32); append_message(label, "end_append_vector"); }
  void append_scalars(const char* label, const ScVec& v) { append_message(label, "begin_append_vector"); for (auto& s : v) append_scalar(label, s); append_message(label, "end_append_vector"); }
  void append_point_bytes(const char* label, const uint8_t b[32]) { app(label, b, 32); }
  Sc challenge_scalar(const char* label) {
    uint8_t b[64];
    if (vt) vt->challenge_bytes(user, (const uint8_t*)label, strlen(label), b, 64); else m.challenge_bytes(label, b, 64);
    return Sc::from_wide_bytes(b);
  }
  ScVec challenge_vector(const char* label, size_t n) { ScVec v; for (size_t i = 0; i < n; i++) v.push_back(challenge_scalar(label)); return v; }
};
// ark-ff Fp::rand on the test RNG (first draw): limbs taken as the Montgomery representation, top 3 bits masked, rejection
inline Sc fr_rand(ChaChaRng& rng) {
  for (;;) {
    uint64_t l[4]; for (int i = 0; i < 4; i++) l[i] = rng.next_u64();
#ifdef LASSO_BN254
    l[3] &= (~(uint64_t)0) >> 2;   // 254-bit modulus
#else
    l[3] &= (~(uint64_t)0) >> 3;
#endif
    fr_t t; memcpy(t.v, l, 32);
    if (!fr_geq_p(t.v)) { Sc s; s.v = t; return s; }
  }
}
class RandomTape {  // utils/random.rs:9-39
  ProofTranscript tape;

 public:
  explicit RandomTape(const char* name) : tape(name) { ChaChaRng prng = ChaChaRng::test_rng(); tape.append_scalar("init_randomness", fr_rand(prng)); }
  // a caller's live RandomTape: its inner transcript (already initialised by RandomTape::new on the caller's side, possibly already drawn from) behind the two callbacks
  RandomTape(const lasso_transcript_vtbl* v, void* u) : tape(v, u) {}
  Sc random_scalar(const char* label) { return tape.challenge_scalar(label); }
  ScVec random_vector(const char* label, size_t n) { return tape.challenge_vector(label, n); }
};

// ------------------------------------------------------------------ generators (poly/commitments.rs:22-44, dot_product.rs:139-150, dense_mlpoly.rs:34-45)
#ifdef LASSO_BN254
inline bool fq_sqrt(const fq_t& a, fq_t& out) {  // q = 3 (mod 4): a^((q+1)/4)
  const uint32_t e[8] = {0xb61f3f52u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
  const fq_t r = fq_pow(a, e);
  if (fq_eq(fq_sqr(r), a)) { out = r; return true; }
  return false;
}
inline bool canonical_less(const fq_t& a, const fq_t& b) { fq_t x = fq_to_canonical(a), y = fq_to_canonical(b); for (int i = 7; i >= 0; i--) if (x.v[i] != y.v[i]) return x.v[i] < y.v[i]; return false; }
// ark-ec `Projective::rand` for a short Weierstrass curve: x <- Fq::rand (limbs taken as the Montgomery representation, top 2 bits masked,
// rejection), bool, y from x (smaller or larger root); cofactor 1
inline Pt point_rand(ChaChaRng& rng) {
  for (;;) {
    uint64_t l[4]; for (int i = 0; i < 4; i++) l[i] = rng.next_u64();
    l[3] &= (~(uint64_t)0) >> 2;
    fq_t x; memcpy(x.v, l, 32);
    if (fq_geq_p(x.v)) continue;
    bool greatest = ((int32_t)rng.next_u32()) < 0;
    fq_t y;
    if (!fq_sqrt(fq_add(fq_mul(fq_sqr(x), x), fq_from_u64(3)), y)) continue;
    fq_t ny = fq_neg(y), ys, yl;
    if (canonical_less(ny, y)) { ys = ny; yl = y; } else { ys = y; yl = ny; }
    return Pt::from_affine_plain(x, greatest ? yl : ys);
  }
}
inline void compress_generator(uint8_t out[32]) { compress_affine(fq_from_u64(1), fq_from_u64(2), out); }   // G1 generator (1, 2)
#else
inline bool fq_sqrt(const fq_t& a, fq_t& out) {  // p = 5 (mod 8)
  const uint32_t e[8] = {0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x0fffffffu};  // (p+3)/8
  fq_t r = fq_pow(a, e);
  if (fq_eq(fq_sqr(r), a)) { out = r; return true; }
  const fq_t sqrtm1 = fq_from_limbs(0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u);
  r = fq_mul(r, sqrtm1);
  if (fq_eq(fq_sqr(r), a)) { out = r; return true; }
  return false;
}
inline bool canonical_less(const fq_t& a, const fq_t& b) { fq_t x = fq_canonical(a), y = fq_canonical(b); for (int i = 7; i >= 0; i--) if (x.v[i] != y.v[i]) return x.v[i] < y.v[i]; return false; }
// ark-ec `Projective::rand` for a twisted Edwards curve: y <- Fq::rand, bool, x from y (smaller or larger root), times the cofactor
inline Pt point_rand(ChaChaRng& rng) {
  for (;;) {
    uint64_t l[4]; for (int i = 0; i < 4; i++) l[i] = rng.next_u64();
    l[3] &= (~(uint64_t)0) >> 1;
    fq_t ym; memcpy(ym.v, l, 32);
    {  // rejection: limbs (Montgomery representation) must be < p
      const uint32_t P[8] = {0xffffffedu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x7fffffffu};
      bool lt = false; for (int i = 7; i >= 0; i--) if (ym.v[i] != P[i]) { lt = ym.v[i] < P[i]; break; }
      if (!lt) continue;
    }
    bool greatest = ((int32_t)rng.next_u32()) < 0;
    fq_t y = fq_from_mont(ym), y2 = fq_sqr(y);
    fq_t den = fq_sub(fq_neg(fq_one()), fq_mul(fq_d(), y2));   // a - d*y^2, a = -1
    if (fq_is_zero(den)) continue;
    fq_t x2 = fq_mul(fq_sub(fq_one(), y2), fq_inv(den)), x;
    if (!fq_sqrt(x2, x)) continue;
    fq_t nx = fq_neg(x), xs, xl;
    if (canonical_less(nx, x)) { xs = nx; xl = x; } else { xs = x; xl = nx; }
    Pt p = Pt::from_affine_plain(greatest ? xl : xs, y);
    return p.dbl().dbl().dbl();
  }
}
inline void compress_generator(uint8_t out[32]) {
  fq_t gx = fq_from_limbs(0x8f25d51au, 0xc9562d60u, 0x9525a7b2u, 0x692cc760u, 0xfdd6dc5cu, 0xc0a4e231u, 0xcd6e53feu, 0x216936d3u);
  fq_t gy = fq_from_limbs(0x66666658u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u);
  compress_affine(gx, gy, out);
}
#endif  // LASSO_BN254
// The generator stream for one label: MultiCommitGens::new(n, label) = first n points as G, point n as h.
struct GenStream {
  std::vector<Pt> pts;
  GenStream(const char* label, size_t count) {
    Shake256 sh; sh.update(label, strlen(label));
    uint8_t buf[32]; compress_generator(buf); sh.update(buf, 32);
    uint8_t seed[32]; sh.read(seed, 32);
    ChaChaRng rng(seed, 20);
    for (size_t i = 0; i < count; i++) pts.push_back(point_rand(rng));
  }
};
// PolyCommitmentGens for one polynomial size: gens_n = {G[0..n), h}, gens_1 = {G[n], h} with h = stream[n+1]  (DotProductProofGens::new(n) =
// MultiCommitGens::new(n+1).split_at(n)); the device table holds [G_0..G_{n-1}, Q = G_n, h] so one MSM covers G, Q and h.
struct PolyCommitmentGens {
  size_t n = 0; Pt Q, h; FixedBase Qmul, hmul; std::vector<lasso_affine> affine; lasso_bases* bases = nullptr; const Dev* dev = nullptr;
  lasso_bases* bases_slab = nullptr;   // slab mode: the generators G_{j*P + rank}, j < n/P (this rank's columns of every Hyrax row), then Q, h
  bool slab_open = false;              // slab mode: every rank can run its share of the opening's MSMs (agreed at construction)
  PolyCommitmentGens() {}
  // From the generator stream of a label (PolyCommitmentGens::new, dense_mlpoly.rs:38-45): the first n + 2 points, normalised with one batch inversion
  PolyCommitmentGens(const Dev& d, const GenStream& gs, size_t num_vars) {
    const size_t nn = (size_t)1 << (num_vars - num_vars / 2);   // right = ell - ell/2 (eq_poly.rs:40-42)
    LASSO_REQUIRE(gs.pts.size() >= nn + 2);
    // affine Montgomery limbs for the ABI, one batch inversion
    std::vector<fq_t> pre(nn + 2); fq_t acc = fq_one();
    for (size_t i = 0; i < nn + 2; i++) { pre[i] = acc; acc = fq_mul(acc, gs.pts[i].p.Z); }
    fq_t inv = fq_inv(acc); std::vector<lasso_affine> aff(nn + 2);
    for (size_t i = nn + 2; i-- > 0;) {
      fq_t zi = fq_mul(inv, pre[i]); inv = fq_mul(inv, gs.pts[i].p.Z);
      fq_t x = fq_to_mont(fq_mul(gs.pts[i].p.X, zi)), y = fq_to_mont(fq_mul(gs.pts[i].p.Y, zi));
      memcpy(aff[i].x, x.v, 32); memcpy(aff[i].y, y.v, 32);
    }
    init(d, std::move(aff), nn);
  }
  // From the CALLER's points — what SparsePolynomialEvaluationProof::prove is handed (surge.rs:119-125 `gens: &SparsePolyCommitmentGens<G>`): the n + 2 points
  // [gens.gens_n.G[0..n), gens.gens_1.G[0], gens.gens_n.h] of one PolyCommitmentGens (dense_mlpoly.rs:34-45, dot_product.rs:139-150, commitments.rs:15-19), affine, ark-ff's
  // Montgomery limbs as CurveGroup::normalize_batch yields them (commitments.rs:87).  Nothing is derived: whatever valid points the caller holds are the generators.
  // num_vars == 0: any power of two (the shim building the object for `commit`, which does not know the strategy's NUM_MEMORIES yet; a wrong size then fails where the reference's
  // does: at use, batch_commit's assert_eq!(gens_n.n, inputs.len()), commitments.rs:85)
  PolyCommitmentGens(const Dev& d, const lasso_affine* pts, size_t count, size_t num_vars) {
    const size_t nn = num_vars ? (size_t)1 << (num_vars - num_vars / 2) : (count >= 3 ? count - 2 : 0);
    if (!pts || count != nn + 2 || !is_pow2(nn)) throw Error("generators: a polynomial of " + std::to_string(num_vars) + " variables needs " + std::to_string(nn + 2) + " points (G[0.." + std::to_string(nn) + "), gens_1.G[0], h), got " + std::to_string(count));
    init(d, std::vector<lasso_affine>(pts, pts + count), nn);
  }
 private:
  void init(const Dev& d, std::vector<lasso_affine>&& aff, size_t nn) {
    dev = &d; n = nn; affine = std::move(aff);
    auto from_aff = [](const lasso_affine& a) { fq_t x, y; memcpy(x.v, a.x, 32); memcpy(y.v, a.y, 32); return Pt::from_affine_plain(fq_from_mont(x), fq_from_mont(y)); };
    Q = from_aff(affine[n]); h = from_aff(affine[n + 1]); Qmul = FixedBase(Q); hmul = FixedBase(h);
    // the byte-multiple tables (459 KB per generator, the openings' MSMs at half the additions) only where they are read: not for the full-width set in slab mode (the rank's
    // residue class serves the openings), not at all in capacity mode
    d.chk(lasso_bases_create_opt(d.ctx, affine.data(), n + 2, (!d.comm.sharded() && !d.capacity) ? 1 : 0, &bases), "lasso_bases_create");
    if (d.comm.sharded()) {
      const size_t P = d.comm.world; LASSO_REQUIRE(n >= P);
      // this rank's residue class of the generators, then Q and h: the same table serves the partial row commitments (first n/P entries) and the rank's share of the
      // opening's MSMs (lasso_bullet_round_slab / lasso_msm_dev_slab, which also need Q and h)
      std::vector<lasso_affine> sub(n / P + 2); for (size_t j = 0; j < n / P; j++) sub[j] = affine[j * P + d.comm.rank];
      sub[n / P] = affine[n]; sub[n / P + 1] = affine[n + 1];
      d.chk(lasso_bases_create_opt(d.ctx, sub.data(), sub.size(), d.capacity ? 0 : 1, &bases_slab), "lasso_bases_create");
      // the sharded opening is used only if EVERY rank has the table it needs (an allocation that failed on one rank must not split the ranks between two protocols)
      static const bool off = [] { const char* e = getenv("LASSO_SLAB_OPEN"); return e && e[0] == '0'; }();
      std::vector<uint8_t> all(P, 0); const uint8_t mine = (!off && lasso_bases_has_direct(bases_slab) == 1) ? 1 : 0;
      d.comm.allgather(&mine, all.data(), 1);
      slab_open = true; for (uint8_t v : all) slab_open = slab_open && v;
    }
  }
 public:
  PolyCommitmentGens(PolyCommitmentGens&& o) noexcept { *this = std::move(o); }
  PolyCommitmentGens& operator=(PolyCommitmentGens&& o) noexcept { std::swap(n, o.n); std::swap(Q, o.Q); std::swap(h, o.h); std::swap(Qmul, o.Qmul); std::swap(hmul, o.hmul); affine.swap(o.affine); std::swap(bases, o.bases); std::swap(bases_slab, o.bases_slab); std::swap(slab_open, o.slab_open); std::swap(dev, o.dev); return *this; }
  ~PolyCommitmentGens() { if (bases && dev) lasso_bases_destroy(dev->ctx, bases); if (bases_slab && dev) lasso_bases_destroy(dev->ctx, bases_slab); }
};
struct SparsePolyCommitmentGens {  // surge.rs:25-59
  PolyCommitmentGens gens_combined_l_variate, gens_combined_log_m_variate, gens_derefs;
  static void num_vars(size_t c, size_t s, size_t num_memories, size_t log_m, size_t& nv_l, size_t& nv_m, size_t& nv_d) {   // surge.rs:39-47
    nv_l = ceil_log2(next_pow2(2 * c * s)); nv_m = ceil_log2(next_pow2(c)) + log_m; nv_d = ceil_log2(next_pow2(num_memories * s));
  }
  // SparsePolyCommitmentGens::new(label, c, s, num_memories, log_m) (surge.rs:32-58): a convenience for callers without generators of their own (C, Python, the bench)
  SparsePolyCommitmentGens(const Dev& d, const char* label, size_t c, size_t s, size_t num_memories, size_t log_m) {
    size_t nv_l, nv_m, nv_d; num_vars(c, s, num_memories, log_m, nv_l, nv_m, nv_d);
    size_t mx = std::max(nv_l, std::max(nv_m, nv_d));
    GenStream gs(label, ((size_t)1 << (mx - mx / 2)) + 2);   // the three sets share one label => one stream, three prefixes
    gens_combined_l_variate = PolyCommitmentGens(d, gs, nv_l);
    gens_combined_log_m_variate = PolyCommitmentGens(d, gs, nv_m);
    gens_derefs = PolyCommitmentGens(d, gs, nv_d);
  }
  // the caller's generators, set by set (lasso_host_gens_from_points): what `prove` receives in the reference (surge.rs:119-125)
  SparsePolyCommitmentGens(const Dev& d, size_t c, size_t s, size_t num_memories, size_t log_m, const lasso_affine* l_variate, size_t n_l, const lasso_affine* log_m_variate, size_t n_m,
                           const lasso_affine* derefs, size_t n_d) {
    size_t nv_l, nv_m, nv_d; num_vars(c, s, num_memories ? num_memories : 1, log_m, nv_l, nv_m, nv_d);
    if (!num_memories) nv_d = 0;   // strategy not known yet: the derefs set only has to be a power of two (+ 2)
    gens_combined_l_variate = PolyCommitmentGens(d, l_variate, n_l, nv_l);
    gens_combined_log_m_variate = PolyCommitmentGens(d, log_m_variate, n_m, nv_m);
    gens_derefs = PolyCommitmentGens(d, derefs, n_d, nv_d);
  }
  const PolyCommitmentGens& set(int which) const { if (which == 0) return gens_combined_l_variate; if (which == 1) return gens_combined_log_m_variate; if (which == 2) return gens_derefs; throw Error("generator set index must be 0 (l-variate), 1 (log_m-variate) or 2 (derefs)"); }
};

// ------------------------------------------------------------------ strategies (host side of subtables/*.rs)
struct Strategy {
  lasso_strategy abi;
  Strategy(int kind, uint32_t c, uint32_t log_m, uint32_t log_r) { abi.kind = kind; abi.c = c; abi.log_m = log_m; abi.log_r = log_r; }
  size_t C() const { return abi.c; }
  size_t M() const { return (size_t)1 << abi.log_m; }
  bool spark() const { return abi.kind == LASSO_SPARK_UNCONFIRMED; }   // NOT in the reference snapshot: see include/lasso_hip.h lasso_strategy_kind
  size_t num_subtables() const { return abi.kind == LASSO_LT ? 2 : abi.kind == LASSO_RANGE ? 3 : spark() ? C() : 1; }
  size_t num_memories() const { return abi.kind == LASSO_LT ? 2 * C() : C(); }
  bool linear() const { return abi.kind != LASSO_LT && !spark(); }   // g = sum_k 2^(k*inc) E_k: and.rs:45-53, range_check.rs:78-86
  bool integer_tables() const { return !spark(); }                  // every subtable of the snapshot holds small integers (u32 on the device); Spark's hold field elements
  ScVec weights() const { ScVec w; size_t inc = abi.kind == LASSO_RANGE ? abi.log_m : abi.log_m / 2; for (size_t i = 0; i < num_memories(); i++) { LASSO_REQUIRE(i * inc < 64); w.push_back(Sc::from_u64((uint64_t)1 << (i * inc))); } return w; }
  size_t sumcheck_poly_degree() const { return (abi.kind == LASSO_LT || spark() ? C() : 1) + 1; }
  size_t memory_to_subtable_index(size_t i) const {
    if (abi.kind == LASSO_RANGE) { size_t lm = abi.log_m; if (i * lm > abi.log_r) return 2; return ((i + 1) * lm > abi.log_r) ? 1 : 0; }  // range_check.rs:62-69
    return i % num_subtables();                                                                                                        // subtables/mod.rs:64-68 (Spark: i)
  }
  size_t memory_to_dimension_index(size_t i) const { return abi.kind == LASSO_RANGE || spark() ? i : i / num_subtables(); }             // mod.rs:70-74, range_check.rs:71-73
  // Spark (unconfirmed): subtable i = EqPolynomial(tau_i).evals(); the snapshot's trait has no per-proof table parameter, so tau is fixed by the strategy:
  // C * log2(M) draws of F::rand from a fresh ark_std::test_rng(), tau_i = draws [i log M, (i + 1) log M)
  std::vector<ScVec> spark_point() const;
  // materialize_subtables: every table of the reference holds small integers, so the host builds u32 and the device lifts to Fr
  // largest entry of any subtable (bounds the scalars of E's commitment): l op r < 2^(log_m / 2); LT / EQ are bits; range tables hold indices
  uint32_t max_table_value() const {
    if (abi.kind == LASSO_LT) return 1;
    if (abi.kind == LASSO_RANGE) return (uint32_t)(M() - 1);
    return (uint32_t)(((size_t)1 << (abi.log_m / 2)) - 1);
  }
  std::vector<std::vector<uint32_t>> materialize_subtables() const {
    size_t m = M(), bits = abi.log_m / 2; std::vector<std::vector<uint32_t>> out;
    auto split = [&](size_t idx, size_t& l, size_t& r) { size_t mask = ((size_t)1 << bits) - 1; r = idx & mask; l = (idx >> bits) & mask; };   // utils/mod.rs:82-89
    if (abi.kind == LASSO_AND || abi.kind == LASSO_OR || abi.kind == LASSO_XOR) {
      std::vector<uint32_t> t(m);
      for (size_t i = 0; i < m; i++) { size_t l, r; split(i, l, r); t[i] = (uint32_t)(abi.kind == LASSO_AND ? (l & r) : abi.kind == LASSO_OR ? (l | r) : (l ^ r)); }
      out.push_back(t);
    } else if (abi.kind == LASSO_LT) {
      std::vector<uint32_t> lt(m), eq(m);
      for (size_t i = 0; i < m; i++) { size_t l, r; split(i, l, r); lt[i] = l < r; eq[i] = l == r; }
      out.push_back(lt); out.push_back(eq);
    } else {
      std::vector<uint32_t> full(m), rem(m), zeros(m, 0);
      size_t cutoff = (size_t)1 << (abi.log_r % abi.log_m);
      for (size_t i = 0; i < m; i++) { full[i] = (uint32_t)i; rem[i] = i < cutoff ? (uint32_t)i : 0; }
      out.push_back(full); out.push_back(rem); out.push_back(zeros);
    }
    return out;
  }
};

inline std::vector<ScVec> Strategy::spark_point() const {
  ChaChaRng rng = ChaChaRng::test_rng(); std::vector<ScVec> tau(C());
  for (size_t i = 0; i < C(); i++) for (size_t b = 0; b < abi.log_m; b++) tau[i].push_back(fr_rand(rng));
  return tau;
}

// ------------------------------------------------------------------ UniPoly (poly/unipoly.rs:13-120)
// from_evals solves the same Vandermonde system as the reference's Gaussian elimination; the solution is unique and the
// arithmetic exact, so the inverse matrix is computed once per degree and applied as a mat-vec.
struct UniPoly {
  ScVec coeffs;
  static const std::vector<ScVec>& inv_vandermonde(size_t n) {
    // per-thread cache: slab mode drives P ranks as P threads in lockstep, which all reach the first from_evals(n) at the same moment
    // (a shared map was a data race); a degree's matrix costs ~n^3 field operations once per thread, the returned reference stays thread-private
    static thread_local std::map<size_t, std::vector<ScVec>> cache;
    auto it = cache.find(n); if (it != cache.end()) return it->second;
    std::vector<ScVec> a(n, ScVec(2 * n, Sc::zero()));
    for (size_t i = 0; i < n; i++) { Sc x = Sc::from_u64(i), pw = Sc::one(); for (size_t j = 0; j < n; j++) { a[i][j] = pw; pw *= x; } a[i][n + i] = Sc::one(); }
    for (size_t col = 0; col < n; col++) {   // Gauss-Jordan; leading minors of a Vandermonde matrix on distinct nodes are non-singular
      size_t piv = col; while (a[piv][col].is_zero()) piv++;
      std::swap(a[piv], a[col]);
      Sc inv = a[col][col].inverse(); for (auto& v : a[col]) v *= inv;
      for (size_t r = 0; r < n; r++) if (r != col && !a[r][col].is_zero()) { Sc f = a[r][col]; for (size_t k = 0; k < 2 * n; k++) a[r][k] -= f * a[col][k]; }
    }
    std::vector<ScVec> inv(n, ScVec(n)); for (size_t i = 0; i < n; i++) for (size_t j = 0; j < n; j++) inv[i][j] = a[i][n + j];
    return cache.emplace(n, inv).first->second;
  }
  static UniPoly from_evals(const ScVec& evals) {
    const auto& V = inv_vandermonde(evals.size()); UniPoly p; p.coeffs.assign(evals.size(), Sc::zero());
    for (size_t i = 0; i < evals.size(); i++) for (size_t j = 0; j < evals.size(); j++) p.coeffs[i] += V[i][j] * evals[j];
    return p;
  }
  Sc evaluate(const Sc& r) const {   // Horner on plain 4 x u64 values: degree products instead of 2 * degree (once per sumcheck round, between the round's sums and its challenge's use)
    if (coeffs.empty()) return Sc::zero();
    const H4 x = h4_from(r.v); H4 e = h4_from(coeffs.back().v);
    for (size_t i = coeffs.size() - 1; i-- > 0;) e = h4_add(h4_mul(e, x), h4_from(coeffs[i].v));
    Sc out; out.v = h4_to(e); return out;
  }
  ScVec compress() const { ScVec c; c.push_back(coeffs[0]); c.insert(c.end(), coeffs.begin() + 2, coeffs.end()); return c; }   // drops the linear term :82-88
  void append_to_transcript(ProofTranscript& t, const char* label) const {   // :112-120
    t.append_message(label, "UniPoly_begin"); for (auto& c : coeffs) t.append_scalar("coeff", c); t.append_message(label, "UniPoly_end");
  }
};
// EqPolynomial::evals on the host, for the sqrt(n)-sized L and R vectors (eq_poly.rs:22-52)
inline ScVec eq_evals_host(const Sc* r, size_t ell) {
  ScVec ev((size_t)1 << ell, Sc::one()); size_t size = 1;
  for (size_t j = 0; j < ell; j++) { size *= 2; for (size_t i = size; i-- > 0;) { if (!(i & 1)) continue; Sc sc = ev[i / 2]; ev[i] = sc * r[j]; ev[i - 1] = sc - ev[i]; } }
  return ev;
}

// init * EqPolynomial(r[0..ell)).evals() as plain 4 x u64 values: the running factor rides on the root instead of costing a product per entry afterwards
inline std::vector<H4> eq_evals_host_scaled(const Sc* r, size_t ell, const Sc& init) {
  std::vector<H4> ev((size_t)1 << ell); ev[0] = h4_from(init.v); size_t size = 1;
  for (size_t j = 0; j < ell; j++) { const H4 rj = h4_from(r[j].v); size *= 2; for (size_t i = size; i-- > 0;) { if (!(i & 1)) continue; const H4 sc = ev[i / 2]; ev[i] = h4_mul(sc, rj); ev[i - 1] = h4_sub(sc, ev[i]); } }
  return ev;
}

// ------------------------------------------------------------------ proof container = ark-serialize (compressed) byte stream
struct ProofWriter {
  std::vector<uint8_t> b;
  void u64le(uint64_t x) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(x >> (8 * i))); }
  void sc(const Sc& s) { uint8_t t[32]; s.to_bytes(t); b.insert(b.end(), t, t + 32); }
  void sc_vec(const ScVec& v) { u64le(v.size()); for (auto& s : v) sc(s); }
  void sc_arr(const ScVec& v) { for (auto& s : v) sc(s); }
  void pt_bytes(const uint8_t p[32]) { b.insert(b.end(), p, p + 32); }
  void pts_vec(const std::vector<uint8_t>& compressed) { u64le(compressed.size() / 32); b.insert(b.end(), compressed.begin(), compressed.end()); }
};
struct SumcheckProof { std::vector<ScVec> compressed_polys; void write(ProofWriter& w) const { w.u64le(compressed_polys.size()); for (auto& c : compressed_polys) w.sc_vec(c); } };
struct DotProductProofLog {
  std::vector<uint8_t> L_vec, R_vec; uint8_t delta[32], beta[32]; Sc z1, z2;
  void write(ProofWriter& w) const { w.pts_vec(L_vec); w.pts_vec(R_vec); w.pt_bytes(delta); w.pt_bytes(beta); w.sc(z1); w.sc(z2); }
};
struct LayerProofBatched { SumcheckProof proof; ScVec claims_prod_left, claims_prod_right; };
struct BatchedGrandProductArgument { std::vector<LayerProofBatched> proof; void write(ProofWriter& w) const { w.u64le(proof.size()); for (auto& l : proof) { l.proof.write(w); w.sc_vec(l.claims_prod_left); w.sc_vec(l.claims_prod_right); } } };

// ------------------------------------------------------------------ Hyrax commitment of a device polynomial (dense_mlpoly.rs:109-181)
struct PolyCommitment { std::vector<uint8_t> compressed; size_t rows = 0; };   // C: Vec<G>, kept in wire form
// d_Z: the whole polynomial, or in slab mode the rank's local array (row-major L x R/P: its columns of every row)
// d_u32 (optional): the polynomial's canonical values as 32-bit integers, all <= max_u32, when the caller has them anyway
// hyrax_commit_rows: any run of l_size whole rows of the matrix (capacity mode commits its compact polynomials block by block)
inline PolyCommitment hyrax_commit_rows(const Dev& d, const lasso_fr* d_Z, size_t l_size, size_t r_size, const PolyCommitmentGens& gens, const uint32_t* d_u32 = nullptr, uint32_t max_u32 = 0) {
  LASSO_REQUIRE(r_size == gens.n && l_size >= 1);
  if (!d.comm.sharded()) {   // rows come back in wire form: the normalisation (one inversion per row) runs on the device
    PolyCommitment c; c.rows = l_size; c.compressed.resize(32 * l_size);
    if (d_u32) d.chk(lasso_hyrax_commit_compressed_u32(d.ctx, d_u32, max_u32, l_size, r_size, gens.bases, c.compressed.data()), "lasso_hyrax_commit_compressed_u32");
    else d.chk(lasso_hyrax_commit_compressed(d.ctx, d_Z, l_size, r_size, gens.bases, c.compressed.data()), "lasso_hyrax_commit_compressed");
    return c;
  }
  if (lasso_rccl_ready(d.ctx) == (int32_t)d.comm.world) {
    // the exchange step of the path on the device: partial row commitments stay in HBM, RCCL all-gathers them over xGMI on the context's stream,
    // and every rank adds the P partials of each row and compresses (lasso_hip.h "slab mode")
    const size_t P = d.comm.world, rb = lasso_point_row_bytes(); LASSO_REQUIRE(gens.bases_slab && r_size >= P);
    void* d_part = d.alloc_bytes(l_size * rb); void* d_all = d.alloc_bytes(P * l_size * rb);
    PolyCommitment c; c.rows = l_size; c.compressed.resize(32 * l_size);
    int32_t rc = lasso_hyrax_commit_rows_dev(d.ctx, d_Z, l_size, r_size / P, gens.bases_slab, d_part);
    // ncclAllGather is a collective and the wait behind it (lasso_points_reduce_compress synchronises the stream) has no time-out: a rank whose partial
    // commitment failed (out of memory, ...) must not leave its peers inside it.  The ranks therefore exchange their status through the host-side
    // all-gather first (bounded waits) and enter the device collective only if EVERY rank succeeded; otherwise all of them throw together.
    {
      std::vector<int32_t> st(P, 0); const int32_t mine = rc;
      d.comm.allgather(&mine, st.data(), sizeof(int32_t));
      for (size_t g = 0; g < P; g++) if (st[g] != 0 && rc == 0) rc = LASSO_ERR_HIP;   // a peer failed: do not enter the collective
      if (rc != 0 && mine == 0) { d.free(d_part); d.free(d_all); throw Error("slab commitment exchange: another rank failed to compute its partial row commitments"); }
    }
    if (!rc) rc = lasso_rccl_allgather(d.ctx, d_part, d_all, l_size * rb);
    if (!rc) rc = lasso_points_reduce_compress(d.ctx, d_all, (uint32_t)P, l_size, c.compressed.data());
    d.free(d_part); d.free(d_all);
    d.chk(rc, "slab commitment exchange (RCCL)");
    return c;
  }
  std::vector<lasso_point> rows(l_size);
  std::vector<Pt> pts(l_size);
  {
    // partial row commitments over this rank's columns, then the exchange step of the path: all-gather the points, add per row
    const size_t P = d.comm.world; LASSO_REQUIRE(gens.bases_slab && r_size >= P);
    d.chk(lasso_hyrax_commit(d.ctx, d_Z, l_size, r_size / P, gens.bases_slab, rows.data()), "lasso_hyrax_commit");
    std::vector<lasso_point> all(l_size * P);
    d.comm.allgather(rows.data(), all.data(), l_size * sizeof(lasso_point));
    for (size_t i = 0; i < l_size; i++) { Pt acc = Pt::from_abi(all[i]); for (size_t g = 1; g < P; g++) acc = acc + Pt::from_abi(all[g * l_size + i]); pts[i] = acc; }
  }
  PolyCommitment c; c.rows = l_size; compress_batch(pts, c.compressed); return c;
}
inline PolyCommitment hyrax_commit(const Dev& d, const lasso_fr* d_Z, size_t num_vars, const PolyCommitmentGens& gens, const uint32_t* d_u32 = nullptr, uint32_t max_u32 = 0) {
  return hyrax_commit_rows(d, d_Z, (size_t)1 << (num_vars / 2), (size_t)1 << (num_vars - num_vars / 2), gens, d_u32, max_u32);
}
inline void append_poly_commitment(ProofTranscript& t, const char* label, const PolyCommitment& c) {  // dense_mlpoly.rs:281-289
  t.append_message(label, "poly_commitment_begin");
  for (size_t i = 0; i < c.rows; i++) t.append_point_bytes("poly_commitment_share", &c.compressed[32 * i]);
  t.append_message(label, "poly_commitment_end");
}

// ------------------------------------------------------------------ DensifiedRepresentation (densified.rs:8-97)
struct SparsePolynomialCommitment { PolyCommitment l_variate_polys_commitment, log_m_variate_polys_commitment; size_t s, log_m, m; };
// capacity mode: local lookup count from which the operations' trees are kept without leaves and dim / read are kept compact
inline size_t capacity_leafless_min() { static const size_t v = [] { const char* e = getenv("LASSO_LEAFLESS_MIN"); const size_t x = e ? (size_t)atoll(e) : ((size_t)1 << 16); return x < 64 ? (size_t)64 : x; }(); return v; }
inline bool capacity_compact_off() { static const bool v = [] { const char* e = getenv("LASSO_CAPACITY_COMPACT"); return e && e[0] == '0'; }(); return v; }
struct DensifiedRepresentation {
  const Dev* dev; size_t C, s, log_m, m;               // global sizes
  size_t s_loc, m_loc;                                  // this rank's share (== s, m when not sharded)
  std::vector<DBufU32> dim_u32;                       // dim_usize, on device (local)
  DBuf combined_l_variate_polys, combined_log_m_variate_polys;   // [dim_1..dim_C, read_1..read_C | 0...], [final_1..final_C | 0...] (local slabs)
  size_t nv_l, nv_m;                                    // global numbers of variables of the two merged polynomials
  // Capacity mode, compact form: dim_i and read_i are integers below 2^32 (addresses, access counts), so the merged polynomial [dim.. read.. | 0..] is held as
  // dim_u32 + read_u32 (4 bytes per entry instead of 32) and NO field-element copy exists; whoever needs field elements (commitment of a row block, evaluation,
  // the opening's L*Z) lifts one polynomial at a time into a scratch array.  Same proof bytes.
  bool compact = false;
  std::vector<DBufU32> read_u32; std::vector<uint32_t> read_max;
  const lasso_fr* dim(size_t i) const { LASSO_REQUIRE(!compact); return combined_l_variate_polys.p + i * s_loc; }
  const lasso_fr* read(size_t i) const { LASSO_REQUIRE(!compact); return combined_l_variate_polys.p + (C + i) * s_loc; }
  // compact form: polynomial b of the merged buffer (b < C: dim_b, b < 2C: read_{b-C}) as integers, and lifted into d_tmp (s_loc elements)
  const uint32_t* block_u32(size_t b) const { return b < C ? dim_u32[b].p : read_u32[b - C].p; }
  uint32_t block_max(size_t b) const { return b < C ? (uint32_t)(m - 1) : read_max[b - C]; }
  void lift_block(size_t b, lasso_fr* d_tmp) const { dev->chk(lasso_fr_from_u32(dev->ctx, block_u32(b), s_loc, d_tmp), "lasso_fr_from_u32"); }
  const lasso_fr* final_(size_t i) const { return combined_log_m_variate_polys.p + i * m_loc; }

  // indices: n_lookups x C, row-major (the reference's Vec<[usize; C]>)
  static std::unique_ptr<DensifiedRepresentation> from_lookup_indices(const Dev& d, const uint64_t* indices, size_t n_lookups, size_t C, size_t log_m) {
    Trace tr_all("DensifiedRepresentation.from_lookup_indices", d.ctx);
    auto D = std::make_unique<DensifiedRepresentation>();
    D->dev = &d; D->C = C; D->s = next_pow2(n_lookups); D->log_m = log_m; D->m = (size_t)1 << log_m;
    const size_t s = D->s, m = D->m, P = d.comm.world;
    if (d.comm.sharded() && (s < 2 * P || m < 2 * P)) throw Error("slab sharding needs at least 2 lookups and 2 table entries per rank");
    D->s_loc = s / P; D->m_loc = m / P;
    size_t n_l = next_pow2(2 * C * s), n_m = next_pow2(C) * m;
    D->nv_l = ceil_log2(n_l); D->nv_m = ceil_log2(n_m);
    if (d.comm.sharded() && (((size_t)1 << (D->nv_m - D->nv_m / 2)) < P || ((size_t)1 << (D->nv_l - D->nv_l / 2)) < P)) throw Error("slab sharding needs every Hyrax matrix to have at least one column per rank");
    // compact form needs a polynomial to be whole rows of the Hyrax matrix (s >= its row length: true from s = 2 * next_pow2(2C) on)
    D->compact = d.capacity && !capacity_compact_off() && D->s_loc >= capacity_leafless_min() && s >= ((size_t)1 << (D->nv_l - D->nv_l / 2));
    D->combined_log_m_variate_polys = DBuf(d, n_m / P);
    if (!D->compact) D->combined_l_variate_polys = DBuf(d, n_l / P);
    // DensePolynomial::merge pads with zeros up to the next power of two (dense_mlpoly.rs:251-261)
    if (!D->compact && n_l > 2 * C * s) d.chk(lasso_zero(d.ctx, D->combined_l_variate_polys.p + 2 * C * D->s_loc, (n_l - 2 * C * s) / P * sizeof(lasso_fr)), "lasso_zero");
    if (n_m > C * m) d.chk(lasso_zero(d.ctx, D->combined_log_m_variate_polys.p + C * D->m_loc, (n_m - C * m) / P * sizeof(lasso_fr)), "lasso_zero");
    // densified.rs:32-57 on the device: the index array is uploaded once as the reference holds it (Vec<[usize; C]>), each dimension is one
    // lasso_densify_dim call (stable radix sort by address -> read/final timestamps), the polynomials are written straight into the merged buffers.
    // Slab mode: every rank sorts the whole sequence (timestamps are a property of the whole sequence) and keeps its residue class.
    DBufU64 d_idx(d, indices, n_lookups * C);
    DBuf tmp_dim, tmp_read; if (D->compact) { tmp_dim = DBuf(d, D->s_loc); tmp_read = DBuf(d, D->s_loc); }
    for (size_t i = 0; i < C; i++) {
      DBufU32 d_access(d, D->s_loc);
      lasso_fr* dim_out = D->compact ? tmp_dim.p : D->combined_l_variate_polys.p + i * D->s_loc;
      lasso_fr* read_out = D->compact ? tmp_read.p : D->combined_l_variate_polys.p + (C + i) * D->s_loc;
      d.chk(lasso_densify_dim_slab(d.ctx, d_idx.p, n_lookups, C, i, s, (uint32_t)log_m, (uint32_t)P, (uint32_t)d.comm.rank, d_access.p, dim_out, read_out, D->combined_log_m_variate_polys.p + i * D->m_loc), "lasso_densify_dim");
      D->dim_u32.push_back(std::move(d_access));
      if (D->compact) {
        DBufU32 r32(d, D->s_loc); uint32_t mx = 0;
        d.chk(lasso_fr_to_u32(d.ctx, tmp_read.p, D->s_loc, r32.p, &mx), "lasso_fr_to_u32");
        D->read_u32.push_back(std::move(r32)); D->read_max.push_back(mx);
      }
    }
    if (D->compact) { tmp_dim.release(); tmp_read.release(); }
    if (d.capacity) (void)lasso_trim(d.ctx);   // the sort's buffers (16 bytes per lookup of scratch) are not needed again
    return D;
  }
  SparsePolynomialCommitment commit(const SparsePolyCommitmentGens& gens) const {  // densified.rs:78-96
    Trace tr_all("DensifiedRepresentation.commit", dev->ctx);
    SparsePolynomialCommitment c;
    if (!compact) c.l_variate_polys_commitment = hyrax_commit(*dev, combined_l_variate_polys.p, nv_l, gens.gens_combined_l_variate);
    else {
      // one polynomial = rows_per whole rows: committed block by block (straight from the integers on one GPU — the commitment has that form anyway —, lifted into a
      // scratch array in slab mode); the zero padding's rows commit to the identity
      const size_t L = (size_t)1 << (nv_l / 2), R = (size_t)1 << (nv_l - nv_l / 2), rows_per = s / R;
      PolyCommitment& pc = c.l_variate_polys_commitment; pc.rows = L; pc.compressed.reserve(32 * L);
      DBuf tmp(*dev, s_loc);
      for (size_t b = 0; b < 2 * C; b++) {
        PolyCommitment part;
        if (!dev->comm.sharded()) part = hyrax_commit_rows(*dev, nullptr, rows_per, R, gens.gens_combined_l_variate, block_u32(b), block_max(b));
        else { lift_block(b, tmp.p); part = hyrax_commit_rows(*dev, tmp.p, rows_per, R, gens.gens_combined_l_variate); }
        pc.compressed.insert(pc.compressed.end(), part.compressed.begin(), part.compressed.end());
      }
      if (2 * C * rows_per < L) {
        dev->chk(lasso_zero(dev->ctx, tmp.p, (R / dev->comm.world) * sizeof(lasso_fr)), "lasso_zero");
        PolyCommitment zero_row = hyrax_commit_rows(*dev, tmp.p, 1, R, gens.gens_combined_l_variate);
        for (size_t r = 2 * C * rows_per; r < L; r++) pc.compressed.insert(pc.compressed.end(), zero_row.compressed.begin(), zero_row.compressed.end());
      }
      tmp.release();
    }
    c.log_m_variate_polys_commitment = hyrax_commit(*dev, combined_log_m_variate_polys.p, nv_m, gens.gens_combined_log_m_variate);
    c.s = s; c.log_m = log_m; c.m = m; return c;
  }
};

// ------------------------------------------------------------------ the prover
class Prover {
  const Dev& d; const Strategy S; DensifiedRepresentation& dense; const SparsePolyCommitmentGens& gens; ProofTranscript& t; RandomTape& tape;
  size_t alpha, s, m, nv_derefs;
  size_t P, lgP, s_loc, m_loc;       // slab mode: world size, its log2, local lengths (P = 1: s_loc = s, m_loc = m)
  std::vector<DBuf> tables;          // subtable_entries, lifted to Fr (whole tables on every rank)
  DBuf combined_E;                   // Subtables::combined_poly = merge(lookup_polys); E_i = slice i (local slab)
  const lasso_fr* E(size_t i) const { return combined_E.p + i * s_loc; }
  std::vector<DBuf> tail_bufs;       // P-element replicated arrays of the current sumcheck's tail
  std::vector<DBuf> side_keep;       // inputs of work in flight on the side context (released after lasso_sync(side))
  static size_t tail_q() { static const size_t q = lasso_sumcheck_tail_capacity(); return q; }
  // A grand-product layer's eq table, not built yet: cubic_rounds builds it INSIDE the first round's launch when the sizes allow (lasso_sumcheck_cubic_eqw2_begin_eq /
  // lasso_sumcheck_cubic_tail_begin_eq) and with lasso_eq_evals_scaled otherwise.  Set by bgpa_prove for the layer's first phase, consumed by cubic_rounds.
  struct LazyEq { bool on = false; std::vector<lasso_fr> rr; lasso_fr scale; lasso_fr* d_table = nullptr; } lazy_eq;
  // The NEXT layer's first launch enqueued while the current layer's resident tail is still answering (include/lasso_hip.h lasso_sumcheck_cubic_*_begin_eq_ahead): bgpa_prove
  // sets next_layer_hook before a layer's sumcheck, cubic_rounds calls it once its own last launch (the tail) is in the stream, and the next layer's cubic_rounds posts the eq
  // point (lasso_point_post) instead of launching — or cancels when the layer turns out to have another shape (a zero coordinate: probability 2^-252, scripted tests).
  struct LayerAhead { bool on = false, tail = false; std::vector<lasso_fr*> A, B; lasso_fr* d_table = nullptr; size_t len = 0, m_stop = 1; uint32_t ell = 0; } layer_ahead;
  std::function<void()> next_layer_hook;
  static bool tail_switched_off() { static const bool off = [] { const char* v = getenv("LASSO_CUBIC_TAIL"); return v && v[0] == '0'; }(); return off; }
  bool layer_ahead_ok() { return P == 1 && !d.no_ahead() && !eq_inline_off() && !tail_switched_off() && lasso_layer_ahead_ok(d.ctx) == 1; }
  // what cubic_rounds would launch first for a whole plain layer of k circuits and `len` elements per circuit, enqueued now (mirrors its choices: tail_from, m_stop, the table sizes)
  void enqueue_layer_ahead(const std::vector<lasso_fr*>& A, const std::vector<lasso_fr*>& B, size_t len, lasso_fr* d_table) {
    layer_ahead.on = false;
    if (!layer_ahead_ok() || len < 4) return;
    const size_t k = A.size(); const uint32_t ell = (uint32_t)ceil_log2(len / 2);
    LayerAhead la; la.A = A; la.B = B; la.d_table = d_table; la.len = len; la.ell = ell;
    int32_t rc;
    if (len / 2 <= tail_q()) {   // the whole layer in the resident kernel (tail_from = 0)
      if (ell > 9) return;
      const size_t m0 = host_m_stop(k); la.m_stop = (m0 >= 2 && m0 < len) ? m0 : 1; la.tail = true;
      if (la.m_stop > 1) d.chk(lasso_tail_handover_next(d.ctx, (uint32_t)la.m_stop), "lasso_tail_handover_next");
      rc = lasso_sumcheck_cubic_tail_begin_eq_ahead(d.ctx, la.A.data(), la.B.data(), (uint32_t)k, len, ell);
    } else {
      if (ell > 32 || len / 2 <= 64) return;
      rc = lasso_sumcheck_cubic_eqw2_begin_eq_ahead(d.ctx, la.A.data(), la.B.data(), (uint32_t)k, d_table, len, ell);
    }
    if (rc == LASSO_ERR_UNSUPPORTED) return;   // a buffer would have had to grow: the layer starts the plain way
    d.chk(rc, "lasso_sumcheck_cubic_*_begin_eq_ahead");
    la.on = true; layer_ahead = std::move(la);
  }
  static bool eq_inline_off() { static const bool v = [] { const char* e = getenv("LASSO_EQ_INLINE"); return e && e[0] == '0'; }(); return v; }
  static bool side_off() { static const bool v = [] { const char* e = getenv("LASSO_SIDE_STREAM"); return e && e[0] == '0'; }(); return v; }

 public:
  std::vector<uint8_t> proof_bytes;
  Prover(const Dev& d_, const Strategy& S_, DensifiedRepresentation& dense_, const SparsePolyCommitmentGens& gens_, ProofTranscript& t_, RandomTape& tape_)
      : d(d_), S(S_), dense(dense_), gens(gens_), t(t_), tape(tape_) {
    alpha = S.num_memories(); s = dense.s; m = dense.m; LASSO_REQUIRE(S.C() == dense.C && S.M() == dense.m);
    P = d.comm.world; lgP = d.comm.log_world(); s_loc = dense.s_loc; m_loc = dense.m_loc;
  }

  // ---- slab helpers
  // this rank's share of EqPolynomial(point).evals(): table over the high variables times the eq factor of the rank's low bits (needs 2^|point| >= P)
  void eq_evals_local(const ScVec& point, lasso_fr* d_out) {
    LASSO_REQUIRE(point.size() >= lgP);
    std::vector<lasso_fr> rr; for (size_t i = 0; i + lgP < point.size(); i++) rr.push_back(point[i].abi());
    if (P == 1) { d.chk(lasso_eq_evals(d.ctx, rr.data(), (uint32_t)rr.size(), d_out), "lasso_eq_evals"); return; }
    lasso_fr sc = d.comm.eq_low(point).abi();
    d.chk(lasso_eq_evals_scaled(d.ctx, rr.data(), (uint32_t)rr.size(), &sc, d_out), "lasso_eq_evals_scaled");
  }
  // The eq-weighted cubic rounds only ever read the first HALF of a layer's eq table (x_0 = 0; cubic_rounds below), and that half is
  // (1 - point[0]) * eq(point[1..]): build just that (half the field multiplications and HBM writes of EqPolynomial::evals, eq_poly.rs:29-42).
  void eq_half_local(const ScVec& point, lasso_fr* d_out) {
    LASSO_REQUIRE(point.size() >= lgP);
    if (point.size() == lgP) return;   // no local round reads a table
    std::vector<lasso_fr> rr; for (size_t i = 1; i + lgP < point.size(); i++) rr.push_back(point[i].abi());
    Sc scale = Sc::one() - point[0]; if (P > 1) scale *= d.comm.eq_low(point);
    lasso_fr sc = scale.abi();
    d.chk(lasso_eq_evals_scaled(d.ctx, rr.data(), (uint32_t)rr.size(), &sc, d_out), "lasso_eq_evals_scaled");
  }
  // the same table as a specification instead of a launch (consumed by cubic_rounds, see LazyEq)
  void eq_half_lazy(const ScVec& point, lasso_fr* d_out) {
    LASSO_REQUIRE(point.size() >= lgP);
    lazy_eq.on = false;
    if (point.size() == lgP) return;
    lazy_eq.rr.clear(); for (size_t i = 1; i + lgP < point.size(); i++) lazy_eq.rr.push_back(point[i].abi());
    Sc scale = Sc::one() - point[0]; if (P > 1) scale *= d.comm.eq_low(point);
    lazy_eq.scale = scale.abi(); lazy_eq.d_table = d_out; lazy_eq.on = true;
  }
  // local arrays are down to ONE element each: all-gather them into P-element replicated arrays (index = rank = the remaining low variables)
  std::vector<lasso_fr*> gather_tail(const std::vector<lasso_fr*>& polys) {
    const size_t k = polys.size();
    std::vector<lasso_fr> mine(k), all(k * P), col(P);
    d.chk(lasso_read_heads(d.ctx, (const lasso_fr* const*)polys.data(), (uint32_t)k, mine.data()), "lasso_read_heads");
    d.comm.allgather(mine.data(), all.data(), k * sizeof(lasso_fr));
    std::vector<lasso_fr*> out;
    for (size_t i = 0; i < k; i++) {
      for (size_t g = 0; g < P; g++) col[g] = all[g * k + i];
      tail_bufs.emplace_back(d, P);
      d.chk(lasso_upload(d.ctx, tail_bufs.back().p, col.data(), P * sizeof(lasso_fr)), "lasso_upload");
      out.push_back(tail_bufs.back().p);
    }
    return out;
  }

  // ---- SumcheckInstanceProof::prove_arbitrary (sumcheck.rs:150-260); polys[0..alpha) = E clones, polys[alpha] = eq.
  // One phase = `rounds` rounds on arrays of current length len; `reduce` = the arrays are slabs, per-round sums are all-gathered and added.
  // LT (the one non-linear strategy): the work arrays of the LT memories carry the factor 32^-(C-1-m) (lasso_lt_prescale, applied once by prove_arbitrary — binding
  // keeps it), which is what lets the round kernel spend one product per memory and point (Horner form, include/lasso_hip.h); the heads are scaled back in read_heads.
  // first_u32 (optional): the UNBOUND polynomials as 32-bit integers with entries 0 / 1 (E_k = T[dim_k] of the LT / EQ subtables): the phase's first round is then exact integer
  // arithmetic (lasso_sumcheck_combine_round_lt_u32) — round 0 is half of this sumcheck's work
  void arbitrary_rounds(size_t rounds, size_t len, std::vector<lasso_fr*>& polys, size_t combined_degree, bool reduce, SumcheckProof& proof, ScVec& r_out, const std::vector<const uint32_t*>* first_u32 = nullptr) {
    std::vector<const lasso_fr*> cp(polys.begin(), polys.begin() + alpha);
    for (size_t round = 0; round < rounds; round++) {
      std::vector<lasso_fr> ev(combined_degree + 1);
      if (S.spark()) d.chk(lasso_sumcheck_combine_round(d.ctx, &S.abi, cp.data(), polys[alpha], len, (uint32_t)combined_degree, ev.data()), "lasso_sumcheck_combine_round");   // g = prod E_m: nothing is pre-scaled
      else if (round == 0 && first_u32) d.chk(lasso_sumcheck_combine_round_lt_u32(d.ctx, &S.abi, first_u32->data(), polys[alpha], len, (uint32_t)combined_degree, ev.data()), "lasso_sumcheck_combine_round_lt_u32");
      else d.chk(lasso_sumcheck_combine_round_lt_scaled(d.ctx, &S.abi, cp.data(), polys[alpha], len, (uint32_t)combined_degree, ev.data()), "lasso_sumcheck_combine_round_lt_scaled");
      if (reduce) d.comm.sum(ev);
      ScVec evals; for (auto& e : ev) evals.push_back(Sc::from_abi(e));
      UniPoly up = UniPoly::from_evals(evals);
      up.append_to_transcript(t, "poly");
      Sc r_j = t.challenge_scalar("challenge_nextround"); r_out.push_back(r_j);
      lasso_fr rj = r_j.abi();
      d.chk(lasso_bind_top(d.ctx, polys.data(), (uint32_t)polys.size(), len, &rj), "lasso_bind_top");
      len /= 2;
      proof.compressed_polys.push_back(up.compress());
    }
  }
  // The same sumcheck for the LINEAR strategies in eq-weighted form (lasso_sumcheck_linear_eqw_round): the eq polynomial is never bound (prefix of
  // d_E + host scalars), a round is one launch (bind of the previous challenge + two dot products per polynomial), the weights 2^(k*inc) are applied
  // here.  One phase = `rounds` rounds over point[v0 .. v0+rounds) on arrays of length len; polys = the alpha E clones only.
  // src (optional): read-only arrays holding the polynomials; the bound arrays go to `polys` (half the length) — no clone of the inputs
  // tail_heads (optional): filled with the alpha final values when the phase ended in the resident tail kernel (then the arrays hold stale data)
  // src_u32 (optional, with src): the same polynomials as 32-bit integers (E_k = T[dim_k]): the first round and the first bind read those instead of the 32-byte form
  void linear_rounds(size_t rounds, size_t len, std::vector<lasso_fr*>& polys, const lasso_fr* d_E, const ScVec& point, size_t v0, bool reduce, Sc& s_run, SumcheckProof& proof, ScVec& r_out,
                     const std::vector<const lasso_fr*>* src = nullptr, std::vector<lasso_fr>* tail_heads = nullptr, const std::vector<const uint32_t*>* src_u32 = nullptr) {
    if (tail_heads) tail_heads->clear();
    if (!rounds) return;
    if (src && rounds < 2) {   // too short for a fused bind to move the data: a one-element copy per polynomial, then in place
      for (size_t i = 0; i < polys.size(); i++) d.chk(lasso_copy(d.ctx, polys[i], (*src)[i], len * sizeof(lasso_fr)), "lasso_copy");
      src = nullptr;
    }
    const ScVec w = S.weights();
    ScVec inv(rounds); bool degenerate = false;
    {
      Sc prod = Sc::one();
      for (size_t j = 0; j < rounds; j++) { Sc om = Sc::one() - point[v0 + j]; if (om.is_zero()) degenerate = true; prod *= om; }
      if (!degenerate) { Sc pi = prod.inverse(); for (size_t j = rounds; j-- > 0;) { inv[j] = pi; pi *= Sc::one() - point[v0 + j]; } }
    }
    DBuf tj; if (degenerate) tj = DBuf(d, len / 2);
    Sc r_prev = Sc::zero();
    // the last rounds (<= 256 indices per polynomial) in one resident kernel, as in cubic_rounds
    static const bool tail_off = [] { const char* v = getenv("LASSO_LINEAR_TAIL"); return v && v[0] == '0'; }();
    size_t tail_from = rounds;
    if (tail_heads && !reduce && !degenerate && !tail_off) {
      size_t j0 = 0, l = len;
      while (j0 < rounds && (j0 == 0 ? l / 2 : l / 4) > tail_q()) { if (j0) l /= 2; j0++; }
      if (j0 < rounds) tail_from = j0;
    }
    bool in_tail = false;
    // streaming rounds launched ahead of their challenge, as in cubic_rounds (in-place rounds only: from round 2 on)
    static const bool ahead_env_off = [] { const char* v = getenv("LASSO_ROUNDS_AHEAD"); return v && v[0] == '0'; }();
    static const bool slab_ahead_off = [] { const char* v = getenv("LASSO_SLAB_AHEAD"); return v && v[0] == '0'; }();
    const bool ahead_ok = !ahead_env_off && !degenerate && (reduce ? (!slab_ahead_off && !d.ranks_share_a_device()) : P == 1) && !d.no_ahead() && lasso_rounds_ahead_ok(d.ctx) == 1;   // with a collective between the rounds too (cubic_rounds says why)
    bool queued = false;
    auto enqueue_next = [&](size_t jn, size_t len_now) {   // round jn (>= 2) on arrays of len_now elements, behind the round in flight
      if (!ahead_ok || jn < 2 || jn >= rounds || jn >= tail_from) return;
      d.chk(lasso_sumcheck_linear_eqw_round_fused_ahead(d.ctx, polys.data(), (uint32_t)alpha, d_E, len_now), "lasso_sumcheck_linear_eqw_round_fused_ahead"); queued = true;
    };
    for (size_t j = 0; j < rounds; j++) {
      const lasso_fr* table = d_E; Sc scale = degenerate ? Sc::one() : inv[j];
      if (degenerate) {
        std::vector<lasso_fr> rr; for (size_t t2 = v0 + j + 1; t2 < v0 + rounds; t2++) rr.push_back(point[t2].abi());
        lasso_fr sc = (reduce ? d.comm.eq_low(point) : Sc::one()).abi();
        d.chk(lasso_eq_evals_scaled(d.ctx, rr.data(), (uint32_t)rr.size(), &sc, tj.p), "lasso_eq_evals_scaled");
        table = tj.p;
      }
      std::vector<lasso_fr> ev(3 * alpha);
      if (queued) {   // this round's kernel is in the stream already: its challenge, the next round behind it, then its sums
        lasso_fr rp = r_prev.abi();
        d.chk(lasso_challenge_post(d.ctx, &rp), "lasso_challenge_post"); queued = false;
        len /= 2;
        enqueue_next(j + 1, len);
        d.chk(lasso_result_wait(d.ctx, ev.data(), 3 * alpha), "lasso_result_wait");
      } else
      if (j >= tail_from) {
        lasso_fr rp = r_prev.abi();
        if (!in_tail) {   // the data is still in src if no bind has moved it yet
          const bool from_src = src && j <= 1;
          d.chk(lasso_sumcheck_linear_tail_begin(d.ctx, from_src ? src->data() : (const lasso_fr* const*)polys.data(), (uint32_t)alpha, table, len, j == 0 ? nullptr : &rp), "lasso_sumcheck_linear_tail_begin");
          in_tail = true;
        } else d.chk(lasso_sumcheck_cubic_tail_next(d.ctx, &rp), "lasso_sumcheck_cubic_tail_next");
        if (j) len /= 2;
        std::vector<lasso_fr> e2(2 * alpha);
        d.chk(lasso_result_wait(d.ctx, e2.data(), 2 * alpha), "lasso_result_wait");
        for (size_t k2 = 0; k2 < alpha; k2++) { ev[3 * k2] = e2[2 * k2]; ev[3 * k2 + 1] = e2[2 * k2 + 1]; }
      } else if (j == 0) {
        if (src && src_u32) d.chk(lasso_sumcheck_linear_eqw_round_u32(d.ctx, src_u32->data(), (uint32_t)alpha, table, len, ev.data()), "lasso_sumcheck_linear_eqw_round_u32");
        else d.chk(lasso_sumcheck_linear_eqw_round(d.ctx, src ? src->data() : (const lasso_fr* const*)polys.data(), (uint32_t)alpha, table, len, ev.data()), "lasso_sumcheck_linear_eqw_round");
      } else {
        lasso_fr rp = r_prev.abi();
        if (j == 1 && src && src_u32) d.chk(lasso_sumcheck_linear_eqw_round_fused_from_u32(d.ctx, src_u32->data(), polys.data(), (uint32_t)alpha, table, len, &rp, ev.data()), "lasso_sumcheck_linear_eqw_round_fused_from_u32");
        else if (j == 1 && src) d.chk(lasso_sumcheck_linear_eqw_round_fused_from(d.ctx, src->data(), polys.data(), (uint32_t)alpha, table, len, &rp, ev.data()), "lasso_sumcheck_linear_eqw_round_fused_from");
        else d.chk(lasso_sumcheck_linear_eqw_round_fused(d.ctx, polys.data(), (uint32_t)alpha, table, len, &rp, ev.data()), "lasso_sumcheck_linear_eqw_round_fused");
        len /= 2;
        enqueue_next(j + 1, len);   // the chain of launched-ahead rounds starts here: round j + 1 goes into the stream before this round's Fiat-Shamir step
      }
      if (reduce) d.comm.sum(ev);
      Sc G0 = Sc::zero(), G1 = Sc::zero();
      for (size_t k2 = 0; k2 < alpha; k2++) { G0 += w[k2] * Sc::from_abi(ev[3 * k2]); G1 += w[k2] * Sc::from_abi(ev[3 * k2 + 1]); }
      const Sc& rj = point[v0 + j]; const Sc base = s_run * scale, om = Sc::one() - rj;
      ScVec evals{base * om * G0, base * rj * G1, base * (rj + rj - om) * (G1 + G1 - G0)};   // x = 0, 1, 2: eq1(r_j, x) * G(x)
      UniPoly up = UniPoly::from_evals(evals);
      up.append_to_transcript(t, "poly");
      Sc r_j = t.challenge_scalar("challenge_nextround"); r_out.push_back(r_j);
      r_prev = r_j;
      s_run *= om * (Sc::one() - r_j) + rj * r_j;
      proof.compressed_polys.push_back(up.compress());
    }
    lasso_fr rp = r_prev.abi();   // the last challenge of the phase (len == 2 here)
    if (in_tail) {
      d.chk(lasso_sumcheck_cubic_tail_next(d.ctx, &rp), "lasso_sumcheck_cubic_tail_next");
      tail_heads->resize(alpha);
      d.chk(lasso_result_wait(d.ctx, tail_heads->data(), alpha), "lasso_result_wait");
      return;
    }
    d.chk(lasso_bind_top(d.ctx, polys.data(), (uint32_t)polys.size(), len, &rp), "lasso_bind_top");
  }
  // polys: local arrays of length len_loc (global length len_loc * P); polys[alpha] = the eq table of `point` (local share in slab mode)
  // heads_out (optional): the final values E_i(r_out) of the first alpha polynomials — the sumcheck's last bind leaves exactly
  // E_i.evaluate(r_z) (surge.rs:175-176) in element 0 of every array, so the prover reads them instead of evaluating E_i again
  // src (optional, linear strategies): the first alpha polynomials are read from src and never modified; polys[i < alpha] then only need len_loc / 2 elements
  SumcheckProof prove_arbitrary(size_t num_rounds, size_t len_loc, std::vector<lasso_fr*>& polys, size_t combined_degree, const ScVec& point, ScVec& r_out, ScVec* heads_out = nullptr,
                                const std::vector<const lasso_fr*>* src = nullptr, const std::vector<const uint32_t*>* src_u32 = nullptr) {
    SumcheckProof proof;
    auto read_heads = [&](const std::vector<lasso_fr*>& arrs) {
      if (!heads_out) return;
      std::vector<lasso_fr> h(alpha);
      d.chk(lasso_read_heads(d.ctx, (const lasso_fr* const*)arrs.data(), (uint32_t)alpha, h.data()), "lasso_read_heads");
      heads_out->clear(); for (auto& x : h) heads_out->push_back(Sc::from_abi(x));
      if (S.abi.kind == LASSO_LT) {   // undo lasso_lt_prescale on the LT memories' final values: LT_m(r) = 32^(C-1-m) * head
        const Sc k32 = Sc::from_u64(32); Sc pw = Sc::one();
        for (size_t m = S.C(); m-- > 0;) { (*heads_out)[2 * m] *= pw; pw *= k32; }
      }
    };
    if (S.linear()) {
      std::vector<lasso_fr*> ep(polys.begin(), polys.begin() + alpha); Sc s_run = Sc::one();
      std::vector<lasso_fr> th;
      auto finish = [&](const std::vector<lasso_fr*>& arrs) {
        if (th.empty()) { read_heads(arrs); return; }
        if (heads_out) { heads_out->clear(); for (auto& x : th) heads_out->push_back(Sc::from_abi(x)); }
      };
      if (P == 1) { linear_rounds(num_rounds, len_loc, ep, polys[alpha], point, 0, false, s_run, proof, r_out, src, heads_out ? &th : nullptr, src_u32); finish(ep); return proof; }
      LASSO_REQUIRE(num_rounds >= lgP && ((size_t)1 << (num_rounds - lgP)) == len_loc);
      const size_t local_rounds = num_rounds - lgP;
      if (src && local_rounds == 0) { for (size_t i = 0; i < alpha; i++) d.chk(lasso_copy(d.ctx, ep[i], (*src)[i], len_loc * sizeof(lasso_fr)), "lasso_copy"); }
      linear_rounds(local_rounds, len_loc, ep, polys[alpha], point, 0, true, s_run, proof, r_out, src);
      std::vector<lasso_fr*> tail = gather_tail(ep);
      tail_bufs.emplace_back(d, P);
      std::vector<lasso_fr> rr; for (size_t i = local_rounds; i < num_rounds; i++) rr.push_back(point[i].abi());
      d.chk(lasso_eq_evals(d.ctx, rr.data(), (uint32_t)rr.size(), tail_bufs.back().p), "lasso_eq_evals");
      linear_rounds(lgP, P, tail, tail_bufs.back().p, point, local_rounds, false, s_run, proof, r_out, nullptr, heads_out ? &th : nullptr);
      finish(tail);
      tail_bufs.clear();
      return proof;
    }
    // the caller's work arrays, once (slab mode: the local arrays; the replicated tails are gathered from them).  With src the call also IS the clone of the lookup polynomials
    if (S.spark()) { if (src) for (size_t i = 0; i < alpha; i++) d.chk(lasso_copy(d.ctx, polys[i], (*src)[i], len_loc * sizeof(lasso_fr)), "lasso_copy"); }   // the clone of surge.rs:151
    else d.chk(lasso_lt_prescale(d.ctx, &S.abi, src ? src->data() : nullptr, polys.data(), len_loc), "lasso_lt_prescale");
    if (P == 1) { arbitrary_rounds(num_rounds, len_loc, polys, combined_degree, false, proof, r_out, src_u32); read_heads(polys); return proof; }
    LASSO_REQUIRE(num_rounds >= lgP && ((size_t)1 << (num_rounds - lgP)) == len_loc);
    arbitrary_rounds(num_rounds - lgP, len_loc, polys, combined_degree, true, proof, r_out);
    std::vector<lasso_fr*> tail = gather_tail(polys);
    arbitrary_rounds(lgP, P, tail, combined_degree, false, proof, r_out);
    read_heads(tail);
    tail_bufs.clear();
    return proof;
  }
  // ---- capacity mode: the bottom layer of the read / write trees without stored leaves.
  // The trees of the operations (read, write) are kept from their layer of n/2 elements upwards (lasso_fingerprint_ops_gp_upper): half their bytes.  Only the bottom layer's
  // sumcheck reads leaves, and only in its two streaming rounds — round 0 (sums over A, B) and round 1 (bind by the first challenge + sums); from round 2 on it works on the
  // bound arrays, which go into the storage of the layer above (dead by then: the argument descends).  Those two rounds run chunk by chunk: the fingerprints of one index range
  // are recomputed into a small mini-layer (lasso_fingerprint_ops_strips), the ordinary round kernel runs on it with the eq table offset to the range, and the host adds the
  // chunks' sums (exact field additions: the same round polynomial).  Costs two extra fingerprint passes and 2 x kLeafChunks hand-offs; saves 7/8 of the leaf bytes.
  struct LeafLayer {
    struct Mem { const lasso_fr* table; const uint32_t* dim; const lasso_fr* read; const uint32_t* read32 = nullptr; };   // read32: compact form (read == nullptr)
    std::vector<Mem> mems; lasso_fr gamma, tau; size_t n_loc = 0;    // circuits 2m (read) and 2m + 1 (write) of memory m, n_loc leaves each
    std::vector<lasso_fr*> work_a, work_b;                          // the arrays bound by the first challenge (n_loc / 8 elements each)
  };
  static constexpr size_t kLeafChunks = 8;
  // below this many leaves per circuit the trees are small and kept whole (LASSO_LEAFLESS_MIN: tests drive the chunked rounds at toy sizes; at least 64 so that a chunk holds an index)
  static size_t leafless_min() { return capacity_leafless_min(); }
  // round j in {0, 1} of the bottom layer from recomputed leaves; len = length of A and B (n_loc / 2); ev receives the 2k sums (q(0), q_inf per circuit)
  void leaf_round(const LeafLayer& L, size_t j, size_t len, const lasso_fr* table, const lasso_fr* rp, std::vector<lasso_fr>& ev) {
    HostClock hc("capacity: chunked leaf rounds");
    const size_t k = 2 * L.mems.size(), nstrips = j == 0 ? 2 : 4, idx = len / nstrips, cs = idx / kLeafChunks;
    LASSO_REQUIRE(len == L.n_loc / 2 && cs >= 1 && cs * kLeafChunks == idx && (j == 0) == (rp == nullptr));
    const size_t per = 2 * nstrips * cs;     // mini-layer of one circuit: [A strips.., B strips..]
    DBuf tmp(d, k * per);
    std::vector<lasso_fr*> am(k), bm(k); for (size_t c = 0; c < k; c++) { am[c] = tmp.p + c * per; bm[c] = am[c] + nstrips * cs; }
    std::vector<Sc> acc(2 * k, Sc::zero()); std::vector<lasso_fr> part(2 * k);
    for (size_t ch = 0; ch < kLeafChunks; ch++) {
      const size_t i0 = ch * cs;
      for (size_t m = 0; m < L.mems.size(); m++) {
        if (L.mems[m].read32) d.chk(lasso_fingerprint_ops_strips_u32(d.ctx, L.mems[m].table, L.mems[m].dim, L.mems[m].read32, L.n_loc, &L.gamma, &L.tau, (uint32_t)nstrips, i0, cs, am[2 * m], am[2 * m + 1]), "lasso_fingerprint_ops_strips_u32");
        else d.chk(lasso_fingerprint_ops_strips(d.ctx, L.mems[m].table, L.mems[m].dim, L.mems[m].read, L.n_loc, &L.gamma, &L.tau, (uint32_t)nstrips, i0, cs, am[2 * m], am[2 * m + 1]), "lasso_fingerprint_ops_strips");
      }
      d.chk(lasso_sumcheck_cubic_eqw2_begin(d.ctx, am.data(), bm.data(), (uint32_t)k, table + i0, nstrips * cs, rp), "lasso_sumcheck_cubic_eqw2_begin");
      d.chk(lasso_result_wait(d.ctx, part.data(), 2 * k), "lasso_result_wait");
      for (size_t i = 0; i < 2 * k; i++) acc[i] += Sc::from_abi(part[i]);
      if (j == 1) for (size_t c = 0; c < k; c++) {   // the bound halves go to their places in the (2 * idx)-element arrays the later rounds work on
        d.chk(lasso_copy(d.ctx, L.work_a[c] + i0, am[c], cs * sizeof(lasso_fr)), "lasso_copy"); d.chk(lasso_copy(d.ctx, L.work_a[c] + idx + i0, am[c] + cs, cs * sizeof(lasso_fr)), "lasso_copy");
        d.chk(lasso_copy(d.ctx, L.work_b[c] + i0, bm[c], cs * sizeof(lasso_fr)), "lasso_copy"); d.chk(lasso_copy(d.ctx, L.work_b[c] + idx + i0, bm[c] + cs, cs * sizeof(lasso_fr)), "lasso_copy");
      }
    }
    ev.resize(2 * k); for (size_t i = 0; i < 2 * k; i++) ev[i] = acc[i].abi();
  }
  // the same layer's leaves in full (the rare paths of cubic_rounds — an eq coordinate of 0 or 1 among the first two — take the ordinary kernels): arrays of n_loc elements per circuit
  std::vector<DBuf> leaf_materialise(const LeafLayer& L, std::vector<lasso_fr*>& A, std::vector<lasso_fr*>& B) {
    HostClock hc("capacity: leaves materialised after all");
    std::vector<DBuf> out; A.clear(); B.clear();
    for (auto& m : L.mems) {
      DBuf lr(d, L.n_loc), lw(d, L.n_loc), lifted;
      if (m.read32) { lifted = DBuf(d, L.n_loc); d.chk(lasso_fr_from_u32(d.ctx, m.read32, L.n_loc, lifted.p), "lasso_fr_from_u32"); }
      d.chk(lasso_fingerprint_ops(d.ctx, m.table, m.dim, m.read32 ? lifted.p : m.read, L.n_loc, &L.gamma, &L.tau, lr.p, lw.p), "lasso_fingerprint_ops");
      A.push_back(lr.p); B.push_back(lr.p + L.n_loc / 2); A.push_back(lw.p); B.push_back(lw.p + L.n_loc / 2);
      out.push_back(std::move(lr)); out.push_back(std::move(lw));
    }
    return out;
  }
  // ---- rounds the HOST finishes (round 5).  A hand-off to the device costs ~7 us per resident turn (5.6 us of device turn + the host's step) and ~25 us per launched round
  // whatever the size; once a layer's arrays are down to a few elements per circuit the round's arithmetic itself is a few dozen field products.  So (one GPU):
  //  * the resident tail hands its ARRAYS over when they are down to m_stop elements each (lasso_tail_handover_next) and the last log2(m_stop) rounds run here;
  //  * the layers of the product trees with at most 2 * m_stop elements — the top of every tree — are proved here entirely, from the tops read once per argument
  //    (lasso_read_runs in memory_checking_prove): no launch, no hand-off.
  // What runs is sumcheck.rs:49-124 literally on (A_c, B_c, C) with C = the bound eq polynomial s * EqPolynomial(rand[first..]).evals(): evaluations at 0, 2, 3 by
  // `prev + hi - lo`, e(1) from the claim (:99-104), UniPoly::from_evals, bind (:116-120) — the same field elements as the device's eq-weighted form, any eq coordinate.
  // The device remains the only place where an O(n) loop runs; this is the O(1) end of the O(log n) host share.
  // budget = elements per array x circuits the host takes over (LASSO_HOST_TAIL, default 32; 0 switches the host rounds off: A/B measurements, byte-identical)
  // default: 32 with the scalar loop; 128 where the rounds run eight elements at a time (field52.hpp, AVX-512 IFMA: a layer of 2 x 2 x 32 elements costs the host less than the two
  // device turns it replaces — profiles/r05_ab_host_ifma.txt)
  static bool host_ifma() {
#ifdef LASSO_HOST_IFMA
    return field52_ok();
#else
    return false;
#endif
  }
  static size_t host_tail_budget() { static const size_t v = [] { const char* e = getenv("LASSO_HOST_TAIL"); const long x = e ? atol(e) : (host_ifma() ? 128 : 32); return (size_t)(x < 0 ? 0 : x > 1024 ? 1024 : x); }(); return v; }
  size_t host_m_stop(size_t k) const {   // elements per array at which the host takes a layer over: a power of two, 1 = never
    if (P != 1 || !host_tail_budget() || !k) return 1;
    size_t m0 = 1; while (2 * m0 * k <= host_tail_budget() && 2 * m0 <= 64) m0 *= 2;
    return m0;
  }
  // a, b: k arrays of m = 2^rounds_left elements; rand[first ..first + rounds_left) the eq coordinates still unbound; s = the running factor prod eq1(rand_t, rho_t) so far
  void host_cubic_rounds(std::vector<ScVec>& a, std::vector<ScVec>& b, size_t rounds_left, const ScVec& rand, size_t first, const ScVec& coeffs, const Sc& s_run, Sc& e, SumcheckProof& proof,
                         ScVec& r_out, std::vector<lasso_fr>& heads) {
    HostClock hc("host rounds (handed-over tails, tree tops)");
    const size_t k = a.size(); size_t m = (size_t)1 << rounds_left;
    LASSO_REQUIRE(k == b.size() && first + rounds_left <= rand.size());
    for (size_t c = 0; c < k; c++) LASSO_REQUIRE(a[c].size() == m && b[c].size() == m);
    auto to_sc = [](const H4& x) { Sc r; r.v = h4_to(x); return r; };
    // the eq weights of the remaining coordinates with the running factor folded in
    std::vector<H4> C = eq_evals_host_scaled(rand.data() + first, rounds_left, s_run);
    // Two forms of the same loops: eight elements at a time in 52-bit limbs on AVX-512 IFMA (field52.hpp HostRounds52) where the CPU has it, plain 4 x u64 values (field_host.hpp H4)
    // otherwise.  Per circuit: A, B and A' = coeffs_c * A — the batching coefficient rides on A' (sumcheck.rs:95-97 applies it to the circuit's sums: the same by linearity), so a
    // term is ONE product per evaluation point; A itself is bound alongside because its final value is a claim (:126-133).  Both forms work on canonical values: same field
    // elements, same bytes.
#ifdef LASSO_HOST_IFMA
    std::unique_ptr<HostRounds52> v52;
    bool w_nonzero = true; for (size_t c = 0; c < k; c++) if (coeffs[c].is_zero()) w_nonzero = false;   // (the vector form recovers A_c's final value as A'_c / coeffs_c)
    if (field52_ok() && HostRounds52::fits(k, m) && w_nonzero) {
      std::vector<std::vector<H4>> ha(k, std::vector<H4>(m)), hb(k, std::vector<H4>(m)); std::vector<H4> w(k);
      for (size_t c = 0; c < k; c++) { w[c] = h4_from(coeffs[c].v); for (size_t i = 0; i < m; i++) { ha[c][i] = h4_from(a[c][i].v); hb[c][i] = h4_from(b[c][i].v); } }
      v52.reset(new HostRounds52(ha, hb, w, C));
    }
    const bool vec = v52 != nullptr;
#else
    const bool vec = false;
#endif
    std::vector<H4> A, B, Aw;
    if (!vec) {
      A.resize(k * m); B.resize(k * m); Aw.resize(k * m);
      for (size_t c = 0; c < k; c++) { const H4 w = h4_from(coeffs[c].v); for (size_t i = 0; i < m; i++) { A[c * m + i] = h4_from(a[c][i].v); B[c * m + i] = h4_from(b[c][i].v); Aw[c * m + i] = h4_mul(A[c * m + i], w); } }
    }
    size_t stride = m;   // arrays keep their stride; the live prefix halves
    for (size_t j = 0; j < rounds_left; j++) {
      const size_t h = m / 2;
      H4 e0 = h4_zero(), e2 = h4_zero(), e3 = h4_zero();
#ifdef LASSO_HOST_IFMA
      if (vec) v52->sums(h, e0, e2, e3); else
#endif
      for (size_t i = 0; i < h; i++) {
        H4 t0 = h4_zero(), t2 = h4_zero(), t3 = h4_zero();   // sum_c coeffs_c A_c(x) B_c(x) at x = 0, 2, 3 (`prev + hi - lo`, :68-89)
        for (size_t c = 0; c < k; c++) {
          const H4 *pa = &Aw[c * stride], *pb = &B[c * stride];
          const H4 da = h4_sub(pa[i + h], pa[i]), db = h4_sub(pb[i + h], pb[i]), a2 = h4_add(pa[i + h], da), b2 = h4_add(pb[i + h], db), a3 = h4_add(a2, da), b3 = h4_add(b2, db);
          t0 = h4_add(t0, h4_mul(pa[i], pb[i])); t2 = h4_add(t2, h4_mul(a2, b2)); t3 = h4_add(t3, h4_mul(a3, b3));
        }
        const H4 dc = h4_sub(C[i + h], C[i]), c2 = h4_add(C[i + h], dc), c3 = h4_add(c2, dc);
        e0 = h4_add(e0, h4_mul(t0, C[i])); e2 = h4_add(e2, h4_mul(t2, c2)); e3 = h4_add(e3, h4_mul(t3, c3));
      }
      // UniPoly::from_evals on (e(0), e(1) = claim - e(0) (:99-104), e(2), e(3)): the unique cubic through four points, in closed form (third and second finite differences) —
      // the same four coefficients the Vandermonde solve of unipoly.rs:30-66 returns, for 2 products instead of 16
      static const H4 inv2 = h4_from(Sc::from_u64(2).inverse().v), inv6 = h4_from(Sc::from_u64(6).inverse().v);
      const H4 e1 = h4_sub(h4_from(e.v), e0);
      const H4 t32 = h4_sub(e3, h4_add(h4_add(e2, e2), e2)), t11 = h4_add(h4_add(e1, e1), e1);                       // e3 - 3 e2, 3 e1
      const H4 k3 = h4_mul(h4_sub(h4_add(t32, t11), e0), inv6);                                                          // (e3 - 3 e2 + 3 e1 - e0) / 6
      const H4 k2 = h4_sub(h4_mul(h4_add(h4_sub(e2, h4_add(e1, e1)), e0), inv2), h4_add(h4_add(k3, k3), k3));          // (e2 - 2 e1 + e0) / 2 - 3 c3
      const H4 k1 = h4_sub(h4_sub(h4_sub(e1, e0), k2), k3);
      UniPoly poly; poly.coeffs = {to_sc(e0), to_sc(k1), to_sc(k2), to_sc(k3)};
      poly.append_to_transcript(t, "poly");
      const Sc r_j = t.challenge_scalar("challenge_nextround"); r_out.push_back(r_j);
      e = poly.evaluate(r_j);
      proof.compressed_polys.push_back(poly.compress());
      const H4 rj = h4_from(r_j.v);
#ifdef LASSO_HOST_IFMA
      if (vec) v52->bind(h, rj); else
#endif
      {
        for (size_t c = 0; c < k; c++) for (H4* arr : {&A[c * stride], &B[c * stride], &Aw[c * stride]}) for (size_t i = 0; i < h; i++) arr[i] = h4_add(arr[i], h4_mul(rj, h4_sub(arr[i + h], arr[i])));   // :116-120
        for (size_t i = 0; i < h; i++) C[i] = h4_add(C[i], h4_mul(rj, h4_sub(C[i + h], C[i])));
      }
      m = h;
    }
    heads.resize(2 * k);
#ifdef LASSO_HOST_IFMA
    if (vec) { std::vector<H4> hh; v52->heads(hh); for (size_t c = 0; c < 2 * k; c++) heads[c] = to_sc(hh[c]).abi(); return; }
#endif
    for (size_t c = 0; c < k; c++) { heads[c] = to_sc(A[c * stride]).abi(); heads[k + c] = to_sc(B[c * stride]).abi(); }
  }
  // ---- SumcheckInstanceProof::prove_cubic_batched (sumcheck.rs:27-135), comb = A*B*C with C = EqPolynomial(rand).evals() (grand_product.rs:122-128).
  // Round j's bind (sumcheck.rs:116-120) is executed by the same kernel that evaluates round j+1, so a round is ONE launch.  The eq polynomial is
  // never bound or stored: after j binds it is  s_j * eq1(rand_j, x_top) * T_j  with T_j a scalar multiple of the PREFIX of the layer's table
  // (lasso_sumcheck_cubic_eqw_round), so the device returns eq-weighted sums and the three scalars below turn them into sumcheck.rs:56-93's
  // evaluations — the same field elements, so the transcript is unchanged.
  // One phase = `rounds` rounds on arrays of length len (>= 2 when rounds > 0) over the variables rand[v0 .. v0+rounds), ending with A and B bound by
  // the last challenge.  d_E: table whose first len/2^(j+1) entries are prod_{t<=j}(1 - rand[v0+t]) * T_j (the phase's eq table); s = running
  // prod eq1(rand_t, rho_t) over all rounds so far (all phases).
  // leaf (capacity mode, bottom layer of the operations' trees): A and B do not exist — rounds 0 and 1 recompute them chunk by chunk (leaf_round) and the later rounds
  // run on leaf->work_a / work_b
  void cubic_rounds(size_t rounds, size_t len, std::vector<lasso_fr*>& A, std::vector<lasso_fr*>& B, const lasso_fr* d_E, const ScVec& rand, size_t v0, const ScVec& coeffs, bool reduce, Sc& s_run,
                    Sc& e, SumcheckProof& proof, ScVec& r_out, std::vector<lasso_fr>* heads_out = nullptr, const LeafLayer* leaf = nullptr) {
    const size_t k = leaf ? 2 * leaf->mems.size() : A.size();
    if (heads_out) heads_out->clear();
    std::unique_ptr<HostClock> hpro(new HostClock("layer transition: sumcheck prologue"));
    LayerAhead pre = std::move(layer_ahead); layer_ahead.on = false;                 // this layer's first launch, enqueued during the previous layer (or not)
    std::function<void()> hook = std::move(next_layer_hook); next_layer_hook = nullptr;   // the next layer's, to be enqueued once this layer's last launch is in the stream
    if (!rounds) { if (pre.on) d.chk(lasso_point_cancel(d.ctx), "lasso_point_cancel"); return; }
    // 1 / prod_{t<=j}(1 - rand[v0+t]) for every round of the phase with one inversion; a zero factor (rand_t = 1) takes the explicit-table path
    ScVec inv(rounds); bool degenerate = false;
    {
      Sc prod = Sc::one();
      for (size_t j = 0; j < rounds; j++) { Sc om = Sc::one() - rand[v0 + j]; if (om.is_zero()) degenerate = true; prod *= om; }
      if (!degenerate) { Sc pi = prod.inverse(); for (size_t j = rounds; j-- > 0;) { inv[j] = pi; pi *= Sc::one() - rand[v0 + j]; } }
    }
    DBuf tj; if (degenerate) tj = DBuf(d, len / 2);
    Sc r_prev = Sc::zero();
    // the layer's eq table may still be a specification (eq_half_lazy): it is built inside round 0's launch where that exists, by its own kernels otherwise
    LazyEq lz; if (lazy_eq.on && lazy_eq.d_table == d_E && v0 == 0) { lz = lazy_eq; } lazy_eq.on = false;
    auto ensure_table = [&] { if (lz.on) { d.chk(lasso_eq_evals_scaled(d.ctx, lz.rr.data(), (uint32_t)lz.rr.size(), &lz.scale, lz.d_table), "lasso_eq_evals_scaled"); lz.on = false; } };
    if (degenerate) lz.on = false;   // the explicit per-round tables below replace it
    // The last rounds of the phase (<= 256 indices per circuit) are served by ONE resident kernel (lasso_sumcheck_cubic_tail_*): no launch per
    // round, the bound arrays stay on chip, and the final bind + heads come back from it.  Not when a collective sits between the rounds
    // (slab-local phase), nor when one of the remaining eq coordinates is 0 or 1 (the per-round paths handle those).
    static const bool tail_off = [] { const char* v = getenv("LASSO_CUBIC_TAIL"); return v && v[0] == '0'; }();
    size_t tail_from = rounds;   // first round served by the resident kernel
    if (heads_out && !reduce && !degenerate && !tail_off) {
      size_t j0 = 0; size_t l = len;   // l = array length before round j's bind
      while (j0 < rounds && (j0 == 0 ? l / 2 : l / 4) > tail_q()) { if (j0) l /= 2; j0++; }   // the resident kernels' capacity
      bool plain = j0 < rounds;
      for (size_t j = j0; j < rounds && plain; j++) if (rand[v0 + j].is_zero()) plain = false;
      if (plain) tail_from = j0;
    }
    bool in_tail = false;
    // the host takes the layer over when its arrays are down to m_stop elements each: the resident tail runs the rounds [tail_from, j_host) and hands the arrays over
    size_t m_stop = 1, j_host = rounds;
    if (tail_from < rounds && len == ((size_t)1 << rounds)) {
      const size_t m0 = host_m_stop(k), lt = len >> tail_from;   // lt = array length at the tail's first round
      if (m0 >= 2 && m0 < lt) { m_stop = m0; j_host = rounds - ceil_log2(m0); }
    }
    // Rounds LAUNCHED AHEAD of their challenge (include/lasso_hip.h lasso_sumcheck_cubic_eqw2_begin_ahead): while round j runs, round j + 1 — a streaming round, or the resident
    // tail — is already in the stream and waits on the device for the challenge this loop posts.  One GPU, plain rounds only (no collective between rounds, no per-round table).
    // the launch enqueued during the previous layer is this layer's first launch only if the layer has the shape that was assumed then
    bool use_pre = false;
    if (pre.on) {
      static const bool three = [] { const char* v = getenv("LASSO_CUBIC_THREE_SUMS"); return v && v[0] == '1'; }();
      use_pre = lz.on && !degenerate && !leaf && v0 == 0 && heads_out && !reduce && pre.len == len && pre.A == A && pre.B == B && pre.d_table == lz.d_table && pre.ell == lz.rr.size() &&
                (pre.tail ? (tail_from == 0 && m_stop == pre.m_stop) : (tail_from > 0 && !three && !rand[0].is_zero() && !s_run.is_zero()));
      if (!use_pre) d.chk(lasso_point_cancel(d.ctx), "lasso_point_cancel");
    }
    static const bool ahead_env_off = [] { const char* v = getenv("LASSO_ROUNDS_AHEAD"); return v && v[0] == '0'; }();
    // Slab mode (reduce, round 6): the ranks' partial sums meet on the host (d.comm.sum below) BEFORE the challenge exists, so posting it is the same step one exchange later —
    // every rank keeps its own next round in its own stream behind its own gate.  (The resident tail is not used with a collective between the rounds: tail_from == rounds.)
    static const bool slab_ahead_off = [] { const char* v = getenv("LASSO_SLAB_AHEAD"); return v && v[0] == '0'; }();   // A/B switch: round 5's schedule for P > 1
    const bool ahead_ok = !ahead_env_off && !degenerate && (reduce ? (!slab_ahead_off && !d.ranks_share_a_device()) : (P == 1 && heads_out != nullptr)) && !d.no_ahead() && lasso_rounds_ahead_ok(d.ctx) == 1;
    bool queued = false, queued_tail = false;   // this round's kernel is already enqueued (a streaming round / the resident tail) and waits for r_prev
    std::vector<DBuf> leaf_full;   // the leaves after all, for the rare shapes the chunked rounds do not cover
    if (leaf) {
      static const bool three = [] { const char* v = getenv("LASSO_CUBIC_THREE_SUMS"); return v && v[0] == '1'; }();
      const bool plain = v0 == 0 && !degenerate && !three && rounds >= 3 && tail_from >= 2 && !rand[0].is_zero() && !rand[1].is_zero() && !s_run.is_zero();
      if (!plain) { leaf_full = leaf_materialise(*leaf, A, B); leaf = nullptr; }
    }
    hpro.reset();
    for (size_t j = 0; j < j_host; j++) {
      const lasso_fr* table = d_E; Sc scale = degenerate ? Sc::one() : inv[j];
      if (degenerate) {   // T_j = eq(rand[v0+j+1 .. v0+rounds)) built explicitly (size len / 2^(j+1) at this point), times the slab factor hidden in d_E[0] / eq-prefix
        std::vector<lasso_fr> rr; for (size_t t2 = v0 + j + 1; t2 < v0 + rounds; t2++) rr.push_back(rand[t2].abi());
        lasso_fr sc = (reduce ? d.comm.eq_low(rand) : Sc::one()).abi();
        d.chk(lasso_eq_evals_scaled(d.ctx, rr.data(), (uint32_t)rr.size(), &sc, tj.p), "lasso_eq_evals_scaled");
        table = tj.p;
      }
      // e(x) = f(x) * q(x) with f(x) = s * scale * eq1(rand_j, x), eq1(r, x) = (1 - r)(1 - x) + r x, and q(x) = sum_c coeffs_c q_c(x) quadratic
      const Sc& rj = rand[v0 + j]; const Sc base = s_run * scale, om = Sc::one() - rj;
      const Sc f0 = base * om, f1 = base * rj;   // f is linear: f(x) = f0 + (f1 - f0) x
      UniPoly poly;
      static const bool three_sums = [] { const char* v = getenv("LASSO_CUBIC_THREE_SUMS"); return v && v[0] == '1'; }();   // A/B switch for measurements
      if (j >= tail_from || queued || (!f1.is_zero() && !three_sums)) {
        // two sums per circuit, q_c(0) and the leading coefficient; q(1) follows from the claim e = e(0) + e(1) (sumcheck.rs:99-104 derives e(1)
        // the same way) and q(2), q(3) by extrapolation.  The inversion of f(1) overlaps the kernel.
        lasso_fr rp = r_prev.abi();
        const uint32_t ell = (uint32_t)lz.rr.size();   // table of 2^ell = len / 2 entries
        std::vector<lasso_fr> ev(2 * k); bool have_ev = false;
        if (queued) {          // enqueued while the previous round ran: it only needs its challenge
          d.chk(lasso_challenge_post(d.ctx, &rp), "lasso_challenge_post"); queued = false;
        } else if (queued_tail) {
          d.chk(lasso_sumcheck_cubic_tail_next(d.ctx, &rp), "lasso_sumcheck_cubic_tail_next"); queued_tail = false; in_tail = true;
        } else
        if (j == 0 && use_pre) {   // enqueued during the previous layer, waiting on the device for this point
          { HostClock hp("layer transition: point post"); d.chk(lasso_point_post(d.ctx, lz.rr.data(), ell, &lz.scale), "lasso_point_post"); } lz.on = false;
          if (pre.tail) { in_tail = true; if (hook) { hook(); hook = nullptr; } }
        } else
        if (leaf && j < 2) {   // capacity mode: this round's A and B are recomputed chunk by chunk; after round 1 the bound arrays are the working arrays
          ensure_table();
          leaf_round(*leaf, j, len, table, j ? &rp : nullptr, ev); have_ev = true;
          if (j == 1) { A = leaf->work_a; B = leaf->work_b; }
        } else
        if (j == 0 && lz.on && j < tail_from && ell <= 32 && len / 2 > 64) {   // round 0 of a streaming layer: the table is built in this launch and left in d_E for the later rounds
          d.chk(lasso_sumcheck_cubic_eqw2_begin_eq(d.ctx, A.data(), B.data(), (uint32_t)k, lz.d_table, len, lz.rr.data(), ell, &lz.scale), "lasso_sumcheck_cubic_eqw2_begin_eq"); lz.on = false;
        } else if (j == 0 && lz.on && j >= tail_from && ell <= 9) {            // the whole layer runs in the resident kernel: no table at all
          if (m_stop > 1) d.chk(lasso_tail_handover_next(d.ctx, (uint32_t)m_stop), "lasso_tail_handover_next");
          d.chk(lasso_sumcheck_cubic_tail_begin_eq(d.ctx, A.data(), B.data(), (uint32_t)k, len, lz.rr.data(), ell, &lz.scale), "lasso_sumcheck_cubic_tail_begin_eq"); in_tail = true; lz.on = false;
          if (hook) { hook(); hook = nullptr; }   // this layer's last launch is in the stream: the next layer's first goes in behind it
        } else {
          ensure_table();
          if (j < tail_from) d.chk(lasso_sumcheck_cubic_eqw2_begin(d.ctx, A.data(), B.data(), (uint32_t)k, table, len, j == 0 ? nullptr : &rp), "lasso_sumcheck_cubic_eqw2_begin");
          else if (!in_tail) {
            if (m_stop > 1) d.chk(lasso_tail_handover_next(d.ctx, (uint32_t)m_stop), "lasso_tail_handover_next");
            d.chk(lasso_sumcheck_cubic_tail_begin(d.ctx, A.data(), B.data(), (uint32_t)k, table, len, j == 0 ? nullptr : &rp), "lasso_sumcheck_cubic_tail_begin"); in_tail = true;
            if (hook) { hook(); hook = nullptr; }
          }
          else d.chk(lasso_sumcheck_cubic_tail_next(d.ctx, &rp), "lasso_sumcheck_cubic_tail_next");
        }
        if (j) len /= 2;
        // round j + 1 into the stream behind round j, before round j's sums are waited for: a streaming round (bind of the challenge to come + its sums), or the resident tail
        if (ahead_ok && !in_tail && !lz.on && j + 1 < j_host && !(leaf && j + 1 < 3) && !rand[v0 + j + 1].is_zero()) {
          if (j + 1 < tail_from) {
            const int32_t rc = lasso_sumcheck_cubic_eqw2_begin_ahead(d.ctx, A.data(), B.data(), (uint32_t)k, table, len);
            if (rc == 0) queued = true; else if (rc != LASSO_ERR_UNSUPPORTED) d.chk(rc, "lasso_sumcheck_cubic_eqw2_begin_ahead");
          } else if (j + 1 == tail_from) {
            if (m_stop > 1) d.chk(lasso_tail_handover_next(d.ctx, (uint32_t)m_stop), "lasso_tail_handover_next");
            d.chk(lasso_sumcheck_cubic_tail_begin_ahead(d.ctx, A.data(), B.data(), (uint32_t)k, table, len), "lasso_sumcheck_cubic_tail_begin_ahead"); queued_tail = true;
            if (hook) { hook(); hook = nullptr; }
          }
        }
        // f(1) = 0 inside the tail can only come from a vanished running factor s (probability 2^-252): then f = 0 identically and q is irrelevant
        const Sc f1_inv = f1.is_zero() ? Sc::zero() : f1.inverse();
        if (!have_ev) d.chk(lasso_result_wait(d.ctx, ev.data(), 2 * k), "lasso_result_wait");
        if (reduce) d.comm.sum(ev);
        HostClock hc("cubic round host work");
        Sc q0 = Sc::zero(), qi = Sc::zero();
        for (size_t i = 0; i < k; i++) { q0 += Sc::from_abi(ev[2 * i]) * coeffs[i]; qi += Sc::from_abi(ev[2 * i + 1]) * coeffs[i]; }
        // q(x) = q0 + ql x + qi x^2 with ql = q(1) - q0 - qi, so the round polynomial e = f q has the coefficients below — the same four field elements
        // UniPoly::from_evals (unipoly.rs:30-66) finds by solving the Vandermonde system on e(0..3) (the interpolant of a cubic is unique, the arithmetic
        // exact), for 6 products instead of the 4 evaluations' + the solve's 22
        const Sc c0 = f0 * q0, q1 = (e - c0) * f1_inv, ql = q1 - q0 - qi, df = f1 - f0;
        poly.coeffs = {c0, f0 * ql + df * q0, f0 * qi + df * ql, df * qi};
      } else {   // rand_j = 0 (or a zero running factor): the claim says nothing about q(1); three sums from the device
        std::vector<lasso_fr> ev(3 * k);
        ensure_table();
        if (j == 0) {
          d.chk(lasso_sumcheck_cubic_eqw_round(d.ctx, (const lasso_fr* const*)A.data(), (const lasso_fr* const*)B.data(), (uint32_t)k, table, len, ev.data()), "lasso_sumcheck_cubic_eqw_round");
        } else {
          lasso_fr rp = r_prev.abi();
          d.chk(lasso_sumcheck_cubic_eqw_round_fused(d.ctx, A.data(), B.data(), (uint32_t)k, table, len, &rp, ev.data()), "lasso_sumcheck_cubic_eqw_round_fused");
          len /= 2;
        }
        if (reduce) d.comm.sum(ev);
        const Sc f2 = base * (rj + rj - om), f3 = base * (rj + rj + rj - om - om);
        Sc c0 = Sc::zero(), c2 = Sc::zero(), c3 = Sc::zero();
        for (size_t i = 0; i < k; i++) { c0 += Sc::from_abi(ev[3 * i]) * coeffs[i]; c2 += Sc::from_abi(ev[3 * i + 1]) * coeffs[i]; c3 += Sc::from_abi(ev[3 * i + 2]) * coeffs[i]; }
        c0 *= f0; c2 *= f2; c3 *= f3;
        poly = UniPoly::from_evals({c0, e - c0, c2, c3});
      }
      HostClock hc2("cubic round host work");
      poly.append_to_transcript(t, "poly");
      Sc r_j = t.challenge_scalar("challenge_nextround"); r_out.push_back(r_j);
      r_prev = r_j;
      e = poly.evaluate(r_j);
      s_run *= om * (Sc::one() - r_j) + rj * r_j;   // eq1(rand_j, rho_j)
      proof.compressed_polys.push_back(poly.compress());
    }
    lasso_fr rp = r_prev.abi();
    if (in_tail) {   // the resident kernel binds the last challenge itself and hands back the heads A_c[0], B_c[0] — or, stopped early, the arrays of m_stop elements
      d.chk(lasso_sumcheck_cubic_tail_next(d.ctx, &rp), "lasso_sumcheck_cubic_tail_next");
      heads_out->resize(2 * k * m_stop);
      d.chk(lasso_result_wait(d.ctx, heads_out->data(), 2 * k * m_stop), "lasso_result_wait");
      if (m_stop > 1) {   // the host finishes the layer: rounds [j_host, rounds) on the handed-over arrays
        std::vector<ScVec> ha(k, ScVec(m_stop)), hb(k, ScVec(m_stop));
        for (size_t c = 0; c < k; c++) for (size_t i = 0; i < m_stop; i++) { ha[c][i] = Sc::from_abi((*heads_out)[c * m_stop + i]); hb[c][i] = Sc::from_abi((*heads_out)[(k + c) * m_stop + i]); }
        host_cubic_rounds(ha, hb, rounds - j_host, rand, v0 + j_host, coeffs, s_run, e, proof, r_out, *heads_out);
      }
      return;
    }
    LASSO_REQUIRE(j_host == rounds && !queued && !queued_tail);
    // the last challenge of the phase still has to be bound (len == 2 here)
    std::vector<lasso_fr*> ab(A); ab.insert(ab.end(), B.begin(), B.end());
    d.chk(lasso_bind_top(d.ctx, ab.data(), (uint32_t)ab.size(), len, &rp), "lasso_bind_top");
  }
  // A, B: local arrays of length 2^num_rounds / P (slab mode) or the whole arrays; d_E = the layer's eq table over `rand` (local share in slab mode)
  SumcheckProof prove_cubic_batched(const Sc& claim, size_t num_rounds, bool slab, std::vector<lasso_fr*>& A, std::vector<lasso_fr*>& B, const lasso_fr* d_E, const ScVec& rand, const ScVec& coeffs,
                                    ScVec& r_out, ScVec& claims_a, ScVec& claims_b, const LeafLayer* leaf = nullptr) {
    SumcheckProof proof; Sc e = claim, s_run = Sc::one(); const size_t k = leaf ? 2 * leaf->mems.size() : A.size();
    LASSO_REQUIRE(rand.size() == num_rounds);
    std::vector<lasso_fr*> fa(A), fb(B);   // where the final values end up
    std::vector<lasso_fr> heads;            // ... unless the resident tail kernel hands them back directly
    if (!slab) {
      cubic_rounds(num_rounds, (size_t)1 << num_rounds, fa, fb, d_E, rand, 0, coeffs, false, s_run, e, proof, r_out, &heads, leaf);
    } else {
      LASSO_REQUIRE(num_rounds >= lgP);
      const size_t local_rounds = num_rounds - lgP;
      cubic_rounds(local_rounds, (size_t)1 << local_rounds, fa, fb, d_E, rand, 0, coeffs, true, s_run, e, proof, r_out, nullptr, leaf);
      std::vector<lasso_fr*> local_heads(fa); local_heads.insert(local_heads.end(), fb.begin(), fb.end());
      static const bool slab_host_tail_off = [] { const char* v = getenv("LASSO_SLAB_HOST_TAIL"); return v && v[0] == '0'; }();   // A/B switch: round 5's device phase
      if (!slab_host_tail_off) {
        // Round 6: the remaining log2 P variables live in P-element arrays that every rank holds whole after one all-gather of the local heads — 2 k P field elements.  Round 5
        // uploaded them (2k hipMemcpy), built a P-entry eq table and ran a resident kernel for log2 P rounds: ~100 us of launches and hand-offs per layer for a few dozen field
        // products.  They are the host's: host_cubic_rounds is sumcheck.rs:49-124 literally on (A_c, B_c, eq), every rank computes the same bytes, nothing touches the device.
        std::vector<lasso_fr> mine(2 * k), all(2 * k * P);
        d.chk(lasso_read_heads(d.ctx, (const lasso_fr* const*)local_heads.data(), (uint32_t)(2 * k), mine.data()), "lasso_read_heads");
        d.comm.allgather(mine.data(), all.data(), 2 * k * sizeof(lasso_fr));
        std::vector<ScVec> ha(k, ScVec(P)), hb(k, ScVec(P));   // element g = rank g's head: the rank index IS the remaining low variables (gather_tail's layout)
        for (size_t g = 0; g < P; g++) for (size_t c = 0; c < k; c++) { ha[c][g] = Sc::from_abi(all[g * 2 * k + c]); hb[c][g] = Sc::from_abi(all[g * 2 * k + k + c]); }
        host_cubic_rounds(ha, hb, lgP, rand, local_rounds, coeffs, s_run, e, proof, r_out, heads);
        claims_a.clear(); claims_b.clear();
        for (size_t i = 0; i < k; i++) { claims_a.push_back(Sc::from_abi(heads[i])); claims_b.push_back(Sc::from_abi(heads[k + i])); }
        return proof;
      }
      std::vector<lasso_fr*> tail = gather_tail(local_heads);
      fa.assign(tail.begin(), tail.begin() + k); fb.assign(tail.begin() + k, tail.begin() + 2 * k);
      // the remaining log2 P variables: replicated P-element arrays and the (whole) eq table over rand[local_rounds..]
      tail_bufs.emplace_back(d, P);
      std::vector<lasso_fr> rr; for (size_t i = local_rounds; i < num_rounds; i++) rr.push_back(rand[i].abi());
      d.chk(lasso_eq_evals(d.ctx, rr.data(), (uint32_t)rr.size(), tail_bufs.back().p), "lasso_eq_evals");
      cubic_rounds(lgP, P, fa, fb, tail_bufs.back().p, rand, local_rounds, coeffs, false, s_run, e, proof, r_out, &heads);
    }
    if (heads.empty()) {
      std::vector<lasso_fr*> ab(fa); ab.insert(ab.end(), fb.begin(), fb.end());
      heads.resize(2 * k);
      d.chk(lasso_read_heads(d.ctx, (const lasso_fr* const*)ab.data(), (uint32_t)(2 * k), heads.data()), "lasso_read_heads");
    }
    claims_a.clear(); claims_b.clear();
    for (size_t i = 0; i < k; i++) { claims_a.push_back(Sc::from_abi(heads[i])); claims_b.push_back(Sc::from_abi(heads[k + i])); }
    tail_bufs.clear();
    return proof;
  }
  // ---- BatchedGrandProductArgument::prove (grand_product.rs:101-201).
  // trees[c]: product tree of n_loc leaves (2*n_loc - 2 elements: layer k at offset n_loc*(2 - 2^(1-k))); in slab mode n_loc = n / P and the tree is the
  // rank's residue class of the global tree down to the global layer of 2P elements; the global layers of P, P/2, .., 2 elements are replicated in tops[c]
  // (P elements = the ranks' local roots, then their product tree).
  // leaf (capacity mode): trees[c] holds the layers ABOVE the leaves only (n_loc - 2 elements: layer k >= 1 at offset n_loc * (1 - 2^(1-k))); the bottom layer's
  // arrays are recomputed by cubic_rounds from *leaf, whose work_a / work_b this function points at the (by then dead) storage of layer 1
  // host_tops (one GPU, optional): the top of every tree on the host — tree c's layers of at most host_tops->len elements, back to back as they lie in the arena (the layer of
  // `len` elements first, the two-element layer last): those layers are proved without the device (host_cubic_rounds)
  struct HostTops { size_t len = 0; std::vector<ScVec> run; };   // run[c]: 2 * len - 2 elements
  // layer `layer_id` of whole trees on one GPU (P == 1, no capacity mode): the arrays bgpa_prove will pass to its sumcheck, enqueued ahead
  void enqueue_next_layer(const std::vector<lasso_fr*>& trees, size_t n, size_t layer_id, lasso_fr* d_table) {
    const size_t len = n >> layer_id, off = 2 * n - 2 * len;
    std::vector<lasso_fr*> A, B; for (auto* tr : trees) { A.push_back(tr + off); B.push_back(tr + off + len / 2); }
    enqueue_layer_ahead(A, B, len / 2, d_table);   // A = the layer's first half, B its second: arrays of len / 2
  }
  BatchedGrandProductArgument bgpa_prove(std::vector<lasso_fr*>& trees, std::vector<lasso_fr*>& tops, size_t n, const ScVec& roots, ScVec& rand_out, LeafLayer* leaf = nullptr, const HostTops* host_tops = nullptr) {
    Trace tr("BatchedGrandProductArgument.prove", d.ctx);
    BatchedGrandProductArgument out; const size_t k = trees.size(), num_layers = ceil_log2(n), n_loc = n / P;
    ScVec claims_to_verify = roots, rand;
    DBuf eq(d, std::max(n_loc / 4, P));   // the layer's (half) eq table: slab layers use n_loc/4 entries, replicated top layers at most P/2
    for (size_t layer_id = num_layers; layer_id-- > 0;) {
      const size_t len = n >> layer_id;                     // global layer `layer_id` has n/2^layer_id elements
      LASSO_REQUIRE(((size_t)1 << rand.size()) == len / 2);
      const size_t num_rounds_prod = ceil_log2(len / 2);
      if (host_tops && len <= host_tops->len && !(leaf && layer_id == 0)) {   // a layer of O(1) elements: the whole layer proof on the host (grand_product.rs:113-190, same transcript schedule)
        LASSO_REQUIRE(host_tops->run.size() == k && (P == 1 || host_tops->len == P));   // P > 1 (round 6): the replicated top layers P, P/2, .., 2 — every rank holds them whole
        const size_t off = 2 * host_tops->len - 2 * len;    // within the run, as in the arena: the layer of `len` elements starts 2 * len elements before the end (+ 2)
        std::vector<ScVec> ha(k), hb(k);
        for (size_t c = 0; c < k; c++) { const ScVec& r = host_tops->run[c]; ha[c].assign(r.begin() + off, r.begin() + off + len / 2); hb[c].assign(r.begin() + off + len / 2, r.begin() + off + len); }
        if (P == 1 && layer_id > 0 && 2 * len > host_tops->len && !leaf) enqueue_next_layer(trees, n, layer_id - 1, eq.p);   // the first layer the device proves: its first launch waits for its point from here on
        ScVec coeff_vec = t.challenge_vector("rand_coeffs_next_layer", claims_to_verify.size());
        Sc claim = Sc::zero(); for (size_t i = 0; i < claims_to_verify.size(); i++) claim += claims_to_verify[i] * coeff_vec[i];
        LayerProofBatched lp; ScVec rand_prod; std::vector<lasso_fr> heads;
        host_cubic_rounds(ha, hb, num_rounds_prod, rand, 0, coeff_vec, Sc::one(), claim, lp.proof, rand_prod, heads);
        for (size_t i = 0; i < k; i++) { lp.claims_prod_left.push_back(Sc::from_abi(heads[i])); lp.claims_prod_right.push_back(Sc::from_abi(heads[k + i])); }
        for (size_t i = 0; i < k; i++) { t.append_scalar("claim_prod_left", lp.claims_prod_left[i]); t.append_scalar("claim_prod_right", lp.claims_prod_right[i]); }
        Sc r_layer = t.challenge_scalar("challenge_r_layer");
        claims_to_verify.clear();
        for (size_t i = 0; i < k; i++) claims_to_verify.push_back(lp.claims_prod_left[i] + r_layer * (lp.claims_prod_right[i] - lp.claims_prod_left[i]));
        ScVec ext{r_layer}; ext.insert(ext.end(), rand_prod.begin(), rand_prod.end()); rand = ext;
        out.proof.push_back(std::move(lp));
        continue;
      }
      const bool slab = P > 1 && len >= 2 * P;              // this layer lives in the local trees; smaller ones in the replicated tops
      std::unique_ptr<HostClock> hco(new HostClock("layer transition: opening (arrays, coefficients, claim)"));
      std::vector<lasso_fr*> A, B;
      const bool bottom_leafless = leaf && layer_id == 0;
      if (bottom_leafless) {
        // layer 1's storage (n_loc / 2 elements, at the start of the leafless arena) is dead: layer 1's sumcheck is over.  It receives the bound arrays A', B' (n_loc / 4 each).
        leaf->work_a.clear(); leaf->work_b.clear();
        for (auto* tr : trees) { leaf->work_a.push_back(tr); leaf->work_b.push_back(tr + n_loc / 4); }
        if (eq_inline_off()) eq_half_local(rand, eq.p); else eq_half_lazy(rand, eq.p);
      } else if (slab || P == 1) {
        const size_t len_l = len / P, off = 2 * n_loc - 2 * len_l - (leaf ? n_loc : 0);
        for (auto* tr : trees) { A.push_back(tr + off); B.push_back(tr + off + len_l / 2); }
        if (eq_inline_off()) eq_half_local(rand, eq.p); else eq_half_lazy(rand, eq.p);        // poly_C_par :122 (the half the rounds read), built inside round 0's launch where possible
      } else {
        const size_t off = 2 * P - 2 * len;
        for (auto* tp : tops) { A.push_back(tp + off); B.push_back(tp + off + len / 2); }
        std::vector<lasso_fr> rr; for (auto& x : rand) rr.push_back(x.abi());
        d.chk(lasso_eq_evals(d.ctx, rr.data(), (uint32_t)rr.size(), eq.p), "lasso_eq_evals");
      }
      ScVec coeff_vec = t.challenge_vector("rand_coeffs_next_layer", claims_to_verify.size());
      Sc claim = Sc::zero(); for (size_t i = 0; i < claims_to_verify.size(); i++) claim += claims_to_verify[i] * coeff_vec[i];
      LayerProofBatched lp; ScVec rand_prod;
      hco.reset();
      next_layer_hook = nullptr;
      if (P == 1 && !leaf && layer_id > 0) next_layer_hook = [this, &trees, n, layer_id, &eq] { enqueue_next_layer(trees, n, layer_id - 1, eq.p); };
      lp.proof = prove_cubic_batched(claim, num_rounds_prod, slab, A, B, eq.p, rand, coeff_vec, rand_prod, lp.claims_prod_left, lp.claims_prod_right, bottom_leafless ? leaf : nullptr);
      next_layer_hook = nullptr;
      HostClock hcl("layer transition: closing claims + r_layer");
      for (size_t i = 0; i < k; i++) { t.append_scalar("claim_prod_left", lp.claims_prod_left[i]); t.append_scalar("claim_prod_right", lp.claims_prod_right[i]); }
      Sc r_layer = t.challenge_scalar("challenge_r_layer");
      claims_to_verify.clear();
      for (size_t i = 0; i < k; i++) claims_to_verify.push_back(lp.claims_prod_left[i] + r_layer * (lp.claims_prod_right[i] - lp.claims_prod_left[i]));
      ScVec ext{r_layer}; ext.insert(ext.end(), rand_prod.begin(), rand_prod.end()); rand = ext;
      out.proof.push_back(std::move(lp));
    }
    rand_out = rand; return out;
  }

  // ---- BulletReductionProof::prove + DotProductProofLog::prove (bullet.rs:40-154, dot_product.rs:167-249), blinds of x and y are zero on this path.
  // The generator vector is never folded: G^(k)_i = sum_b w_b G_{b*n_k+i} with w the running tensor of u^{+-1}, so every L_k / R_k / g_hat is one
  // MSM over the ORIGINAL (precomputed) generators with scalars a (x) w — same group elements as the reference's fold-then-MSM.
  Pt msm_dev(const PolyCommitmentGens& g, const lasso_fr* d_scalars, size_t n) {
    lasso_point out; d.chk(lasso_msm_dev(d.ctx, g.bases, d_scalars, n, &out), "lasso_msm_dev");
    return Pt::from_abi(out);
  }
  // d_a0 = x (here L*Z), d_b0 = a (here the R half of the eq table), both of length n and resident on the device; a_bytes = serialize(a)
  DotProductProofLog dot_product_log_prove(const PolyCommitmentGens& g, DBuf& d_a0, DBuf& d_b0, const std::vector<uint8_t>& a_bytes, const Sc& y) {
    static_assert(sizeof(Sc) == sizeof(lasso_fr), "ScVec is uploaded as an array of lasso_fr");
    Trace tr("DotProductProofLog.prove", d.ctx);
    t.append_protocol_name("dot product proof (log)");
    const size_t n = g.n; LASSO_REQUIRE(d_a0.n == n && d_b0.n == n && a_bytes.size() == 32 * n);
    const size_t lg_n = ceil_log2(n);
    Sc dd, r_delta, r_beta; ScVec v1, v2;
    {
      HostClock hc("opening: random tape");
      dd = tape.random_scalar("d"); r_delta = tape.random_scalar("r_delta"); r_beta = tape.random_scalar("r_delta");   // sic: dot_product.rs:189
      v1 = tape.random_vector("blinds_vec_1", 2 * lg_n); v2 = tape.random_vector("blinds_vec_2", 2 * lg_n);
    }
    // Slab mode (one proof over P GPUs): a, b and the fold weights are sqrt(N)-sized and replicated, but the MSMs — Cx, every round's L and R, delta: the dependent
    // chain of the opening — are SHARED: each rank adds up the terms of its residue class of the generators (lasso_*_slab over gens.bases_slab) and the ranks
    // all-gather and add the partial points (rank order everywhere: the same group element on every rank, the same compressed bytes as one GPU).
    const bool shard = d.comm.sharded() && g.slab_open;
    const uint32_t cw = (uint32_t)d.comm.world, cr = (uint32_t)d.comm.rank;
    auto sum_points = [&](const lasso_point* mine, size_t count, Pt* out) {
      std::vector<lasso_point> all(count * cw);
      d.comm.allgather(mine, all.data(), count * sizeof(lasso_point));
      for (size_t i = 0; i < count; i++) { Pt acc = Pt::from_abi(all[i]); for (size_t r2 = 1; r2 < cw; r2++) acc = acc + Pt::from_abi(all[r2 * count + i]); out[i] = acc; }
    };
    DotProductProofLog P; uint8_t buf[32];
    // a, b and the generator-fold weights live on the device for the whole reduction, in ping-pong pairs: round k's fold (bullet.rs:127-132)
    // is applied by the same call that computes round k+1's c_L, c_R, L, R (lasso_bullet_round) — one host round trip per round.
    DBuf d_a1(d, n / 2 ? n / 2 : 1), d_b1(d, n / 2 ? n / 2 : 1), d_w0(d, n), d_w1(d, n);
    // the initial fold weight w = [1] = EqPolynomial([]).evals(): written by a kernel on the stream (a synchronous 32-byte upload cost every opening ~40 us of idle device)
    d.chk(lasso_eq_evals(d.ctx, nullptr, 0, d_w0.p), "lasso_eq_evals");
    {   // Cx = <x, G> + 0*h (commitments.rs:84-93) on the device while the host computes Cy = y*G_1[0] + 0*h (commitments.rs:78-82)
      lasso_point cx;
      d.chk(lasso_defer_next(d.ctx), "lasso_defer_next");
      if (shard) d.chk(lasso_msm_dev_slab(d.ctx, g.bases_slab, d_a0.p, n, cw, cr, nullptr, nullptr, &cx), "lasso_msm_dev_slab");
      else d.chk(lasso_msm_dev(d.ctx, g.bases, d_a0.p, n, &cx), "lasso_msm_dev");
      uint8_t cy[32]; compress_one(g.Qmul.mul(y), cy);
      d.chk(lasso_result_wait(d.ctx, (lasso_fr*)&cx, 4), "lasso_result_wait");
      Pt cxp = Pt::from_abi(cx); if (shard) sum_points(&cx, 1, &cxp);
      compress_one(cxp, buf); t.append_point_bytes("Cx", buf);
      t.append_point_bytes("Cy", cy);
    }
    // bullet reduction (bullet.rs:40-154), blind = blind_x + blind_y = 0
    lasso_fr *a_cur = d_a0.p, *a_nxt = d_a1.p, *b_cur = d_b0.p, *b_nxt = d_b1.p, *w_cur = d_w0.p, *w_nxt = d_w1.p;
    Sc blind_fin = Sc::zero(); size_t nk = n, nw = 1, round = 0;
    bool have_u = false; lasso_fr ua, uia;
    // Rounds launched AHEAD (one GPU): the kernel of round k+1 is enqueued behind round k before round k's L, R are back; it waits on the device for the challenge the host posts
    // (lasso_bullet_post) — the host turn between two rounds is ~4 us of work, the launch and its dispatch were 27 us more (include/lasso_hip.h).  `queued`: this round's kernel is
    // already in the stream, its ping-pong buffers already swapped.
    const bool ahead = !shard && !d.no_ahead() && lasso_bullet_ahead_ok(d.ctx, g.bases) == 1;
    bool queued = false;
    // The END of the opening enqueued ahead too (round 6, lasso_bullet_tail_ahead): when the round in flight is the last one, the chain fold -> heads -> delta MSM goes into the
    // stream behind it, waiting for the challenge this loop draws last; three launches and two hand-offs leave the critical path.  tail_queued: posted after the loop.
    const bool tail_ahead = ahead && lasso_bullet_tail_ahead_ok(d.ctx, g.bases) == 1;
    bool tail_queued = false;
    auto enqueue_tail = [&](size_t nw_now) {   // a_cur, b_cur: the two-element state the round in flight is writing; w_cur: its nw_now = n / 2 weights
      lasso_fr sc = dd.abi(); lasso_fr tl[2] = {Sc::zero().abi(), r_delta.abi()};
      const int32_t rc = lasso_bullet_tail_ahead(d.ctx, g.bases, n, a_cur, b_cur, w_cur, nw_now, w_nxt, &sc, tl);
      if (rc == 0) { std::swap(w_cur, w_nxt); tail_queued = true; } else if (rc != LASSO_ERR_UNSUPPORTED) d.chk(rc, "lasso_bullet_tail_ahead");
    };
    auto enqueue_next = [&](size_t nk_next) {   // the round after the one in flight: folds (a_cur, b_cur, w_cur) — being written by the launch in flight, stream order — to length nk_next
      lasso_fr bl[2] = {v1[round + 1].abi(), v2[round + 1].abi()};
      d.chk(lasso_bullet_round_ahead(d.ctx, g.bases, n, a_cur, b_cur, w_cur, a_nxt, b_nxt, w_nxt, nk_next, bl), "lasso_bullet_round_ahead");
      std::swap(a_cur, a_nxt); std::swap(b_cur, b_nxt); std::swap(w_cur, w_nxt);
      queued = true;
    };
    while (nk != 1) {
      const Sc& blind_L = v1[round]; const Sc& blind_R = v2[round];
      lasso_fr blinds[2] = {blind_L.abi(), blind_R.abi()};
      lasso_point LR[2];
      if (have_u && queued) {   // this round's kernel is waiting for the challenge drawn at the end of the previous iteration
        d.chk(lasso_bullet_post(d.ctx, &ua, &uia), "lasso_bullet_post");
        queued = false;
        if (nk / 2 != 1) enqueue_next(nk / 2); else if (tail_ahead) enqueue_tail(nw);   // (nk == 2: this round writes w of n / nk = nw entries)
        d.chk(lasso_result_wait(d.ctx, (lasso_fr*)LR, 8), "lasso_result_wait");
      } else if (have_u) {   // fold with the previous challenge (length 2nk -> nk), then this round's L, R
        if (shard) d.chk(lasso_bullet_round_slab(d.ctx, g.bases_slab, n, cw, cr, a_cur, b_cur, w_cur, a_nxt, b_nxt, w_nxt, nk, &ua, &uia, blinds, LR), "lasso_bullet_round_slab");
        else d.chk(lasso_bullet_round(d.ctx, g.bases, n, a_cur, b_cur, w_cur, a_nxt, b_nxt, w_nxt, nk, &ua, &uia, blinds, LR), "lasso_bullet_round");
        std::swap(a_cur, a_nxt); std::swap(b_cur, b_nxt); std::swap(w_cur, w_nxt);
      } else {
        // round 0 needs no challenge: it runs on the device while the host absorbs the a-vector (dot_product.rs:196), the longest message of the opening
        d.chk(lasso_defer_next(d.ctx), "lasso_defer_next");
        if (shard) d.chk(lasso_bullet_round_slab(d.ctx, g.bases_slab, n, cw, cr, a_cur, b_cur, w_cur, nullptr, nullptr, nullptr, nk, nullptr, nullptr, blinds, LR), "lasso_bullet_round_slab");
        else d.chk(lasso_bullet_round(d.ctx, g.bases, n, a_cur, b_cur, w_cur, nullptr, nullptr, nullptr, nk, nullptr, nullptr, blinds, LR), "lasso_bullet_round");
        if (ahead && nk / 2 != 1) enqueue_next(nk / 2);
        { HostClock hc("opening: append a_vec"); t.append_scalars_bytes("a", a_bytes); }
        d.chk(lasso_result_wait(d.ctx, (lasso_fr*)LR, 8), "lasso_result_wait");
      }
      std::vector<uint8_t> cb; Sc u, u_inv;
      {
        HostClock hc("opening: round host work");
        std::vector<Pt> two{Pt::from_abi(LR[0]), Pt::from_abi(LR[1])};
        if (shard) sum_points(LR, 2, two.data());
        compress_batch(two, cb);
        t.append_point_bytes("L", &cb[0]); t.append_point_bytes("R", &cb[32]);
        u = t.challenge_scalar("u"); u_inv = u.inverse();
      }
      ua = u.abi(); uia = u_inv.abi(); have_u = true;
      blind_fin = blind_fin + blind_L * u * u + blind_R * u_inv * u_inv;
      P.L_vec.insert(P.L_vec.end(), cb.begin(), cb.begin() + 32); P.R_vec.insert(P.R_vec.end(), cb.begin() + 32, cb.end());
      nk /= 2; nw *= 2; round++;
    }
    if (!have_u) { HostClock hc("opening: append a_vec"); t.append_scalars_bytes("a", a_bytes); }   // n = 1: no round absorbed it
    if (tail_queued) {   // fold, heads and the delta MSM are in the stream: release them, compute beta under them, collect the point and the two heads in one hand-off
      d.chk(lasso_bullet_post(d.ctx, &ua, &uia), "lasso_bullet_post");
      { HostClock hc("opening: delta/beta scalar mults"); compress_one(g.Qmul.mul(dd) + g.hmul.mul(r_beta), P.beta); }
      lasso_fr six[6];
      d.chk(lasso_result_wait(d.ctx, six, 6), "lasso_result_wait");
      lasso_point dl; memcpy(&dl, six, sizeof(dl));
      compress_one(Pt::from_abi(dl), P.delta);
      const Sc x_hat = Sc::from_abi(six[4]), a_hat = Sc::from_abi(six[5]), y_hat = x_hat * a_hat;
      t.append_point_bytes("delta", P.delta); t.append_point_bytes("beta", P.beta);
      Sc c = t.challenge_scalar("c");
      P.z1 = dd + c * y_hat;
      P.z2 = a_hat * (c * blind_fin + r_beta) + r_delta;
      return P;
    }
    if (have_u) {   // the last challenge still folds a, b (length 2 -> 1) and the weights
      d.chk(lasso_bullet_fold(d.ctx, a_cur, b_cur, 2, w_cur, nw / 2, w_nxt, &ua, &uia), "lasso_bullet_fold");
      std::swap(w_cur, w_nxt);
    }
    lasso_fr heads[2]; const lasso_fr* hp[2] = {a_cur, b_cur};
    d.chk(lasso_read_heads(d.ctx, hp, 2, heads), "lasso_read_heads");
    Sc x_hat = Sc::from_abi(heads[0]), a_hat = Sc::from_abi(heads[1]), y_hat = x_hat * a_hat;
    // delta = d*g_hat + r_delta*h with g_hat = G[0] after all folds = sum_j w_j G_j: one MSM over the resident weights scaled by d, plus the h term
    lasso_point dl; Pt dlp;
    {
      lasso_fr sc = dd.abi(); lasso_fr tl[2] = {Sc::zero().abi(), r_delta.abi()};
      if (shard) {
        d.chk(lasso_msm_dev_slab(d.ctx, g.bases_slab, w_cur, n, cw, cr, &sc, cr == 0 ? tl : nullptr, &dl), "lasso_msm_dev_slab"); sum_points(&dl, 1, &dlp);
        HostClock hc("opening: delta/beta scalar mults");
        compress_one(dlp, P.delta); compress_one(g.Qmul.mul(dd) + g.hmul.mul(r_beta), P.beta);
      } else {
        // beta = d * Q + r_beta * h (two fixed-base scalar multiplications, ~60 us of host time) while the device computes delta — as Cx / Cy above
        d.chk(lasso_defer_next(d.ctx), "lasso_defer_next");
        d.chk(lasso_msm_dev_scaled(d.ctx, g.bases, w_cur, n, &sc, tl, &dl), "lasso_msm_dev_scaled");
        { HostClock hc("opening: delta/beta scalar mults"); compress_one(g.Qmul.mul(dd) + g.hmul.mul(r_beta), P.beta); }
        d.chk(lasso_result_wait(d.ctx, (lasso_fr*)&dl, 4), "lasso_result_wait");
        dlp = Pt::from_abi(dl); compress_one(dlp, P.delta);
      }
    }
    t.append_point_bytes("delta", P.delta); t.append_point_bytes("beta", P.beta);
    Sc c = t.challenge_scalar("c");
    P.z1 = dd + c * y_hat;
    P.z2 = a_hat * (c * blind_fin + r_beta) + r_delta;
    return P;
  }
  // ---- PolyEvalProof::prove (dense_mlpoly.rs:302-359), blinds None
  DotProductProofLog poly_eval_prove(const lasso_fr* d_poly, size_t num_vars, const ScVec& r, const Sc& Zr, const PolyCommitmentGens& g) {
    Trace tr("DensePolyEval.prove", d.ctx);
    t.append_protocol_name("polynomial evaluation proof");
    LASSO_REQUIRE(r.size() == num_vars);
    size_t left = num_vars / 2, right = num_vars - left;
    const size_t Ln = (size_t)1 << left, Rn = (size_t)1 << right;
    DBuf d_LZ(d, Rn), d_R(d, Rn);
    std::vector<uint8_t> a_bytes(32 * Rn);
    std::vector<lasso_fr> rl, rr; for (size_t i = 0; i < left; i++) rl.push_back(r[i].abi()); for (size_t i = left; i < num_vars; i++) rr.push_back(r[i].abi());
    // d_poly == nullptr: the compact merged polynomial of the operations (capacity mode).  L*Z is a sum over row blocks — one polynomial each, lifted into a
    // scratch array in turn: L*Z = sum_b L[b * rows_per ..) * Z_b; the zero padding contributes nothing
    const bool blocks = d_poly == nullptr;
    const size_t rows_per = blocks ? dense.s / Rn : 0, nb = blocks ? 2 * dense.C : 0;
    if (blocks) LASSO_REQUIRE(dense.compact && num_vars == dense.nv_l && rows_per >= 1 && nb * rows_per <= Ln);
    if (this->P == 1 && blocks) {
      DBuf d_L(d, Ln), Mb(d, nb * Rn), ones(d, nb), tmp(d, s_loc);
      d.chk(lasso_eq_evals(d.ctx, rl.data(), (uint32_t)left, d_L.p), "lasso_eq_evals");
      d.chk(lasso_eq_evals(d.ctx, rr.data(), (uint32_t)right, d_R.p), "lasso_eq_evals");
      for (size_t b = 0; b < nb; b++) { dense.lift_block(b, tmp.p); d.chk(lasso_matvec_left_dev(d.ctx, tmp.p, d_L.p + b * rows_per, rows_per, Rn, Mb.p + b * Rn), "lasso_matvec_left_dev"); }
      std::vector<lasso_fr> one_h(nb, Sc::one().abi());
      d.chk(lasso_upload(d.ctx, ones.p, one_h.data(), nb * sizeof(lasso_fr)), "lasso_upload");
      d.chk(lasso_matvec_left_dev(d.ctx, Mb.p, ones.p, nb, Rn, d_LZ.p), "lasso_matvec_left_dev");
      d.chk(lasso_fr_to_bytes(d.ctx, d_R.p, Rn, a_bytes.data()), "lasso_fr_to_bytes");
    } else if (this->P == 1) {
      // everything stays on the device: the two halves of the eq table (eq_poly.rs:44-52), L*Z (dense_mlpoly.rs:184-207); only serialize(R) comes back
      DBuf d_L(d, Ln);
      d.chk(lasso_eq_evals(d.ctx, rl.data(), (uint32_t)left, d_L.p), "lasso_eq_evals");
      d.chk(lasso_eq_evals(d.ctx, rr.data(), (uint32_t)right, d_R.p), "lasso_eq_evals");
      d.chk(lasso_matvec_left_dev(d.ctx, d_poly, d_L.p, Ln, Rn, d_LZ.p), "lasso_matvec_left_dev");
      d.chk(lasso_fr_to_bytes(d.ctx, d_R.p, Rn, a_bytes.data()), "lasso_fr_to_bytes");
    } else {
      // slab mode: d_poly holds this rank's columns (= rank mod P) of every row, so the rank computes its entries of L*Z in full; the vector is
      // all-gathered and the opening's bullet reduction (latency-bound, sqrt(n)-sized) runs replicated on every rank with the same transcript
      ScVec L = eq_evals_host(r.data(), left);
      std::vector<lasso_fr> Lh(Ln), LZh(Rn); for (size_t i = 0; i < Ln; i++) Lh[i] = L[i].abi();
      const size_t Pw = this->P, r_loc = Rn / Pw; LASSO_REQUIRE(Rn >= Pw);
      std::vector<lasso_fr> mine(r_loc), all(Rn);
      if (blocks) {
        DBuf tmp(d, s_loc); std::vector<lasso_fr> part(r_loc); ScVec acc(r_loc, Sc::zero());
        for (size_t b = 0; b < nb; b++) {
          dense.lift_block(b, tmp.p);
          d.chk(lasso_matvec_left(d.ctx, tmp.p, Lh.data() + b * rows_per, rows_per, r_loc, part.data()), "lasso_matvec_left");
          for (size_t j = 0; j < r_loc; j++) acc[j] += Sc::from_abi(part[j]);
        }
        for (size_t j = 0; j < r_loc; j++) mine[j] = acc[j].abi();
      } else
      d.chk(lasso_matvec_left(d.ctx, d_poly, Lh.data(), Ln, r_loc, mine.data()), "lasso_matvec_left");
      d.comm.allgather(mine.data(), all.data(), r_loc * sizeof(lasso_fr));
      for (size_t g2 = 0; g2 < Pw; g2++) for (size_t j = 0; j < r_loc; j++) LZh[j * Pw + g2] = all[g2 * r_loc + j];
      d.chk(lasso_upload(d.ctx, d_LZ.p, LZh.data(), Rn * sizeof(lasso_fr)), "lasso_upload");
      d.chk(lasso_eq_evals(d.ctx, rr.data(), (uint32_t)right, d_R.p), "lasso_eq_evals");
      d.chk(lasso_fr_to_bytes(d.ctx, d_R.p, Rn, a_bytes.data()), "lasso_fr_to_bytes");
    }
    return dot_product_log_prove(g, d_LZ, d_R, a_bytes, Zr);
  }
  // ---- openings prepared ahead of the transcript.  joint_open's point is (ch, r): k challenges ch drawn when the opening starts, r known earlier
  // (the grand-product argument's random point).  The L*Z of PolyEvalProof::prove (dense_mlpoly.rs:184-207, :336-340) factors through ch:
  //   L = eq(ch) (x) eq(r[0 .. left-k))   =>   L*Z = sum_b eq(ch)[b] * M_b,   M_b = eq(r[0 .. left-k)) * Z_b   (Z_b = block b of 2^(left-k) rows),
  // so the pass over the whole polynomial (the M_b) and the right-half table depend on r only.  They are computed on the side context (own
  // stream) while the main context sits in a latency-bound phase — the second grand-product argument, or the previous opening's bullet rounds —
  // and at opening time a 2^k-row mat-vec combines them.
  struct OpenPrep { DBuf M, R; size_t k = 0, left = 0, right = 0; bool ready = false; };
  OpenPrep prep_open(const lasso_fr* d_poly, size_t k, const ScVec& r) {
    OpenPrep pr; const size_t nv = k + r.size();
    pr.k = k; pr.left = nv / 2; pr.right = nv - pr.left;
    if (P != 1 || pr.left < k || side_off()) return pr;
    lasso_ctx* sc = d.side();
    const size_t lrows = (size_t)1 << (pr.left - k), Rn = (size_t)1 << pr.right, nb = (size_t)1 << k;
    pr.M = DBuf(d, nb * Rn); pr.R = DBuf(d, Rn);
    DBuf Lp(d, lrows);
    std::vector<lasso_fr> rl, rr; for (size_t i = 0; i + k < pr.left; i++) rl.push_back(r[i].abi()); for (size_t i = pr.left - k; i < r.size(); i++) rr.push_back(r[i].abi());
    d.chk_side(lasso_eq_evals(sc, rl.data(), (uint32_t)rl.size(), Lp.p), "lasso_eq_evals");
    d.chk_side(lasso_eq_evals(sc, rr.data(), (uint32_t)rr.size(), pr.R.p), "lasso_eq_evals");
    for (size_t b = 0; b < nb; b++) d.chk_side(lasso_matvec_left_dev(sc, d_poly + b * lrows * Rn, Lp.p, lrows, Rn, pr.M.p + b * Rn), "lasso_matvec_left_dev");
    side_keep.push_back(std::move(Lp));
    pr.ready = true; return pr;
  }
  void side_sync() { if (!side_keep.empty() || true) { d.chk_side(lasso_sync(d.side()), "lasso_sync"); side_keep.clear(); } }
  DotProductProofLog poly_eval_prove_prepped(OpenPrep& pr, const ScVec& ch, const Sc& Zr, const PolyCommitmentGens& g) {
    Trace tr("DensePolyEval.prove", d.ctx);
    t.append_protocol_name("polynomial evaluation proof");
    LASSO_REQUIRE(pr.ready && ch.size() == pr.k);
    side_sync();
    const size_t Rn = (size_t)1 << pr.right, nb = (size_t)1 << pr.k;
    DBuf d_LZ;
    if (pr.k == 0) d_LZ = std::move(pr.M);
    else {
      DBuf w(d, nb); d_LZ = DBuf(d, Rn);
      std::vector<lasso_fr> cc; for (auto& c : ch) cc.push_back(c.abi());
      d.chk(lasso_eq_evals(d.ctx, cc.data(), (uint32_t)cc.size(), w.p), "lasso_eq_evals");
      d.chk(lasso_matvec_left_dev(d.ctx, pr.M.p, w.p, nb, Rn, d_LZ.p), "lasso_matvec_left_dev");
    }
    std::vector<uint8_t> a_bytes(32 * Rn);
    d.chk(lasso_fr_to_bytes(d.ctx, pr.R.p, Rn, a_bytes.data()), "lasso_fr_to_bytes");
    return dot_product_log_prove(g, d_LZ, pr.R, a_bytes, Zr);
  }
  // ---- CombinedTableEvalProof::prove (subtables/mod.rs:285-313) / the two n-to-1 reductions of HashLayerProof (memory_checking.rs:370-449)
  DotProductProofLog joint_open(const char* evals_label, const char* challenge_label, const char* joint_label, ScVec evals, bool pad_before_append,
                                const lasso_fr* d_poly, size_t num_vars, const ScVec& r, const PolyCommitmentGens& g, OpenPrep* prep = nullptr) {
    if (pad_before_append) evals.resize(next_pow2(evals.size()), Sc::zero());
    t.append_scalars(evals_label, evals);
    ScVec ch = t.challenge_vector(challenge_label, ceil_log2(evals.size()));
    evals.resize(next_pow2(evals.size()), Sc::zero());     // DensePolynomial::new_padded for the unpadded case (:420)
    size_t len = evals.size();
    for (size_t i = ch.size(); i-- > 0;) { len /= 2; for (size_t k = 0; k < len; k++) evals[k] = evals[2 * k] + ch[i] * (evals[2 * k + 1] - evals[2 * k]); }   // bound_poly_var_bot
    LASSO_REQUIRE(len == 1);
    Sc joint = evals[0];
    ScVec r_joint = ch; r_joint.insert(r_joint.end(), r.begin(), r.end());
    t.append_scalar(joint_label, joint);
    if (prep && prep->ready) { LASSO_REQUIRE(prep->k == ch.size() && prep->k + r.size() == num_vars); return poly_eval_prove_prepped(*prep, ch, joint, g); }
    return poly_eval_prove(d_poly, num_vars, r_joint, joint, g);
  }

  // ---- SparsePolynomialEvaluationProof::prove (surge.rs:119-211)
  void prove(const ScVec& r) {
    Trace tr_all("SparsePoly.prove", d.ctx);
    t.append_protocol_name("Lasso SparsePolynomialEvaluationProof");
    LASSO_REQUIRE(r.size() == ceil_log2(s));
    // Subtables::new (subtables/mod.rs:116-129)
    std::unique_ptr<Trace> sp(new Trace("Subtables.new", d.ctx));
    // the subtables as integers, written by the device (64 K entries each: nothing to compute on the host and upload); they stay until E is
    // committed (the commitment's scalars are T[dim] as integers)
    std::vector<DBufU32> tables_u32; const bool ints = S.integer_tables(); const uint32_t table_max = ints ? S.max_table_value() : 0;
    if (ints) for (size_t i = 0; i < S.num_subtables(); i++) {
      tables_u32.emplace_back(d, m); tables.emplace_back(d, m);
      d.chk(lasso_materialize_subtable_u32(d.ctx, &S.abi, (uint32_t)i, tables_u32.back().p), "lasso_materialize_subtable_u32");
    } else {   // Spark (unconfirmed): subtable i = EqPolynomial(tau_i).evals(), field elements from the start
      const std::vector<ScVec> tau = S.spark_point();
      for (size_t i = 0; i < S.num_subtables(); i++) {
        tables.emplace_back(d, m);
        std::vector<lasso_fr> tv; for (auto& x : tau[i]) tv.push_back(x.abi());
        d.chk(lasso_eq_evals(d.ctx, tv.data(), (uint32_t)tv.size(), tables.back().p), "lasso_eq_evals");
      }
    }
    if (side_off() || P != 1) {} else d.chk(lasso_sync(d.ctx), "lasso_sync");   // the side context reads them
    size_t n_E = next_pow2(alpha * s); nv_derefs = ceil_log2(n_E);
    combined_E = DBuf(d, n_E / P);
    DBuf eq(d, s_loc);
    // The commitment of E needs only E's INTEGER values (a 4-byte gather per lookup), so on one GPU the field-element side of Subtables::new — the
    // tables lifted to Fr, E = T[dim] as 32-byte elements — and the eq table of r run on the side context UNDER the commitment's MSM (VALU-bound;
    // these are HBM-bound) instead of in front of it.
    const bool side_new = P == 1 && !side_off() && ints;
    lasso_ctx* fc = side_new ? d.side() : d.ctx;
    auto fchk = [&](int32_t rc, const char* what) { if (side_new) d.chk_side(rc, what); else d.chk(rc, what); };
    auto field_side = [&] {
      if (ints) for (size_t i = 0; i < tables.size(); i++) fchk(lasso_fr_from_u32(fc, tables_u32[i].p, m, tables[i].p), "lasso_fr_from_u32");
      if (n_E > alpha * s) fchk(lasso_zero(fc, combined_E.p + alpha * s_loc, (n_E - alpha * s) / P * sizeof(lasso_fr)), "lasso_zero");
      for (size_t i = 0; i < alpha; i++)
        fchk(lasso_gather(fc, tables[S.memory_to_subtable_index(i)].p, dense.dim_u32[S.memory_to_dimension_index(i)].p, s_loc, combined_E.p + i * s_loc), "lasso_gather");
    };
    ProofWriter W;
    PolyCommitment comm_derefs;
    DBufU32 E_u32;   // E as integers (one GPU): the commitment's scalars, and what the primary sumcheck's first round and first bind read (4 bytes per element instead of 32)
    if (P == 1 && ints) {   // one 4-byte gather per lookup instead of converting the 32-byte elements back
      E_u32 = DBufU32(d, n_E);
      if (n_E > alpha * s) d.chk(lasso_zero(d.ctx, E_u32.p + alpha * s, (n_E - alpha * s) * sizeof(uint32_t)), "lasso_zero");
      for (size_t i = 0; i < alpha; i++)
        d.chk(lasso_gather_u32(d.ctx, tables_u32[S.memory_to_subtable_index(i)].p, dense.dim_u32[S.memory_to_dimension_index(i)].p, s, E_u32.p + i * s), "lasso_gather_u32");
      if (side_new) {
        field_side();
        std::vector<lasso_fr> rr; for (auto& x : r) rr.push_back(x.abi());
        d.chk_side(lasso_eq_evals(fc, rr.data(), (uint32_t)rr.size(), eq.p), "lasso_eq_evals");
      } else field_side();
      sp.reset(), sp.reset(new Trace("Subtables.commit", d.ctx));
      comm_derefs = hyrax_commit(d, combined_E.p, nv_derefs, gens.gens_derefs, E_u32.p, table_max);
      if (side_new) side_sync();
    } else {
      field_side();
      sp.reset(), sp.reset(new Trace("Subtables.commit", d.ctx));
      comm_derefs = hyrax_commit(d, combined_E.p, nv_derefs, gens.gens_derefs);
    }
    tables_u32.clear();
    // The claim (subtables/mod.rs:187-216) depends on r and E only, not on the transcript: its kernels run while the host absorbs the
    // commitment (4096 compressed rows at 2^24: 0.4 ms of Keccak during which the device would otherwise idle)
    if (!side_new) eq_evals_local(r, eq.p);
    std::vector<const lasso_fr*> Eptr; for (size_t i = 0; i < alpha; i++) Eptr.push_back(E(i));
    std::vector<lasso_fr> claim_abi(1);
    d.chk(lasso_defer_next(d.ctx), "lasso_defer_next");
    d.chk(lasso_combine_claim(d.ctx, &S.abi, Eptr.data(), eq.p, s_loc, claim_abi.data()), "lasso_combine_claim");
    t.append_message("subtable_evals_commitment", "begin_subtable_evals_commitment");
    append_poly_commitment(t, "comm_poly_row_col_ops_val", comm_derefs);
    t.append_message("subtable_evals_commitment", "end_subtable_evals_commitment");
    W.pts_vec(comm_derefs.compressed);
    // claim
    sp.reset(), sp.reset(new Trace("Subtables.compute_sumcheck_claim", d.ctx));
    d.chk(lasso_result_wait(d.ctx, claim_abi.data(), 1), "lasso_result_wait");
    d.comm.sum(claim_abi);
    Sc claimed_eval = Sc::from_abi(claim_abi[0]);
    t.append_scalar("claim_eval_scalar_product", claimed_eval);
    // primary sumcheck on clones of E_i and the eq polynomial (surge.rs:151-172)
    ScVec r_z;
    sp.reset(), sp.reset(new Trace("Sumcheck.prove", d.ctx));
    ScVec sumcheck_heads;
    {
      // the sumcheck binds its polynomials; E itself must survive (the openings read it).  Linear strategies: the first bind reads E and writes the
      // half-length work arrays (no clone, surge.rs:151); LT (every polynomial enters the combine kernel and is bound in place): a clone
      const bool no_clone = S.linear();
      const size_t wl = no_clone && s_loc >= 4 ? s_loc / 2 : s_loc;   // fewer than two local rounds: linear_rounds copies the (tiny) arrays instead
      // capacity mode: in pieces of s_loc elements — the size of an operations' tree, which takes their place afterwards (no buffer of another size to return to the driver)
      const size_t per_piece = d.capacity ? s_loc / wl : alpha, n_pieces = (alpha + per_piece - 1) / per_piece;
      std::vector<DBuf> work; for (size_t i = 0; i < n_pieces; i++) work.emplace_back(d, d.capacity ? s_loc : alpha * wl);
      // LT: the clone of surge.rs:151 and the scaling of the LT memories are one pass (lasso_lt_prescale with a source): E itself is only read
      std::vector<lasso_fr*> polys; for (size_t i = 0; i < alpha; i++) polys.push_back(work[i / per_piece].p + (i % per_piece) * wl); polys.push_back(eq.p);
      static const bool u32_off = [] { const char* e = getenv("LASSO_SUMCHECK_U32"); return e && e[0] == '0'; }();   // A/B switch
      // linear strategies: any table values; LT: only because its subtables hold bits (the integer round needs entries 0 / 1)
      std::vector<const uint32_t*> Eu32; if (P == 1 && (no_clone || table_max <= 1) && E_u32.p && !u32_off && ceil_log2(s) > 0) for (size_t i = 0; i < alpha; i++) Eu32.push_back(E_u32.p + i * s);
      SumcheckProof sp = prove_arbitrary(ceil_log2(s), s_loc, polys, S.sumcheck_poly_degree(), r, r_z, &sumcheck_heads, &Eptr, Eu32.empty() ? nullptr : &Eu32);
      sp.write(W);
    }
    E_u32 = DBufU32();
    if (d.capacity) eq.reset();   // bound down to one element by the sumcheck: nothing reads it again (parked: a tree of the operations takes its place)
    W.sc(claimed_eval);
    // eval_derefs = E_i(r_z) (surge.rs:175-176)
    sp.reset(), sp.reset(new Trace("CombinedEval.prove", d.ctx));
    DBuf chis(d, s_loc);
    auto evaluate_at = [&](const std::vector<const lasso_fr*>& polys, const ScVec& point, size_t n_loc, DBuf& chi) {
      eq_evals_local(point, chi.p);
      std::vector<lasso_fr> out(polys.size());
      d.chk(lasso_multi_dot(d.ctx, polys.data(), (uint32_t)polys.size(), chi.p, n_loc, out.data()), "lasso_multi_dot");
      d.comm.sum(out);
      ScVec v; for (auto& o : out) v.push_back(Sc::from_abi(o)); return v;
    };
    // E_i(r_z): the value the sumcheck's last bind left behind (same field element as DensePolynomial::evaluate, without another pass over E_i)
    ScVec eval_derefs = sumcheck_heads.size() == alpha && ceil_log2(s) > 0 ? sumcheck_heads : evaluate_at(Eptr, r_z, s_loc, chis);
    W.sc_arr(eval_derefs);
    t.append_protocol_name("Lasso CombinedTableEvalProof");
    joint_open("evals_ops_val", "challenge_combine_n_to_one", "joint_claim_eval", eval_derefs, true, combined_E.p, nv_derefs, r_z, gens.gens_derefs).write(W);
    // memory checking (surge.rs:186-199)
    sp.reset(), sp.reset(new Trace("MemoryChecking.prove", d.ctx));
    ScVec r_hash = t.challenge_vector("challenge_r_hash", 2);
    memory_checking_prove(r_hash[0], r_hash[1], Eptr, chis, W);
    sp.reset();
    proof_bytes.swap(W.b);
    HostClock::dump();
  }

  void memory_checking_prove(const Sc& gamma, const Sc& tau, const std::vector<const lasso_fr*>& Eptr, DBuf& chis, ProofWriter& W) {
    t.append_protocol_name("Lasso MemoryCheckingProof");
    // Subtables::to_grand_products -> GrandProducts::new (subtables/mod.rs:134-175, memory_checking.rs:175-217)
    lasso_fr g = gamma.abi(), ta = tau.abi();
    std::unique_ptr<Trace> sp(new Trace("Subtables.to_grand_products", d.ctx));
    std::vector<DBuf> t_init, t_read, t_write, t_final;
    // capacity mode: the read / write trees without their leaf layers (half of each tree); the bottom layer's sumcheck recomputes the fingerprints (LeafLayer above)
    const bool leafless = (d.capacity || dense.compact) && s_loc >= leafless_min();   // a compact representation implies the leafless trees (same size condition)
    // the chi table is next needed after the operations' argument (re-allocated there); and what the earlier phases parked in the recycling pool goes back to the driver
    // before the peak (the primary sumcheck's work arrays: no later buffer has their size) — except buffers of a tree's size, which the loop below takes
    // (+ 1 when the chunked leaf rounds' mini-layers, alpha * s_loc / 4 elements, happen to be of a tree's size — alpha = 4 — or they are allocated anew every proof: tests/test_buffer_policy_cpu.py)
    if (d.capacity) { chis.reset(); d.trim_keep((leafless ? s_loc : 2 * s_loc) * sizeof(lasso_fr), 2 * alpha + (leafless && alpha == 4 ? 1 : 0)); }
    LeafLayer leaf; leaf.gamma = g; leaf.tau = ta; leaf.n_loc = s_loc;
    for (size_t i = 0; i < alpha; i++) {
      size_t j = S.memory_to_dimension_index(i); const lasso_fr* table = tables[S.memory_to_subtable_index(i)].p;
      DBuf ti(d, 2 * m_loc), tf(d, 2 * m_loc), tr(d, leafless ? s_loc : 2 * s_loc), tw(d, leafless ? s_loc : 2 * s_loc);
      d.chk(lasso_fingerprint_mem_slab(d.ctx, table, dense.final_(j), m_loc, (uint32_t)P, (uint32_t)d.comm.rank, &g, &ta, ti.p, tf.p), "lasso_fingerprint_mem");
      d.chk(lasso_gp_build(d.ctx, ti.p, m_loc), "lasso_gp_build"); d.chk(lasso_gp_build(d.ctx, tf.p, m_loc), "lasso_gp_build");
      if (leafless && dense.compact) {
        d.chk(lasso_fingerprint_ops_gp_upper_u32(d.ctx, table, dense.dim_u32[j].p, dense.read_u32[j].p, s_loc, &g, &ta, tr.p, tw.p), "lasso_fingerprint_ops_gp_upper_u32");
        leaf.mems.push_back({table, dense.dim_u32[j].p, nullptr, dense.read_u32[j].p});
      } else if (leafless) {
        d.chk(lasso_fingerprint_ops_gp_upper(d.ctx, table, dense.dim_u32[j].p, dense.read(j), s_loc, &g, &ta, tr.p, tw.p), "lasso_fingerprint_ops_gp_upper");
        leaf.mems.push_back({table, dense.dim_u32[j].p, dense.read(j)});
      } else if (s_loc >= 4) {   // read / write leaves and both trees in one call: the first product layer is taken while the leaves are in registers (no re-read of 2 x 32 s bytes)
        d.chk(lasso_fingerprint_ops_gp(d.ctx, table, dense.dim_u32[j].p, dense.read(j), s_loc, &g, &ta, tr.p, tw.p), "lasso_fingerprint_ops_gp");
      } else {
        d.chk(lasso_fingerprint_ops(d.ctx, table, dense.dim_u32[j].p, dense.read(j), s_loc, &g, &ta, tr.p, tw.p), "lasso_fingerprint_ops");
        d.chk(lasso_gp_build(d.ctx, tr.p, s_loc), "lasso_gp_build"); d.chk(lasso_gp_build(d.ctx, tw.p, s_loc), "lasso_gp_build");
      }
      t_init.push_back(std::move(ti)); t_final.push_back(std::move(tf)); t_read.push_back(std::move(tr)); t_write.push_back(std::move(tw));
    }
    if (Trace::mem()) d.dump_live("trees built");
    // ProductLayerProof::prove (memory_checking.rs:674-731)
    sp.reset(), sp.reset(new Trace("ProductLayer.prove", d.ctx));
    t.append_protocol_name("Lasso ProductLayerProof");
    // GrandProductCircuit::evaluate = product of the last layer's two elements; in slab mode that is the rank's LOCAL root, i.e. element `rank` of the
    // global layer of P elements: the roots are all-gathered, the global layers P, P/2, .., 2 are built from them (replicated) and the hash is the top product
    std::vector<DBuf> tops_store;
    auto root_and_top = [&](const DBuf& tree, size_t n_loc, lasso_fr*& top_out, size_t missing = 0) {   // missing: elements the arena lacks in front (a leafless tree: n_loc)
      lasso_fr two[2]; const lasso_fr* last[2] = {tree.p + (2 * n_loc - 4 - missing), tree.p + (2 * n_loc - 3 - missing)};
      d.chk(lasso_read_heads(d.ctx, last, 2, two), "lasso_read_heads");   // through the mapped result buffer: no memcpy, no stream synchronisation
      Sc local = Sc::from_abi(two[0]) * Sc::from_abi(two[1]);
      top_out = nullptr;
      if (P == 1) return local;
      std::vector<lasso_fr> mine{local.abi()}, all(P);
      d.comm.allgather(mine.data(), all.data(), sizeof(lasso_fr));
      tops_store.emplace_back(d, 2 * P);
      d.chk(lasso_upload(d.ctx, tops_store.back().p, all.data(), P * sizeof(lasso_fr)), "lasso_upload");
      if (P > 2) d.chk(lasso_gp_build(d.ctx, tops_store.back().p, P), "lasso_gp_build");
      top_out = tops_store.back().p;
      Sc prod = Sc::one(); for (auto& x : all) prod *= Sc::from_abi(x);
      return prod;
    };
    // Slab mode (round 6): the global layers of P, P/2, .., 2 elements are built from the all-gathered local roots — which arrive ON THE HOST.  Round 5 uploaded them, built the
    // layers on the device and proved them with device rounds (log2 P layers per argument, each a launch or a resident kernel's worth of hand-offs for <= P field products);
    // now the layers are P - 1 host products and bgpa_prove proves them through host_cubic_rounds, as it does a single GPU's tree tops.  LASSO_SLAB_HOST_TOPS=0: round 5's form.
    static const bool slab_host_tops_off = [] { const char* v = getenv("LASSO_SLAB_HOST_TOPS"); return v && v[0] == '0'; }();
    const bool slab_host_tops = P > 1 && !slab_host_tops_off;
    auto root_and_host_top = [&](const DBuf& tree, size_t n_loc, size_t missing, HostTops& T, size_t c, size_t kk) {
      lasso_fr two[2]; const lasso_fr* last[2] = {tree.p + (2 * n_loc - 4 - missing), tree.p + (2 * n_loc - 3 - missing)};
      d.chk(lasso_read_heads(d.ctx, last, 2, two), "lasso_read_heads");
      const Sc local = Sc::from_abi(two[0]) * Sc::from_abi(two[1]);
      std::vector<lasso_fr> mine{local.abi()}, all(P);
      d.comm.allgather(mine.data(), all.data(), sizeof(lasso_fr));
      if (T.run.empty()) { T.len = P; T.run.assign(kk, ScVec()); }
      ScVec& run = T.run[c]; run.clear(); run.reserve(2 * P - 2);
      for (auto& x : all) run.push_back(Sc::from_abi(x));                                  // the layer of P elements: element g = rank g's local root
      for (size_t len = P, off = 0; len > 2; off += len, len /= 2) for (size_t i = 0; i < len / 2; i++) run.push_back(run[off + i] * run[off + i + len / 2]);   // grand_product.rs:25-30's pairs (i, i + len/2)
      return run[run.size() - 2] * run[run.size() - 1];
    };
    ScVec roots_rw, roots_if;
    std::vector<lasso_fr*> rw, inf, rw_top, inf_top;
    // One GPU: the TOP of every tree — its layers of at most 2 * m_stop elements — comes to the host in one read per argument (lasso_read_runs through the mapped buffer:
    // was 4 alpha two-element reads for the roots alone); the roots are the products of the two-element layers, and bgpa_prove proves those layers without the device.
    HostTops tops_rw, tops_if;
    auto read_tops = [&](const std::vector<DBuf>& ta, const std::vector<DBuf>& tb, size_t n_leaves, size_t missing, HostTops& out) {   // trees interleaved (ta[0], tb[0], ta[1], ..): bgpa's circuit order
      const size_t kk = 2 * ta.size(), m0 = host_m_stop(kk);
      if (P != 1 || !host_tail_budget() || kk > LASSO_HOST_TOPS_MAX_PTRS) return;   // LASSO_HOST_TAIL=0: everything through the device, as before round 5
      size_t hl = 2 * m0; if (hl > n_leaves) hl = n_leaves;                     // a tree of fewer leaves is all top
      if (missing && hl >= n_leaves) return;                                       // leafless trees: the leaf layer is not there to be read
      if (hl < 2) return;
      const size_t count = 2 * hl - 2, start = 2 * n_leaves - 2 * hl - missing;   // the layers of hl, hl / 2, .., 2 elements: the last 2 hl - 2 elements of the arena
      std::vector<const lasso_fr*> ptr; for (size_t i = 0; i < ta.size(); i++) { ptr.push_back(ta[i].p + start); ptr.push_back(tb[i].p + start); }
      std::vector<lasso_fr> flat(kk * count);
      d.chk(lasso_read_runs(d.ctx, ptr.data(), (uint32_t)kk, (uint32_t)count, flat.data()), "lasso_read_runs");
      out.len = hl; out.run.assign(kk, ScVec(count));
      for (size_t c = 0; c < kk; c++) for (size_t i = 0; i < count; i++) out.run[c][i] = Sc::from_abi(flat[c * count + i]);
    };
    read_tops(t_read, t_write, s_loc, leafless ? s_loc : 0, tops_rw);
    read_tops(t_init, t_final, m_loc, 0, tops_if);
    auto root_of = [](const HostTops& T, size_t c) { const ScVec& r = T.run[c]; return r[r.size() - 2] * r[r.size() - 1]; };   // GrandProductCircuit::evaluate: the last layer's two elements
    for (size_t i = 0; i < alpha; i++) {
      lasso_fr *ti_top = nullptr, *tr_top = nullptr, *tw_top = nullptr, *tf_top = nullptr;
      const size_t lm = leafless ? s_loc : 0;
      Sc hi, hr, hw, hf;
      if (slab_host_tops) {   // (the four all-gathers in the same order on every rank)
        hi = root_and_host_top(t_init[i], m_loc, 0, tops_if, 2 * i, 2 * alpha); hf = root_and_host_top(t_final[i], m_loc, 0, tops_if, 2 * i + 1, 2 * alpha);
        hr = root_and_host_top(t_read[i], s_loc, lm, tops_rw, 2 * i, 2 * alpha); hw = root_and_host_top(t_write[i], s_loc, lm, tops_rw, 2 * i + 1, 2 * alpha);
      } else {
      if (tops_if.len) { hi = root_of(tops_if, 2 * i); hf = root_of(tops_if, 2 * i + 1); } else { hi = root_and_top(t_init[i], m_loc, ti_top); hf = root_and_top(t_final[i], m_loc, tf_top); }
      if (tops_rw.len) { hr = root_of(tops_rw, 2 * i); hw = root_of(tops_rw, 2 * i + 1); } else { hr = root_and_top(t_read[i], s_loc, tr_top, lm); hw = root_and_top(t_write[i], s_loc, tw_top, lm); }
      }
      if (!(hi * hw == hr * hf)) throw Error("memory checking: hash_init * hash_write != hash_read * hash_final (memory_checking.rs:689)");
      t.append_scalar("claim_hash_init", hi); t.append_scalar("claim_hash_read", hr); t.append_scalar("claim_hash_write", hw); t.append_scalar("claim_hash_final", hf);
      W.sc(hi); W.sc(hr); W.sc(hw); W.sc(hf);
      roots_rw.push_back(hr); roots_rw.push_back(hw); roots_if.push_back(hi); roots_if.push_back(hf);
      rw.push_back(t_read[i].p); rw.push_back(t_write[i].p); inf.push_back(t_init[i].p); inf.push_back(t_final[i].p);
      rw_top.push_back(tr_top); rw_top.push_back(tw_top); inf_top.push_back(ti_top); inf_top.push_back(tf_top);
    }
    ScVec rand_ops, rand_mem;
    BatchedGrandProductArgument proof_ops = bgpa_prove(rw, rw_top, s, roots_rw, rand_ops, leafless ? &leaf : nullptr, tops_rw.len ? &tops_rw : nullptr);
    t_read.clear(); t_write.clear();
    if (!chis.p) chis = DBuf(d, s_loc);
    // Everything HashLayerProof needs at rand_ops that does not depend on the transcript — the evaluations of E / dim / read (one pass over all of
    // them) and the big mat-vecs of the two openings at rand_ops — starts now on the side context and runs under the second grand-product
    // argument, which is latency-bound (two 2^16-leaf circuits per memory) and leaves the device mostly idle.
    const size_t C = S.C();
    const size_t k_derefs = ceil_log2(next_pow2(alpha)), k_ops = ceil_log2(next_pow2(2 * C)), k_mem = ceil_log2(next_pow2(C));
    std::vector<const lasso_fr*> at_ops(Eptr);
    if (!dense.compact) { for (size_t i = 0; i < C; i++) at_ops.push_back(dense.dim(i)); for (size_t i = 0; i < C; i++) at_ops.push_back(dense.read(i)); }
    OpenPrep prep_derefs, prep_ops, prep_mem;
    const bool side = P == 1 && !side_off() && !dense.compact;   // compact form: dim / read exist as field elements one at a time, on the main context
    if (side) {
      lasso_ctx* sc = d.side();
      std::vector<lasso_fr> rr; for (auto& x : rand_ops) rr.push_back(x.abi());
      d.chk_side(lasso_eq_evals(sc, rr.data(), (uint32_t)rr.size(), chis.p), "lasso_eq_evals");
      d.chk_side(lasso_defer_next(sc), "lasso_defer_next");
      std::vector<lasso_fr> dummy(at_ops.size());
      d.chk_side(lasso_multi_dot(sc, at_ops.data(), (uint32_t)at_ops.size(), chis.p, s_loc, dummy.data()), "lasso_multi_dot");
      prep_derefs = prep_open(combined_E.p, k_derefs, rand_ops);
      prep_ops = prep_open(dense.combined_l_variate_polys.p, k_ops, rand_ops);
    }
    BatchedGrandProductArgument proof_mem = bgpa_prove(inf, inf_top, m, roots_if, rand_mem, nullptr, tops_if.len ? &tops_if : nullptr);
    t_init.clear(); t_final.clear(); tops_store.clear();
    proof_mem.write(W); proof_ops.write(W);    // field order of ProductLayerProof: grand_product_evals, proof_mem, proof_ops (:656-660)
    // HashLayerProof::prove (memory_checking.rs:338-460)
    sp.reset(), sp.reset(new Trace("HashLayer.prove", d.ctx));
    t.append_protocol_name("Lasso HashLayerProof");
    ScVec ev_ops;
    DBuf chim(d, m_loc);
    std::vector<const lasso_fr*> fin; for (size_t i = 0; i < C; i++) fin.push_back(dense.final_(i));
    {
      std::vector<lasso_fr> out(at_ops.size());
      if (side) {
        d.chk_side(lasso_result_wait(d.side(), out.data(), out.size()), "lasso_result_wait");
        // the same for rand_mem (known now): the evaluations of final and the mat-vec of the last opening run under the first opening's bullet rounds
        std::vector<lasso_fr> rr; for (auto& x : rand_mem) rr.push_back(x.abi());
        d.chk_side(lasso_eq_evals(d.side(), rr.data(), (uint32_t)rr.size(), chim.p), "lasso_eq_evals");
        d.chk_side(lasso_defer_next(d.side()), "lasso_defer_next");
        std::vector<lasso_fr> dummy(C);
        d.chk_side(lasso_multi_dot(d.side(), fin.data(), (uint32_t)C, chim.p, m_loc, dummy.data()), "lasso_multi_dot");
        prep_mem = prep_open(dense.combined_log_m_variate_polys.p, k_mem, rand_mem);
      } else {
        eq_evals_local(rand_ops, chis.p);
        d.chk(lasso_multi_dot(d.ctx, at_ops.data(), (uint32_t)at_ops.size(), chis.p, s_loc, out.data()), "lasso_multi_dot");
        if (dense.compact) {   // dim_i, read_i: lifted into one scratch array and evaluated one after the other
          DBuf tmp(d, s_loc); const lasso_fr* one[1] = {tmp.p};
          out.resize(alpha + 2 * C);
          for (size_t b = 0; b < 2 * C; b++) { dense.lift_block(b, tmp.p); d.chk(lasso_multi_dot(d.ctx, one, 1, chis.p, s_loc, &out[alpha + b]), "lasso_multi_dot"); }
        }
        d.comm.sum(out);
      }
      for (auto& o : out) ev_ops.push_back(Sc::from_abi(o));
    }
    ScVec eval_derefs(ev_ops.begin(), ev_ops.begin() + alpha), eval_dim(ev_ops.begin() + alpha, ev_ops.begin() + alpha + C), eval_read(ev_ops.begin() + alpha + C, ev_ops.end());
    t.append_protocol_name("Lasso CombinedTableEvalProof");
    DotProductProofLog proof_derefs = joint_open("evals_ops_val", "challenge_combine_n_to_one", "joint_claim_eval", eval_derefs, true, combined_E.p, nv_derefs, rand_ops, gens.gens_derefs, &prep_derefs);
    ScVec eval_final;
    {
      std::vector<lasso_fr> out(C);
      if (side) d.chk_side(lasso_result_wait(d.side(), out.data(), C), "lasso_result_wait");
      else {
        eq_evals_local(rand_mem, chim.p);
        d.chk(lasso_multi_dot(d.ctx, fin.data(), (uint32_t)C, chim.p, m_loc, out.data()), "lasso_multi_dot");
        d.comm.sum(out);
      }
      for (auto& o : out) eval_final.push_back(Sc::from_abi(o));
    }
    ScVec evals_ops = eval_dim; evals_ops.insert(evals_ops.end(), eval_read.begin(), eval_read.end());
    DotProductProofLog proof_ops_open = joint_open("claim_evals_ops", "challenge_combine_n_to_one", "joint_claim_eval_ops", evals_ops, true, dense.compact ? nullptr : dense.combined_l_variate_polys.p, dense.nv_l, rand_ops, gens.gens_combined_l_variate, &prep_ops);
    DotProductProofLog proof_mem_open = joint_open("claim_evals_mem", "challenge_combine_two_to_one", "joint_claim_eval_mem", eval_final, false, dense.combined_log_m_variate_polys.p, dense.nv_m, rand_mem, gens.gens_combined_log_m_variate, &prep_mem);
    if (side) side_sync();
    W.sc_arr(eval_dim); W.sc_arr(eval_read); W.sc_arr(eval_final); W.sc_arr(eval_derefs);     // HashLayerProof field order (:314-329)
    proof_ops_open.write(W); proof_mem_open.write(W); proof_derefs.write(W);
  }
};

}  // namespace lasso

// liblasso_prover.so — include/lasso_prover.h over lasso_amd/host/prover.hpp.  Nothing unwinds across the ABI.
#include "prover.hpp"
#include "verifier.hpp"
#include "shm_comm.hpp"
#include <atomic>
#include "../../include/lasso_prover.h"

using namespace lasso;

// Generator and dense-representation objects hold device buffers that belong to their host's context (DBuf keeps a `const Dev*`), so the host
// must outlive them: it is reference-counted by its children, and lasso_host_destroy only drops the caller's reference — the context goes away
// with the last child (Python's GC may release the three in any order).
struct lasso_host {
  Dev dev; std::atomic<int> refs{1};
  std::unique_ptr<ShmComm> shm;   // slab mode's native intra-node exchange (lasso_host_set_comm_shm); outlives every proof of this host
  explicit lasso_host(int device) : dev(device) {}
  void retain() { refs.fetch_add(1, std::memory_order_relaxed); }
  void release() { if (refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete this; }
};
struct lasso_host_gens {
  lasso_host* owner; std::unique_ptr<SparsePolyCommitmentGens> g;
  lasso_host_gens(lasso_host* h, const char* label, size_t c, size_t s, size_t nm, size_t log_m) : owner(h) { g.reset(new SparsePolyCommitmentGens(h->dev, label, c, s, nm, log_m)); h->retain(); }
  lasso_host_gens(lasso_host* h, size_t c, size_t s, size_t nm, size_t log_m, const lasso_affine* p1, size_t n1, const lasso_affine* p2, size_t n2, const lasso_affine* p3, size_t n3) : owner(h) {
    g.reset(new SparsePolyCommitmentGens(h->dev, c, s, nm, log_m, p1, n1, p2, n2, p3, n3)); h->retain();
  }
  ~lasso_host_gens() { g.reset(); owner->release(); }
};
struct lasso_host_dense {
  lasso_host* owner; std::unique_ptr<DensifiedRepresentation> d;
  explicit lasso_host_dense(lasso_host* h) : owner(h) { h->retain(); }
  ~lasso_host_dense() { d.reset(); owner->release(); }
};

static thread_local std::string g_err;
#define GUARD(body) try { body } catch (const std::exception& e) { g_err = e.what(); return -1; } catch (...) { g_err = "unknown error"; return -1; }
static int32_t emit(const std::vector<uint8_t>& b, uint8_t* out, size_t cap, size_t* len) { if (len) *len = b.size(); if (!out || b.size() > cap) { g_err = "output buffer too small"; return -2; } memcpy(out, b.data(), b.size()); return 0; }

extern "C" {
const char* lasso_host_last_error(void) { return g_err.c_str(); }
// The library may have been built with -march=x86-64-v3 (lasso_amd/build.py host_march_flags): refuse, with a message, to run on a CPU without those extensions — the alternative
// is a SIGILL somewhere inside the first Keccak permutation.  This function itself must not need them: it is compiled for the baseline ISA.
#if defined(LASSO_HOST_V3) && defined(__x86_64__)
__attribute__((target("arch=x86-64"))) static bool host_cpu_ok() { __builtin_cpu_init(); return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("fma"); }
#else
static bool host_cpu_ok() { return true; }
#endif
int32_t lasso_host_create(int32_t device, lasso_host** out) {
  if (!host_cpu_ok()) { g_err = "liblasso_prover was built with -march=x86-64-v3 (AVX2 / BMI2 / FMA) and this CPU lacks them: rebuild with LASSO_HOST_MARCH=none"; return -1; }
  GUARD(*out = new lasso_host(device); return 0;)
}
void lasso_host_destroy(lasso_host* h) { if (h) h->release(); }
lasso_ctx* lasso_host_ctx(lasso_host* h) { return h ? h->dev.ctx : nullptr; }
int32_t lasso_host_mem_stats(lasso_host* h, uint64_t* live_bytes, uint64_t* peak_bytes, uint64_t* prover_peak_bytes, int32_t reset) {
  GUARD(if (!h) throw Error("lasso_host_mem_stats: null host"); h->dev.mem_stats(live_bytes, peak_bytes, prover_peak_bytes, reset != 0); return 0;)
}
int32_t lasso_host_set_capacity(lasso_host* h, int32_t on) { GUARD(if (!h) throw Error("lasso_host_set_capacity: null host"); h->dev.capacity = on != 0; if (on) h->dev.trim(); return 0;) }
int32_t lasso_host_set_throughput_mode(lasso_host* h, int32_t on) { GUARD(if (!h) throw Error("lasso_host_set_throughput_mode: null host"); h->dev.throughput = on != 0; return 0;) }
int32_t lasso_host_set_comm(lasso_host* h, int32_t rank, int32_t world, lasso_host_allgather_fn fn, void* user) {
  GUARD(
    if (!h || world < 1 || (world & (world - 1)) || rank < 0 || rank >= world || (world > 1 && !fn)) throw Error("lasso_host_set_comm: world must be a power of two, 0 <= rank < world, and a collective is needed when world > 1");
    h->dev.comm.rank = (size_t)rank; h->dev.comm.world = (size_t)world; h->dev.comm.fn = fn; h->dev.comm.user = user; return 0;)
}
// Slab mode with the library's own exchange: the ranks of one node meet in a POSIX shared-memory segment `name` (same on every rank, e.g. derived from
// MASTER_PORT) and all-gather their per-round partial sums there (shm_comm.hpp) — no callback into the embedding language per round.
int32_t lasso_host_set_comm_shm(lasso_host* h, int32_t rank, int32_t world, const char* name) {
  GUARD(
    if (!h || !name || world < 1 || (world & (world - 1)) || rank < 0 || rank >= world) throw Error("lasso_host_set_comm_shm: world must be a power of two and 0 <= rank < world");
    h->shm.reset(new ShmComm(name, rank, world));
    h->dev.comm.rank = (size_t)rank; h->dev.comm.world = (size_t)world; h->dev.comm.fn = &ShmComm::trampoline; h->dev.comm.user = h->shm.get();
    // The bulk exchange (partial row commitments) goes over RCCL on the device stream when every rank can join one communicator: rank 0 draws the
    // unique id and publishes it through the segment; the ranks then agree (an all-gather of one status byte) — all use RCCL or none does.
    // Agreement protocol (every step an all-gather through the segment, so no rank can be left inside a collective its peers never enter):
    //   1. every rank publishes whether IT can use RCCL at all (librccl loads, LASSO_SLAB_RCCL != 0) — ncclCommInitRank is only entered if ALL can;
    //   2. rank 0 draws the unique id and publishes it with its status — the others read it, and all enter ncclCommInitRank or none does;
    //   3. every rank publishes its ncclCommInitRank outcome — the communicator is kept only if it came up on every rank.
    const char* e = getenv("LASSO_SLAB_RCCL");
    if (world > 1) {
      auto all_ok = [&](uint8_t mine, const char* what) {
        std::vector<uint8_t> all((size_t)world, 0);
        if (h->shm->allgather(&mine, all.data(), 1) != 0) throw Error(std::string("lasso_host_set_comm_shm: ranks did not agree on ") + what);
        bool every = true; for (uint8_t v : all) every = every && v; return every;
      };
      const uint8_t can = (!(e && e[0] == '0') && lasso_rccl_available() == 1) ? 1 : 0;
      if (all_ok(can, "RCCL availability")) {
        uint8_t blob[129] = {0};
        if (rank == 0) blob[128] = lasso_rccl_unique_id(blob) == 0 ? 1 : 0;
        h->shm->broadcast_blob(blob, sizeof(blob));
        if (blob[128]) {     // the same byte on every rank: all enter the collective init, or none
          const uint8_t mine = lasso_rccl_init(h->dev.ctx, rank, world, blob) == 0 ? 1 : 0;
          if (!all_ok(mine, "the device-side exchange")) { if (mine) (void)lasso_rccl_shutdown(h->dev.ctx); }
          else {
            //   4. first contact: every rank all-gathers 1 KB through the new communicator and checks every slot (lasso_rccl_selftest, bounded wait); the communicator is kept only
            //      if that worked on EVERY rank — otherwise the fall-back is announced on stderr, not silent (a node where xGMI / RCCL is half-configured must say so once)
            const uint8_t st = lasso_rccl_selftest(h->dev.ctx) == 0 ? 1 : 0;
            if (!st) fprintf(stderr, "[lasso] rank %d: RCCL self-test failed (%s) — slab mode falls back to the shared-memory exchange of partial row commitments\n", rank, lasso_last_error(h->dev.ctx));
            if (!all_ok(st, "the RCCL self-test")) { if (lasso_rccl_ready(h->dev.ctx)) (void)lasso_rccl_shutdown(h->dev.ctx); if (st) fprintf(stderr, "[lasso] rank %d: a peer's RCCL self-test failed — shared-memory exchange on every rank\n", rank); }
          }
        }
      }
    }
    return 0;)
}
int32_t lasso_host_gens_new(lasso_host* h, const char* label, size_t c, size_t s, size_t nm, size_t log_m, lasso_host_gens** out) { GUARD(*out = new lasso_host_gens(h, label, c, s, nm, log_m); return 0;) }
int32_t lasso_host_gens_from_points(lasso_host* h, size_t c, size_t s, size_t nm, size_t log_m, const lasso_affine* l_variate, size_t n_l, const lasso_affine* log_m_variate, size_t n_m,
                                    const lasso_affine* derefs, size_t n_d, lasso_host_gens** out) {
  GUARD(if (!h || !out) throw Error("lasso_host_gens_from_points: null argument"); *out = new lasso_host_gens(h, c, s, nm, log_m, l_variate, n_l, log_m_variate, n_m, derefs, n_d); return 0;)
}
int32_t lasso_host_gens_points(lasso_host_gens* g, int32_t which, lasso_affine* out, size_t cap, size_t* count) {
  GUARD(
    if (!g || !count) throw Error("lasso_host_gens_points: null argument");
    const PolyCommitmentGens& pg = g->g->set(which);
    *count = pg.affine.size();
    if (!out || cap < pg.affine.size()) { g_err = "output buffer too small"; return -2; }
    memcpy(out, pg.affine.data(), pg.affine.size() * sizeof(lasso_affine)); return 0;)
}
// every table a proof over these generators reads, built now: the byte-multiple tables of the small-scalar commitments (dim / read / final: two byte windows; E: one) are otherwise
// built by the first lasso_host_commit / lasso_host_prove that meets the object — inside whatever span the caller times
int32_t lasso_host_gens_prepare(lasso_host_gens* g) {
  GUARD(
    if (!g) throw Error("lasso_host_gens_prepare: null argument");
    for (int which = 0; which < 3; which++) {
      const PolyCommitmentGens& pg = g->g->set(which);
      if (pg.bases) pg.dev->chk(lasso_bases_prepare(pg.dev->ctx, pg.bases, 2), "lasso_bases_prepare");
      if (pg.bases_slab) pg.dev->chk(lasso_bases_prepare(pg.dev->ctx, pg.bases_slab, 2), "lasso_bases_prepare");
    }
    return 0;)
}
void lasso_host_gens_free(lasso_host_gens* g) { delete g; }
int32_t lasso_host_densify(lasso_host* h, const uint64_t* indices, size_t n, size_t c, size_t log_m, lasso_host_dense** out) {
  GUARD(std::unique_ptr<lasso_host_dense> d(new lasso_host_dense(h)); d->d = DensifiedRepresentation::from_lookup_indices(h->dev, indices, n, c, log_m); *out = d.release(); return 0;)
}
void lasso_host_dense_free(lasso_host_dense* d) { delete d; }
int32_t lasso_host_dense_info(lasso_host_dense* d, uint64_t* device_bytes, int32_t* compact) {
  if (!d || !d->d) return LASSO_ERR_INVALID;
  const DensifiedRepresentation& D = *d->d;
  uint64_t b = (uint64_t)(D.combined_l_variate_polys.n + D.combined_log_m_variate_polys.n) * sizeof(lasso_fr);
  for (auto& x : D.dim_u32) b += 4 * (uint64_t)x.n;
  for (auto& x : D.read_u32) b += 4 * (uint64_t)x.n;
  if (device_bytes) *device_bytes = b;
  if (compact) *compact = D.compact ? 1 : 0;
  return 0;
}
int32_t lasso_host_commit(lasso_host_dense* d, lasso_host_gens* g, uint8_t* out, size_t cap, size_t* len) {
  GUARD(SparsePolynomialCommitment c = d->d->commit(*g->g); ProofWriter w; w.pts_vec(c.l_variate_polys_commitment.compressed); w.pts_vec(c.log_m_variate_polys_commitment.compressed); return emit(w.b, out, cap, len);)
}
// one proof (capacity mode's bound on what stays parked between proofs is in Dev::alloc_bytes: returning everything to the driver here was measured at 1.4 s per proof
// of hipFree / hipMalloc at configs[3])
static std::vector<uint8_t> run_prover(lasso_host* h, const Strategy& S, DensifiedRepresentation& D, const SparsePolyCommitmentGens& G, ProofTranscript& t, RandomTape& tape, const ScVec& rv) {
  std::vector<uint8_t> bytes;
  {
    Prover P(h->dev, S, D, G, t, tape);
    try { P.prove(rv); } catch (...) { h->dev.abort_all(); throw; }
    bytes.swap(P.proof_bytes);
  }
  return bytes;
}
int32_t lasso_host_prove(lasso_host* h, lasso_host_dense* d, lasso_host_gens* g, const lasso_strategy* st, const lasso_fr* r, size_t r_len, const char* tl, const char* pl,
                         uint8_t* out, size_t cap, size_t* len) {
  GUARD(
    Strategy S(st->kind, st->c, st->log_m, st->log_r);
    ProofTranscript t(tl); RandomTape tape(pl);
    ScVec rv; for (size_t i = 0; i < r_len; i++) rv.push_back(Sc::from_abi(r[i]));
    std::vector<uint8_t> bytes = run_prover(h, S, *d->d, *g->g, t, tape, rv);
    return emit(bytes, out, cap, len);)
}
// ---- live transcripts (include/lasso_prover.h lasso_transcript_vtbl)
struct lasso_merlin { Merlin m; explicit lasso_merlin(const char* label) : m(label) {} };
static void merlin_append(void* u, const uint8_t* label, size_t ll, const uint8_t* msg, size_t n) { ((lasso_merlin*)u)->m.append_message_l(label, ll, msg, n); }
static void merlin_challenge(void* u, const uint8_t* label, size_t ll, uint8_t* dest, size_t n) { ((lasso_merlin*)u)->m.challenge_bytes_l(label, ll, dest, n); }
static const lasso_transcript_vtbl g_merlin_vtbl = {merlin_append, merlin_challenge};
const lasso_transcript_vtbl* lasso_host_merlin_vtbl(void) { return &g_merlin_vtbl; }
lasso_merlin* lasso_host_merlin_new(const char* label) { try { return label ? new lasso_merlin(label) : nullptr; } catch (...) { return nullptr; } }
lasso_merlin* lasso_host_random_tape_new(const char* name) {
  try {
    if (!name) return nullptr;
    lasso_merlin* t = new lasso_merlin(name);
    ChaChaRng prng = ChaChaRng::test_rng(); uint8_t b[32]; fr_rand(prng).to_bytes(b);
    t->m.append_message("init_randomness", b, 32);   // RandomTape::new (utils/random.rs:15-31)
    return t;
  } catch (...) { return nullptr; }
}
void lasso_host_merlin_free(lasso_merlin* m) { delete m; }
int32_t lasso_host_prove_cb(lasso_host* h, lasso_host_dense* d, lasso_host_gens* g, const lasso_strategy* st, const lasso_fr* r, size_t r_len,
                            const lasso_transcript_vtbl* tv, void* tu, const lasso_transcript_vtbl* pv, void* pu, uint8_t* out, size_t cap, size_t* len) {
  GUARD(
    if (!h || !d || !g || !st || !r || !tv || !pv) throw Error("lasso_host_prove_cb: null argument");
    Strategy S(st->kind, st->c, st->log_m, st->log_r);
    ProofTranscript t(tv, tu); RandomTape tape(pv, pu);
    ScVec rv; for (size_t i = 0; i < r_len; i++) rv.push_back(Sc::from_abi(r[i]));
    std::vector<uint8_t> bytes = run_prover(h, S, *d->d, *g->g, t, tape, rv);
    return emit(bytes, out, cap, len);)
}
int32_t lasso_host_verify_cb(lasso_host* h, lasso_host_gens* g, const lasso_strategy* st, size_t s, const lasso_fr* r, size_t r_len, const lasso_transcript_vtbl* tv, void* tu,
                             const uint8_t* proof, size_t proof_len, const uint8_t* commitment, size_t commitment_len, int32_t* ok) {
  GUARD(
    if (!h || !g || !st || !r || !proof || !commitment || !ok || !tv) throw Error("lasso_host_verify_cb: null argument");
    Strategy S(st->kind, st->c, st->log_m, st->log_r);
    ProofTranscript t(tv, tu);
    ScVec rv; for (size_t i = 0; i < r_len; i++) rv.push_back(Sc::from_abi(r[i]));
    Verifier V(h->dev, S, *g->g, t);
    *ok = V.verify(proof, proof_len, commitment, commitment_len, s, st->log_m, rv) ? 1 : 0;
    return 0;)
}
int32_t lasso_host_verify(lasso_host* h, lasso_host_gens* g, const lasso_strategy* st, size_t s, const lasso_fr* r, size_t r_len, const char* tl,
                          const uint8_t* proof, size_t proof_len, const uint8_t* commitment, size_t commitment_len, int32_t* ok) {
  GUARD(
    if (!h || !g || !st || !r || !proof || !commitment || !ok) throw Error("lasso_host_verify: null argument");
    Strategy S(st->kind, st->c, st->log_m, st->log_r);
    ProofTranscript t(tl);
    ScVec rv; for (size_t i = 0; i < r_len; i++) rv.push_back(Sc::from_abi(r[i]));
    Verifier V(h->dev, S, *g->g, t);
    *ok = V.verify(proof, proof_len, commitment, commitment_len, s, st->log_m, rv) ? 1 : 0;
    return 0;)
}
// Test support: Prover::prove_cubic_batched (sumcheck.rs:27-135 with C = EqPolynomial(rand).evals(), grand_product.rs:122-128) on caller-supplied
// arrays with a SCRIPTED eq point, so that the eq points a transcript never produces (rand_t = 0: the claim-derived evaluation is unavailable;
// rand_t = 1: the table prefix vanishes) can be driven through the device path.  A, B: k arrays of 2^ell elements each, contiguous.
// out = per round the three compressed coefficients, then the ell challenges, then the k final claims of A and the k of B (32-byte scalars).
int32_t lasso_host_debug_cubic_batched(lasso_host* h, lasso_host_dense* dn, lasso_host_gens* g, const lasso_strategy* st, size_t k, size_t ell, const lasso_fr* A, const lasso_fr* B,
                                       const lasso_fr* rand, const lasso_fr* coeffs, const lasso_fr* claim, const char* tl, uint8_t* out, size_t cap, size_t* len) {
  try {
    Strategy S(st->kind, st->c, st->log_m, st->log_r);
    ProofTranscript t(tl); RandomTape tape("unused");
    Prover P(h->dev, S, *dn->d, *g->g, t, tape);
    const size_t n = (size_t)1 << ell;
    std::vector<DBuf> bufs; std::vector<lasso_fr*> pa, pb;
    for (size_t c = 0; c < 2 * k; c++) {
      bufs.emplace_back(h->dev, n);
      h->dev.chk(lasso_upload(h->dev.ctx, bufs.back().p, (c < k ? A + c * n : B + (c - k) * n), n * sizeof(lasso_fr)), "lasso_upload");
      (c < k ? pa : pb).push_back(bufs.back().p);
    }
    ScVec rv, cv; for (size_t i = 0; i < ell; i++) rv.push_back(Sc::from_abi(rand[i])); for (size_t i = 0; i < k; i++) cv.push_back(Sc::from_abi(coeffs[i]));
    { const char* e = getenv("LASSO_DEBUG_CUBIC_HOST");   // read per call: the tests drive the HOST rounds (Prover::host_cubic_rounds: the tree tops' layers) through the same scripted points
      if (e && e[0] == '1') {
        std::vector<ScVec> ha(k, ScVec(n)), hb(k, ScVec(n));
        for (size_t c = 0; c < k; c++) for (size_t i = 0; i < n; i++) { ha[c][i] = Sc::from_abi(A[c * n + i]); hb[c][i] = Sc::from_abi(B[c * n + i]); }
        SumcheckProof sp; ScVec r_out; std::vector<lasso_fr> heads; Sc e0 = Sc::from_abi(*claim);
        P.host_cubic_rounds(ha, hb, ell, rv, 0, cv, Sc::one(), e0, sp, r_out, heads);
        ProofWriter w;
        for (auto& c : sp.compressed_polys) w.sc_arr(c);
        w.sc_arr(r_out); for (auto& x : heads) w.sc(Sc::from_abi(x));
        return emit(w.b, out, cap, len);
      } }
    DBuf eq(h->dev, n / 2 ? n / 2 : 1);
    P.eq_half_local(rv, eq.p);
    ScVec r_out, ca, cb;
    SumcheckProof sp = P.prove_cubic_batched(Sc::from_abi(*claim), ell, false, pa, pb, eq.p, rv, cv, r_out, ca, cb);
    ProofWriter w;
    for (auto& c : sp.compressed_polys) w.sc_arr(c);
    w.sc_arr(r_out); w.sc_arr(ca); w.sc_arr(cb);
    return emit(w.b, out, cap, len);
  } catch (const std::exception& e) { g_err = e.what(); return -1; } catch (...) { g_err = "unknown error"; return -1; }
}
void lasso_host_gen_indices(size_t sparsity, size_t memory_size, uint64_t* out) { ChaChaRng rng = ChaChaRng::test_rng(); for (size_t i = 0; i < sparsity; i++) out[i] = rng.next_u64() % memory_size; }
void lasso_host_gen_random_point(size_t bits, lasso_fr* out) { ChaChaRng rng = ChaChaRng::test_rng(); for (size_t i = 0; i < bits; i++) out[i] = fr_rand(rng).abi(); }
}

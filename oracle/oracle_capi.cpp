// ORACLE — TEST INFRASTRUCTURE ONLY (see lasso_oracle.hpp header).
// C-ABI surface of the CPU oracle for ctypes-driven tests, smoke() and bench.py's cpu_baseline leg.
#include "lasso_oracle.hpp"
#include <chrono>
#include <memory>
#include <map>

using namespace orc;

static thread_local std::string g_err;
#define GUARD(body) try { body } catch (const std::exception& e) { g_err = e.what(); return -1; }

static Strategy mk_strategy(int kind, size_t C, size_t M, size_t log_r) { Strategy s; s.kind = (StrategyKind)kind; s.C = C; s.M = M; s.LOG_R = log_r; return s; }

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// ---- field (Montgomery limbs in/out, ark-ff in-memory form); which: 0 = Fr, 1 = Fq
void orc_f_mul(int which, const u64* a, const u64* b, u64* o) { if (which) { Fq r = Fq::from_raw(a) * Fq::from_raw(b); memcpy(o, r.v, 32); } else { Fr r = Fr::from_raw(a) * Fr::from_raw(b); memcpy(o, r.v, 32); } }
void orc_f_add(int which, const u64* a, const u64* b, u64* o) { if (which) { Fq r = Fq::from_raw(a) + Fq::from_raw(b); memcpy(o, r.v, 32); } else { Fr r = Fr::from_raw(a) + Fr::from_raw(b); memcpy(o, r.v, 32); } }
void orc_f_sub(int which, const u64* a, const u64* b, u64* o) { if (which) { Fq r = Fq::from_raw(a) - Fq::from_raw(b); memcpy(o, r.v, 32); } else { Fr r = Fr::from_raw(a) - Fr::from_raw(b); memcpy(o, r.v, 32); } }
void orc_f_inv(int which, const u64* a, u64* o) { if (which) { Fq r = Fq::from_raw(a).inverse(); memcpy(o, r.v, 32); } else { Fr r = Fr::from_raw(a).inverse(); memcpy(o, r.v, 32); } }
void orc_f_from_canonical(int which, const u64* a, u64* o) { if (which) { Fq r = Fq::from_canonical(a); memcpy(o, r.v, 32); } else { Fr r = Fr::from_canonical(a); memcpy(o, r.v, 32); } }
void orc_f_to_canonical(int which, const u64* a, u64* o) { if (which) Fq::from_raw(a).to_canonical(o); else Fr::from_raw(a).to_canonical(o); }
void orc_fr_from_le_bytes_mod_order(const uint8_t* b, size_t n, u64* o) { Fr r = Fr::from_le_bytes_mod_order(b, n); memcpy(o, r.v, 32); }

// which group this build restates: 0 = curve25519 (the reference's own instantiation), 1 = BN254 G1 (-DORC_BN254)
int orc_curve_id() {
#ifdef ORC_BN254
  return 1;
#else
  return 0;
#endif
}
// ---- curve: affine canonical coordinates in/out (4 limbs each), for Python big-int cross checks
static Point pt_from_canon(const u64* xy) { return Point::from_affine(Fq::from_canonical(xy), Fq::from_canonical(xy + 4)); }
static void pt_to_canon(const Point& p, u64* xy) { Fq x, y; p.to_affine(x, y); x.to_canonical(xy); y.to_canonical(xy + 4); }
void orc_pt_generator(u64* xy) { pt_to_canon(Point::generator(), xy); }
void orc_pt_add(const u64* a, const u64* b, u64* o) { pt_to_canon(pt_from_canon(a) + pt_from_canon(b), o); }
void orc_pt_dbl(const u64* a, u64* o) { pt_to_canon(pt_from_canon(a).dbl(), o); }
void orc_pt_mul(const u64* a, const u64* scalar_canon, u64* o) { pt_to_canon(pt_from_canon(a).mul_limbs(scalar_canon), o); }
void orc_pt_compress(const u64* a, uint8_t* out32) { pt_from_canon(a).compress(out32); }
int orc_pt_decompress(const uint8_t* in32, u64* o) { Point p; if (!curve_decompress(in32, p)) return -1; pt_to_canon(p, o); return 0; }
// MSM over affine canonical bases and Montgomery Fr scalars (oracle of msm/mod.rs:36-40)
void orc_msm(const u64* bases_xy, const u64* scalars_mont, size_t n, u64* out_xy) {
  std::vector<Point> b; std::vector<Fr> s;
  for (size_t i = 0; i < n; i++) { b.push_back(pt_from_canon(bases_xy + 8 * i)); s.push_back(Fr::from_raw(scalars_mont + 4 * i)); }
  pt_to_canon(msm(b, s), out_xy);
}

// ---- hashes
void orc_shake256(const uint8_t* in, size_t n, uint8_t* out, size_t m) { Shake256 s; s.absorb(in, n); s.squeeze(out, m); }
void orc_keccak_f1600(uint64_t* st) { keccak_f1600(st); }
void orc_chacha_block(const uint8_t* key32, uint64_t counter, int rounds, uint32_t* out16) { ChaChaRng r(key32, rounds); ChaChaRng::block(r.key, counter, rounds, out16); }
// Merlin: transcript(label); append(msg_label, msg); challenge(ch_label) -> n bytes
void orc_merlin_simple(const char* label, const char* msg_label, const uint8_t* msg, size_t msg_n, const char* ch_label, uint8_t* out, size_t n) {
  Transcript t(label); t.append_message(msg_label, msg, msg_n); t.challenge_bytes(ch_label, out, n);
}
// test_rng stream: first n u64 draws
void orc_test_rng_u64(uint64_t* out, size_t n) { ChaChaRng r = test_rng(); for (size_t i = 0; i < n; i++) out[i] = r.next_u64(); }

// ---- generators (poly/commitments.rs:22-44): n+1 points, affine canonical; last is h
int orc_gens(const char* label, size_t n, u64* out_xy) { GUARD(
  MultiCommitGens g = MultiCommitGens::create(n, label);
  for (size_t i = 0; i < n; i++) pt_to_canon(g.G[i], out_xy + 8 * i);
  pt_to_canon(g.h, out_xy + 8 * n); return 0; ) }

// ---- harness inputs (benches/bench.rs:13-34)
void orc_gen_indices(size_t sparsity, size_t memory_size, uint64_t* out) { auto v = gen_indices(1, sparsity, memory_size); for (size_t i = 0; i < sparsity; i++) out[i] = v[i][0]; }
void orc_gen_random_point(size_t bits, u64* out_mont) { auto r = gen_random_point(bits); for (size_t i = 0; i < bits; i++) memcpy(out_mont + 4 * i, r[i].v, 32); }
void orc_random_tape_init_scalar(u64* out_mont) { ChaChaRng p = test_rng(); Fr f = fr_rand(p); memcpy(out_mont, f.v, 32); }

// ---- strategy helpers
int orc_strategy_info(int kind, size_t C, size_t M, size_t log_r, size_t* num_subtables, size_t* num_memories, size_t* degree) { GUARD(
  Strategy S = mk_strategy(kind, C, M, log_r); *num_subtables = S.num_subtables(); *num_memories = S.num_memories(); *degree = S.g_poly_degree(); return 0; ) }
// materialised subtable k as canonical u64 (all tables hold small integers)
int orc_subtable(int kind, size_t C, size_t M, size_t log_r, size_t k, uint64_t* out) { GUARD(
  Strategy S = mk_strategy(kind, C, M, log_r); auto t = S.materialize_subtables(); ORC_ASSERT(k < t.size());
  for (size_t i = 0; i < M; i++) { u64 c[4]; t[k][i].to_canonical(c); ORC_ASSERT(!c[1] && !c[2] && !c[3]); out[i] = c[0]; } return 0; ) }
int orc_subtable_mle(int kind, size_t C, size_t M, size_t log_r, size_t k, const u64* point_mont, size_t n, u64* out_mont) { GUARD(
  Strategy S = mk_strategy(kind, C, M, log_r); std::vector<Fr> p; for (size_t i = 0; i < n; i++) p.push_back(Fr::from_raw(point_mont + 4 * i));
  Fr r = S.evaluate_subtable_mle(k, p); memcpy(out_mont, r.v, 32); return 0; ) }
int orc_combine_lookups(int kind, size_t C, size_t M, size_t log_r, const u64* vals_mont, u64* out_mont) { GUARD(
  Strategy S = mk_strategy(kind, C, M, log_r); std::vector<Fr> v; for (size_t i = 0; i < S.num_memories(); i++) v.push_back(Fr::from_raw(vals_mont + 4 * i));
  Fr r = S.combine_lookups(v.data()); memcpy(out_mont, r.v, 32); return 0; ) }

// ---- session: dense representation + gens + commitment for one (strategy, indices) instance
struct Session {
  Strategy S; DensifiedRepresentation dense; SparsePolyCommitmentGens gens; SparsePolynomialCommitment commitment; bool committed = false;
  std::vector<Fr> r;
};
static std::map<std::string, SparsePolyCommitmentGens> g_gens_cache;
static const SparsePolyCommitmentGens& cached_gens(size_t c, size_t s, size_t nm, size_t log_m) {
  std::string key = std::to_string(c) + "/" + std::to_string(s) + "/" + std::to_string(nm) + "/" + std::to_string(log_m);
  auto it = g_gens_cache.find(key);
  if (it == g_gens_cache.end()) it = g_gens_cache.emplace(key, SparsePolyCommitmentGens::create("gens_sparse_poly", c, s, nm, log_m)).first;
  return it->second;
}
// indices: n_lookups x C (row-major); r: log2(s) Montgomery scalars
void* orc_session_new(int kind, size_t C, size_t M, size_t log_r, const uint64_t* indices, size_t n_lookups, const u64* r_mont) {
  try {
    auto* se = new Session(); se->S = mk_strategy(kind, C, M, log_r);
    std::vector<std::vector<size_t>> idx(n_lookups, std::vector<size_t>(C));
    for (size_t i = 0; i < n_lookups; i++) for (size_t j = 0; j < C; j++) idx[i][j] = (size_t)indices[i * C + j];
    se->dense = DensifiedRepresentation::from_lookup_indices(idx, C, ark_log2(M));
    se->gens = cached_gens(C, se->dense.s, se->S.num_memories(), ark_log2(M));
    for (size_t i = 0; i < ark_log2(se->dense.s); i++) se->r.push_back(Fr::from_raw(r_mont + 4 * i));
    return se;
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void orc_session_free(void* s) { delete (Session*)s; }
// Replace the session's generators by CALLER-SUPPLIED points — `prove` and `commit` take `gens: &SparsePolyCommitmentGens<G>` (surge.rs:119-125, densified.rs:78-81), whatever
// points the caller holds.  Each set = n + 2 affine points, canonical (x, y) as 8 u64: gens_n.G[0..n), gens_1.G[0], h  (dense_mlpoly.rs:34-45, dot_product.rs:139-150,
// commitments.rs:15-19 / :54-71: both MultiCommitGens of a DotProductProofGens share h).  Sizes are checked against surge.rs:39-47.  The checker for lasso_host_gens_from_points.
int orc_session_set_gens(void* sv, const u64* l_xy, size_t n_l, const u64* m_xy, size_t n_m, const u64* d_xy, size_t n_d) { GUARD(
  Session* se = (Session*)sv;
  auto mk = [](const PolyCommitmentGens& like, const u64* xy, size_t cnt) {
    const size_t n = like.gens.n; if (cnt != n + 2) throw std::runtime_error("orc_session_set_gens: a set needs n + 2 points");
    PolyCommitmentGens g; g.gens.n = n; g.gens.gens_n.n = n; g.gens.gens_1.n = 1;
    for (size_t i = 0; i < n; i++) g.gens.gens_n.G.push_back(pt_from_canon(xy + 8 * i));
    g.gens.gens_1.G.push_back(pt_from_canon(xy + 8 * n));
    g.gens.gens_n.h = pt_from_canon(xy + 8 * (n + 1)); g.gens.gens_1.h = g.gens.gens_n.h; return g; };
  SparsePolyCommitmentGens ng;
  ng.gens_combined_l_variate = mk(se->gens.gens_combined_l_variate, l_xy, n_l);
  ng.gens_combined_log_m_variate = mk(se->gens.gens_combined_log_m_variate, m_xy, n_m);
  ng.gens_derefs = mk(se->gens.gens_derefs, d_xy, n_d);
  se->gens = ng; se->committed = false; return 0; ) }
// commitment bytes: [u64 n1][n1 x 32B compressed][u64 n2][n2 x 32B]
int orc_session_commit(void* sv, uint8_t* out, size_t cap, size_t* len) { GUARD(
  Session* se = (Session*)sv;
  if (!se->committed) { se->commitment = se->dense.commit(se->gens); se->committed = true; }
  ByteWriter w; w.pts_vec(se->commitment.l_variate_polys_commitment); w.pts_vec(se->commitment.log_m_variate_polys_commitment);
  *len = w.b.size(); if (w.b.size() > cap) return -2; memcpy(out, w.b.data(), w.b.size()); return 0; ) }
int orc_session_prove(void* sv, uint8_t* out, size_t cap, size_t* len) { GUARD(
  Session* se = (Session*)sv;
  RandomTape tape("proof"); MerlinTranscript t("example");
  auto P = surge_prove(se->S, se->dense, se->r, se->gens, t, tape);
  auto b = serialize_proof(P);
  *len = b.size(); if (b.size() > cap) return -2; memcpy(out, b.data(), b.size()); return 0; ) }
// The same with LIVE transcripts, as surge.rs:119-125 takes them (&mut Transcript, &mut RandomTape): `pre_*` is a message the caller's protocol absorbed into the transcript
// (label "example") before calling prove, `tape_pre_*` likewise into the random tape's transcript (label "proof", after init_randomness); NULL labels = nothing absorbed.
// The checker for lasso_host_prove_cb's claim that a caller's pre-seeded transcript changes the proof exactly as it changes the reference's.
int orc_session_prove_seeded(void* sv, const char* pre_label, const uint8_t* pre_msg, size_t pre_len, const char* tape_pre_label, const uint8_t* tape_pre_msg, size_t tape_pre_len,
                             uint8_t* out, size_t cap, size_t* len) { GUARD(
  Session* se = (Session*)sv;
  RandomTape tape("proof"); MerlinTranscript t("example");
  if (pre_label) t.append_message(pre_label, pre_msg, pre_len);
  if (tape_pre_label) tape.tape.append_message(tape_pre_label, tape_pre_msg, tape_pre_len);
  auto P = surge_prove(se->S, se->dense, se->r, se->gens, t, tape);
  auto b = serialize_proof(P);
  *len = b.size(); if (b.size() > cap) return -2; memcpy(out, b.data(), b.size()); return 0; ) }
int orc_session_verify_seeded(void* sv, const char* pre_label, const uint8_t* pre_msg, size_t pre_len, const uint8_t* proof, size_t n) { GUARD(
  Session* se = (Session*)sv;
  if (!se->committed) { se->commitment = se->dense.commit(se->gens); se->committed = true; }
  SparsePolynomialEvaluationProof P;
  if (!deserialize_proof(se->S, proof, n, P)) { g_err = "deserialize failed"; return 0; }
  MerlinTranscript t("example");
  if (pre_label) t.append_message(pre_label, pre_msg, pre_len);
  return surge_verify(se->S, P, se->commitment, se->r, se->gens, t) ? 1 : 0; ) }
// prove_cubic_batched (sumcheck.rs:27-135) on caller-supplied arrays with C = EqPolynomial(rand).evals() (grand_product.rs:122-128) and a scripted
// eq point: the literal three-polynomial loop.  A, B: k contiguous arrays of 2^ell Montgomery elements.  out = the honest claim, then (same
// layout as lasso_host_debug_cubic_batched) 3 compressed coefficients per round, the challenges, the final claims of A and of B.
int orc_cubic_batched(size_t k, size_t ell, const uint64_t* A, const uint64_t* B, const uint64_t* rand, const uint64_t* coeffs, const char* tl, uint8_t* out, size_t cap, size_t* len) { try {
  const size_t n = (size_t)1 << ell;
  auto fr_at = [](const uint64_t* p, size_t i) { Fr x; memcpy(&x, p + 4 * i, 32); return x; };
  std::vector<DensePolynomial> pa, pb;
  for (size_t c = 0; c < k; c++) { std::vector<Fr> a(n), b(n); for (size_t i = 0; i < n; i++) { a[i] = fr_at(A, c * n + i); b[i] = fr_at(B, c * n + i); } pa.emplace_back(a); pb.emplace_back(b); }
  std::vector<Fr> rv(ell), cv(k); for (size_t i = 0; i < ell; i++) rv[i] = fr_at(rand, i); for (size_t i = 0; i < k; i++) cv[i] = fr_at(coeffs, i);
  DensePolynomial pc(EqPolynomial(rv).evals());
  Fr claim = Fr::zero();
  for (size_t c = 0; c < k; c++) { Fr sum = Fr::zero(); for (size_t i = 0; i < n; i++) sum += pa[c][i] * pb[c][i] * pc[i]; claim += sum * cv[c]; }
  std::vector<DensePolynomial*> ap, bp; for (size_t c = 0; c < k; c++) { ap.push_back(&pa[c]); bp.push_back(&pb[c]); }
  MerlinTranscript t(tl);
  std::vector<Fr> r_out; CubicClaims cl;
  SumcheckInstanceProof sp = prove_cubic_batched(claim, ell, ap, bp, pc, cv, t, r_out, cl);
  ByteWriter w; w.fr(claim);
  for (auto& c : sp.compressed_polys) w.frs_arr(c.coeffs_except_linear_term);
  w.frs_arr(r_out); w.frs_arr(cl.a); w.frs_arr(cl.b);
  *len = w.b.size(); if (w.b.size() > cap) return -2; memcpy(out, w.b.data(), w.b.size()); return 0;
  } catch (const std::exception& e) { g_err = e.what(); return -1; } }
// returns 1 = verified, 0 = rejected, <0 = error
int orc_session_verify(void* sv, const uint8_t* proof, size_t n) { GUARD(
  Session* se = (Session*)sv;
  if (!se->committed) { se->commitment = se->dense.commit(se->gens); se->committed = true; }
  SparsePolynomialEvaluationProof P;
  if (!deserialize_proof(se->S, proof, n, P)) { g_err = "deserialize failed"; return 0; }
  MerlinTranscript t("example");
  bool ok = surge_verify(se->S, P, se->commitment, se->r, se->gens, t);
  return ok ? 1 : 0; ) }
// verify against an externally produced commitment (same byte layout as orc_session_commit)
int orc_session_verify_with_commitment(void* sv, const uint8_t* proof, size_t n, const uint8_t* comm, size_t cn) { GUARD(
  Session* se = (Session*)sv;
  ByteReader cr(comm, cn); SparsePolynomialCommitment c; c.l_variate_polys_commitment = cr.pts_vec(); c.log_m_variate_polys_commitment = cr.pts_vec();
  if (!cr.ok || cr.pos != cn) { g_err = "bad commitment bytes"; return 0; }
  c.s = se->dense.s; c.log_m = se->dense.log_m; c.m = se->dense.m;
  SparsePolynomialEvaluationProof P;
  if (!deserialize_proof(se->S, proof, n, P)) { g_err = "deserialize failed"; return 0; }
  MerlinTranscript t("example");
  return surge_verify(se->S, P, c, se->r, se->gens, t) ? 1 : 0; ) }

// Verifier only (surge.rs:214-271): needs the strategy, the sizes, the generators, the point, the commitment and the proof — NOT the lookups.
// Lets the -m gpu tests check proofs at BASELINE.json's full sizes (2^20 .. 2^24 lookups) without building a CPU-side dense representation.
int orc_verify_only(int kind, size_t C, size_t M, size_t log_r, size_t s, const u64* r_mont, const uint8_t* proof, size_t n, const uint8_t* comm, size_t cn) { GUARD(
  Strategy S = mk_strategy(kind, C, M, log_r);
  const size_t log_m = ark_log2(M);
  const auto& gens = cached_gens(C, s, S.num_memories(), log_m);
  std::vector<Fr> r; for (size_t i = 0; i < ark_log2(s); i++) r.push_back(Fr::from_raw(r_mont + 4 * i));
  ByteReader cr(comm, cn); SparsePolynomialCommitment c; c.l_variate_polys_commitment = cr.pts_vec(); c.log_m_variate_polys_commitment = cr.pts_vec();
  if (!cr.ok || cr.pos != cn) { g_err = "bad commitment bytes"; return 0; }
  c.s = s; c.log_m = log_m; c.m = M;
  SparsePolynomialEvaluationProof P;
  if (!deserialize_proof(S, proof, n, P)) { g_err = "deserialize failed"; return 0; }
  MerlinTranscript t("example");
  return surge_verify(S, P, c, r, gens, t) ? 1 : 0; ) }

// ---- threads (par.hpp): the oracle's data-parallel loops run on OpenMP; bytes never depend on the count
int orc_max_threads() { return par_max_threads(); }
// physical cores of this host: distinct (physical id, core id) pairs of /proc/cpuinfo; 0 when that cannot be read.  The all-core runs use one thread per
// PHYSICAL core (the reference's RAYON_NUM_THREADS convention of SURVEY 8(d)); filling the SMT siblings as well leaves no idle hardware thread for anything
// else in the process, and a spinning OpenMP barrier then waits out whole scheduler time slices (measured: 2^24 lookups 121 s on 256 threads of a 2 x 64-core box).
int orc_physical_cores() {
  FILE* f = fopen("/proc/cpuinfo", "r"); if (!f) return 0;
  std::map<std::pair<long, long>, int> seen; char line[512]; long phys = -1;
  while (fgets(line, sizeof line, f)) {
    long v; if (sscanf(line, "physical id : %ld", &v) == 1) phys = v; else if (sscanf(line, "core id : %ld", &v) == 1) seen[{phys, v}] = 1;
  }
  fclose(f); return (int)seen.size();
}
// n <= 0: one thread per physical core (capped by what OpenMP was given, e.g. OMP_NUM_THREADS / a cgroup limit)
void orc_set_threads(int n) {
  static const int initial = par_max_threads();
  if (n <= 0) { const int pc = orc_physical_cores(); n = pc > 0 && pc < initial ? pc : initial; }
  par_set_threads(n);
}

// ---- timing leg for bench.py cpu_baseline ("port"): harness inputs, on the threads set by orc_set_threads / OMP_NUM_THREADS.
// proof_out / comm_out (optional, may be null): the serialized proof and commitment (orc_session_commit's layout), so the
// caller can compare them byte for byte with the GPU prover's output on the same instance.
int orc_bench_bytes(int kind, size_t C, size_t M, size_t log_r, size_t s, double* t_densify, double* t_commit, double* t_prove, int do_verify,
                    uint8_t* proof_out, size_t proof_cap, size_t* proof_len, uint8_t* comm_out, size_t comm_cap, size_t* comm_len) { GUARD(
  using clk = std::chrono::steady_clock;
  Strategy S = mk_strategy(kind, C, M, log_r);
  size_t log_m = ark_log2(M);
  auto r = gen_random_point(ark_log2(s));
  auto nz = gen_indices(C, s, M);
  const auto& gens = cached_gens(C, s, S.num_memories(), log_m);
  auto t0 = clk::now();
  auto dense = DensifiedRepresentation::from_lookup_indices(nz, C, log_m);
  auto t1 = clk::now();
  auto commitment = dense.commit(gens);
  auto t2 = clk::now();
  RandomTape tape("proof"); MerlinTranscript t("example");
  auto P = surge_prove(S, dense, r, gens, t, tape);
  auto t3 = clk::now();
  *t_densify = std::chrono::duration<double>(t1 - t0).count();
  *t_commit = std::chrono::duration<double>(t2 - t1).count();
  *t_prove = std::chrono::duration<double>(t3 - t2).count();
  if (proof_out) { auto b = serialize_proof(P); *proof_len = b.size(); if (b.size() > proof_cap) return -2; memcpy(proof_out, b.data(), b.size()); }
  if (comm_out) {
    ByteWriter w; w.pts_vec(commitment.l_variate_polys_commitment); w.pts_vec(commitment.log_m_variate_polys_commitment);
    *comm_len = w.b.size(); if (w.b.size() > comm_cap) return -2; memcpy(comm_out, w.b.data(), w.b.size());
  }
  if (do_verify) { MerlinTranscript tv("example"); if (!surge_verify(S, P, commitment, r, gens, tv)) { g_err = "verify failed"; return -3; } }
  return 0; ) }
int orc_bench(int kind, size_t C, size_t M, size_t log_r, size_t s, double* t_densify, double* t_commit, double* t_prove, int do_verify) {
  return orc_bench_bytes(kind, C, M, log_r, s, t_densify, t_commit, t_prove, do_verify, nullptr, 0, nullptr, nullptr, 0, nullptr); }

}  // extern "C"

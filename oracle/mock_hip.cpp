// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the product path.
//
// CPU implementation of include/lasso_hip.h on top of the oracle's arithmetic ("device" pointers are host
// pointers).  Two uses, both in tests/:
//   * the expected value for every entry point of the real liblasso_hip.so in the `-m gpu` parity tests;
//   * a fake backend so the C++ host prover (lasso_amd/host/) can be exercised end-to-end on a machine
//     without a GPU (`-m "not gpu"`: transcript schedule, proof bytes == oracle proof bytes).
// Each function restates the reference loop it stands for, serially, exactly as cited in lasso_hip.h.
#include "lasso_oracle.hpp"
#include "../include/lasso_hip.h"
#include <cstdlib>
#include <map>
#include <mutex>

using namespace orc;

struct lasso_ctx {
  std::string err;
  // launch/wait split (per context, as on the device): the result of a deferred / *_begin call, handed over by lasso_result_wait
  std::vector<Fr> pending; bool defer = false;
  std::vector<std::vector<Fr>> tail_a, tail_b; std::vector<Fr> tail_e;   // resident tail: private copies of the arrays
  bool tail_linear = false;   // k_linear_tail: tail_a holds the alpha polynomials, tail_b is unused
  size_t tail_stop = 1;       // lasso_tail_handover_next: the tail in progress hands its arrays over at this length; handover_next: the value for the next begin
  uint32_t handover_next = 0;
  // sumcheck rounds / the resident tail launched ahead of their challenge: the arguments wait here for lasso_challenge_post / the first lasso_sumcheck_cubic_tail_next
  // a layer enqueued ahead of its eq point (lasso_sumcheck_cubic_*_begin_eq_ahead): parked until lasso_point_post runs the plain entry point / lasso_point_cancel drops it
  struct LayerAhead { bool on = false, tail = false; std::vector<lasso_fr*> A, B; lasso_fr* E_out = nullptr; size_t n = 0; uint32_t ell = 0, m_stop = 0; } lahead;
  struct RoundAhead { bool on = false, tail = false, linear = false; std::vector<lasso_fr*> A, B; const lasso_fr* E; size_t n; uint32_t m_stop = 0; } rahead;
  // a bullet round launched ahead of its challenge: the arguments wait here for lasso_bullet_post
  std::mutex mem_mu; std::map<void*, size_t> mem_sizes; uint64_t mem_live = 0, mem_peak = 0, alloc_calls = 0;   // lasso_mem_stats of the mock
  struct Ahead { bool on = false, tail = false; const lasso_bases* bs; size_t n, nk; const lasso_fr *a_in, *b_in, *w_in; lasso_fr *a_out, *b_out, *w_out; lasso_fr blinds[2]; lasso_fr scale, tl[2]; } ahead;   // tail: lasso_bullet_tail_ahead (a_out / b_out = the two-element state, nk = nw)
};
struct lasso_bases { std::vector<Point> pts; };

static int32_t fail(lasso_ctx* c, const char* m) { if (c) c->err = m; return LASSO_ERR_INVALID; }
#define REQ(c, cond) do { if (!(cond)) return fail((c), "invalid argument: " #cond); } while (0)
static inline const Fr* F(const lasso_fr* p) { return reinterpret_cast<const Fr*>(p); }
static inline Fr* F(lasso_fr* p) { return reinterpret_cast<Fr*>(p); }
static Strategy mk(const lasso_strategy* s) { Strategy S; S.kind = (StrategyKind)s->kind; S.C = s->c; S.M = (size_t)1 << s->log_m; S.LOG_R = s->log_r; return S; }
#ifdef ORC_BN254
// BN254 build: lasso_point is homogeneous projective (x : y : z), t unused; the oracle's Jacobian point goes out in affine form, the identity as (0 : 1 : 0)
static void put_point(const Point& p, lasso_point* o) {
  memset(o, 0, sizeof(*o));
  if (p.is_identity()) { Fq one = Fq::one(); memcpy(o->y, one.v, 32); return; }
  Fq x, y, one = Fq::one(); p.to_affine(x, y); memcpy(o->x, x.v, 32); memcpy(o->y, y.v, 32); memcpy(o->z, one.v, 32);
}
static Point get_point(const lasso_point* p) {
  Fq X = Fq::from_raw(p->x), Y = Fq::from_raw(p->y), Z = Fq::from_raw(p->z);
  if (Z.is_zero()) return Point::identity();
  Fq zi = Z.inverse(); return Point::from_affine(X * zi, Y * zi);
}
#else
static void put_point(const Point& p, lasso_point* o) { memcpy(o->x, p.X.v, 32); memcpy(o->y, p.Y.v, 32); memcpy(o->t, p.T.v, 32); memcpy(o->z, p.Z.v, 32); }
static Point get_point(const lasso_point* p) { Point q; memcpy(q.X.v, p->x, 32); memcpy(q.Y.v, p->y, 32); memcpy(q.T.v, p->t, 32); memcpy(q.Z.v, p->z, 32); return q; }
#endif

extern "C" {
int32_t lasso_ctx_create(int32_t, lasso_ctx** out) { *out = new lasso_ctx(); return 0; }
int32_t lasso_ctx_create_background(int32_t, int32_t, lasso_ctx** out) { *out = new lasso_ctx(); return 0; }
// every mock context is its own "device" (nothing is shared between two of them: no queues, no gates that could wait behind each other), so slab mode's launched-ahead schedule is
// exercised by the CPU tests; the real library reports the physical device
int32_t lasso_ctx_device_uuid(lasso_ctx* c, uint8_t out[16]) { if (!c || !out) return LASSO_ERR_INVALID; memset(out, 0, 16); const uintptr_t p = (uintptr_t)c; memcpy(out, &p, sizeof(p)); out[15] = 0x4d; return 0; }
void lasso_ctx_destroy(lasso_ctx* c) { delete c; }
const char* lasso_last_error(lasso_ctx* c) { return c ? c->err.c_str() : ""; }
void* lasso_stream(lasso_ctx*) { return nullptr; }
// lasso_alloc / lasso_free with the device's byte accounting (lasso_mem_stats), so that the host prover's buffer policy — recycling pool, capacity mode's evictions and trims —
// can be checked on the CPU: per context live / peak bytes, and the number of lasso_alloc calls (mock_alloc_calls: a steady state of repeated proofs must make none)
int32_t lasso_alloc(lasso_ctx* c, size_t bytes, void** d) {
  REQ(c, d); *d = malloc(bytes ? bytes : 1); if (!*d) return LASSO_ERR_OOM;
  std::lock_guard<std::mutex> g(c->mem_mu); c->mem_sizes[*d] = bytes ? bytes : 1; c->mem_live += bytes ? bytes : 1; if (c->mem_live > c->mem_peak) c->mem_peak = c->mem_live; c->alloc_calls++;
  return 0;
}
int32_t lasso_free(lasso_ctx* c, void* p) {
  if (c && p) { std::lock_guard<std::mutex> g(c->mem_mu); auto it = c->mem_sizes.find(p); if (it != c->mem_sizes.end()) { c->mem_live -= it->second; c->mem_sizes.erase(it); } }
  free(p); return 0;
}
int32_t lasso_trim(lasso_ctx*) { return 0; }
int32_t lasso_mem_stats(lasso_ctx* c, uint64_t* live, uint64_t* peak, int32_t reset) {
  REQ(c, c); std::lock_guard<std::mutex> g(c->mem_mu);
  if (live) *live = c->mem_live; if (peak) *peak = c->mem_peak; if (reset) c->mem_peak = c->mem_live; return 0;
}
extern "C" uint64_t mock_alloc_calls(lasso_ctx* c) { return c ? c->alloc_calls : 0; }
int32_t lasso_upload(lasso_ctx*, void* d, const void* s, size_t n) { memcpy(d, s, n); return 0; }
int32_t lasso_download(lasso_ctx*, void* d, const void* s, size_t n) { memcpy(d, s, n); return 0; }
int32_t lasso_copy(lasso_ctx*, void* d, const void* s, size_t n) { memmove(d, s, n); return 0; }
int32_t lasso_zero(lasso_ctx*, void* d, size_t n) { memset(d, 0, n); return 0; }
int32_t lasso_sync(lasso_ctx*) { return 0; }
// the device-side exchange of slab mode has no CPU statement: the mock reports it unsupported and the host prover keeps to its host-memory exchange
int32_t lasso_rccl_available(void) { return 0; }
int32_t lasso_rccl_unique_id(uint8_t*) { return LASSO_ERR_UNSUPPORTED; }
int32_t lasso_rccl_init(lasso_ctx*, int32_t, int32_t, const uint8_t*) { return LASSO_ERR_UNSUPPORTED; }
int32_t lasso_rccl_ready(lasso_ctx*) { return 0; }
int32_t lasso_rccl_shutdown(lasso_ctx*) { return 0; }
int32_t lasso_rccl_selftest(lasso_ctx*) { return LASSO_ERR_UNSUPPORTED; }
int32_t lasso_rccl_allgather(lasso_ctx*, const void*, void*, size_t) { return LASSO_ERR_UNSUPPORTED; }
size_t lasso_point_row_bytes(void) { return 144; }
int32_t lasso_hyrax_commit_rows_dev(lasso_ctx*, const lasso_fr*, size_t, size_t, const lasso_bases*, void*) { return LASSO_ERR_UNSUPPORTED; }
int32_t lasso_points_reduce_compress(lasso_ctx*, const void*, uint32_t, size_t, uint8_t*) { return LASSO_ERR_UNSUPPORTED; }
int32_t lasso_abort(lasso_ctx* c) { REQ(c, c); c->pending.clear(); c->defer = false; c->ahead.on = false; c->rahead.on = false; c->lahead.on = false; c->handover_next = 0; c->tail_stop = 1; c->tail_a.clear(); c->tail_b.clear(); c->tail_e.clear(); return 0; }
int32_t lasso_prof_get_large(lasso_ctx*, int32_t, uint64_t* n, double* ms, double* b) { if (n) *n = 0; if (ms) *ms = 0; if (b) *b = 0; return 0; }
int32_t lasso_wait_stats(lasso_ctx*, uint64_t* w, double* us, int32_t) { if (w) *w = 0; if (us) *us = 0; return 0; }
int32_t lasso_prof_get_units(lasso_ctx*, int32_t, int32_t, double* u) { if (u) *u = 0; return 0; }
int32_t lasso_prof_enable(lasso_ctx*, int32_t) { return 0; }
int32_t lasso_prof_reset(lasso_ctx*) { return 0; }
int32_t lasso_prof_get(lasso_ctx*, int32_t, uint64_t* l, double* ms, double* b) { if (l) *l = 0; if (ms) *ms = 0; if (b) *b = 0; return 0; }

int32_t lasso_fr_from_u32(lasso_ctx*, const uint32_t* s, size_t n, lasso_fr* d) { for (size_t i = 0; i < n; i++) F(d)[i] = Fr::from_u64(s[i]); return 0; }
int32_t lasso_fr_to_u32(lasso_ctx* c, const lasso_fr* s, size_t n, uint32_t* d, uint32_t* max_out) {
  uint32_t mx = 0;
  for (size_t i = 0; i < n; i++) {
    uint8_t b[32]; F(s)[i].to_bytes_le(b);
    for (int k = 4; k < 32; k++) REQ(c, b[k] == 0);
    d[i] = (uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24; if (d[i] > mx) mx = d[i];
  }
  if (max_out) *max_out = mx;
  return 0;
}
int32_t lasso_gather(lasso_ctx*, const lasso_fr* t, const uint32_t* idx, size_t n, lasso_fr* o) { for (size_t i = 0; i < n; i++) o[i] = t[idx[i]]; return 0; }
int32_t lasso_eq_evals(lasso_ctx*, const lasso_fr* r, uint32_t ell, lasso_fr* o) {
  std::vector<Fr> rr(F(r), F(r) + ell); auto ev = EqPolynomial(rr).evals(); memcpy(o, ev.data(), ev.size() * 32); return 0;
}
int32_t lasso_eq_evals_scaled(lasso_ctx* c, const lasso_fr* r, uint32_t ell, const lasso_fr* scale, lasso_fr* o) {
  lasso_eq_evals(c, r, ell, o);
  if (scale) for (size_t i = 0; i < ((size_t)1 << ell); i++) F(o)[i] = F(o)[i] * *F(scale);
  return 0;
}
int32_t lasso_bind_top(lasso_ctx* c, lasso_fr* const* polys, uint32_t np, size_t n, const lasso_fr* r) {
  REQ(c, n >= 2 && (n & (n - 1)) == 0);
  for (uint32_t p = 0; p < np; p++) { Fr* Z = F(polys[p]); size_t h = n / 2; for (size_t i = 0; i < h; i++) Z[i] = Z[i] + *F(r) * (Z[i + h] - Z[i]); }  // dense_mlpoly.rs:209-216
  return 0;
}
int32_t lasso_sumcheck_cubic_round(lasso_ctx* c, const lasso_fr* const* A, const lasso_fr* const* B, uint32_t nc, const lasso_fr* C, size_t n, lasso_fr* out) {
  REQ(c, n >= 2 && (n & (n - 1)) == 0);
  size_t len = n / 2; const Fr* pc = F(C);
  for (uint32_t k = 0; k < nc; k++) {  // sumcheck.rs:56-93
    const Fr* pa = F(A[k]); const Fr* pb = F(B[k]);
    Fr p0 = Fr::zero(), p2 = Fr::zero(), p3 = Fr::zero();
    for (size_t i = 0; i < len; i++) {
      p0 += pa[i] * pb[i] * pc[i];
      Fr a2 = pa[len + i] + pa[len + i] - pa[i], b2 = pb[len + i] + pb[len + i] - pb[i], c2 = pc[len + i] + pc[len + i] - pc[i];
      p2 += a2 * b2 * c2;
      Fr a3 = a2 + pa[len + i] - pa[i], b3 = b2 + pb[len + i] - pb[i], c3 = c2 + pc[len + i] - pc[i];
      p3 += a3 * b3 * c3;
    }
    F(out)[3 * k] = p0; F(out)[3 * k + 1] = p2; F(out)[3 * k + 2] = p3;
  }
  return 0;
}
// eq-weighted rounds of prove_arbitrary for linear g: two dot products per polynomial against the eq-table prefix
int32_t lasso_sumcheck_linear_eqw_round(lasso_ctx* c, const lasso_fr* const* polys, uint32_t alpha, const lasso_fr* E, size_t n, lasso_fr* out) {
  REQ(c, n >= 2 && (n & (n - 1)) == 0);
  size_t h = n / 2;
  for (uint32_t k = 0; k < alpha; k++) {
    Fr s0 = Fr::zero(), s1 = Fr::zero();
    for (size_t i = 0; i < h; i++) { s0 += F(polys[k])[i] * F(E)[i]; s1 += F(polys[k])[i + h] * F(E)[i]; }
    F(out)[3 * k] = s0; F(out)[3 * k + 1] = s1; F(out)[3 * k + 2] = Fr::zero();
  }
  return 0;
}
int32_t lasso_sumcheck_linear_eqw_round_fused_from(lasso_ctx* c, const lasso_fr* const* src, lasso_fr* const* polys, uint32_t alpha, const lasso_fr* E, size_t n, const lasso_fr* r, lasso_fr* out) {
  REQ(c, n >= 4 && (n & (n - 1)) == 0);
  size_t h = n / 2;
  for (uint32_t k = 0; k < alpha; k++) for (size_t i = 0; i < h; i++) F(polys[k])[i] = F(src[k])[i] + *F(r) * (F(src[k])[i + h] - F(src[k])[i]);
  return lasso_sumcheck_linear_eqw_round(c, polys, alpha, E, h, out);
}
// the same two calls from the polynomials' integer values: lift to the field, then the literal forms
int32_t lasso_sumcheck_linear_eqw_round_u32(lasso_ctx* c, const uint32_t* const* u, uint32_t alpha, const lasso_fr* E, size_t n, lasso_fr* out) {
  std::vector<std::vector<Fr>> z(alpha, std::vector<Fr>(n)); std::vector<const lasso_fr*> ptr(alpha);
  for (uint32_t k = 0; k < alpha; k++) { for (size_t i = 0; i < n; i++) z[k][i] = Fr::from_u64(u[k][i]); ptr[k] = (const lasso_fr*)z[k].data(); }
  return lasso_sumcheck_linear_eqw_round(c, ptr.data(), alpha, E, n, out);
}
int32_t lasso_sumcheck_linear_eqw_round_fused_from_u32(lasso_ctx* c, const uint32_t* const* u, lasso_fr* const* polys, uint32_t alpha, const lasso_fr* E, size_t n, const lasso_fr* r, lasso_fr* out) {
  std::vector<std::vector<Fr>> z(alpha, std::vector<Fr>(n)); std::vector<const lasso_fr*> ptr(alpha);
  for (uint32_t k = 0; k < alpha; k++) { for (size_t i = 0; i < n; i++) z[k][i] = Fr::from_u64(u[k][i]); ptr[k] = (const lasso_fr*)z[k].data(); }
  return lasso_sumcheck_linear_eqw_round_fused_from(c, ptr.data(), polys, alpha, E, n, r, out);
}
int32_t lasso_sumcheck_linear_eqw_round_fused(lasso_ctx* c, lasso_fr* const* polys, uint32_t alpha, const lasso_fr* E, size_t n, const lasso_fr* r, lasso_fr* out) {
  return lasso_sumcheck_linear_eqw_round_fused_from(c, polys, polys, alpha, E, n, r, out);
}
// eq-weighted form: sum_i A(x)[i] B(x)[i] E[i] at x = 0, 2, 3 (the host turns these into sumcheck.rs:56-93's evaluations with three scalars)
int32_t lasso_sumcheck_cubic_eqw_round(lasso_ctx* c, const lasso_fr* const* A, const lasso_fr* const* B, uint32_t nc, const lasso_fr* E, size_t n, lasso_fr* out) {
  REQ(c, n >= 2 && (n & (n - 1)) == 0);
  size_t len = n / 2; const Fr* pe = F(E);
  for (uint32_t k = 0; k < nc; k++) {
    const Fr* pa = F(A[k]); const Fr* pb = F(B[k]);
    Fr p0 = Fr::zero(), p2 = Fr::zero(), p3 = Fr::zero();
    for (size_t i = 0; i < len; i++) {
      p0 += pa[i] * pb[i] * pe[i];
      Fr a2 = pa[len + i] + pa[len + i] - pa[i], b2 = pb[len + i] + pb[len + i] - pb[i];
      p2 += a2 * b2 * pe[i];
      Fr a3 = a2 + pa[len + i] - pa[i], b3 = b2 + pb[len + i] - pb[i];
      p3 += a3 * b3 * pe[i];
    }
    F(out)[3 * k] = p0; F(out)[3 * k + 1] = p2; F(out)[3 * k + 2] = p3;
  }
  return 0;
}
int32_t lasso_sumcheck_cubic_eqw_round_fused(lasso_ctx* c, lasso_fr* const* A, lasso_fr* const* B, uint32_t nc, const lasso_fr* E, size_t n, const lasso_fr* r, lasso_fr* out) {
  REQ(c, n >= 4 && (n & (n - 1)) == 0);
  size_t h = n / 2;
  for (uint32_t k = 0; k < nc; k++) for (size_t i = 0; i < h; i++) { F(A[k])[i] = F(A[k])[i] + *F(r) * (F(A[k])[i + h] - F(A[k])[i]); F(B[k])[i] = F(B[k])[i] + *F(r) * (F(B[k])[i + h] - F(B[k])[i]); }
  return lasso_sumcheck_cubic_eqw_round(c, A, B, nc, E, h, out);
}
// two-sum form: the CPU mock computes at "begin" and hands the result over at "wait"
int32_t lasso_sumcheck_cubic_eqw2_begin(lasso_ctx* c, lasso_fr* const* A, lasso_fr* const* B, uint32_t nc, const lasso_fr* E, size_t n, const lasso_fr* r) {
  REQ(c, n >= (r ? 4u : 2u) && (n & (n - 1)) == 0);
  if (r) {
    size_t h = n / 2;
    for (uint32_t k = 0; k < nc; k++) for (size_t i = 0; i < h; i++) { F(A[k])[i] = F(A[k])[i] + *F(r) * (F(A[k])[i + h] - F(A[k])[i]); F(B[k])[i] = F(B[k])[i] + *F(r) * (F(B[k])[i + h] - F(B[k])[i]); }
    n = h;
  }
  const size_t len = n / 2; const Fr* pe = F(E);
  c->pending.assign(2 * (size_t)nc, Fr::zero());
  for (uint32_t k = 0; k < nc; k++) {
    const Fr* pa = F(A[k]); const Fr* pb = F(B[k]);
    Fr q0 = Fr::zero(), qi = Fr::zero();
    for (size_t i = 0; i < len; i++) { q0 += pa[i] * pb[i] * pe[i]; qi += (pa[len + i] - pa[i]) * (pb[len + i] - pb[i]) * pe[i]; }
    c->pending[2 * k] = q0; c->pending[2 * k + 1] = qi;
  }
  return 0;
}
int32_t lasso_result_wait(lasso_ctx* c, lasso_fr* out, size_t count) {
  REQ(c, out && count == c->pending.size());
  for (size_t i = 0; i < count; i++) F(out)[i] = c->pending[i];
  c->pending.clear();
  return 0;
}
// resident tail: the mock keeps private copies of the arrays (the device kernel keeps them in LDS and never writes them back either)
static void tail_publish(lasso_ctx* ctx) {
  auto& ta = ctx->tail_a; auto& tb = ctx->tail_b; auto& te = ctx->tail_e; auto& pend = ctx->pending;
  const size_t k = ta.size(), m = ta[0].size();
  if (ctx->tail_linear) {
    if (m == 1) { pend.assign(k, Fr::zero()); for (size_t c = 0; c < k; c++) pend[c] = ta[c][0]; ta.clear(); ctx->tail_linear = false; return; }
    pend.assign(2 * k, Fr::zero());
    const size_t h = m / 2;
    for (size_t c = 0; c < k; c++) { Fr s0 = Fr::zero(), s1 = Fr::zero(); for (size_t i = 0; i < h; i++) { s0 += ta[c][i] * te[i]; s1 += ta[c][h + i] * te[i]; } pend[2 * c] = s0; pend[2 * c + 1] = s1; }
    return;
  }
  pend.assign(2 * k, Fr::zero());
  if (m <= ctx->tail_stop) {   // the heads (tail_stop = 1) or the arrays for the host to finish: A_0[0..m), A_1.., B_0..
    pend.assign(2 * k * m, Fr::zero());
    for (size_t c = 0; c < k; c++) for (size_t i = 0; i < m; i++) { pend[c * m + i] = ta[c][i]; pend[(k + c) * m + i] = tb[c][i]; }
    ta.clear(); tb.clear(); ctx->tail_stop = 1; return;
  }
  const size_t h = m / 2;
  for (size_t c = 0; c < k; c++) {
    Fr q0 = Fr::zero(), qi = Fr::zero();
    for (size_t i = 0; i < h; i++) { q0 += ta[c][i] * tb[c][i] * te[i]; qi += (ta[c][h + i] - ta[c][i]) * (tb[c][h + i] - tb[c][i]) * te[i]; }
    pend[2 * c] = q0; pend[2 * c + 1] = qi;
  }
}
static void tail_bind(lasso_ctx* c, const Fr& r) {
  for (auto* arrs : {&c->tail_a, &c->tail_b}) for (auto& v : *arrs) { const size_t h = v.size() / 2; for (size_t i = 0; i < h; i++) v[i] = v[i] + r * (v[i + h] - v[i]); v.resize(h); }
}
uint32_t lasso_sumcheck_tail_capacity(void) { return 512; }
int32_t lasso_sumcheck_cubic_eqw2_begin(lasso_ctx* c, lasso_fr* const* A, lasso_fr* const* B, uint32_t nc, const lasso_fr* E, size_t n, const lasso_fr* r);
int32_t lasso_sumcheck_cubic_tail_begin(lasso_ctx* c, lasso_fr* const* A, lasso_fr* const* B, uint32_t nc, const lasso_fr* E, size_t n, const lasso_fr* r);
// the first round of a layer with the eq table built on the way: here simply the table, then the plain call
int32_t lasso_sumcheck_cubic_eqw2_begin_eq(lasso_ctx* c, lasso_fr* const* A, lasso_fr* const* B, uint32_t nc, lasso_fr* E_out, size_t n, const lasso_fr* point, uint32_t ell, const lasso_fr* scale) {
  REQ(c, E_out && n >= 2 && ((size_t)1 << ell) == n / 2);
  if (ell > 32 || n / 2 <= 64) { c->err = "unsupported table size"; return LASSO_ERR_UNSUPPORTED; }
  int32_t rc = lasso_eq_evals_scaled(c, point, ell, scale, E_out); if (rc) return rc;
  return lasso_sumcheck_cubic_eqw2_begin(c, A, B, nc, E_out, n, nullptr);
}
int32_t lasso_sumcheck_cubic_tail_begin_eq(lasso_ctx* c, lasso_fr* const* A, lasso_fr* const* B, uint32_t nc, size_t n, const lasso_fr* point, uint32_t ell, const lasso_fr* scale) {
  REQ(c, n >= 2 && ((size_t)1 << ell) == n / 2 && ell <= 9);
  std::vector<lasso_fr> E((size_t)1 << ell);
  int32_t rc = lasso_eq_evals_scaled(c, point, ell, scale, E.data()); if (rc) return rc;
  return lasso_sumcheck_cubic_tail_begin(c, A, B, nc, E.data(), n, nullptr);
}
int32_t lasso_tail_handover_next(lasso_ctx* c, uint32_t m_stop) { REQ(c, c && (m_stop & (m_stop - 1)) == 0 && m_stop <= 128); c->handover_next = m_stop <= 1 ? 0 : m_stop; return 0; }
int32_t lasso_layer_ahead_ok(lasso_ctx* c) { const char* v = getenv("LASSO_LAYER_AHEAD"); return c && !(v && v[0] == '0') ? 1 : 0; }
int32_t lasso_sumcheck_cubic_eqw2_begin_eq_ahead(lasso_ctx* c, lasso_fr* const* A, lasso_fr* const* B, uint32_t nc, lasso_fr* E_out, size_t n, uint32_t ell) {
  REQ(c, A && B && E_out && nc >= 1 && n >= 2 && (n & (n - 1)) == 0 && ell < 48 && ((size_t)1 << ell) == n / 2 && !c->lahead.on && !(c->rahead.on && !c->rahead.tail) && !c->ahead.on && !c->defer);
  if (!lasso_layer_ahead_ok(c) || ell > 32 || n / 2 <= 64) { c->err = "unsupported table size"; return LASSO_ERR_UNSUPPORTED; }
  c->lahead.on = true; c->lahead.tail = false; c->lahead.A.assign(A, A + nc); c->lahead.B.assign(B, B + nc); c->lahead.E_out = E_out; c->lahead.n = n; c->lahead.ell = ell; return 0;
}
int32_t lasso_sumcheck_cubic_tail_begin_eq_ahead(lasso_ctx* c, lasso_fr* const* A, lasso_fr* const* B, uint32_t nc, size_t n, uint32_t ell) {
  const uint32_t m_stop = c ? c->handover_next : 0; if (c) c->handover_next = 0;
  REQ(c, A && B && nc >= 1 && n >= 2 && ((size_t)1 << ell) == n / 2 && ell <= 9 && n > (m_stop ? m_stop : 1u) && !c->lahead.on && !(c->rahead.on && !c->rahead.tail) && !c->ahead.on && !c->defer);
  if (!lasso_layer_ahead_ok(c)) { c->err = "switched off"; return LASSO_ERR_UNSUPPORTED; }
  c->lahead.on = true; c->lahead.tail = true; c->lahead.A.assign(A, A + nc); c->lahead.B.assign(B, B + nc); c->lahead.n = n; c->lahead.ell = ell; c->lahead.m_stop = m_stop; return 0;
}
int32_t lasso_point_post(lasso_ctx* c, const lasso_fr* point, uint32_t ell, const lasso_fr* scale) {
  REQ(c, c && c->lahead.on && ell == c->lahead.ell && (point || !ell) && c->pending.empty() && c->tail_a.empty() && !c->rahead.on);
  c->lahead.on = false;
  if (c->lahead.tail) {
    c->handover_next = c->lahead.m_stop;
    return lasso_sumcheck_cubic_tail_begin_eq(c, c->lahead.A.data(), c->lahead.B.data(), (uint32_t)c->lahead.A.size(), c->lahead.n, point, ell, scale);
  }
  return lasso_sumcheck_cubic_eqw2_begin_eq(c, c->lahead.A.data(), c->lahead.B.data(), (uint32_t)c->lahead.A.size(), c->lahead.E_out, c->lahead.n, point, ell, scale);
}
int32_t lasso_point_cancel(lasso_ctx* c) { REQ(c, c && c->lahead.on); c->lahead.on = false; return 0; }
int32_t lasso_sumcheck_cubic_tail_begin(lasso_ctx* c, lasso_fr* const* A, lasso_fr* const* B, uint32_t nc, const lasso_fr* E, size_t n, const lasso_fr* r) {
  const size_t m_stop = c->handover_next ? c->handover_next : 1; c->handover_next = 0;
  REQ(c, n >= (r ? 4u : 2u) && (n & (n - 1)) == 0 && c->tail_a.empty() && c->pending.empty() && !c->rahead.on);
  const size_t q = r ? n / 4 : n / 2; REQ(c, q >= 1 && q <= 512 && 2 * q > m_stop);
  c->tail_stop = m_stop;
  c->tail_a.clear(); c->tail_b.clear();
  for (uint32_t k = 0; k < nc; k++) { c->tail_a.emplace_back(F(A[k]), F(A[k]) + n); c->tail_b.emplace_back(F(B[k]), F(B[k]) + n); }
  c->tail_e.assign(F(E), F(E) + q);
  if (r) tail_bind(c, *F(r));
  tail_publish(c);
  return 0;
}
int32_t lasso_sumcheck_linear_tail_begin(lasso_ctx* c, const lasso_fr* const* src, uint32_t alpha, const lasso_fr* E, size_t n, const lasso_fr* r) {
  REQ(c, n >= (r ? 4u : 2u) && (n & (n - 1)) == 0 && c->tail_a.empty() && c->pending.empty() && !c->handover_next && !c->rahead.on);
  const size_t q = r ? n / 4 : n / 2; REQ(c, q >= 1 && q <= 512);
  c->tail_a.clear(); c->tail_b.clear(); c->tail_linear = true;
  for (uint32_t k = 0; k < alpha; k++) c->tail_a.emplace_back(F(src[k]), F(src[k]) + n);
  c->tail_e.assign(F(E), F(E) + q);
  if (r) tail_bind(c, *F(r));
  tail_publish(c);
  return 0;
}
// launched ahead: the arguments are parked; lasso_challenge_post / the first lasso_sumcheck_cubic_tail_next run the ordinary call with the challenge
int32_t lasso_rounds_ahead_ok(lasso_ctx* c) { const char* v = getenv("LASSO_ROUNDS_AHEAD"); return c && !(v && v[0] == '0') ? 1 : 0; }
int32_t lasso_sumcheck_cubic_eqw2_begin_ahead(lasso_ctx* c, lasso_fr* const* A, lasso_fr* const* B, uint32_t nc, const lasso_fr* E, size_t n) {
  REQ(c, A && B && E && nc >= 1 && n >= 4 && (n & (n - 1)) == 0 && !c->rahead.on && !c->ahead.on && c->tail_a.empty() && !c->defer);
  if (n / 4 <= 64) { c->err = "streaming rounds only"; return LASSO_ERR_UNSUPPORTED; }
  c->rahead.on = true; c->rahead.tail = false; c->rahead.A.assign(A, A + nc); c->rahead.B.assign(B, B + nc); c->rahead.E = E; c->rahead.n = n; return 0;
}
int32_t lasso_sumcheck_linear_eqw_round_fused(lasso_ctx* c, lasso_fr* const* polys, uint32_t alpha, const lasso_fr* E, size_t n, const lasso_fr* r, lasso_fr* out);
int32_t lasso_sumcheck_linear_eqw_round_fused_ahead(lasso_ctx* c, lasso_fr* const* polys, uint32_t alpha, const lasso_fr* E, size_t n) {
  REQ(c, polys && E && alpha >= 1 && n >= 4 && (n & (n - 1)) == 0 && !c->rahead.on && !c->ahead.on && c->tail_a.empty() && !c->defer);
  c->rahead.on = true; c->rahead.tail = false; c->rahead.linear = true; c->rahead.A.assign(polys, polys + alpha); c->rahead.B.clear(); c->rahead.E = E; c->rahead.n = n; return 0;
}
int32_t lasso_challenge_post(lasso_ctx* c, const lasso_fr* r) {
  REQ(c, r && c->rahead.on && !c->rahead.tail && c->pending.empty());
  c->rahead.on = false;
  if (c->rahead.linear) {   // parked for lasso_result_wait, as after lasso_defer_next
    c->rahead.linear = false;
    std::vector<lasso_fr> out(3 * c->rahead.A.size());
    int32_t rc = lasso_sumcheck_linear_eqw_round_fused(c, c->rahead.A.data(), (uint32_t)c->rahead.A.size(), c->rahead.E, c->rahead.n, r, out.data()); if (rc) return rc;
    c->pending.assign(F(out.data()), F(out.data()) + out.size()); return 0;
  }
  return lasso_sumcheck_cubic_eqw2_begin(c, c->rahead.A.data(), c->rahead.B.data(), (uint32_t)c->rahead.A.size(), c->rahead.E, c->rahead.n, r);
}
int32_t lasso_sumcheck_cubic_tail_begin_ahead(lasso_ctx* c, lasso_fr* const* A, lasso_fr* const* B, uint32_t nc, const lasso_fr* E, size_t n) {
  REQ(c, A && B && E && nc >= 1 && n >= 4 && (n & (n - 1)) == 0 && n / 4 <= 512 && !c->rahead.on && !c->ahead.on && c->tail_a.empty() && !c->defer);
  REQ(c, n / 2 > (c->handover_next ? c->handover_next : 1u));
  c->rahead.on = true; c->rahead.tail = true; c->rahead.A.assign(A, A + nc); c->rahead.B.assign(B, B + nc); c->rahead.E = E; c->rahead.n = n;
  c->rahead.m_stop = c->handover_next; c->handover_next = 0;   // consumed by THIS begin, as on the device (another begin may be enqueued before this tail starts)
  return 0;
}
int32_t lasso_sumcheck_cubic_tail_next(lasso_ctx* c, const lasso_fr* r) {
  if (c && c->rahead.on && c->rahead.tail) {   // the first challenge of a tail launched ahead
    REQ(c, r && c->pending.empty());
    c->rahead.on = false;
    { const uint32_t later = c->handover_next; c->handover_next = c->rahead.m_stop; const int32_t rc = lasso_sumcheck_cubic_tail_begin(c, c->rahead.A.data(), c->rahead.B.data(), (uint32_t)c->rahead.A.size(), c->rahead.E, c->rahead.n, r); c->handover_next = later; return rc; }
    return lasso_sumcheck_cubic_tail_begin(c, c->rahead.A.data(), c->rahead.B.data(), (uint32_t)c->rahead.A.size(), c->rahead.E, c->rahead.n, r);
  }
  REQ(c, r && !c->tail_a.empty() && c->pending.empty());
  tail_bind(c, *F(r));
  tail_publish(c);
  return 0;
}
int32_t lasso_sumcheck_combine_round(lasso_ctx* c, const lasso_strategy* s, const lasso_fr* const* polys, const lasso_fr* eq, size_t n, uint32_t degree, lasso_fr* out) {
  REQ(c, n >= 2 && (n & (n - 1)) == 0);
  Strategy S = mk(s); size_t alpha = S.num_memories(), half = n / 2;
  REQ(c, degree == S.sumcheck_poly_degree());
  std::vector<Fr> ev(degree + 1, Fr::zero()), lo(alpha + 1), hi(alpha + 1), cur(alpha + 1);
  for (size_t i = 0; i < half; i++) {  // sumcheck.rs:179-218
    for (size_t j = 0; j < alpha; j++) { lo[j] = F(polys[j])[i]; hi[j] = F(polys[j])[half + i]; }
    lo[alpha] = F(eq)[i]; hi[alpha] = F(eq)[half + i];
    ev[0] += S.combine_lookups_eq(lo.data()); ev[1] += S.combine_lookups_eq(hi.data());
    cur = hi;
    for (uint32_t k = 2; k <= degree; k++) { for (size_t j = 0; j <= alpha; j++) cur[j] = cur[j] + hi[j] - lo[j]; ev[k] += S.combine_lookups_eq(cur.data()); }
  }
  memcpy(out, ev.data(), ev.size() * 32); return 0;
}
// LT_m <- 32^-(C-1-m) LT_m (include/lasso_hip.h); the scaled round = the literal round on unscaled copies
static Fr pow32(size_t e, bool inverse) { Fr b = Fr::from_u64(32); if (inverse) b = b.inverse(); Fr r = Fr::one(); for (size_t i = 0; i < e; i++) r = r * b; return r; }
int32_t lasso_lt_prescale(lasso_ctx* c, const lasso_strategy* s, const lasso_fr* const* src, lasso_fr* const* polys, size_t n) {
  REQ(c, s && s->kind == LASSO_LT && polys && n >= 1);
  if (src) for (size_t i = 0; i < 2 * (size_t)s->c; i++) if (src[i] != polys[i]) memcpy(polys[i], src[i], n * sizeof(lasso_fr));
  for (size_t m = 0; m + 1 < s->c; m++) { const Fr k = pow32(s->c - 1 - m, true); for (size_t i = 0; i < n; i++) F(polys[2 * m])[i] = F(polys[2 * m])[i] * k; }
  return 0;
}
int32_t lasso_sumcheck_combine_round_lt_scaled(lasso_ctx* c, const lasso_strategy* s, const lasso_fr* const* polys, const lasso_fr* eq, size_t n, uint32_t degree, lasso_fr* out) {
  REQ(c, s && s->kind == LASSO_LT);
  std::vector<std::vector<lasso_fr>> tmp(s->c); std::vector<const lasso_fr*> ptrs(polys, polys + 2 * s->c);
  for (size_t m = 0; m + 1 < s->c; m++) {
    const Fr k = pow32(s->c - 1 - m, false); tmp[m].assign(polys[2 * m], polys[2 * m] + n);
    for (size_t i = 0; i < n; i++) F(tmp[m].data())[i] = F(tmp[m].data())[i] * k;
    ptrs[2 * m] = tmp[m].data();
  }
  return lasso_sumcheck_combine_round(c, s, ptrs.data(), eq, n, degree, out);
}
int32_t lasso_sumcheck_combine_round_lt_u32(lasso_ctx* c, const lasso_strategy* s, const uint32_t* const* u, const lasso_fr* eq, size_t n, uint32_t degree, lasso_fr* out) {
  REQ(c, s && s->kind == LASSO_LT && u);
  const size_t alpha = 2 * (size_t)s->c;
  std::vector<std::vector<Fr>> z(alpha, std::vector<Fr>(n)); std::vector<const lasso_fr*> ptr(alpha);
  for (size_t k = 0; k < alpha; k++) { for (size_t i = 0; i < n; i++) { REQ(c, u[k][i] <= 1); z[k][i] = Fr::from_u64(u[k][i]); } ptr[k] = (const lasso_fr*)z[k].data(); }
  return lasso_sumcheck_combine_round(c, s, ptr.data(), eq, n, degree, out);
}
int32_t lasso_combine_claim(lasso_ctx* c, const lasso_strategy* s, const lasso_fr* const* polys, const lasso_fr* eq, size_t n, lasso_fr* out) {
  Strategy S = mk(s); size_t alpha = S.num_memories(); std::vector<Fr> v(alpha); Fr claim = Fr::zero();
  for (size_t k = 0; k < n; k++) { for (size_t j = 0; j < alpha; j++) v[j] = F(polys[j])[k]; claim += F(eq)[k] * S.combine_lookups(v.data()); }  // subtables/mod.rs:197-213
  if (c->defer) { c->defer = false; c->pending.assign(1, claim); return 0; }   // lasso_defer_next
  *F(out) = claim; return 0;
}
int32_t lasso_multi_dot(lasso_ctx* c, const lasso_fr* const* polys, uint32_t k, const lasso_fr* w, size_t n, lasso_fr* out) {
  if (c->defer) { c->defer = false; c->pending.resize(k); for (uint32_t p = 0; p < k; p++) c->pending[p] = compute_dotproduct(F(polys[p]), F(w), n); return 0; }   // lasso_defer_next
  for (uint32_t p = 0; p < k; p++) F(out)[p] = compute_dotproduct(F(polys[p]), F(w), n); return 0;
}
int32_t lasso_read_heads(lasso_ctx*, const lasso_fr* const* polys, uint32_t k, lasso_fr* out) { for (uint32_t p = 0; p < k; p++) out[p] = polys[p][0]; return 0; }
int32_t lasso_read_runs(lasso_ctx* c, const lasso_fr* const* polys, uint32_t k, uint32_t count, lasso_fr* out) { REQ(c, polys && out && k >= 1 && count >= 1 && (size_t)k * count <= 16384); for (uint32_t p = 0; p < k; p++) for (uint32_t j = 0; j < count; j++) out[p * count + j] = polys[p][j]; return 0; }
int32_t lasso_gp_build(lasso_ctx* c, lasso_fr* tree, size_t n) {
  REQ(c, n >= 2 && (n & (n - 1)) == 0);
  Fr* in = F(tree); size_t len = n;
  while (len > 2) { size_t h = len / 2; Fr* o = in + len; for (size_t i = 0; i < h; i++) o[i] = in[i] * in[i + h]; in = o; len = h; }  // grand_product.rs:20-58
  return 0;
}
int32_t lasso_fingerprint_ops(lasso_ctx*, const lasso_fr* table, const uint32_t* dim, const lasso_fr* read, size_t s, const lasso_fr* gamma, const lasso_fr* tau, lasso_fr* ro, lasso_fr* wo) {
  Fr g = *F(gamma), g2 = g.square(), t = *F(tau);
  for (size_t i = 0; i < s; i++) {  // memory_checking.rs:252, :284-301
    Fr a = Fr::from_u64(dim[i]), v = F(table)[dim[i]];
    F(ro)[i] = F(read)[i] * g2 + v * g + a - t;
    F(wo)[i] = (F(read)[i] + Fr::one()) * g2 + v * g + a - t;
  }
  return 0;
}
int32_t lasso_fingerprint_ops_gp(lasso_ctx* c, const lasso_fr* table, const uint32_t* dim, const lasso_fr* read, size_t s, const lasso_fr* gamma, const lasso_fr* tau, lasso_fr* tr, lasso_fr* tw) {
  REQ(c, s >= 4 && (s & (s - 1)) == 0);   // the device's fused form == the literal steps: fingerprints (memory_checking.rs:284-301), then GrandProductCircuit::new of each (grand_product.rs:38-58)
  int32_t rc = lasso_fingerprint_ops(c, table, dim, read, s, gamma, tau, tr, tw); if (rc) return rc;
  rc = lasso_gp_build(c, tr, s); if (rc) return rc;
  return lasso_gp_build(c, tw, s);
}
// capacity mode: the literal steps into temporaries, then the layers above the leaves / the requested strips
int32_t lasso_fingerprint_ops_gp_upper(lasso_ctx* c, const lasso_fr* table, const uint32_t* dim, const lasso_fr* read, size_t s, const lasso_fr* gamma, const lasso_fr* tau, lasso_fr* ur, lasso_fr* uw) {
  REQ(c, s >= 4 && (s & (s - 1)) == 0);
  std::vector<lasso_fr> tr(2 * s), tw(2 * s);
  int32_t rc = lasso_fingerprint_ops_gp(c, table, dim, read, s, gamma, tau, tr.data(), tw.data()); if (rc) return rc;
  memcpy(ur, tr.data() + s, (s - 2) * 32); memcpy(uw, tw.data() + s, (s - 2) * 32); return 0;
}
int32_t lasso_fingerprint_ops_strips(lasso_ctx* c, const lasso_fr* table, const uint32_t* dim, const lasso_fr* read, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                     uint32_t nstrips, size_t i0, size_t cs, lasso_fr* out_r, lasso_fr* out_w) {
  REQ(c, s >= 4 && (s & (s - 1)) == 0 && (nstrips == 2 || nstrips == 4) && cs >= 1 && i0 + cs <= s / 2 / nstrips);
  std::vector<lasso_fr> lr(s), lw(s);
  int32_t rc = lasso_fingerprint_ops(c, table, dim, read, s, gamma, tau, lr.data(), lw.data()); if (rc) return rc;
  const size_t stride = s / 2 / nstrips;
  for (size_t arr = 0; arr < 2; arr++) for (size_t t = 0; t < nstrips; t++) for (size_t i = 0; i < cs; i++) {
    const size_t k = arr * (s / 2) + t * stride + i0 + i, j = (arr * nstrips + t) * cs + i;
    out_r[j] = lr[k]; out_w[j] = lw[k];
  }
  return 0;
}
int32_t lasso_fingerprint_ops_gp_upper_u32(lasso_ctx* c, const lasso_fr* table, const uint32_t* dim, const uint32_t* read32, size_t s, const lasso_fr* gamma, const lasso_fr* tau, lasso_fr* ur, lasso_fr* uw) {
  std::vector<lasso_fr> read(s); lasso_fr_from_u32(c, read32, s, read.data());
  return lasso_fingerprint_ops_gp_upper(c, table, dim, read.data(), s, gamma, tau, ur, uw);
}
int32_t lasso_fingerprint_ops_strips_u32(lasso_ctx* c, const lasso_fr* table, const uint32_t* dim, const uint32_t* read32, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                         uint32_t nstrips, size_t i0, size_t cs, lasso_fr* out_r, lasso_fr* out_w) {
  std::vector<lasso_fr> read(s); lasso_fr_from_u32(c, read32, s, read.data());
  return lasso_fingerprint_ops_strips(c, table, dim, read.data(), s, gamma, tau, nstrips, i0, cs, out_r, out_w);
}
int32_t lasso_fingerprint_mem(lasso_ctx*, const lasso_fr* table, const lasso_fr* fin, size_t m, const lasso_fr* gamma, const lasso_fr* tau, lasso_fr* io, lasso_fr* fo) {
  Fr g = *F(gamma), g2 = g.square(), t = *F(tau);
  for (size_t i = 0; i < m; i++) {  // memory_checking.rs:257-273
    F(io)[i] = Fr::zero() * g2 + F(table)[i] * g + Fr::from_u64(i) - t;
    F(fo)[i] = F(fin)[i] * g2 + F(table)[i] * g + Fr::from_u64(i) - t;
  }
  return 0;
}
// slab mode: local index i stands for the global address i*world + rank (table is the whole subtable; fin and the outputs are local)
int32_t lasso_fingerprint_mem_slab(lasso_ctx*, const lasso_fr* table, const lasso_fr* fin, size_t m, uint32_t world, uint32_t rank, const lasso_fr* gamma, const lasso_fr* tau, lasso_fr* io, lasso_fr* fo) {
  Fr g = *F(gamma), g2 = g.square(), t = *F(tau);
  for (size_t i = 0; i < m; i++) {
    const size_t a = i * world + rank;
    F(io)[i] = F(table)[a] * g + Fr::from_u64(a) - t;
    F(fo)[i] = F(fin)[i] * g2 + F(table)[a] * g + Fr::from_u64(a) - t;
  }
  return 0;
}
int32_t lasso_matvec_left(lasso_ctx*, const lasso_fr* Z, const lasso_fr* L, size_t ls, size_t rs, lasso_fr* out) {
  for (size_t i = 0; i < rs; i++) { Fr s = Fr::zero(); for (size_t j = 0; j < ls; j++) s += F(L)[j] * F(Z)[j * rs + i]; F(out)[i] = s; }  // dense_mlpoly.rs:184-207
  return 0;
}
// densified.rs:32-57, the reference's serial loop for one dimension
int32_t lasso_densify_dim(lasso_ctx* c, const uint64_t* idx, size_t n_lookups, size_t C, size_t dim, size_t s, uint32_t log_m, uint32_t* dim_u32, lasso_fr* d_dim, lasso_fr* d_read, lasso_fr* d_final) {
  REQ(c, dim < C && n_lookups <= s);
  const size_t m = (size_t)1 << log_m;
  std::vector<size_t> access(s, 0), final_ts(m, 0), read_ts(s, 0);
  for (size_t k = 0; k < n_lookups; k++) { access[k] = (size_t)idx[k * C + dim]; REQ(c, access[k] < m); }
  for (size_t k = 0; k < s; k++) { size_t a = access[k]; size_t ts = final_ts[a]; read_ts[k] = ts; final_ts[a] = ts + 1; }
  for (size_t k = 0; k < s; k++) { dim_u32[k] = (uint32_t)access[k]; F(d_dim)[k] = Fr::from_u64(access[k]); F(d_read)[k] = Fr::from_u64(read_ts[k]); }
  for (size_t a = 0; a < m; a++) F(d_final)[a] = Fr::from_u64(final_ts[a]);
  return 0;
}
// slab mode: the same global timestamps, only the rank's residue class kept (global k -> local k / world)
int32_t lasso_densify_dim_slab(lasso_ctx* c, const uint64_t* idx, size_t n_lookups, size_t C, size_t dim, size_t s, uint32_t log_m, uint32_t world, uint32_t rank, uint32_t* dim_u32, lasso_fr* d_dim,
                               lasso_fr* d_read, lasso_fr* d_final) {
  const size_t m = (size_t)1 << log_m;
  REQ(c, world >= 1 && rank < world && s % world == 0 && m % world == 0);
  std::vector<uint32_t> u(s); std::vector<lasso_fr> fd(s), fr(s), ff(m);
  int32_t rc = lasso_densify_dim(c, idx, n_lookups, C, dim, s, log_m, u.data(), fd.data(), fr.data(), ff.data()); if (rc) return rc;
  for (size_t k = rank; k < s; k += world) { dim_u32[k / world] = u[k]; d_dim[k / world] = fd[k]; d_read[k / world] = fr[k]; }
  for (size_t a = rank; a < m; a += world) d_final[a / world] = ff[a];
  return 0;
}
int32_t lasso_bases_create(lasso_ctx*, const lasso_affine* pts, size_t n, lasso_bases** out);
int32_t lasso_bases_create_opt(lasso_ctx* c, const lasso_affine* pts, size_t n, int32_t, lasso_bases** out) { return lasso_bases_create(c, pts, n, out); }
int32_t lasso_bases_create(lasso_ctx*, const lasso_affine* pts, size_t n, lasso_bases** out) {
  auto* b = new lasso_bases();
  for (size_t i = 0; i < n; i++) b->pts.push_back(Point::from_affine(Fq::from_raw(pts[i].x), Fq::from_raw(pts[i].y)));
  *out = b; return 0;
}
void lasso_bases_destroy(lasso_ctx*, lasso_bases* b) { delete b; }
int32_t lasso_bases_prepare(lasso_ctx* c, const lasso_bases* b, uint32_t byte_windows) { (void)c; return (b && byte_windows >= 1 && byte_windows <= 2) ? 0 : LASSO_ERR_INVALID; }   // the mock holds no tables
int32_t lasso_hyrax_commit(lasso_ctx* c, const lasso_fr* Z, size_t ls, size_t rs, const lasso_bases* b, lasso_point* out) {
  REQ(c, rs <= b->pts.size());
  std::vector<Point> bases(b->pts.begin(), b->pts.begin() + rs);
  for (size_t i = 0; i < ls; i++) { std::vector<Fr> sc(F(Z) + i * rs, F(Z) + (i + 1) * rs); put_point(msm(bases, sc), out + i); }  // dense_mlpoly.rs:118-127, commitments.rs:84-93 with blind = 0
  return 0;
}
int32_t lasso_hyrax_commit_compressed(lasso_ctx* c, const lasso_fr* Z, size_t ls, size_t rs, const lasso_bases* b, uint8_t* out32) {
  REQ(c, rs <= b->pts.size());
  std::vector<Point> bases(b->pts.begin(), b->pts.begin() + rs);
  for (size_t i = 0; i < ls; i++) { std::vector<Fr> sc(F(Z) + i * rs, F(Z) + (i + 1) * rs); msm(bases, sc).compress(out32 + 32 * i); }
  return 0;
}
int32_t lasso_hyrax_commit_compressed_u32(lasso_ctx* c, const uint32_t* Z, uint32_t max_value, size_t ls, size_t rs, const lasso_bases* b, uint8_t* out32) {
  REQ(c, rs <= b->pts.size());
  std::vector<Point> bases(b->pts.begin(), b->pts.begin() + rs);
  for (size_t i = 0; i < ls; i++) { std::vector<Fr> sc(rs); for (size_t j = 0; j < rs; j++) { REQ(c, Z[i * rs + j] <= max_value); sc[j] = Fr::from_u64(Z[i * rs + j]); } msm(bases, sc).compress(out32 + 32 * i); }
  return 0;
}
int32_t lasso_materialize_subtable_u32(lasso_ctx* c, const lasso_strategy* s, uint32_t sub, uint32_t* o) {
  Strategy S = mk(s); auto tabs = S.materialize_subtables();   // the oracle's restatement of and.rs / or.rs / xor.rs / lt.rs / range_check.rs
  REQ(c, sub < tabs.size());
  for (size_t i = 0; i < tabs[sub].size(); i++) { u64 cc[4]; tabs[sub][i].to_canonical(cc); REQ(c, cc[1] == 0 && cc[2] == 0 && cc[3] == 0 && cc[0] < ((u64)1 << 32)); o[i] = (uint32_t)cc[0]; }
  return 0;
}
int32_t lasso_gather_u32(lasso_ctx*, const uint32_t* t, const uint32_t* idx, size_t n, uint32_t* o) { for (size_t i = 0; i < n; i++) o[i] = t[idx[i]]; return 0; }
int32_t lasso_msm(lasso_ctx* c, const lasso_bases* b, const lasso_fr* scalars, size_t n, lasso_point* out) {
  REQ(c, n <= b->pts.size());
  std::vector<Point> bases(b->pts.begin(), b->pts.begin() + n); std::vector<Fr> sc(F(scalars), F(scalars) + n);
  put_point(msm(bases, sc), out); return 0;
}

// lasso_defer_next in the mock: the deferred call computes at once and parks its result for lasso_result_wait (points as 4 field elements each)
int32_t lasso_defer_next(lasso_ctx* c) { REQ(c, !c->defer && c->pending.empty()); c->defer = true; return 0; }
static int32_t deliver_points(lasso_ctx* c, const lasso_point* pts, size_t n, lasso_point* out) {
  if (c->defer) { c->defer = false; c->pending.resize(4 * n); memcpy((void*)c->pending.data(), pts, n * sizeof(lasso_point)); return 0; }
  memcpy(out, pts, n * sizeof(lasso_point)); return 0;
}
int32_t lasso_msm_dev(lasso_ctx* c, const lasso_bases* b, const lasso_fr* scalars, size_t n, lasso_point* out) {
  lasso_point tmp; int32_t rc = lasso_msm(c, b, scalars, n, &tmp); if (rc) return rc;
  return deliver_points(c, &tmp, 1, out);
}
int32_t lasso_matvec_left_dev(lasso_ctx* c, const lasso_fr* Z, const lasso_fr* L, size_t ls, size_t rs, lasso_fr* out) { return lasso_matvec_left(c, Z, L, ls, rs, out); }
int32_t lasso_fr_to_bytes(lasso_ctx*, const lasso_fr* src, size_t n, uint8_t* out) { for (size_t i = 0; i < n; i++) { u64 cc[4]; F(src)[i].to_canonical(cc); memcpy(out + 32 * i, cc, 32); } return 0; }
int32_t lasso_msm_dev_scaled(lasso_ctx* c, const lasso_bases* b, const lasso_fr* sc, size_t n, const lasso_fr* scale, const lasso_fr* tail, lasso_point* out) {
  REQ(c, n + 2 <= b->pts.size());
  std::vector<Point> bases(b->pts.begin(), b->pts.begin() + n + 2); std::vector<Fr> s;
  for (size_t i = 0; i < n; i++) s.push_back(F(sc)[i] * *F(scale));
  s.push_back(F(tail)[0]); s.push_back(F(tail)[1]);
  lasso_point tmp; put_point(msm(bases, s), &tmp); return deliver_points(c, &tmp, 1, out);
}
int32_t lasso_inner_products_lr(lasso_ctx*, const lasso_fr* a, const lasso_fr* b, size_t nk, lasso_fr* out) {
  size_t h = nk / 2; F(out)[0] = inner_product(F(a), F(b) + h, h); F(out)[1] = inner_product(F(a) + h, F(b), h); return 0;   // bullet.rs:79-80
}
// bullet.rs:84-118 on the virtually folded generators G^(k)_i = sum_blk w_blk G_{blk*nk+i}: fold G explicitly (as the reference does) then MSM
int32_t lasso_bullet_lr(lasso_ctx* c, const lasso_bases* b, size_t n, const lasso_fr* a, size_t nk, const lasso_fr* w, const lasso_fr* tail, lasso_point* out) {
  REQ(c, n + 2 <= b->pts.size() && nk >= 2 && nk <= n);
  size_t nw = n / nk, h = nk / 2;
  std::vector<Point> G(nk, Point::identity());
  for (size_t i = 0; i < nk; i++) for (size_t blk = 0; blk < nw; blk++) G[i] += b->pts[blk * nk + i] * F(w)[blk];
  const Point& Q = b->pts[n]; const Point& H = b->pts[n + 1];
  std::vector<Point> bs(G.begin() + h, G.end()); bs.push_back(Q); bs.push_back(H);
  std::vector<Fr> sc(F(a), F(a) + h); sc.push_back(F(tail)[0]); sc.push_back(F(tail)[1]);
  put_point(msm(bs, sc), out);
  bs.assign(G.begin(), G.begin() + h); bs.push_back(Q); bs.push_back(H);
  sc.assign(F(a) + h, F(a) + nk); sc.push_back(F(tail)[2]); sc.push_back(F(tail)[3]);
  put_point(msm(bs, sc), out + 1);
  return 0;
}
// the fused round = (optional) fold of the previous challenge, then inner products + L/R on the folded state, restated with the pieces above
int32_t lasso_bullet_fold(lasso_ctx*, lasso_fr* a, lasso_fr* b, size_t nk, const lasso_fr* w, size_t nw, lasso_fr* w_out, const lasso_fr* u, const lasso_fr* u_inv);
int32_t lasso_bullet_round(lasso_ctx* c, const lasso_bases* bs, size_t n, const lasso_fr* a_in, const lasso_fr* b_in, const lasso_fr* w_in, lasso_fr* a_out, lasso_fr* b_out, lasso_fr* w_out, size_t nk,
                           const lasso_fr* u, const lasso_fr* u_inv, const lasso_fr* blinds, lasso_point* out) {
  const lasso_fr *a = a_in, *b = b_in, *w = w_in;
  std::vector<lasso_fr> ta, tb;
  if (u) {
    REQ(c, u_inv && a_out && b_out && w_out && 2 * nk <= n);
    ta.assign(a_in, a_in + 2 * nk); tb.assign(b_in, b_in + 2 * nk);
    lasso_bullet_fold(c, ta.data(), tb.data(), 2 * nk, w_in, n / (2 * nk), w_out, u, u_inv);
    memcpy(a_out, ta.data(), nk * sizeof(lasso_fr)); memcpy(b_out, tb.data(), nk * sizeof(lasso_fr));
    a = a_out; b = b_out; w = w_out;
  }
  lasso_fr cc[2]; lasso_inner_products_lr(c, a, b, nk, cc);
  lasso_fr tail[4] = {cc[0], blinds[0], cc[1], blinds[1]};
  lasso_point lr[2]; int32_t rc = lasso_bullet_lr(c, bs, n, a, nk, w, tail, lr); if (rc) return rc;
  return deliver_points(c, lr, 2, out);
}
// a folding round launched ahead of its challenge: nothing can be computed before the challenge exists, so the mock records the launch and runs it when the challenge is posted —
// the device's order of effects (round k reads what round k-1 wrote; the result is collected by lasso_result_wait)
int32_t lasso_bullet_ahead_ok(lasso_ctx* c, const lasso_bases* b) { const char* v = getenv("LASSO_BULLET_AHEAD"); return c && b && !(v && v[0] == '0') ? 1 : 0; }
int32_t lasso_bullet_round_ahead(lasso_ctx* c, const lasso_bases* bs, size_t n, const lasso_fr* a_in, const lasso_fr* b_in, const lasso_fr* w_in, lasso_fr* a_out, lasso_fr* b_out, lasso_fr* w_out, size_t nk,
                                 const lasso_fr* blinds) {
  REQ(c, bs && a_in && b_in && w_in && a_out && b_out && w_out && blinds && nk >= 2 && 2 * nk <= n && !c->ahead.on && !c->defer);
  c->ahead.on = true; c->ahead.bs = bs; c->ahead.n = n; c->ahead.nk = nk; c->ahead.a_in = a_in; c->ahead.b_in = b_in; c->ahead.w_in = w_in; c->ahead.a_out = a_out; c->ahead.b_out = b_out; c->ahead.w_out = w_out;
  c->ahead.blinds[0] = blinds[0]; c->ahead.blinds[1] = blinds[1];
  return 0;
}
int32_t lasso_bullet_tail_ahead_ok(lasso_ctx* c, const lasso_bases* b) { const char* v = getenv("LASSO_BULLET_TAIL_AHEAD"); return lasso_bullet_ahead_ok(c, b) && !(v && v[0] == '0') ? 1 : 0; }
int32_t lasso_bullet_tail_ahead(lasso_ctx* c, const lasso_bases* bs, size_t n, lasso_fr* a, lasso_fr* b, const lasso_fr* w, size_t nw, lasso_fr* w_out, const lasso_fr* scale, const lasso_fr* tail) {
  REQ(c, bs && a && b && w && w_out && scale && tail && n >= 2 && 2 * nw == n && !c->ahead.on && !c->defer);
  c->ahead.on = true; c->ahead.tail = true; c->ahead.bs = bs; c->ahead.n = n; c->ahead.nk = nw; c->ahead.a_out = a; c->ahead.b_out = b; c->ahead.w_in = w; c->ahead.w_out = w_out;
  c->ahead.scale = *scale; c->ahead.tl[0] = tail[0]; c->ahead.tl[1] = tail[1];
  return 0;
}
int32_t lasso_msm_dev_scaled(lasso_ctx* c, const lasso_bases* b, const lasso_fr* sc, size_t n, const lasso_fr* scale, const lasso_fr* tail, lasso_point* out);
int32_t lasso_bullet_post(lasso_ctx* c, const lasso_fr* u, const lasso_fr* u_inv) {
  REQ(c, u && u_inv && c->ahead.on && c->pending.empty() && !c->defer);
  if (c->ahead.tail) {   // the opening's tail chain: fold, heads, delta MSM — the result is the point (4 values) and the two heads
    c->ahead.on = false; c->ahead.tail = false;
    lasso_bullet_fold(c, c->ahead.a_out, c->ahead.b_out, 2, c->ahead.w_in, c->ahead.nk, c->ahead.w_out, u, u_inv);
    lasso_point dl; const int32_t rc = lasso_msm_dev_scaled(c, c->ahead.bs, c->ahead.w_out, c->ahead.n, &c->ahead.scale, c->ahead.tl, &dl); if (rc) return rc;
    c->pending.resize(6); memcpy((void*)c->pending.data(), &dl, sizeof(lasso_point));
    c->pending[4] = F(c->ahead.a_out)[0]; c->pending[5] = F(c->ahead.b_out)[0];
    return 0;
  }
  c->ahead.on = false; c->defer = true;   // the result is parked for lasso_result_wait, as after lasso_defer_next
  lasso_point unused[2];
  return lasso_bullet_round(c, c->ahead.bs, c->ahead.n, c->ahead.a_in, c->ahead.b_in, c->ahead.w_in, c->ahead.a_out, c->ahead.b_out, c->ahead.w_out, c->ahead.nk, u, u_inv, c->ahead.blinds, unused);
}
// ---- slab mode of the opening: the rank's share of the MSMs over its residue class of the generators (include/lasso_hip.h); bases = [G_{jl*world + rank}.., Q, H]
int32_t lasso_bases_has_direct(const lasso_bases* b) { return b ? 1 : 0; }
int32_t lasso_msm_dev_slab(lasso_ctx* c, const lasso_bases* b, const lasso_fr* sc, size_t n, uint32_t world, uint32_t rank, const lasso_fr* scale, const lasso_fr* tail, lasso_point* out) {
  REQ(c, b && sc && out && world >= 1 && rank < world && n % world == 0 && n / world + 2 <= b->pts.size());
  const size_t nl = n / world; Point acc = Point::identity();
  for (size_t jl = 0; jl < nl; jl++) { Fr v = F(sc)[jl * world + rank]; if (scale) v = v * *F(scale); acc += b->pts[jl] * v; }
  if (tail) { acc += b->pts[nl] * F(tail)[0]; acc += b->pts[nl + 1] * F(tail)[1]; }
  lasso_point tmp; put_point(acc, &tmp);
  return deliver_points(c, &tmp, 1, out);
}
int32_t lasso_bullet_round_slab(lasso_ctx* c, const lasso_bases* bs, size_t n, uint32_t world, uint32_t rank, const lasso_fr* a_in, const lasso_fr* b_in, const lasso_fr* w_in, lasso_fr* a_out,
                                lasso_fr* b_out, lasso_fr* w_out, size_t nk, const lasso_fr* u, const lasso_fr* u_inv, const lasso_fr* blinds, lasso_point* out) {
  REQ(c, bs && world >= 1 && rank < world && world <= n && n / world + 2 <= bs->pts.size() && nk >= 2 && nk <= n);
  const lasso_fr *a = a_in, *b = b_in, *w = w_in;
  std::vector<lasso_fr> ta, tb;
  if (u) {
    REQ(c, u_inv && a_out && b_out && w_out && 2 * nk <= n);
    ta.assign(a_in, a_in + 2 * nk); tb.assign(b_in, b_in + 2 * nk);
    lasso_bullet_fold(c, ta.data(), tb.data(), 2 * nk, w_in, n / (2 * nk), w_out, u, u_inv);
    memcpy(a_out, ta.data(), nk * sizeof(lasso_fr)); memcpy(b_out, tb.data(), nk * sizeof(lasso_fr));
    a = a_out; b = b_out; w = w_out;
  }
  lasso_fr cc[2]; lasso_inner_products_lr(c, a, b, nk, cc);
  const size_t nl = n / world, h = nk / 2;
  Point L = Point::identity(), R = Point::identity();
  for (size_t jl = 0; jl < nl; jl++) {   // generator j = blk*nk + pos of G^(k)_pos = sum_blk w_blk G_{blk*nk + pos}: L = <a_L, G_R>, R = <a_R, G_L>  (bullet.rs:84-118)
    const size_t j = jl * world + rank, blk = j / nk, pos = j % nk;
    if (pos >= h) L += bs->pts[jl] * (F(w)[blk] * F(a)[pos - h]); else R += bs->pts[jl] * (F(w)[blk] * F(a)[pos + h]);
  }
  if (rank == 0) { L += bs->pts[nl] * *F(&cc[0]); L += bs->pts[nl + 1] * F(blinds)[0]; R += bs->pts[nl] * *F(&cc[1]); R += bs->pts[nl + 1] * F(blinds)[1]; }
  lasso_point lr[2]; put_point(L, &lr[0]); put_point(R, &lr[1]);
  return deliver_points(c, lr, 2, out);
}
int32_t lasso_bullet_fold(lasso_ctx*, lasso_fr* a, lasso_fr* b, size_t nk, const lasso_fr* w, size_t nw, lasso_fr* w_out, const lasso_fr* u, const lasso_fr* u_inv) {
  size_t h = nk / 2; Fr uu = *F(u), ui = *F(u_inv);
  for (size_t i = 0; i < h; i++) { F(a)[i] = F(a)[i] * uu + ui * F(a)[h + i]; F(b)[i] = F(b)[i] * ui + uu * F(b)[h + i]; }   // bullet.rs:127-130
  for (size_t k = 0; k < nw; k++) { F(w_out)[2 * k] = F(w)[k] * ui; F(w_out)[2 * k + 1] = F(w)[k] * uu; }                   // bullet.rs:131 as weights
  return 0;
}

// ---- test helpers (not part of lasso_hip.h): compare projective points produced by two implementations
void mock_point_compress(const lasso_point* p, uint8_t* out32) { get_point(p).compress(out32); }
#ifdef ORC_BN254
int mock_point_on_curve(const lasso_point* p) {  // y^2 z = x^3 + 3 z^3 (the identity (0 : 1 : 0) included)
  Fq X = Fq::from_raw(p->x), Y = Fq::from_raw(p->y), Z = Fq::from_raw(p->z);
  if (X.is_zero() && Y.is_zero() && Z.is_zero()) return 0;
  return Y.square() * Z == X.square() * X + Fq::from_u64(3) * Z.square() * Z;
}
#else
int mock_point_on_curve(const lasso_point* p) {  // -x^2 + y^2 = 1 + d x^2 y^2 and T*Z = X*Y
  Point q = get_point(p);
  if (q.Z.is_zero()) return 0;
  Fq x, y; q.to_affine(x, y);
  bool on = (y.square() - x.square()) == (Fq::one() + EdConsts::d() * x.square() * y.square());
  return on && (q.T * q.Z == q.X * q.Y);
}
#endif
// n+1 generators from the reference's derivation (commitments.rs:22-44), as lasso_affine (Montgomery limbs); last = h
void mock_gens(const char* label, size_t n, lasso_affine* out) {
  MultiCommitGens g = MultiCommitGens::create(n, label);
  for (size_t i = 0; i <= n; i++) { const Point& p = i < n ? g.G[i] : g.h; Fq x, y; p.to_affine(x, y); memcpy(out[i].x, x.v, 32); memcpy(out[i].y, y.v, 32); }
}
}  // extern "C"

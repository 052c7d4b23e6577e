/* lasso_hip.h — C ABI of the MI355X (gfx950) device library `liblasso_hip.so`.
 *
 * This is the drop-in boundary for the hot path of a16z/Lasso's SparsePolynomialEvaluationProof
 * (SURVEY.md §8b): every O(n) loop of the reference prover is one entry point here; the host
 * (Rust in the reference, the C++ mirror in lasso_amd/host/ in this repo) keeps the protocol,
 * the Merlin transcript and all O(log n) scalar work.  Each entry point names the reference
 * code it replaces (paths relative to /root/reference).
 *
 * Conventions
 *  - lasso_fr     = ark-ff `Fp256<MontBackend<FrConfig,4>>` in-memory form: 4 x u64 little-endian limbs
 *                   holding a*2^256 mod p, p = 2^252 + 27742317777372353535851937790883648493
 *                   (`ark_curve25519::Fr`).  Passed through unchanged; device arrays of lasso_fr are
 *                   byte-identical to a Rust `Vec<Fr>`.
 *  - lasso_affine = ark-ec `twisted_edwards::Affine<EdwardsConfig>` {x, y}, Fq limbs in Montgomery form.
 *  - lasso_point  = ark-ec `twisted_edwards::Projective<EdwardsConfig>` {x, y, t, z}, Montgomery form; any
 *                   valid projective representative is returned (the transcript only ever sees the
 *                   compressed affine form, src/utils/transcript.rs:47-51).
 *  - The BN254 build of the library (liblasso_hip_bn254.so, the same source compiled with -DLASSO_BN254; BASELINE.json configs[1] names
 *    G = BN254) exports the SAME entry points with the same layouts over ark-bn254: lasso_fr = `ark_bn254::Fr` (a*2^256 mod r, r the
 *    254-bit order of G1), lasso_affine = `short_weierstrass::Affine<g1::Config>` {x, y} in Montgomery form (the host never passes the
 *    point at infinity: generators are not), lasso_point = a HOMOGENEOUS projective representative {x, y, t, z} = (X : Y : Z) with t unused
 *    (zero) and the identity (0 : 1 : 0) — ark's `Projective` for this model is Jacobian, so the binding reconstructs a `G1Projective` from
 *    the affine point (x/z, y/z) or, as the prover does, only ever serialises it.  Compressed rows use ark-ec's SWFlags encoding.
 *    A process may load both libraries (dlopen RTLD_LOCAL); a context belongs to the library that created it.
 *  - `d_` parameters are DEVICE pointers obtained from lasso_alloc (or any hipMalloc'd memory on the
 *    context's device, e.g. a torch tensor's data_ptr); all other pointers are HOST memory owned by the
 *    caller for the duration of the call only.
 *  - Every function returns 0 on success and a negative lasso_status otherwise; lasso_last_error()
 *    gives the message.  Nothing unwinds across the ABI.  The reference prover panics on contract
 *    violations (assert!); a host binding turns non-zero into panic!.
 *  - A context is entered from one thread at a time; it owns one HIP stream.  Functions that return
 *    results to host memory synchronise that stream before returning; all others are asynchronous.
 */
#ifndef LASSO_HIP_H
#define LASSO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lasso_ctx lasso_ctx;
typedef struct lasso_bases lasso_bases;
typedef struct { uint64_t l[4]; } lasso_fr;
typedef struct { uint64_t x[4], y[4]; } lasso_affine;
typedef struct { uint64_t x[4], y[4], t[4], z[4]; } lasso_point;

enum lasso_status { LASSO_OK = 0, LASSO_ERR_INVALID = -1, LASSO_ERR_OOM = -2, LASSO_ERR_HIP = -3, LASSO_ERR_UNSUPPORTED = -4 };

/* src/subtables/{and,or,xor,lt,range_check}.rs — the table plug-in (`SubtableStrategy`) as a runtime descriptor */
enum lasso_strategy_kind { LASSO_AND = 0, LASSO_OR = 1, LASSO_XOR = 2, LASSO_LT = 3, LASSO_RANGE = 4,
  /* NOT in the reference snapshot (src/subtables/mod.rs:22-26): BASELINE.json configs[4] names a "SparkSubtableStrategy"; upstream Lasso's history had one — subtable i =
   * EqPolynomial(tau_i).evals(), combine_lookups = the product of the C values, sumcheck degree C (to be confirmed against upstream).  Restated from that
   * description so that configs[4]'s shape runs under its own name; tau = C * log2(M) draws of F::rand from a fresh ark_std::test_rng() (the snapshot's trait has no
   * per-proof table parameter).  The tables hold field elements, so the integer shortcuts (lasso_materialize_subtable_u32, the *_u32 entry points) do not apply. */
  LASSO_SPARK_UNCONFIRMED = 5 };
typedef struct {
  int32_t kind;      /* lasso_strategy_kind */
  uint32_t c;        /* const generic C */
  uint32_t log_m;    /* log2 of const generic M */
  uint32_t log_r;    /* RangeCheckSubtableStrategy<LOG_R> only */
} lasso_strategy;

/* ---- context, memory ---------------------------------------------------------------------- */
int32_t lasso_ctx_create(int32_t device, lasso_ctx** out);
/* The same with a stream priority: background != 0 gives the context's stream the lowest priority of the device (bulk work issued beside a
 * latency-bound context, see lasso_amd/host/prover.hpp Dev::side), 0 the highest (what lasso_ctx_create does). */
int32_t lasso_ctx_create_background(int32_t device, int32_t background, lasso_ctx** out);
/* 16 bytes that identify the PHYSICAL device behind the context (hipDeviceGetUuid).  Slab mode uses it to learn whether two ranks of one proof share a GPU (P contexts of one
 * device: the one-box tests): launches that wait ON THE DEVICE for a challenge that depends on another rank's result must then not be used — streams of one device share its few
 * hardware queues, and a waiting gate would sit in front of the very kernel it waits for. */
int32_t lasso_ctx_device_uuid(lasso_ctx* ctx, uint8_t out[16]);
void lasso_ctx_destroy(lasso_ctx* ctx);
const char* lasso_last_error(lasso_ctx* ctx);          /* ctx may be NULL: last error of a failed create */
int32_t lasso_alloc(lasso_ctx* ctx, size_t bytes, void** d_out);
int32_t lasso_free(lasso_ctx* ctx, void* d_ptr);
/* Device memory held through this context: bytes currently allocated — lasso_alloc'd buffers, the context's own scratch and result buffers, and the generator
 * tables of every lasso_bases object created with it (window, digit-multiple and byte-multiple tables) — and the high-water mark of that figure since the
 * context was created or since the last call with reset_peak != 0.  What one rank of slab mode needs of its GPU's 288 GB (bench.py `peak_bytes_per_rank`);
 * the reference's counterpart is the resident set of its Vec<F>s (src/poly/dense_mlpoly.rs:28-32, src/subprotocols/grand_product.rs:19-58). */
int32_t lasso_mem_stats(lasso_ctx* ctx, uint64_t* live_bytes, uint64_t* peak_bytes, int32_t reset_peak);
/* The context's scratch buffer grows to the largest request it has served (densify's sort buffers, a commitment's digit arrays: 16 bytes per lookup and more) and is
 * kept; lasso_trim drains the stream and shrinks it back to its initial 4 MiB.  Not while a deferred result or a resident tail is outstanding (LASSO_ERR_INVALID). */
int32_t lasso_trim(lasso_ctx* ctx);
int32_t lasso_upload(lasso_ctx* ctx, void* d_dst, const void* src, size_t bytes);     /* synchronous */
int32_t lasso_download(lasso_ctx* ctx, void* dst, const void* d_src, size_t bytes);   /* synchronous */
int32_t lasso_copy(lasso_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);     /* DensePolynomial::clone / merge: src/poly/dense_mlpoly.rs:97-99,:251-261 */
int32_t lasso_zero(lasso_ctx* ctx, void* d_dst, size_t bytes);                        /* merge's zero padding :258 */
int32_t lasso_sync(lasso_ctx* ctx);
/* Error recovery after a call sequence was abandoned half way (the host prover threw between lasso_sumcheck_*_tail_begin and the last
 * lasso_sumcheck_cubic_tail_next, or left a deferred result uncollected): releases a resident kernel that is waiting for a challenge, drains the
 * stream and resets the protocol state, so that the context is usable again.  lasso_ctx_destroy calls it when needed. */
int32_t lasso_abort(lasso_ctx* ctx);
void* lasso_stream(lasso_ctx* ctx);                    /* the context's hipStream_t */

/* ---- per-kernel timing (HIP events on the context's stream), for bench.py's roofline ------- */
enum lasso_kernel_id { LASSO_K_BIND = 0, LASSO_K_CUBIC = 1, LASSO_K_COMBINE = 2, LASSO_K_EQ = 3, LASSO_K_GP = 4, LASSO_K_FINGERPRINT = 5,
                       LASSO_K_DOT = 6, LASSO_K_MATVEC = 7, LASSO_K_MSM = 8 /* bucket kernel: commitments, many rows */, LASSO_K_MISC = 9,
                       LASSO_K_MSM_DIRECT = 10 /* latency-shaped kernel: the openings' few rows of full-width scalars */, LASSO_K_COUNT = 11 };
#define LASSO_PROF_LARGE_ONLY 0x40000000   /* OR into the mask: bracket only launches of at least LASSO_PROF_LARGE_BYTES (a handful per proof: no measurable overhead) */
int32_t lasso_prof_enable(lasso_ctx* ctx, int32_t family_mask);   /* bit k = bracket launches of lasso_kernel_id k; 0 = off */
int32_t lasso_prof_reset(lasso_ctx* ctx);
/* launches, total milliseconds and algorithmic bytes (SURVEY.md §8d definitions) recorded for one kernel family */
int32_t lasso_prof_get(lasso_ctx* ctx, int32_t kernel_id, uint64_t* launches, double* total_ms, double* alg_bytes);
/* the same, restricted to launches with at least LASSO_PROF_LARGE_BYTES algorithmic bytes (past the 256 MiB Infinity Cache: the HBM-bound regime) */
#define LASSO_PROF_LARGE_BYTES 268435456.0
int32_t lasso_prof_get_large(lasso_ctx* ctx, int32_t kernel_id, uint64_t* launches, double* total_ms, double* alg_bytes);
/* family-specific work units recorded beside the bytes.  MSM families: group additions of the REFERENCE's algorithm for the same inputs (SURVEY.md §8d:
 * L*(R+1)*W bucket accumulation + L*W*2*2^c bucket reduction + L*(W-1)*(c+1) window combine, src/msm/mod.rs:91-164); 0 for the streaming families.
 * large_only bit 0: the launches lasso_prof_get_large counts (for LASSO_K_MSM: the row-parallel commitments, more than 16 rows); bit 1: the second counter
 * instead — MSM families: the mixed additions the kernel itself issues at most (one per scalar digit it looks at; zero digits are skipped); bit 2 (value 4): the
 * mixed additions the kernels EXECUTED, counted on the device (non-zero digits / bytes) — collected only while every launch is bracketed (a mask without
 * LASSO_PROF_LARGE_ONLY: the one untimed profiled step of bench.py), 0 otherwise.  This is the numerator of bench.py's roofline_msm fractions. */
int32_t lasso_prof_get_units(lasso_ctx* ctx, int32_t kernel_id, int32_t large_only, double* units);

/* Host-side latency accounting: number of device->host result hand-offs (flag waits) and the host time spent spinning on them since the last reset. */
int32_t lasso_wait_stats(lasso_ctx* ctx, uint64_t* waits, double* wait_us, int32_t reset);

/* ---- polynomial kernels -------------------------------------------------------------------- */
/* DensePolynomial::from_usize (src/poly/dense_mlpoly.rs:263-269): d_dst[i] = Fr::from(d_src[i]) */
int32_t lasso_fr_from_u32(lasso_ctx* ctx, const uint32_t* d_src, size_t n, lasso_fr* d_dst);
/* ... and its inverse for the values that are integers (capacity mode keeps dim / read_ts as 4-byte integers between the phases that need them as field elements):
 * d_dst[i] = d_src[i] as an integer; LASSO_ERR_INVALID if some value is >= 2^32; *max_out (may be NULL) = the largest value. */
int32_t lasso_fr_to_u32(lasso_ctx* ctx, const lasso_fr* d_src, size_t n, uint32_t* d_dst, uint32_t* max_out);
/* SubtableStrategy::to_lookup_polys (src/subtables/mod.rs:78-92): d_out[j] = d_table[d_idx[j]] */
int32_t lasso_gather(lasso_ctx* ctx, const lasso_fr* d_table, const uint32_t* d_idx, size_t n, lasso_fr* d_out);
/* EqPolynomial::evals (src/poly/eq_poly.rs:22-38): d_out[x] = prod_j (x_j ? r_j : 1-r_j), r[0] <-> top bit */
int32_t lasso_eq_evals(lasso_ctx* ctx, const lasso_fr* r, uint32_t ell, lasso_fr* d_out);
/* the same table multiplied by *scale (NULL = 1): in slab mode a rank's share of eq(r, .) is eq over the high variables times the eq factor of
 * the rank's low index bits */
int32_t lasso_eq_evals_scaled(lasso_ctx* ctx, const lasso_fr* r, uint32_t ell, const lasso_fr* scale, lasso_fr* d_out);
/* DensePolynomial::bound_poly_var_top (src/poly/dense_mlpoly.rs:209-216) on `npolys` polynomials of current
 * length n: Z[i] <- Z[i] + r*(Z[i+n/2] - Z[i]) for i < n/2, in place.  `d_polys` is a HOST array of device pointers. */
int32_t lasso_bind_top(lasso_ctx* ctx, lasso_fr* const* d_polys, uint32_t npolys, size_t n, const lasso_fr* r);
/* One round of SumcheckInstanceProof::prove_cubic_batched (src/subprotocols/sumcheck.rs:49-93): for each circuit c,
 * out[3c+0..3c+2] = sum_i comb(A,B,C) at x = 0, 2, 3 with comb = A*B*C (grand_product.rs:126-128).  n = current length. */
int32_t lasso_sumcheck_cubic_round(lasso_ctx* ctx, const lasso_fr* const* d_A, const lasso_fr* const* d_B, uint32_t ncirc,
                                   const lasso_fr* d_C, size_t n, lasso_fr* out);
/* The same round in EQ-WEIGHTED form — what the prover calls.  In prove_cubic_batched the third polynomial is always EqPolynomial(rand).evals()
 * (grand_product.rs:122-128); after binding its top j variables it equals  s_j * eq1(rand_j, x_top) * T_j  with T_j = eq(rand[j+1..]), and T_j is
 * the prefix of the ORIGINAL table d_E up to the scalar prod_{t<=j}(1 - rand_t).  So the device never binds or stores C: for the current length n
 * of A and B it returns  out[3c+{0,1,2}] = sum_{i < n/2} A_c(x)[i] * B_c(x)[i] * d_E[i]  at x = 0, 2, 3 (A_c(x) = the line through A_c[i], A_c[i+n/2]),
 * and the host multiplies by  s_j * eq1(rand_j, x) / prod_{t<=j}(1 - rand_t)  to obtain sumcheck.rs:56-93's evaluations — the same field elements.
 * d_E: device table with at least n/2 entries (the layer's eq table, or in the degenerate case rand_t = 1 a table proportional to T_j). */
int32_t lasso_sumcheck_cubic_eqw_round(lasso_ctx* ctx, const lasso_fr* const* d_A, const lasso_fr* const* d_B, uint32_t ncirc,
                                       const lasso_fr* d_E, size_t n, lasso_fr* out);
/* Rounds j >= 1 in one pass: first bind A and B with the previous challenge r (sumcheck.rs:116-120, in place, n = length BEFORE the bind, n >= 4),
 * then the eq-weighted sums of the next round on the bound values: out as lasso_sumcheck_cubic_eqw_round for length n/2 (sums over i < n/4). */
int32_t lasso_sumcheck_cubic_eqw_round_fused(lasso_ctx* ctx, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n,
                                             const lasso_fr* r, lasso_fr* out);
/* The eq-weighted round with TWO sums per circuit, split into launch and wait.  q_c(x) = sum_i A_c(x)[i] B_c(x)[i] d_E[i] is quadratic in x and the
 * round's claim already fixes q(0) + (a known multiple of) q(1) (sumcheck.rs:99-104 derives the evaluation at 1 from it the same way), so
 * (q_c(0), q_c,inf) with q_c,inf = sum_i (A_c[i+n/2] - A_c[i]) (B_c[i+n/2] - B_c[i]) d_E[i], the leading coefficient, determine the round
 * polynomial: one product per index and circuit less than the three-sum form, and the host's per-round inversion overlaps the kernel.
 * r == NULL: first round of a layer (length n, n >= 2, evaluation only); otherwise bind the previous challenge first (n = length before the
 * bind, n >= 4) exactly as lasso_sumcheck_cubic_eqw_round_fused.  Returns after the launch; lasso_result_wait(ctx, out, 2*ncirc) then yields
 * out[2c] = q_c(0), out[2c+1] = q_c,inf.  No other call may be made on the context between the two.
 * Memory form of the bound arrays: between rounds only kernels read them, so this call leaves them LAZILY REDUCED — each element is some 256-bit
 * representative (< 2^254 + 2^130) of the right residue, not necessarily the canonical one.  Every entry point of this library accepts such
 * input; lasso_bind_top (the last bind of a layer), the resident tail and lasso_sumcheck_cubic_eqw_round_fused write canonical elements again, and
 * nothing lazily reduced ever reaches the host (the heads are read after the last bind). */
int32_t lasso_sumcheck_cubic_eqw2_begin(lasso_ctx* ctx, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n,
                                        const lasso_fr* r);
int32_t lasso_result_wait(lasso_ctx* ctx, lasso_fr* out, size_t count);
/* The FIRST round of a layer (r == NULL above) with the layer's eq table built inside the same launch.  Every layer of BatchedGrandProductArgument::prove starts with
 * poly_C = EqPolynomial(rand).evals() (grand_product.rs:122) — here  E = *scale * EqPolynomial(point[0..ell)).evals(),  2^ell = n/2 entries, scale == NULL: 1 — and as kernels of
 * their own those tables are 40 launch-bound steps on the proof's critical path.  Afterwards d_E_out holds the table (byte-identical to lasso_eq_evals_scaled's) for the later
 * rounds of the layer and the pending result is lasso_sumcheck_cubic_eqw2_begin's.  Tables of 2^7 .. 2^32 entries (up to 2^14 the two factor tables are built in LDS by every workgroup and the table is written on the way; above, the call enqueues what lasso_eq_evals_scaled enqueues — k_eq_small2, k_eq_outer — in front of the round, or with LASSO_EQ_INLINE_BIG=1 forms the entries inside the round from the factor tables: measured, not the default); LASSO_ERR_UNSUPPORTED otherwise (build the table, call the plain form). */
int32_t lasso_sumcheck_cubic_eqw2_begin_eq(lasso_ctx* ctx, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, lasso_fr* d_E_out, size_t n, const lasso_fr* point, uint32_t ell,
                                           const lasso_fr* scale);
/* Launch/wait split for any call whose result comes back through the mapped result buffer (sumcheck rounds, MSMs of up to 16 rows,
 * lasso_bullet_round, lasso_read_heads ...): after lasso_defer_next the NEXT such call returns right after its launch, ignoring its `out`
 * argument; lasso_result_wait(ctx, out, count) then delivers the values (count in field-element units; a lasso_point is 4).  The prover uses it
 * to absorb the opening's a-vector into the transcript (dot_product.rs:196) while the first bullet round runs.  If the next call turns out not to
 * use the mapped buffer it completes synchronously and lasso_result_wait fails with LASSO_ERR_INVALID (the flag is cleared by either). */
int32_t lasso_defer_next(lasso_ctx* ctx);
/* The tail of prove_cubic_batched (sumcheck.rs:49-133) in one resident kernel: once a layer is down to q <= lasso_sumcheck_tail_capacity() indices per circuit, the remaining
 * log2(2q) rounds and the final bind are served without a launch per round — the host posts each challenge into a host-mapped mailbox, the kernel
 * answers through the mapped result buffer (2.5 us per turn against 6.5 us for a launch, tools/pingpong_bench.hip).
 *   begin: r == NULL: first round of a layer, arrays of length n = 2q; otherwise bind r first (n = 4q, as lasso_sumcheck_cubic_eqw2_begin).
 *          Afterwards the first round's sums are pending: lasso_result_wait(ctx, out, 2*ncirc) -> out[2c] = q_c(0), out[2c+1] = q_c,inf.
 *   next:  posts the round's challenge; pending afterwards: the next round's sums, or — after log2(2q) calls — the bound heads
 *          out[c] = A_c[0], out[ncirc + c] = B_c[0] (the final claims, sumcheck.rs:126-133).
 * The arrays in device memory are left as they were (stale): nothing reads a layer's arrays after its sumcheck.  No other call may be made on the
 * context until the heads have been collected.  A host that stops answering cannot hang the device: every wait in the kernel gives up after 5 s. */
int32_t lasso_sumcheck_cubic_tail_begin(lasso_ctx* ctx, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n,
                                        const lasso_fr* r);
/* lasso_sumcheck_cubic_tail_begin(.., r = NULL) for a layer that fits the resident kernel from its first round on: no table at all — the kernel derives
 * E = *scale * EqPolynomial(point[0..ell)).evals(), 2^ell = n/2 <= capacity, from the point (two factor tables of <= 32 entries in LDS, one product per use). */
int32_t lasso_sumcheck_cubic_tail_begin_eq(lasso_ctx* ctx, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, size_t n, const lasso_fr* point, uint32_t ell, const lasso_fr* scale);
/* largest q the two *_tail_begin calls accept (512: one workgroup per circuit holds its arrays in 147 KB of the CU's LDS) */
/* The resident tail stops EARLY and hands its arrays to the host (round 5): the next lasso_sumcheck_cubic_tail_begin* ends when the arrays are down to m_stop elements each
 * (a power of two, 2 <= m_stop <= 128, smaller than the arrays at the tail's first round) and its LAST pending result is the arrays instead of the heads:
 * 2 * ncirc * m_stop values, A_0[0..m_stop), A_1[..], .., B_0[..], ..  A resident turn costs ~7 us of round trip whatever the size; the last log2(m_stop) rounds of a layer are a
 * few dozen field products, which the host does in less (prover.hpp cubic_rounds: sumcheck.rs:49-124 on the handed-over arrays — the same field elements).  m_stop <= 1: the heads. */
int32_t lasso_tail_handover_next(lasso_ctx* ctx, uint32_t m_stop);
/* Sumcheck rounds LAUNCHED AHEAD of their challenge (round 5; lasso_bullet_round_ahead is the same device for the openings).  Between two launched rounds the host turn was ~12 us of
 * which ~1.5 us Fiat-Shamir: the rest is the launch and its dispatch.  So round j + 1 is enqueued behind round j BEFORE round j's sums are back:
 *     lasso_sumcheck_cubic_eqw2_begin_ahead(A, B, ncirc, d_E, n)   — the round lasso_sumcheck_cubic_eqw2_begin(.., r) would run, minus r; legal while a result is pending
 *     lasso_result_wait(..)                                          — round j's sums; transcript; challenge r_j
 *     lasso_challenge_post(&r_j)                                     — the waiting kernel proceeds; its sums are now the pending result
 * The kernel (k_cubic_eqw_fused<.., AHEAD>) waits on the device: workgroup 0 polls the host-mapped mailbox, republishes the scalar in device memory, all workgroups go on.  It ends
 * WITHOUT a result on lasso_abort's poison tag or after 5 s without a post (then lasso_result_wait fails: nothing hangs).  Between begin_ahead and the post ONLY lasso_result_wait
 * is legal on the context: entry points that would synchronise the stream (lasso_free / upload / download / sync / trim, buffer growth) return LASSO_ERR_INVALID instead of blocking
 * behind the waiting kernel.  Streaming two-sum rounds only (n / 4 > 64); LASSO_ERR_UNSUPPORTED otherwise.  lasso_rounds_ahead_ok: 1 unless LASSO_ROUNDS_AHEAD=0.
 * lasso_sumcheck_cubic_tail_begin_ahead: the resident tail enqueued ahead of the challenge it binds first (n = 4q); the first lasso_sumcheck_cubic_tail_next posts it. */
int32_t lasso_rounds_ahead_ok(lasso_ctx* ctx);
int32_t lasso_sumcheck_cubic_eqw2_begin_ahead(lasso_ctx* ctx, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n);
int32_t lasso_challenge_post(lasso_ctx* ctx, const lasso_fr* r);
/* lasso_sumcheck_linear_eqw_round_fused (in place) enqueued ahead of its challenge: lasso_challenge_post, then lasso_result_wait delivers its 3 * alpha values */
int32_t lasso_sumcheck_linear_eqw_round_fused_ahead(lasso_ctx* ctx, lasso_fr* const* d_polys, uint32_t alpha, const lasso_fr* d_E, size_t n);
int32_t lasso_sumcheck_cubic_tail_begin_ahead(lasso_ctx* ctx, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n);
/* A grand-product LAYER enqueued ahead of its eq point (round 5).  Between two layers the device idled for the host's last rounds, the layer's closing Fiat-Shamir step AND the
 * launch + dispatch of the next layer's first kernel(s).  The next layer's first launch — lasso_sumcheck_cubic_eqw2_begin_eq's round 0, or lasso_sumcheck_cubic_tail_begin_eq's
 * resident kernel (after lasso_tail_handover_next, as there) — may be enqueued while the CURRENT layer's resident tail is still active and a result is pending:
 *     lasso_sumcheck_cubic_eqw2_begin_eq_ahead(A, B, ncirc, d_E_out, n, ell)  /  lasso_sumcheck_cubic_tail_begin_eq_ahead(A, B, ncirc, n, ell)
 *     ... the current layer's remaining lasso_sumcheck_cubic_tail_next / lasso_result_wait turns, the host's rounds, the layer's closing transcript step ...
 *     lasso_point_post(point, ell, scale)    — the context is now where the plain entry point would have left it (first sums pending; the tail active)
 *  or lasso_point_cancel()                   — the enqueued kernels end without a result and without touching anything
 * A one-wave gate kernel (k_gate_point) in front of the launch waits for the post (host-mapped area, one self-validating three-chunk entry per field element, all under the launch's
 * sequence number) and leaves the point in device memory for the kernels behind it; lasso_abort's poison tag and a 5 s wall-clock bound end the wait as everywhere else.  Nothing may
 * grow while a resident kernel is active: LASSO_ERR_UNSUPPORTED when a buffer would have to, or when LASSO_LAYER_AHEAD=0 (lasso_layer_ahead_ok: 0) — the caller then starts the layer
 * the plain way.  Until the post / cancel, entry points that synchronise the stream return LASSO_ERR_INVALID (as for rounds launched ahead). */
int32_t lasso_layer_ahead_ok(lasso_ctx* ctx);
int32_t lasso_sumcheck_cubic_eqw2_begin_eq_ahead(lasso_ctx* ctx, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, lasso_fr* d_E_out, size_t n, uint32_t ell);
int32_t lasso_sumcheck_cubic_tail_begin_eq_ahead(lasso_ctx* ctx, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, size_t n, uint32_t ell);
int32_t lasso_point_post(lasso_ctx* ctx, const lasso_fr* point, uint32_t ell, const lasso_fr* scale);
int32_t lasso_point_cancel(lasso_ctx* ctx);
uint32_t lasso_sumcheck_tail_capacity(void);
int32_t lasso_sumcheck_cubic_tail_next(lasso_ctx* ctx, const lasso_fr* r);
/* The resident tail of the primary sumcheck for the linear strategies (the rounds lasso_sumcheck_linear_eqw_round[_fused] serve one launch at a time):
 * begin as above on the alpha polynomials d_src (only read); per round 2*alpha values out[2k] = S0_k, out[2k+1] = S1_k; challenges through
 * lasso_sumcheck_cubic_tail_next; after log2(2q) of them the pending result is the alpha heads out[k] = E_k(r_z) (surge.rs:175-176). */
int32_t lasso_sumcheck_linear_tail_begin(lasso_ctx* ctx, const lasso_fr* const* d_src, uint32_t alpha, const lasso_fr* d_E, size_t n, const lasso_fr* r);
/* One round of SumcheckInstanceProof::prove_arbitrary (src/subprotocols/sumcheck.rs:165-237) with
 * comb_func = S::combine_lookups_eq (src/subtables/mod.rs:53-57): out[x] = sum_i g(E_1..E_alpha)(x) * eq(x), x = 0..degree.
 * d_polys holds alpha = NUM_MEMORIES device pointers; d_eq is the eq polynomial. */
int32_t lasso_sumcheck_combine_round(lasso_ctx* ctx, const lasso_strategy* s, const lasso_fr* const* d_polys, const lasso_fr* d_eq,
                                     size_t n, uint32_t degree, lasso_fr* out);
/* LT strategy, the prover's form of the degree-(C+1) round (lt.rs:62-71 inside sumcheck.rs:179-218).  In Horner form  g = LT_0 + EQ_0 (LT_1 + EQ_1 (... + EQ_{C-2} LT_{C-1}))  a round
 * costs ONE field product per memory and evaluation point instead of two, provided the factor 2^5 that a product of two in-memory (Montgomery 2^256) operands loses against the kernels'
 * 2^261 radix is carried by the data: lasso_lt_prescale multiplies polynomial 2m (= LT_m, length n) by 32^-(C-1-m), once, before the first round — binding is linear, so the
 * arrays stay scaled through every lasso_bind_top — and lasso_sumcheck_combine_round_lt_scaled takes such arrays and returns exactly what lasso_sumcheck_combine_round returns on the
 * unscaled ones.  After the last bind the caller multiplies the head of polynomial 2m by 32^(C-1-m) to obtain LT_m(r); the EQ polynomials (odd indices) are never touched.
 * (lasso_sumcheck_combine_round on an LT strategy scales private copies itself: the literal call.) */
/* d_src == NULL: in place.  Otherwise d_polys[i] <- (scaled) d_src[i] for ALL 2C polynomials (the ones the scaling leaves alone are copied): the sumcheck's clone of the lookup
 * polynomials (surge.rs:151) and the scaling in one pass over the data. */
int32_t lasso_lt_prescale(lasso_ctx* ctx, const lasso_strategy* s, const lasso_fr* const* d_src, lasso_fr* const* d_polys, size_t n);
int32_t lasso_sumcheck_combine_round_lt_scaled(lasso_ctx* ctx, const lasso_strategy* s, const lasso_fr* const* d_polys, const lasso_fr* d_eq, size_t n, uint32_t degree, lasso_fr* out);
/* The FIRST round of that sumcheck (before any bind) from the lookup polynomials' INTEGER values, E_k[i] = F::from(d_u32[k][i]) with EVERY ENTRY 0 OR 1 — the LT and EQ subtables
 * hold bits (lt.rs:17-44).  All lines are then small integers at the evaluation points and the Horner walk is exact 128-bit integer arithmetic; same out[x] as
 * lasso_sumcheck_combine_round on the lifted arrays.  Round 0 is half of the sumcheck's work.
 * Enforced, not assumed: the kernel checks every entry it reads and the call returns LASSO_ERR_INVALID when one exceeds 1 (out is then meaningless); C <= 16 (with bits the
 * walk's intermediate values stay below 17 (17^16 - 1) / 16 < 2^67, inside the three 29-bit limbs the eq weighting multiplies by; a larger C is refused). */
int32_t lasso_sumcheck_combine_round_lt_u32(lasso_ctx* ctx, const lasso_strategy* s, const uint32_t* const* d_u32, const lasso_fr* d_eq, size_t n, uint32_t degree, lasso_fr* out);
/* The same round in EQ-WEIGHTED form for the LINEAR strategies (AND / OR / XOR / RangeCheck: g = sum_k w_k E_k, src/subtables/and.rs:45-53) — what the
 * prover calls.  The eq polynomial is factored exactly as in lasso_sumcheck_cubic_eqw_round (prefix of the original table d_E + host scalars), and by
 * linearity of g a round needs per polynomial only  out[3k] = sum_{i<n/2} E_k[i] d_E[i]  and  out[3k+1] = sum_{i<n/2} E_k[i+n/2] d_E[i]  (out[3k+2] unused):
 * the host forms G(x) = sum_k w_k (out[3k] + x (out[3k+1] - out[3k])) and e(x) = s_j eq1(r_j, x) / prod_{t<=j}(1 - r_t) G(x), x = 0, 1, 2. */
int32_t lasso_sumcheck_linear_eqw_round(lasso_ctx* ctx, const lasso_fr* const* d_polys, uint32_t alpha, const lasso_fr* d_E, size_t n, lasso_fr* out);
/* bind the alpha polynomials with the previous challenge r (in place, n = length before the bind, n >= 4), then the sums of the next round (length n/2) */
int32_t lasso_sumcheck_linear_eqw_round_fused(lasso_ctx* ctx, lasso_fr* const* d_polys, uint32_t alpha, const lasso_fr* d_E, size_t n, const lasso_fr* r, lasso_fr* out);
/* The same with separate source arrays: reads d_src[k] (length n), writes the bound arrays (length n/2) to d_polys[k]; d_src[k] == d_polys[k] is the
 * in-place call above.  The prover's first bind takes the lookup polynomials E_k themselves as source, so surge.rs:151's clones are never made. */
int32_t lasso_sumcheck_linear_eqw_round_fused_from(lasso_ctx* ctx, const lasso_fr* const* d_src, lasso_fr* const* d_polys, uint32_t alpha, const lasso_fr* d_E,
                                                   size_t n, const lasso_fr* r, lasso_fr* out);
/* The first round and the first bind of the same sumcheck from the lookup polynomials' INTEGER values: E_k[i] = F::from(d_u32[k][i]) (E_k = T[dim_k] holds subtable entries, which the
 * prover has as 32-bit integers anyway: lasso_gather_u32).  Same results as the two calls above on the 32-byte field form — lasso_sumcheck_linear_eqw_round resp.
 * lasso_sumcheck_linear_eqw_round_fused_from with d_src = that form — for an eighth of the bytes read per element. */
int32_t lasso_sumcheck_linear_eqw_round_u32(lasso_ctx* ctx, const uint32_t* const* d_u32, uint32_t alpha, const lasso_fr* d_E, size_t n, lasso_fr* out);
int32_t lasso_sumcheck_linear_eqw_round_fused_from_u32(lasso_ctx* ctx, const uint32_t* const* d_u32, lasso_fr* const* d_polys, uint32_t alpha, const lasso_fr* d_E, size_t n,
                                                       const lasso_fr* r, lasso_fr* out);
/* Subtables::compute_sumcheck_claim (src/subtables/mod.rs:187-216): out = sum_k eq[k] * g(E_1[k],...,E_alpha[k]) */
int32_t lasso_combine_claim(lasso_ctx* ctx, const lasso_strategy* s, const lasso_fr* const* d_polys, const lasso_fr* d_eq, size_t n, lasso_fr* out);
/* compute_dotproduct for k polynomials against one weight vector (src/utils/mod.rs:64-73 via DensePolynomial::evaluate
 * src/poly/dense_mlpoly.rs:229-235): out[p] = sum_i d_polys[p][i] * d_w[i] */
int32_t lasso_multi_dot(lasso_ctx* ctx, const lasso_fr* const* d_polys, uint32_t k, const lasso_fr* d_w, size_t n, lasso_fr* out);
/* out[p] = d_polys[p][0] for k polynomials — the final claims a sumcheck hands back after the last bind
 * (src/subprotocols/sumcheck.rs:126-132 `poly_A_vec_par[i][0]`, :257 `poly[0]`) in one transfer. */
int32_t lasso_read_heads(lasso_ctx* ctx, const lasso_fr* const* d_polys, uint32_t k, lasso_fr* out);
/* out[i * count + j] = d_polys[i][j], j < count: short runs of k arrays (k * count <= 16384) through the mapped result buffer — the tops of the product trees, whose
 * layers of up to 32 elements the host proves without the device (lasso_amd/host/prover.hpp host_layers) */
int32_t lasso_read_runs(lasso_ctx* ctx, const lasso_fr* const* d_polys, uint32_t k, uint32_t count, lasso_fr* out);
/* GrandProductCircuit::new (src/subprotocols/grand_product.rs:38-58).  d_tree holds 2n-2 elements: layer 0 (the n
 * inputs, left half | right half) at [0,n) must be filled by the caller; layer k (n/2^k elements) follows layer k-1
 * and is computed here as layer_k[i] = left_{k-1}[i] * right_{k-1}[i]. */
int32_t lasso_gp_build(lasso_ctx* ctx, lasso_fr* d_tree, size_t n);
/* GrandProducts::build_grand_product_inputs, read/write sets (src/lasso/memory_checking.rs:277-302):
 * d_read_out[i]  = d_read[i]*gamma^2 + d_table[d_dim[i]]*gamma + d_dim[i] - tau,  d_write_out[i] = d_read_out[i] + gamma^2 */
int32_t lasso_fingerprint_ops(lasso_ctx* ctx, const lasso_fr* d_table, const uint32_t* d_dim, const lasso_fr* d_read, size_t s,
                              const lasso_fr* gamma, const lasso_fr* tau, lasso_fr* d_read_out, lasso_fr* d_write_out);
/* lasso_fingerprint_ops followed by lasso_gp_build of both trees, as ONE call: the first product layer (grand_product.rs:20-36) is taken while the leaves are still
 * in registers, so they are written once and not read back.  d_tree_read / d_tree_write: 2s-element tree arenas (leaves, then the layers); s >= 4, a power of two.
 * Bit-identical to the three separate calls. */
int32_t lasso_fingerprint_ops_gp(lasso_ctx* ctx, const lasso_fr* d_table, const uint32_t* d_dim, const lasso_fr* d_read, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                 lasso_fr* d_tree_read, lasso_fr* d_tree_write);
/* Capacity mode (one proof larger than the reference's "every layer a Vec<F>" layout allows, grand_product.rs:38-58): the same two trees WITHOUT their leaf layers —
 * d_upper_read / d_upper_write receive the layers of s/2, s/4, .., 2 elements back to back (s - 2 elements; what lasso_fingerprint_ops_gp leaves at d_tree + s); the
 * fingerprints exist only inside the launch.  Half the resident bytes of the read / write trees. */
int32_t lasso_fingerprint_ops_gp_upper(lasso_ctx* ctx, const lasso_fr* d_table, const uint32_t* d_dim, const lasso_fr* d_read, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                       lasso_fr* d_upper_read, lasso_fr* d_upper_write);
/* the same, the read timestamps given as 32-bit integers (identical bytes out) */
int32_t lasso_fingerprint_ops_gp_upper_u32(lasso_ctx* ctx, const lasso_fr* d_table, const uint32_t* d_dim, const uint32_t* d_read_u32, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                           lasso_fr* d_upper_read, lasso_fr* d_upper_write);
/* ... and the leaves recomputed for ONE strip set of the bottom layer, where its two streaming sumcheck rounds need them.  The layer's arrays are A = leaves[0 .. s/2),
 * B = leaves[s/2 .. s).  A round on the index range [i0, i0 + cs) reads `nstrips` strips of each array, stride = s / 2 / nstrips apart (nstrips = 2: the layer's first round;
 * 4: the second round, which binds); d_out_read / d_out_write (2 * nstrips * cs elements each) receive [A strips.., B strips..]:
 *   out[(arr * nstrips + t) * cs + i] = leaf[arr * s/2 + t * stride + i0 + i],   arr in {0 (A), 1 (B)}, t < nstrips, i < cs,
 * i.e. the arrays lasso_sumcheck_cubic_eqw2_begin takes as (A = out, B = out + nstrips * cs, n = nstrips * cs) with the eq table offset by i0.  Same bytes as
 * lasso_fingerprint_ops at those positions.  i0 + cs <= stride. */
int32_t lasso_fingerprint_ops_strips(lasso_ctx* ctx, const lasso_fr* d_table, const uint32_t* d_dim, const lasso_fr* d_read, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                     uint32_t nstrips, size_t i0, size_t cs, lasso_fr* d_out_read, lasso_fr* d_out_write);
int32_t lasso_fingerprint_ops_strips_u32(lasso_ctx* ctx, const lasso_fr* d_table, const uint32_t* d_dim, const uint32_t* d_read_u32, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                         uint32_t nstrips, size_t i0, size_t cs, lasso_fr* d_out_read, lasso_fr* d_out_write);
/* init/final sets (memory_checking.rs:254-273): d_init_out[i] = d_table[i]*gamma + i - tau, d_final_out[i] = d_init_out[i] + d_final[i]*gamma^2 */
int32_t lasso_fingerprint_mem(lasso_ctx* ctx, const lasso_fr* d_table, const lasso_fr* d_final, size_t m,
                              const lasso_fr* gamma, const lasso_fr* tau, lasso_fr* d_init_out, lasso_fr* d_final_out);
/* Slab mode of the same (one proof sharded over `world` GPUs by low index bits): local index i stands for address i*world + rank; d_table is the whole
 * subtable, d_final and the outputs hold the rank's m local entries. */
int32_t lasso_fingerprint_mem_slab(lasso_ctx* ctx, const lasso_fr* d_table, const lasso_fr* d_final, size_t m, uint32_t world, uint32_t rank,
                                   const lasso_fr* gamma, const lasso_fr* tau, lasso_fr* d_init_out, lasso_fr* d_final_out);
/* DensePolynomial::bound (src/poly/dense_mlpoly.rs:184-207): out[i] = sum_j L[j] * d_Z[j*r_size + i], i < r_size */
int32_t lasso_matvec_left(lasso_ctx* ctx, const lasso_fr* d_Z, const lasso_fr* L, size_t l_size, size_t r_size, lasso_fr* out);

/* lasso_matvec_left with L already on the device and the result left on the device (asynchronous): the opening keeps L*Z resident */
int32_t lasso_matvec_left_dev(lasso_ctx* ctx, const lasso_fr* d_Z, const lasso_fr* d_L, size_t l_size, size_t r_size, lasso_fr* d_out);
/* CanonicalSerialize of n device field elements: out[32*i..] = the canonical integer, little endian (what ProofTranscript::append_scalar absorbs,
 * src/utils/transcript.rs:33-45) */
int32_t lasso_fr_to_bytes(lasso_ctx* ctx, const lasso_fr* d_src, size_t n, uint8_t* out);

/* ---- densify ------------------------------------------------------------------------------- */
/* DensifiedRepresentation::from_lookup_indices for ONE dimension (src/lasso/densified.rs:32-57; serial in the reference, TODO(#29)).
 * d_indices: the reference's `Vec<[usize; C]>` uploaded as is (n_lookups x C u64, row-major).  For k < s (s = n_lookups.next_power_of_two()):
 * access[k] = indices[k][dim] (0 for the padded tail), read_ts[k] = #{j < k : access[j] == access[k]}, final_ts[a] = #{k : access[k] == a}.
 * Outputs: d_dim_u32[s] (dim_usize), d_dim[s], d_read[s], d_final[2^log_m] as Fr (DensePolynomial::from_usize).  Fails with
 * LASSO_ERR_INVALID if an index is >= 2^log_m (the reference panics on the out-of-bounds `final_timestamps[memory_address]`). Synchronous. */
int32_t lasso_densify_dim(lasso_ctx* ctx, const uint64_t* d_indices, size_t n_lookups, size_t C, size_t dim, size_t s, uint32_t log_m,
                          uint32_t* d_dim_u32, lasso_fr* d_dim, lasso_fr* d_read, lasso_fr* d_final);

/* Slab mode: every rank sorts the whole access sequence (timestamps are a property of the whole sequence) but writes only its residue class:
 * global index k (address a) with k mod world == rank (a mod world == rank) lands at local index k / world (a / world).  Outputs have s/world
 * (2^log_m / world) entries. */
int32_t lasso_densify_dim_slab(lasso_ctx* ctx, const uint64_t* d_indices, size_t n_lookups, size_t C, size_t dim, size_t s, uint32_t log_m, uint32_t world, uint32_t rank,
                               uint32_t* d_dim_u32, lasso_fr* d_dim, lasso_fr* d_read, lasso_fr* d_final);

/* ---- curve kernels (Hyrax commitment, src/poly/commitments.rs + src/msm/mod.rs) ------------- */
/* Upload a generator vector once (MultiCommitGens: G[0..n) then any extra points such as gens_1.G[0] and h) and
 * precompute the per-window multiples used by both MSM entry points. */
int32_t lasso_bases_create(lasso_ctx* ctx, const lasso_affine* points, size_t n, lasso_bases** out);
/* Tables a generator set holds (bytes per generator): the window table 16^w G (64 x 112 = 7 KB: the commitments' bucket MSM), the signed digit multiples m 16^w G, m = 1..8
 * (57 KB: the openings' latency-shaped MSMs), byte multiples of the low windows on demand (the commitments of small scalars), and — for sets of at most 2^14 + 64 generators —
 * the signed BYTE multiples m 256^w G, m = 1..128 (459 KB: the same openings with half the additions per scalar).  byte_multiples = 0 leaves the last ones out (a caller
 * that wants the bytes more than the time: slab mode's unused full-width set, capacity mode); lasso_bases_create = byte_multiples 1.  LASSO_MSM_DIRECT8=0 in the
 * environment leaves them out everywhere. */
int32_t lasso_bases_create_opt(lasso_ctx* ctx, const lasso_affine* points, size_t n, int32_t byte_multiples, lasso_bases** out);
/* Device memory per bases object: the window table 64 * n * 112 B; for n <= 2^17 the digit-multiple table of the latency-shaped MSMs, 8x that (57 KB per generator); and, built by the
 * first commitment of small scalars that uses the object (serialised by a mutex inside the object; ~3 ms, waited for), one byte-multiple table of 255 * n * 112 B per byte window
 * (117 MB for n = 4096; at most two windows). */
void lasso_bases_destroy(lasso_ctx* ctx, lasso_bases* b);
/* Build the byte-multiple tables of byte windows 0 .. byte_windows-1 (1 or 2) NOW instead of inside the first commitment of small scalars that uses the object, so that a caller
 * timing DensifiedRepresentation::commit (src/benches/bench.rs:54-66) does not time a one-off table build (3 x 10 ms at n = 4096).  A failed allocation is not an error: that
 * commitment then takes the bucket kernel.  Not while a launch of this context waits for the host. */
int32_t lasso_bases_prepare(lasso_ctx* ctx, const lasso_bases* bases, uint32_t byte_windows);
/* DensePolynomial::commit / commit_inner with zero blinds (src/poly/dense_mlpoly.rs:109-181): for each of l_size rows,
 * out[row] = sum_{j < r_size} d_Z[row*r_size + j] * bases[j]   (Commitments::batch_commit, src/poly/commitments.rs:84-93) */
int32_t lasso_hyrax_commit(lasso_ctx* ctx, const lasso_fr* d_Z, size_t l_size, size_t r_size, const lasso_bases* bases, lasso_point* out);
/* the same with the rows returned in wire form: out32[32*row..] = serialize_compressed(normalised row commitment) — what the transcript absorbs
 * (src/poly/dense_mlpoly.rs:281-289 -> utils/transcript.rs:47-51) and what the proof stores; normalisation (one inversion per row) runs on the device */
int32_t lasso_hyrax_commit_compressed(lasso_ctx* ctx, const lasso_fr* d_Z, size_t l_size, size_t r_size, const lasso_bases* bases, uint8_t* out32);
/* The same commitment for a polynomial whose canonical values the caller already holds as 32-bit integers, Z[i] = F::from(d_u32[i]) — the lookup
 * polynomials E_i = T[dim_i] of a small-valued subtable (subtables/mod.rs:116-129), dim, the timestamps.  Skips the pass that converts the
 * 32-byte elements to integers and its max-bit readback (msm/mod.rs:95-106 finds the same bound by scanning).  max_value >= every d_u32[i]. */
int32_t lasso_hyrax_commit_compressed_u32(lasso_ctx* ctx, const uint32_t* d_u32, uint32_t max_value, size_t l_size, size_t r_size,
                                          const lasso_bases* bases, uint8_t* out32);

/* ---- slab mode (ONE proof over the P GPUs of a node): the device-side exchange of the partial row commitments ---------------
 * Each rank holds the columns = rank (mod P) of every Hyrax row and commits to them (lasso_hyrax_commit_rows_dev leaves the L partial row sums on the
 * device, lasso_point_row_bytes() bytes each, in the kernels' own point form); the partials are all-gathered with RCCL on the context's stream
 * (ncclAllGather of raw bytes over xGMI) and every rank adds the P partials of each row and compresses (lasso_points_reduce_compress).  No reference
 * counterpart: the reference is single-process (its rows are rayon tasks, src/poly/dense_mlpoly.rs:118-127).
 * librccl is loaded on first use (dlopen); LASSO_ERR_UNSUPPORTED when it is absent.  The unique id travels between the ranks by the caller's means
 * (the host prover broadcasts it through its shared-memory segment). */
int32_t lasso_rccl_available(void);                                                               /* 1 = librccl loads and has every entry point used; no communicator is touched.  Exchange it between the ranks
                                                                                                     BEFORE lasso_rccl_init: ncclCommInitRank blocks until every rank has entered it */
int32_t lasso_rccl_unique_id(uint8_t out[128]);                                                   /* ncclGetUniqueId, rank 0 */
int32_t lasso_rccl_init(lasso_ctx* ctx, int32_t rank, int32_t world, const uint8_t id[128]);      /* ncclCommInitRank on the context's device; collective */
int32_t lasso_rccl_shutdown(lasso_ctx* ctx);                                                      /* ncclCommDestroy (also done by lasso_ctx_destroy) */
int32_t lasso_rccl_ready(lasso_ctx* ctx);                                                         /* world size of the communicator, 0 = none */
int32_t lasso_rccl_selftest(lasso_ctx* ctx);                                                      /* collective: all-gather 1 KB of a rank-dependent pattern and check every slot; bounded wait (20 s), the
                                                                                                     communicator is aborted and dropped on a time-out — the caller then falls back to its host exchange */
int32_t lasso_rccl_allgather(lasso_ctx* ctx, const void* d_send, void* d_recv, size_t bytes);     /* d_recv[g*bytes..) <- rank g's d_send; asynchronous on the stream */
size_t lasso_point_row_bytes(void);
int32_t lasso_hyrax_commit_rows_dev(lasso_ctx* ctx, const lasso_fr* d_Z, size_t l_size, size_t r_size, const lasso_bases* bases, void* d_rows);
int32_t lasso_points_reduce_compress(lasso_ctx* ctx, const void* d_parts, uint32_t groups, size_t rows, uint8_t* out32);
/* Subtable `sub` of the strategy as 32-bit integers, 2^log_m entries (SubtableStrategy::materialize_subtables: and.rs:16-28, or.rs, xor.rs,
 * lt.rs:17-44 (0 = LT, 1 = EQ), range_check.rs:19-51 (0 = full, 1 = remainder below 2^(LOG_R mod log_m), 2 = zeros)), written by the device. */
int32_t lasso_materialize_subtable_u32(lasso_ctx* ctx, const lasso_strategy* strategy, uint32_t sub, uint32_t* d_out);
/* d_out[i] = d_table[d_idx[i]] on 32-bit integers (the integer twin of lasso_gather, feeding lasso_hyrax_commit_compressed_u32) */
int32_t lasso_gather_u32(lasso_ctx* ctx, const uint32_t* d_table, const uint32_t* d_idx, size_t n, uint32_t* d_out);
/* VariableBaseMSM::msm (src/msm/mod.rs:36-40): out = sum_{j < n} scalars[j] * bases[j]; n <= number of bases.  Zero scalars cost nothing. */
int32_t lasso_msm(lasso_ctx* ctx, const lasso_bases* bases, const lasso_fr* scalars, size_t n, lasso_point* out);

/* Same as lasso_msm with the scalars already resident on the device (Montgomery form). */
int32_t lasso_msm_dev(lasso_ctx* ctx, const lasso_bases* bases, const lasso_fr* d_scalars, size_t n, lasso_point* out);

/* out = sum_{j<n} (scale * d_scalars[j]) * bases[j] + tail[0]*bases[n] + tail[1]*bases[n+1] — e.g. delta = d*g_hat + r_delta*h of
 * DotProductProofLog::prove (src/subprotocols/dot_product.rs:219-224) in one MSM over the resident fold weights (bases = [G.., Q, h]) */
int32_t lasso_msm_dev_scaled(lasso_ctx* ctx, const lasso_bases* bases, const lasso_fr* d_scalars, size_t n, const lasso_fr* scale, const lasso_fr* tail, lasso_point* out);

/* ---- Hyrax opening tail: BulletReductionProof::prove (src/subprotocols/bullet.rs:40-154) with the vectors resident on
 * the device.  State kept by the caller: d_a, d_b (current length nk, folded in place), d_w (n/nk tensor weights, see below).
 * The generator vector is never folded: after k rounds G^(k)_i = sum_blk w_blk * G_{blk*nk + i}, so every L / R is one MSM
 * over the original (precomputed) generators — the same group elements as the reference's fold-then-MSM. */
/* out[0] = c_L = <a_L, b_R>, out[1] = c_R = <a_R, b_L>   (bullet.rs:79-80), halves of the current length nk */
int32_t lasso_inner_products_lr(lasso_ctx* ctx, const lasso_fr* d_a, const lasso_fr* d_b, size_t nk, lasso_fr* out);
/* bullet.rs:84-118: out[0] = L = <a_L, G_R> + c_L*Q + blind_L*H, out[1] = R = <a_R, G_L> + c_R*Q + blind_R*H, where G is the
 * current (virtually folded) generator vector, `bases` holds [G_0..G_{n-1}, Q, H] and tail = {c_L, blind_L, c_R, blind_R}. */
int32_t lasso_bullet_lr(lasso_ctx* ctx, const lasso_bases* bases, size_t n, const lasso_fr* d_a, size_t nk, const lasso_fr* d_w, const lasso_fr* tail, lasso_point* out);
/* One whole round of BulletReductionProof::prove (bullet.rs:66-132) in a single call — the form the host prover uses.
 * If u != NULL the previous round's fold is applied on the way in: (d_a_in, d_b_in) of length 2*nk and the n/(2*nk) weights d_w_in are
 * read, (d_a_out, d_b_out) of length nk and the n/nk weights d_w_out are written (ping-pong buffers, must not alias the inputs):
 *   a'[i] = a_L[i]*u + u_inv*a_R[i], b'[i] = b_L[i]*u_inv + u*b_R[i], w'[2k] = w[k]*u_inv, w'[2k+1] = w[k]*u   (:127-132).
 * If u == NULL the inputs (length nk, n/nk weights) are the state itself and the *_out pointers are ignored.
 * Then, for the state of length nk: c_L = <a_L, b_R>, c_R = <a_R, b_L> (:79-80) stay on the device and
 *   out[0] = L = <a_L, G_R> + c_L*Q + blinds[0]*H,  out[1] = R = <a_R, G_L> + c_R*Q + blinds[1]*H   (:84-118)
 * with G the virtually folded generators as in lasso_bullet_lr. */
int32_t lasso_bullet_round(lasso_ctx* ctx, const lasso_bases* bases, size_t n, const lasso_fr* d_a_in, const lasso_fr* d_b_in, const lasso_fr* d_w_in,
                           lasso_fr* d_a_out, lasso_fr* d_b_out, lasso_fr* d_w_out, size_t nk, const lasso_fr* u, const lasso_fr* u_inv, const lasso_fr* blinds, lasso_point* out);
/* The same folding round LAUNCHED AHEAD of its challenge.  A bullet round is on the proof's critical path and the host turn between two rounds is ~4 us of work inside ~31 us of
 * launch and completion latency; so the host enqueues round k+1 behind round k BEFORE it has round k's L and R — the kernel starts the moment round k ends and waits for the
 * challenge in a host-mapped mailbox (the resident sumcheck tails' protocol) — and posts u, u^-1 when it has drawn them:
 *   lasso_bullet_round_ahead(..., nk, blinds)   enqueue (arguments as lasso_bullet_round's folding form, without u / u_inv / out); returns at once
 *   lasso_bullet_post(ctx, u, u_inv)            release it: from here on it is a deferred result
 *   lasso_result_wait(ctx, (lasso_fr*)LR, 8)    L and R
 * Between the first and the second call no other launch of this context may be made except... none: only host work and lasso_result_wait of the PREVIOUS round.
 * lasso_abort releases a round that never got its challenge (the kernel also leaves by itself after 5 s).  lasso_bullet_ahead_ok: 1 if the form is available for this generator
 * set now (its digit-multiple table exists, the fused launch is not switched off, LASSO_BULLET_AHEAD != 0, the opening MSMs are not being profiled launch by launch). */
int32_t lasso_bullet_ahead_ok(lasso_ctx* ctx, const lasso_bases* bases);
int32_t lasso_bullet_round_ahead(lasso_ctx* ctx, const lasso_bases* bases, size_t n, const lasso_fr* d_a_in, const lasso_fr* d_b_in, const lasso_fr* d_w_in,
                                 lasso_fr* d_a_out, lasso_fr* d_b_out, lasso_fr* d_w_out, size_t nk, const lasso_fr* blinds);
int32_t lasso_bullet_post(lasso_ctx* ctx, const lasso_fr* u, const lasso_fr* u_inv);
/* The END of an opening enqueued ahead of its last challenge (round 6): what dot_product.rs:198-231 does after the last folding round — the last fold of a, b (two elements each)
 * and of the generators' weights (bullet.rs:127-132), x_hat = a[0], a_hat = b[0], and delta = d * g_hat + r_delta * h (dot_product.rs:219-224: one MSM over the folded weights) —
 * as ONE chain in the stream: gate, fold, MSM (which also publishes the two heads).  Enqueue it right after lasso_bullet_post of the last round (its result may still be pending),
 * collect that round's L, R, draw u, then
 *   lasso_bullet_post(ctx, u, u_inv)               release the chain
 *   lasso_result_wait(ctx, out, 6)                 out[0..4) = delta as a lasso_point, out[4] = d_a[0], out[5] = d_b[0]
 * d_a, d_b: the two-element state after the last round (folded in place); d_w: n / 2 weights, d_w_out: n; scale = d; tail = {0, r_delta} (the Q and h terms).
 * Same group element and field elements as lasso_bullet_fold + lasso_read_heads + lasso_msm_dev_scaled.  lasso_bullet_tail_ahead_ok: lasso_bullet_ahead_ok, tagged results on,
 * LASSO_BULLET_TAIL_AHEAD != 0.  LASSO_ERR_UNSUPPORTED when a buffer would have to grow (first proof of a context: the plain calls size them). */
int32_t lasso_bullet_tail_ahead_ok(lasso_ctx* ctx, const lasso_bases* bases);
int32_t lasso_bullet_tail_ahead(lasso_ctx* ctx, const lasso_bases* bases, size_t n, lasso_fr* d_a, lasso_fr* d_b, const lasso_fr* d_w, size_t nw, lasso_fr* d_w_out,
                                const lasso_fr* scale, const lasso_fr* tail);
/* bullet.rs:127-132: a[i] <- a_L[i]*u + u_inv*a_R[i], b[i] <- b_L[i]*u_inv + u*b_R[i] for i < nk/2 (in place), and the
 * generator fold G[i] <- G_L[i]*u_inv + G_R[i]*u recorded as weights: d_w_out[2*blk] = d_w[blk]*u_inv, d_w_out[2*blk+1] = d_w[blk]*u. */
int32_t lasso_bullet_fold(lasso_ctx* ctx, lasso_fr* d_a, lasso_fr* d_b, size_t nk, const lasso_fr* d_w, size_t nw, lasso_fr* d_w_out, const lasso_fr* u, const lasso_fr* u_inv);


/* ---- slab mode of the opening (ONE proof over P GPUs): the generator vector is split by residue class like every other array.
 * `bases` = lasso_bases_create over [G_{rank}, G_{P+rank}, ..., G_{n-P+rank}, Q, H] — this rank's n/P generators, then the two extra points.  a, b and the fold weights are
 * sqrt(N)-sized and stay replicated (every rank folds them in full, identically), but the MSMs — the dependent chain of the opening, 12-14 rounds of it — are shared:
 * each rank adds up only its generators' terms and the host sums the P partial points of a result (the transcript sees the compressed sum: same bytes as one GPU).
 * No reference counterpart (the reference is single-process: bullet.rs:84-132 folds G and runs its MSMs serially). */
/* 1 if `bases` carries the digit-multiple table the two calls below need (built for generator sets up to 2^17 when memory allows); the ranks agree on it before using them */
int32_t lasso_bases_has_direct(const lasso_bases* bases);
/* lasso_bullet_round with out[0], out[1] = this rank's PARTIAL L and R: the terms of the generators j = rank (mod world); rank 0 alone adds c_L*Q + blinds[0]*H resp.
 * c_R*Q + blinds[1]*H.  The state (d_a_*, d_b_*, d_w_*: whole vectors, replicated) is read and written exactly as lasso_bullet_round does. */
int32_t lasso_bullet_round_slab(lasso_ctx* ctx, const lasso_bases* bases, size_t n, uint32_t world, uint32_t rank, const lasso_fr* d_a_in, const lasso_fr* d_b_in, const lasso_fr* d_w_in,
                                lasso_fr* d_a_out, lasso_fr* d_b_out, lasso_fr* d_w_out, size_t nk, const lasso_fr* u, const lasso_fr* u_inv, const lasso_fr* blinds, lasso_point* out);
/* out = sum_{jl < n/world} (scale *) d_scalars[jl*world + rank] * bases[jl]  (+ tail[0]*bases[n/world] + tail[1]*bases[n/world + 1]): this rank's share of an MSM over a whole,
 * replicated scalar vector of length n — Cx = <x, G> (dot_product.rs:183-186) and delta = d*g_hat + r_delta*h (:219-224; rank 0 passes the tail, the others NULL).
 * scale == NULL: 1.  Returns through the mapped result buffer (lasso_defer_next applies). */
int32_t lasso_msm_dev_slab(lasso_ctx* ctx, const lasso_bases* bases, const lasso_fr* d_scalars, size_t n, uint32_t world, uint32_t rank, const lasso_fr* scale, const lasso_fr* tail,
                           lasso_point* out);

#ifdef __cplusplus
}
#endif
#endif

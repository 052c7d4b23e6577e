/* lasso_prover.h — C ABI of `liblasso_prover.so`, the C++ mirror of the reference's Rust host for the north-star path.
 *
 * In the reference the host is Rust and these are methods (src/lasso/densified.rs, src/lasso/surge.rs); with no Rust
 * toolchain in this image the same surface is offered to Python (tests/, bench.py) through this header:
 *   lasso_host_densify  = DensifiedRepresentation::<F,C>::from_lookup_indices(&indices, log_m)      densified.rs:22
 *   lasso_host_gens_from_points = the `gens: &SparsePolyCommitmentGens<G>` argument of commit / prove    surge.rs:119-125 (the caller's points)
 *   lasso_host_gens_new = SparsePolyCommitmentGens::<G>::new(label, c, s, num_memories, log_m)      surge.rs:32 (convenience)
 *   lasso_host_commit   = DensifiedRepresentation::commit(&gens)                                     densified.rs:78
 *   lasso_host_prove    = SparsePolynomialEvaluationProof::<G,C,M,S>::prove(&mut dense, &r, &gens,
 *                             &mut Transcript::new(transcript_label), &mut RandomTape::new(tape_label))   surge.rs:119
 * Proofs and commitments are returned in ark-serialize's compressed wire format (CanonicalSerialize), the artefact the
 * unmodified Rust `verify` (surge.rs:214) would consume.  All device work goes through include/lasso_hip.h.
 * Every function returns 0 on success, negative on error (message: lasso_host_last_error); -2 = output buffer too small
 * (the needed size is written to *len).
 */
#ifndef LASSO_PROVER_H
#define LASSO_PROVER_H
#include <stddef.h>
#include <stdint.h>
#include "lasso_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct lasso_host lasso_host;
typedef struct lasso_host_gens lasso_host_gens;
typedef struct lasso_host_dense lasso_host_dense;

const char* lasso_host_last_error(void);
int32_t lasso_host_create(int32_t device, lasso_host** out);
void lasso_host_destroy(lasso_host* h);
lasso_ctx* lasso_host_ctx(lasso_host* h);   /* the device context, e.g. for lasso_prof_* */
/* Device memory of this host (its main and side contexts, include/lasso_hip.h lasso_mem_stats): bytes held now, their high-water mark, and the high-water mark of
 * what the prover itself had in use (buffers parked in the host's recycling pool excluded) since creation / the last call with reset != 0.  Any pointer may be NULL. */
int32_t lasso_host_mem_stats(lasso_host* h, uint64_t* live_bytes, uint64_t* peak_bytes, uint64_t* prover_peak_bytes, int32_t reset);
/* Capacity mode (slab mode's purpose: proofs whose polynomials do not fit one GPU — the reference keeps every DensePolynomial and every product-tree layer as a
 * Vec<F>, src/subprotocols/grand_product.rs:38-58): the prover keeps the read / write product trees without their leaf layers (half of each tree) and recomputes the
 * fingerprints strip by strip for the two streaming rounds of the bottom layer that read them; and a representation densified while the mode is on keeps dim_i / read_i
 * (integers: addresses and access counts, densified.rs:32-57) as 4-byte integers instead of field elements — whoever needs them as field elements (a row block's
 * commitment in slab mode, their evaluation, the opening's L*Z) lifts one polynomial at a time into a scratch array.  Same proof bytes, less resident memory, more time.
 * Off by default (LASSO_CAPACITY=1 turns it on for every host). */
int32_t lasso_host_set_capacity(lasso_host* h, int32_t on);
/* Throughput mode: this host is one of several proving concurrently on the same GPU (own context, stream and thread each).  The one latency device that costs throughput is
 * then switched off — the openings' folding rounds launched ahead of their challenge (lasso_bullet_round_ahead): a kernel that waits on the device for its host occupies its
 * compute units while other proofs' kernels could run (measured: -5 % at 16 concurrent proofs, +1.5 % for a single one).  Same proof bytes.  Default: off. */
int32_t lasso_host_set_throughput_mode(lasso_host* h, int32_t on);
/* what a densified representation holds on the device: bytes, and whether dim / read are in the compact form */
int32_t lasso_host_dense_info(lasso_host_dense* d, uint64_t* device_bytes, int32_t* compact);

/* Slab mode: ONE proof sharded over `world` GPUs (world a power of two, one lasso_host per rank, every rank given the SAME lookups and point).
 * Every polynomial is split by low index bits (rank g holds the indices = g mod world), the transcript is replicated, and `allgather` is the only
 * collective: it must gather `bytes` bytes from every rank into recv (rank order) and return 0 (any transport the embedder has; the gloo test uses
 * lasso_amd/parallel.py's).  Prefer lasso_host_set_comm_shm below on one node: no callback per round, and the partial row commitments go over RCCL.
 * Must be called before lasso_host_gens_new / lasso_host_densify.  All ranks return the same commitment and proof bytes (the bytes a single GPU
 * produces).  world = 1 restores the single-GPU path. */
typedef int32_t (*lasso_host_allgather_fn)(void* user, const void* send, void* recv, size_t bytes);
int32_t lasso_host_set_comm(lasso_host* h, int32_t rank, int32_t world, lasso_host_allgather_fn allgather, void* user);
/* The same with the library's own intra-node exchange instead of a callback: the P ranks of one node (one process per GPU) meet in the POSIX
 * shared-memory segment `name` ("/..." — identical on every rank) and all-gather their per-round partial sums through it (lasso_amd/host/shm_comm.hpp):
 * about a microsecond per exchange, nothing of the embedding language in the loop.  When every rank can join one RCCL communicator (one GPU per rank; rank 0's
 * ncclUniqueId travels through the segment and the ranks agree on the outcome), the bulk exchange — the partial row commitments of the Hyrax matrices — runs as
 * ncclAllGather on the context's stream with the per-row sums on the device (include/lasso_hip.h "slab mode"); otherwise it too goes through the segment.
 * LASSO_SLAB_RCCL=0 keeps RCCL out.  Must precede gens_new / densify, like lasso_host_set_comm. */
int32_t lasso_host_set_comm_shm(lasso_host* h, int32_t rank, int32_t world, const char* name);

/* The CALLER's generators: SparsePolynomialEvaluationProof::prove and DensifiedRepresentation::commit take `gens: &SparsePolyCommitmentGens<G>` (src/lasso/surge.rs:119-125,
 * src/lasso/densified.rs:78-81) — points the caller already holds, however it came by them.  This is that argument: nothing is derived inside the library.
 * Each of the three sets is one PolyCommitmentGens (src/poly/dense_mlpoly.rs:34-45 -> DotProductProofGens, src/subprotocols/dot_product.rs:139-150 -> two MultiCommitGens,
 * src/poly/commitments.rs:15-19) passed as n + 2 affine points in the order
 *     gens.gens_n.G[0], ..., gens.gens_n.G[n-1],   gens.gens_1.G[0],   gens.gens_n.h          (gens_1.h is the same point by construction: split_at, commitments.rs:54-71)
 * with n = 2^(num_vars - num_vars/2) for the set's polynomial size (surge.rs:39-47: l_variate = log2(next_pow2(2 c s)), log_m_variate = log2(next_pow2(c)) + log_m,
 * derefs = log2(next_pow2(num_memories s))); a count other than n + 2 is refused (num_memories = 0: the strategy is not known yet — `commit` reads only the first two sets —
 * and the derefs set is only required to be a power of two + 2; a wrong size then fails at use, as batch_commit's assert does, commitments.rs:85).  Coordinates are ark-ff's in-memory form (4 x u64 Montgomery limbs of x then y: what
 * CurveGroup::normalize_batch yields, commitments.rs:87); points must be finite and on the curve (as the reference, the library does not check).  The points are copied:
 * the caller's buffers are not retained.  The Rust shim fills the three arrays from the caller's `&SparsePolyCommitmentGens<G>` (integration/rust/hip.rs HipGens::from_gens). */
int32_t lasso_host_gens_from_points(lasso_host* h, size_t c, size_t s, size_t num_memories, size_t log_m,
                                    const lasso_affine* l_variate, size_t n_l_variate, const lasso_affine* log_m_variate, size_t n_log_m_variate,
                                    const lasso_affine* derefs, size_t n_derefs, lasso_host_gens** out);
/* SparsePolyCommitmentGens::new(label, c, s, num_memories, log_m) (surge.rs:32-58) as a CONVENIENCE for callers with no generators of their own (C, Python, bench.py):
 * the library's restatement of MultiCommitGens::new (SHAKE256(label || compressed generator) -> ChaCha20Rng -> G::rand, commitments.rs:22-44).  A Rust caller should not
 * rely on it reproducing arkworks' stream: it passes its own points through lasso_host_gens_from_points. */
int32_t lasso_host_gens_new(lasso_host* h, const char* label, size_t c, size_t s, size_t num_memories, size_t log_m, lasso_host_gens** out);
/* the points of set `which` (0 = gens_combined_l_variate, 1 = gens_combined_log_m_variate, 2 = gens_derefs) in lasso_host_gens_from_points' layout; *count = n + 2
 * (returns -2 with *count set when cap is smaller) */
int32_t lasso_host_gens_points(lasso_host_gens* g, int32_t which, lasso_affine* out, size_t cap, size_t* count);
/* Build every device table a proof over these generators reads NOW (lasso_bases_prepare on the three sets): the byte-multiple tables of the small-scalar commitments are
 * otherwise built by the first lasso_host_commit / lasso_host_prove that meets the object — inside whatever span the caller times.  The Rust harness calls it before its
 * `DensifiedRepresentation.commit` span (integration/rust/bench_types.rs, src/benches/bench.rs:54-66); bench.py reports its time as `gens_tables_s`. */
int32_t lasso_host_gens_prepare(lasso_host_gens* g);
void lasso_host_gens_free(lasso_host_gens* g);
/* indices: n_lookups x c, row-major (Vec<[usize; C]>) */
int32_t lasso_host_densify(lasso_host* h, const uint64_t* indices, size_t n_lookups, size_t c, size_t log_m, lasso_host_dense** out);
void lasso_host_dense_free(lasso_host_dense* d);
/* out = [u64 L1][L1 x 32 B][u64 L2][L2 x 32 B]: l_variate_polys_commitment.C then log_m_variate_polys_commitment.C */
int32_t lasso_host_commit(lasso_host_dense* d, lasso_host_gens* g, uint8_t* out, size_t cap, size_t* len);
/* LIVENESS.  While a proof runs, kernels of this context wait ON THE DEVICE for the calling thread's next Fiat-Shamir challenge (resident tails, gate kernels
 * in front of rounds / layers / bullet rounds launched ahead, include/lasso_hip.h).  Every such wait ends by itself after 5 s of wall clock (or at once on lasso_abort's poison
 * tag), so a host that stops answering can never hang the GPU — but the bound cuts both ways: if the thread inside lasso_host_prove* is descheduled or stopped (debugger,
 * SIGSTOP, a starved container) for more than 5 s in the middle of a proof, the waiting kernel leaves WITHOUT a result, the next lasso_result_wait fails ("flag not raised"),
 * and lasso_host_prove* returns an error after restoring the context (lasso_abort: both contexts usable again, nothing leaked).  The proof is NOT retried inside the library: with
 * the caller's live transcript (lasso_host_prove_cb) a replay would absorb into a transcript that has already advanced.  A caller that owns fresh transcripts (this entry
 * point, the bench harness) simply calls again; the proof bytes are deterministic.  The prover thread should not share its core with work that can starve it for seconds. */
int32_t lasso_host_prove(lasso_host* h, lasso_host_dense* d, lasso_host_gens* g, const lasso_strategy* strategy, const lasso_fr* r, size_t r_len,
                         const char* transcript_label, const char* tape_label, uint8_t* out, size_t cap, size_t* len);

/* ---- the caller's LIVE transcript and random tape --------------------------------------------------------------------------------------------------
 * SparsePolynomialEvaluationProof::prove takes `&mut merlin::Transcript` and `&mut RandomTape<G>` (src/lasso/surge.rs:119-125): a caller whose transcript already holds
 * state — anything but the bench harness, whose transcripts are fresh — needs the proof bound to THAT state.  Everything src/utils/transcript.rs:20-72 does to a
 * merlin::Transcript is one of two operations, so the transcript crosses the ABI as two callbacks; `user` is the caller's object (the Rust shim passes the
 * `&mut Transcript` itself, integration/rust/hip.rs).  RandomTape<G> is a merlin::Transcript too (src/utils/random.rs:9-39: `tape`), initialised by the caller's
 * RandomTape::new; it crosses the same way.  Labels are NOT NUL-terminated (pointer + length) and live as long as the library is loaded (string literals: the Rust
 * side may treat them as &'static [u8], which merlin's signatures ask for).  Callbacks are called from the thread that called lasso_host_prove_cb, never concurrently. */
typedef struct {
  /* merlin::Transcript::append_message(label, message); append_u64(label, x) arrives as its 8 little-endian bytes (what merlin absorbs) */
  void (*append_message)(void* user, const uint8_t* label, size_t label_len, const uint8_t* message, size_t message_len);
  /* merlin::Transcript::challenge_bytes(label, dest) */
  void (*challenge_bytes)(void* user, const uint8_t* label, size_t label_len, uint8_t* dest, size_t dest_len);
} lasso_transcript_vtbl;
/* lasso_host_prove with the caller's live transcript and tape.  lasso_host_prove(.., "example", "proof", ..) is this call with the library's own Merlin objects
 * (below) behind the callbacks: same bytes (tests/test_transcript_callbacks_cpu.py). */
int32_t lasso_host_prove_cb(lasso_host* h, lasso_host_dense* d, lasso_host_gens* g, const lasso_strategy* strategy, const lasso_fr* r, size_t r_len,
                            const lasso_transcript_vtbl* transcript, void* transcript_user, const lasso_transcript_vtbl* tape, void* tape_user,
                            uint8_t* out, size_t cap, size_t* len);
/* lasso_host_verify (below) against the caller's live transcript (surge.rs:214-220 takes `&mut Transcript` as well) */
int32_t lasso_host_verify_cb(lasso_host* h, lasso_host_gens* g, const lasso_strategy* strategy, size_t s, const lasso_fr* r, size_t r_len,
                             const lasso_transcript_vtbl* transcript, void* transcript_user,
                             const uint8_t* proof, size_t proof_len, const uint8_t* commitment, size_t commitment_len, int32_t* ok);
/* The library's own Merlin (STROBE-128 / Keccak-f[1600], merlin 3.0's framing) as one implementation of that interface — for callers without a merlin of their own
 * (C, Python) that need a transcript living across several calls.  lasso_host_merlin_new(label) = Transcript::new(label);
 * lasso_host_random_tape_new(name) = RandomTape::new(name) (init_randomness = F::rand(&mut test_rng()) absorbed, src/utils/random.rs:15-31).
 * Pass the object as `user` with lasso_host_merlin_vtbl(). */
typedef struct lasso_merlin lasso_merlin;
lasso_merlin* lasso_host_merlin_new(const char* label);
lasso_merlin* lasso_host_random_tape_new(const char* name);
void lasso_host_merlin_free(lasso_merlin* m);
const lasso_transcript_vtbl* lasso_host_merlin_vtbl(void);

/* the bench harness's inputs (src/benches/bench.rs:13-34): one `next_u64() % memory_size` per lookup from ark_std::test_rng(),
 * and log2(s) field elements from a fresh test_rng() */
void lasso_host_gen_indices(size_t sparsity, size_t memory_size, uint64_t* out);
void lasso_host_gen_random_point(size_t bits, lasso_fr* out);
/* SparsePolynomialEvaluationProof::verify(&commitment, &r, &gens, &mut transcript)  (src/lasso/surge.rs:214-271), over the proof's wire format
 * (ark-serialize CanonicalSerialize, compressed — the bytes lasso_host_prove returns) and the commitment's (lasso_host_commit's layout).  The lookups
 * are not needed: s = the (padded) number of lookups the commitment was made for.  The openings' two MSMs per proof run on the device; the rest is host
 * arithmetic (lasso_amd/host/verifier.hpp).  Returns 0 and sets *ok to 1 (Ok(())) or 0 (Err(ProofVerifyError)); a proof that does not deserialize, or
 * that trips one of the reference's assert!s on shapes, returns -1 with the reason in lasso_host_last_error(). */
int32_t lasso_host_verify(lasso_host* h, lasso_host_gens* g, const lasso_strategy* strategy, size_t s, const lasso_fr* r, size_t r_len, const char* transcript_label,
                          const uint8_t* proof, size_t proof_len, const uint8_t* commitment, size_t commitment_len, int32_t* ok);
/* Test support (not part of the reference's surface): prove_cubic_batched (sumcheck.rs:27-135, C = EqPolynomial(rand).evals()) on caller-supplied
 * arrays with a scripted eq point, for the degenerate points (rand_t = 0 or 1) no transcript produces.  A, B: k contiguous arrays of 2^ell elements.
 * out = 3 compressed coefficients per round, the ell challenges, the k final claims of A, the k of B (32-byte canonical scalars). */
int32_t lasso_host_debug_cubic_batched(lasso_host* h, lasso_host_dense* dense, lasso_host_gens* gens, const lasso_strategy* strategy, size_t k, size_t ell,
                                       const lasso_fr* A, const lasso_fr* B, const lasso_fr* rand, const lasso_fr* coeffs, const lasso_fr* claim,
                                       const char* transcript_label, uint8_t* out, size_t cap, size_t* len);

#ifdef __cplusplus
}
#endif
#endif

/* The drop-in boundary used from plain C — no Python, no C++, no torch: what a cgo / Rust-FFI / N-API binding of the reference's harness calls
 * (include/lasso_prover.h; the reference's three calls are src/benches/bench.rs:54-66).
 *
 *   cc -std=c99 -Iinclude examples/prove_c_abi.c -o prove_c_abi -Llasso_amd -llasso_prover -Wl,-rpath,$PWD/lasso_amd
 *   ./prove_c_abi [log2 lookups, default 10]
 *
 * densify -> commit -> prove (fresh transcript by label) -> prove again through the callback interface with the library's own Merlin objects (must be the same bytes) ->
 * verify.  Exit code 0 = the proof verified and the two proving paths agree.  tests/test_c_abi_example_cpu.py builds it against the test mock of the device library
 * (no GPU needed), which also checks that both headers are valid C. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lasso_prover.h"

#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, lasso_host_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
  const size_t log_s = argc > 1 ? (size_t)atoi(argv[1]) : 10, s = (size_t)1 << log_s, log_m = 16, c = 1;
  lasso_strategy strategy; strategy.kind = LASSO_AND; strategy.c = (uint32_t)c; strategy.log_m = (uint32_t)log_m; strategy.log_r = 0;

  lasso_host* h = NULL; lasso_host_gens* gens = NULL; lasso_host_dense* dense = NULL;
  CHECK(lasso_host_create(0, &h));
  /* the bench harness's inputs: gen_indices / gen_random_point from a fresh test_rng (benches/bench.rs:13-34) */
  uint64_t* idx = (uint64_t*)malloc(s * c * sizeof(uint64_t));
  lasso_fr* r = (lasso_fr*)malloc(log_s * sizeof(lasso_fr));
  lasso_host_gen_indices(s, (size_t)1 << log_m, idx);
  lasso_host_gen_random_point(log_s, r);

  CHECK(lasso_host_gens_new(h, "gens_sparse_poly", c, s, c, log_m, &gens));            /* SparsePolyCommitmentGens::new   surge.rs:32 */
  CHECK(lasso_host_densify(h, idx, s, c, log_m, &dense));                              /* from_lookup_indices            densified.rs:22 */
  size_t cap = (size_t)1 << 22, comm_len = 0, proof_len = 0, proof2_len = 0;
  uint8_t* comm = (uint8_t*)malloc(cap); uint8_t* proof = (uint8_t*)malloc(cap); uint8_t* proof2 = (uint8_t*)malloc(cap);
  CHECK(lasso_host_commit(dense, gens, comm, cap, &comm_len));                         /* commit                         densified.rs:78 */
  CHECK(lasso_host_prove(h, dense, gens, &strategy, r, log_s, "example", "proof", proof, cap, &proof_len));   /* prove   surge.rs:119 */

  /* the same through the caller-owned transcript interface (surge.rs:119-125 takes &mut Transcript, &mut RandomTape) */
  lasso_merlin* t = lasso_host_merlin_new("example"); lasso_merlin* tape = lasso_host_random_tape_new("proof");
  CHECK(lasso_host_prove_cb(h, dense, gens, &strategy, r, log_s, lasso_host_merlin_vtbl(), t, lasso_host_merlin_vtbl(), tape, proof2, cap, &proof2_len));
  lasso_host_merlin_free(t); lasso_host_merlin_free(tape);
  const int same = proof_len == proof2_len && memcmp(proof, proof2, proof_len) == 0;

  int32_t ok = 0;
  CHECK(lasso_host_verify(h, gens, &strategy, s, r, log_s, "example", proof, proof_len, comm, comm_len, &ok));   /* verify  surge.rs:214 */
  uint64_t live = 0, peak = 0, used = 0;
  CHECK(lasso_host_mem_stats(h, &live, &peak, &used, 0));
  printf("2^%zu AND lookups: commitment %zu bytes, proof %zu bytes, verified %d, callback path identical %d, device peak %llu bytes\n", log_s, comm_len, proof_len, (int)ok, same,
         (unsigned long long)peak);

  lasso_host_dense_free(dense); lasso_host_gens_free(gens); lasso_host_destroy(h);
  free(idx); free(r); free(comm); free(proof); free(proof2);
  return ok == 1 && same ? 0 : 2;
}

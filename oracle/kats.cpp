// ORACLE — TEST INFRASTRUCTURE ONLY (see lasso_oracle.hpp header).
// The reference's own unit-test known answers, restated against the oracle.
// Each orc_kat_* returns 0 on success, a positive failure code otherwise; tests/test_oracle_kats.py
// runs them all.  Citations name the reference test each one mirrors.
#include "lasso_oracle.hpp"
using namespace orc;

static Fr F(u64 x) { return Fr::from_u64(x); }
static std::vector<Fr> bitvec(size_t value, size_t bits) {  // utils/mod.rs:33-46 index_to_field_bitvector
  std::vector<Fr> v; for (size_t i = bits; i-- > 0;) v.push_back(((value >> i) & 1) ? Fr::one() : Fr::zero()); return v;
}
#define CHECK(cond, code) do { if (!(cond)) return code; } while (0)
#define KAT(name) extern "C" int orc_kat_##name() { try {
#define KAT_END return 0; } catch (const std::exception& e) { fprintf(stderr, "%s\n", e.what()); return 99; } }

// poly/dense_mlpoly.rs:435-458 check_polynomial_evaluation: Z=[1,2,1,4], r=[4,3] -> 28, and L*Z*R agrees
KAT(poly_evaluation_28)
  DensePolynomial p({F(1), F(2), F(1), F(4)});
  std::vector<Fr> r{F(4), F(3)};
  CHECK(p.evaluate(r) == F(28), 1);
  auto LR = EqPolynomial(r).compute_factored_evals();
  auto LZ = p.bound(LR.first);
  CHECK(compute_dotproduct(LZ.data(), LR.second.data(), LZ.size()) == F(28), 2);
KAT_END
// poly/dense_mlpoly.rs:627-648 evaluation: constant 8 -> 8
KAT(poly_evaluation_const8)
  DensePolynomial p({F(8), F(8), F(8), F(8)});
  CHECK(p.evaluate({F(3), F(4)}) == F(8), 1);
  CHECK(p.evaluate({F(5), F(10)}) == F(8), 2);
KAT_END
// poly/dense_mlpoly.rs:528-583: eq-evals vs the naive product form; factored L (x) R == full
KAT(eq_evals_vs_naive)
  ChaChaRng rng = test_rng();
  size_t s = 10; std::vector<Fr> r; for (size_t i = 0; i < s; i++) r.push_back(fr_rand(rng));
  auto chis = EqPolynomial(r).evals();
  for (size_t i = 0; i < chis.size(); i++) {  // compute_chis_at_r dense_mlpoly.rs:460-480: bit j of i, MSB first
    Fr c = Fr::one();
    for (size_t j = 0; j < s; j++) { bool bit = (i >> (s - j - 1)) & 1; c *= bit ? r[j] : Fr::one() - r[j]; }
    CHECK(c == chis[i], 1);
  }
  auto LR = EqPolynomial(r).compute_factored_evals();
  CHECK(LR.first.size() * LR.second.size() == chis.size(), 2);
  for (size_t i = 0; i < LR.first.size(); i++) for (size_t j = 0; j < LR.second.size(); j++) CHECK(LR.first[i] * LR.second[j] == chis[i * LR.second.size() + j], 3);
KAT_END
// poly/unipoly.rs:128-156: evals [1,6,15] -> coeffs [1,3,2]; compress/decompress; evaluate(3) = 28
KAT(unipoly_quad)
  UniPoly p = UniPoly::from_evals({F(1), F(6), F(15)});
  CHECK(p.coeffs.size() == 3 && p.coeffs[0] == F(1) && p.coeffs[1] == F(3) && p.coeffs[2] == F(2), 1);
  CHECK(p.eval_at_zero() == F(1) && p.eval_at_one() == F(6), 2);
  UniPoly d = decompress(p.compress(), F(1) + F(6));
  for (size_t i = 0; i < 3; i++) CHECK(d.coeffs[i] == p.coeffs[i], 3);
  CHECK(p.evaluate(F(3)) == F(28), 4);
KAT_END
// poly/unipoly.rs:158-189: evals [1,7,23,55] -> coeffs [1,3,2,1]; evaluate(4) = 109
KAT(unipoly_cubic)
  UniPoly p = UniPoly::from_evals({F(1), F(7), F(23), F(55)});
  CHECK(p.coeffs.size() == 4 && p.coeffs[0] == F(1) && p.coeffs[1] == F(3) && p.coeffs[2] == F(2) && p.coeffs[3] == F(1), 1);
  UniPoly d = decompress(p.compress(), F(1) + F(7));
  for (size_t i = 0; i < 4; i++) CHECK(d.coeffs[i] == p.coeffs[i], 2);
  CHECK(p.evaluate(F(4)) == F(109), 3);
KAT_END
// utils/gaussian_elimination.rs:72-81
KAT(gauss)
  std::vector<std::vector<Fr>> m{{F(1), F(0), F(0), F(2)}, {F(1), F(1), F(1), F(17)}, {F(1), F(2), F(4), F(38)}};
  auto r = gaussian_elimination(m);
  CHECK(r.size() == 3 && r[0] == F(2) && r[1] == F(12) && r[2] == F(3), 1);
KAT_END
// utils/mod.rs:95-98 split_bits
KAT(split_bits)
  CHECK(split_bits(0b0001, 2) == std::make_pair((size_t)0, (size_t)1), 1);
  CHECK(split_bits(0b1001, 2) == std::make_pair((size_t)2, (size_t)1), 2);
KAT_END
// subprotocols/grand_product.rs:269-283: [1,2,3,4] -> 24, prove/verify with real Merlin
KAT(grand_product_24)
  DensePolynomial f({F(1), F(2), F(3), F(4)});
  GrandProductCircuit c(f);
  CHECK(c.evaluate() == F(24), 1);
  MerlinTranscript t("test_transcript");
  std::vector<GrandProductCircuit*> cs{&c}; std::vector<Fr> rand;
  auto proof = bgpa_prove(cs, t, rand);
  MerlinTranscript tv("test_transcript");
  std::vector<Fr> claims, randv;
  bgpa_verify(proof, {F(24)}, 4, tv, claims, randv);
  CHECK(rand.size() == randv.size(), 2);
  for (size_t i = 0; i < rand.size(); i++) CHECK(rand[i] == randv[i], 3);
  CHECK(DensePolynomial({F(1), F(2), F(3), F(4)}).evaluate(randv) == claims[0], 4);  // final claim = MLE of the input at rand
KAT_END
// subprotocols/sumcheck.rs:458-513 sumcheck_arbitrary_cubic with scripted challenges r = [3,1,3]
KAT(sumcheck_scripted_313)
  size_t num_vars = 3, n = 8;
  std::vector<Fr> ev; for (size_t i = 0; i < n; i++) ev.push_back(F(8 + i));
  DensePolynomial A(ev), B(ev), C(ev);
  Fr claim = Fr::zero();
  for (size_t i = 0; i < n; i++) claim += A.evaluate(bitvec(i, num_vars)) * B.evaluate(bitvec(i, num_vars)) * C.evaluate(bitvec(i, num_vars));
  std::vector<DensePolynomial> polys{A.clone(), B.clone(), C.clone()};
  std::vector<Fr> r{F(3), F(1), F(3)};
  ScriptedTranscript t(r, {});
  std::vector<Fr> pr, fin;
  auto proof = prove_arbitrary(num_vars, polys, [](const Fr* v) { return v[0] * v[1] * v[2]; }, 3, t, pr, fin);
  ScriptedTranscript tv(r, {});
  Fr e; std::vector<Fr> vr;
  CHECK(sumcheck_verify(proof, claim, num_vars, 3, tv, e, vr), 1);
  CHECK(pr.size() == 3 && vr.size() == 3, 2);
  for (size_t i = 0; i < 3; i++) CHECK(pr[i] == vr[i] && pr[i] == r[i], 3);
  CHECK(e == A.evaluate(pr) * B.evaluate(pr) * C.evaluate(pr), 4);
  for (size_t i = 0; i < 3; i++) CHECK(fin[i] == A.evaluate(pr), 5);
KAT_END
// lasso/memory_checking.rs:794-831: the 8-cell / 4-op memory; check the multiset identity the reference
// asserts in ProductLayerProof::prove (memory_checking.rs:689)
KAT(memory_checking_multiset)
  std::vector<Fr> table; for (u64 i = 10; i < 18; i++) table.push_back(F(i));
  DensePolynomial dim({F(1), F(2), F(1), F(5)}), rd({F(0), F(0), F(1), F(0)}), fin({F(0), F(2), F(1), F(0), F(0), F(1), F(0), F(0)});
  std::vector<size_t> dim_usize{1, 2, 1, 5};
  DensePolynomial gi, gr, gw, gf;
  GrandProducts::build_grand_product_inputs(table, dim, dim_usize, rd, fin, F(100), F(200), gi, gr, gw, gf);
  GrandProducts gp(gi, gr, gw, gf);
  CHECK(gp.init.evaluate() * gp.write.evaluate() == gp.read.evaluate() * gp.final_.evaluate(), 1);
  // h(a,v,t) = t*gamma^2 + v*gamma + a - tau with small integers: first read tuple (a=1, v=11, t=0)
  CHECK(gr[0] == F(11 * 100 + 1) - F(200), 2);
  CHECK(gw[2] == F(2 * 10000 + 11 * 100 + 1) - F(200), 3);
KAT_END

static Strategy strat(StrategyKind k, size_t C, size_t M, size_t log_r = 0) { Strategy s; s.kind = k; s.C = C; s.M = M; s.LOG_R = log_r; return s; }
static int mle_parity(const Strategy& S) {  // subtables/test.rs:15-40 materialization_mle_parity_test
  auto tabs = S.materialize_subtables();
  size_t bits = ark_log2(S.M);
  for (size_t k = 0; k < tabs.size(); k++) for (size_t i = 0; i < S.M; i++) if (!(tabs[k][i] == S.evaluate_subtable_mle(k, bitvec(i, bits)))) return 1;
  return 0;
}
// subtables/and.rs:69-110,139-146
KAT(and_table)
  auto t = strat(STRAT_AND, 4, 16).materialize_subtables();
  u64 exp[11] = {0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 2};
  CHECK(t.size() == 1 && t[0].size() == 16, 1);
  for (int i = 0; i < 11; i++) CHECK(t[0][i] == F(exp[i]), 2);
  Fr vals[4] = {F(100), F(200), F(300), F(400)};
  CHECK(strat(STRAT_AND, 4, 1 << 16).combine_lookups(vals) == F(100 + (1ull << 8) * 200 + (1ull << 16) * 300 + (1ull << 24) * 400), 3);
  CHECK(mle_parity(strat(STRAT_AND, 4, 16)) == 0, 4);
KAT_END
// subtables/and.rs:112-137 valid_merged_poly
KAT(and_merged_poly)
  Strategy S = strat(STRAT_AND, 2, 16);
  Subtables st(S, {{0, 2}, {5, 9}}, 2);
  u64 exp[4] = {0, 0, 1, 0};
  for (size_t x = 0; x < 4; x++) CHECK(st.combined_poly.evaluate(bitvec(x, 2)) == F(exp[x]), 1);
KAT_END
// subtables/or.rs:71-139
KAT(or_table)
  auto t = strat(STRAT_OR, 4, 16).materialize_subtables();
  u64 exp[11] = {0, 1, 2, 3, 1, 1, 3, 3, 2, 3, 2};
  for (int i = 0; i < 11; i++) CHECK(t[0][i] == F(exp[i]), 1);
  Fr vals[4] = {F(100), F(200), F(300), F(400)};
  CHECK(strat(STRAT_OR, 4, 1 << 16).combine_lookups(vals) == F(100 + (1ull << 8) * 200 + (1ull << 16) * 300 + (1ull << 24) * 400), 2);
  CHECK(mle_parity(strat(STRAT_OR, 4, 16)) == 0, 3);
KAT_END
// subtables/xor.rs:71-139
KAT(xor_table)
  auto t = strat(STRAT_XOR, 4, 16).materialize_subtables();
  u64 exp[11] = {0, 1, 2, 3, 1, 0, 3, 2, 2, 3, 0};
  for (int i = 0; i < 11; i++) CHECK(t[0][i] == F(exp[i]), 1);
  Fr vals[4] = {F(100), F(200), F(300), F(400)};
  CHECK(strat(STRAT_XOR, 4, 1 << 16).combine_lookups(vals) == F(100 + (1ull << 8) * 200 + (1ull << 16) * 300 + (1ull << 24) * 400), 2);
  CHECK(mle_parity(strat(STRAT_XOR, 4, 16)) == 0, 3);
KAT_END
// subtables/lt.rs:85-147
KAT(lt_table)
  Fr vals[8] = {F(10), F(1), F(20), F(0), F(30), F(1), F(40), F(1)};
  CHECK(strat(STRAT_LT, 4, 16).combine_lookups(vals) == F(30), 1);
  auto t = strat(STRAT_LT, 2, 16).materialize_subtables();
  u64 lt[7] = {0, 1, 1, 1, 0, 0, 1}, eq[7] = {1, 0, 0, 0, 0, 1, 0};
  CHECK(t.size() == 2, 2);
  for (int i = 0; i < 7; i++) CHECK(t[0][i] == F(lt[i]) && t[1][i] == F(eq[i]), 3);
  CHECK(mle_parity(strat(STRAT_LT, 4, 16)) == 0, 4);
KAT_END
// subtables/range_check.rs:101-136
KAT(range_table)
  Strategy S = strat(STRAT_RANGE, 4, 1 << 16, 40);
  auto t = S.materialize_subtables();
  CHECK(t.size() == 3, 1);
  for (size_t i = 0; i < S.M; i++) {
    CHECK(t[0][i] == F(i), 2);
    CHECK(t[1][i] == (i < 256 ? F(i) : Fr::zero()), 3);
    CHECK(t[2][i].is_zero(), 4);
  }
  CHECK(mle_parity(S) == 0, 5);
  // memory_to_subtable_index (range_check.rs:62-69) at LOG_R=40, log_m=16: memories 0,1 full; 2 remainder; 3 zeros
  CHECK(S.memory_to_subtable_index(0) == 0 && S.memory_to_subtable_index(1) == 0 && S.memory_to_subtable_index(2) == 1 && S.memory_to_subtable_index(3) == 2, 6);
KAT_END
// STRAT_SPARK_UNCONFIRMED (not in the reference snapshot; lasso_oracle.hpp states what it restates): no reference KAT exists, so the self-consistency properties the reference's
// own macro checks for every strategy (subtables/test.rs:15-40): each subtable's MLE agrees with the materialised table on the hypercube; plus the shape of the strategy
KAT(spark_unconfirmed)
  for (size_t C : {(size_t)1, (size_t)3, (size_t)16}) {
    Strategy S = strat(STRAT_SPARK_UNCONFIRMED, C, 16);
    CHECK(S.num_subtables() == C && S.num_memories() == C && S.g_poly_degree() == C && S.sumcheck_poly_degree() == C + 1, 1);
    for (size_t i = 0; i < C; i++) CHECK(S.memory_to_subtable_index(i) == i && S.memory_to_dimension_index(i) == i, 2);
    CHECK(mle_parity(S) == 0, 3);
    auto t = S.materialize_subtables();
    for (size_t k = 0; k < C; k++) { Fr sum = Fr::zero(); for (auto& x : t[k]) sum += x; CHECK(sum == Fr::one(), 4); }   // an eq table sums to 1
  }
  Fr vals[3] = {F(3), F(5), F(7)};
  CHECK(strat(STRAT_SPARK_UNCONFIRMED, 3, 16).combine_lookups(vals) == F(105), 5);
KAT_END
// poly/dense_mlpoly.rs:585-625 check_polynomial_commit: commit -> open -> verify
KAT(poly_commit_open_verify)
  DensePolynomial p({F(1), F(2), F(1), F(4)});
  std::vector<Fr> r{F(4), F(3)};
  Fr ev = p.evaluate(r);
  CHECK(ev == F(28), 1);
  PolyCommitmentGens gens = PolyCommitmentGens::create(p.num_vars, "test-two");
  PolyCommitment comm = p.commit(gens);
  RandomTape tape("proof"); MerlinTranscript t("example");
  PolyEvalProof proof = poly_eval_prove(p, r, ev, gens, t, tape);
  MerlinTranscript tv("example");
  CHECK(poly_eval_verify_plain(proof, gens, tv, r, ev, comm), 2);
  MerlinTranscript tw("example");
  CHECK(!poly_eval_verify_plain(proof, gens, tw, r, F(29), comm), 3);  // wrong evaluation must be rejected
KAT_END
// subprotocols/dot_product.rs:349-384 check_dotproductproof_log (n = 1024 there; 64 here for time)
KAT(dot_product_log)
  ChaChaRng rng = test_rng();
  size_t n = 64;
  DotProductProofGens gens = DotProductProofGens::create(n, "test-1024");
  std::vector<Fr> x, a; for (size_t i = 0; i < n; i++) { x.push_back(fr_rand(rng)); a.push_back(fr_rand(rng)); }
  Fr y = compute_dotproduct(x.data(), a.data(), n), r_x = fr_rand(rng), r_y = fr_rand(rng);
  RandomTape tape("proof"); MerlinTranscript t("example");
  Point Cx, Cy;
  auto proof = dot_product_log_prove(gens, t, tape, x, r_x, a, y, r_y, &Cx, &Cy);
  MerlinTranscript tv("example");
  CHECK(dot_product_log_verify(proof, n, gens, tv, a, Cx, Cy), 1);
KAT_END

// e2e_test.rs:64-99 — the four prove -> verify configurations, inputs from utils/test.rs:15-32
static int e2e(StrategyKind k, size_t C, size_t M, size_t log_r, size_t sparsity) {
  Strategy S = strat(k, C, M, log_r);
  size_t log_M = log_2(M), log_s = ark_log2(sparsity);
  auto nz = gen_indices(C, sparsity, M);
  auto dense = DensifiedRepresentation::from_lookup_indices(nz, C, log_M);
  auto gens = SparsePolyCommitmentGens::create("gens_sparse_poly", C, sparsity, S.num_memories(), log_M);
  auto commitment = dense.commit(gens);
  auto r = gen_random_point(log_s);
  RandomTape tape("proof"); MerlinTranscript t("example");
  auto proof = surge_prove(S, dense, r, gens, t, tape);
  MerlinTranscript tv("example");
  if (!surge_verify(S, proof, commitment, r, gens, tv)) return 1;
  // serialisation round trip, then verify the parsed copy
  auto bytes = serialize_proof(proof);
  SparsePolynomialEvaluationProof back;
  if (!deserialize_proof(S, bytes.data(), bytes.size(), back)) return 2;
  if (serialize_proof(back) != bytes) return 3;
  MerlinTranscript tv2("example");
  if (!surge_verify(S, back, commitment, r, gens, tv2)) return 4;
  // a tampered proof must be rejected (flip one byte of claimed_evaluation region: find it by re-serialising)
  back.primary_sumcheck.eval_derefs[0] += Fr::one();
  MerlinTranscript tv3("example");
  bool rejected;
  try { rejected = !surge_verify(S, back, commitment, r, gens, tv3); } catch (...) { rejected = true; }
  if (!rejected) return 5;
  return 0;
}
KAT(e2e_prove_4d_lt) return e2e(STRAT_LT, 4, 16, 0, 16); KAT_END
KAT(e2e_prove_4d_lt_big_s) return e2e(STRAT_LT, 4, 16, 0, 128); KAT_END
KAT(e2e_prove_4d_and) return e2e(STRAT_AND, 4, 16, 0, 16); KAT_END
KAT(e2e_prove_3d_range) return e2e(STRAT_RANGE, 3, 256, 40, 16); KAT_END
KAT(e2e_prove_1d_and_s64) return e2e(STRAT_AND, 1, 16, 0, 64); KAT_END
KAT(e2e_prove_2d_xor) return e2e(STRAT_XOR, 2, 16, 0, 32); KAT_END
KAT(e2e_prove_2d_or) return e2e(STRAT_OR, 2, 16, 0, 8); KAT_END
KAT(e2e_prove_spark_unconfirmed) return e2e(STRAT_SPARK_UNCONFIRMED, 3, 16, 0, 16); KAT_END   // the same acceptance criterion for the restated strategy (not a reference test)

// rand 0.8 `StdRng` value stability (rand/src/rngs/std.rs test_stdrng_construction) — the seed is ark_std::test_rng()'s (ark-std 0.4 rand_helper.rs), the source of the
// harness's indices and points (benches/bench.rs:13-34) and of RandomTape's init scalar (utils/random.rs:17): first next_u64 and, through SeedableRng::from_rng
// (fill_bytes of 32 = the next eight words), the second generator's first next_u64.  Published by the rand crate, not derived here.
KAT(rand_stdrng_value_stability)
  ChaChaRng r0 = test_rng();
  CHECK(r0.next_u64() == 10719222850664546238ull, 1);
  uint8_t seed[32]; for (int i = 0; i < 8; i++) { const uint32_t w = r0.next_u32(); for (int k = 0; k < 4; k++) seed[4 * i + k] = (uint8_t)(w >> (8 * k)); }
  ChaChaRng r1(seed, 12);
  CHECK(r1.next_u64() == 14064965282130556830ull, 2);
KAT_END
// rand_chacha 0.3 test_chacha_true_values_a (src/chacha.rs): ChaCha20Rng::from_seed([0; 32]) — the generator stream's RNG (commitments.rs:31) — first two blocks
KAT(rand_chacha20_true_values_a)
  const uint32_t e1[16] = {0xade0b876, 0x903df1a0, 0xe56a5d40, 0x28bd8653, 0xb819d2bd, 0x1aed8da0, 0xccef36a8, 0xc70d778b, 0x7c5941da, 0x8d485751, 0x3fe02477, 0x374ad8b8, 0xf4b8436a, 0x1ca11815, 0x69b687c3, 0x8665eeb2};
  const uint32_t e2[16] = {0xbee7079f, 0x7a385155, 0x7c97ba98, 0x0d082d73, 0xa0290fcb, 0x6965e348, 0x3e53c612, 0xed7aee32, 0x7621b729, 0x434ee69c, 0xb03371d5, 0xd539d874, 0x281fed31, 0x45fb0a51, 0x1f0ae1ac, 0x6f4d794b};
  const uint8_t z[32] = {0}; ChaChaRng r(z, 20);
  for (int i = 0; i < 16; i++) CHECK(r.next_u32() == e1[i], 1);
  for (int i = 0; i < 16; i++) CHECK(r.next_u32() == e2[i], 2);
KAT_END

extern "C" const char* orc_kat_names() {
  return "poly_evaluation_28,poly_evaluation_const8,eq_evals_vs_naive,unipoly_quad,unipoly_cubic,gauss,split_bits,grand_product_24,"
         "sumcheck_scripted_313,memory_checking_multiset,and_table,and_merged_poly,or_table,xor_table,lt_table,range_table,"
         "poly_commit_open_verify,dot_product_log,e2e_prove_4d_lt,e2e_prove_4d_lt_big_s,e2e_prove_4d_and,e2e_prove_3d_range,"
         "e2e_prove_1d_and_s64,e2e_prove_2d_xor,e2e_prove_2d_or,spark_unconfirmed,e2e_prove_spark_unconfirmed,rand_stdrng_value_stability,rand_chacha20_true_values_a";
}

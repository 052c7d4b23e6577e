// SparsePolynomialEvaluationProof::verify (src/lasso/surge.rs:214-271) — the product-side verifier and the reader of the proof's wire format.
//
// SURVEY.md §8 row (f4).  The verifier is O(log^2 s + sqrt s) field work on the host plus three kinds of group work: decompressing the commitment
// rows, one MSM over them per opening (C_LZ = <L, C>, dense_mlpoly.rs:376-379) and one MSM over the generators per opening (G_hat = <s, G>,
// bullet.rs:232-236).  The two MSMs run on the device through the same C ABI the prover uses (lasso_bases_create / lasso_msm); everything else
// is host arithmetic over the shared field headers.  Function by function it follows:
//   surge.rs:214-271                       SparsePolynomialEvaluationProof::verify
//   subprotocols/sumcheck.rs:286-328       SumcheckInstanceProof::verify           poly/unipoly.rs:96-109  CompressedUniPoly::decompress
//   subtables/mod.rs:315-375               CombinedTableEvalProof::verify / verify_single
//   poly/dense_mlpoly.rs:361-400           PolyEvalProof::verify / verify_plain
//   subprotocols/dot_product.rs:251-296    DotProductProofLog::verify
//   subprotocols/bullet.rs:158-257         verification_scalars / verify
//   lasso/memory_checking.rs:96-143        MemoryCheckingProof::verify            :733-785 ProductLayerProof::verify
//   lasso/memory_checking.rs:525-648       HashLayerProof::verify                 :462-523 check_reed_solomon_fingerprints
//   subprotocols/grand_product.rs:203-261  BatchedGrandProductArgument::verify
//   subtables/{and,or,xor,lt,range_check}.rs evaluate_subtable_mle / combine_lookups;  poly/identity_poly.rs:14-20
// Wire format: ark-serialize 0.4 CanonicalSerialize, compressed (what `#[derive(CanonicalSerialize)]` gives the structs of surge.rs:61-104,
// memory_checking.rs:26,:313,:655, dot_product.rs:152, bullet.rs:23, grand_product.rs:68,:94, sumcheck.rs:263, unipoly.rs:19): Vec<T> = u64 LE
// length + items, arrays = items, Fr = 32 canonical LE bytes (rejected when >= p), points = 32 bytes with the curve's flag bits.
// A rejected proof returns false; a proof the reference would `assert!` on / fail to deserialize throws Error (-> non-zero status at the C ABI).
#pragma once
#include "prover.hpp"

namespace lasso {

// ------------------------------------------------------------------ points from the wire (ark-ec deserialize_compressed with Validate::Yes)
#ifdef LASSO_BN254
// short Weierstrass (ark-ec SWFlags): canonical x, bit 7 of the last byte = "y is the larger root", bit 6 = infinity.  G1 has cofactor 1.
inline bool decompress_point(const uint8_t in[32], Pt& out, bool& infinity) {
  uint8_t b[32]; memcpy(b, in, 32);
  const bool neg = (b[31] & 0x80) != 0; infinity = (b[31] & 0x40) != 0; b[31] &= 0x3f;
  fq_t c; memcpy(c.v, b, 32);
  // ark-ec 0.4 (SWCurveConfig::deserialize_with_mode): both flags set is no SWFlags value; x must be a canonical field element whatever the flags say;
  // with the infinity flag the point is the identity WHATEVER x is (the transcript then absorbs the re-serialised identity: x = 0 with the flag)
  if (neg && infinity) return false;
  if (fq_geq_p(c.v)) return false;
  if (infinity) { out = Pt::identity(); return true; }
  const fq_t x = fq_from_canonical(c);
  fq_t y;
  if (!fq_sqrt(fq_add(fq_mul(fq_sqr(x), x), fq_from_u64(3)), y)) return false;
  const fq_t ny = fq_neg(y);
  const bool y_larger = canonical_less(ny, y);
  out = Pt::from_affine_plain(x, (neg == y_larger) ? y : ny);
  return true;
}
inline void affine_to_abi(const Pt& p, lasso_affine& a) { memcpy(a.x, p.p.X.v, 32); memcpy(a.y, p.p.Y.v, 32); }   // Z = 1, Montgomery limbs as stored
inline void compress_affine_pt(const Pt& p, bool infinity, uint8_t out[32]) { if (infinity) compress_infinity(out); else compress_affine(p.p.X, p.p.Y, out); }
#else
// twisted Edwards (ark-ec TEFlags): canonical y, bit 7 of the last byte = "x is the larger root"; the point must lie in the prime-order subgroup
// (Affine::check: is_on_curve && is_in_correct_subgroup_assuming_on_curve, cofactor 8)
inline bool decompress_point(const uint8_t in[32], Pt& out, bool& infinity) {
  infinity = false;
  uint8_t b[32]; memcpy(b, in, 32);
  const bool neg = (b[31] & 0x80) != 0; b[31] &= 0x7f;
  fq_t y; memcpy(y.v, b, 32);
  {
    const uint32_t P[8] = {0xffffffedu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x7fffffffu};
    bool lt = false; for (int i = 7; i >= 0; i--) if (y.v[i] != P[i]) { lt = y.v[i] < P[i]; break; }
    if (!lt) return false;
  }
  const fq_t y2 = fq_sqr(y), den = fq_sub(fq_neg(fq_one()), fq_mul(fq_d(), y2));   // x^2 = (1 - y^2) / (a - d y^2), a = -1
  if (fq_is_zero(den)) return false;
  fq_t x;
  if (!fq_sqrt(fq_mul(fq_sub(fq_one(), y2), fq_inv_host(den)), x)) return false;
  const fq_t nx = fq_neg(x);
  const bool x_larger = canonical_less(nx, x);
  out = Pt::from_affine_plain((neg == x_larger) ? x : nx, y);
  uint32_t order[8]; for (int i = 0; i < 8; i++) order[i] = fr_p_limb(i);
  const ed_point chk = ed_mul_limbs(out.p, order, 253);
  return ed_eq(chk, ed_identity());
}
inline void affine_to_abi(const Pt& p, lasso_affine& a) { const fq_t x = fq_to_mont(p.p.X), y = fq_to_mont(p.p.Y); memcpy(a.x, x.v, 32); memcpy(a.y, y.v, 32); }
inline void compress_affine_pt(const Pt& p, bool, uint8_t out[32]) { compress_affine(p.p.X, p.p.Y, out); }
#endif

struct WirePoint { Pt p; bool infinity = false; uint8_t bytes[32]; };   // bytes = serialize_compressed of the decoded point (what the transcript absorbs)

// ------------------------------------------------------------------ reader of the ark-serialize byte stream
struct ProofReader {
  const uint8_t* p; size_t n, pos = 0;
  ProofReader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
  void need(size_t k) const { if (pos + k > n) throw Error("proof bytes: truncated"); }
  uint64_t u64le() { need(8); uint64_t x = 0; for (int i = 0; i < 8; i++) x |= (uint64_t)p[pos + i] << (8 * i); pos += 8; return x; }
  Sc sc() {
    need(32); fr_t c; memcpy(c.v, p + pos, 32); pos += 32;
    if (fr_geq_p(c.v)) throw Error("proof bytes: scalar is not canonical");
    Sc s; s.v = fr_mul(c, fr_r2()); return s;   // canonical integer -> Montgomery form
  }
  WirePoint pt() {
    need(32); WirePoint w;
    if (!decompress_point(p + pos, w.p, w.infinity)) throw Error("proof bytes: invalid point encoding");
    compress_affine_pt(w.p, w.infinity, w.bytes); pos += 32; return w;
  }
  // a length prefix is believed only as far as the bytes behind it can back it: k elements of at least `elem` bytes each must still fit, so a
  // crafted prefix cannot make the reader reserve more than the proof's own size (a WirePoint is ~200 B in memory for 32 B on the wire)
  size_t len(size_t elem) { const uint64_t k = u64le(); if (k > (n - pos) / elem) throw Error("proof bytes: implausible vector length"); return (size_t)k; }
  ScVec sc_vec() { const size_t k = len(32); ScVec v; v.reserve(k); for (size_t i = 0; i < k; i++) v.push_back(sc()); return v; }
  ScVec sc_arr(size_t k) { ScVec v; for (size_t i = 0; i < k; i++) v.push_back(sc()); return v; }
  std::vector<WirePoint> pts_vec() { const size_t k = len(32); std::vector<WirePoint> v; v.reserve(k); for (size_t i = 0; i < k; i++) v.push_back(pt()); return v; }
  SumcheckProof sumcheck() { SumcheckProof s; const size_t k = len(8); for (size_t i = 0; i < k; i++) s.compressed_polys.push_back(sc_vec()); return s; }   // every inner vector carries at least its own 8-byte prefix
  bool done() const { return pos == n; }
};
struct WireDotProductProofLog { std::vector<WirePoint> L_vec, R_vec; WirePoint delta, beta; Sc z1, z2; };   // dot_product.rs:152-159 + bullet.rs:23-28
struct WireBgpa { std::vector<LayerProofBatched> proof; };
struct WireProof {   // surge.rs:92-104 with its nested structs flattened in declaration (= serialization) order
  std::vector<WirePoint> comm_derefs;
  SumcheckProof primary; Sc claimed_evaluation; ScVec eval_derefs; WireDotProductProofLog proof_derefs;
  std::vector<std::array<Sc, 4>> grand_product_evals;   // (init, read, write, final) memory_checking.rs:655-662
  WireBgpa proof_mem, proof_ops;
  ScVec eval_dim, eval_read, eval_final, eval_derefs_hash; WireDotProductProofLog open_ops, open_mem, open_derefs;   // memory_checking.rs:313-336
};
inline WireDotProductProofLog read_dpl(ProofReader& r) { WireDotProductProofLog d; d.L_vec = r.pts_vec(); d.R_vec = r.pts_vec(); d.delta = r.pt(); d.beta = r.pt(); d.z1 = r.sc(); d.z2 = r.sc(); return d; }
inline WireBgpa read_bgpa(ProofReader& r) {
  WireBgpa g; const size_t k = r.len(24);   // a layer = three length prefixes at least
  for (size_t i = 0; i < k; i++) { LayerProofBatched l; l.proof = r.sumcheck(); l.claims_prod_left = r.sc_vec(); l.claims_prod_right = r.sc_vec(); g.proof.push_back(std::move(l)); }
  return g;
}
inline WireProof read_proof(const Strategy& S, const uint8_t* bytes, size_t n) {
  ProofReader r(bytes, n); WireProof P;
  const size_t alpha = S.num_memories(), C = S.C();
  P.comm_derefs = r.pts_vec();
  P.primary = r.sumcheck(); P.claimed_evaluation = r.sc(); P.eval_derefs = r.sc_arr(alpha); P.proof_derefs = read_dpl(r);
  for (size_t i = 0; i < alpha; i++) { std::array<Sc, 4> e; for (auto& x : e) x = r.sc(); P.grand_product_evals.push_back(e); }
  P.proof_mem = read_bgpa(r); P.proof_ops = read_bgpa(r);
  P.eval_dim = r.sc_arr(C); P.eval_read = r.sc_arr(C); P.eval_final = r.sc_arr(C); P.eval_derefs_hash = r.sc_arr(alpha);
  P.open_ops = read_dpl(r); P.open_mem = read_dpl(r); P.open_derefs = read_dpl(r);
  if (!r.done()) throw Error("proof bytes: trailing data");
  return P;
}

// ------------------------------------------------------------------ host side of SubtableStrategy the verifier needs (subtables/*.rs)
inline Sc evaluate_subtable_mle(const Strategy& S, size_t k, const ScVec& point) {
  const int kind = S.abi.kind; const size_t n = point.size();
  if (kind == LASSO_SPARK_UNCONFIRMED) {   // the MLE of eq(tau_k, .) at a point is eq(tau_k, point) (eq_poly.rs:14-20)
    const std::vector<ScVec> tau = S.spark_point(); LASSO_REQUIRE(k < tau.size() && n == tau[k].size());
    Sc res = Sc::one();
    for (size_t b = 0; b < n; b++) res *= tau[k][b] * point[b] + (Sc::one() - tau[k][b]) * (Sc::one() - point[b]);
    return res;
  }
  if (kind == LASSO_RANGE) {   // range_check.rs:42-66
    if (k == 2) return Sc::zero();
    const size_t cutoff = S.abi.log_r % S.abi.log_m; Sc res = Sc::zero();
    for (size_t i = 0; i < n; i++) {
      if (k == 0 || i < cutoff) res += Sc::from_u64((uint64_t)1 << i) * point[n - i - 1];
      else res *= Sc::one() - point[n - i - 1];
    }
    return res;
  }
  LASSO_REQUIRE(n % 2 == 0);
  const size_t b = n / 2; const Sc* x = point.data(); const Sc* y = point.data() + b;
  if (kind == LASSO_LT) {   // lt.rs:33-57: LT = sum_i (1-x_i) y_i eq(x_<i, y_<i); EQ = prod_i eq(x_i, y_i)
    Sc res = Sc::zero(), eq = Sc::one();
    for (size_t i = 0; i < b; i++) { res += (Sc::one() - x[i]) * y[i] * eq; eq *= Sc::one() - x[i] - y[i] + Sc::from_u64(2) * x[i] * y[i]; }
    return k % 2 == 0 ? res : eq;
  }
  Sc res = Sc::zero();   // and.rs:30-40, or.rs:28-41, xor.rs:28-41
  for (size_t i = 0; i < b; i++) {
    const Sc& xi = x[b - i - 1]; const Sc& yi = y[b - i - 1];
    const Sc bit = kind == LASSO_AND ? xi * yi : kind == LASSO_OR ? Sc::one() - (Sc::one() - xi) * (Sc::one() - yi) : (Sc::one() - xi) * yi + xi * (Sc::one() - yi);
    res += Sc::from_u64((uint64_t)1 << i) * bit;
  }
  return res;
}
inline Sc combine_lookups(const Strategy& S, const ScVec& vals) {
  if (S.spark()) { Sc prod = Sc::one(); for (size_t i = 0; i < S.C(); i++) prod *= vals[i]; return prod; }
  if (S.abi.kind == LASSO_LT) {   // lt.rs:62-71
    Sc sum = Sc::zero(), eq = Sc::one();
    for (size_t i = 0; i < S.C(); i++) { sum += vals[2 * i] * eq; eq *= vals[2 * i + 1]; }
    return sum;
  }
  const ScVec w = S.weights(); Sc sum = Sc::zero();   // and.rs:45-53, range_check.rs:78-86
  for (size_t i = 0; i < vals.size(); i++) sum += w[i] * vals[i];
  return sum;
}

// ------------------------------------------------------------------ the verifier
class Verifier {
  const Dev& d; const Strategy& S; const SparsePolyCommitmentGens& gens; ProofTranscript& t;

  // sumcheck.rs:286-328
  bool sumcheck_verify(const SumcheckProof& proof, const Sc& claim, size_t num_rounds, size_t degree_bound, Sc& e_out, ScVec& r_out) {
    Sc e = claim; r_out.clear();
    if (proof.compressed_polys.size() != num_rounds) throw Error("sumcheck proof: wrong number of rounds");
    for (const ScVec& c : proof.compressed_polys) {
      if (c.empty()) throw Error("sumcheck proof: empty polynomial");
      Sc lin = e - c[0] - c[0]; for (size_t i = 1; i < c.size(); i++) lin -= c[i];   // decompress: unipoly.rs:96-109
      UniPoly poly; poly.coeffs.push_back(c[0]); poly.coeffs.push_back(lin); poly.coeffs.insert(poly.coeffs.end(), c.begin() + 1, c.end());
      if (poly.coeffs.size() - 1 != degree_bound) return false;   // ProofVerifyError::InvalidInputLength
      poly.append_to_transcript(t, "poly");
      const Sc r_i = t.challenge_scalar("challenge_nextround");
      r_out.push_back(r_i);
      e = poly.evaluate(r_i);
    }
    e_out = e; return true;
  }
  // grand_product.rs:203-261
  bool bgpa_verify(const WireBgpa& p, const ScVec& claims_prod_vec, size_t len, ScVec& claims_out, ScVec& rand_out) {
    const size_t num_layers = ceil_log2(len);
    if (p.proof.size() != num_layers) throw Error("grand product proof: wrong number of layers");
    ScVec rand, claims = claims_prod_vec;
    for (size_t i = 0; i < num_layers; i++) {
      const ScVec coeff = t.challenge_vector("rand_coeffs_next_layer", claims.size());
      Sc claim = Sc::zero(); for (size_t k = 0; k < claims.size(); k++) claim += claims[k] * coeff[k];
      Sc claim_last; ScVec rand_prod;
      if (!sumcheck_verify(p.proof[i].proof, claim, i, 3, claim_last, rand_prod)) return false;
      const ScVec& cl = p.proof[i].claims_prod_left; const ScVec& cr = p.proof[i].claims_prod_right;
      if (cl.size() != claims_prod_vec.size() || cr.size() != claims_prod_vec.size()) throw Error("grand product proof: wrong number of claims");
      for (size_t k = 0; k < cl.size(); k++) { t.append_scalar("claim_prod_left", cl[k]); t.append_scalar("claim_prod_right", cr[k]); }
      LASSO_REQUIRE(rand.size() == rand_prod.size());
      Sc eq = Sc::one(); for (size_t k = 0; k < rand.size(); k++) eq *= rand[k] * rand_prod[k] + (Sc::one() - rand[k]) * (Sc::one() - rand_prod[k]);
      Sc expected = Sc::zero(); for (size_t k = 0; k < cl.size(); k++) expected += coeff[k] * (cl[k] * cr[k] * eq);
      if (!(expected == claim_last)) return false;   // assert_eq!(claim_expected, claim_last)
      const Sc r_layer = t.challenge_scalar("challenge_r_layer");
      claims.clear(); for (size_t k = 0; k < cl.size(); k++) claims.push_back(cl[k] + r_layer * (cr[k] - cl[k]));
      ScVec ext{r_layer}; ext.insert(ext.end(), rand_prod.begin(), rand_prod.end()); rand.swap(ext);
    }
    claims_out = claims; rand_out = rand; return true;
  }
  // VariableBaseMSM::msm over arbitrary points (the commitment rows): a base table built for this call, then the device MSM
  Pt msm_points(const std::vector<WirePoint>& pts, const ScVec& scalars) {
    LASSO_REQUIRE(pts.size() == scalars.size());
    std::vector<lasso_affine> aff; std::vector<lasso_fr> sc;
    for (size_t i = 0; i < pts.size(); i++) { if (pts[i].infinity) continue; lasso_affine a; affine_to_abi(pts[i].p, a); aff.push_back(a); sc.push_back(scalars[i].abi()); }
    if (aff.empty()) return Pt::identity();
    lasso_bases* b = nullptr;
    d.chk(lasso_bases_create(d.ctx, aff.data(), aff.size(), &b), "lasso_bases_create");
    lasso_point out; const int32_t rc = lasso_msm(d.ctx, b, sc.data(), sc.size(), &out);
    lasso_bases_destroy(d.ctx, b);
    d.chk(rc, "lasso_msm");
    return Pt::from_abi(out);
  }
  static void wire(const Pt& p, uint8_t out[32]) { compress_one(p, out); }
  // bullet.rs:158-257 inside dot_product.rs:251-296 inside dense_mlpoly.rs:361-400 (verify_plain: the claimed evaluation is committed with blind 0)
  bool poly_eval_verify_plain(const WireDotProductProofLog& p, const PolyCommitmentGens& g, const ScVec& r, const Sc& Zr, const std::vector<WirePoint>& comm) {
    const Pt C_Zr = g.Qmul.mul(Zr);   // Zr.commit(&0, &gens.gens.gens_1) = Zr * G[0] + 0 * h
    t.append_protocol_name("polynomial evaluation proof");
    const size_t left = r.size() / 2;   // EqPolynomial::compute_factored_evals eq_poly.rs:44-52
    const ScVec L = eq_evals_host(r.data(), left), R = eq_evals_host(r.data() + left, r.size() - left);
    if (comm.size() != L.size()) throw Error("commitment: wrong number of rows for this opening");
    const Pt C_LZ = msm_points(comm, L);
    // DotProductProofLog::verify(n = R.len(), gens, a = R, Cx = C_LZ, Cy = C_Zr)
    const size_t n = R.size();
    if (g.n != n) throw Error("generators: wrong size for this opening");
    t.append_protocol_name("dot product proof (log)");
    uint8_t buf[32];
    wire(C_LZ, buf); t.append_point_bytes("Cx", buf);
    wire(C_Zr, buf); t.append_point_bytes("Cy", buf);
    t.append_scalars("a", R);
    const Pt Gamma = C_LZ + C_Zr;
    // verification_scalars
    const size_t lg_n = p.L_vec.size();
    if (lg_n >= 32 || p.R_vec.size() != lg_n) return false;            // InputTooLarge
    if (n != ((size_t)1 << lg_n)) return false;                        // InvalidInputLength
    ScVec ch;
    for (size_t i = 0; i < lg_n; i++) { t.append_point_bytes("L", p.L_vec[i].bytes); t.append_point_bytes("R", p.R_vec[i].bytes); ch.push_back(t.challenge_scalar("u")); }
    ScVec ch_inv; for (auto& c : ch) { if (c.is_zero()) return false; ch_inv.push_back(c.inverse()); }
    Sc all_inv = Sc::one(); for (auto& c : ch_inv) all_inv *= c;
    for (size_t i = 0; i < lg_n; i++) { ch[i] = ch[i].square(); ch_inv[i] = ch_inv[i].square(); }
    ScVec s(n); s[0] = all_inv;
    for (size_t i = 1; i < n; i++) { const size_t lg_i = 31 - (size_t)__builtin_clz((uint32_t)i), k = (size_t)1 << lg_i; s[i] = s[i - k] * ch[(lg_n - 1) - lg_i]; }
    // G_hat = <s, G>: the generators' device table (first n entries of [G.., Q, h])
    std::vector<lasso_fr> s_abi(n); for (size_t i = 0; i < n; i++) s_abi[i] = s[i].abi();
    lasso_point gh; d.chk(lasso_msm(d.ctx, g.bases, s_abi.data(), n, &gh), "lasso_msm");
    const Pt G_hat = Pt::from_abi(gh);
    Sc a_hat = Sc::zero(); for (size_t i = 0; i < n; i++) a_hat += R[i] * s[i];
    Pt Gamma_hat = Gamma;   // <u^2, L> + <u^-2, R> + Gamma: 2 lg n + 1 terms, host
    for (size_t i = 0; i < lg_n; i++) Gamma_hat = Gamma_hat + p.L_vec[i].p * ch[i] + p.R_vec[i].p * ch_inv[i];
    t.append_point_bytes("delta", p.delta.bytes); t.append_point_bytes("beta", p.beta.bytes);
    const Sc c = t.challenge_scalar("c");
    const Pt lhs = (Gamma_hat * c + p.beta.p) * a_hat + p.delta.p;
    const Pt rhs = (G_hat + g.Qmul.mul(a_hat)) * p.z1 + g.hmul.mul(p.z2);
    return ed_eq(lhs.p, rhs.p);
  }
  // subtables/mod.rs:315-375
  bool combined_table_eval_verify(const WireDotProductProofLog& proof, const ScVec& r, const ScVec& evals_in, const std::vector<WirePoint>& comm) {
    t.append_protocol_name("Lasso CombinedTableEvalProof");
    ScVec evals = evals_in; evals.resize(next_pow2(evals.size()), Sc::zero());
    t.append_scalars("evals_ops_val", evals);
    const ScVec ch = t.challenge_vector("challenge_combine_n_to_one", ceil_log2(evals.size()));
    const Sc joint = bound_bot_all(evals, ch);
    ScVec r_joint = ch; r_joint.insert(r_joint.end(), r.begin(), r.end());
    t.append_scalar("joint_claim_eval", joint);
    return poly_eval_verify_plain(proof, gens.gens_derefs, r_joint, joint, comm);
  }
  // for i in (0..challenges.len()).rev() { poly.bound_poly_var_bot(&challenges[i]) }  (dense_mlpoly.rs:218-225)
  static Sc bound_bot_all(ScVec z, const ScVec& ch) {
    for (size_t i = ch.size(); i-- > 0;) { const size_t n = z.size() / 2; for (size_t k = 0; k < n; k++) z[k] = z[2 * k] + ch[i] * (z[2 * k + 1] - z[2 * k]); z.resize(n); }
    LASSO_REQUIRE(z.size() == 1); return z[0];
  }

 public:
  Verifier(const Dev& d_, const Strategy& S_, const SparsePolyCommitmentGens& g_, ProofTranscript& t_) : d(d_), S(S_), gens(g_), t(t_) {}

  // commitment: SparsePolynomialCommitment's two PolyCommitments in wire form ([u64 n][n x 32 B] twice: lasso_host_commit's layout) for s lookups, M = 2^log_m
  bool verify(const uint8_t* proof_bytes, size_t proof_len, const uint8_t* comm_bytes, size_t comm_len, size_t s, size_t log_m, const ScVec& eq_randomness) {
    const WireProof P = read_proof(S, proof_bytes, proof_len);
    ProofReader cr(comm_bytes, comm_len);
    const std::vector<WirePoint> comm_l = cr.pts_vec(), comm_m = cr.pts_vec();
    if (!cr.done()) throw Error("commitment bytes: trailing data");
    const size_t alpha = S.num_memories(), m = (size_t)1 << log_m;
    LASSO_REQUIRE(log_m == S.abi.log_m && eq_randomness.size() == ceil_log2(s));
    // surge.rs:214-271
    t.append_protocol_name("Lasso SparsePolynomialEvaluationProof");
    t.append_message("subtable_evals_commitment", "begin_subtable_evals_commitment");   // CombinedTableCommitment::append_to_transcript subtables/mod.rs:382-393
    t.append_message("comm_poly_row_col_ops_val", "poly_commitment_begin");
    for (auto& w : P.comm_derefs) t.append_point_bytes("poly_commitment_share", w.bytes);
    t.append_message("comm_poly_row_col_ops_val", "poly_commitment_end");
    t.append_message("subtable_evals_commitment", "end_subtable_evals_commitment");
    t.append_scalar("claim_eval_scalar_product", P.claimed_evaluation);
    Sc claim_last; ScVec r_z;
    if (!sumcheck_verify(P.primary, P.claimed_evaluation, ceil_log2(s), S.sumcheck_poly_degree(), claim_last, r_z)) return false;
    Sc eq_eval = Sc::one();   // EqPolynomial::evaluate eq_poly.rs:14-19
    for (size_t i = 0; i < r_z.size(); i++) eq_eval *= eq_randomness[i] * r_z[i] + (Sc::one() - eq_randomness[i]) * (Sc::one() - r_z[i]);
    if (!(eq_eval * combine_lookups(S, P.eval_derefs) == claim_last)) return false;   // :245-249
    if (!combined_table_eval_verify(P.proof_derefs, r_z, P.eval_derefs, P.comm_derefs)) return false;
    const ScVec r_mem_check = t.challenge_vector("challenge_r_hash", 2);
    const Sc& gamma = r_mem_check[0]; const Sc& tau = r_mem_check[1];
    // MemoryCheckingProof::verify memory_checking.rs:96-143
    t.append_protocol_name("Lasso MemoryCheckingProof");
    const size_t num_ops = next_pow2(s), num_cells = m;
    // ProductLayerProof::verify :733-785
    t.append_protocol_name("Lasso ProductLayerProof");
    ScVec rw_claims, if_claims;
    for (auto& e : P.grand_product_evals) {
      if (!(e[0] * e[2] == e[1] * e[3])) return false;   // multiset equality: init * write == read * final
      t.append_scalar("claim_hash_init", e[0]); t.append_scalar("claim_hash_read", e[1]); t.append_scalar("claim_hash_write", e[2]); t.append_scalar("claim_hash_final", e[3]);
      rw_claims.push_back(e[1]); rw_claims.push_back(e[2]); if_claims.push_back(e[0]); if_claims.push_back(e[3]);
    }
    ScVec claims_ops, rand_ops, claims_mem, rand_mem;
    if (!bgpa_verify(P.proof_ops, rw_claims, num_ops, claims_ops, rand_ops)) return false;
    if (!bgpa_verify(P.proof_mem, if_claims, num_cells, claims_mem, rand_mem)) return false;
    // HashLayerProof::verify :525-648
    t.append_protocol_name("Lasso HashLayerProof");
    if (!combined_table_eval_verify(P.open_derefs, rand_ops, P.eval_derefs_hash, P.comm_derefs)) return false;
    ScVec evals_ops = P.eval_dim; evals_ops.insert(evals_ops.end(), P.eval_read.begin(), P.eval_read.end());
    evals_ops.resize(next_pow2(evals_ops.size()), Sc::zero());
    t.append_scalars("claim_evals_ops", evals_ops);
    const ScVec ch_ops = t.challenge_vector("challenge_combine_n_to_one", ceil_log2(evals_ops.size()));
    const Sc joint_ops = bound_bot_all(evals_ops, ch_ops);
    ScVec r_joint_ops = ch_ops; r_joint_ops.insert(r_joint_ops.end(), rand_ops.begin(), rand_ops.end());
    t.append_scalar("joint_claim_eval_ops", joint_ops);
    if (!poly_eval_verify_plain(P.open_ops, gens.gens_combined_l_variate, r_joint_ops, joint_ops, comm_l)) return false;
    t.append_scalars("claim_evals_mem", P.eval_final);
    ScVec fin = P.eval_final; fin.resize(next_pow2(fin.size()), Sc::zero());   // DensePolynomial::new_padded
    const ScVec ch_mem = t.challenge_vector("challenge_combine_two_to_one", ceil_log2(P.eval_final.size()));
    const Sc joint_mem = bound_bot_all(fin, ch_mem);
    ScVec r_joint_mem = ch_mem; r_joint_mem.insert(r_joint_mem.end(), rand_mem.begin(), rand_mem.end());
    t.append_scalar("joint_claim_eval_mem", joint_mem);
    if (!poly_eval_verify_plain(P.open_mem, gens.gens_combined_log_m_variate, r_joint_mem, joint_mem, comm_m)) return false;
    // check_reed_solomon_fingerprints :462-523 with h(a, v, t) = t*gamma^2 + v*gamma + a - tau
    Sc init_addr = Sc::zero();   // IdentityPolynomial::evaluate identity_poly.rs:14-20
    for (size_t i = 0; i < rand_mem.size(); i++) init_addr += Sc::from_u64((uint64_t)1 << (rand_mem.size() - i - 1)) * rand_mem[i];
    const Sc g2 = gamma.square();
    auto h = [&](const Sc& a, const Sc& v, const Sc& ts) { return ts * g2 + v * gamma + a - tau; };
    for (size_t i = 0; i < alpha; i++) {
      const size_t j = S.memory_to_dimension_index(i), k = S.memory_to_subtable_index(i);
      const Sc init_memory = evaluate_subtable_mle(S, k, rand_mem);
      if (!(h(init_addr, init_memory, Sc::zero()) == claims_mem[2 * i])) return false;                                              // init
      if (!(h(P.eval_dim[j], P.eval_derefs_hash[i], P.eval_read[j]) == claims_ops[2 * i])) return false;                           // read
      if (!(h(P.eval_dim[j], P.eval_derefs_hash[i], P.eval_read[j] + Sc::one()) == claims_ops[2 * i + 1])) return false;           // write
      if (!(h(init_addr, init_memory, P.eval_final[j]) == claims_mem[2 * i + 1])) return false;                                    // final
    }
    return true;
  }
};

}  // namespace lasso

// CPU check of lasso_amd/csrc/fe29.cuh (29-bit signed-limb Fq and the point formulas built on it) against the oracle.
#include "../../lasso_amd/csrc/fe29.cuh"
#include "../../oracle/lasso_oracle.hpp"
#include <random>
#include <cstdio>
using namespace orc;
static std::mt19937_64 rng(777);
static Fq rand_fq() { u64 l[4]; for (;;) { for (int i = 0; i < 4; i++) l[i] = rng(); l[3] &= (~0ull) >> 1; if (!Fq::geq_p(l)) return Fq::from_raw(l); } }
static Fr rand_fr() { u64 l[4]; for (;;) { for (int i = 0; i < 4; i++) l[i] = rng(); l[3] &= (~0ull) >> 3; if (!Fr::geq_p(l)) return Fr::from_raw(l); } }
static fq_t fq32(const Fq& a) { u64 c[4]; a.to_canonical(c); fq_t r; memcpy(r.v, c, 32); return r; }
static bool sameq(const fq_t& a, const Fq& b) { fq_t c = fq_canonical(a); u64 e[4]; b.to_canonical(e); return memcmp(c.v, e, 32) == 0; }
static bool same29(const fe29& a, const Fq& b) { return sameq(fe_to_fq(a), b); }
#define CHECK(c) do { if (!(c)) { printf("FAIL %s line %d\n", #c, __LINE__); return 1; } } while (0)
static bool reduced(const fe29& a) { for (int i = 0; i < 9; i++) if (a.v[i] > (1 << 29) + (1 << 15) || a.v[i] < -((1 << 29) + (1 << 15))) return false; return true; }

int main() {
  std::vector<Fq> v{Fq::zero(), Fq::one(), Fq::zero() - Fq::one(), Fq::from_u64(19), Fq::from_u64(1216), Fq::zero() - Fq::from_u64(1216)};
  for (int i = 0; i < 400; i++) v.push_back(rand_fq());
  for (size_t i = 0; i < v.size(); i++) {
    const Fq &a = v[i], &b = v[(i * 17 + 5) % v.size()], &c = v[(i * 29 + 11) % v.size()];
    fe29 fa = fe_from_fq(fq32(a)), fb = fe_from_fq(fq32(b)), fc = fe_from_fq(fq32(c));
    CHECK(same29(fa, a));
    CHECK(same29(fe_add(fa, fb), a + b)); CHECK(same29(fe_sub(fa, fb), a - b)); CHECK(same29(fe_neg(fa), -a)); CHECK(same29(fe_dbl(fa), a + a));
    fe29 m = fe_mul(fa, fb); CHECK(reduced(m)); CHECK(same29(m, a * b));
    // loose x reduced, including negative loose values
    fe29 l1 = fe_sub(fa, fc), l2 = fe_add(fa, fc);
    fe29 m1 = fe_mul(l1, fb), m2 = fe_mul(l2, fb); CHECK(reduced(m1) && reduced(m2));
    CHECK(same29(m1, (a - c) * b)); CHECK(same29(m2, (a + c) * b));
    fe29 w = fe_weak(fe_sub(fe_dbl(fa), fc)); CHECK(reduced(w)); CHECK(same29(w, a + a - c));
    CHECK(same29(fe_mul(fe_mul(m1, m2), m), (a - c) * b * (a + c) * b * a * b));   // chains of reduced outputs
  }
  // worst-case magnitudes: all limbs at the loose / reduced bounds, both signs
  // (second pair of magnitudes: what the four-lane tree feeds fe_mul since round 6 — sums / differences of two product outputs whose limb 1 may carry 2^16 extra)
  for (int mag = 0; mag < 2; mag++) for (int sa = -1; sa <= 1; sa += 2) for (int sb = -1; sb <= 1; sb += 2) {
    const uint64_t ml = (1ull << 30) + (mag ? (1ull << 17) : (1ull << 16)), mr = (1ull << 29) + (mag ? (1ull << 16) : (1ull << 15));
    fe29 x, y; for (int i = 0; i < 9; i++) { x.v[i] = sa > 0 ? (int32_t)ml : -(int32_t)ml; y.v[i] = sb > 0 ? (int32_t)mr : -(int32_t)mr; }
    Fq X = Fq::zero(), Y = Fq::zero(), pw = Fq::one(), two29 = Fq::from_u64(1ull << 29);
    for (int i = 0; i < 9; i++) { Fq lx = Fq::from_u64(ml), ly = Fq::from_u64(mr); X += (sa > 0 ? lx : -lx) * pw; Y += (sb > 0 ? ly : -ly) * pw; pw *= two29; }
    fe29 m = fe_mul(x, y); CHECK(reduced(m)); CHECK(same29(m, X * Y)); CHECK(same29(x, X)); CHECK(same29(y, Y));
    // the small constant applied to a LOOSE signed operand (stage 1 of the tree): reduced, and the right value
    for (int32_t k : {121666, 243330, 243332, (1 << 18) - 1}) { const fe29 s = fe_mul_small(x, k); CHECK(reduced(s)); CHECK(same29(s, X * Fq::from_u64((uint64_t)k))); const fe29 m2 = fe_mul(x, fe_mul_small(x, k)); CHECK(reduced(m2)); CHECK(same29(m2, X * X * Fq::from_u64((uint64_t)k))); }
  }
  // group law vs oracle
  Point G = Point::generator();
  auto to29 = [](const Point& p) { pt29 e; e.X = fe_from_fq(fq32(p.X)); e.Y = fe_from_fq(fq32(p.Y)); e.T = fe_from_fq(fq32(p.T)); e.Z = fe_from_fq(fq32(p.Z)); return e; };
  auto same_pt = [&](const pt29& e, const Point& p) {
    ed_point q = pt_to_ed(e); ed_point o; o.X = fq32(p.X); o.Y = fq32(p.Y); o.T = fq32(p.T); o.Z = fq32(p.Z);
    return ed_eq(q, o) && fq_eq(fq_mul(q.T, q.Z), fq_mul(q.X, q.Y)) && reduced(e.X) && reduced(e.Y) && reduced(e.T) && reduced(e.Z); };
  std::vector<Point> pts{Point::identity(), G};
  for (int i = 0; i < 14; i++) pts.push_back(G * rand_fr());
  fe29 d2 = fe_d2();
  for (size_t i = 0; i < pts.size(); i++) {
    const Point &p = pts[i], &q = pts[(i * 5 + 2) % pts.size()];
    CHECK(same_pt(pt_add(to29(p), to29(q), d2), p + q));
    CHECK(same_pt(pt_add(to29(p), to29(p), d2), p.dbl()));
    {   // the addition split over four lanes (msm_coop_tree): stage 1 = A, B, Cs, Ds (one product + the small curve constants), stage 2 = the four coordinates; lanes emulated
      for (int rep = 0; rep < 3; rep++) {
        const pt29 pa = to29(p), pb = rep == 0 ? to29(q) : (rep == 1 ? to29(p) : pt_identity());
        fe29 st[4]; for (uint32_t c = 0; c < 4; c++) st[c] = pt_coop4_stage1(pa, pb, c);
        pt29 r; for (uint32_t c = 0; c < 4; c++) reinterpret_cast<fe29*>(&r)[c] = pt_coop4_stage2(st[0], st[1], st[2], st[3], c);
        CHECK(same_pt(r, rep == 0 ? p + q : (rep == 1 ? p.dbl() : p)));
        // round 6: stage 2 by address (what msm_coop_tree runs): the same point, every coordinate reduced, and stage 1's outputs reduced (fe_mul's second operand must be)
        pt29 r2; for (uint32_t c = 0; c < 4; c++) { CHECK(reduced(st[c])); reinterpret_cast<fe29*>(&r2)[c] = pt_coop4_stage2p(st, c); CHECK(reduced(reinterpret_cast<fe29*>(&r2)[c])); }
        CHECK(same_pt(r2, rep == 0 ? p + q : (rep == 1 ? p.dbl() : p)));
      }
    }
    {   // small-constant multiplication on reduced and on product-output operands
      const fe29 x = fe_from_fq(fq32(p.X)), y = fe_mul(x, fe_from_fq(fq32(q.Y)));
      for (int32_t k : {1, 2, 243330, 243332, (1 << 18) - 1}) { CHECK(reduced(fe_mul_small(x, k))); CHECK(same29(fe_mul_small(x, k), p.X * Fq::from_u64((uint64_t)k))); CHECK(same29(fe_mul_small(y, k), p.X * q.Y * Fq::from_u64((uint64_t)k))); }
    }
    {   // a chain of 300 cooperative additions stays reduced and correct (the tree's levels feed each other)
      pt29 acc = to29(p); Point ref = p;
      for (int k2 = 0; k2 < 300; k2++) {
        const pt29 other = (k2 & 1) ? to29(q) : acc;
        fe29 st[4]; for (uint32_t c = 0; c < 4; c++) st[c] = pt_coop4_stage1(acc, other, c);
        pt29 r; for (uint32_t c = 0; c < 4; c++) reinterpret_cast<fe29*>(&r)[c] = pt_coop4_stage2p(st, c);
        ref = (k2 & 1) ? ref + q : ref.dbl(); acc = r;
      }
      CHECK(same_pt(acc, ref));
    }
    CHECK(same_pt(pt_dbl(to29(p)), p.dbl()));
    Fq qx, qy; q.to_affine(qx, qy);
    niels29 n = niels_from_affine(fq32(qx), fq32(qy));
    CHECK(same_pt(pt_madd(to29(p), n), p + q));
    {   // round 6: the mixed addition shared by four lanes (msm_coop_leftover), both signs of the table entry, and on an accumulator that is itself a cooperative sum
      for (int neg = 0; neg < 2; neg++) {
        const niels29 e = niels_cond_neg(n, neg != 0);
        const fe29 nf[3] = {e.ypx, e.ymx, e.t2d};
        pt29 acc = to29(p); Point ref = p;
        for (int rep = 0; rep < 3; rep++) {
          fe29 st[4]; for (uint32_t c = 0; c < 4; c++) { st[c] = pt_coop4_madd_stage1(acc, nf, c); CHECK(reduced(st[c])); }
          pt29 r; for (uint32_t c = 0; c < 4; c++) reinterpret_cast<fe29*>(&r)[c] = pt_coop4_stage2p(st, c);
          ref = neg ? ref - q : ref + q; acc = r;
          CHECK(same_pt(acc, ref));
        }
      }
    }
    // long chain: 200 mixed adds then doublings stay reduced and correct
    pt29 acc = to29(p); Point ref = p;
    for (int k = 0; k < 200; k++) { acc = pt_madd(acc, n); ref = ref + q; }
    for (int k = 0; k < 20; k++) { acc = pt_dbl(acc); ref = ref.dbl(); }
    CHECK(same_pt(acc, ref));
  }
  printf("OK\n");
  return 0;
}

// CPU check of the BN254 build of the product's arithmetic headers (lasso_amd/csrc/bn254_*.cuh + mont29.cuh, selected by -DLASSO_BN254)
// against the oracle's independent 64-bit arithmetic and Jacobian group law (oracle/ff.hpp, bn254.hpp under -DORC_BN254).
// Test-only: links oracle code as the checker.
#include "../../lasso_amd/csrc/fe29.cuh"
#include "../../oracle/lasso_oracle.hpp"
#include <random>
#include <cstdio>
#include <array>
using namespace orc;

static std::mt19937_64 rng(777);
static Fr rand_fr() { u64 l[4]; for (;;) { for (int i = 0; i < 4; i++) l[i] = rng(); l[3] &= (~0ull) >> 2; if (!Fr::geq_p(l)) return Fr::from_raw(l); } }
static Fq rand_fq() { u64 l[4]; for (;;) { for (int i = 0; i < 4; i++) l[i] = rng(); l[3] &= (~0ull) >> 2; if (!Fq::geq_p(l)) return Fq::from_raw(l); } }
static fr_t r32(const Fr& a) { fr_t r; memcpy(r.v, a.v, 32); return r; }
static fq_t q32(const Fq& a) { fq_t r; memcpy(r.v, a.v, 32); return r; }
static bool same(const fr_t& a, const Fr& b) { return memcmp(a.v, b.v, 32) == 0; }
static bool same(const fq_t& a, const Fq& b) { return memcmp(a.v, b.v, 32) == 0; }
#define CHECK(c) do { if (!(c)) { printf("FAIL %s line %d\n", #c, __LINE__); return 1; } } while (0)

static Point from_ed(const ed_point& e) {   // (X : Y : Z) -> oracle point
  Fq X = Fq::from_raw((const u64*)e.X.v), Y = Fq::from_raw((const u64*)e.Y.v), Z = Fq::from_raw((const u64*)e.Z.v);
  if (Z.is_zero()) return Point::identity();
  Fq zi = Z.inverse(); return Point::from_affine(X * zi, Y * zi);
}
static ed_point to_ed(const Point& p, const Fq& scale) {   // a non-trivial projective representative
  if (p.is_identity()) { ed_point e = ed_identity(); e.Y = q32(scale); return e; }
  Fq x, y; p.to_affine(x, y); ed_point e; e.X = q32(x * scale); e.Y = q32(y * scale); e.Z = q32(scale); e.T = fq_zero(); return e;
}
static niels29 to_niels(const Point& p) { Fq x, y; p.to_affine(x, y); return niels_from_affine(q32(x), q32(y)); }

int main() {
  std::vector<Fr> frs{Fr::zero(), Fr::one(), Fr::zero() - Fr::one(), Fr::from_u64(2), Fr::from_u64(~0ull)};
  for (int i = 0; i < 300; i++) frs.push_back(rand_fr());
  for (size_t i = 0; i < frs.size(); i++) {
    const Fr &a = frs[i], &b = frs[(i * 13 + 7) % frs.size()];
    CHECK(same(fr_mul(r32(a), r32(b)), a * b)); CHECK(same(fr_add(r32(a), r32(b)), a + b)); CHECK(same(fr_sub(r32(a), r32(b)), a - b)); CHECK(same(fr_neg(r32(a)), -a));
    u64 c[4]; a.to_canonical(c); fr_t cc = fr_to_canonical(r32(a)); CHECK(memcmp(cc.v, c, 32) == 0);
    CHECK(same(fr_from_canonical(cc), a)); CHECK(fr_canonical_bits(cc) == a.num_bits());
    if (i < 20 && !a.is_zero()) CHECK(same(fr_inv(r32(a)), a.inverse()));
  }
  CHECK(same(fr_one(), Fr::one())); CHECK(same(fr_from_u64(123456789012345ull), Fr::from_u64(123456789012345ull)));
  { fr_t big; for (int i = 0; i < 8; i++) big.v[i] = 0xffffffffu; u64 l[4] = {~0ull, ~0ull, ~0ull, ~0ull}; CHECK(same(fr_from_canonical(big), Fr::from_canonical(l))); }

  std::vector<Fq> fqs{Fq::zero(), Fq::one(), Fq::zero() - Fq::one(), Fq::from_u64(3), Fq::from_u64(9)};
  for (int i = 0; i < 300; i++) fqs.push_back(rand_fq());
  for (size_t i = 0; i < fqs.size(); i++) {
    const Fq &a = fqs[i], &b = fqs[(i * 11 + 3) % fqs.size()];
    CHECK(same(fq_mul(q32(a), q32(b)), a * b)); CHECK(same(fq_add(q32(a), q32(b)), a + b)); CHECK(same(fq_sub(q32(a), q32(b)), a - b)); CHECK(same(fq_neg(q32(a)), -a));
    CHECK(same(fq_mul9(q32(a)), a * Fq::from_u64(9))); CHECK(same(fq_mul3(q32(a)), a * Fq::from_u64(3)));
    if (i < 10 && !a.is_zero()) CHECK(same(fq_inv(q32(a)), a.inverse()));
    // 29-bit-limb form: round trip, products, sums, small multiples, plain integer
    const fe29 ea = fe_from_fq(q32(a)), eb = fe_from_fq(q32(b));
    CHECK(same(fe_to_fq(ea), a));
    CHECK(same(fe_to_fq(fe_mul(ea, eb)), a * b));
    CHECK(same(fe_to_fq(fe_mul(fe_sub(ea, eb), fe_weak(fe_add(ea, eb)))), (a - b) * (a + b)));
    CHECK(same(fe_to_fq(fe_small(fe_sub(fe_sub(ea, eb), eb), 9)), (a - b - b) * Fq::from_u64(9)));
    CHECK(same(fe_to_fq(fe_x3(ea)), a * Fq::from_u64(3)));
    CHECK(same(fe_to_fq(fe_mul(ea, fe_one())), a));
    { uint32_t w[8]; fe_to_plain_words(ea, w); u64 c[4]; a.to_canonical(c); CHECK(memcmp(w, c, 32) == 0); }
    if (i < 6 && !a.is_zero()) CHECK(same(fe_to_fq(fe_inv_chain(ea)), a.inverse()));
  }

  // group law: 8x32 form (host tails) and 29-bit form (kernels) vs the oracle's Jacobian law
  Point G = Point::generator();
  std::vector<Point> pts{Point::identity(), G, G.neg()};
  for (int i = 0; i < 14; i++) pts.push_back(G * rand_fr());
  const fe29 dummy = fe_d2();
  for (size_t i = 0; i < pts.size(); i++) {
    const Point &p = pts[i], &q = pts[(i * 5 + 2) % pts.size()];
    const ed_point ep = to_ed(p, rand_fq()), eq = to_ed(q, rand_fq());
    CHECK(from_ed(ed_add(ep, eq)) == p + q); CHECK(from_ed(ed_add(ep, ep)) == p.dbl()); CHECK(from_ed(ed_dbl(ep)) == p.dbl());
    CHECK(from_ed(ed_add(ep, ed_neg(ep))).is_identity()); CHECK(ed_eq(ep, to_ed(p, rand_fq()))); CHECK(p == q || !ed_eq(ep, eq));
    Fr k = rand_fr(); u64 kc[4]; k.to_canonical(kc);
    CHECK(from_ed(ed_mul_limbs(ep, (const uint32_t*)kc)) == p * k);
    const pt29 pp = pt_from_ed(ep), pq = pt_from_ed(eq);
    CHECK(from_ed(pt_to_ed(pp)) == p);
    CHECK(from_ed(pt_to_ed(pt_add(pp, pq, dummy))) == p + q); CHECK(from_ed(pt_to_ed(pt_add(pp, pp, dummy))) == p.dbl()); CHECK(from_ed(pt_to_ed(pt_dbl(pp))) == p.dbl());
    CHECK(from_ed(pt_to_abi(pt_add(pp, pt_identity(), dummy))) == p); CHECK(from_ed(pt_to_abi(pt_add(pt_identity(), pt_identity(), dummy))).is_identity());
    {   // the six-lane split of the full addition (msm_coop_tree), lanes emulated in a loop
      fe29 m[6], pr[6];
      for (uint32_t c = 0; c < 6; c++) m[c] = pt_coop_layer1(pp, pq, c);
      for (uint32_t c = 0; c < 6; c++) pr[c] = pt_coop_layer2(m, c);
      pt29 r; r.X = pt_coop_out(pr[0], pr[1], 0); r.Y = pt_coop_out(pr[2], pr[3], 1); r.Z = pt_coop_out(pr[4], pr[5], 2); r.T = fe_zero();
      CHECK(from_ed(pt_to_ed(r)) == p + q);
      for (uint32_t c = 0; c < 6; c++) m[c] = pt_coop_layer1(pp, pp, c);
      for (uint32_t c = 0; c < 6; c++) pr[c] = pt_coop_layer2(m, c);
      r.X = pt_coop_out(pr[0], pr[1], 0); r.Y = pt_coop_out(pr[2], pr[3], 1); r.Z = pt_coop_out(pr[4], pr[5], 2);
      CHECK(from_ed(pt_to_ed(r)) == p.dbl());
      // round 4's pass: the linear step as a lane step of its own (pt_coop_form), operands of the second layer read by index (pt_coop_prod2) — the same limbs as above
      fe29 f[6], pr2[6];
      for (uint32_t c = 0; c < 6; c++) m[c] = pt_coop_layer1(pp, pq, c);
      for (uint32_t c = 0; c < 6; c++) pr[c] = pt_coop_layer2(m, c);
      for (uint32_t c = 0; c < 6; c++) f[c] = pt_coop_form(m, c);
      for (uint32_t c = 0; c < 6; c++) { for (int k = 0; k < 8; k++) CHECK(f[c].v[k] >= 0 && f[c].v[k] < (1 << 29)); CHECK(f[c].v[8] > -(1 << 28) && f[c].v[8] < (1 << 28)); }
      for (uint32_t c = 0; c < 6; c++) pr2[c] = pt_coop_prod2(f, c);
      for (uint32_t c = 0; c < 6; c++) CHECK(memcmp(&pr[c], &pr2[c], sizeof(fe29)) == 0);
      pt29 r2; r2.X = pt_coop_out(pr2[0], pr2[1], 0); r2.Y = pt_coop_out(pr2[2], pr2[3], 1); r2.Z = pt_coop_out(pr2[4], pr2[5], 2); r2.T = fe_zero();
      CHECK(from_ed(pt_to_ed(r2)) == p + q);
    }
    if (!q.is_identity()) {
      const niels29 n = to_niels(q);
      CHECK(from_ed(pt_to_ed(pt_madd(pp, n))) == p + q);
      CHECK(from_ed(pt_to_ed(pt_madd(pp, niels_cond_neg(n, true)))) == p - q);
      CHECK(from_ed(pt_to_ed(pt_madd(pt_identity(), n))) == q);
      CHECK(from_ed(pt_to_ed(pt_madd(pt_from_ed(to_ed(q, rand_fq())), n))) == q.dbl());                       // P + P through the mixed addition
      CHECK(from_ed(pt_to_ed(pt_madd(pt_from_ed(to_ed(q.neg(), rand_fq())), n))).is_identity());              // P + (-P)
    }
    uint8_t want[32]; p.compress(want); uint32_t got[8]; pt_compress(pp, got); CHECK(memcmp(want, got, 32) == 0);
  }
  // long dependent chains (the shape of a bucket accumulation followed by a tree and a Horner tail): magnitudes must stay in bounds
  {
    pt29 acc = pt_identity(); Point ref = Point::identity();
    for (int i = 0; i < 600; i++) {
      const Point& q = pts[1 + (rng() % (pts.size() - 1))];
      const int op = (int)(rng() % 4);
      if (op == 0) { acc = pt_madd(acc, to_niels(q)); ref = ref + q; }
      else if (op == 1) { acc = pt_madd(acc, niels_cond_neg(to_niels(q), true)); ref = ref - q; }
      else if (op == 2 && (i & 1)) { acc = pt_add(acc, pt_from_ed(to_ed(q, rand_fq())), dummy); ref = ref + q; }
      else if (op == 2) {
        const pt29 pq2 = pt_from_ed(to_ed(q, rand_fq())); fe29 m[6], pr[6];
        fe29 f[6];
        for (uint32_t c = 0; c < 6; c++) m[c] = pt_coop_layer1(acc, pq2, c);
        for (uint32_t c = 0; c < 6; c++) f[c] = pt_coop_form(m, c);
        for (uint32_t c = 0; c < 6; c++) pr[c] = pt_coop_prod2(f, c);
        acc.X = pt_coop_out(pr[0], pr[1], 0); acc.Y = pt_coop_out(pr[2], pr[3], 1); acc.Z = pt_coop_out(pr[4], pr[5], 2); ref = ref + q;
      }
      else { acc = pt_dbl(acc); ref = ref.dbl(); }
      for (int k = 0; k < 8; k++) { CHECK(acc.X.v[k] >= 0 && acc.X.v[k] < (1 << 29)); CHECK(acc.Z.v[k] >= 0 && acc.Z.v[k] < (1 << 29)); }
      CHECK(acc.X.v[8] > -(1 << 26) && acc.X.v[8] < (1 << 26) && acc.Y.v[8] > -(1 << 26) && acc.Y.v[8] < (1 << 26) && acc.Z.v[8] > -(1 << 26) && acc.Z.v[8] < (1 << 26));
      if (i % 50 == 49) CHECK(from_ed(pt_to_ed(acc)) == ref);
    }
    uint8_t want[32]; ref.compress(want); uint32_t got[8]; pt_compress(acc, got); CHECK(memcmp(want, got, 32) == 0);
  }
  // the cooperative tree's schedule (msm_kernels.cuh msm_coop_tree, BN254 build), lanes and barriers emulated: 256 threads in groups of six, 42
  // additions per pass, one exchange buffer used twice per pass; pts[0..live) -> pts[0]
  for (uint32_t live : {1u, 2u, 3u, 5u, 42u, 43u, 85u, 100u, 255u, 256u}) {
    const uint32_t NT = 256, GROUPS = NT / 6;
    std::vector<pt29> v(NT); std::vector<fe29> st(NT); Point ref = Point::identity();
    for (uint32_t i = 0; i < NT; i++) { const Point& q = pts[rng() % pts.size()]; v[i] = pt_from_ed(to_ed(q, rand_fq())); if (i < live) ref = ref + q; }
    uint32_t p2 = 1; while (p2 < live) p2 <<= 1;
    for (uint32_t sft = p2 >> 1; sft > 0; sft >>= 1) {
      for (uint32_t i0 = 0; i0 < sft; i0 += GROUPS) {
        std::vector<std::array<fe29, 6>> regs(NT);
        auto act = [&](uint32_t t) { const uint32_t g = t / 6, i = i0 + g; return g < GROUPS && i < sft && i + sft < live; };
        for (uint32_t t = 0; t < NT; t++) if (act(t)) { const uint32_t g = t / 6, c = t - g * 6, i = i0 + g; st[g * 6 + c] = pt_coop_layer1(v[i], v[i + sft], c); }
        for (uint32_t t = 0; t < NT; t++) if (act(t)) { const uint32_t g = t / 6; for (int k = 0; k < 6; k++) regs[t][k] = st[g * 6 + k]; }
        for (uint32_t t = 0; t < NT; t++) if (act(t)) { const uint32_t g = t / 6, c = t - g * 6; st[g * 6 + c] = pt_coop_layer2(regs[t].data(), c); }
        for (uint32_t t = 0; t < NT; t++) if (act(t)) { const uint32_t g = t / 6, c = t - g * 6, i = i0 + g; if (c < 3) reinterpret_cast<fe29*>(&v[i])[c == 2 ? 3 : c] = pt_coop_out(st[g * 6 + 2 * c], st[g * 6 + 2 * c + 1], c); }
      }
    }
    CHECK(from_ed(pt_to_ed(v[0])) == ref);
  }
  printf("OK\n");
  return 0;
}

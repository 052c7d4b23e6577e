// CPU check of the product's __host__ __device__ arithmetic headers (lasso_amd/csrc/fr.cuh, fq.cuh)
// against the oracle (oracle/ff.hpp, ed25519.hpp).  Test-only: links oracle code as the checker.
#include "../../lasso_amd/csrc/fq.cuh"
#include "../../oracle/lasso_oracle.hpp"
#include <random>
#include <cstdio>
using namespace orc;

static std::mt19937_64 rng(12345);
static Fr rand_fr() { u64 l[4]; for (;;) { for (int i = 0; i < 4; i++) l[i] = rng(); l[3] &= (~0ull) >> 3; if (!Fr::geq_p(l)) return Fr::from_raw(l); } }
static Fq rand_fq() { u64 l[4]; for (;;) { for (int i = 0; i < 4; i++) l[i] = rng(); l[3] &= (~0ull) >> 1; if (!Fq::geq_p(l)) return Fq::from_raw(l); } }
static fr_t to32(const Fr& a) { fr_t r; memcpy(r.v, a.v, 32); return r; }
static bool same(const fr_t& a, const Fr& b) { return memcmp(a.v, b.v, 32) == 0; }
// product Fq is plain+lazy; oracle Fq is Montgomery: compare through canonical integers
static fq_t fq32(const Fq& a) { u64 c[4]; a.to_canonical(c); fq_t r; memcpy(r.v, c, 32); return r; }
static bool sameq(const fq_t& a, const Fq& b) { fq_t c = fq_canonical(a); u64 e[4]; b.to_canonical(e); return memcmp(c.v, e, 32) == 0; }
static fq_t lazy(const fq_t& a, int k) {  // add k*p to make a non-canonical representative (still < 2^256)
  fq_t r = a; if (k == 0) return r;
  // canonical a < p = 2^255-19, so a + p < 2^256
  uint64_t c = 0; const uint32_t P[8] = {0xffffffedu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x7fffffffu};
  for (int i = 0; i < 8; i++) { c += (uint64_t)r.v[i] + P[i]; r.v[i] = (uint32_t)c; c >>= 32; }
  return r;
}
#define CHECK(c) do { if (!(c)) { printf("FAIL %s line %d\n", #c, __LINE__); return 1; } } while (0)

int main() {
  std::vector<Fr> frs{Fr::zero(), Fr::one(), Fr::zero() - Fr::one(), Fr::from_u64(2), Fr::from_u64(0xffffffffffffffffull)};
  for (int i = 0; i < 300; i++) frs.push_back(rand_fr());
  for (size_t i = 0; i < frs.size(); i++) {
    const Fr &a = frs[i], &b = frs[(i * 13 + 7) % frs.size()];
    CHECK(same(fr_mul(to32(a), to32(b)), a * b));
    CHECK(same(fr_add(to32(a), to32(b)), a + b));
    CHECK(same(fr_sub(to32(a), to32(b)), a - b));
    CHECK(same(fr_neg(to32(a)), -a));
    u64 c[4]; a.to_canonical(c); fr_t cc = fr_to_canonical(to32(a)); CHECK(memcmp(cc.v, c, 32) == 0);
    CHECK(same(fr_from_canonical(cc), a));
    CHECK(fr_canonical_bits(cc) == a.num_bits());
    if (i < 20 && !a.is_zero()) CHECK(same(fr_inv(to32(a)), a.inverse()));
  }
  CHECK(same(fr_one(), Fr::one())); CHECK(same(fr_from_u64(123456789012345ull), Fr::from_u64(123456789012345ull)));
  { fr_t big; for (int i = 0; i < 8; i++) big.v[i] = 0xffffffffu; u64 l[4] = {~0ull, ~0ull, ~0ull, ~0ull}; CHECK(same(fr_from_canonical(big), Fr::from_canonical(l))); }

  std::vector<Fq> fqs{Fq::zero(), Fq::one(), Fq::zero() - Fq::one(), Fq::from_u64(19), Fq::from_u64(38)};
  for (int i = 0; i < 300; i++) fqs.push_back(rand_fq());
  for (size_t i = 0; i < fqs.size(); i++) {
    const Fq &a = fqs[i], &b = fqs[(i * 11 + 3) % fqs.size()];
    for (int ka = 0; ka < 2; ka++) for (int kb = 0; kb < 2; kb++) {
      fq_t x = lazy(fq32(a), ka), y = lazy(fq32(b), kb);
      CHECK(sameq(fq_mul(x, y), a * b));
      CHECK(sameq(fq_add(x, y), a + b));
      CHECK(sameq(fq_sub(x, y), a - b));
      CHECK(sameq(fq_neg(x), -a));
    }
    fq_t m; memcpy(m.v, a.v, 32);  // oracle in-memory form IS ark's Montgomery form
    CHECK(sameq(fq_from_mont(m), a));
    fq_t back = fq_to_mont(fq32(a)); CHECK(memcmp(back.v, a.v, 32) == 0);
    if (i < 10 && !a.is_zero()) CHECK(sameq(fq_inv(fq32(a)), a.inverse()));
  }
  // extreme lazy values
  { fq_t mx; for (int i = 0; i < 8; i++) mx.v[i] = 0xffffffffu; Fq v = Fq::from_u64(37);  // 2^256-1 = 38-1 mod p
    CHECK(sameq(mx, v)); CHECK(sameq(fq_add(mx, mx), v + v)); CHECK(sameq(fq_mul(mx, mx), v * v)); CHECK(sameq(fq_sub(fq_zero(), mx), -v));
    fq_t one = fq_one(); CHECK(sameq(fq_sub(one, mx), Fq::one() - v)); CHECK(sameq(fq_add(mx, one), v + Fq::one())); }

  // group law vs oracle
  Point G = Point::generator();
  auto to_ed = [](const Point& p) { ed_point e; e.X = fq32(p.X); e.Y = fq32(p.Y); e.T = fq32(p.T); e.Z = fq32(p.Z); return e; };
  auto same_pt = [&](const ed_point& e, const Point& p) { ed_point o = to_ed(p); return ed_eq(e, o) && fq_eq(fq_mul(e.T, e.Z), fq_mul(e.X, e.Y)); };
  std::vector<Point> pts{Point::identity(), G};
  for (int i = 0; i < 12; i++) pts.push_back(G * rand_fr());
  for (size_t i = 0; i < pts.size(); i++) {
    const Point &p = pts[i], &q = pts[(i * 5 + 2) % pts.size()];
    CHECK(same_pt(ed_add(to_ed(p), to_ed(q)), p + q));
    CHECK(same_pt(ed_add(to_ed(p), to_ed(p)), p.dbl()));
    CHECK(same_pt(ed_dbl(to_ed(p)), p.dbl()));
    Fq qx, qy; q.to_affine(qx, qy);
    ed_niels n = ed_to_niels_affine(fq32(qx), fq32(qy));
    CHECK(same_pt(ed_madd(to_ed(p), n), p + q));
    CHECK(same_pt(ed_msub(to_ed(p), n), p - q));
    Fr k = rand_fr(); u64 kc[4]; k.to_canonical(kc);
    CHECK(same_pt(ed_mul_limbs(to_ed(p), (const uint32_t*)kc), p * k));
  }
  printf("OK\n");
  return 0;
}

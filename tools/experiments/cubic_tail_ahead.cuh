// EXPERIMENT, not part of the product (round 3; measured slower than k_cubic_tail, see DESIGN.md 7.9 and profiles/r03_tail_ahead_experiment.txt): the resident sumcheck tail
// one round ahead of the host.  Included by tools/tail_phase_bench.hip right after poly_kernels.cuh (it uses that file's helpers and its TAIL_EQ_S macro, re-stated here).
#pragma once
#define TAIL_EQ_S(idx) (EQI ? fr29_mul(eq_hi[(idx) >> elb], eq_lo[(idx) & ((1u << elb) - 1u)]) : fr29_unpack_s(E[(idx)]))

// The same tail, one round AHEAD of the host (round 3).  In k_cubic_tail the challenge is followed by bind -> products -> reduction -> publish before the host can go on:
// 5.3 us of device work per turn on the critical path.  But the next round's two sums are QUADRATICS in the coming challenge r: with arrays A, B of 2h entries and
// G[i] = A[i] * E[i mod h/2] the bound arrays are X'[j] = X[j] + r (X[j+h] - X[j]), so (n = h/2 pairs of the next round, D = "upper half minus lower half" inside a bound array)
//   q'(0)  = sum_{j<n} B'[j] G'[j]       = V0 + r (V1 - V0 - Vi) + r^2 Vi     V0 = sum B[j] G[j],  V1 = sum B[j+h] G[j+h],  Vi = sum (B[j+h]-B[j]) (G[j+h]-G[j])
//   q'_inf = sum_{j<n} DB'[j] DG'[j]     = W0 + r (W1 - W0 - Wi) + r^2 Wi     W0 = sum DB[j] DG[j] (lower halves),  W1 = the same over the upper halves,  Wi = sum (DB_up - DB_lo)(DG_up - DG_lo)
// Six sums that need no challenge: they are formed while the previous answer travels to the host and the host works on its transcript.  When the challenge arrives two lanes
// evaluate the quadratics (two products each) and publish; everybody else binds and forms the next six sums in the shadow of that hand-off.  Same results, same sequence
// numbers, same mailbox as k_cubic_tail (every sum is an exact field sum).  Lane tasks are dealt from the LAST lane down, so that wave 0 (poll, evaluate, publish) has none at small h.
// LDS at Q = 512: arrays 74 KB + one work area of 55 KB (G while the terms are formed, then the 6 x h/2 term rows) + strips: 134 KB.
template <bool BIND, int Q, bool EQI = false>
__global__ void __launch_bounds__(Q) k_cubic_tail_ahead(MutPtrTable A, MutPtrTable B, const fr_t* __restrict__ E, uint32_t q, fr_t r0, const uint32_t* mailbox, uint32_t* counters,
                                                                   fr_t* __restrict__ out, uint32_t* flag, uint32_t seq0, EqInline EQ = EqInline()) {
  __shared__ fr29 eq_hi[EQI ? 32 : 1], eq_lo[EQI ? 32 : 1];
  const uint32_t elb = EQ.ell / 2;
  if (EQI) {
    const uint32_t tt = threadIdx.x, hb = EQ.ell - elb; const fr29 one_s = fr29_one_s();
    if (tt < (1u << hb)) { fr29 p = fr29_unpack_s(EQ.scale); for (uint32_t j = 0; j < hb; j++) { const bool bit = (tt >> (hb - 1 - j)) & 1u; const fr29 rs = fr29_unpack_s(EQ.r[j]); p = fr29_mul(bit ? rs : fr29_sub(one_s, rs), p); } eq_hi[tt] = p; }
    else if (tt >= 64 && tt - 64 < (1u << elb)) { const uint32_t x = tt - 64; fr29 p = one_s; for (uint32_t j = 0; j < elb; j++) { const bool bit = (x >> (elb - 1 - j)) & 1u; const fr29 rs = fr29_unpack_s(EQ.r[hb + j]); p = fr29_mul(bit ? rs : fr29_sub(one_s, rs), p); } eq_lo[x] = p; }
    __syncthreads();
  }
  constexpr uint32_t NS = Q >= 512 ? 8 : 4;   // strips per column of the reductions (NS * 54 threads)
  __shared__ fr29 bound[2][2 * Q];            // A, B of the current round (2h values each)
  __shared__ int32_t work[3 * Q * 9];         // G[2h] while terms are formed, then the term rows
  __shared__ int64_t strips[NS * 54];
  __shared__ int64_t cols[54];
  __shared__ fr29 coef[6];
  __shared__ fr_t chal;
  __shared__ uint32_t alive;
  fr29* const G = reinterpret_cast<fr29*>(work);
  int32_t* const rows = work;
  const uint32_t t = threadIdx.x, tw = Q - 1 - t, y = blockIdx.x, ncirc = gridDim.x;
  const uint64_t t_end = wall_clock64() + 500000000ull;   // 5 s at 100 MHz
  // column sums of nsums groups of R term rows each (row (s, i) at (s * R + i) * 9): cols[s * 9 + k]; all threads, ends with a barrier
  auto column_sums = [&](uint32_t nsums, uint32_t R) {
    const uint32_t NC = nsums * 9;
    if (R > 16) {
      if (t < NS * NC) {
        const uint32_t col = t % NC, strip = t / NC, sm = col / 9, k = col - sm * 9;
        const uint32_t per = (R + NS - 1) / NS, i0 = strip * per, i1 = i0 + per < R ? i0 + per : R;
        int64_t sum = 0;
        for (uint32_t i = i0; i < i1; i++) sum += rows[(sm * R + i) * 9 + k];
        strips[strip * NC + col] = sum;
      }
      __syncthreads();
      if (t < NC) { int64_t sum = 0; for (uint32_t g = 0; g < NS; g++) sum += strips[g * NC + t]; cols[t] = sum; }
    } else if (t < NC) {
      const uint32_t sm = t / 9, k = t - sm * 9;
      int64_t sum = 0;
      for (uint32_t i = 0; i < R; i++) sum += rows[(sm * R + i) * 9 + k];
      cols[t] = sum;
    }
    __syncthreads();
  };
  uint32_t h = q;   // pairs of the current round; arrays hold 2h values
  {
    // load (or bind the previous layer round's challenge) and weight A with the eq table: G[i] = A[i] * E[i mod h]
    const fr29 rs = fr29_unpack_s(r0);
    const uint32_t m = 2 * h;
    for (uint32_t item = tw; item < 2 * m; item += Q) {
      const uint32_t p = item / m, i = item - p * m;
      const fr_t* src = p == 0 ? A.p[y] : B.p[y];
      const fr29 v = BIND ? bind29(src[i], src[i + m], rs) : fr29_unpack_u(src[i]);
      bound[p][i] = v;
      if (p == 0) G[i] = fr29_mul(v, TAIL_EQ_S(i < h ? i : i - h));
    }
    __syncthreads();
    // this round's two sums directly (its challenge-free form): 2h terms, at most two per lane, parked in registers until every lane has read G
    fr29 term[2];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const uint32_t u = tw + pass * Q;
      if (u < 2 * h) {
        const uint32_t v = u >= h ? 1u : 0u, i = u - v * h;
        const fr29 g0 = G[i], g1 = G[i + h], b0 = bound[1][i], b1 = bound[1][i + h];
        const fr29 dg = fr29_sub(g1, g0), db = fr29_sub(b1, b0);
        fr29 fa, fb;
#pragma unroll
        for (int k = 0; k < 9; k++) { fa.v[k] = v ? dg.v[k] : b0.v[k]; fb.v[k] = v ? db.v[k] : g0.v[k]; }
        term[pass] = fr29_mul(fa, fb);
      }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const uint32_t u = tw + pass * Q;
      if (u < 2 * h) {
#pragma unroll
        for (int k = 0; k < 9; k++) rows[u * 9 + k] = term[pass].v[k];
      }
    }
    __syncthreads();
    column_sums(2, h);
    if (t < 2) {
      int64_t c[9];
#pragma unroll
      for (int k = 0; k < 9; k++) c[k] = cols[t * 9 + k];
      result_store(out, (size_t)y * 2 + t, fr29_pack(fr29_reduce_columns(c, 5)), flag, seq0);
    }
    row_done(ncirc, counters, flag, seq0);
  }
  for (uint32_t turn = 0;; turn++) {
    const uint32_t n = h / 2;   // pairs of the NEXT round
    TAIL_STAMP(0);
    if (n) {
      // ---- the next round's sums as quadratics in the challenge that has not arrived yet (the work area is free: column_sums and the bind both end with a barrier)
      for (uint32_t i = tw; i < 2 * h; i += Q) G[i] = fr29_mul(bound[0][i], TAIL_EQ_S(i & (n - 1u)));
      __syncthreads();
      fr29 term[3];
#pragma unroll
      for (int pass = 0; pass < 3; pass++) {
        const uint32_t u = tw + pass * Q;
        if (u < 6 * n) {
          const uint32_t ty = u / n, j = u - ty * n;
          // a = sum_k c_k B[i_k], b = sum_k c_k G[i_k] over i = (j, j+n, j+h, j+h+n) with the sign pattern of the sum `ty`
          const int32_t c0 = (ty == 0 || ty == 5) ? 1 : (ty == 2 || ty == 3) ? -1 : 0;
          const int32_t c1 = ty == 3 ? 1 : ty == 5 ? -1 : 0;
          const int32_t c2 = (ty == 1 || ty == 2) ? 1 : (ty == 4 || ty == 5) ? -1 : 0;
          const int32_t c3 = (ty == 4 || ty == 5) ? 1 : 0;
          fr29 fa = fr29_zero(), fb = fr29_zero();
          if (c0) { const fr29 bv = bound[1][j], gv = G[j];
#pragma unroll
            for (int k = 0; k < 9; k++) { fa.v[k] += c0 * bv.v[k]; fb.v[k] += c0 * gv.v[k]; } }
          if (c1) { const fr29 bv = bound[1][j + n], gv = G[j + n];
#pragma unroll
            for (int k = 0; k < 9; k++) { fa.v[k] += c1 * bv.v[k]; fb.v[k] += c1 * gv.v[k]; } }
          if (c2) { const fr29 bv = bound[1][j + h], gv = G[j + h];
#pragma unroll
            for (int k = 0; k < 9; k++) { fa.v[k] += c2 * bv.v[k]; fb.v[k] += c2 * gv.v[k]; } }
          if (c3) { const fr29 bv = bound[1][j + h + n], gv = G[j + h + n];
#pragma unroll
            for (int k = 0; k < 9; k++) { fa.v[k] += c3 * bv.v[k]; fb.v[k] += c3 * gv.v[k]; } }
          term[pass] = fr29_mul(fa, fr29_weak(fb));   // |fa limb| < 2^30 (four canonical values), fb carried back to reduced limbs
        }
      }
      __syncthreads();   // every lane has read G: the work area becomes the term rows
#pragma unroll
      for (int pass = 0; pass < 3; pass++) {
        const uint32_t u = tw + pass * Q;
        if (u < 6 * n) {
#pragma unroll
          for (int k = 0; k < 9; k++) rows[u * 9 + k] = term[pass].v[k];
        }
      }
      __syncthreads();
      TAIL_STAMP(1);
      column_sums(6, n);
      if (t >= 64 && t < 70) {   // six lanes of wave 1: the sums as canonical values (still 2^5 short, like the terms)
        int64_t c[9];
#pragma unroll
        for (int k = 0; k < 9; k++) c[k] = cols[(t - 64) * 9 + k];
        coef[t - 64] = fr29_reduce_columns(c, 0);
      }
    }
    TAIL_STAMP(2);
    // ---- the host's answer to the last publication: this round's challenge
    if (t == 0) {
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const u32x4* m4 = reinterpret_cast<const u32x4*>(mailbox);
      uint32_t ok = 1; u32x4 c0, c1, c2; uint32_t spins = 0;
      for (;;) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        c0 = __builtin_nontemporal_load(m4); c1 = __builtin_nontemporal_load(m4 + 1); c2 = __builtin_nontemporal_load(m4 + 2);
        if (c0.x == seq0 + turn + 1 && c1.x == seq0 + turn + 1 && c2.x == seq0 + turn + 1 && c2.w == (c0.y ^ c0.z ^ c0.w ^ c1.y ^ c1.z ^ c1.w ^ c2.y ^ c2.z) + (seq0 + turn + 1) * 0x9E3779B9u) break;
        if (c0.x == LASSO_MAIL_POISON || ((++spins & 63u) == 0 && wall_clock64() > t_end)) { ok = 0; break; }
      }
      if (ok) { chal.v[0] = c0.y; chal.v[1] = c0.z; chal.v[2] = c0.w; chal.v[3] = c1.y; chal.v[4] = c1.z; chal.v[5] = c1.w; chal.v[6] = c2.y; chal.v[7] = c2.z; }
      alive = ok;
    }
    __syncthreads();
    if (!alive) return;
    TAIL_STAMP(3);
    const fr29 rs = fr29_unpack_s(chal);
    if (n && t < 2) {   // the next round's sums at the challenge: c0 + r ((c1 - c0 - ci) + r ci), then the 2^5 the terms are short of
      const fr29 c0 = coef[3 * t], c1 = coef[3 * t + 1], ci = coef[3 * t + 2];
      const fr29 mid = fr29_sub(fr29_sub(c1, c0), ci);
      const fr29 inner = fr29_add(mid, fr29_mul(ci, rs));      // limbs within (-2^30, 2^30)
      const fr29 val = fr29_add(c0, fr29_mul(inner, rs));
      int64_t c[9];
#pragma unroll
      for (int k = 0; k < 9; k++) c[k] = val.v[k];
      result_store(out, (size_t)y * 2 + t, fr29_pack(fr29_reduce_columns(c, 5)), flag, seq0 + turn + 1);
    }
    if (n) row_done(ncirc, counters, flag, seq0 + turn + 1);
    TAIL_STAMP(4);
    // ---- bind in place: task (p, j) reads j and j + h of array p and writes j
    for (uint32_t u = tw; u < 2 * h; u += Q) {
      const uint32_t p = u >= h ? 1u : 0u, j = u - p * h;
      const fr29 lo = bound[p][j];
      bound[p][j] = fr29_canonical(fr29_add(lo, fr29_mul(fr29_sub(bound[p][j + h], lo), rs)));
    }
    __syncthreads();
    TAIL_STAMP(5);
    if (n == 0) {   // h was 1: the layer is bound to a point
      if (t < 2) result_store(out, (size_t)t * ncirc + y, fr29_pack(bound[t][0]), flag, seq0 + turn + 1);
      row_done(ncirc, counters, flag, seq0 + turn + 1);
      return;
    }
    h = n;
  }
}
#undef TAIL_EQ_S

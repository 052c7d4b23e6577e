// ORACLE — TEST INFRASTRUCTURE ONLY (see lasso_oracle.hpp header).
//
// Data-parallel helpers for the CPU restatement.  The reference fans its loops out on the global rayon pool
// (`multicore` feature: subprotocols/sumcheck.rs:174-218, poly/dense_mlpoly.rs:118-127, lasso/memory_checking.rs:280-302,
// ark-ec's msm over windows); this is the OpenMP statement of the same thing, used so that the oracle can (a) prove the
// metric-size instance (2^24 lookups) inside the GPU-test budget and (b) serve as the all-core CPU baseline of bench.py.
//
// Exactness: every value is a canonical field element (4 x u64 Montgomery limbs, fully reduced after every operation) or
// a group element compared after canonical compression, and modular addition is associative and commutative, so ANY
// partition of a sum over threads gives the same bytes as the serial loop.  tests/test_oracle_parallel.py holds that as a
// test: proofs at OMP_NUM_THREADS=1 and =N are byte-identical.
//
// Built with -fopenmp; without it the pragmas vanish and everything below is the serial loop.
#pragma once
#include <cstddef>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

inline int par_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
inline void par_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

// Threads for a region of n iterations when one thread should have at least `per_thread` of them: small regions run on few threads (or inline) so that
// their fork/join cost stays proportional to the work — a late sumcheck round of 4096 terms on 128 threads would be nothing but barrier.
inline int par_threads_for(size_t n, size_t per_thread) {
#ifdef _OPENMP
  if (omp_in_parallel()) return 1;
  const size_t mx = (size_t)omp_get_max_threads(), want = per_thread ? n / per_thread : n;
  return (int)(want < 1 ? 1 : want > mx ? mx : want);
#else
  (void)n; (void)per_thread; return 1;
#endif
}

// for i in [0, n): f(i), iterations independent.  Bodies must not throw (an exception cannot leave an OpenMP region).
// per_thread: iterations that justify one more thread (light bodies, a few field operations: thousands; an MSM row or a scalar multiplication: 1).
template <class Fn>
inline void par_for(size_t n, Fn&& f, size_t per_thread = 4096) {
#ifdef _OPENMP
  const int nt = par_threads_for(n, per_thread);
  if (nt > 1) {
#pragma omp parallel for schedule(static) num_threads(nt)
    for (size_t i = 0; i < n; i++) f(i);
    return;
  }
#endif
  for (size_t i = 0; i < n; i++) f(i);
}

// K running sums over i in [0, n): f(i, acc) adds its terms into acc[0..K).  T needs zero() and operator+=.
// Per-thread accumulators joined in thread order; identical to the serial sum because the addition is exact.
template <class T, class Fn>
inline std::vector<T> par_sums(size_t n, size_t K, Fn&& f, size_t per_thread = 2048) {
  std::vector<T> total(K, T::zero());
#ifdef _OPENMP
  const int nt = par_threads_for(n, per_thread);
  if (nt > 1) {
    std::vector<std::vector<T>> part((size_t)nt, std::vector<T>(K, T::zero()));
#pragma omp parallel num_threads(nt)
    {
      std::vector<T> acc(K, T::zero());
#pragma omp for schedule(static) nowait
      for (size_t i = 0; i < n; i++) f(i, acc.data());
      part[(size_t)omp_get_thread_num()] = acc;
    }
    for (auto& p : part) for (size_t k = 0; k < K; k++) total[k] += p[k];
    return total;
  }
#endif
  for (size_t i = 0; i < n; i++) f(i, total.data());
  return total;
}

}  // namespace orc

// liblasso_hip.so — implementation of include/lasso_hip.h for MI355X (gfx950).
// One context = one device + one HIP stream + scratch.  See the header for the contract of each entry point.
#include <hip/hip_runtime.h>
#if defined(__SSE2__)
#include <emmintrin.h>   // 16-byte single-copy stores / loads of the hand-off chunks; other hosts take the per-word fallbacks (the check word covers tearing)
#endif
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include "../../include/lasso_hip.h"
#include "poly_kernels.cuh"
#include "msm_kernels.cuh"
#include "densify_kernels.cuh"

__global__ void __launch_bounds__(256) k_inner_lr(const fr_t* __restrict__ a, const fr_t* __restrict__ b, size_t half, fr_t* __restrict__ partials) {
  __shared__ RedScratch S;
  fr29 acc[3] = {fr29_zero(), fr29_zero(), fr29_zero()}; uint32_t cnt = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    acc[0] = fr29_weak(fr29_add(acc[0], fr29_mul(fr29_unpack_u(a[i]), fr29_unpack_s(b[half + i]))));
    acc[1] = fr29_weak(fr29_add(acc[1], fr29_mul(fr29_unpack_u(a[half + i]), fr29_unpack_s(b[i]))));
    if ((++cnt & 127u) == 0) { acc[0] = fr29_mul(acc[0], fr29_one_s()); acc[1] = fr29_mul(acc[1], fr29_one_s()); }
  }
  store_block_partials<3>(acc, 2, partials + 2 * (size_t)blockIdx.x, 0, S);
}

static_assert(sizeof(lasso_fr) == 32 && sizeof(fr_t) == 32, "Fr layout");
static_assert(sizeof(lasso_affine) == 64 && sizeof(lasso_point) == 128 && sizeof(ed_point) == 128 && sizeof(niels29) == (MSM_NIELS_ALIGN > 16 ? 128 : 112) && sizeof(pt29) == 144, "curve layouts");

struct EventPair { hipEvent_t a, b; int kid; double bytes, units, units2; bool large; bool counted; };
#define LASSO_PROF_COUNT_SLOTS 4096   // device counters of exactly executed additions, one group of 64 words per bracketed launch of the fully-profiled step (the waves of a launch spread their atomics over the group)
struct lasso_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  void* d_scratch = nullptr; size_t scratch_cap = 0;      // reduction partials / converted scalars
  void* d_pip = nullptr; size_t pip_cap = 0;              // k_msm_pip_*: a group of rows' sorted pairs, bucket offsets, size ranks and bucket sums (run_msm)
  // Small results (round polynomials, claims) return through HOST-MAPPED pinned memory: the producing kernel stores straight into
  // h_small (d_small is the device alias of the same pages), then a sequence number is stored to h_flag and the host spins on it.
  // No hipMemcpy, no stream synchronisation: 6.7 us per round trip instead of 13.6 us (tools/latency_bench.hip on MI355X).
  fr_t* d_small = nullptr;                                // device alias of h_small
  fr_t* h_small = nullptr;                                // pinned, mapped, coherent
  size_t small_cap = 0;
  uint32_t* h_flag = nullptr; uint32_t* d_flag = nullptr; // sequence flag (mapped), device alias
  // Since round 3 the round kernels of the sumchecks hand their results over as SELF-VALIDATING chunks instead (poly_kernels.cuh "results for the host": three tagged 16-byte
  // chunks per element in h_tag, one release fence on the device, no ticket and no flag: 2.1 us per resident turn instead of 4.1, tools/handoff_bench.hip).  LASSO_TAGGED_RESULTS=0: the flag protocol.
  uint32_t* h_tag = nullptr; uint32_t* d_tag = nullptr;    // small_cap elements of 48 bytes (mapped, zero-initialised: sequence numbers start at 1)
  bool tagged = true, pending_tagged = false;
  uint32_t pending_groups = 1, pending_K = 0;   // a launch whose workgroups each published their own block sums (LASSO_TAGGED_DIRECT): groups per row, values per row
  std::vector<lasso_fr> group_tmp;
  // The resident tails' mailbox: three tagged 16-byte chunks of host-mapped memory (h_flag + 32) the kernel polls (k_cubic_tail).  Putting it in device memory the host writes through the BAR
  // (hsa_amd_agents_allow_access) was tried in round 3: 2.5 -> 2.1 us per empty turn in tools/pingpong_bench.hip, nothing measurable in a proof, and hand-offs lost with four contexts proving at once — dropped.
  uint32_t* mail_h = nullptr; uint32_t* mail_d = nullptr;
  uint32_t seq = 0;
  uint64_t stat_waits = 0; double stat_wait_us = 0;        // host time spent spinning on the flag (lasso_wait_stats)
  fr_t* d_big = nullptr; fr_t* h_big = nullptr; size_t big_cap = 0;   // large results (matvec rows): device buffer + pinned mirror, hipMemcpyAsync
  uint32_t* d_flags = nullptr;
  uint32_t* d_counters = nullptr;                         // arrival tickets of the in-launch reductions (zero between launches)
  bool pending = false, defer_next = false; uint32_t pending_seq = 0; size_t pending_count = 0;   // a deferred result not yet collected by lasso_result_wait
  // a bullet round launched ahead of its challenge (k_bullet_msm with a mailbox): enqueued by lasso_bullet_round_ahead, released by lasso_bullet_post (u, u^-1 in the two
  // mailboxes mail_h, mail_h + 12), after which it is an ordinary pending result.  d_gmail: 18 words of device memory, workgroup (0, 0)'s republication of the two scalars
  bool ahead_active = false; uint32_t ahead_seq = 0; uint32_t* d_gmail = nullptr;
  bool tail_active = false; uint32_t tail_seq0 = 0, tail_turn = 0, tail_turns = 0; size_t tail_count = 0, tail_final = 0;   // resident sumcheck-tail kernel (k_cubic_tail); its mailbox = h_flag + 32 (bytes 128..163)
  bool tail_unstarted = false;    // the resident tail was launched AHEAD of its first challenge (lasso_sumcheck_cubic_tail_begin_ahead): the first lasso_sumcheck_cubic_tail_next starts it
  uint32_t handover_next = 0;     // lasso_tail_handover_next: the next cubic tail stops at this many elements per array and hands the arrays over
  // a sumcheck round launched ahead (lasso_sumcheck_cubic_eqw2_begin_ahead): what lasso_challenge_post turns into the pending result
  size_t ahead_count = 0; bool ahead_tagged = false; uint32_t ahead_groups = 1, ahead_K = 0; bool ahead_bullet = false;
  // a LAYER's first launch enqueued ahead of the layer's eq point (lasso_sumcheck_cubic_eqw2_begin_eq_ahead / lasso_sumcheck_cubic_tail_begin_eq_ahead): legal while the previous
  // layer's tail is still active; lasso_point_post delivers the point through pmail (host-mapped, one mailbox entry per field element) and turns the launch into the pending
  // result / the active tail; lasso_point_cancel ends it without a result.  d_gpoint: where the gate (k_gate_point) leaves the point for the kernels behind it.
  uint32_t* pmail_h = nullptr; uint32_t* pmail_d = nullptr; uint32_t* d_gpoint = nullptr;
  uint32_t gate_sent = 0;         // sequence number of the last point gate launched: the mailbox area is its until pmail_h[LASSO_PMAIL_ACK_WORD] shows it (gate_free)
  bool lay_active = false, lay_tail = false, lay_tagged = false, no_grow = false; uint32_t lay_seq = 0, lay_ell = 0, lay_groups = 1, lay_K = 0, lay_turns = 0; size_t lay_count = 0, lay_final = 0;
  uint32_t prof_mask = 0;   // bit k set = kernel family k is bracketed with events
  std::vector<EventPair> events; size_t events_used = 0;
  uint64_t prof_launches[LASSO_K_COUNT] = {0}; double prof_ms[LASSO_K_COUNT] = {0}; double prof_bytes[LASSO_K_COUNT] = {0};
  // the same, restricted to launches whose algorithmic bytes exceed LASSO_PROF_LARGE_BYTES (past the 256 MiB Infinity Cache: the HBM-bound regime)
  uint64_t big_launches[LASSO_K_COUNT] = {0}; double big_ms[LASSO_K_COUNT] = {0}; double big_bytes[LASSO_K_COUNT] = {0};
  // family-specific work units beside the bytes (the MSM families: group additions of the reference's algorithm), all / large launches
  double prof_units[LASSO_K_COUNT] = {0}; double big_units[LASSO_K_COUNT] = {0};
  double prof_units2[LASSO_K_COUNT] = {0}; double big_units2[LASSO_K_COUNT] = {0};   // MSM families: mixed additions the kernel itself issues at most (one per scalar digit it looks at)
  double prof_units3[LASSO_K_COUNT] = {0}; double big_units3[LASSO_K_COUNT] = {0};   // MSM families: mixed additions EXECUTED, counted by the kernels (only while every launch is bracketed: no LASSO_PROF_LARGE_ONLY)
  uint32_t* d_prof_counts = nullptr;
  void* rccl_comm = nullptr; int rccl_world = 0; int rccl_rank = -1;   // slab mode's device-side exchange (lasso_rccl_*): an ncclComm_t bound to this context's device and stream
  // device memory held through this context (lasso_mem_stats): lasso_alloc'd buffers, the context's scratch / result buffers and the generator tables built with it
  std::unordered_map<void*, size_t> mem_sizes; uint64_t mem_live = 0, mem_peak = 0; std::mutex mem_mu;   // bases tables may be built from another host thread (ensure_tab8)
};
struct lasso_bases {
  size_t n = 0; niels29* d_table = nullptr;
  lasso_ctx* owner = nullptr;                // the context its tables' bytes are accounted to (lasso_mem_stats)
  niels29* d_mult = nullptr;                 // signed digit multiples for the latency-shaped MSM (k_msm_direct), optional
  niels29* d_mult8 = nullptr;                // signed BYTE multiples m * 256^w * G, m = 1..128 (32 windows): half the additions of the same MSMs; generator sets up to LASSO_MSM_DIRECT8_MAX_N
  niels29* d_tab8[2] = {nullptr, nullptr};   // byte multiples m * 256^w * G_j, m = 1..255, for the row-parallel commitments of small scalars (k_msm_rows8); built on first use
  bool tab8_failed = false;
  std::mutex tab8_mu;                        // the lazy build is serialised: two contexts (or host threads) sharing one bases object may reach first use together
};

static thread_local std::string g_create_err;
// contexts alive in this process: a lasso_bases keeps a raw pointer to the context its tables are accounted to, and may outlive it
static std::mutex g_ctx_mu; static std::vector<lasso_ctx*> g_ctx_live;
static lasso_ctx* live_or_null(lasso_ctx* c) { std::lock_guard<std::mutex> g(g_ctx_mu); for (lasso_ctx* x : g_ctx_live) if (x == c) return c; return nullptr; }

static int32_t fail(lasso_ctx* c, int32_t code, const std::string& msg) { if (c) c->err = msg; else g_create_err = msg; return code; }
#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((c), e_ == hipErrorOutOfMemory ? LASSO_ERR_OOM : LASSO_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
// Every entry point runs on the context's device whatever the calling thread's current device is (two contexts on different devices driven from one
// thread, or worker threads that default to device 0): hipGetDevice/hipSetDevice are thread-local state in the runtime, a few tens of nanoseconds.
static inline bool bind_device(lasso_ctx* c) { if (!c) return true; int d = -1; if (hipGetDevice(&d) == hipSuccess && d == c->device) return true; return hipSetDevice(c->device) == hipSuccess; }
#define REQUIRE(c, cond) do { if (!bind_device(c)) return fail((c), LASSO_ERR_HIP, "hipSetDevice failed for the context's device"); \
                              if (!(cond)) return fail((c), LASSO_ERR_INVALID, std::string("invalid argument: ") + #cond); } while (0)

// hipMalloc / hipFree with the context's byte accounting (lasso_mem_stats)
static hipError_t dmalloc(lasso_ctx* c, void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess && c) { std::lock_guard<std::mutex> g(c->mem_mu); c->mem_sizes[*p] = bytes; c->mem_live += bytes; if (c->mem_live > c->mem_peak) c->mem_peak = c->mem_live; }
  return e;
}
static hipError_t dfree(lasso_ctx* c, void* p) {
  if (c && p) { std::lock_guard<std::mutex> g(c->mem_mu); auto it = c->mem_sizes.find(p); if (it != c->mem_sizes.end()) { c->mem_live -= it->second; c->mem_sizes.erase(it); } }
  return hipFree(p);
}

static int32_t ensure_scratch(lasso_ctx* c, size_t bytes) {
  if (bytes <= c->scratch_cap) return 0;
  if (c->no_grow) return LASSO_ERR_UNSUPPORTED;   // a launch enqueued behind a resident kernel: the caller takes the ordinary path instead
  if (c->ahead_active || c->lay_active) return fail(c, LASSO_ERR_INVALID, "a launch is waiting on the device for its challenge: only lasso_result_wait and the post of that challenge are legal until then");   // growing would synchronise the stream for the kernel's whole 5 s bail-out and lose the round
  if (c->d_scratch) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, dfree(c, c->d_scratch)); c->d_scratch = nullptr; c->scratch_cap = 0; }
  size_t cap = bytes < ((size_t)1 << 22) ? ((size_t)1 << 22) : bytes;
  HIPCHK(c, dmalloc(c, &c->d_scratch, cap)); c->scratch_cap = cap; return 0;
}
static int32_t ensure_pip(lasso_ctx* c, size_t bytes) {
  if (bytes <= c->pip_cap) return 0;
  if (c->no_grow || c->ahead_active || c->lay_active) return LASSO_ERR_UNSUPPORTED;   // the caller takes the bucket kernel instead
  if (c->d_pip) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, dfree(c, c->d_pip)); c->d_pip = nullptr; c->pip_cap = 0; }
  if (dmalloc(c, &c->d_pip, bytes) != hipSuccess) { (void)hipGetLastError(); c->d_pip = nullptr; return LASSO_ERR_UNSUPPORTED; }   // no room: not an error, the bucket kernel serves the commitment
  c->pip_cap = bytes; return 0;
}
static int32_t ensure_small(lasso_ctx* c, size_t count) {
  if (count <= c->small_cap) return 0;
  if (c->no_grow) return LASSO_ERR_UNSUPPORTED;   // a launch enqueued behind a resident kernel: the caller takes the ordinary path instead
  if (c->ahead_active || c->lay_active) return fail(c, LASSO_ERR_INVALID, "a launch is waiting on the device for its challenge: only lasso_result_wait and the post of that challenge are legal until then");   // growing would synchronise the stream for the kernel's whole 5 s bail-out and lose the round
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->h_small) (void)hipHostFree(c->h_small);
  c->h_small = nullptr; c->d_small = nullptr; c->small_cap = 0;
  size_t cap = count < 4096 ? 4096 : count;
  HIPCHK(c, hipHostMalloc((void**)&c->h_small, cap * sizeof(fr_t), hipHostMallocMapped | hipHostMallocCoherent));
  HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_small, c->h_small, 0));
  if (c->h_tag) (void)hipHostFree(c->h_tag);
  c->h_tag = nullptr; c->d_tag = nullptr;
  HIPCHK(c, hipHostMalloc((void**)&c->h_tag, cap * 48, hipHostMallocMapped | hipHostMallocCoherent));
  memset(c->h_tag, 0, cap * 48);
  HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_tag, c->h_tag, 0));
  c->small_cap = cap; return 0;
}
static int32_t ensure_big(lasso_ctx* c, size_t count) {
  if (count <= c->big_cap) return 0;
  if (c->no_grow) return LASSO_ERR_UNSUPPORTED;   // a launch enqueued behind a resident kernel: the caller takes the ordinary path instead
  if (c->ahead_active || c->lay_active) return fail(c, LASSO_ERR_INVALID, "a launch is waiting on the device for its challenge: only lasso_result_wait and the post of that challenge are legal until then");   // growing would synchronise the stream for the kernel's whole 5 s bail-out and lose the round
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->d_big) (void)dfree(c, c->d_big);
  if (c->h_big) (void)hipHostFree(c->h_big);
  c->d_big = nullptr; c->h_big = nullptr; c->big_cap = 0;
  size_t cap = count < 8192 ? 8192 : count;
  HIPCHK(c, dmalloc(c, (void**)&c->d_big, cap * sizeof(fr_t)));
  HIPCHK(c, hipHostMalloc((void**)&c->h_big, cap * sizeof(fr_t), hipHostMallocDefault));
  c->big_cap = cap; return 0;
}
// profiling: bracket a launch with an event pair on the context's stream
struct ProfScope {
  lasso_ctx* c; int idx = -1;
  // `large`: the launch belongs to the throughput regime whatever its byte count (the row-parallel commitment MSMs: 64 MiB of u32 scalars, milliseconds of VALU work)
  ProfScope(lasso_ctx* c_, int kid, double bytes, double units = 0, bool large = false, double units2 = 0) : c(c_) {
    if (!((c->prof_mask >> kid) & 1u)) return;
    large = large || bytes >= LASSO_PROF_LARGE_BYTES;
    if ((c->prof_mask & 0x40000000u) && !large) return;   // LASSO_PROF_LARGE_ONLY: leave the latency-bound launches unbracketed
    if (c->events_used == c->events.size()) { EventPair p; if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return; c->events.push_back(p); }
    idx = (int)c->events_used++;
    c->events[idx].kid = kid; c->events[idx].bytes = bytes; c->events[idx].units = units; c->events[idx].units2 = units2; c->events[idx].large = large; c->events[idx].counted = false;
    (void)hipEventRecord(c->events[idx].a, c->stream);
  }
  ~ProfScope() { if (idx >= 0) (void)hipEventRecord(c->events[idx].b, c->stream); }
  // device counter the MSM kernels add their executed mixed additions to; NULL (no counting, no cost) unless every launch is being bracketed
  uint32_t* counter() {
    if (idx < 0 || idx >= LASSO_PROF_COUNT_SLOTS || (c->prof_mask & 0x40000000u)) return nullptr;
    if (!c->d_prof_counts) { if (hipMalloc((void**)&c->d_prof_counts, LASSO_PROF_COUNT_SLOTS * 256) != hipSuccess || hipMemset(c->d_prof_counts, 0, LASSO_PROF_COUNT_SLOTS * 256) != hipSuccess) { (void)hipGetLastError(); c->d_prof_counts = nullptr; return nullptr; } }
    c->events[idx].counted = true;
    return c->d_prof_counts + (size_t)idx * 64;
  }
};
static void prof_flush(lasso_ctx* c) {
  if (!c->events_used) return;
  (void)hipStreamSynchronize(c->stream);
  std::vector<uint64_t> counts;   // 64-bit: a full-width commitment of 2^28 scalars executes 1.6e10 additions in ONE launch (the 32-bit sum wrapped: Spark C=16 2^24 read 0.12 for 0.67)
  if (c->d_prof_counts) {
    const size_t m = c->events_used < LASSO_PROF_COUNT_SLOTS ? c->events_used : LASSO_PROF_COUNT_SLOTS;
    std::vector<uint32_t> raw(m * 64); counts.assign(m, 0);
    if (hipMemcpy(raw.data(), c->d_prof_counts, m * 256, hipMemcpyDeviceToHost) != hipSuccess || hipMemset(c->d_prof_counts, 0, m * 256) != hipSuccess) { (void)hipGetLastError(); counts.clear(); }
    else for (size_t i = 0; i < m; i++) for (size_t k = 0; k < 64; k++) counts[i] += raw[i * 64 + k];
  }
  for (size_t i = 0; i < c->events_used; i++) {
    float ms = 0; if (hipEventElapsedTime(&ms, c->events[i].a, c->events[i].b) != hipSuccess) continue;
    int k = c->events[i].kid; c->prof_launches[k]++; c->prof_ms[k] += ms; c->prof_bytes[k] += c->events[i].bytes; c->prof_units[k] += c->events[i].units; c->prof_units2[k] += c->events[i].units2;
    const double exact = c->events[i].counted && i < counts.size() ? (double)counts[i] : 0.0;
    c->prof_units3[k] += exact;
    if (c->events[i].large) { c->big_launches[k]++; c->big_ms[k] += ms; c->big_bytes[k] += c->events[i].bytes; c->big_units[k] += c->events[i].units; c->big_units2[k] += c->events[i].units2; c->big_units3[k] += exact; }
  }
  c->events_used = 0;
}
static inline unsigned grid_for(size_t n, unsigned cap = 2048) { size_t g = (n + LASSO_BLOCK - 1) / LASSO_BLOCK; if (g < 1) g = 1; if (g > cap) g = cap; return (unsigned)g; }
static inline fr_t to_fr(const lasso_fr* p) { fr_t r; memcpy(r.v, p, 32); return r; }
// Wait until the device has stored sequence number `seq` to the mapped flag, then copy `count` results out of the mapped buffer.
// The producer's stores to h_small are ordered before the flag by a system-scope release on the device (k_publish / publish_flag).
static inline double now_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
// One element of the tagged result area (three 16-byte chunks, poly_kernels.cuh result_store): true once all three carry `seq` and the check word agrees.  Each chunk is read with
// one aligned 16-byte load (the device wrote it with one aligned 16-byte store); the check word also covers a platform that would tear either.
static inline bool tagged_element(const uint32_t* e, uint32_t seq, uint32_t* w8) {
  uint32_t c[12];
#if defined(__SSE2__)
  for (int k = 0; k < 3; k++) _mm_storeu_si128((__m128i*)(c + 4 * k), _mm_load_si128((const __m128i*)(e + 4 * k)));
#else
  for (int k = 0; k < 12; k++) c[k] = __atomic_load_n(e + k, __ATOMIC_ACQUIRE);
#endif
  if (c[0] != seq || c[4] != seq || c[8] != seq) return false;
  w8[0] = c[1]; w8[1] = c[2]; w8[2] = c[3]; w8[3] = c[5]; w8[4] = c[6]; w8[5] = c[7]; w8[6] = c[9]; w8[7] = c[10];
  return c[11] == (w8[0] ^ w8[1] ^ w8[2] ^ w8[3] ^ w8[4] ^ w8[5] ^ w8[6] ^ w8[7]) + seq * 0x9E3779B9u;
}
static int32_t wait_flag(lasso_ctx* c, uint32_t seq, size_t count, lasso_fr* out, bool tagged = false, uint32_t groups = 1, uint32_t K = 0) {
  if (c->defer_next) { c->defer_next = false; c->pending = true; c->pending_seq = seq; c->pending_count = count; c->pending_tagged = tagged; c->pending_groups = groups; c->pending_K = K; return 0; }
  if (groups > 1) {   // every workgroup of a row published its block sums, slot (row * groups + bx) * K + k: the row's K values are the sums over bx
    c->group_tmp.resize(count * groups);
    int32_t rc = wait_flag(c, seq, count * groups, c->group_tmp.data(), tagged); if (rc) return rc;
    const size_t rows = count / K;
    for (size_t y = 0; y < rows; y++) for (uint32_t k = 0; k < K; k++) {
      fr_t acc = to_fr(&c->group_tmp[(y * groups) * K + k]);
      for (uint32_t bx = 1; bx < groups; bx++) acc = fr_add(acc, to_fr(&c->group_tmp[(y * groups + bx) * K + k]));
      memcpy(out + y * K + k, acc.v, 32);
    }
    return 0;
  }   // lasso_defer_next: collected by lasso_result_wait
  uint64_t spins = 0;
  const double t0 = now_us();
  if (tagged) {
    for (size_t e = 0; e < count; e++) {
      bool last_look = false;
      while (!tagged_element(c->h_tag + 12 * e, seq, (uint32_t*)(out + e))) {
        if (last_look) return fail(c, LASSO_ERR_HIP, "a result was not delivered by the device");
        if ((++spins & 0xffff) == 0) {   // a faulted or finished stream can never deliver it: look once more, then stop
          hipError_t q = hipStreamQuery(c->stream);
          if (q == hipSuccess) last_look = true;
          else if (q != hipErrorNotReady) return fail(c, LASSO_ERR_HIP, std::string("stream error while waiting for a result: ") + hipGetErrorString(q));
        }
        __builtin_ia32_pause();
      }
    }
    c->stat_waits++; c->stat_wait_us += now_us() - t0;
    return 0;
  }
  while (__atomic_load_n(c->h_flag, __ATOMIC_ACQUIRE) != seq) {
    if ((++spins & 0xffff) == 0) {   // a faulted or finished stream can never raise the flag: stop spinning
      hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) {
        if (__atomic_load_n(c->h_flag, __ATOMIC_ACQUIRE) == seq) break;
        uint32_t gm[18] = {0}; if (c->d_gmail) (void)hipMemcpy(gm, c->d_gmail, sizeof(gm), hipMemcpyDeviceToHost);
        return fail(c, LASSO_ERR_HIP, "result flag was not raised by the device (flag " + std::to_string(*c->h_flag) + ", waiting for " + std::to_string(seq) + ", last sequence number " + std::to_string(c->seq) +
                                      ", launched-ahead tags " + std::to_string(gm[16]) + " / " + std::to_string(gm[17]) + ", mailbox tags " + std::to_string(c->mail_h[0]) + " / " + std::to_string(c->mail_h[12]) + ")");
      }
      if (q != hipErrorNotReady) return fail(c, LASSO_ERR_HIP, std::string("stream error while waiting for a result: ") + hipGetErrorString(q));
    }
    __builtin_ia32_pause();
  }
  c->stat_waits++; c->stat_wait_us += now_us() - t0;
  memcpy(out, c->h_small, count * sizeof(fr_t));
  return 0;
}
// Sequence numbers tag every hand-off (result chunks, the tails' mailbox) and must be unique among everything the tagged areas can still hold.  They are 32 bits: at ~1e5
// hand-offs per second a long-lived context would wrap after about 12 hours, reach LASSO_MAIL_POISON (0xFFFFFFFF) and then meet 0 (= "never written") and its own old tags
// (ADVICE r3).  next_seq hands out `span` consecutive numbers and, long before the top of the range, starts a new epoch instead: drain the stream (nothing in flight can
// still publish), zero the tagged result area and the mailbox, and restart at 1.  Costs one stream synchronisation per ~4e9 hand-offs.
static uint32_t next_seq(lasso_ctx* c, uint32_t span = 1) {
  // not while a result is still uncollected (lasso_defer_next) or a resident tail holds a block of numbers: the 2^20 numbers of slack cover any such stretch
  if (c->seq > 0xFFF00000u - span && ((!c->pending && !c->tail_active && !c->ahead_active && !c->lay_active) || c->seq > 0xFFFFFF00u - span)) {
    (void)hipStreamSynchronize(c->stream);
    if (c->pmail_h) { memset(c->pmail_h, 0, LASSO_PMAIL_BYTES); c->gate_sent = 0; }
    if (c->d_gpoint) (void)hipMemset(c->d_gpoint, 0, LASSO_GPOINT_BYTES);
    if (c->h_tag) memset(c->h_tag, 0, c->small_cap * 48);
    if (c->mail_h) memset(c->mail_h, 0, 96);   // both mailboxes
    if (c->d_gmail) (void)hipMemset(c->d_gmail, 0, 128);   // ... and the launched-ahead rounds' republication tags
    if (c->h_flag) *c->h_flag = 0;
    c->seq = 0;
  }
  const uint32_t first = c->seq + 1; c->seq += span; return first;
}
// where a converted round kernel's results go: (out, flag) arguments of the launch
#define RES(c) ((c)->tagged ? (fr_t*)(c)->d_tag : (c)->d_small), ((c)->tagged ? LASSO_TAGGED : (c)->d_flag)
__global__ void k_publish(uint32_t* flag, uint32_t seq) { __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
// results were stored to d_small (= mapped h_small) by kernels already enqueued on the stream: raise the flag behind them
static int32_t fetch_small(lasso_ctx* c, size_t count, lasso_fr* out) {
  const uint32_t seq = next_seq(c);
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, c->stream, c->d_flag, seq);
  HIPCHK(c, hipGetLastError());
  return wait_flag(c, seq, count, out);
}

// ------------------------------------------------------------------ RCCL over xGMI: the bulk exchange of slab mode (one proof over the P GPUs of a node)
// Each rank commits to ITS columns of every Hyrax row; the L partial row sums per rank are all-gathered on the context's stream (ncclAllGather of raw
// bytes: RCCL cannot add curve points) and every rank adds the P partials of each row itself (k_points_reduce_compress) — the north star's "RCCL reduce
// over xGMI for partial bucket sums".  librccl is resolved with dlopen the first time a communicator is asked for, so single-GPU use never loads it.
#include <dlfcn.h>
#include <chrono>
#include <thread>
#include <rccl/rccl.h>
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // optional: only the self-test's time-out path uses it
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi* rccl_api(std::string* why) {
  static RcclApi api; static bool tried = false; static std::string err;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (api.lib) break; }
    if (!api.lib) err = std::string("librccl not found: ") + dlerror();
    else {
      api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId"); api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
      api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather"); api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
      api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString"); api.CommAbort = (decltype(api.CommAbort))dlsym(api.lib, "ncclCommAbort");
      if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy || !api.GetErrorString) { err = "librccl lacks an expected symbol"; api.lib = nullptr; }
    }
  }
  if (!api.lib) { if (why) *why = err; return nullptr; }
  return &api;
}
static void rccl_release(lasso_ctx* c) { if (c->rccl_comm) { RcclApi* a = rccl_api(nullptr); if (a) (void)a->CommDestroy((ncclComm_t)c->rccl_comm); c->rccl_comm = nullptr; c->rccl_world = 0; } }
// one thread per row: sum of the P ranks' partial row commitments, then ark-serialize's compressed form (32 bytes per row)
__global__ void __launch_bounds__(256) k_points_reduce_compress(const pt29* __restrict__ parts, uint32_t groups, size_t rows, uint32_t* __restrict__ out32) {
  const fe29 d2 = fe_d2();
  for (size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x; row < rows; row += (size_t)gridDim.x * blockDim.x) {
    pt29 acc = parts[row];
    for (uint32_t g = 1; g < groups; g++) acc = pt_add(acc, parts[(size_t)g * rows + row], d2);
    pt_compress(acc, out32 + 8 * row);
  }
}

extern "C" {

// ---- slab mode's device-side exchange
int32_t lasso_rccl_unique_id(uint8_t out[128]) {
  std::string why; RcclApi* a = rccl_api(&why);
  if (!a) return fail(nullptr, LASSO_ERR_UNSUPPORTED, why);
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id; ncclResult_t r = a->GetUniqueId(&id);
  if (r != ncclSuccess) return fail(nullptr, LASSO_ERR_HIP, std::string("ncclGetUniqueId: ") + a->GetErrorString(r));
  memcpy(out, &id, 128); return 0;
}
int32_t lasso_rccl_init(lasso_ctx* c, int32_t rank, int32_t world, const uint8_t id[128]) {
  REQUIRE(c, id && world >= 1 && rank >= 0 && rank < world && !c->rccl_comm);
  std::string why; RcclApi* a = rccl_api(&why);
  if (!a) return fail(c, LASSO_ERR_UNSUPPORTED, why);
  ncclUniqueId uid; memcpy(&uid, id, 128);
  ncclComm_t comm = nullptr; ncclResult_t r = a->CommInitRank(&comm, world, uid, rank);
  if (r != ncclSuccess) return fail(c, LASSO_ERR_HIP, std::string("ncclCommInitRank: ") + a->GetErrorString(r));
  c->rccl_comm = comm; c->rccl_world = world; c->rccl_rank = rank; return 0;
}
int32_t lasso_rccl_ready(lasso_ctx* c) { return c && c->rccl_comm ? c->rccl_world : 0; }
// 1 if librccl can be loaded and has the entry points this library uses, 0 otherwise — WITHOUT touching a communicator.  ncclCommInitRank is a collective: a rank
// that cannot load librccl must say so BEFORE its peers enter it (they would wait for it forever), so the ranks exchange this value first.
int32_t lasso_rccl_available(void) { return rccl_api(nullptr) ? 1 : 0; }
int32_t lasso_rccl_shutdown(lasso_ctx* c) { REQUIRE(c, c); (void)hipStreamSynchronize(c->stream); rccl_release(c); return 0; }
// First contact with a communicator (ncclAllGather with world > 1 had never executed anywhere when this was written): all-gather 1 KB of a rank-dependent
// pattern on the context's stream and check every byte.  Collective — every rank of the communicator calls it.  The wait is BOUNDED (20 s of polling, not a stream
// synchronisation): on a time-out the communicator is aborted (ncclCommAbort, when the library has it) and dropped, so the caller falls back instead of hanging.
int32_t lasso_rccl_selftest(lasso_ctx* c) {
  REQUIRE(c, c && c->rccl_comm && !c->ahead_active && !c->lay_active && !c->tail_active);
  RcclApi* a = rccl_api(nullptr);
  const int world = c->rccl_world;
  const size_t bytes = 1024;
  uint8_t* d = nullptr;
  if (hipMalloc((void**)&d, bytes * (size_t)(world + 1)) != hipSuccess) { (void)hipGetLastError(); return fail(c, LASSO_ERR_OOM, "lasso_rccl_selftest: alloc"); }
  // the pattern is built from what the receiver can recompute: byte i of rank g = (g * 131 + i * 7 + 1) mod 251; this rank's g comes back in the gathered buffer itself
  // (slot g must hold g's pattern for EVERY g, which also proves the slots are in rank order)
  std::vector<uint8_t> mine(bytes), all(bytes * (size_t)world);
  int32_t rc = 0; std::string msg;
  const int my_rank = c->rccl_rank;
  for (size_t i = 0; i < bytes; i++) mine[i] = (uint8_t)(((size_t)my_rank * 131 + i * 7 + 1) % 251);
  hipEvent_t ev = nullptr;
  if (hipMemcpyAsync(d, mine.data(), bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { rc = LASSO_ERR_HIP; msg = "lasso_rccl_selftest: upload"; }
  if (!rc) {
    const ncclResult_t r = a->AllGather(d, d + bytes, bytes, ncclUint8, (ncclComm_t)c->rccl_comm, c->stream);
    if (r != ncclSuccess) { rc = LASSO_ERR_HIP; msg = std::string("lasso_rccl_selftest: ncclAllGather: ") + a->GetErrorString(r); }
  }
  if (!rc) {
    (void)hipEventRecord(ev, c->stream);
    const auto t0 = std::chrono::steady_clock::now(); bool done = false;
    while (std::chrono::steady_clock::now() - t0 < std::chrono::seconds(20)) { const hipError_t q = hipEventQuery(ev); if (q == hipSuccess) { done = true; break; } if (q != hipErrorNotReady) break; std::this_thread::sleep_for(std::chrono::microseconds(200)); }
    if (!done) {
      rc = LASSO_ERR_HIP; msg = "lasso_rccl_selftest: the 1 KB all-gather did not complete within 20 s (a peer never entered it, or the fabric is down)";
      if (a->CommAbort) (void)a->CommAbort((ncclComm_t)c->rccl_comm); c->rccl_comm = nullptr; c->rccl_world = 0;   // never destroy a communicator with a collective stuck in it
    }
  }
  if (!rc) {
    if (hipMemcpy(all.data(), d + bytes, bytes * (size_t)world, hipMemcpyDeviceToHost) != hipSuccess) { rc = LASSO_ERR_HIP; msg = "lasso_rccl_selftest: download"; }
    else for (int g = 0; g < world && !rc; g++) for (size_t i = 0; i < bytes; i++) if (all[(size_t)g * bytes + i] != (uint8_t)(((size_t)g * 131 + i * 7 + 1) % 251)) { rc = LASSO_ERR_HIP; msg = "lasso_rccl_selftest: slot " + std::to_string(g) + " does not hold rank " + std::to_string(g) + "'s bytes"; break; }
  }
  if (ev) (void)hipEventDestroy(ev);
  if (c->rccl_comm || rc == 0) (void)hipFree(d);   // (after an abort the buffer is left to the driver: a stuck collective may still reference it)
  (void)hipGetLastError();
  return rc ? fail(c, rc, msg) : 0;
}
// d_recv[g * bytes ..) <- rank g's d_send[0 .. bytes), enqueued on the context's stream (no host synchronisation)
int32_t lasso_rccl_allgather(lasso_ctx* c, const void* d_send, void* d_recv, size_t bytes) {
  REQUIRE(c, d_send && d_recv && c->rccl_comm);
  RcclApi* a = rccl_api(nullptr);
  ncclResult_t r = a->AllGather(d_send, d_recv, bytes, ncclUint8, (ncclComm_t)c->rccl_comm, c->stream);
  if (r != ncclSuccess) return fail(c, LASSO_ERR_HIP, std::string("ncclAllGather: ") + a->GetErrorString(r));
  return 0;
}
// d_parts: groups x rows points in the kernels' own form (lasso_point_row_bytes() each, rank-major, as lasso_rccl_allgather leaves them);
// out32: rows x 32 wire bytes (host) of the per-row sums
int32_t lasso_points_reduce_compress(lasso_ctx* c, const void* d_parts, uint32_t groups, size_t rows, uint8_t* out32) {
  REQUIRE(c, d_parts && out32 && groups >= 1 && rows >= 1);
  int32_t rc = ensure_scratch(c, rows * 32 + 256); if (rc) return rc;
  {
    ProfScope ps(c, LASSO_K_MSM, (double)groups * rows * sizeof(pt29), (double)(groups - 1) * rows);
    hipLaunchKernelGGL(k_points_reduce_compress, dim3(grid_for(rows)), dim3(256), 0, c->stream, (const pt29*)d_parts, groups, rows, (uint32_t*)c->d_scratch);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(out32, c->d_scratch, rows * 32, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// three self-validating chunks [tag, w, w, w] [tag, w, w, w] [tag, w, w, check], each ONE aligned 16-byte store; the check word (result_check's formula) lets the kernel reject a chunk
// that arrived in pieces
static inline void mail_chunks(uint32_t* mail, uint32_t tag, const uint32_t w[8]) {
  const uint32_t chk = (w[0] ^ w[1] ^ w[2] ^ w[3] ^ w[4] ^ w[5] ^ w[6] ^ w[7]) + tag * 0x9E3779B9u;
#if defined(__SSE2__)
  _mm_store_si128((__m128i*)(mail + 0), _mm_set_epi32((int)w[2], (int)w[1], (int)w[0], (int)tag));
  _mm_store_si128((__m128i*)(mail + 4), _mm_set_epi32((int)w[5], (int)w[4], (int)w[3], (int)tag));
  _mm_store_si128((__m128i*)(mail + 8), _mm_set_epi32((int)chk, (int)w[7], (int)w[6], (int)tag));
#else   // no 16-byte store: data words first, the tags last (a reader that sees all three tags with a matching check word has the whole message)
  const uint32_t m[12] = {tag, w[0], w[1], w[2], tag, w[3], w[4], w[5], tag, w[6], w[7], chk};
  for (int k : {1, 2, 3, 5, 6, 7, 9, 10, 11}) __atomic_store_n(mail + k, m[k], __ATOMIC_RELAXED);
  for (int k : {0, 4, 8}) __atomic_store_n(mail + k, m[k], __ATOMIC_RELEASE);
#endif
}
static inline void post_mail(lasso_ctx* c, uint32_t tag, const uint32_t w[8]) {
  mail_chunks(c->mail_h, tag, w);
#if defined(__SSE2__)
  _mm_sfence();   // release: the chunks are globally visible before anything the host does next
#else
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}
int32_t lasso_ctx_create(int32_t device, lasso_ctx** out) { return lasso_ctx_create_background(device, 0, out); }
int32_t lasso_ctx_device_uuid(lasso_ctx* c, uint8_t out[16]) {
  REQUIRE(c, c && out);
  hipUUID u; memset(&u, 0, sizeof(u));
  HIPCHK(c, hipDeviceGetUuid(&u, c->device));
  static_assert(sizeof(u.bytes) == 16, "hipUUID");
  memcpy(out, u.bytes, 16); return 0;
}
// background != 0: the context's stream gets the LOWEST priority the device offers (the prover's side context: bulk work that must not delay
// the latency-bound kernels of the main context); otherwise the highest.
int32_t lasso_ctx_create_background(int32_t device, int32_t background, lasso_ctx** out) {
  if (!out) return fail(nullptr, LASSO_ERR_INVALID, "out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return fail(nullptr, LASSO_ERR_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(nullptr, LASSO_ERR_INVALID, "device index out of range");
  lasso_ctx* c = new lasso_ctx(); c->device = device;
  int prio_lo = 0, prio_hi = 0;   // numerically lower = higher priority
  if ((e = hipSetDevice(device)) == hipSuccess) (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (e != hipSuccess || (e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, background ? prio_lo : prio_hi)) != hipSuccess) {
    std::string m = hipGetErrorString(e); delete c; return fail(nullptr, LASSO_ERR_HIP, m);
  }
  if (hipMalloc((void**)&c->d_flags, 64) != hipSuccess) { delete c; return fail(nullptr, LASSO_ERR_OOM, "flags alloc"); }
  if (hipMalloc((void**)&c->d_counters, (LASSO_MAX_PTRS + 40) * 4) != hipSuccess || hipMemset(c->d_counters, 0, (LASSO_MAX_PTRS + 40) * 4) != hipSuccess) { delete c; return fail(nullptr, LASSO_ERR_OOM, "counters alloc"); }
  if (hipHostMalloc((void**)&c->h_flag, 256, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || hipHostGetDevicePointer((void**)&c->d_flag, c->h_flag, 0) != hipSuccess) { delete c; return fail(nullptr, LASSO_ERR_OOM, "mapped flag alloc"); }
  *c->h_flag = 0;
  { const char* v = getenv("LASSO_TAGGED_RESULTS"); c->tagged = !(v && v[0] == '0'); }   // A/B switch: the flag protocol for every hand-off
  { const char* v = getenv("LASSO_SEQ_START"); if (v) c->seq = (uint32_t)strtoul(v, nullptr, 0); }   // tests: start close to the end of a sequence epoch (next_seq)
  c->mail_h = c->h_flag + 32; c->mail_d = c->d_flag + 32;   // 128-byte offset: 16-byte aligned chunks
  memset(c->h_flag, 0, 256);   // the flag word, the LT round's "not a bit" word and BOTH mailboxes (hipHostMalloc does not promise zeroed memory: a stale poison tag in the second mailbox ended the first launched-ahead round)
  int32_t rc = ensure_small(c, (size_t)1 << 16); if (rc) { g_create_err = c->err; delete c; return rc; }   // 2 MiB of mapped result buffer: the largest a-vector / row-commitment hand-off without a reallocation
  rc = ensure_scratch(c, (size_t)1 << 22); if (rc) { g_create_err = c->err; delete c; return rc; }
  if (dmalloc(c, (void**)&c->d_gmail, 128) != hipSuccess || hipMemset(c->d_gmail, 0, 128) != hipSuccess) { g_create_err = "gmail alloc"; delete c; return LASSO_ERR_OOM; }
  if (hipHostMalloc((void**)&c->pmail_h, LASSO_PMAIL_BYTES, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || hipHostGetDevicePointer((void**)&c->pmail_d, c->pmail_h, 0) != hipSuccess) { g_create_err = "point mailbox alloc"; delete c; return LASSO_ERR_OOM; }
  memset(c->pmail_h, 0, LASSO_PMAIL_BYTES);
  if (dmalloc(c, (void**)&c->d_gpoint, LASSO_GPOINT_BYTES) != hipSuccess || hipMemset(c->d_gpoint, 0, LASSO_GPOINT_BYTES) != hipSuccess) { g_create_err = "gpoint alloc"; delete c; return LASSO_ERR_OOM; }
  { std::lock_guard<std::mutex> g(g_ctx_mu); g_ctx_live.push_back(c); }
  *out = c; return 0;
}
// Error recovery: a host that stops between *_tail_begin and the last tail_next (an exception in the prover) leaves a resident kernel waiting for a
// challenge, tail_active / pending set, and possibly non-zero arrival tickets.  Post the poison tag (the kernels leave at their next poll instead of
// after the 5 s bail-out), drain the stream, clear the mailbox and the protocol state, and restore the "tickets are zero between launches" invariant.
int32_t lasso_abort(lasso_ctx* c) {
  REQUIRE(c, c);
  const uint32_t zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c->lay_active && c->pmail_h) for (uint32_t j = 0; j < LASSO_POINT_MAX + 2; j++) mail_chunks(c->pmail_h + 12 * j, LASSO_MAIL_POISON, zero8);
  if (c->tail_active || c->ahead_active || c->lay_active) { mail_chunks(c->mail_h + 12, LASSO_MAIL_POISON, zero8); post_mail(c, LASSO_MAIL_POISON, zero8); }
  (void)hipStreamSynchronize(c->stream);   // bounded: every device-side wait has the poison check and a wall-clock bail-out
  (void)hipGetLastError();
  if (c->pmail_h) { memset(c->pmail_h, 0, LASSO_PMAIL_BYTES); c->gate_sent = 0; }   // the stream is drained: no gate is left to read or acknowledge anything
  mail_chunks(c->mail_h + 12, 0, zero8); post_mail(c, 0, zero8);
  c->lay_active = false; c->lay_tail = false; c->no_grow = false;
  c->ahead_active = false; c->ahead_bullet = false; c->tail_unstarted = false; c->handover_next = 0;
  c->tail_active = false; c->pending = false; c->defer_next = false; c->events_used = 0; c->pending_groups = 1; c->pending_K = 0;
  HIPCHK(c, hipMemsetAsync(c->d_counters, 0, (LASSO_MAX_PTRS + 40) * 4, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
void lasso_ctx_destroy(lasso_ctx* c) {
  if (!c) return;
  { std::lock_guard<std::mutex> g(g_ctx_mu); for (size_t i = 0; i < g_ctx_live.size(); i++) if (g_ctx_live[i] == c) { g_ctx_live.erase(g_ctx_live.begin() + i); break; } }
  (void)hipSetDevice(c->device);
  if (c->tail_active || c->pending || c->ahead_active || c->lay_active) (void)lasso_abort(c);   // never block in the synchronise below for a kernel's 5 s bail-out
  (void)hipStreamSynchronize(c->stream);
  rccl_release(c);
  for (auto& p : c->events) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->d_pip) (void)hipFree(c->d_pip);
  if (c->d_gmail) (void)hipFree(c->d_gmail);
  if (c->d_gpoint) (void)hipFree(c->d_gpoint);
  if (c->pmail_h) (void)hipHostFree(c->pmail_h);
  if (c->h_small) (void)hipHostFree(c->h_small);
  if (c->h_tag) (void)hipHostFree(c->h_tag);
  if (c->h_flag) (void)hipHostFree(c->h_flag);
  if (c->d_big) (void)hipFree(c->d_big);
  if (c->h_big) (void)hipHostFree(c->h_big);
  if (c->d_flags) (void)hipFree(c->d_flags);
  if (c->d_counters) (void)hipFree(c->d_counters);
  if (c->d_prof_counts) (void)hipFree(c->d_prof_counts);
  (void)hipStreamDestroy(c->stream);
  delete c;
}
const char* lasso_last_error(lasso_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }
void* lasso_stream(lasso_ctx* c) { return c ? (void*)c->stream : nullptr; }
int32_t lasso_alloc(lasso_ctx* c, size_t bytes, void** d_out) { REQUIRE(c, d_out); HIPCHK(c, dmalloc(c, d_out, bytes ? bytes : 1)); return 0; }
int32_t lasso_free(lasso_ctx* c, void* p) { if (!p) return 0; REQUIRE(c, c && !c->ahead_active && !c->lay_active); HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, dfree(c, p)); return 0; }
int32_t lasso_mem_stats(lasso_ctx* c, uint64_t* live_bytes, uint64_t* peak_bytes, int32_t reset_peak) {
  REQUIRE(c, c); std::lock_guard<std::mutex> g(c->mem_mu);
  if (live_bytes) *live_bytes = c->mem_live; if (peak_bytes) *peak_bytes = c->mem_peak; if (reset_peak) c->mem_peak = c->mem_live; return 0;
}
int32_t lasso_trim(lasso_ctx* c) {
  REQUIRE(c, c && !c->pending && !c->tail_active && !c->ahead_active && !c->lay_active);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->d_pip) { HIPCHK(c, dfree(c, c->d_pip)); c->d_pip = nullptr; c->pip_cap = 0; }
  if (c->scratch_cap <= ((size_t)1 << 22)) return 0;
  HIPCHK(c, dfree(c, c->d_scratch)); c->d_scratch = nullptr; c->scratch_cap = 0;
  return ensure_scratch(c, (size_t)1 << 22);
}
int32_t lasso_upload(lasso_ctx* c, void* d, const void* s, size_t n) { REQUIRE(c, d && s && !c->ahead_active && !c->lay_active); HIPCHK(c, hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); return 0; }
int32_t lasso_download(lasso_ctx* c, void* d, const void* s, size_t n) { REQUIRE(c, d && s && !c->ahead_active && !c->lay_active); HIPCHK(c, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); return 0; }
int32_t lasso_copy(lasso_ctx* c, void* d, const void* s, size_t n) { REQUIRE(c, d && s); ProfScope ps(c, LASSO_K_MISC, 2.0 * n); HIPCHK(c, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, c->stream)); return 0; }
int32_t lasso_zero(lasso_ctx* c, void* d, size_t n) { REQUIRE(c, d); HIPCHK(c, hipMemsetAsync(d, 0, n, c->stream)); return 0; }
int32_t lasso_sync(lasso_ctx* c) { REQUIRE(c, c && !c->ahead_active && !c->lay_active); HIPCHK(c, hipStreamSynchronize(c->stream)); return 0; }

int32_t lasso_prof_get_large(lasso_ctx* c, int32_t k, uint64_t* launches, double* ms, double* bytes) {
  REQUIRE(c, k >= 0 && k < LASSO_K_COUNT); prof_flush(c);
  if (launches) *launches = c->big_launches[k]; if (ms) *ms = c->big_ms[k]; if (bytes) *bytes = c->big_bytes[k]; return 0;
}
int32_t lasso_wait_stats(lasso_ctx* c, uint64_t* waits, double* wait_us, int32_t reset) { REQUIRE(c, waits && wait_us); *waits = c->stat_waits; *wait_us = c->stat_wait_us; if (reset) { c->stat_waits = 0; c->stat_wait_us = 0; } return 0; }
int32_t lasso_prof_enable(lasso_ctx* c, int32_t mask) { prof_flush(c); c->prof_mask = (uint32_t)mask; return 0; }
int32_t lasso_prof_reset(lasso_ctx* c) { prof_flush(c); for (int i = 0; i < LASSO_K_COUNT; i++) { c->prof_launches[i] = 0; c->prof_ms[i] = 0; c->prof_bytes[i] = 0; c->big_launches[i] = 0; c->big_ms[i] = 0; c->big_bytes[i] = 0; c->prof_units[i] = 0; c->big_units[i] = 0; c->prof_units2[i] = 0; c->big_units2[i] = 0; c->prof_units3[i] = 0; c->big_units3[i] = 0; } return 0; }
int32_t lasso_prof_get_units(lasso_ctx* c, int32_t k, int32_t large_only, double* units) {
  REQUIRE(c, k >= 0 && k < LASSO_K_COUNT && units); prof_flush(c);
  const bool large = large_only & 1, second = large_only & 2, third = large_only & 4;   // bit 1: the kernel's own mixed additions, upper bound; bit 2: the same counted exactly by the kernels
  *units = third ? (large ? c->big_units3[k] : c->prof_units3[k]) : second ? (large ? c->big_units2[k] : c->prof_units2[k]) : (large ? c->big_units[k] : c->prof_units[k]); return 0;
}
int32_t lasso_prof_get(lasso_ctx* c, int32_t k, uint64_t* launches, double* ms, double* bytes) {
  REQUIRE(c, k >= 0 && k < LASSO_K_COUNT); prof_flush(c);
  if (launches) *launches = c->prof_launches[k]; if (ms) *ms = c->prof_ms[k]; if (bytes) *bytes = c->prof_bytes[k]; return 0;
}

// ------------------------------------------------------------------ polynomial entry points
int32_t lasso_fr_from_u32(lasso_ctx* c, const uint32_t* d_src, size_t n, lasso_fr* d_dst) {
  REQUIRE(c, d_src && d_dst); if (!n) return 0;
  ProfScope ps(c, LASSO_K_MISC, 36.0 * n);
  hipLaunchKernelGGL(k_from_u32, dim3(grid_for(n)), dim3(LASSO_BLOCK), 0, c->stream, d_src, n, (fr_t*)d_dst);
  HIPCHK(c, hipGetLastError()); return 0;
}
// SubtableStrategy::materialize_subtables (and.rs:16-28, or.rs, xor.rs, lt.rs:17-44, range_check.rs:19-51) as integers, on the device: entry i of
// subtable `sub` from the two halves (l, r) of the index (utils/mod.rs:82-89 split_bits).  64 K entries: built where they are used instead of on
// the host and uploaded.
__global__ void __launch_bounds__(256) k_subtable_u32(int32_t kind, uint32_t sub, uint32_t log_m, uint32_t log_r, uint32_t* __restrict__ out) {
  const size_t m = (size_t)1 << log_m; const uint32_t bits = log_m / 2, mask = (1u << bits) - 1u;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t r = (uint32_t)i & mask, l = (uint32_t)(i >> bits) & mask;
    uint32_t v;
    if (kind == LASSO_AND) v = l & r;
    else if (kind == LASSO_OR) v = l | r;
    else if (kind == LASSO_XOR) v = l ^ r;
    else if (kind == LASSO_LT) v = sub == 0 ? (l < r ? 1u : 0u) : (l == r ? 1u : 0u);
    else { const size_t cutoff = (size_t)1 << (log_r % log_m); v = sub == 0 ? (uint32_t)i : (sub == 1 ? (i < cutoff ? (uint32_t)i : 0u) : 0u); }
    out[i] = v;
  }
}
int32_t lasso_materialize_subtable_u32(lasso_ctx* c, const lasso_strategy* s, uint32_t sub, uint32_t* d_out) {
  REQUIRE(c, s && d_out && s->kind >= LASSO_AND && s->kind <= LASSO_RANGE && s->log_m >= 1 && s->log_m <= 31);
  const uint32_t nsub = s->kind == LASSO_LT ? 2u : (s->kind == LASSO_RANGE ? 3u : 1u);
  REQUIRE(c, sub < nsub);
  hipLaunchKernelGGL(k_subtable_u32, dim3(grid_for((size_t)1 << s->log_m)), dim3(256), 0, c->stream, s->kind, sub, s->log_m, s->log_r, d_out);
  HIPCHK(c, hipGetLastError()); return 0;
}
int32_t lasso_gather(lasso_ctx* c, const lasso_fr* d_table, const uint32_t* d_idx, size_t n, lasso_fr* d_out) {
  REQUIRE(c, d_table && d_idx && d_out); if (!n) return 0;
  ProfScope ps(c, LASSO_K_MISC, 68.0 * n);
  hipLaunchKernelGGL(k_gather, dim3(grid_for(n)), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)d_table, d_idx, n, (fr_t*)d_out);
  HIPCHK(c, hipGetLastError()); return 0;
}
int32_t lasso_eq_evals(lasso_ctx* c, const lasso_fr* r, uint32_t ell, lasso_fr* d_out) { return lasso_eq_evals_scaled(c, r, ell, nullptr, d_out); }
int32_t lasso_eq_evals_scaled(lasso_ctx* c, const lasso_fr* r, uint32_t ell, const lasso_fr* scale, lasso_fr* d_out) {
  REQUIRE(c, d_out && ell <= 40 && (r || ell == 0));
  const fr_t sc = scale ? to_fr(scale) : fr_one();
  const size_t n = (size_t)1 << ell;
  ProfScope ps(c, LASSO_K_EQ, 32.0 * n);
  if (ell <= 12) {
    RTable R; for (uint32_t j = 0; j < ell; j++) R.r[j] = to_fr(r + j);
    hipLaunchKernelGGL(k_eq_small, dim3(grid_for(n)), dim3(LASSO_BLOCK), 0, c->stream, R, ell, sc, (fr_t*)d_out);
  } else {
    // out[x] = hi[x >> lo_bits] * lo[x & mask]: two small tables (the factored evals of eq_poly.rs:44-52) then one outer-product pass
    const uint32_t lo_bits = ell / 2, hi_bits = ell - lo_bits;
    int32_t rc = ensure_scratch(c, (((size_t)1 << hi_bits) + ((size_t)1 << lo_bits)) * sizeof(fr_t)); if (rc) return rc;
    fr_t* hi = (fr_t*)c->d_scratch; fr_t* lo = hi + ((size_t)1 << hi_bits);
    if (hi_bits <= 16) {   // both factor tables in one launch
      RTable16 Rh, Rl; for (uint32_t j = 0; j < 16; j++) { Rh.r[j] = j < hi_bits ? to_fr(r + j) : fr_zero(); Rl.r[j] = j < lo_bits ? to_fr(r + hi_bits + j) : fr_zero(); }
      const unsigned hb = grid_for((size_t)1 << hi_bits), lb = grid_for((size_t)1 << lo_bits);
      hipLaunchKernelGGL(k_eq_small2, dim3(hb + lb), dim3(LASSO_BLOCK), 0, c->stream, Rh, hi_bits, sc, hi, hb, Rl, lo_bits, lo);
    } else {
      RTable Rh, Rl; for (uint32_t j = 0; j < hi_bits; j++) Rh.r[j] = to_fr(r + j); for (uint32_t j = 0; j < lo_bits; j++) Rl.r[j] = to_fr(r + hi_bits + j);
      hipLaunchKernelGGL(k_eq_small, dim3(grid_for((size_t)1 << hi_bits)), dim3(LASSO_BLOCK), 0, c->stream, Rh, hi_bits, sc, hi);
      hipLaunchKernelGGL(k_eq_small, dim3(grid_for((size_t)1 << lo_bits)), dim3(LASSO_BLOCK), 0, c->stream, Rl, lo_bits, fr_one(), lo);
    }
    hipLaunchKernelGGL(k_eq_outer, dim3(grid_for(n, 4096)), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)hi, (const fr_t*)lo, lo_bits, n, (fr_t*)d_out);
  }
  HIPCHK(c, hipGetLastError()); return 0;
}
int32_t lasso_bind_top(lasso_ctx* c, lasso_fr* const* d_polys, uint32_t npolys, size_t n, const lasso_fr* r) {
  REQUIRE(c, d_polys && r && npolys >= 1 && npolys <= LASSO_MAX_PTRS && n >= 2 && (n & (n - 1)) == 0);
  MutPtrTable T; for (uint32_t i = 0; i < npolys; i++) { REQUIRE(c, d_polys[i]); T.p[i] = (fr_t*)d_polys[i]; }
  const size_t half = n / 2;
  ProfScope ps(c, LASSO_K_BIND, 48.0 * n * npolys);   // read 32n + write 16n per polynomial
  hipLaunchKernelGGL(k_bind_top, dim3(grid_for(half, 4096), npolys), dim3(LASSO_BLOCK), 0, c->stream, T, half, to_fr(r));
  HIPCHK(c, hipGetLastError()); return 0;
}
// x-extent of the cubic-round grids: ~512 workgroups over the whole grid.  Inside a proof (random data, tools/gpu_sweep_roofline.sh) 512 total beats 1024 by 3-10% and 256 by 10%
// for both the cubic (ny = 2) and the linear (ny = 1) rounds, although in isolation on constant data 1024 is 6% faster (profiles/r01_microbench_v2.txt); the kernels are VALU-bound
// and every extra workgroup adds a reduction epilogue), LASSO_CUBIC_NX overrides for experiments
static unsigned cubic_nx_cap(unsigned ny) { static const long e = [] { const char* v = getenv("LASSO_CUBIC_NX"); return v ? atol(v) : 0L; }(); if (e > 0) return (unsigned)e; unsigned c = 512 / (ny ? ny : 1); return c < 64 ? 64 : c; }
static bool cubic_wide() { static const bool on = [] { const char* v = getenv("LASSO_CUBIC_WIDE"); return !(v && v[0] == '0'); }(); return on; }   // double-width accumulators in the two-sum fused round (A/B switch)
#define CUBIC_SMALL_Q 64   // rounds with at most this many indices per circuit take the latency-shaped kernel
// the reference's loop with an explicit third polynomial (any C): kept as the literal counterpart of sumcheck.rs:49-93
int32_t lasso_sumcheck_cubic_round(lasso_ctx* c, const lasso_fr* const* d_A, const lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_C, size_t n, lasso_fr* out) {
  REQUIRE(c, d_A && d_B && d_C && out && ncirc >= 1 && ncirc <= LASSO_MAX_PTRS && n >= 2 && (n & (n - 1)) == 0);
  const size_t half = n / 2;
  int32_t rc = ensure_small(c, (size_t)ncirc * 3); if (rc) return rc;
  const uint32_t seq = next_seq(c);
  PtrTable A, B; for (uint32_t i = 0; i < ncirc; i++) { REQUIRE(c, d_A[i] && d_B[i]); A.p[i] = (const fr_t*)d_A[i]; B.p[i] = (const fr_t*)d_B[i]; }
  const unsigned ny = ncirc, nx = grid_for(half, cubic_nx_cap(ny));
  rc = ensure_scratch(c, (size_t)nx * ny * 3 * sizeof(fr_t)); if (rc) return rc;
  {
    ProfScope ps(c, LASSO_K_CUBIC, 32.0 * n * (2.0 * ncirc + 1.0));
    hipLaunchKernelGGL(k_cubic_round_lb, dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, A, B, nx, ny, (const fr_t*)d_C, half, (fr_t*)c->d_scratch, c->d_counters, RES(c), seq);
  }
  HIPCHK(c, hipGetLastError());
  return wait_flag(c, seq, (size_t)ncirc * 3, out, c->tagged);
}
// eq-weighted forms (see k_cubic_eqw_* in poly_kernels.cuh): what the prover calls.  Algorithmic bytes are SURVEY.md §8d's for the reference's round
// (2k+1 polynomials), although the kernels read only the 2k of A and B plus n/4 .. n/2 table entries.
// One eq-weighted cubic round, enqueued only: r == nullptr is the first round of a layer (arrays of length n, evaluation only), otherwise the
// previous challenge is bound first (length n -> n/2) and the sums are those of the next round.  NT sums per circuit land in the mapped
// result buffer under sequence number *seq_out.
// eqi (first round of a layer, two-sum form, streaming size only): the layer's eq table is built inside the launch and WRITTEN to d_E (k_cubic_eqw_lb<2, true>)
// a launch of a few workgroups per circuit hands over EVERY workgroup's block sums (LASSO_TAGGED_DIRECT; the host adds them in wait_flag) instead of running the in-launch second stage
static unsigned direct_nx_max() { static const unsigned v = [] { const char* e = getenv("LASSO_DIRECT_NX"); const long x = e ? atol(e) : 16; return (unsigned)(x < 0 ? 0 : x > 64 ? 64 : x); }(); return v; }
#define CUBIC_RESULT_ARGS(nx_) \
  const bool direct = groups_out && c->tagged && (nx_) > 1 && (nx_) <= direct_nx_max(); \
  if (direct) { *groups_out = (nx_); rc = ensure_small(c, (size_t)ncirc * 3 * (nx_)); if (rc) return rc; } \
  fr_t* const r_out = c->tagged ? (fr_t*)c->d_tag : c->d_small; uint32_t* const r_flag = direct ? LASSO_TAGGED_DIRECT : c->tagged ? LASSO_TAGGED : c->d_flag
extern "C++" {
struct EqPointArg { const lasso_fr* point; uint32_t ell; fr_t scale; };   // a table too large for the in-LDS build: factor tables by k_eq_small2 into the scratch, the product inside round 0 (EqGlobal)
template <class TM, class TP>   // pointer tables sized for the number of circuits (MutPtrTable8 / PtrTable8 up to 8: 64 bytes of kernel arguments each instead of 1088)
static int32_t cubic_eqw_launch_t(lasso_ctx* c, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n, const lasso_fr* r, int NT, uint32_t* seq_out, const EqInline* eqi, uint32_t* groups_out, bool ahead = false, const EqPointArg* eqg = nullptr, int gate_ell = -1) {
  if (groups_out) *groups_out = 1;
  if (ahead && (NT != 2 || n / 4 <= CUBIC_SMALL_Q)) return fail(c, LASSO_ERR_UNSUPPORTED, "a round launched ahead of its challenge: two-sum streaming rounds only (more than 64 index quadruples per circuit)");
  TM A, B; for (uint32_t i = 0; i < ncirc; i++) { REQUIRE(c, d_A[i] && d_B[i]); A.p[i] = (fr_t*)d_A[i]; B.p[i] = (fr_t*)d_B[i]; }
  int32_t rc = ensure_small(c, (size_t)ncirc * 3); if (rc) return rc;
  const uint32_t seq = next_seq(c); *seq_out = seq;
  if (!r && !ahead) {
    const size_t half = n / 2;
    if (half <= CUBIC_SMALL_Q) {   // arrays are read-only in this mode
      ProfScope ps(c, LASSO_K_CUBIC, 32.0 * n * (2.0 * ncirc + 1.0));
      if (NT == 3) hipLaunchKernelGGL((k_cubic_eqw_small<false, 3, TM>), dim3(ncirc), dim3(LASSO_BLOCK), 0, c->stream, A, B, (const fr_t*)d_E, (uint32_t)half, fr_zero(), c->d_counters, RES(c), seq);
      else hipLaunchKernelGGL((k_cubic_eqw_small<false, 2, TM>), dim3(ncirc), dim3(LASSO_BLOCK), 0, c->stream, A, B, (const fr_t*)d_E, (uint32_t)half, fr_zero(), c->d_counters, RES(c), seq);
    } else {
      TP Ac, Bc; for (uint32_t i = 0; i < ncirc; i++) { Ac.p[i] = A.p[i]; Bc.p[i] = B.p[i]; }
      const unsigned ny = ncirc, nx = grid_for(half, cubic_nx_cap(ny));
      static const bool big_inline = [] { const char* v = getenv("LASSO_EQ_INLINE_BIG"); return v && v[0] == '1'; }();   // A/B switch: tables above 2^14 entries formed inside round 0 (EqGlobal) instead of by k_eq_outer in front of it
      const bool gated = gate_ell >= 0, gbig = gated && gate_ell > 14;   // the point comes through the gate (the layer is enqueued ahead of it); above 2^14 entries with factor tables in memory
      const uint32_t g_ell = gbig ? (uint32_t)gate_ell : eqg ? eqg->ell : 0, g_lo = g_ell / 2, g_hi = g_ell - g_lo;
      const size_t part_elems = (size_t)nx * ny * 3;
      rc = ensure_scratch(c, (part_elems + ((eqg || gbig) ? ((size_t)1 << g_hi) + ((size_t)1 << g_lo) : 0)) * sizeof(fr_t)); if (rc) return rc;
      CUBIC_RESULT_ARGS(nx);
      static const uint32_t pipe = [] { const char* v = getenv("LASSO_LB_PIPELINE"); return (v && v[0] == '0') ? 0u : 1u; }();
      // what runs in front of the round is outside the round's profiling bracket: the wait for the point (one wave), the two factor tables (their own bracket)
      fr_t* const f_hi = (fr_t*)c->d_scratch + part_elems; fr_t* const f_lo = f_hi + ((size_t)1 << g_hi);
      if (gated && NT == 2) hipLaunchKernelGGL(k_gate_point, dim3(1), dim3(64), 0, c->stream, (const uint32_t*)c->pmail_d, c->d_gpoint, seq, (uint32_t)gate_ell + 2u, c->pmail_d + LASSO_PMAIL_ACK_WORD); if (gated && NT == 2) c->gate_sent = seq;
      const uint32_t* const gate_gp = gated ? (const uint32_t*)c->d_gpoint : (const uint32_t*)nullptr;
      if ((gbig || eqg) && NT == 2) {
        ProfScope pe(c, LASSO_K_EQ, big_inline ? 32.0 * (((size_t)1 << g_hi) + ((size_t)1 << g_lo)) : 32.0 * half);
        const unsigned hb = grid_for((size_t)1 << g_hi), lb2 = grid_for((size_t)1 << g_lo);
        if (gbig) hipLaunchKernelGGL(k_eq_small2_mem, dim3(hb + lb2), dim3(LASSO_BLOCK), 0, c->stream, (const uint32_t*)c->d_gpoint, seq, g_hi, f_hi, hb, g_lo, f_lo);
        else {
          RTable16 Rh, Rl; for (uint32_t j = 0; j < 16; j++) { Rh.r[j] = j < g_hi ? to_fr(eqg->point + j) : fr_zero(); Rl.r[j] = j < g_lo ? to_fr(eqg->point + g_hi + j) : fr_zero(); }
          hipLaunchKernelGGL(k_eq_small2, dim3(hb + lb2), dim3(LASSO_BLOCK), 0, c->stream, Rh, g_hi, eqg->scale, f_hi, hb, Rl, g_lo, f_lo);
        }
        if (!big_inline) hipLaunchKernelGGL(k_eq_outer, dim3(grid_for(half, 4096)), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)f_hi, (const fr_t*)f_lo, g_lo, half, (fr_t*)d_E, gate_gp, seq);   // the table itself, as lasso_eq_evals_scaled writes it
      }
      ProfScope ps(c, LASSO_K_CUBIC, 32.0 * n * (2.0 * ncirc + 1.0));
      if ((gbig || eqg) && NT == 2 && !big_inline) {   // round 0 reads the table the kernels above left in d_E
        EqNone EN; EN.ell = 0; EN.gp = gate_gp; EN.seq = seq;
        hipLaunchKernelGGL((k_cubic_eqw_lb<2, false, TP, EqNone>), dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, Ac, Bc, nx, ny, (const fr_t*)d_E, half, (fr_t*)c->d_scratch, c->d_counters, r_out, r_flag, seq, pipe, EN, (fr_t*)nullptr);
      } else
      if (gated && NT == 2) {
        if (gbig) {
          fr_t* hi = f_hi; fr_t* lo = f_lo;
          EqGlobal G; G.hi = hi; G.lo = lo; G.lo_bits = g_lo; G.ell = g_ell; G.gp = c->d_gpoint; G.seq = seq;
          hipLaunchKernelGGL((k_cubic_eqw_lb<2, true, TP, EqGlobal>), dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, Ac, Bc, nx, ny, (const fr_t*)nullptr, half, (fr_t*)c->d_scratch, c->d_counters, r_out, r_flag, seq, 1u, G, (fr_t*)d_E);
        } else {
          EqInlineMem M; M.gp = c->d_gpoint; M.seq = seq; M.ell = (uint32_t)gate_ell;
          hipLaunchKernelGGL((k_cubic_eqw_lb<2, true, TP, EqInlineMem>), dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, Ac, Bc, nx, ny, (const fr_t*)nullptr, half, (fr_t*)c->d_scratch, c->d_counters, r_out, r_flag, seq, 1u, M, (fr_t*)d_E);
        }
      } else
      if (eqg && NT == 2) {   // factor tables behind the partials in the scratch (above), then round 0 with the product inside (the table goes to d_E on the way)
        fr_t* hi = f_hi; fr_t* lo = f_lo;
        EqGlobal G; G.hi = hi; G.lo = lo; G.lo_bits = g_lo; G.ell = eqg->ell; G.gp = nullptr; G.seq = 0;
        hipLaunchKernelGGL((k_cubic_eqw_lb<2, true, TP, EqGlobal>), dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, Ac, Bc, nx, ny, (const fr_t*)nullptr, half, (fr_t*)c->d_scratch, c->d_counters, r_out, r_flag, seq, 1u, G, (fr_t*)d_E);
      } else
      if (NT == 3) hipLaunchKernelGGL((k_cubic_eqw_lb<3, false, TP, EqNone>), dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, Ac, Bc, nx, ny, (const fr_t*)d_E, half, (fr_t*)c->d_scratch, c->d_counters, r_out, r_flag, seq, 0u, EqNone(), (fr_t*)nullptr);
      else if (eqi) hipLaunchKernelGGL((k_cubic_eqw_lb<2, true, TP, EqInline>), dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, Ac, Bc, nx, ny, (const fr_t*)nullptr, half, (fr_t*)c->d_scratch, c->d_counters, r_out, r_flag, seq, 1u, *eqi, (fr_t*)d_E);
      else {
        static const bool lb_nt = [] { const char* v = getenv("LASSO_LB_NT"); return v && v[0] == '1'; }();   // A/B switch: non-temporal loads of A and B in the evaluation-only round
        if (lb_nt && pipe) hipLaunchKernelGGL((k_cubic_eqw_lb<2, false, TP, EqNone, true>), dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, Ac, Bc, nx, ny, (const fr_t*)d_E, half, (fr_t*)c->d_scratch, c->d_counters, r_out, r_flag, seq, pipe, EqNone(), (fr_t*)nullptr);
        else hipLaunchKernelGGL((k_cubic_eqw_lb<2, false, TP, EqNone>), dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, Ac, Bc, nx, ny, (const fr_t*)d_E, half, (fr_t*)c->d_scratch, c->d_counters, r_out, r_flag, seq, pipe, EqNone(), (fr_t*)nullptr);
      }
    }
  } else if (ahead) {
    const size_t q = n / 4;
    const unsigned ny = ncirc, nx = grid_for(q, cubic_nx_cap(ny));
    rc = ensure_scratch(c, (size_t)nx * ny * 3 * sizeof(fr_t)); if (rc) return rc;
    CUBIC_RESULT_ARGS(nx);
    // where the wait lives: in a one-wave gate kernel in front of the round (many workgroups), or in the round's own kernel (few: the gate's second launch costs the host more
    // than a handful of spinning workgroups cost the device — profiles/r05_ahead_wait_forms.txt).  LASSO_AHEAD_INKERNEL_WGS: the bound (default 32; 0 = always the gate)
    static const unsigned inkernel_max = [] { const char* v = getenv("LASSO_AHEAD_INKERNEL_WGS"); const long x = v ? atol(v) : 32; return (unsigned)(x < 0 ? 0 : x > 4096 ? 4096 : x); }();
    const bool bracketed = ((c->prof_mask >> LASSO_K_CUBIC) & 1u) && !(c->prof_mask & 0x40000000u);   // every launch of the family is between profiling events: the wait must not be inside them -> always the gate
    const bool inkernel = nx * ny <= inkernel_max && !bracketed;
    const uint32_t* wait_mail = inkernel ? (const uint32_t*)c->mail_d : (const uint32_t*)nullptr;
    if (!inkernel) hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, c->stream, (const uint32_t*)c->mail_d, c->d_gmail, seq, 0u);   // the wait: one wave; the round starts when it ends
    ProfScope ps(c, LASSO_K_CUBIC, 48.0 * n * (2.0 * ncirc + 1.0));   // the events bracket the round's kernel, not the gate's wait for the host
    if (cubic_wide()) hipLaunchKernelGGL((k_cubic_eqw_fused<2, true, TM, true>), dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, A, B, nx, ny, (const fr_t*)d_E, q, fr_zero(), (fr_t*)c->d_scratch, c->d_counters, r_out, r_flag, seq, (const uint32_t*)c->d_gmail, wait_mail);
    else hipLaunchKernelGGL((k_cubic_eqw_fused<2, false, TM, true>), dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, A, B, nx, ny, (const fr_t*)d_E, q, fr_zero(), (fr_t*)c->d_scratch, c->d_counters, r_out, r_flag, seq, (const uint32_t*)c->d_gmail, wait_mail);
  } else {
    const size_t q = n / 4;
    // bind: read 32n + write 16n per polynomial (SURVEY.md §8d's "fused bind+next-eval" over the reference's 2*ncirc + 1 polynomials)
    ProfScope ps(c, LASSO_K_CUBIC, 48.0 * n * (2.0 * ncirc + 1.0));
    if (q <= CUBIC_SMALL_Q) {
      if (NT == 3) hipLaunchKernelGGL((k_cubic_eqw_small<true, 3, TM>), dim3(ncirc), dim3(LASSO_BLOCK), 0, c->stream, A, B, (const fr_t*)d_E, (uint32_t)q, to_fr(r), c->d_counters, RES(c), seq);
      else hipLaunchKernelGGL((k_cubic_eqw_small<true, 2, TM>), dim3(ncirc), dim3(LASSO_BLOCK), 0, c->stream, A, B, (const fr_t*)d_E, (uint32_t)q, to_fr(r), c->d_counters, RES(c), seq);
    } else {
      const unsigned ny = ncirc, nx = grid_for(q, cubic_nx_cap(ny));
      rc = ensure_scratch(c, (size_t)nx * ny * 3 * sizeof(fr_t)); if (rc) return rc;
      CUBIC_RESULT_ARGS(nx);
      if (NT == 3) hipLaunchKernelGGL((k_cubic_eqw_fused<3, false, TM>), dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, A, B, nx, ny, (const fr_t*)d_E, q, to_fr(r), (fr_t*)c->d_scratch, c->d_counters, r_out, r_flag, seq);
      else if (cubic_wide()) hipLaunchKernelGGL((k_cubic_eqw_fused<2, true, TM>), dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, A, B, nx, ny, (const fr_t*)d_E, q, to_fr(r), (fr_t*)c->d_scratch, c->d_counters, r_out, r_flag, seq);
      else hipLaunchKernelGGL((k_cubic_eqw_fused<2, false, TM>), dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, A, B, nx, ny, (const fr_t*)d_E, q, to_fr(r), (fr_t*)c->d_scratch, c->d_counters, r_out, r_flag, seq);
    }
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
}   // extern "C++"
static int32_t cubic_eqw_launch(lasso_ctx* c, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n, const lasso_fr* r, int NT, uint32_t* seq_out, const EqInline* eqi = nullptr, uint32_t* groups_out = nullptr, bool ahead = false, const EqPointArg* eqg = nullptr, int gate_ell = -1) {
  return ncirc <= 8 ? cubic_eqw_launch_t<MutPtrTable8, PtrTable8>(c, d_A, d_B, ncirc, d_E, n, r, NT, seq_out, eqi, groups_out, ahead, eqg, gate_ell) : cubic_eqw_launch_t<MutPtrTable, PtrTable>(c, d_A, d_B, ncirc, d_E, n, r, NT, seq_out, eqi, groups_out, ahead, eqg, gate_ell);
}
int32_t lasso_sumcheck_cubic_eqw_round(lasso_ctx* c, const lasso_fr* const* d_A, const lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n, lasso_fr* out) {
  REQUIRE(c, d_A && d_B && d_E && out && ncirc >= 1 && ncirc <= LASSO_MAX_PTRS && n >= 2 && (n & (n - 1)) == 0);
  uint32_t seq, groups; int32_t rc = cubic_eqw_launch(c, (lasso_fr* const*)d_A, (lasso_fr* const*)d_B, ncirc, d_E, n, nullptr, 3, &seq, nullptr, &groups); if (rc) return rc;
  return wait_flag(c, seq, (size_t)ncirc * 3, out, c->tagged, groups, 3);
}
int32_t lasso_sumcheck_cubic_eqw_round_fused(lasso_ctx* c, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n, const lasso_fr* r, lasso_fr* out) {
  REQUIRE(c, d_A && d_B && d_E && r && out && ncirc >= 1 && ncirc <= LASSO_MAX_PTRS && n >= 4 && (n & (n - 1)) == 0);
  uint32_t seq, groups; int32_t rc = cubic_eqw_launch(c, d_A, d_B, ncirc, d_E, n, r, 3, &seq, nullptr, &groups); if (rc) return rc;
  return wait_flag(c, seq, (size_t)ncirc * 3, out, c->tagged, groups, 3);
}
// Two-sum form, split into launch and wait so that the host can prepare the round's scalars (one field inversion) while the kernel runs.
// out (lasso_result_wait) = ncirc pairs (q(0), q_inf) — see cubic_eqw_terms2 in poly_kernels.cuh.  One result may be pending per context.
int32_t lasso_sumcheck_cubic_eqw2_begin(lasso_ctx* c, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n, const lasso_fr* r) {
  REQUIRE(c, d_A && d_B && d_E && ncirc >= 1 && ncirc <= LASSO_MAX_PTRS && n >= (r ? 4u : 2u) && (n & (n - 1)) == 0 && !c->pending);
  uint32_t seq, groups; int32_t rc = cubic_eqw_launch(c, d_A, d_B, ncirc, d_E, n, r, 2, &seq, nullptr, &groups); if (rc) return rc;
  c->pending = true; c->pending_seq = seq; c->pending_count = (size_t)ncirc * 2; c->pending_tagged = c->tagged; c->pending_groups = groups; c->pending_K = 2;
  return 0;
}
// The same round enqueued AHEAD of its challenge (k_cubic_eqw_fused<.., AHEAD>): legal while the previous round's result is still pending; the kernel waits on the device for
// lasso_challenge_post, which turns the launch into the context's pending result.  Only streaming two-sum rounds (n / 4 > 64); LASSO_ERR_UNSUPPORTED otherwise (the caller launches
// the ordinary round once it has the challenge).
int32_t lasso_sumcheck_cubic_eqw2_begin_ahead(lasso_ctx* c, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n) {
  REQUIRE(c, d_A && d_B && d_E && ncirc >= 1 && ncirc <= LASSO_MAX_PTRS && n >= 4 && (n & (n - 1)) == 0 && !c->ahead_active && !c->tail_active && !c->defer_next);
  if (n / 4 <= CUBIC_SMALL_Q) return fail(c, LASSO_ERR_UNSUPPORTED, "lasso_sumcheck_cubic_eqw2_begin_ahead: streaming rounds only");
  // everything that could synchronise the stream happens here, BEFORE the launch that waits
  { const unsigned nx = grid_for(n / 4, cubic_nx_cap(ncirc)); int32_t rc0 = ensure_small(c, (size_t)ncirc * 3 * (nx <= direct_nx_max() ? nx : 1)); if (rc0) return rc0; rc0 = ensure_scratch(c, (size_t)nx * ncirc * 3 * sizeof(fr_t)); if (rc0) return rc0; }
  uint32_t seq, groups; int32_t rc = cubic_eqw_launch(c, d_A, d_B, ncirc, d_E, n, nullptr, 2, &seq, nullptr, &groups, true); if (rc) return rc;
  c->ahead_active = true; c->ahead_bullet = false; c->ahead_seq = seq; c->ahead_count = (size_t)ncirc * 2; c->ahead_tagged = c->tagged; c->ahead_groups = groups; c->ahead_K = 2;
  return 0;
}
// the challenge of the round launched ahead: after this call its sums are the context's pending result (lasso_result_wait)
int32_t lasso_challenge_post(lasso_ctx* c, const lasso_fr* r) {
  REQUIRE(c, r && c->ahead_active && !c->ahead_bullet && !c->pending);
  post_mail(c, c->ahead_seq, (const uint32_t*)r);
  c->ahead_active = false;
  c->pending = true; c->pending_seq = c->ahead_seq; c->pending_count = c->ahead_count; c->pending_tagged = c->ahead_tagged; c->pending_groups = c->ahead_groups; c->pending_K = c->ahead_K;
  return 0;
}
// 1 when rounds may be launched ahead on this context (LASSO_ROUNDS_AHEAD=0: A/B switch)
int32_t lasso_rounds_ahead_ok(lasso_ctx* c) { static const bool off = [] { const char* v = getenv("LASSO_ROUNDS_AHEAD"); return v && v[0] == '0'; }(); return c && !off && c->mail_d && c->d_gmail ? 1 : 0; }
// The next entry point that hands its result over through the mapped buffer (the sumcheck rounds, the few-row MSMs, lasso_bullet_round ...)
// returns right after its launch; its `out` argument is ignored and lasso_result_wait(ctx, out, count) delivers the same values
// (count in field-element units: a point is 4).  Lets the host absorb transcript data or do scalar work while the device computes.
// The tail of a layer's sumcheck in ONE resident kernel (k_cubic_tail): from q <= 64 indices per circuit on, the remaining rounds are served
// without a launch per round — the host posts each challenge into a host-mapped mailbox and the kernel answers through the result buffer.
// begin: r == NULL starts at the first round of a layer (arrays of length n = 2q), otherwise the challenge r is bound first (n = 4q).  The
// result of the first of the log2(2q) rounds is pending afterwards (lasso_result_wait, 2*ncirc values: (q(0), q_inf) per circuit).
// next: posts a challenge; pending: the next round's sums, or after the last round the 2*ncirc bound heads (A_0.., B_0..).
// The arrays in device memory are NOT updated (nothing reads a layer's arrays after its sumcheck).
static int32_t cubic_tail_begin_impl(lasso_ctx* c, uint32_t m_stop, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n, const lasso_fr* r, const EqInline* eqi, bool ahead, int gate_ell = -1);
// lasso_tail_handover_next's one-shot setting is consumed by the NEXT tail entry point whatever its outcome: every public wrapper takes it first thing, before any argument check can
// return (a REQUIRE in a wrapper used to leave it armed for the next, unrelated tail)
static uint32_t take_handover(lasso_ctx* c) { if (!c) return 0; const uint32_t m = c->handover_next; c->handover_next = 0; return m; }
int32_t lasso_sumcheck_cubic_tail_begin(lasso_ctx* c, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n, const lasso_fr* r) {
  const uint32_t ho = take_handover(c);
  REQUIRE(c, d_E);
  return cubic_tail_begin_impl(c, ho, d_A, d_B, ncirc, d_E, n, r, nullptr, false);
}
static bool make_eq_inline(const lasso_fr* point, uint32_t ell, const lasso_fr* scale, EqInline& Q) {
  if (ell > 14 || (ell && !point)) return false;
  for (uint32_t j = 0; j < 14; j++) Q.r[j] = j < ell ? to_fr(point + j) : fr_zero();
  Q.scale = scale ? to_fr(scale) : fr_one(); Q.ell = ell; return true;
}
// The first round of a layer with the layer's eq table  E = *scale * EqPolynomial(point[0..ell)).evals()  (2^ell = n/2 entries) built inside the launch (k_cubic_eqw_lb<2, true>):
// the same pending result as lasso_sumcheck_cubic_eqw2_begin(.., d_E, n, NULL) after lasso_eq_evals_scaled(point, ell, scale, d_E), and d_E holds the same table afterwards.
int32_t lasso_sumcheck_cubic_eqw2_begin_eq(lasso_ctx* c, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, lasso_fr* d_E_out, size_t n, const lasso_fr* point, uint32_t ell, const lasso_fr* scale) {
  REQUIRE(c, d_A && d_B && d_E_out && ncirc >= 1 && ncirc <= LASSO_MAX_PTRS && n >= 2 && (n & (n - 1)) == 0 && !c->pending && ((size_t)1 << ell) == n / 2);
  EqInline Q;
  if (n / 2 <= CUBIC_SMALL_Q || (ell && !point) || ell > 32) return fail(c, LASSO_ERR_UNSUPPORTED, "lasso_sumcheck_cubic_eqw2_begin_eq: tables of 2^7 .. 2^32 entries only");
  uint32_t seq, groups; int32_t rc;
  if (make_eq_inline(point, ell, scale, Q)) rc = cubic_eqw_launch(c, d_A, d_B, ncirc, d_E_out, n, nullptr, 2, &seq, &Q, &groups);     // <= 2^14 entries: factor tables in LDS
  else { EqPointArg G; G.point = point; G.ell = ell; G.scale = scale ? to_fr(scale) : fr_one(); rc = cubic_eqw_launch(c, d_A, d_B, ncirc, d_E_out, n, nullptr, 2, &seq, nullptr, &groups, false, &G); }   // larger: factor tables in memory
  if (rc) return rc;
  c->pending = true; c->pending_seq = seq; c->pending_count = (size_t)ncirc * 2; c->pending_tagged = c->tagged; c->pending_groups = groups; c->pending_K = 2;
  return 0;
}
// lasso_sumcheck_cubic_tail_begin(.., r = NULL) without a table: the resident kernel derives E = *scale * EqPolynomial(point[0..ell)).evals(), 2^ell = n/2 <= capacity, itself
int32_t lasso_sumcheck_cubic_tail_begin_eq(lasso_ctx* c, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, size_t n, const lasso_fr* point, uint32_t ell, const lasso_fr* scale) {
  const uint32_t ho = take_handover(c);
  REQUIRE(c, n >= 2 && ((size_t)1 << ell) == n / 2 && ell <= 9);
  EqInline Q; if (!make_eq_inline(point, ell, scale, Q)) return fail(c, LASSO_ERR_INVALID, "lasso_sumcheck_cubic_tail_begin_eq: bad point");
  return cubic_tail_begin_impl(c, ho, d_A, d_B, ncirc, nullptr, n, nullptr, &Q, false);
}
extern "C++" {
template <class TM>
static int32_t cubic_tail_begin_t(lasso_ctx* c, uint32_t m_stop_arg, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n, const lasso_fr* r, const EqInline* eqi, bool ahead, int gate_ell) {
  const uint32_t m_stop = m_stop_arg ? m_stop_arg : 1u;   // the caller's take_handover(): explicit, not sticky context state
  const bool gated = gate_ell >= 0;   // the whole layer enqueued ahead of its eq point, possibly behind the previous layer's (still active) tail: the point comes through k_gate_point
  REQUIRE(c, d_A && d_B && (d_E || eqi || gated) && ncirc >= 1 && ncirc <= LASSO_MAX_PTRS && n >= ((r || ahead) ? 4u : 2u) && (n & (n - 1)) == 0 && (ahead || gated || !c->pending) && (gated || !c->tail_active) && !c->defer_next && !c->ahead_active && !c->lay_active);
  const size_t q = (r || ahead) ? n / 4 : n / 2;
  REQUIRE(c, q >= 1 && q <= CUBIC_TAIL_Q && (m_stop & (m_stop - 1)) == 0 && m_stop <= 128 && 2 * q > m_stop);   // at least one round of sums before the arrays are handed over
  TM A, B; for (uint32_t i = 0; i < ncirc; i++) { REQUIRE(c, d_A[i] && d_B[i]); A.p[i] = (fr_t*)d_A[i]; B.p[i] = (fr_t*)d_B[i]; }
  int32_t rc = ensure_small(c, (size_t)ncirc * 2 * (m_stop > 2 ? m_stop : 2)); if (rc) return rc;
  uint32_t turns = 0; while (((size_t)m_stop << turns) < 2 * q) turns++;   // rounds of sums; one more publication carries the heads (or the arrays of m_stop elements)
  const uint32_t seq0 = next_seq(c, turns + 1);
  // workgroup = capacity: 256 threads / 74 KB of LDS up to 256 indices per circuit, 512 threads / 147 KB above
#define LAUNCH_CTAIL(B_, Q_, I_, TE_, R_, EQ_) hipLaunchKernelGGL((k_cubic_tail<B_, Q_, I_, TM, TE_>), dim3(ncirc), dim3(Q_), 0, c->stream, A, B, (const fr_t*)d_E, (uint32_t)q, R_, (const uint32_t*)c->mail_d, c->d_counters, RES(c), seq0, EQ_, m_stop, ahead ? 1u : 0u)
  if (gated) {
    hipLaunchKernelGGL(k_gate_point, dim3(1), dim3(64), 0, c->stream, (const uint32_t*)c->pmail_d, c->d_gpoint, seq0, (uint32_t)gate_ell + 2u, c->pmail_d + LASSO_PMAIL_ACK_WORD); c->gate_sent = seq0;
    EqInlineMem M; M.gp = c->d_gpoint; M.seq = seq0; M.ell = (uint32_t)gate_ell;
    if (q <= 256) LAUNCH_CTAIL(false, 256, true, EqInlineMem, fr_zero(), M); else LAUNCH_CTAIL(false, 512, true, EqInlineMem, fr_zero(), M);
    HIPCHK(c, hipGetLastError());
    c->lay_active = true; c->lay_tail = true; c->lay_seq = seq0; c->lay_ell = (uint32_t)gate_ell; c->lay_turns = turns; c->lay_count = (size_t)ncirc * 2; c->lay_final = (size_t)ncirc * 2 * m_stop; c->lay_tagged = c->tagged;
    return 0;   // the tail state of the context still belongs to the previous layer: lasso_point_post installs this one
  }
  if (eqi) { if (q <= 256) LAUNCH_CTAIL(false, 256, true, EqInline, fr_zero(), *eqi); else LAUNCH_CTAIL(false, 512, true, EqInline, fr_zero(), *eqi); }
  else if (q <= 256) { if (r || ahead) LAUNCH_CTAIL(true, 256, false, EqNone, r ? to_fr(r) : fr_zero(), EqNone()); else LAUNCH_CTAIL(false, 256, false, EqNone, fr_zero(), EqNone()); }
  else { if (r || ahead) LAUNCH_CTAIL(true, 512, false, EqNone, r ? to_fr(r) : fr_zero(), EqNone()); else LAUNCH_CTAIL(false, 512, false, EqNone, fr_zero(), EqNone()); }
  HIPCHK(c, hipGetLastError());
  c->tail_active = true; c->tail_seq0 = seq0; c->tail_turn = 0; c->tail_turns = turns; c->tail_count = (size_t)ncirc * 2; c->tail_final = (size_t)ncirc * 2 * m_stop;
  if (ahead) { c->tail_unstarted = true; return 0; }   // nothing is pending until the first challenge has been posted (lasso_sumcheck_cubic_tail_next)
  c->pending = true; c->pending_seq = seq0; c->pending_count = c->tail_count; c->pending_tagged = c->tagged;
  return 0;
}
}   // extern "C++"
static int32_t cubic_tail_begin_impl(lasso_ctx* c, uint32_t m_stop, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n, const lasso_fr* r, const EqInline* eqi, bool ahead, int gate_ell) {
  return ncirc <= 8 ? cubic_tail_begin_t<MutPtrTable8>(c, m_stop, d_A, d_B, ncirc, d_E, n, r, eqi, ahead, gate_ell) : cubic_tail_begin_t<MutPtrTable>(c, m_stop, d_A, d_B, ncirc, d_E, n, r, eqi, ahead, gate_ell);
}
// ---- A LAYER enqueued ahead of its eq point (round 5).  Between two layers of a grand-product argument the device used to idle for the host's last rounds, the layer's closing
// Fiat-Shamir step AND the launch + dispatch of the next layer's first kernel(s) (20-40 us per transition, ~30 transitions per proof).  With these entry points the next layer's
// first launch — round 0 with its eq table built inside (lasso_sumcheck_cubic_eqw2_begin_eq), or the resident tail that serves a small layer whole
// (lasso_sumcheck_cubic_tail_begin_eq) — is enqueued while the CURRENT layer's resident tail is still answering (legal with a tail active and a result pending), behind a one-wave
// gate that waits for the point; lasso_point_post delivers point and scale and turns the launch into the context's pending result (and active tail), exactly as if the plain entry
// point had been called then; lasso_point_cancel ends the enqueued kernels without a result (a layer whose shape turned out different).  Nothing may grow while a tail is resident:
// LASSO_ERR_UNSUPPORTED when a buffer would have to (the caller takes the plain path after the layer).
// the ONE mailbox area is free when the last gate launched has ended (it acknowledges with its sequence number); until then a new layer may not be enqueued ahead — its
// post / cancel would overwrite a message that gate has not read yet, and the gate would spin to its 5 s bail-out.  prover.hpp never gets here with a gate in
// flight (a layer's first result is collected before the next one is enqueued); the entry points enforce it for every other caller: LASSO_ERR_UNSUPPORTED = take the plain path.
static bool gate_free(lasso_ctx* c) { return !c->gate_sent || __atomic_load_n(c->pmail_h + LASSO_PMAIL_ACK_WORD, __ATOMIC_ACQUIRE) == c->gate_sent; }
int32_t lasso_layer_ahead_ok(lasso_ctx* c) { static const bool off = [] { const char* v = getenv("LASSO_LAYER_AHEAD"); return v && v[0] == '0'; }(); return c && !off && c->pmail_d && c->d_gpoint ? 1 : 0; }
int32_t lasso_sumcheck_cubic_eqw2_begin_eq_ahead(lasso_ctx* c, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, lasso_fr* d_E_out, size_t n, uint32_t ell) {
  REQUIRE(c, d_A && d_B && d_E_out && ncirc >= 1 && ncirc <= LASSO_MAX_PTRS && n >= 2 && (n & (n - 1)) == 0 && ell < 48 && ((size_t)1 << ell) == n / 2 && !c->lay_active && !c->ahead_active && !c->defer_next);
  if (!lasso_layer_ahead_ok(c) || n / 2 <= CUBIC_SMALL_Q || ell > 32 || ell > LASSO_POINT_MAX) return fail(c, LASSO_ERR_UNSUPPORTED, "lasso_sumcheck_cubic_eqw2_begin_eq_ahead: tables of 2^7 .. 2^32 entries only");
  if (!gate_free(c)) return fail(c, LASSO_ERR_UNSUPPORTED, "lasso_sumcheck_cubic_eqw2_begin_eq_ahead: the previous point gate has not consumed its message yet");
  uint32_t seq, groups;
  c->no_grow = true;
  const int32_t rc = cubic_eqw_launch(c, d_A, d_B, ncirc, d_E_out, n, nullptr, 2, &seq, nullptr, &groups, false, nullptr, (int)ell);
  c->no_grow = false;
  if (rc == LASSO_ERR_UNSUPPORTED) return fail(c, rc, "lasso_sumcheck_cubic_eqw2_begin_eq_ahead: a buffer would have to grow while kernels are in flight");
  if (rc) return rc;
  HIPCHK(c, hipGetLastError());
  c->lay_active = true; c->lay_tail = false; c->lay_seq = seq; c->lay_ell = ell; c->lay_count = (size_t)ncirc * 2; c->lay_tagged = c->tagged; c->lay_groups = groups; c->lay_K = 2;
  return 0;
}
int32_t lasso_sumcheck_cubic_tail_begin_eq_ahead(lasso_ctx* c, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, size_t n, uint32_t ell) {
  const uint32_t ho = take_handover(c);
  REQUIRE(c, n >= 2 && ell <= 9 && ((size_t)1 << ell) == n / 2);
  if (!lasso_layer_ahead_ok(c)) return fail(c, LASSO_ERR_UNSUPPORTED, "lasso_sumcheck_cubic_tail_begin_eq_ahead: switched off");
  if (!gate_free(c)) return fail(c, LASSO_ERR_UNSUPPORTED, "lasso_sumcheck_cubic_tail_begin_eq_ahead: the previous point gate has not consumed its message yet");
  c->no_grow = true;
  const int32_t rc = cubic_tail_begin_impl(c, ho, d_A, d_B, ncirc, nullptr, n, nullptr, nullptr, false, (int)ell);
  c->no_grow = false;
  if (rc == LASSO_ERR_UNSUPPORTED) return fail(c, rc, "lasso_sumcheck_cubic_tail_begin_eq_ahead: a buffer would have to grow while kernels are in flight");
  return rc;
}
static void point_mail(lasso_ctx* c, uint32_t ell, const lasso_fr* point, const lasso_fr* scale, uint32_t ctrl) {
  const uint32_t zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const fr_t one = fr_one();
  for (uint32_t j = 0; j < ell; j++) mail_chunks(c->pmail_h + 12 * j, c->lay_seq, point ? (const uint32_t*)(point + j) : zero8);
  mail_chunks(c->pmail_h + 12 * ell, c->lay_seq, scale ? (const uint32_t*)scale : one.v);
  const uint32_t ctl[8] = {ctrl, 0, 0, 0, 0, 0, 0, 0};
  mail_chunks(c->pmail_h + 12 * (ell + 1), c->lay_seq, ctl);
#if defined(__SSE2__)
  _mm_sfence();
#else
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}
// point[0..ell) and *scale (NULL = 1) of the layer enqueued ahead: afterwards the context is where the plain entry point would have left it (first round's sums pending;
// the resident tail active).  Legal once the previous layer's tail has ended and its last result has been collected.
int32_t lasso_point_post(lasso_ctx* c, const lasso_fr* point, uint32_t ell, const lasso_fr* scale) {
  REQUIRE(c, c && c->lay_active && ell == c->lay_ell && (point || !ell) && !c->pending && !c->tail_active && !c->ahead_active);
  point_mail(c, ell, point, scale, 0u);
  c->lay_active = false;
  if (c->lay_tail) {
    c->tail_active = true; c->tail_seq0 = c->lay_seq; c->tail_turn = 0; c->tail_turns = c->lay_turns; c->tail_count = c->lay_count; c->tail_final = c->lay_final; c->tail_unstarted = false;
    c->pending = true; c->pending_seq = c->lay_seq; c->pending_count = c->lay_count; c->pending_tagged = c->lay_tagged; c->pending_groups = 1; c->pending_K = 0;
  } else {
    c->pending = true; c->pending_seq = c->lay_seq; c->pending_count = c->lay_count; c->pending_tagged = c->lay_tagged; c->pending_groups = c->lay_groups; c->pending_K = c->lay_K;
  }
  c->lay_tail = false;
  return 0;
}
// the layer enqueued ahead is not wanted after all: its kernels end without touching anything (stream order: whatever is enqueued next runs after them)
int32_t lasso_point_cancel(lasso_ctx* c) {
  REQUIRE(c, c && c->lay_active);
  point_mail(c, c->lay_ell, nullptr, nullptr, 1u);
  c->lay_active = false; c->lay_tail = false;
  return 0;
}
// The resident tail enqueued AHEAD of the challenge it binds first (n = 4q): legal while the previous round's result is pending; the first lasso_sumcheck_cubic_tail_next
// posts that challenge and makes the first round's sums the pending result.
int32_t lasso_sumcheck_cubic_tail_begin_ahead(lasso_ctx* c, lasso_fr* const* d_A, lasso_fr* const* d_B, uint32_t ncirc, const lasso_fr* d_E, size_t n) {
  const uint32_t ho = take_handover(c);
  REQUIRE(c, d_E);
  return cubic_tail_begin_impl(c, ho, d_A, d_B, ncirc, d_E, n, nullptr, nullptr, true);
}
// The next lasso_sumcheck_cubic_tail_begin* stops when its arrays are down to m_stop elements each (a power of two, 2 <= m_stop <= 128, below the arrays' length at the first
// round) and its LAST publication is the arrays instead of the heads: 2 * ncirc * m_stop values, A_0[0..m_stop), A_1[..], .., B_0[..], ...  m_stop = 1 or 0: the heads.
int32_t lasso_tail_handover_next(lasso_ctx* c, uint32_t m_stop) {
  REQUIRE(c, c && (m_stop & (m_stop - 1)) == 0 && m_stop <= 128);   // (legal with a tail active: the next begin may be a layer enqueued ahead, lasso_sumcheck_cubic_tail_begin_eq_ahead)
  c->handover_next = m_stop <= 1 ? 0 : m_stop; return 0;
}
// The same for the primary sumcheck of a linear strategy (k_linear_tail): per round two dot products per polynomial, out[2k] = S0_k, out[2k+1] = S1_k
// (as lasso_sumcheck_linear_eqw_round, without the unused third slot); after the last challenge the heads out[k] = polys_k[0] (alpha values).
// d_src is only read (r == NULL: arrays of length n = 2q; otherwise bound with r first, n = 4q).  Challenges go through lasso_sumcheck_cubic_tail_next.
int32_t lasso_sumcheck_linear_tail_begin(lasso_ctx* c, const lasso_fr* const* d_src, uint32_t alpha, const lasso_fr* d_E, size_t n, const lasso_fr* r) {
  const uint32_t ho = take_handover(c);   // the linear tail has no hand-over form: refused, and the one-shot setting does not survive the refusal
  REQUIRE(c, d_src && d_E && alpha >= 1 && alpha <= LASSO_MAX_PTRS && n >= (r ? 4u : 2u) && (n & (n - 1)) == 0 && !c->pending && !c->tail_active && !c->defer_next && !ho && !c->ahead_active);
  const size_t q = r ? n / 4 : n / 2;
  REQUIRE(c, q >= 1 && q <= CUBIC_TAIL_Q);
  PtrTable Src; for (uint32_t i = 0; i < alpha; i++) { REQUIRE(c, d_src[i]); Src.p[i] = (const fr_t*)d_src[i]; }
  int32_t rc = ensure_small(c, (size_t)alpha * 3); if (rc) return rc;
  uint32_t turns = 0; while (((size_t)1 << turns) < 2 * q) turns++;
  const uint32_t seq0 = next_seq(c, turns + 1);
#define LAUNCH_LTAIL(B_, Q_, R_) hipLaunchKernelGGL((k_linear_tail<B_, Q_>), dim3(alpha), dim3(Q_), 0, c->stream, Src, (const fr_t*)d_E, (uint32_t)q, R_, (const uint32_t*)c->mail_d, c->d_counters, RES(c), seq0)
  if (q <= 256) { if (r) LAUNCH_LTAIL(true, 256, to_fr(r)); else LAUNCH_LTAIL(false, 256, fr_zero()); }
  else { if (r) LAUNCH_LTAIL(true, 512, to_fr(r)); else LAUNCH_LTAIL(false, 512, fr_zero()); }
  HIPCHK(c, hipGetLastError());
  c->tail_active = true; c->tail_seq0 = seq0; c->tail_turn = 0; c->tail_turns = turns; c->tail_count = (size_t)alpha * 2; c->tail_final = alpha;
  c->pending = true; c->pending_seq = seq0; c->pending_count = c->tail_count; c->pending_tagged = c->tagged;
  return 0;
}
int32_t lasso_sumcheck_cubic_tail_next(lasso_ctx* c, const lasso_fr* r) {
  REQUIRE(c, r && c->tail_active && !c->pending);
  const fr_t rr = to_fr(r);
  if (c->tail_unstarted) {   // launched ahead: this is the challenge the kernel binds first; it enables publication tail_seq0 (the first round's sums)
    c->tail_unstarted = false;
    post_mail(c, c->tail_seq0, rr.v);
    c->pending = true; c->pending_seq = c->tail_seq0; c->pending_count = c->tail_count; c->pending_tagged = c->tagged;
    return 0;
  }
  c->tail_turn++;
  const uint32_t tn = c->tail_seq0 + c->tail_turn;   // = the sequence number of the publication this challenge enables: tags are unique, the mailbox is never reset
                                                      // (a reset could erase a challenge some workgroup of a multi-workgroup kernel has not read yet)
  post_mail(c, tn, rr.v);
  const size_t cnt = c->tail_turn == c->tail_turns ? c->tail_final : c->tail_count;
  if (cnt) { c->pending = true; c->pending_seq = c->tail_seq0 + c->tail_turn; c->pending_count = cnt; c->pending_tagged = c->tagged; }
  if (c->tail_turn == c->tail_turns) c->tail_active = false;
  return 0;
}
// largest q (indices per circuit / polynomial) lasso_sumcheck_{cubic,linear}_tail_begin accept; LASSO_TAIL_Q=256 restores round 2's capacity (A/B measurements)
uint32_t lasso_sumcheck_tail_capacity(void) { static const uint32_t q = [] { const char* v = getenv("LASSO_TAIL_Q"); const long x = v ? atol(v) : 0; return (uint32_t)(x == 256 ? 256 : CUBIC_TAIL_Q); }(); return q; }
int32_t lasso_defer_next(lasso_ctx* c) { REQUIRE(c, !c->pending && !c->defer_next); c->defer_next = true; return 0; }
int32_t lasso_result_wait(lasso_ctx* c, lasso_fr* out, size_t count) {
  REQUIRE(c, out && c->pending && count == c->pending_count);
  c->pending = false;
  { const uint32_t g = c->pending_groups, K = c->pending_K; c->pending_groups = 1; c->pending_K = 0; return wait_flag(c, c->pending_seq, count, out, c->pending_tagged, g, K); }
}
// eq-weighted rounds of prove_arbitrary for the linear strategies (k_dot_eqw_* in poly_kernels.cuh)
int32_t lasso_sumcheck_linear_eqw_round(lasso_ctx* c, const lasso_fr* const* d_polys, uint32_t alpha, const lasso_fr* d_E, size_t n, lasso_fr* out) {
  REQUIRE(c, d_polys && d_E && out && alpha >= 1 && alpha <= LASSO_MAX_PTRS && n >= 2 && (n & (n - 1)) == 0);
  PtrTable P; for (uint32_t i = 0; i < alpha; i++) { REQUIRE(c, d_polys[i]); P.p[i] = (const fr_t*)d_polys[i]; }
  const size_t half = n / 2; const unsigned ny = alpha, nx = grid_for(half, cubic_nx_cap(ny));
  int32_t rc = ensure_small(c, (size_t)alpha * 3); if (rc) return rc;
  rc = ensure_scratch(c, (size_t)nx * ny * 3 * sizeof(fr_t)); if (rc) return rc;
  const uint32_t seq = next_seq(c);
  {
    ProfScope ps(c, LASSO_K_COMBINE, 32.0 * n * (alpha + 1.0));
    hipLaunchKernelGGL(k_dot_eqw_lb, dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, P, nx, ny, (const fr_t*)d_E, half, (fr_t*)c->d_scratch, c->d_counters, RES(c), seq);
  }
  HIPCHK(c, hipGetLastError());
  return wait_flag(c, seq, (size_t)alpha * 3, out, c->tagged);
}
int32_t lasso_sumcheck_linear_eqw_round_fused(lasso_ctx* c, lasso_fr* const* d_polys, uint32_t alpha, const lasso_fr* d_E, size_t n, const lasso_fr* r, lasso_fr* out) {
  return lasso_sumcheck_linear_eqw_round_fused_from(c, (const lasso_fr* const*)d_polys, d_polys, alpha, d_E, n, r, out);
}
int32_t lasso_sumcheck_linear_eqw_round_fused_from(lasso_ctx* c, const lasso_fr* const* d_src, lasso_fr* const* d_polys, uint32_t alpha, const lasso_fr* d_E, size_t n, const lasso_fr* r, lasso_fr* out) {
  REQUIRE(c, d_src && d_polys && d_E && r && out && alpha >= 1 && alpha <= LASSO_MAX_PTRS && n >= 4 && (n & (n - 1)) == 0);
  MutPtrTable P; PtrTable Src; for (uint32_t i = 0; i < alpha; i++) { REQUIRE(c, d_polys[i] && d_src[i]); P.p[i] = (fr_t*)d_polys[i]; Src.p[i] = (const fr_t*)d_src[i]; }
  const size_t q = n / 4; const unsigned ny = alpha, nx = grid_for(q, cubic_nx_cap(ny));
  int32_t rc = ensure_small(c, (size_t)alpha * 3); if (rc) return rc;
  rc = ensure_scratch(c, (size_t)nx * ny * 3 * sizeof(fr_t)); if (rc) return rc;
  const uint32_t seq = next_seq(c);
  {
    // bind (48 n per polynomial, the reference's alpha + 1 of them) with the next round's sums riding on the same pass
    ProfScope ps(c, LASSO_K_BIND, 48.0 * n * (alpha + 1.0));
    hipLaunchKernelGGL(k_dot_eqw_fused<false>, dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, Src, P, nx, ny, (const fr_t*)d_E, q, to_fr(r), (fr_t*)c->d_scratch, c->d_counters, RES(c), seq);
  }
  HIPCHK(c, hipGetLastError());
  return wait_flag(c, seq, (size_t)alpha * 3, out, c->tagged);
}
// lasso_sumcheck_linear_eqw_round_fused(.., r, out) enqueued AHEAD of r (in place; gate kernel + k_dot_eqw_fused<AHEAD>): lasso_challenge_post releases it, lasso_result_wait
// delivers the 3 * alpha values.  Same rules as lasso_sumcheck_cubic_eqw2_begin_ahead.
int32_t lasso_sumcheck_linear_eqw_round_fused_ahead(lasso_ctx* c, lasso_fr* const* d_polys, uint32_t alpha, const lasso_fr* d_E, size_t n) {
  REQUIRE(c, d_polys && d_E && alpha >= 1 && alpha <= LASSO_MAX_PTRS && n >= 4 && (n & (n - 1)) == 0 && !c->ahead_active && !c->tail_active && !c->defer_next);
  MutPtrTable P; PtrTable Src; for (uint32_t i = 0; i < alpha; i++) { REQUIRE(c, d_polys[i]); P.p[i] = (fr_t*)d_polys[i]; Src.p[i] = (const fr_t*)d_polys[i]; }
  const size_t q = n / 4; const unsigned ny = alpha, nx = grid_for(q, cubic_nx_cap(ny));
  int32_t rc = ensure_small(c, (size_t)alpha * 3); if (rc) return rc;
  rc = ensure_scratch(c, (size_t)nx * ny * 3 * sizeof(fr_t)); if (rc) return rc;
  const uint32_t seq = next_seq(c);
  hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, c->stream, (const uint32_t*)c->mail_d, c->d_gmail, seq, 0u);
  {
    ProfScope ps(c, LASSO_K_BIND, 48.0 * n * (alpha + 1.0));
    hipLaunchKernelGGL(k_dot_eqw_fused<true>, dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, Src, P, nx, ny, (const fr_t*)d_E, q, fr_zero(), (fr_t*)c->d_scratch, c->d_counters, RES(c), seq, (const uint32_t*)c->d_gmail);
  }
  HIPCHK(c, hipGetLastError());
  c->ahead_active = true; c->ahead_bullet = false; c->ahead_seq = seq; c->ahead_count = (size_t)alpha * 3; c->ahead_tagged = c->tagged; c->ahead_groups = 1; c->ahead_K = 0;
  return 0;
}
// the first round and the first bind of the primary sumcheck from the lookup polynomials' integer values (k_dot_eqw_lb_u32 / k_dot_eqw_fused_from_u32)
int32_t lasso_sumcheck_linear_eqw_round_u32(lasso_ctx* c, const uint32_t* const* d_u32, uint32_t alpha, const lasso_fr* d_E, size_t n, lasso_fr* out) {
  REQUIRE(c, d_u32 && d_E && out && alpha >= 1 && alpha <= LASSO_MAX_PTRS && n >= 2 && (n & (n - 1)) == 0);
  PtrTableU32 P; for (uint32_t i = 0; i < alpha; i++) { REQUIRE(c, d_u32[i]); P.p[i] = d_u32[i]; }
  const size_t half = n / 2; const unsigned ny = alpha, nx = grid_for(half, cubic_nx_cap(ny));
  int32_t rc = ensure_small(c, (size_t)alpha * 3); if (rc) return rc;
  rc = ensure_scratch(c, (size_t)nx * ny * 3 * sizeof(fr_t)); if (rc) return rc;
  const uint32_t seq = next_seq(c);
  {
    ProfScope ps(c, LASSO_K_COMBINE, 32.0 * n * (alpha + 1.0));   // SURVEY 8(d)'s bytes of the reference's round; the kernel reads 4 n per polynomial + 16 n of the table
    hipLaunchKernelGGL(k_dot_eqw_lb_u32, dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, P, nx, ny, (const fr_t*)d_E, half, (fr_t*)c->d_scratch, c->d_counters, RES(c), seq);
  }
  HIPCHK(c, hipGetLastError());
  return wait_flag(c, seq, (size_t)alpha * 3, out, c->tagged);
}
int32_t lasso_sumcheck_linear_eqw_round_fused_from_u32(lasso_ctx* c, const uint32_t* const* d_u32, lasso_fr* const* d_polys, uint32_t alpha, const lasso_fr* d_E, size_t n, const lasso_fr* r, lasso_fr* out) {
  REQUIRE(c, d_u32 && d_polys && d_E && r && out && alpha >= 1 && alpha <= LASSO_MAX_PTRS && n >= 4 && (n & (n - 1)) == 0);
  MutPtrTable P; PtrTableU32 Src; for (uint32_t i = 0; i < alpha; i++) { REQUIRE(c, d_polys[i] && d_u32[i]); P.p[i] = (fr_t*)d_polys[i]; Src.p[i] = d_u32[i]; }
  const size_t q = n / 4; const unsigned ny = alpha, nx = grid_for(q, cubic_nx_cap(ny));
  int32_t rc = ensure_small(c, (size_t)alpha * 3); if (rc) return rc;
  rc = ensure_scratch(c, (size_t)nx * ny * 3 * sizeof(fr_t)); if (rc) return rc;
  const uint32_t seq = next_seq(c);
  {
    ProfScope ps(c, LASSO_K_BIND, 48.0 * n * (alpha + 1.0));
    hipLaunchKernelGGL(k_dot_eqw_fused_from_u32, dim3(nx * ny), dim3(LASSO_BLOCK), 0, c->stream, Src, P, nx, ny, (const fr_t*)d_E, q, to_fr(r), (fr_t*)c->d_scratch, c->d_counters, RES(c), seq);
  }
  HIPCHK(c, hipGetLastError());
  return wait_flag(c, seq, (size_t)alpha * 3, out, c->tagged);
}
static int32_t make_strategy(lasso_ctx* c, const lasso_strategy* s, StrategyDev& S, WeightTable& W) {
  REQUIRE(c, s && s->kind >= LASSO_AND && s->kind <= LASSO_SPARK_UNCONFIRMED && s->c >= 1);
  S.kind = s->kind; S.c = s->c; S.log_m = s->log_m; S.log_r = s->log_r;
  S.alpha = s->kind == LASSO_LT ? 2 * s->c : s->c;
  REQUIRE(c, S.alpha <= LASSO_MAX_ALPHA);
  for (uint32_t i = 0; i < LASSO_MAX_ALPHA; i++) W.w[i] = fr_zero();
  if (s->kind != LASSO_LT && s->kind != LASSO_SPARK_UNCONFIRMED) {
    const uint32_t inc = s->kind == LASSO_RANGE ? s->log_m : s->log_m / 2;  // and.rs:46 / range_check.rs:79
    for (uint32_t i = 0; i < S.alpha; i++) { REQUIRE(c, i * inc < 64); W.w[i] = fr_from_u64((uint64_t)1 << (i * inc)); }  // `1u64 << ...` in the reference overflows beyond 63
  }
  return 0;
}
#define DISPATCH_A(alpha, FN) do { if ((alpha) <= 2) { FN(2, 2); } else if ((alpha) <= 4) { FN(4, 3); } else if ((alpha) <= 8) { FN(8, 5); } else if ((alpha) <= 16) { FN(16, 9); } else { FN(32, 17); } } while (0)
// the LT round kernel: (bound on NUM_MEMORIES, bound on the degree, lanes per index) — at most 6 evaluation points per lane
#define DISPATCH_LT(alpha, FN) do { if ((alpha) <= 2) { FN(2, 2, 1); } else if ((alpha) <= 4) { FN(4, 3, 1); } else if ((alpha) <= 8) { FN(8, 5, 1); } else if ((alpha) <= 16) { FN(16, 9, 2); } else { FN(32, 17, 3); } } while (0)
static fr_t lt_pow32(uint32_t e, bool inverse) { static const fr_t inv32 = fr_inv(fr_from_u64(32)); const fr_t b = inverse ? inv32 : fr_from_u64(32); fr_t r = fr_one(); for (uint32_t i = 0; i < e; i++) r = fr_mul(r, b); return r; }
// LT: the round kernel works on arrays whose LT memories carry the factor 32^-(C-1-m) (k_combine_round_lt).  lasso_sumcheck_combine_round keeps the literal contract (plain arrays in,
// sumcheck.rs:165-237's evaluations out): it scales COPIES of the C LT arrays first; the prover scales its work arrays once (lasso_lt_prescale) and calls the _lt_scaled form every round.
static int32_t combine_round_impl(lasso_ctx* c, const lasso_strategy* s, const lasso_fr* const* d_polys, const lasso_fr* d_eq, size_t n, uint32_t degree, lasso_fr* out, bool lt_scaled) {
  StrategyDev S; WeightTable W; int32_t rc = make_strategy(c, s, S, W); if (rc) return rc;
  REQUIRE(c, d_polys && d_eq && out && n >= 2 && (n & (n - 1)) == 0);
  const bool spark = s->kind == LASSO_SPARK_UNCONFIRMED;
  REQUIRE(c, degree == ((s->kind == LASSO_LT || spark) ? s->c + 1 : 2));   // sumcheck_poly_degree(): subtables/mod.rs:60-62
  PtrTable P; for (uint32_t i = 0; i < S.alpha; i++) { REQUIRE(c, d_polys[i]); P.p[i] = (const fr_t*)d_polys[i]; }
  const size_t half = n / 2; const unsigned nx = grid_for(half, 1024); const uint32_t K = degree + 1;
  const bool lt = s->kind == LASSO_LT;
  const size_t copy_elems = lt && !lt_scaled && S.c > 1 ? (size_t)(S.c - 1) * n : 0;
  rc = ensure_scratch(c, ((size_t)nx * K + copy_elems) * sizeof(fr_t)); if (rc) return rc;
  rc = ensure_small(c, K); if (rc) return rc;
  if (copy_elems) {   // literal form: scaled copies of LT_0 .. LT_{C-2} behind the partials (LT_{C-1} has kappa = 1)
    fr_t* cp = (fr_t*)c->d_scratch + (size_t)nx * K; MutPtrTable M; PtrTable Src; LtKappa LK;
    for (uint32_t m = 0; m + 1 < S.c; m++) { Src.p[2 * m] = (const fr_t*)d_polys[2 * m]; M.p[2 * m] = cp + (size_t)m * n; P.p[2 * m] = cp + (size_t)m * n; LK.k[m] = lt_pow32(S.c - 1 - m, true); }
    hipLaunchKernelGGL(k_lt_prescale, dim3(grid_for(n, 1024), S.c - 1), dim3(LASSO_BLOCK), 0, c->stream, Src, M, LK, n);
  }
  {
    ProfScope ps(c, LASSO_K_COMBINE, 32.0 * n * (S.alpha + 1.0));
    if (spark) {   // g = prod_m E_m: the LT walk without its LT terms (k_combine_round_lt<.., PROD>), same 32^C correction of the block sums
      const fr_t scale = lt_pow32(S.c, false);
#define LAUNCH_COMBINE_PROD(A_, D_, T_) hipLaunchKernelGGL((k_combine_round_lt<A_, D_, T_, true>), dim3(nx), dim3(LASSO_BLOCK), 0, c->stream, S, P, (const fr_t*)d_eq, scale, half, degree, (fr_t*)c->d_scratch)
      DISPATCH_LT(2 * S.c, LAUNCH_COMBINE_PROD);   // the dispatch table is keyed by LT's memory count 2C: same degree bounds
    } else if (!lt) hipLaunchKernelGGL(k_combine_round_linear, dim3(nx), dim3(LASSO_BLOCK), 0, c->stream, S, P, (const fr_t*)d_eq, W, half, (fr_t*)c->d_scratch);
    else {
      const fr_t scale = lt_pow32(S.c, false);
#define LAUNCH_COMBINE(A_, D_, T_) hipLaunchKernelGGL((k_combine_round_lt<A_, D_, T_>), dim3(nx), dim3(LASSO_BLOCK), 0, c->stream, S, P, (const fr_t*)d_eq, scale, half, degree, (fr_t*)c->d_scratch)
      DISPATCH_LT(S.alpha, LAUNCH_COMBINE);
    }
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)c->d_scratch, nx, K, c->d_small);
  }
  HIPCHK(c, hipGetLastError());
  return fetch_small(c, K, out);
}
int32_t lasso_sumcheck_combine_round(lasso_ctx* c, const lasso_strategy* s, const lasso_fr* const* d_polys, const lasso_fr* d_eq, size_t n, uint32_t degree, lasso_fr* out) {
  return combine_round_impl(c, s, d_polys, d_eq, n, degree, out, false);
}
int32_t lasso_sumcheck_combine_round_lt_scaled(lasso_ctx* c, const lasso_strategy* s, const lasso_fr* const* d_polys, const lasso_fr* d_eq, size_t n, uint32_t degree, lasso_fr* out) {
  REQUIRE(c, s && s->kind == LASSO_LT);
  return combine_round_impl(c, s, d_polys, d_eq, n, degree, out, true);
}
// the FIRST round of the LT sumcheck from the polynomials' integer values (entries 0 / 1: the LT and EQ subtables hold bits): exact integer Horner walk, one field product per point for the eq weight
int32_t lasso_sumcheck_combine_round_lt_u32(lasso_ctx* c, const lasso_strategy* s, const uint32_t* const* d_u32, const lasso_fr* d_eq, size_t n, uint32_t degree, lasso_fr* out) {
  StrategyDev S; WeightTable W; int32_t rc = make_strategy(c, s, S, W); if (rc) return rc;
  REQUIRE(c, s->kind == LASSO_LT && d_u32 && d_eq && out && n >= 2 && (n & (n - 1)) == 0 && degree == s->c + 1 && s->c <= 16 && !c->defer_next);   // C <= 16: |t| < 2^67 fits the three-limb form (header)
  PtrTableU32 P; for (uint32_t i = 0; i < S.alpha; i++) { REQUIRE(c, d_u32[i]); P.p[i] = d_u32[i]; }
  const size_t half = n / 2; const unsigned nx = grid_for(half, 1024); const uint32_t K = degree + 1;
  rc = ensure_scratch(c, (size_t)nx * K * sizeof(fr_t)); if (rc) return rc;
  rc = ensure_small(c, K); if (rc) return rc;
  {
    ProfScope ps(c, LASSO_K_COMBINE, 32.0 * n * (S.alpha + 1.0));
#define LAUNCH_COMBINE_U32(A_, D_, T_) hipLaunchKernelGGL((k_combine_round_lt_u32<A_, D_, T_>), dim3(nx), dim3(LASSO_BLOCK), 0, c->stream, S, P, (const fr_t*)d_eq, half, degree, (fr_t*)c->d_scratch, c->d_flag + 16)
    __atomic_store_n(c->h_flag + 16, 0u, __ATOMIC_RELEASE);   // "an entry was not 0 / 1": a host-mapped word the kernel sets, read behind the result's flag
    DISPATCH_LT(S.alpha, LAUNCH_COMBINE_U32);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)c->d_scratch, nx, K, c->d_small);
  }
  HIPCHK(c, hipGetLastError());
  rc = fetch_small(c, K, out); if (rc) return rc;
  if (__atomic_load_n(c->h_flag + 16, __ATOMIC_ACQUIRE) != 0) return fail(c, LASSO_ERR_INVALID, "lasso_sumcheck_combine_round_lt_u32: an entry is neither 0 nor 1 (the integer round is exact only for the LT / EQ subtables' bits)");
  return 0;
}
int32_t lasso_lt_prescale(lasso_ctx* c, const lasso_strategy* s, const lasso_fr* const* d_src, lasso_fr* const* d_polys, size_t n) {
  StrategyDev S; WeightTable W; int32_t rc = make_strategy(c, s, S, W); if (rc) return rc;
  REQUIRE(c, s->kind == LASSO_LT && d_polys && n >= 1);
  MutPtrTable M; PtrTable Src; LtKappa LK;
  for (uint32_t m = 0; m + 1 < S.c; m++) { REQUIRE(c, d_polys[2 * m] && (!d_src || d_src[2 * m])); M.p[2 * m] = (fr_t*)d_polys[2 * m]; Src.p[2 * m] = d_src ? (const fr_t*)d_src[2 * m] : (const fr_t*)d_polys[2 * m]; LK.k[m] = lt_pow32(S.c - 1 - m, true); }
  ProfScope ps(c, LASSO_K_MISC, 64.0 * n * (S.c - 1) + (d_src ? 64.0 * n * (S.c + 1) : 0.0));
  if (S.c >= 2) hipLaunchKernelGGL(k_lt_prescale, dim3(grid_for(n, 1024), S.c - 1), dim3(LASSO_BLOCK), 0, c->stream, Src, M, LK, n);
  if (d_src) {   // out of place: the polynomials the scaling leaves alone (LT_{C-1} and every EQ_m) are plain copies — together the clone of sumcheck.rs / surge.rs:151
    for (uint32_t i = 0; i < S.alpha; i++) if ((i & 1u) || i == 2 * (S.c - 1)) { REQUIRE(c, d_src[i] && d_polys[i]); HIPCHK(c, hipMemcpyAsync(d_polys[i], d_src[i], n * sizeof(fr_t), hipMemcpyDeviceToDevice, c->stream)); }
  }
  HIPCHK(c, hipGetLastError()); return 0;
}
int32_t lasso_combine_claim(lasso_ctx* c, const lasso_strategy* s, const lasso_fr* const* d_polys, const lasso_fr* d_eq, size_t n, lasso_fr* out) {
  StrategyDev S; WeightTable W; int32_t rc = make_strategy(c, s, S, W); if (rc) return rc;
  REQUIRE(c, d_polys && d_eq && out && n >= 1);
  PtrTable P; for (uint32_t i = 0; i < S.alpha; i++) { REQUIRE(c, d_polys[i]); P.p[i] = (const fr_t*)d_polys[i]; }
  const unsigned nx = grid_for(n, 1024);
  rc = ensure_scratch(c, (size_t)nx * sizeof(fr_t)); if (rc) return rc;
  {
    ProfScope ps(c, LASSO_K_COMBINE, 32.0 * n * (S.alpha + 1.0));
#define LAUNCH_CLAIM(A_, D_) hipLaunchKernelGGL((k_combine_claim<A_>), dim3(nx), dim3(LASSO_BLOCK), 0, c->stream, S, P, (const fr_t*)d_eq, W, n, (fr_t*)c->d_scratch)
    DISPATCH_A(S.alpha, LAUNCH_CLAIM);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)c->d_scratch, nx, 1u, c->d_small);
  }
  HIPCHK(c, hipGetLastError());
  return fetch_small(c, 1, out);
}
int32_t lasso_multi_dot(lasso_ctx* c, const lasso_fr* const* d_polys, uint32_t k, const lasso_fr* d_w, size_t n, lasso_fr* out) {
  REQUIRE(c, d_polys && d_w && out && k >= 1 && k <= LASSO_MAX_PTRS && n >= 1);
  PtrTable P; for (uint32_t i = 0; i < k; i++) { REQUIRE(c, d_polys[i]); P.p[i] = (const fr_t*)d_polys[i]; }
  const unsigned nx = grid_for(n, 512);
  int32_t rc = ensure_scratch(c, (size_t)nx * k * sizeof(fr_t)); if (rc) return rc;
  rc = ensure_small(c, k); if (rc) return rc;
  {
    ProfScope ps(c, LASSO_K_DOT, 32.0 * n * (k + 1.0));
    hipLaunchKernelGGL(k_multi_dot, dim3(nx * k), dim3(LASSO_BLOCK), 0, c->stream, P, nx, k, (const fr_t*)d_w, n, (fr_t*)c->d_scratch);
    hipLaunchKernelGGL(k_reduce_partials, dim3(k), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)c->d_scratch, nx, 1u, c->d_small);
  }
  HIPCHK(c, hipGetLastError());
  return fetch_small(c, k, out);
}
int32_t lasso_read_heads(lasso_ctx* c, const lasso_fr* const* d_polys, uint32_t k, lasso_fr* out) {
  REQUIRE(c, d_polys && out && k >= 1 && k <= LASSO_MAX_PTRS);
  PtrTable P; for (uint32_t i = 0; i < k; i++) { REQUIRE(c, d_polys[i]); P.p[i] = (const fr_t*)d_polys[i]; }
  int32_t rc = ensure_small(c, k); if (rc) return rc;
  hipLaunchKernelGGL(k_read_heads, dim3(1), dim3(LASSO_BLOCK), 0, c->stream, P, k, c->d_small);
  HIPCHK(c, hipGetLastError());
  return fetch_small(c, k, out);
}
// out[i * count + j] = d_polys[i][j], j < count (k * count <= 16384 elements), through the mapped result buffer: no memcpy, no stream synchronisation
int32_t lasso_read_runs(lasso_ctx* c, const lasso_fr* const* d_polys, uint32_t k, uint32_t count, lasso_fr* out) {
  REQUIRE(c, d_polys && out && k >= 1 && k <= LASSO_MAX_PTRS && count >= 1 && (size_t)k * count <= 16384);
  PtrTable P; for (uint32_t i = 0; i < k; i++) { REQUIRE(c, d_polys[i]); P.p[i] = (const fr_t*)d_polys[i]; }
  int32_t rc = ensure_small(c, (size_t)k * count); if (rc) return rc;
  hipLaunchKernelGGL(k_read_runs, dim3((k * count + LASSO_BLOCK - 1) / LASSO_BLOCK), dim3(LASSO_BLOCK), 0, c->stream, P, k, count, c->d_small);
  HIPCHK(c, hipGetLastError());
  return fetch_small(c, (size_t)k * count, out);
}
// layers above `in` (len elements, the layers laid out back to back behind it): one launch per large layer, the small ones in one workgroup
static void gp_layers_from(lasso_ctx* c, fr_t* in, size_t len) {
  while (len >= 16 * LASSO_BLOCK) {   // two layers per launch while they are large: the layer in between is written but not read back
    const size_t q = len / 4;
    hipLaunchKernelGGL(k_gp_layer2, dim3(grid_for(q, 4096)), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)in, q, in + len, in + len + len / 2);
    in += len + len / 2; len = q;
  }
  while (len > 2 * LASSO_BLOCK) {
    size_t half = len / 2;
    hipLaunchKernelGGL(k_gp_layer, dim3(grid_for(half, 4096)), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)in, half, in + len);
    in += len; len = half;
  }
  if (len > 2) hipLaunchKernelGGL(k_gp_tail, dim3(1), dim3(LASSO_BLOCK), 0, c->stream, in, len);
}
int32_t lasso_gp_build(lasso_ctx* c, lasso_fr* d_tree, size_t n) {
  REQUIRE(c, d_tree && n >= 2 && (n & (n - 1)) == 0);
  ProfScope ps(c, LASSO_K_GP, 48.0 * n * 2.0);
  gp_layers_from(c, (fr_t*)d_tree, n);
  HIPCHK(c, hipGetLastError()); return 0;
}
// lasso_fingerprint_ops + lasso_gp_build of both trees in one call, with the first product layer taken while the leaves are still in registers
// (k_fingerprint_ops_l1): d_tree_r / d_tree_w are 2s-element arenas, leaves first.  Same bytes as the three separate calls.
int32_t lasso_fingerprint_ops_gp(lasso_ctx* c, const lasso_fr* d_table, const uint32_t* d_dim, const lasso_fr* d_read, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                 lasso_fr* d_tree_r, lasso_fr* d_tree_w) {
  REQUIRE(c, d_table && d_dim && d_read && gamma && tau && d_tree_r && d_tree_w && s >= 4 && (s & (s - 1)) == 0);
  fr_t g = to_fr(gamma), g2 = fr_sqr(g), t = to_fr(tau);
  fr_t* tr = (fr_t*)d_tree_r; fr_t* tw = (fr_t*)d_tree_w;
  {
    ProfScope ps(c, LASSO_K_FINGERPRINT, (32.0 * 3 + 64.0) * s + 2 * 48.0 * s);   // the fingerprints + the first layer of two trees (48 n per layer)
    // LASSO_EXP_NO_LEAF_STORE=1: TIMING EXPERIMENT ONLY (the proof that follows is invalid): the kernel without its 2 x 32 s bytes of leaf stores = what "compact leaves" would leave of it
    static const uint32_t store_leaves = [] { const char* v = getenv("LASSO_EXP_NO_LEAF_STORE"); return (v && v[0] == '1') ? 0u : 1u; }();
    hipLaunchKernelGGL(k_fingerprint_ops_l1<false>, dim3(grid_for(s / 2, 4096)), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)d_table, d_dim, (const void*)d_read, s, g, g2, t, tr, tw, tr + s, tw + s, store_leaves);
  }
  {
    ProfScope ps(c, LASSO_K_GP, 2 * 48.0 * s);   // the remaining layers of both trees
    gp_layers_from(c, tr + s, s / 2);
    gp_layers_from(c, tw + s, s / 2);
  }
  HIPCHK(c, hipGetLastError()); return 0;
}
// Capacity mode: the two trees WITHOUT their leaf layers.  d_upper_r / d_upper_w: s - 2 (allocate s) elements each = the layers of s/2, s/4, .., 2 elements back to back, i.e.
// what lasso_fingerprint_ops_gp leaves at d_tree + s.  The leaves exist only inside the launch (k_fingerprint_ops_l1 without its leaf stores).
static int32_t fingerprint_gp_upper_impl(lasso_ctx* c, const lasso_fr* d_table, const uint32_t* d_dim, const void* d_read, bool read_u32, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                         lasso_fr* d_upper_r, lasso_fr* d_upper_w) {
  REQUIRE(c, d_table && d_dim && d_read && gamma && tau && d_upper_r && d_upper_w && s >= 4 && (s & (s - 1)) == 0);
  fr_t g = to_fr(gamma), g2 = fr_sqr(g), t = to_fr(tau);
  fr_t* ur = (fr_t*)d_upper_r; fr_t* uw = (fr_t*)d_upper_w;
  {
    ProfScope ps(c, LASSO_K_FINGERPRINT, (32.0 * 3 + 64.0) * s + 2 * 48.0 * s);   // the reference's bytes for the same step; the leaf stores are not made
    if (read_u32) hipLaunchKernelGGL(k_fingerprint_ops_l1<true>, dim3(grid_for(s / 2, 4096)), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)d_table, d_dim, d_read, s, g, g2, t, (fr_t*)nullptr, (fr_t*)nullptr, ur, uw, 0u);
    else hipLaunchKernelGGL(k_fingerprint_ops_l1<false>, dim3(grid_for(s / 2, 4096)), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)d_table, d_dim, d_read, s, g, g2, t, (fr_t*)nullptr, (fr_t*)nullptr, ur, uw, 0u);
  }
  {
    ProfScope ps(c, LASSO_K_GP, 2 * 48.0 * s);
    gp_layers_from(c, ur, s / 2);
    gp_layers_from(c, uw, s / 2);
  }
  HIPCHK(c, hipGetLastError()); return 0;
}
// Capacity mode: the two trees WITHOUT their leaf layers.  d_upper_r / d_upper_w: s - 2 (allocate s) elements each = the layers of s/2, s/4, .., 2 elements back to back, i.e.
// what lasso_fingerprint_ops_gp leaves at d_tree + s.  The leaves exist only inside the launch (k_fingerprint_ops_l1 without its leaf stores).
int32_t lasso_fingerprint_ops_gp_upper(lasso_ctx* c, const lasso_fr* d_table, const uint32_t* d_dim, const lasso_fr* d_read, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                       lasso_fr* d_upper_r, lasso_fr* d_upper_w) { return fingerprint_gp_upper_impl(c, d_table, d_dim, d_read, false, s, gamma, tau, d_upper_r, d_upper_w); }
// the same with the read timestamps as 32-bit integers (capacity mode keeps dim / read compact: 4 bytes per entry instead of 32)
int32_t lasso_fingerprint_ops_gp_upper_u32(lasso_ctx* c, const lasso_fr* d_table, const uint32_t* d_dim, const uint32_t* d_read_u32, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                           lasso_fr* d_upper_r, lasso_fr* d_upper_w) { return fingerprint_gp_upper_impl(c, d_table, d_dim, d_read_u32, true, s, gamma, tau, d_upper_r, d_upper_w); }
static int32_t fingerprint_strips_impl(lasso_ctx* c, const lasso_fr* d_table, const uint32_t* d_dim, const void* d_read, bool read_u32, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                       uint32_t nstrips, size_t i0, size_t cs, lasso_fr* d_out_r, lasso_fr* d_out_w) {
  REQUIRE(c, d_table && d_dim && d_read && gamma && tau && d_out_r && d_out_w && s >= 4 && (s & (s - 1)) == 0 && (nstrips == 2 || nstrips == 4) && cs >= 1);
  const size_t stride = s / 2 / nstrips;
  REQUIRE(c, stride >= 1 && i0 + cs <= stride);
  fr_t g = to_fr(gamma), g2 = fr_sqr(g), t = to_fr(tau);
  const size_t total = 2 * (size_t)nstrips * cs;
  ProfScope ps(c, LASSO_K_FINGERPRINT, (32.0 * 3 + 64.0) * total);
  if (read_u32) hipLaunchKernelGGL(k_fingerprint_ops_strips<true>, dim3(grid_for(total, 4096)), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)d_table, d_dim, d_read, s, g, g2, t, nstrips, stride, i0, cs, (fr_t*)d_out_r, (fr_t*)d_out_w);
  else hipLaunchKernelGGL(k_fingerprint_ops_strips<false>, dim3(grid_for(total, 4096)), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)d_table, d_dim, d_read, s, g, g2, t, nstrips, stride, i0, cs, (fr_t*)d_out_r, (fr_t*)d_out_w);
  HIPCHK(c, hipGetLastError()); return 0;
}
// ... and the leaves of one strip set of the bottom layer, recomputed for a round on the index range [i0, i0 + cs) (k_fingerprint_ops_strips states the layout)
int32_t lasso_fingerprint_ops_strips(lasso_ctx* c, const lasso_fr* d_table, const uint32_t* d_dim, const lasso_fr* d_read, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                     uint32_t nstrips, size_t i0, size_t cs, lasso_fr* d_out_r, lasso_fr* d_out_w) { return fingerprint_strips_impl(c, d_table, d_dim, d_read, false, s, gamma, tau, nstrips, i0, cs, d_out_r, d_out_w); }
int32_t lasso_fingerprint_ops_strips_u32(lasso_ctx* c, const lasso_fr* d_table, const uint32_t* d_dim, const uint32_t* d_read_u32, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                                         uint32_t nstrips, size_t i0, size_t cs, lasso_fr* d_out_r, lasso_fr* d_out_w) { return fingerprint_strips_impl(c, d_table, d_dim, d_read_u32, true, s, gamma, tau, nstrips, i0, cs, d_out_r, d_out_w); }
int32_t lasso_fingerprint_ops(lasso_ctx* c, const lasso_fr* d_table, const uint32_t* d_dim, const lasso_fr* d_read, size_t s, const lasso_fr* gamma, const lasso_fr* tau,
                              lasso_fr* d_read_out, lasso_fr* d_write_out) {
  REQUIRE(c, d_table && d_dim && d_read && gamma && tau && d_read_out && d_write_out); if (!s) return 0;
  fr_t g = to_fr(gamma), g2 = fr_sqr(g), t = to_fr(tau);
  ProfScope ps(c, LASSO_K_FINGERPRINT, (32.0 * 3 + 64.0) * s);
  hipLaunchKernelGGL(k_fingerprint_ops, dim3(grid_for(s, 4096)), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)d_table, d_dim, (const fr_t*)d_read, s, g, g2, t, (fr_t*)d_read_out, (fr_t*)d_write_out);
  HIPCHK(c, hipGetLastError()); return 0;
}
int32_t lasso_fingerprint_mem(lasso_ctx* c, const lasso_fr* d_table, const lasso_fr* d_final, size_t m, const lasso_fr* gamma, const lasso_fr* tau, lasso_fr* d_init_out, lasso_fr* d_final_out) {
  return lasso_fingerprint_mem_slab(c, d_table, d_final, m, 1, 0, gamma, tau, d_init_out, d_final_out);
}
int32_t lasso_fingerprint_mem_slab(lasso_ctx* c, const lasso_fr* d_table, const lasso_fr* d_final, size_t m, uint32_t world, uint32_t rank, const lasso_fr* gamma, const lasso_fr* tau,
                                   lasso_fr* d_init_out, lasso_fr* d_final_out) {
  REQUIRE(c, d_table && d_final && gamma && tau && d_init_out && d_final_out && world >= 1 && rank < world); if (!m) return 0;
  fr_t g = to_fr(gamma), g2 = fr_sqr(g), t = to_fr(tau);
  ProfScope ps(c, LASSO_K_FINGERPRINT, (64.0 + 64.0) * m);
  hipLaunchKernelGGL(k_fingerprint_mem, dim3(grid_for(m, 4096)), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)d_table, (const fr_t*)d_final, m, world, rank, g, g2, t, (fr_t*)d_init_out, (fr_t*)d_final_out);
  HIPCHK(c, hipGetLastError()); return 0;
}
int32_t lasso_matvec_left(lasso_ctx* c, const lasso_fr* d_Z, const lasso_fr* L, size_t l_size, size_t r_size, lasso_fr* out) {
  REQUIRE(c, d_Z && L && out && l_size >= 1 && r_size >= 1);
  // enough row chunks to fill the chip: ~1024 workgroups
  size_t col_blocks = (r_size + LASSO_BLOCK - 1) / LASSO_BLOCK;
  size_t nchunks = (1024 + col_blocks - 1) / col_blocks; if (nchunks > l_size) nchunks = l_size; if (nchunks < 1) nchunks = 1;
  size_t rows_per_chunk = (l_size + nchunks - 1) / nchunks; nchunks = (l_size + rows_per_chunk - 1) / rows_per_chunk;
  int32_t rc = ensure_scratch(c, (l_size + nchunks * r_size) * sizeof(fr_t)); if (rc) return rc;
  rc = ensure_big(c, r_size); if (rc) return rc;
  fr_t* dL = (fr_t*)c->d_scratch; fr_t* partials = dL + l_size;
  HIPCHK(c, hipMemcpyAsync(dL, L, l_size * sizeof(fr_t), hipMemcpyHostToDevice, c->stream));
  {
    ProfScope ps(c, LASSO_K_MATVEC, 32.0 * l_size * r_size);
    hipLaunchKernelGGL(k_matvec_left, dim3((unsigned)col_blocks, (unsigned)nchunks), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)d_Z, (const fr_t*)dL, l_size, r_size, rows_per_chunk, partials);
    hipLaunchKernelGGL(k_matvec_reduce, dim3((unsigned)col_blocks), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)partials, nchunks, r_size, c->d_big);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_big, c->d_big, r_size * sizeof(fr_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  memcpy(out, c->h_big, r_size * sizeof(fr_t));
  return 0;
}

// device-resident forms used by the opening: L already on the device, L*Z left on the device (no host round trip)
int32_t lasso_matvec_left_dev(lasso_ctx* c, const lasso_fr* d_Z, const lasso_fr* d_L, size_t l_size, size_t r_size, lasso_fr* d_out) {
  REQUIRE(c, d_Z && d_L && d_out && l_size >= 1 && r_size >= 1);
  size_t col_blocks = (r_size + LASSO_BLOCK - 1) / LASSO_BLOCK;
  size_t nchunks = (1024 + col_blocks - 1) / col_blocks; if (nchunks > l_size) nchunks = l_size; if (nchunks < 1) nchunks = 1;
  size_t rows_per_chunk = (l_size + nchunks - 1) / nchunks; nchunks = (l_size + rows_per_chunk - 1) / rows_per_chunk;
  int32_t rc = ensure_scratch(c, nchunks * r_size * sizeof(fr_t)); if (rc) return rc;
  fr_t* partials = (fr_t*)c->d_scratch;
  ProfScope ps(c, LASSO_K_MATVEC, 32.0 * l_size * r_size);
  hipLaunchKernelGGL(k_matvec_left, dim3((unsigned)col_blocks, (unsigned)nchunks), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)d_Z, (const fr_t*)d_L, l_size, r_size, rows_per_chunk, partials);
  hipLaunchKernelGGL(k_matvec_reduce, dim3((unsigned)col_blocks), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)partials, nchunks, r_size, (fr_t*)d_out);
  HIPCHK(c, hipGetLastError()); return 0;
}
// ark-serialize of n field elements (canonical integers, 32 little-endian bytes each) — what append_scalar feeds the transcript (utils/transcript.rs:33-45)
// d_dst[i] = the canonical value of d_src[i] as a 32-bit integer; LASSO_ERR_INVALID if some value does not fit; *max_out (optional) = the largest one.  How capacity mode
// turns the timestamps densify wrote as field elements into its compact form (DensePolynomial::from_usize's inverse, dense_mlpoly.rs:263-269).
int32_t lasso_fr_to_u32(lasso_ctx* c, const lasso_fr* d_src, size_t n, uint32_t* d_dst, uint32_t* max_out) {
  REQUIRE(c, d_src && d_dst && n >= 1);
  HIPCHK(c, hipMemsetAsync(c->d_flags, 0, 8, c->stream));
  hipLaunchKernelGGL(k_fr_to_u32, dim3(grid_for(n, 4096)), dim3(256), 0, c->stream, (const fr_t*)d_src, n, d_dst, c->d_flags);
  HIPCHK(c, hipGetLastError());
  uint32_t flags[2];
  HIPCHK(c, hipMemcpyAsync(flags, c->d_flags, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (flags[1]) return fail(c, LASSO_ERR_INVALID, "lasso_fr_to_u32: a value does not fit 32 bits");
  if (max_out) *max_out = flags[0];
  return 0;
}
int32_t lasso_fr_to_bytes(lasso_ctx* c, const lasso_fr* d_src, size_t n, uint8_t* out) {
  REQUIRE(c, d_src && out && n >= 1);
  // up to 2^16 elements (the a-vector of an opening: 128-256 KiB) go straight into the host-mapped result buffer and come back behind the
  // sequence flag: a hipMemcpy to pageable memory plus a stream synchronisation cost 100 us of idle device per opening
  if (n <= ((size_t)1 << 16)) {
    int32_t rc = ensure_small(c, n); if (rc) return rc;
    hipLaunchKernelGGL(k_fr_to_canonical, dim3(grid_for(n)), dim3(256), 0, c->stream, (const fr_t*)d_src, n, c->d_small);
    HIPCHK(c, hipGetLastError());
    return fetch_small(c, n, (lasso_fr*)out);
  }
  int32_t rc = ensure_scratch(c, n * sizeof(fr_t)); if (rc) return rc;
  hipLaunchKernelGGL(k_fr_to_canonical, dim3(grid_for(n)), dim3(256), 0, c->stream, (const fr_t*)d_src, n, (fr_t*)c->d_scratch);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(out, c->d_scratch, n * 32, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ------------------------------------------------------------------ densify (densified.rs:22-75)
int32_t lasso_densify_dim(lasso_ctx* c, const uint64_t* d_indices, size_t n_lookups, size_t C, size_t dim, size_t s, uint32_t log_m, uint32_t* d_dim_u32, lasso_fr* d_dim, lasso_fr* d_read,
                          lasso_fr* d_final) {
  return lasso_densify_dim_slab(c, d_indices, n_lookups, C, dim, s, log_m, 1, 0, d_dim_u32, d_dim, d_read, d_final);
}
int32_t lasso_densify_dim_slab(lasso_ctx* c, const uint64_t* d_indices, size_t n_lookups, size_t C, size_t dim, size_t s, uint32_t log_m, uint32_t world, uint32_t rank, uint32_t* d_dim_u32,
                               lasso_fr* d_dim, lasso_fr* d_read, lasso_fr* d_final) {
  REQUIRE(c, d_indices && d_dim_u32 && d_dim && d_read && d_final && C >= 1 && dim < C && s >= 1 && (s & (s - 1)) == 0 && n_lookups <= s && s < ((size_t)1 << 32) && log_m <= 32);
  REQUIRE(c, world >= 1 && (world & (world - 1)) == 0 && rank < world && world <= s && world <= ((size_t)1 << log_m));
  const size_t m = (size_t)1 << log_m;
  const uint32_t ntiles = (uint32_t)((s + RADIX_TILE - 1) / RADIX_TILE);
  const size_t nh = (size_t)256 * ntiles, nb = (nh + 4095) / 4096;
  const size_t words = 4 * s + nh + nb + 2 * m + 64;
  int32_t rc = ensure_scratch(c, words * 4); if (rc) return rc;
  uint32_t* kA = (uint32_t*)c->d_scratch; uint32_t* vA = kA + s; uint32_t* kB = vA + s; uint32_t* vB = kB + s;
  uint32_t* hist = vB + s; uint32_t* sums = hist + nh; uint32_t* run_start = sums + nb; uint32_t* run_end = run_start + m;
  HIPCHK(c, hipMemsetAsync(c->d_flags, 0, 8, c->stream));
  HIPCHK(c, hipMemsetAsync(run_start, 0, 2 * m * 4, c->stream));
  ProfScope ps(c, LASSO_K_MISC, 8.0 * n_lookups + (32.0 * 2 + 4.0) * s + 32.0 * m);
  hipLaunchKernelGGL(k_densify_extract, dim3(grid_for(s, 4096)), dim3(256), 0, c->stream, d_indices, n_lookups, C, dim, s, (uint64_t)m, world, rank, kA, vA, d_dim_u32, (fr_t*)d_dim, c->d_flags);
  const uint32_t npass = log_m == 0 ? 1 : (log_m + 7) / 8;
  for (uint32_t p = 0; p < npass; p++) {
    hipLaunchKernelGGL(k_radix_hist, dim3(ntiles), dim3(RADIX_THREADS), 0, c->stream, (const uint32_t*)kA, s, 8 * p, hist, ntiles);
    hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(256), 0, c->stream, (const uint32_t*)hist, nh, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, c->stream, sums, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, c->stream, hist, nh, (const uint32_t*)sums);
    hipLaunchKernelGGL(k_radix_scatter, dim3(ntiles), dim3(RADIX_THREADS), 0, c->stream, (const uint32_t*)kA, (const uint32_t*)vA, s, 8 * p, (const uint32_t*)hist, ntiles, kB, vB);
    std::swap(kA, kB); std::swap(vA, vB);
  }
  hipLaunchKernelGGL(k_densify_runs, dim3(grid_for(s, 4096)), dim3(256), 0, c->stream, (const uint32_t*)kA, s, run_start, run_end);
  hipLaunchKernelGGL(k_densify_read, dim3(grid_for(s, 4096)), dim3(256), 0, c->stream, (const uint32_t*)kA, (const uint32_t*)vA, s, (const uint32_t*)run_start, world, rank, (fr_t*)d_read);
  hipLaunchKernelGGL(k_densify_final, dim3(grid_for(m / world, 4096)), dim3(256), 0, c->stream, (const uint32_t*)run_start, (const uint32_t*)run_end, m, world, rank, (fr_t*)d_final);
  HIPCHK(c, hipGetLastError());
  uint32_t flags[2];
  HIPCHK(c, hipMemcpyAsync(flags, c->d_flags, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (flags[0]) return fail(c, LASSO_ERR_INVALID, "lookup index out of range (memory_address >= M, densified.rs:46)");
  return 0;
}

// ------------------------------------------------------------------ curve entry points
int32_t lasso_bases_create(lasso_ctx* c, const lasso_affine* points, size_t n, lasso_bases** out) { return lasso_bases_create_opt(c, points, n, 1, out); }
int32_t lasso_bases_create_opt(lasso_ctx* c, const lasso_affine* points, size_t n, int32_t byte_multiples, lasso_bases** out) {
  REQUIRE(c, points && out && n >= 1 && n * MSM_WINDOWS < ((size_t)1 << 32));
  lasso_bases* b = new lasso_bases(); b->n = n; b->owner = c;
  void* d_aff = nullptr;
  if (dmalloc(c, &d_aff, n * sizeof(lasso_affine)) != hipSuccess || dmalloc(c, (void**)&b->d_table, n * MSM_WINDOWS * sizeof(niels29)) != hipSuccess) {
    if (d_aff) (void)dfree(c, d_aff); delete b; return fail(c, LASSO_ERR_OOM, "bases alloc");
  }
  hipError_t e = hipMemcpyAsync(d_aff, points, n * sizeof(lasso_affine), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) { hipLaunchKernelGGL(k_precompute_table, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, (const fq_t*)d_aff, n, b->d_table); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)dfree(c, d_aff);
  if (e != hipSuccess) { (void)dfree(c, b->d_table); delete b; return fail(c, LASSO_ERR_HIP, hipGetErrorString(e)); }
  // digit multiples for the few-row full-width MSMs of the opening tail: 8 x the window table (57 KB per generator).  Optional: if the
  // allocation fails or the generator set is larger than LASSO_MSM_DIRECT_MAX_N the bucket kernel serves those MSMs too.
  static const size_t direct_max = [] { const char* v = getenv("LASSO_MSM_DIRECT_MAX_N"); return v ? (size_t)atoll(v) : (((size_t)1 << 17) + 64); }();
  if (n <= direct_max) {
    if (dmalloc(c, (void**)&b->d_mult, n * MSM_WINDOWS * MSM_MULTS * sizeof(niels29)) == hipSuccess) {
      hipLaunchKernelGGL(k_precompute_multiples, dim3((unsigned)((n * MSM_WINDOWS + 63) / 64)), dim3(64), 0, c->stream, (const niels29*)b->d_table, n, b->d_mult);
      e = hipGetLastError(); if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) { (void)dfree(c, b->d_mult); (void)dfree(c, b->d_table); delete b; return fail(c, LASSO_ERR_HIP, hipGetErrorString(e)); }
    } else { (void)hipGetLastError(); b->d_mult = nullptr; }
  }
  // ... and their byte multiples (8x the bytes of d_mult: 459 KB per generator, 3.8 GB for the 8194 generators of the headline's widest opening; HBM is 288 GB and the
  // generators are fixed for the life of the object).  LASSO_MSM_DIRECT8=0 turns them off, LASSO_MSM_DIRECT8_MAX_N moves the size limit; a failed allocation is not an error.
  static const size_t direct8_max = [] { const char* off = getenv("LASSO_MSM_DIRECT8"); if (off && off[0] == '0') return (size_t)0; const char* v = getenv("LASSO_MSM_DIRECT8_MAX_N"); return v ? (size_t)atoll(v) : (((size_t)1 << 14) + 64); }();
  if (b->d_mult && byte_multiples && n <= direct8_max) {
    typedef MsmD<8> D8;
    if (dmalloc(c, (void**)&b->d_mult8, n * D8::WINDOWS * D8::MULTS * sizeof(niels29)) == hipSuccess) {
      hipLaunchKernelGGL(k_precompute_tab8, dim3((unsigned)((n + 63) / 64), D8::WINDOWS), dim3(64), 0, c->stream, (const niels29*)b->d_table, n, 0u, b->d_mult8, D8::MULTS);   // all 32 windows in one launch
      e = hipGetLastError(); if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) { (void)dfree(c, b->d_mult8); (void)dfree(c, b->d_mult); (void)dfree(c, b->d_table); delete b; return fail(c, LASSO_ERR_HIP, hipGetErrorString(e)); }
    } else { (void)hipGetLastError(); b->d_mult8 = nullptr; }
  }
  *out = b; return 0;
}
void lasso_bases_destroy(lasso_ctx* c, lasso_bases* b) {
  if (!b) return; if (c) (void)hipStreamSynchronize(c->stream);
  lasso_ctx* o = live_or_null(b->owner);   // the tables' bytes are accounted to the context that built them; a caller that passes another context (or NULL) only loses the bookkeeping
  if (b->d_table) (void)dfree(o, b->d_table); if (b->d_mult) (void)dfree(o, b->d_mult); if (b->d_mult8) (void)dfree(o, b->d_mult8);
  for (niels29* t : b->d_tab8) if (t) (void)dfree(o, t);
  delete b;
}
// the byte-multiple table of window w8 (k_precompute_tab8), built the first time a commitment asks for it: 255 * n * 112 bytes (117 MB for n = 4096).
// LASSO_MSM_ROWS8=0 switches the path off (A/B measurements); an allocation failure falls back to the bucket kernel for the life of the bases object.
static bool msm_rows8_enabled() { static const bool on = [] { const char* v = getenv("LASSO_MSM_ROWS8"); return !(v && v[0] == '0'); }(); return on; }
static const niels29* ensure_tab8(lasso_ctx* c, const lasso_bases* cb, uint32_t w8) {
  lasso_bases* b = const_cast<lasso_bases*>(cb);
  std::lock_guard<std::mutex> lock(b->tab8_mu);   // one builder at a time; the pointer is published only after the build has been waited for (below)
  if (b->d_tab8[w8]) return b->d_tab8[w8];
  if (b->tab8_failed) return nullptr;
  niels29* t = nullptr;
  lasso_ctx* const owner = live_or_null(b->owner);   // accounted to the creating context while it lives, to nobody afterwards
  if (dmalloc(owner, (void**)&t, (size_t)MSM8_MULTS * b->n * sizeof(niels29)) != hipSuccess) { (void)hipGetLastError(); b->tab8_failed = true; return nullptr; }
  hipLaunchKernelGGL(k_precompute_tab8, dim3((unsigned)((b->n + 63) / 64)), dim3(64), 0, c->stream, (const niels29*)b->d_table, b->n, w8, t);
  // one-time build (~3 ms): waited for, so that another context sharing this bases object can never see the pointer before the table is complete
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); (void)dfree(owner, t); b->tab8_failed = true; return nullptr; }
  b->d_tab8[w8] = t;
  return t;
}
// the byte-multiple tables the commitments of small scalars read (k_msm_rows8 / k_msm_rows8w), built NOW instead of inside the first commitment that uses them: a caller that times
// `commit` (benches/bench.rs:54-66) prepares its generators first.  byte_windows = 1 (values < 2^8) or 2 (< 2^16).  A table that cannot be allocated is not an error (the bucket kernel serves).
int32_t lasso_bases_prepare(lasso_ctx* c, const lasso_bases* b, uint32_t byte_windows) {
  REQUIRE(c, b && byte_windows >= 1 && byte_windows <= 2 && !c->ahead_active && !c->lay_active && !c->tail_active);
  if (!msm_rows8_enabled()) return 0;
  for (uint32_t w = 0; w < byte_windows; w++) if (!ensure_tab8(c, b, w)) break;
  return 0;
}

// chunks per row.  Measured on MI355X (profiles/): the bucket kernel is VALU-issue-bound even at one wave per SIMD (the 81 independent
// multiply-adds of a field product pipeline back to back), so extra workgroups beyond one per CU only multiply the fixed per-workgroup
// reduction tree (2 rows x 482 chunks ran 310 us, 2 x 129 ran 180 us).  Aim for rows*K = 256 workgroups, never below 1024 pairs a chunk.
// group additions the REFERENCE's msm_bigint_wnaf (msm/mod.rs:91-164) performs for `rows` MSMs of n terms with num_bits-bit scalars:
// bucket accumulation n*W + bucket reduction W*2*2^c + window combine (W-1)*(c+1), c = ln_without_floats(n)+2 (:112-119,:322-325), W = ceil(num_bits/c).
// This is the algorithmic work unit of the MSM families' roofline (bench.py roofline_msm); the kernels here execute a different schedule
// (precomputed 4-bit window tables: one mixed addition per non-zero nibble, no per-window reduction, no doubling chain).
#ifdef LASSO_BN254
#define FR_MODULUS_BITS 254u
#else
#define FR_MODULUS_BITS 253u
#endif
static double msm_ref_adds(size_t rows, size_t n, uint32_t num_bits) {
  size_t lg = n <= 1 ? 0 : 64 - (size_t)__builtin_clzll((unsigned long long)(n - 1));
  const size_t cw = n < 32 ? 3 : lg * 69 / 100 + 2, W = (num_bits + cw - 1) / cw;
  return (double)rows * ((double)n * W + (double)W * 2.0 * (double)((size_t)1 << cw) + (double)(W > 0 ? W - 1 : 0) * (cw + 1.0));
}
static size_t msm_chunks(size_t rows, size_t n_cols, uint32_t W) {
  size_t pairs = n_cols * W, K = 1;
  if (rows < 256) { K = 256 / rows; size_t kmax = (pairs + 1023) / 1024; if (kmax < 1) kmax = 1; if (K > kmax) K = kmax; }
  size_t cols_per_chunk = (n_cols + K - 1) / K;
  return (n_cols + cols_per_chunk - 1) / cols_per_chunk;
}
// shared tail: bucket kernel over `rows` rows of `n_cols` scalars, then per-row sum of the chunk partials, then hand the points to the host
#define MSM_SMALL_ROWS 16   // results of up to this many rows return through the mapped buffer + flag (no memcpy, no stream sync)
// latency-shaped path (k_msm_direct): rows <= MSM_SMALL_ROWS of full-width canonical scalars, results through the mapped buffer + flag.
// Chunking: one workgroup per CU over all rows, whole multiples of 256 items per workgroup (every thread the same number of mixed adds),
// at most 8192 items (128 columns of LDS-staged scalars).  Scratch after the scalars: rows * K partial points.
static bool msm_direct_enabled() { static const bool on = [] { const char* v = getenv("LASSO_MSM_DIRECT"); return !(v && v[0] == '0'); }(); return on; }
static size_t msm_direct_chunks(size_t rows, size_t n_cols, uint32_t* items_per_chunk, size_t windows = MSM_WINDOWS) {
  const size_t total = n_cols * windows;
  // workgroups per launch: one per CU by default.  LASSO_MSM_DIRECT_WGS overrides it for tuning (more workgroups = shorter per-thread addition chains,
  // a larger cross-workgroup tree): the BN254 build's additions cost ~2.5x the Edwards ones and its balance point has not been measured yet
  static const size_t wgs = [] { const char* v = getenv("LASSO_MSM_DIRECT_WGS"); const long x = v ? atol(v) : 0; return (size_t)(x >= 1 && x <= 4096 ? x : 256); }();
  size_t K = wgs / rows; if (K < 1) K = 1;
  size_t ipc = ((total + K - 1) / K + 255) / 256 * 256;
  const size_t ipc_max = windows * 128;   // 128 columns of LDS-staged scalars
  if (ipc > ipc_max) ipc = ipc_max;
  *items_per_chunk = (uint32_t)ipc;
  return (total + ipc - 1) / ipc;
}
// bytes of point scratch an MSM of `rows` x `n_cols` may need after its scalars (whichever kernel serves it)
static size_t msm_pts_bytes(size_t rows, size_t n_cols) {
  uint32_t ipc; const size_t kd = rows <= MSM_SMALL_ROWS ? msm_direct_chunks(rows, n_cols, &ipc) : 0, kb = msm_chunks(rows, n_cols, MSM_WINDOWS);
  return (rows * (kd > kb ? kd : kb) + 2 * rows + 4) * sizeof(pt29) + 512;
}
// round 6: the few-row MSMs hand their points over as tagged elements (msm_direct_finish) when the context is in tagged mode; LASSO_MSM_TAGGED=0: the flag protocol (A/B switch)
static bool msm_tagged(lasso_ctx* c) { static const bool off = [] { const char* v = getenv("LASSO_MSM_TAGGED"); return v && v[0] == '0'; }(); return c->tagged && !off; }
#define MSM_RES(c) (msm_tagged(c) ? (ed_point*)(c)->d_tag : (ed_point*)(c)->d_small), (c)->d_counters + LASSO_MAX_PTRS + 8, (msm_tagged(c) ? LASSO_TAGGED : (c)->d_flag)
// mode 0: d_scal = canonical integers; 1: field elements in memory (Montgomery) form, converted by the kernel; 2: as 1 with the first n_cols - 2 columns
// multiplied by *scale and the last two columns = tail[0], tail[1] (k_msm_direct<MODE>)
// heads / gate_seq (the opening's tail chain): the launch also publishes heads[0][0], heads[1][0] behind the point, sits behind the gate of sequence number *gate_seq (whose tag it
// checks and whose number its result carries) and is NOT waited for here
static int32_t run_msm_direct(lasso_ctx* c, const uint8_t* d_scal, size_t row_stride, size_t rows, size_t n_cols, const MsmColMap& cm, const lasso_bases* b, uint8_t* scratch_after, lasso_point* out,
                              int mode = 0, const lasso_fr* scale = nullptr, const lasso_fr* tail = nullptr, uint32_t sstride = 1, uint32_t soffset = 0,
                              const lasso_fr* const* heads = nullptr, const uint32_t* gate_seq = nullptr) {
  const bool w8 = b->d_mult8 != nullptr; const size_t windows = w8 ? MsmD<8>::WINDOWS : MsmD<4>::WINDOWS;
  uint32_t ipc = 0; const size_t K = msm_direct_chunks(rows, n_cols, &ipc, windows);
  const uint32_t seq = gate_seq ? *gate_seq : next_seq(c);
  const fr_t* const h0 = heads ? (const fr_t*)heads[0] : nullptr; const fr_t* const h1 = heads ? (const fr_t*)heads[1] : nullptr;
  const uint32_t* const ggm = gate_seq ? (const uint32_t*)c->d_gmail : nullptr;
  {
    ProfScope ps(c, LASSO_K_MSM_DIRECT, (double)rows * n_cols * 32, msm_ref_adds(rows, n_cols, FR_MODULUS_BITS), false, (double)rows * n_cols * windows);
    const fr_t z = fr_zero();
#define LAUNCH_DIRECT(M, WB_, TAB_, SC, T0, T1) hipLaunchKernelGGL((k_msm_direct<M, WB_>), dim3((unsigned)K, (unsigned)rows), dim3(MSM_THREADS), 0, c->stream, (const uint32_t*)d_scal, row_stride / 4, (uint32_t)n_cols, ipc, cm, \
                       (const niels29*)TAB_, b->n, (pt29*)scratch_after, MSM_RES(c), seq, SC, T0, T1, ps.counter(), sstride, soffset, h0, h1, ggm)
    if (w8) {
      if (mode == 0) LAUNCH_DIRECT(0, 8, b->d_mult8, z, z, z);
      else if (mode == 1) LAUNCH_DIRECT(1, 8, b->d_mult8, z, z, z);
      else LAUNCH_DIRECT(2, 8, b->d_mult8, to_fr(scale), to_fr(tail), to_fr(tail + 1));
    } else {
      if (mode == 0) LAUNCH_DIRECT(0, 4, b->d_mult, z, z, z);
      else if (mode == 1) LAUNCH_DIRECT(1, 4, b->d_mult, z, z, z);
      else LAUNCH_DIRECT(2, 4, b->d_mult, to_fr(scale), to_fr(tail), to_fr(tail + 1));
    }
  }
  HIPCHK(c, hipGetLastError());
  if (gate_seq) return 0;   // released by lasso_bullet_post, collected by lasso_result_wait
  return wait_flag(c, seq, rows * (sizeof(ed_point) / sizeof(fr_t)), (lasso_fr*)out, msm_tagged(c));
}
static bool msm_direct_fused() { static const bool on = [] { const char* v = getenv("LASSO_MSM_FUSED"); return !(v && v[0] == '0'); }(); return on; }   // A/B switch: conversions and the bullet fold inside the MSM launch
// d_rows_out (device, rows x sizeof(pt29)): leave the row sums on the device in the kernels' own point form instead of handing them to the host —
// slab mode's partial row commitments, which go through lasso_rccl_allgather and lasso_points_reduce_compress
static int32_t run_msm(lasso_ctx* c, const uint8_t* d_scal, uint32_t bps, uint32_t W, size_t row_stride, size_t rows, size_t n_cols, const lasso_bases* b, uint8_t* scratch_after, lasso_point* out,
                       uint8_t* out_compressed = nullptr, void* d_rows_out = nullptr) {
  if (d_rows_out) out_compressed = (uint8_t*)d_rows_out;   // same kernel path as the compressed form: k_points_sum leaves pt29 row sums in d_final
  if (bps == 32 && rows <= MSM_SMALL_ROWS && !out_compressed && b->d_mult && msm_direct_enabled()) { const MsmColMap id = {0, 0, 0, 0}; return run_msm_direct(c, d_scal, row_stride, rows, n_cols, id, b, scratch_after, out); }
  const size_t K = msm_chunks(rows, n_cols, W);
  const size_t cols_per_chunk = (n_cols + K - 1) / K;
  pt29* d_partial = (pt29*)scratch_after;
  const bool small = rows <= MSM_SMALL_ROWS && !out_compressed;
  ed_point* d_final = small ? (ed_point*)c->d_small : (ed_point*)(((uintptr_t)(d_partial + rows * K) + 15) & ~(uintptr_t)15);
  const uint32_t seq = small ? next_seq(c) : 0;
  // many rows of small scalars (<= 16 bits): one table entry per non-zero byte (k_msm_rows8) instead of nibble buckets
  const niels29* t8[2] = {nullptr, nullptr};
  const uint32_t W8 = (W + 1) / 2;
  if (bps == 4 && W <= 4 && rows >= 32 && msm_rows8_enabled()) { t8[0] = ensure_tab8(c, b, 0); t8[1] = W8 > 1 && t8[0] ? ensure_tab8(c, b, 1) : t8[0]; if (!t8[1]) t8[0] = nullptr; }
  // full-width scalars over the signed byte-multiple table (k_msm_rows_full: 32 additions per scalar, no buckets) — MEASURED AND NOT THE DEFAULT (round 6, LASSO_MSM_FULL8=1 turns
  // it on): Spark C=16 2^22's E commitment 198 ms against the bucket kernel's 182 ms.  Half the additions, but every one of them reads its own 128-byte line of a 4.3 GB table
  // (8194 generators x 32 windows x 128 multiples): 2.1e9 random line reads = 1.38 TB/s, the rate HBM serves random lines at; the bucket kernel's 64-window table is 67 MB and
  // stays in the Infinity Cache.  profiles/r06_full_width_commit_ab.txt
  static const bool full8_on = [] { const char* v = getenv("LASSO_MSM_FULL8"); return v && v[0] == '1'; }();
  const bool full8 = bps == 32 && b->d_mult8 != nullptr && full8_on && !t8[0];
  // many long rows of full-width scalars: 12-bit signed windows over the SAME nibble-window table, 2048 buckets per row, 22 additions per scalar instead of 60
  // (msm_kernels.cuh k_msm_pip_*; round 6).  Rows go through in groups that keep the scratch (sorted pairs 84 B per column, bucket sums 288 KB per row) near 1.2 GB
  // (LASSO_MSM_PIP_SCRATCH_MB).  From 256 rows of 512 columns on (measured, profiles/r06_full_width_commit_ab.txt: 512 x 512 0.69 against 1.12 ms, 1024 x 1024 1.86 against 3.70,
  // 4096 x 4096 23.5 against 47.9).  LASSO_MSM_PIP=0: the bucket kernel (A/B switch).  A refused allocation falls back to it as well.
  // (both switches are read per call — a commitment of this size is milliseconds — so that one test process can run both forms)
  const bool pip_on = [] { const char* v = getenv("LASSO_MSM_PIP"); return !(v && v[0] == '0'); }();
  const size_t pip_min_cols = [] { const char* v = getenv("LASSO_MSM_PIP_MIN_COLS"); const long x = v ? atol(v) : 512; return (size_t)(x < 1 ? 1 : x); }();
  size_t pip_group = 0, pip_row_bytes = 0, pip_items = 0;
  if (bps == 32 && pip_on && !full8 && K == 1 && rows >= 256 && n_cols >= pip_min_cols && n_cols < ((size_t)1 << 26) && b->n * MSM_WINDOWS < ((size_t)1 << 31)) {
    pip_items = n_cols * MSM_PIP_WINDOWS;
    pip_row_bytes = ((pip_items * 4 + (MSM_PIP_BUCKETS + 1) * 4 + MSM_PIP_BUCKETS * 2 + MSM_PIP_BUCKETS * sizeof(pt29) + n_cols) + 255) & ~(size_t)255;
    const size_t pip_mb = [] { const char* v = getenv("LASSO_MSM_PIP_SCRATCH_MB"); const long x = v ? atol(v) : 1200; return (size_t)(x < 16 ? 16 : x); }();
    pip_group = (pip_mb << 20) / pip_row_bytes; if (pip_group < 64) pip_group = 64; if (pip_group > rows) pip_group = rows;
    if (ensure_pip(c, pip_group * pip_row_bytes + 256) != 0) pip_group = 0;
  }
  {
    ProfScope ps(c, LASSO_K_MSM, (double)rows * n_cols * bps, msm_ref_adds(rows, n_cols, bps == 4 ? 4 * W : FR_MODULUS_BITS), rows > MSM_SMALL_ROWS,
                 (double)rows * n_cols * (t8[0] ? W8 : full8 ? 32 : pip_group ? MSM_PIP_WINDOWS : W));
    // many SHORT rows: one wave per row (k_msm_rows8w: 64 additions per lane and a 6-level tree inside the wave instead of 16 per thread and a 256-point tree).  Measured
    // (profiles/r04_ab_rows8w.txt): -9 % on the headline's E (4096 one-byte columns), -8 % on BN254 configs[1], +2 % on configs[2]'s 16384-column rows, where a thread of the
    // 256-lane kernel already runs 64 additions — hence the column bound.  LASSO_MSM_ROWS8W=0: A/B switch
    static const bool rows8w = [] { const char* v = getenv("LASSO_MSM_ROWS8W"); return !(v && v[0] == '0'); }();
    // rows per wave (round 6 experiment, NOT the default: LASSO_MSM_ROWS8W_WAVES=2048 caps a launch at two waves per SIMD, all resident at once — measured 0.966 ms against
    // 0.896 ms at one row per wave on the headline's E, profiles/r06_madd_bench_curve25519.txt section C: two waves per SIMD hide less than three, the tail of the
    // three-wave schedule costs less than that)
    static const size_t w_max = [] { const char* v = getenv("LASSO_MSM_ROWS8W_WAVES"); const long x = v ? atol(v) : 0; return (size_t)(x < 0 ? 0 : x); }();
    const size_t rpw = w_max ? (rows + w_max - 1) / w_max : 1, waves = (rows + rpw - 1) / rpw;
    if (t8[0] && rows8w && K == 1 && rows >= 1024 && n_cols * W8 <= 8192) hipLaunchKernelGGL(k_msm_rows8w, dim3((unsigned)((waves + MSM_THREADS / 64 - 1) / (MSM_THREADS / 64))), dim3(MSM_THREADS), 0, c->stream, (const uint32_t*)d_scal, row_stride / 4,
                                                                      (uint32_t)n_cols, W8, t8[0], t8[1], b->n, d_partial, (uint32_t)rows, ps.counter(), (uint32_t)rpw);
    else if (t8[0]) hipLaunchKernelGGL(k_msm_rows8, dim3((unsigned)K, (unsigned)rows), dim3(MSM_THREADS), 0, c->stream, (const uint32_t*)d_scal, row_stride / 4, (uint32_t)n_cols, (uint32_t)cols_per_chunk, W8,
                                  t8[0], t8[1], b->n, d_partial, ps.counter());
    else if (pip_group) {
      uint8_t* base = (uint8_t*)c->d_pip;
      uint32_t* d_sorted = (uint32_t*)base; uint32_t* d_offs = d_sorted + pip_group * pip_items; uint16_t* d_perm = (uint16_t*)(d_offs + pip_group * (MSM_PIP_BUCKETS + 1));
      uint8_t* d_vtop = (uint8_t*)(d_perm + pip_group * MSM_PIP_BUCKETS);
      pt29* d_bk = (pt29*)((((uintptr_t)(d_vtop + pip_group * n_cols)) + 15) & ~(uintptr_t)15);
      for (size_t r0 = 0; r0 < rows; r0 += pip_group) {
        const unsigned g = (unsigned)(rows - r0 < pip_group ? rows - r0 : pip_group);
        hipLaunchKernelGGL(k_msm_pip_sort, dim3(g), dim3(MSM_THREADS), 0, c->stream, d_scal + r0 * row_stride, row_stride, (uint32_t)n_cols, (uint32_t)b->n, d_sorted, pip_items, d_offs, d_perm, d_vtop, ps.counter());
        hipLaunchKernelGGL(k_msm_pip_accumulate, dim3(g, MSM_PIP_PER_THREAD), dim3(MSM_THREADS), 0, c->stream, (const uint32_t*)d_sorted, pip_items, (const uint32_t*)d_offs, (const uint16_t*)d_perm, (const niels29*)b->d_table, d_bk);
        hipLaunchKernelGGL(k_msm_pip_reduce, dim3(g), dim3(MSM_THREADS), 0, c->stream, (const pt29*)d_bk, (const uint8_t*)d_vtop, (uint32_t)n_cols, (const niels29*)b->d_table + (size_t)(MSM_WINDOWS - 1) * b->n, d_partial + r0);
      }
    }
    else if (full8) hipLaunchKernelGGL((k_msm_rows_full<8>), dim3((unsigned)K, (unsigned)rows), dim3(MSM_THREADS), 0, c->stream, (const uint32_t*)d_scal, row_stride / 4, (uint32_t)n_cols, (uint32_t)cols_per_chunk,
                                       (const niels29*)b->d_mult8, b->n, d_partial, ps.counter());
    else hipLaunchKernelGGL(k_msm_buckets, dim3((unsigned)K, (unsigned)rows), dim3(MSM_THREADS), 0, c->stream, d_scal, bps, W, row_stride, n_cols, cols_per_chunk, (const niels29*)b->d_table, b->n, d_partial, ps.counter());
    hipLaunchKernelGGL(k_points_sum, dim3((unsigned)rows), dim3(MSM_THREADS), 0, c->stream, (const pt29*)d_partial, (uint32_t)K, d_final, out_compressed ? (uint32_t*)d_final : (uint32_t*)nullptr,
                       c->d_counters + LASSO_MAX_PTRS + 1, small ? c->d_flag : (uint32_t*)nullptr, seq);
  }
  HIPCHK(c, hipGetLastError());
  if (d_rows_out) { HIPCHK(c, hipMemcpyAsync(d_rows_out, d_final, rows * sizeof(pt29), hipMemcpyDeviceToDevice, c->stream)); return 0; }
  if (out_compressed) {   // d_final holds the row sums as pt29 (144 B per row: the scratch is sized for it, see hyrax_commit_impl); 32 wire bytes per row go out
    if (rows <= ((size_t)1 << 16)) {   // wire bytes straight into the host-mapped result buffer, handed over behind the sequence flag
      int32_t rc = ensure_small(c, rows); if (rc) return rc;
      hipLaunchKernelGGL(k_points_compress, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, (const pt29*)d_final, rows, (uint32_t*)c->d_small);
      HIPCHK(c, hipGetLastError());
      return fetch_small(c, rows, (lasso_fr*)out_compressed);
    }
    uint32_t* d_wire = (uint32_t*)(((uintptr_t)((pt29*)d_final + rows) + 15) & ~(uintptr_t)15);
    hipLaunchKernelGGL(k_points_compress, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, (const pt29*)d_final, rows, d_wire);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out_compressed, d_wire, rows * 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
  }
  if (small) return wait_flag(c, seq, rows * (sizeof(ed_point) / sizeof(fr_t)), (lasso_fr*)out);
  HIPCHK(c, hipMemcpyAsync(out, d_final, rows * sizeof(ed_point), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
static int32_t hyrax_commit_impl(lasso_ctx* c, const lasso_fr* d_Z, size_t l_size, size_t r_size, const lasso_bases* b, lasso_point* out, uint8_t* out_compressed, void* d_rows_out = nullptr);
int32_t lasso_hyrax_commit_rows_dev(lasso_ctx* c, const lasso_fr* d_Z, size_t l_size, size_t r_size, const lasso_bases* b, void* d_rows) { REQUIRE(c, d_rows); return hyrax_commit_impl(c, d_Z, l_size, r_size, b, nullptr, nullptr, d_rows); }
size_t lasso_point_row_bytes(void) { return sizeof(pt29); }
int32_t lasso_hyrax_commit(lasso_ctx* c, const lasso_fr* d_Z, size_t l_size, size_t r_size, const lasso_bases* b, lasso_point* out) { REQUIRE(c, out); return hyrax_commit_impl(c, d_Z, l_size, r_size, b, out, nullptr); }
int32_t lasso_hyrax_commit_compressed(lasso_ctx* c, const lasso_fr* d_Z, size_t l_size, size_t r_size, const lasso_bases* b, uint8_t* out32) { REQUIRE(c, out32); return hyrax_commit_impl(c, d_Z, l_size, r_size, b, nullptr, out32); }
static int32_t hyrax_commit_impl(lasso_ctx* c, const lasso_fr* d_Z, size_t l_size, size_t r_size, const lasso_bases* b, lasso_point* out, uint8_t* out_compressed, void* d_rows_out) {
  REQUIRE(c, d_Z && b && l_size >= 1 && r_size >= 1 && r_size <= b->n && l_size < ((size_t)1 << 31));
  const size_t n = l_size * r_size;
  const size_t pts_bytes = msm_pts_bytes(l_size, r_size);   // chunk partials + row sums (pt29 or ed_point) + wire bytes
  // small-scalar regime first (4 bytes per scalar); the 32-byte form is only allocated if some scalar needs it — a 2^25-element polynomial of
  // table indices / timestamps then needs 128 MiB of scratch instead of 1 GiB (and no hipMalloc at all after densify)
  int32_t rc = ensure_scratch(c, n * 4 + 256 + pts_bytes); if (rc) return rc;
  uint8_t* d_scal = (uint8_t*)c->d_scratch;
  HIPCHK(c, hipMemsetAsync(c->d_flags, 0, 8, c->stream));
  {
    ProfScope ps(c, LASSO_K_MISC, 36.0 * n);
    hipLaunchKernelGGL(k_fr_to_u32, dim3(grid_for(n, 4096)), dim3(256), 0, c->stream, (const fr_t*)d_Z, n, (uint32_t*)d_scal, c->d_flags);
  }
  uint32_t flags[2];
  HIPCHK(c, hipMemcpyAsync(flags, c->d_flags, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (!flags[1]) {  // every scalar < 2^32: the reference's small-scalar regime (msm/mod.rs:95-106)
    uint32_t bits = 0; while (bits < 32 && (flags[0] >> bits)) bits++;
    uint32_t W = (bits + 3) / 4; if (W == 0) W = 1;   // 4-bit windows actually populated
    return run_msm(c, d_scal, 4, W, r_size * 4, l_size, r_size, b, d_scal + ((n * 4 + 255) & ~(size_t)255), out, out_compressed, d_rows_out);
  }
  rc = ensure_scratch(c, n * 32 + pts_bytes); if (rc) return rc;
  d_scal = (uint8_t*)c->d_scratch;
  hipLaunchKernelGGL(k_fr_to_canonical, dim3(grid_for(n, 4096)), dim3(256), 0, c->stream, (const fr_t*)d_Z, n, (fr_t*)d_scal);
  return run_msm(c, d_scal, 32, MSM_WINDOWS, r_size * 32, l_size, r_size, b, d_scal + n * 32, out, out_compressed, d_rows_out);
}
// The commitment of a polynomial whose canonical values the caller already holds as u32 (Z[i] = F::from(d_u32[i]), e.g. E = T[dim] with a small
// table T, or the dim / timestamp polynomials): no conversion pass over the 32-byte elements and no max-bit readback.  max_value bounds the values.
int32_t lasso_hyrax_commit_compressed_u32(lasso_ctx* c, const uint32_t* d_u32, uint32_t max_value, size_t l_size, size_t r_size, const lasso_bases* b, uint8_t* out32) {
  REQUIRE(c, d_u32 && out32 && b && l_size >= 1 && r_size >= 1 && r_size <= b->n && l_size < ((size_t)1 << 31));
  int32_t rc = ensure_scratch(c, msm_pts_bytes(l_size, r_size) + 256); if (rc) return rc;
  uint32_t bits = 0; while (bits < 32 && (max_value >> bits)) bits++;
  uint32_t W = (bits + 3) / 4; if (W == 0) W = 1;
  return run_msm(c, (const uint8_t*)d_u32, 4, W, r_size * 4, l_size, r_size, b, (uint8_t*)c->d_scratch, nullptr, out32);
}
__global__ void __launch_bounds__(256) k_gather_u32(const uint32_t* __restrict__ table, const uint32_t* __restrict__ idx, size_t n, uint32_t* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = table[idx[i]];
}
int32_t lasso_gather_u32(lasso_ctx* c, const uint32_t* d_table, const uint32_t* d_idx, size_t n, uint32_t* d_out) {
  REQUIRE(c, d_table && d_idx && d_out); if (!n) return 0;
  ProfScope ps(c, LASSO_K_MISC, 12.0 * n);
  hipLaunchKernelGGL(k_gather_u32, dim3(grid_for(n)), dim3(256), 0, c->stream, d_table, d_idx, n, d_out);
  HIPCHK(c, hipGetLastError()); return 0;
}
int32_t lasso_msm(lasso_ctx* c, const lasso_bases* b, const lasso_fr* scalars, size_t n, lasso_point* out) {
  REQUIRE(c, b && scalars && out && n >= 1 && n <= b->n);
  int32_t rc = ensure_scratch(c, n * 64 + msm_pts_bytes(1, n)); if (rc) return rc;
  fr_t* d_in = (fr_t*)c->d_scratch; fr_t* d_can = d_in + n;
  HIPCHK(c, hipMemcpyAsync(d_in, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_fr_to_canonical, dim3(grid_for(n)), dim3(256), 0, c->stream, (const fr_t*)d_in, n, d_can);
  return run_msm(c, (const uint8_t*)d_can, 32, MSM_WINDOWS, n * 32, 1, n, b, (uint8_t*)(d_can + n), out);
}

int32_t lasso_msm_dev(lasso_ctx* c, const lasso_bases* b, const lasso_fr* d_scalars, size_t n, lasso_point* out) {
  REQUIRE(c, b && d_scalars && out && n >= 1 && n <= b->n);
  int32_t rc = ensure_scratch(c, n * 32 + msm_pts_bytes(1, n)); if (rc) return rc;
  fr_t* d_can = (fr_t*)c->d_scratch;
  if (b->d_mult && msm_direct_enabled() && msm_direct_fused()) {   // the latency-shaped kernel converts its own columns: one launch instead of two
    const MsmColMap id = {0, 0, 0, 0};
    return run_msm_direct(c, (const uint8_t*)d_scalars, n * 32, 1, n, id, b, (uint8_t*)c->d_scratch, out, 1);
  }
  hipLaunchKernelGGL(k_fr_to_canonical, dim3(grid_for(n)), dim3(256), 0, c->stream, (const fr_t*)d_scalars, n, d_can);
  return run_msm(c, (const uint8_t*)d_can, 32, MSM_WINDOWS, n * 32, 1, n, b, (uint8_t*)(d_can + n), out);
}
__global__ void __launch_bounds__(256) k_scale_to_integers(const fr_t* __restrict__ src, size_t n, fr_t scale, fr_t t0, fr_t t1, fr_t* __restrict__ dst) {
  const fr29 ss = fr29_unpack_s(scale); fr29 k32 = fr29_zero(); k32.v[0] = 32;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = fr29_store(fr29_mul(fr29_mul(fr29_unpack_u(src[i]), ss), k32));   // (u * s) = u-form, then -> integer
  if (blockIdx.x == 0 && threadIdx.x == 0) { dst[n] = fr29_to_integer(fr29_unpack_u(t0)); dst[n + 1] = fr29_to_integer(fr29_unpack_u(t1)); }
}
int32_t lasso_msm_dev_scaled(lasso_ctx* c, const lasso_bases* b, const lasso_fr* d_scalars, size_t n, const lasso_fr* scale, const lasso_fr* tail, lasso_point* out) {
  REQUIRE(c, b && d_scalars && scale && tail && out && n >= 1 && n + 2 <= b->n);
  const size_t row = n + 2;
  int32_t rc = ensure_scratch(c, row * 32 + msm_pts_bytes(1, row)); if (rc) return rc;
  fr_t* d_can = (fr_t*)c->d_scratch;
  if (b->d_mult && msm_direct_enabled() && msm_direct_fused()) {   // scaling and conversion inside the MSM launch (the two tail columns are kernel arguments, never read from d_scalars)
    const MsmColMap id = {0, 0, 0, 0};
    return run_msm_direct(c, (const uint8_t*)d_scalars, row * 32, 1, row, id, b, (uint8_t*)c->d_scratch, out, 2, scale, tail);
  }
  hipLaunchKernelGGL(k_scale_to_integers, dim3(grid_for(n)), dim3(256), 0, c->stream, (const fr_t*)d_scalars, n, to_fr(scale), to_fr(tail), to_fr(tail + 1), d_can);
  return run_msm(c, (const uint8_t*)d_can, 32, MSM_WINDOWS, row * 32, 1, row, b, (uint8_t*)(d_can + row), out);
}
int32_t lasso_inner_products_lr(lasso_ctx* c, const lasso_fr* d_a, const lasso_fr* d_b, size_t nk, lasso_fr* out) {
  REQUIRE(c, d_a && d_b && out && nk >= 2 && (nk & (nk - 1)) == 0);
  const size_t half = nk / 2; const unsigned nx = grid_for(half, 256);
  int32_t rc = ensure_scratch(c, (size_t)nx * 2 * sizeof(fr_t)); if (rc) return rc;
  {
    ProfScope ps(c, LASSO_K_DOT, 64.0 * nk);
    hipLaunchKernelGGL(k_inner_lr, dim3(nx), dim3(256), 0, c->stream, (const fr_t*)d_a, (const fr_t*)d_b, half, (fr_t*)c->d_scratch);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(LASSO_BLOCK), 0, c->stream, (const fr_t*)c->d_scratch, nx, 2u, c->d_small);
  }
  HIPCHK(c, hipGetLastError());
  return fetch_small(c, 2, out);
}
int32_t lasso_bullet_lr(lasso_ctx* c, const lasso_bases* b, size_t n, const lasso_fr* d_a, size_t nk, const lasso_fr* d_w, const lasso_fr* tail, lasso_point* out) {
  REQUIRE(c, b && d_a && d_w && tail && out && n >= 2 && (n & (n - 1)) == 0 && nk >= 2 && nk <= n && (nk & (nk - 1)) == 0 && n + 2 <= b->n);
  const size_t row = n + 2;
  int32_t rc = ensure_scratch(c, 2 * row * 32 + msm_pts_bytes(2, row)); if (rc) return rc;
  fr_t* SL = (fr_t*)c->d_scratch; fr_t* SR = SL + row;
  hipLaunchKernelGGL(k_bullet_expand, dim3(grid_for(n)), dim3(256), 0, c->stream, (const fr_t*)d_a, nk, (const fr_t*)d_w, n, to_fr(tail), to_fr(tail + 1), to_fr(tail + 2), to_fr(tail + 3), SL, SR);
  return run_msm(c, (const uint8_t*)SL, 32, MSM_WINDOWS, row * 32, 2, row, b, (uint8_t*)(SR + row), out);
}
// fold + scalars + both MSMs in one launch (k_bullet_msm): K chunk workgroups per row over the row's local columns, plus one per row for a', b', the inner product and c*Q + blind*H.
// world / rank: slab mode (the table `b` holds the rank's n / world generators, then Q, H; the result is the rank's PARTIAL L, R).
static int32_t bullet_round_fused(lasso_ctx* c, const lasso_bases* b, size_t n, const lasso_fr* d_a_in, const lasso_fr* d_b_in, const lasso_fr* d_w_in, lasso_fr* d_a_out, lasso_fr* d_b_out,
                                  lasso_fr* d_w_out, size_t nk, const lasso_fr* u, const lasso_fr* u_inv, const lasso_fr* blinds, lasso_point* out, uint32_t world, uint32_t rank, bool ahead = false) {
  const bool fold = u != nullptr || ahead;   // ahead: a folding round enqueued before its challenge exists (the kernel waits for lasso_bullet_post)
  // chunks per row: (workgroups of the launch - 2 extra) / 2 rows, items shared out evenly (a multiple of 64 keeps whole columns together where it can)
  static const size_t wgs = [] { const char* v = getenv("LASSO_MSM_DIRECT_WGS"); const long x = v ? atol(v) : 0; return (size_t)(x >= 4 && x <= 4096 ? x : 256); }();
  const size_t n_loc = n / world, cols = (nk / 2 >= world) ? n_loc / 2 : n_loc;   // the longest row's local columns
  const bool w8 = b->d_mult8 != nullptr; const size_t windows = w8 ? MsmD<8>::WINDOWS : MsmD<4>::WINDOWS;
  const size_t total = cols * windows, kmax = (wgs - 2) / 2;
  size_t ipc_ = (total + kmax - 1) / kmax; ipc_ = (ipc_ + windows - 1) / windows * windows; if (ipc_ < 256) ipc_ = 256; if (ipc_ > windows * 128) ipc_ = windows * 128;
  const uint32_t ipc = (uint32_t)ipc_; const size_t K = (total + ipc_ - 1) / ipc_;
  int32_t rc = ensure_scratch(c, 2 * (K + 1) * sizeof(pt29) + 512); if (rc) return rc;
  const uint32_t seq = next_seq(c);
  if (ahead) hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, c->stream, (const uint32_t*)c->mail_d, c->d_gmail, seq, 1u);   // waits for lasso_bullet_post's two scalars; the round starts when it ends
  {
    const size_t row = n_loc / 2 + 2;
    ProfScope ps(c, LASSO_K_MSM_DIRECT, 2.0 * row * 32 + (fold ? 96.0 * 2 * nk : 64.0 * nk), msm_ref_adds(2, row, FR_MODULUS_BITS), false, 2.0 * row * windows);
    const fr_t z = fr_zero();
#define LAUNCH_BULLET(FOLD_, WB_, TAB_, AO, BO, WO, U, UI) hipLaunchKernelGGL((k_bullet_msm<FOLD_, WB_>), dim3((unsigned)K + 1, 2), dim3(MSM_THREADS), 0, c->stream, (const fr_t*)d_a_in, (const fr_t*)d_b_in, (const fr_t*)d_w_in, \
                                 (fr_t*)AO, (fr_t*)BO, (fr_t*)WO, (uint32_t)nk, (uint32_t)n, U, UI, to_fr(blinds), to_fr(blinds + 1), ipc, (const niels29*)TAB_, b->n, (pt29*)c->d_scratch, \
                                 MSM_RES(c), seq, ps.counter(), world, rank, (const uint32_t*)(ahead ? c->mail_d : nullptr), c->d_gmail)
    if (fold) { const fr_t uu = ahead ? z : to_fr(u), ui = ahead ? z : to_fr(u_inv); if (w8) LAUNCH_BULLET(true, 8, b->d_mult8, d_a_out, d_b_out, d_w_out, uu, ui); else LAUNCH_BULLET(true, 4, b->d_mult, d_a_out, d_b_out, d_w_out, uu, ui); }
    else { if (w8) LAUNCH_BULLET(false, 8, b->d_mult8, nullptr, nullptr, nullptr, z, z); else LAUNCH_BULLET(false, 4, b->d_mult, nullptr, nullptr, nullptr, z, z); }
  }
  HIPCHK(c, hipGetLastError());
  if (ahead) { c->ahead_active = true; c->ahead_bullet = true; c->ahead_seq = seq; c->ahead_tagged = msm_tagged(c); c->ahead_count = 2 * (sizeof(ed_point) / sizeof(fr_t)); return 0; }
  return wait_flag(c, seq, 2 * (sizeof(ed_point) / sizeof(fr_t)), (lasso_fr*)out, msm_tagged(c));
}
// slab mode of the opening (include/lasso_hip.h): this rank's share of L and R over its residue class of the generators
int32_t lasso_bullet_round_slab(lasso_ctx* c, const lasso_bases* b, size_t n, uint32_t world, uint32_t rank, const lasso_fr* d_a_in, const lasso_fr* d_b_in, const lasso_fr* d_w_in, lasso_fr* d_a_out,
                                lasso_fr* d_b_out, lasso_fr* d_w_out, size_t nk, const lasso_fr* u, const lasso_fr* u_inv, const lasso_fr* blinds, lasso_point* out) {
  REQUIRE(c, b && d_a_in && d_b_in && d_w_in && blinds && out && n >= 2 && (n & (n - 1)) == 0 && nk >= 2 && nk <= n && (nk & (nk - 1)) == 0);
  REQUIRE(c, world >= 1 && (world & (world - 1)) == 0 && rank < world && world <= n && n / world + 2 <= b->n);
  if (u) REQUIRE(c, u_inv && d_a_out && d_b_out && d_w_out && 2 * nk <= n && d_a_out != d_a_in && d_b_out != d_b_in && d_w_out != d_w_in);
  if (!b->d_mult) return fail(c, LASSO_ERR_UNSUPPORTED, "lasso_bullet_round_slab needs the digit-multiple table of the bases (lasso_bases_has_direct)");
  return bullet_round_fused(c, b, n, d_a_in, d_b_in, d_w_in, d_a_out, d_b_out, d_w_out, nk, u, u_inv, blinds, out, world, rank);
}
// out = sum_{jl < n/world} (scale *) d_scalars[jl * world + rank] * bases[jl]  (+ tail[0] * bases[n/world] + tail[1] * bases[n/world + 1]): the rank's share of an MSM over the
// whole replicated scalar vector — Cx = <x, G> and delta = d * g_hat + r_delta * h of the opening (dot_product.rs:183-186, :219-224).  scale / tail may be NULL (1 / no extra terms).
int32_t lasso_msm_dev_slab(lasso_ctx* c, const lasso_bases* b, const lasso_fr* d_scalars, size_t n, uint32_t world, uint32_t rank, const lasso_fr* scale, const lasso_fr* tail, lasso_point* out) {
  REQUIRE(c, b && d_scalars && out && n >= 1 && world >= 1 && (world & (world - 1)) == 0 && rank < world && n % world == 0 && n / world + 2 <= b->n);
  if (!b->d_mult) return fail(c, LASSO_ERR_UNSUPPORTED, "lasso_msm_dev_slab needs the digit-multiple table of the bases (lasso_bases_has_direct)");
  const size_t nl = n / world, row = nl + 2;
  int32_t rc = ensure_scratch(c, msm_pts_bytes(1, row) + 512); if (rc) return rc;
  const lasso_fr one = [] { lasso_fr o; const fr_t r = fr_one(); memcpy(&o, r.v, 32); return o; }();
  const lasso_fr zeros[2] = {{{0, 0, 0, 0}}, {{0, 0, 0, 0}}};
  const MsmColMap id = {0, 0, 0, 0};
  return run_msm_direct(c, (const uint8_t*)d_scalars, 0, 1, row, id, b, (uint8_t*)c->d_scratch, out, 2, scale ? scale : &one, tail ? tail : zeros, world, rank);
}
int32_t lasso_bases_has_direct(const lasso_bases* b) { return b && b->d_mult ? 1 : 0; }
int32_t lasso_bullet_round(lasso_ctx* c, const lasso_bases* b, size_t n, const lasso_fr* d_a_in, const lasso_fr* d_b_in, const lasso_fr* d_w_in, lasso_fr* d_a_out, lasso_fr* d_b_out,
                           lasso_fr* d_w_out, size_t nk, const lasso_fr* u, const lasso_fr* u_inv, const lasso_fr* blinds, lasso_point* out) {
  REQUIRE(c, b && d_a_in && d_b_in && d_w_in && blinds && out && n >= 2 && (n & (n - 1)) == 0 && nk >= 2 && nk <= n && (nk & (nk - 1)) == 0 && n + 2 <= b->n);
  const bool fold = u != nullptr;
  if (fold) REQUIRE(c, u_inv && d_a_out && d_b_out && d_w_out && 2 * nk <= n && d_a_out != d_a_in && d_b_out != d_b_in && d_w_out != d_w_in);
  const bool direct = b->d_mult && msm_direct_enabled();
  if (direct && msm_direct_fused()) return bullet_round_fused(c, b, n, d_a_in, d_b_in, d_w_in, d_a_out, d_b_out, d_w_out, nk, u, u_inv, blinds, out, 1, 0);
  const size_t row = direct ? n / 2 + 2 : n + 2;   // compact rows for k_msm_direct: only the non-zero half
  const unsigned nx = grid_for(n / 2, 64);
  int32_t rc = ensure_scratch(c, 2 * row * 32 + (size_t)nx * 2 * sizeof(fr_t) + msm_pts_bytes(2, row)); if (rc) return rc;
  fr_t* SL = (fr_t*)c->d_scratch; fr_t* SR = SL + row; fr_t* partials = SR + row;
  {
    ProfScope ps(c, LASSO_K_MISC, 64.0 * row + (fold ? 96.0 * 2 * nk : 64.0 * nk));
    const fr_t z = fr_zero();
    if (fold) hipLaunchKernelGGL((k_bullet_step<true>), dim3(nx), dim3(256), 0, c->stream, (const fr_t*)d_a_in, (const fr_t*)d_b_in, (const fr_t*)d_w_in, (fr_t*)d_a_out, (fr_t*)d_b_out, (fr_t*)d_w_out, nk, n,
                                 to_fr(u), to_fr(u_inv), to_fr(blinds), to_fr(blinds + 1), SL, SR, partials, c->d_counters + LASSO_MAX_PTRS + 2, direct ? 1u : 0u);
    else hipLaunchKernelGGL((k_bullet_step<false>), dim3(nx), dim3(256), 0, c->stream, (const fr_t*)d_a_in, (const fr_t*)d_b_in, (const fr_t*)d_w_in, (fr_t*)nullptr, (fr_t*)nullptr, (fr_t*)nullptr, nk, n,
                            z, z, to_fr(blinds), to_fr(blinds + 1), SL, SR, partials, c->d_counters + LASSO_MAX_PTRS + 2, direct ? 1u : 0u);
  }
  HIPCHK(c, hipGetLastError());
  if (direct) { const MsmColMap cm = {(uint32_t)nk, (uint32_t)(nk / 2), (uint32_t)(n / 2), (uint32_t)n}; return run_msm_direct(c, (const uint8_t*)SL, row * 32, 2, row, cm, b, (uint8_t*)(partials + (size_t)nx * 2), out); }
  return run_msm(c, (const uint8_t*)SL, 32, MSM_WINDOWS, row * 32, 2, row, b, (uint8_t*)(partials + (size_t)nx * 2), out);
}
// ---- a folding round LAUNCHED AHEAD of its challenge (include/lasso_hip.h): enqueue, later post u / u^-1, then collect like any deferred result
static bool bullet_ahead_possible(lasso_ctx* c, const lasso_bases* b) {
  static const bool off = [] { const char* v = getenv("LASSO_BULLET_AHEAD"); return v && v[0] == '0'; }();
  const bool bracketed = ((c->prof_mask >> LASSO_K_MSM_DIRECT) & 1u) && !(c->prof_mask & 0x40000000u);   // a launch between profiling events would count the wait for the host as kernel time
  return !off && !bracketed && b && b->d_mult && msm_direct_enabled() && msm_direct_fused();
}
int32_t lasso_bullet_ahead_ok(lasso_ctx* c, const lasso_bases* b) { return c && bullet_ahead_possible(c, b) ? 1 : 0; }
int32_t lasso_bullet_round_ahead(lasso_ctx* c, const lasso_bases* b, size_t n, const lasso_fr* d_a_in, const lasso_fr* d_b_in, const lasso_fr* d_w_in, lasso_fr* d_a_out, lasso_fr* d_b_out,
                                 lasso_fr* d_w_out, size_t nk, const lasso_fr* blinds) {
  REQUIRE(c, b && d_a_in && d_b_in && d_w_in && d_a_out && d_b_out && d_w_out && blinds && n >= 2 && (n & (n - 1)) == 0 && nk >= 2 && 2 * nk <= n && (nk & (nk - 1)) == 0 && n + 2 <= b->n &&
             d_a_out != d_a_in && d_b_out != d_b_in && d_w_out != d_w_in && !c->ahead_active && !c->tail_active && !c->defer_next);
  if (!bullet_ahead_possible(c, b)) return fail(c, LASSO_ERR_UNSUPPORTED, "lasso_bullet_round_ahead: not available for this generator set / configuration (lasso_bullet_ahead_ok)");
  return bullet_round_fused(c, b, n, d_a_in, d_b_in, d_w_in, d_a_out, d_b_out, d_w_out, nk, nullptr, nullptr, blinds, nullptr, 1, 0, true);
}
// ---- THE END OF AN OPENING enqueued ahead of its last challenge (round 6).  After the last folding round the host used to: draw u, launch the fold of the two-element
// a, b and of the weights, launch lasso_read_heads and wait for it, launch the delta MSM over the folded weights, wait.  Three launches and two hand-offs behind one host turn.  Here
// the whole chain — gate, k_bullet_fold (u, u^-1 from the gate), k_msm_direct<2> over d_w_out scaled by *scale with the two tail terms, which also publishes a[0] and b[0] — is in
// the stream before the host has the last round's L and R; lasso_bullet_post releases it and lasso_result_wait(ctx, out, 6) delivers the point (4 values) and the two heads.
//   = lasso_bullet_fold(d_a, d_b, 2, d_w, nw, d_w_out, u, u_inv); lasso_read_heads({d_a, d_b}); lasso_msm_dev_scaled(bases, d_w_out, n, scale, tail)     with n = 2 nw
int32_t lasso_bullet_tail_ahead_ok(lasso_ctx* c, const lasso_bases* b) {
  static const bool off = [] { const char* v = getenv("LASSO_BULLET_TAIL_AHEAD"); return v && v[0] == '0'; }();
  return c && !off && bullet_ahead_possible(c, b) && msm_tagged(c) ? 1 : 0;
}
int32_t lasso_bullet_tail_ahead(lasso_ctx* c, const lasso_bases* b, size_t n, lasso_fr* d_a, lasso_fr* d_b, const lasso_fr* d_w, size_t nw, lasso_fr* d_w_out, const lasso_fr* scale, const lasso_fr* tail) {
  REQUIRE(c, b && d_a && d_b && d_w && d_w_out && scale && tail && n >= 2 && (n & (n - 1)) == 0 && 2 * nw == n && n + 2 <= b->n && d_w != d_w_out && !c->ahead_active && !c->tail_active && !c->defer_next && !c->lay_active);
  if (!lasso_bullet_tail_ahead_ok(c, b)) return fail(c, LASSO_ERR_UNSUPPORTED, "lasso_bullet_tail_ahead: not available for this generator set / configuration (lasso_bullet_tail_ahead_ok)");
  const size_t row = n + 2;
  c->no_grow = true;   // nothing may synchronise the stream here: the previous round's kernel may still be running, its result uncollected
  int32_t rc = ensure_scratch(c, msm_pts_bytes(1, row) + 512); if (!rc) rc = ensure_small(c, 8);
  c->no_grow = false;
  if (rc) return rc == LASSO_ERR_UNSUPPORTED ? fail(c, rc, "lasso_bullet_tail_ahead: a buffer would have to grow while kernels are in flight") : rc;
  const uint32_t seq = next_seq(c);
  hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, c->stream, (const uint32_t*)c->mail_d, c->d_gmail, seq, 1u);
  const fr_t z = fr_zero();
  hipLaunchKernelGGL(k_bullet_fold, dim3(grid_for(nw)), dim3(256), 0, c->stream, (fr_t*)d_a, (fr_t*)d_b, (size_t)1, (const fr_t*)d_w, nw, (fr_t*)d_w_out, z, z, (const uint32_t*)c->d_gmail, seq);
  HIPCHK(c, hipGetLastError());
  const MsmColMap id = {0, 0, 0, 0};
  const lasso_fr* heads[2] = {d_a, d_b};
  rc = run_msm_direct(c, (const uint8_t*)d_w_out, row * 32, 1, row, id, b, (uint8_t*)c->d_scratch, nullptr, 2, scale, tail, 1, 0, heads, &seq); if (rc) return rc;
  c->ahead_active = true; c->ahead_bullet = true; c->ahead_seq = seq; c->ahead_tagged = true; c->ahead_count = sizeof(ed_point) / sizeof(fr_t) + 2;
  return 0;
}
int32_t lasso_bullet_post(lasso_ctx* c, const lasso_fr* u, const lasso_fr* u_inv) {
  REQUIRE(c, u && u_inv && c->ahead_active && c->ahead_bullet && !c->pending);
  mail_chunks(c->mail_h + 12, c->ahead_seq, (const uint32_t*)u_inv);
  post_mail(c, c->ahead_seq, (const uint32_t*)u);
  c->ahead_active = false;
  c->pending = true; c->pending_seq = c->ahead_seq; c->pending_count = c->ahead_count; c->pending_tagged = c->ahead_tagged; c->pending_groups = 1; c->pending_K = 0;
  return 0;
}
int32_t lasso_bullet_fold(lasso_ctx* c, lasso_fr* d_a, lasso_fr* d_b, size_t nk, const lasso_fr* d_w, size_t nw, lasso_fr* d_w_out, const lasso_fr* u, const lasso_fr* u_inv) {
  REQUIRE(c, d_a && d_b && d_w && d_w_out && u && u_inv && nk >= 2 && (nk & (nk - 1)) == 0 && nw >= 1 && d_w != d_w_out);
  const size_t half = nk / 2;
  ProfScope ps(c, LASSO_K_MISC, 96.0 * nk + 96.0 * nw);
  hipLaunchKernelGGL(k_bullet_fold, dim3(grid_for(half > nw ? half : nw)), dim3(256), 0, c->stream, (fr_t*)d_a, (fr_t*)d_b, half, (const fr_t*)d_w, nw, (fr_t*)d_w_out, to_fr(u), to_fr(u_inv));
  HIPCHK(c, hipGetLastError()); return 0;
}

}  // extern "C"

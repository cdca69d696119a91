// Host side of the PairHMM path: context, wire-format parsing, job partitioning, launches, results.
// Boundary it stands in for: compute_fpga (pairhmm/host/PairHMMFpga.cpp:125-162), the Blaze task's
// prepare()/compute() (pairhmm/task/xlnx/PairHMMTask.cpp:27-143) and the pair loop + post-process of
// FalconPairHMM::computePairhmmAVX (pairhmm/xlnx/host/FalconPairHMM.cpp:69-95).
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <sched.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <numeric>
#include <condition_variable>
#include <thread>
#include "accg_internal.h"

using namespace accg;

namespace accg {
static thread_local std::string g_hip_err;
void set_hip_error(hipError_t e, const char* what) {
  g_hip_err = std::string(what) + ": " + hipGetErrorString(e);
}
void set_error_text(const char* text) { g_hip_err = text; }
}  // namespace accg

extern "C" const char* accg_last_hip_error(void) { return g_hip_err.c_str(); }

extern "C" const char* accg_strerror(int st) {
  switch (st) {
    case ACCG_OK: return "ok";
    case ACCG_ERR_NO_DEVICE: return "no gfx950 HIP device";
    case ACCG_ERR_NOT_INITIALISED: return "context not initialised";
    case ACCG_ERR_BAD_ARG: return "bad argument";
    case ACCG_ERR_BAD_WIRE: return "malformed serialized reads/haps";
    case ACCG_ERR_BAD_BASE: return "base other than A,C,G,T,N";
    case ACCG_ERR_EMPTY_SEQ: return "zero-length read or haplotype";
    case ACCG_ERR_TOO_LONG: return "sequence longer than the kernel supports";
    case ACCG_ERR_HIP: return "HIP runtime error";
    case ACCG_ERR_NOMEM: return "out of memory";
    case ACCG_ERR_RCCL: return "RCCL call failed";
    case ACCG_ERR_NO_RCCL: return "librccl could not be loaded";
  }
  return "unknown status";
}

// ---- context ------------------------------------------------------------------------------------
// libomp only (the symbol is weak: a host built against another OpenMP runtime links and simply keeps its setting); per calling thread;
// KMP_BLOCKTIME in the environment wins, ACCG_KEEP_OMP_BLOCKTIME=1 leaves the host application's setting alone
#pragma weak kmp_set_blocktime
namespace accg {
void omp_short_blocktime() {
  static const bool keep = getenv("KMP_BLOCKTIME") != nullptr || (getenv("ACCG_KEEP_OMP_BLOCKTIME") && getenv("ACCG_KEEP_OMP_BLOCKTIME")[0] == '1');
  if (!keep && &kmp_set_blocktime != nullptr) kmp_set_blocktime(1);
}
}  // namespace accg
extern "C" int accg_init(int device, accg_ctx** out) {
  if (!out) return ACCG_ERR_BAD_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return ACCG_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  ACCG_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ACCG_ERR_NO_DEVICE;   // code objects are gfx950 only
  ACCG_HIP(hipSetDevice(device));
  // The host loops' OpenMP threads go to sleep a millisecond after a parallel region instead of libomp's 200 ms of spinning: under a
  // CPU quota (a container's share of a large host) spinners eat the quota the working threads need (a threaded ring next to a
  // 16-thread team measured 30 ms for a stream that takes 7).  KMP_BLOCKTIME in the environment wins.
  omp_short_blocktime();
  // a failure below hands the half-built context to accg_shutdown (streams, events, tables made so far are released)
  struct Guard { accg_ctx* c; ~Guard() { if (c) accg_shutdown(c); } } guard{new accg_ctx};
  accg_ctx* c = guard.c;
  c->device = device;
  c->n_cu = prop.multiProcessorCount;
  snprintf(c->name, sizeof c->name, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, c->n_cu);
  { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) c->wall_khz = khz; }
  if (const char* e = getenv("ACCG_COPY_KERNEL_MAX")) c->kernel_copy_max = (size_t)strtoull(e, nullptr, 10);
  ACCG_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  ACCG_HIP(hipEventCreate(&c->ev0));
  ACCG_HIP(hipEventCreate(&c->ev1));
  // (the forked streams and the tail stream are made when a batch first needs them -- ctx_need_aux / ctx_need_tail: a context that only
  // ever sees regions one at a time runs on ONE stream, and sixteen such contexts do not crowd the device's few hardware queues)
  const HostTables& t = host_tables();
  const size_t nf = 128 * 3 + 8256, bytes = nf * sizeof(float) + nf * sizeof(double);
  ACCG_HIP(hipMalloc(&c->tab_mem, bytes));
  double* d = (double*)c->tab_mem;
  float* f = (float*)(d + nf);
  auto up = [&](void* dst, const void* src, size_t b) { return hipMemcpy(dst, src, b, hipMemcpyHostToDevice); };
  ACCG_HIP(up(d, t.ph_d, 128 * 8)); ACCG_HIP(up(d + 128, t.omph_d, 128 * 8)); ACCG_HIP(up(d + 256, t.phd3_d, 128 * 8));
  ACCG_HIP(up(d + 384, t.m2m_d, 8256 * 8));
  ACCG_HIP(up(f, t.ph_f, 128 * 4)); ACCG_HIP(up(f + 128, t.omph_f, 128 * 4)); ACCG_HIP(up(f + 256, t.phd3_f, 128 * 4));
  ACCG_HIP(up(f + 384, t.m2m_f, 8256 * 4));
  c->tab_d = {d, d + 128, d + 256, d + 384, t.init_d};
  c->tab_f = {f, f + 128, f + 256, f + 384, t.init_f};
  *out = c;
  guard.c = nullptr;
  return ACCG_OK;
}

extern "C" void accg_shutdown(accg_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  for (int i = 0; i < accg_ctx::N_AUX; i++) {
    if (c->aux[i]) { hipStreamSynchronize(c->aux[i]); hipStreamDestroy(c->aux[i]); }
    if (c->ev_join[i]) hipEventDestroy(c->ev_join[i]);
  }
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  for (int i = 0; i < accg_ctx::N_AUX; i++) {
    if (c->aux_t[i]) { hipStreamSynchronize(c->aux_t[i]); hipStreamDestroy(c->aux_t[i]); }
    if (c->ev_join_t[i]) hipEventDestroy(c->ev_join_t[i]);
  }
  if (c->tail) { hipStreamSynchronize(c->tail); hipStreamDestroy(c->tail); }
  if (c->ev_fork_t) hipEventDestroy(c->ev_fork_t);
  c->pool.drain();
  if (c->h_stage) hipHostFree(c->h_stage);
  if (c->h_flags) hipHostFree(c->h_flags);
  if (c->tab_mem) hipFree(c->tab_mem);
  delete c;
}
namespace accg {
namespace {
constexpr size_t POOL_MAX_BLOCK = 256ull << 20, POOL_MAX_CACHED = 4ull << 30;
size_t pool_round(size_t b) {
  if (b <= 256) return 256;
  if (b <= (1u << 20)) { size_t r = 256; while (r < b) r <<= 1; return r; }
  // above 1 MiB: quarter-octave steps (at most 25 % over).  Batches of a stream differ by a few per cent in size; with 1 MiB steps every
  // new largest one missed the cache, and a hipMalloc under load waits for the device to go idle (measured: 5 to 10 ms, and it
  // holds up every other thread's allocation meanwhile).
  size_t p2 = (size_t)1 << 20;
  while (p2 * 2 < b) p2 <<= 1;                 // p2 < b <= 2 p2
  const size_t q = p2 / 4;
  return p2 + (b - p2 + q - 1) / q * q;
}
}  // namespace
hipError_t DevPool::get(size_t bytes, void** p) {
  const size_t cap = pool_round(bytes);
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = free_.lower_bound(cap);
    if (it != free_.end() && it->first <= cap * 2) {
      *p = it->second; live_[*p] = it->first; cached -= it->first; free_.erase(it);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, cap);
  if (e != hipSuccess) {               // give the cache back to the driver and try once more
    drain();
    e = hipMalloc(p, cap);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> g(mu);
  live_[*p] = cap;
  return hipSuccess;
}
void DevPool::put(void* p) {
  if (!p) return;
  size_t cap = 0;
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = live_.find(p);
    if (it == live_.end()) return;
    cap = it->second; live_.erase(it);
    if (cap <= POOL_MAX_BLOCK && cached + cap <= POOL_MAX_CACHED) { free_.emplace(cap, p); cached += cap; return; }
  }
  hipFree(p);
}
void DevPool::drain() {
  std::multimap<size_t, void*> f;
  { std::lock_guard<std::mutex> g(mu); f.swap(free_); cached = 0; }
  for (auto& kv : f) hipFree(kv.second);
}
thread_local int tls_host_threads = 0;      // a ring worker's share of the host threads (accg_phmm_ring_create_threaded)
int host_threads() {
  if (tls_host_threads > 0) return tls_host_threads;
  static const int n = [] {
    if (const char* e = getenv("ACCG_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return v; }
    int cpus = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = std::max(1, CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32]; long per = 0;
      if (fscanf(f, "%31s %ld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) cpus = std::max(1, std::min(cpus, (int)(atol(q) / per)));
      fclose(f);
    }
    return std::min(cpus, 64);
  }();
  return n;
}
hipError_t ctx_stage(accg_ctx* c, size_t bytes, void** p) {
  if (bytes > c->h_stage_bytes) {
    // (a copy kernel queued on the context's stream may still be reading the block that is about to go)
    if (c->h_stage) { hipError_t es = hipStreamSynchronize(c->stream); if (es != hipSuccess) return es; hipHostFree(c->h_stage); }
    c->h_stage = nullptr; c->h_stage_bytes = 0;
    const size_t cap = std::max<size_t>((bytes + (1u << 20) - 1) >> 20 << 20, 1u << 20);
    hipError_t e = hipHostMalloc(&c->h_stage, cap, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    c->h_stage_bytes = cap;
  }
  *p = c->h_stage;
  return hipSuccess;
}
hipError_t ctx_need_aux(accg_ctx* c) {
  if (c->ev_fork) return hipSuccess;
  hipError_t e = hipSuccess;
  for (int i = 0; i < accg_ctx::N_AUX && e == hipSuccess; i++) {
    e = hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
  return e;
}
hipError_t ctx_need_tail(accg_ctx* c) {
  if (c->ev_fork_t) return hipSuccess;
  hipError_t e = hipStreamCreateWithFlags(&c->tail, hipStreamNonBlocking);
  for (int i = 0; i < accg_ctx::N_AUX && e == hipSuccess; i++) {
    e = hipStreamCreateWithFlags(&c->aux_t[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join_t[i], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork_t, hipEventDisableTiming);
  return e;
}
hipError_t ctx_fork(accg_ctx* c) {
  hipError_t e = ctx_need_aux(c);
  if (e != hipSuccess) return e;
  e = hipEventRecord(c->ev_fork, c->stream);
  for (int i = 0; i < accg_ctx::N_AUX && e == hipSuccess; i++) e = hipStreamWaitEvent(c->aux[i], c->ev_fork, 0);
  return e;
}
hipError_t ctx_join(accg_ctx* c) {
  hipError_t e = hipSuccess;
  for (int i = 0; i < accg_ctx::N_AUX && e == hipSuccess; i++) {
    e = hipEventRecord(c->ev_join[i], c->aux[i]);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join[i], 0);
  }
  return e;
}
hipError_t ctx_fork_tail(accg_ctx* c) {
  hipError_t e = ctx_need_tail(c);
  if (e != hipSuccess) return e;
  e = hipEventRecord(c->ev_fork_t, c->tail);
  for (int i = 0; i < accg_ctx::N_AUX && e == hipSuccess; i++) e = hipStreamWaitEvent(c->aux_t[i], c->ev_fork_t, 0);
  return e;
}
hipError_t ctx_join_tail(accg_ctx* c) {
  hipError_t e = hipSuccess;
  for (int i = 0; i < accg_ctx::N_AUX && e == hipSuccess; i++) {
    e = hipEventRecord(c->ev_join_t[i], c->aux_t[i]);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->tail, c->ev_join_t[i], 0);
  }
  return e;
}
}  // namespace accg
extern "C" void* accg_stream(accg_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int accg_device_name(accg_ctx* c, char* buf, size_t n) {
  if (!c || !buf || !n) return ACCG_ERR_BAD_ARG;
  snprintf(buf, n, "%s", c->name);
  return ACCG_OK;
}
extern "C" void accg_counters_pack(const accg_counters* c, uint64_t out[4]) {
  out[0] = c->cells; out[1] = c->pairs; out[2] = c->kernel_ns; out[3] = c->rescued;
}
// Context<T> tables as the device sees them (tests pin them to the golden table fixture).
extern "C" void accg_phmm_tables_f32(float* ph128, float* m2m8256, float* init, float* log10_init) {
  const HostTables& t = host_tables();
  memcpy(ph128, t.ph_f, sizeof t.ph_f); memcpy(m2m8256, t.m2m_f, sizeof t.m2m_f); *init = t.init_f; *log10_init = t.log10_init_f;
}
extern "C" void accg_phmm_tables_f64(double* ph128, double* m2m8256, double* init, double* log10_init) {
  const HostTables& t = host_tables();
  memcpy(ph128, t.ph_d, sizeof t.ph_d); memcpy(m2m8256, t.m2m_d, sizeof t.m2m_d); *init = t.init_d; *log10_init = t.log10_init_d;
}

// ---- batch --------------------------------------------------------------------------------------
namespace {

struct Region { uint32_t read0, n_reads, hap0, n_haps; uint64_t out0; };
// form: 7, 6 or 5 operations per cell; wg = 2: the work items come in pairs (same reads, two runs of haplotypes; the second may be
// empty) that the fast kernel runs as one workgroup of two wavefronts sharing the dist table
// 0: a class with a launch of its own; else (window of K, workgroup size) of the merged launch it can join
inline int merge_class(int K, int lpp, int form, bool striped, int wg) {
  if (striped || form != 5 || lpp * K <= 16 || (lpp != 8 && lpp != 16) || (wg != 1 && wg != 2)) return 0;
  const int w = phmm_multi_window(K);
  return w ? w * 4 + (wg & 3) : 0;
}
// k_hi != 0: a merged launch of the fast mode, the classes K..k_hi of both lane counts in one grid (phmm_launch_f32_multi)
// (then K = smallest K, lpp = the lane count with the larger LDS request, wpc_min = the smallest occupancy of the classes in it)
struct KLaunch { int K, lpp, form; bool striped; int wg; uint32_t work0, n_work; int stream_cap, haps_cap; uint64_t cost = 0; int aux = 0; int k_hi = 0; size_t lds = 0; int wpc_min = 0; };

thread_local DevPool* tls_pool = nullptr;   // set for the duration of accg_phmm_batch_create
struct PoolScope { DevPool* prev; explicit PoolScope(DevPool* p) : prev(tls_pool) { tls_pool = p; } ~PoolScope() { tls_pool = prev; } };

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevPool* pool = nullptr;
  bool view = false;             // points into the batch's arena; nothing to free
  void place(void* base, size_t byte_off, size_t count) { p = (T*)((uint8_t*)base + byte_off); n = count; view = true; }
  int alloc(size_t count) {
    n = count;
    if (count == 0) return ACCG_OK;
    pool = tls_pool;
    if (pool) ACCG_HIP(pool->get(count * sizeof(T), (void**)&p));
    else ACCG_HIP(hipMalloc((void**)&p, count * sizeof(T)));
    return ACCG_OK;
  }
  int upload(const std::vector<T>& v, hipStream_t s) {
    int st = alloc(v.size());
    if (st != ACCG_OK || v.empty()) return st;
    ACCG_HIP(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return ACCG_OK;
  }
  void release() { if (p && !view) { if (pool) pool->put(p); else hipFree(p); } p = nullptr; n = 0; }
  ~DevBuf() { release(); }      // error paths of batch_create free what was allocated so far
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

bool valid_base_lut(uint8_t b) { return b == 'A' || b == 'C' || b == 'G' || b == 'T' || b == 'N'; }
// 1 if some byte of p[0, n) is not one of ACGTN (a table lookup per byte, four independent chains: the five comparisons per byte of
// valid_base_lut were a third of a batch's parse time)
int any_invalid_base(const uint8_t* p, size_t n) {
  struct Lut { uint8_t bad[256]; Lut() { for (int i = 0; i < 256; i++) bad[i] = !valid_base_lut((uint8_t)i); } };
  static const Lut lut;
  unsigned b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  size_t k = 0;
  for (; k + 4 <= n; k += 4) { b0 |= lut.bad[p[k]]; b1 |= lut.bad[p[k + 1]]; b2 |= lut.bad[p[k + 2]]; b3 |= lut.bad[p[k + 3]]; }
  for (; k < n; k++) b0 |= lut.bad[p[k]];
  return (int)((b0 | b1 | b2 | b3) != 0);
}

}  // namespace

struct accg_phmm_batch {
  accg_ctx* ctx = nullptr;
  std::vector<Region> regions;
  std::vector<SeqRef> rd, hp;
  std::vector<uint32_t> rd_out, hp_local, hap_ids;
  std::vector<uint32_t> rd_row0; // per read: index of its first record in the per-row records of the five-operation sweep
  std::vector<uint32_t> rd_shape; // per read: K | LPP << 8 of the wavefront it runs in, 0 if that one does not run the five-operation sweep
  std::vector<const uint8_t*> hp_ptr;   // per haplotype: its bases in the caller's blob (valid during accg_phmm_batch_create only)
  std::vector<uint8_t> streams;  // the haplotype streams of all runs, laid out for the kernels that copy theirs in (phmm_dev.h: PHMM_STREAM_TAIL)
  std::vector<uint32_t> chunk_stream16, chunk_stream_len;   // per run (index into chunks_dev): offset in 16-byte units / length
  uint64_t n_rows = 0;           // read bases in all = per-row records
  bool redo_possible = false;    // some read's fp64 result may land next to the denormal range (parse_reads: `deep`): the strict re-run launches are needed
  bool all_form5 = true;         // every read passes the five-operation form's range tests: the fp64 rescue pass may use it too
  bool any_form5 = false;        // some launch runs the five-operation sweep: phmm_prepare_rows runs at the start of a fast pass
  std::vector<uint8_t> rd_form;  // per read: the cheapest form of the fast sweep it passes the range tests of: 5, 6 or 7 (phmm_dev.h)
  std::vector<PhmmWork> work;
  std::vector<KLaunch> launches;        // one per (lanes, K, form, workgroup size) class, in work-list order
  std::vector<KLaunch> launches_fast;   // what the fast mode launches: the same list with the mergeable classes merged
  uint64_t pairs = 0, cells = 0, algo_bytes = 0;
  bool has_n = false;          // some haplotype contains an 'N': the dist table needs its fifth slab
  int force_wpc = 0;           // ACCG_PHMM_WPC: > 0 pins every launch to that many wavefronts per CU, < 0 pins nothing
  DevBuf<uint8_t> d_arena;       // one allocation behind every buffer below
  uint8_t* res_ptr = nullptr;    // the results block [n_rescued u64][out f32 x (pairs+1)] of the pass buffers in use
  // Pass buffers (d_out, d_out64, d_state, d_rescue_jobs, d_redo, d_flagged, res_ptr): everything a pass writes.  A batch that is run
  // again gets a SECOND set (d_alt, allocated at its second run) and alternates between the two, so that a pass's sweep can start
  // while the tail of the pass before -- planner, fp64 rescue, re-runs, on the context's tail stream -- is still at work on the
  // other set (run_direct).  `alt` holds the set not in use; tail_done of a set = its last tail has been queued up to there.
  struct PassSet { float* out = nullptr; double* out64 = nullptr; uint32_t *state = nullptr, *redo = nullptr, *flagged = nullptr; PhmmWork* jobs = nullptr;
                   uint8_t* res = nullptr; hipEvent_t tail_done = nullptr; bool tail_pending = false; };
  PassSet alt;
  hipEvent_t tail_done = nullptr, ev_sweep = nullptr;
  bool tail_pending = false;     // the set in use: a tail has been queued on the tail stream and not yet been joined into the main stream
  DevBuf<uint8_t> d_alt;
  size_t pass_bytes = 0, off_alt_out64 = 0, off_alt_state = 0, off_alt_out = 0, off_alt_jobs = 0, off_alt_redo = 0, off_alt_flagged = 0, alt_clear_bytes = 0;
  uint64_t runs = 0;
  DevBuf<uint8_t> d_rblob, d_hblob;
  DevBuf<SeqRef> d_rd, d_hp;
  DevBuf<uint32_t> d_rd_out, d_hp_local;
  DevBuf<PhmmHapDesc> d_hap_desc;   // hap_ids with the haplotype's descriptor next to each id
  DevBuf<PhmmWork> d_work;
  DevBuf<uint32_t> d_rd_row0, d_rd_shape;
  DevBuf<uint64_t> d_clock;      // {shader-clock ticks, wall-clock ticks} of the first wavefront of the last sweep launch
  DevBuf<uint8_t> d_streams;
  DevBuf<float4> d_rec_coef, d_rec_dist, d_rec_misc;
  DevBuf<float> d_out;
  DevBuf<double> d_out64;
  // device-side rescue planning
  std::vector<PhmmRegionDev> regions_dev;
  std::vector<PhmmChunkDev> chunks_dev;
  std::vector<uint32_t> sorted_reads;
  uint64_t rescue_bound[PHMM_RESCUE_CLASSES] = {0};   // host-side upper bound of rescue jobs per class (64 bits: checked before narrowing)
  uint32_t rescue_off[PHMM_RESCUE_CLASSES + 1] = {0}; // class c's job array starts at rescue_off[c] (prefix sums of the bounds)
  int rescue_stream_cap = 0, rescue_haps_cap = 0;
  DevBuf<PhmmRegionDev> d_regions;
  DevBuf<PhmmChunkDev> d_chunks;
  DevBuf<uint32_t> d_sorted_reads, d_flagged;
  DevBuf<uint32_t> d_state;   // zeroed per run: [n_reads] read flags, rescue job counts per class, [2] n_rescued (u64)
  DevBuf<PhmmWork> d_rescue_jobs;
  DevBuf<uint32_t> d_redo;       // per rescue class (at rescue_off[c]): indices of the jobs to re-run in the strict form
  uint64_t last_kernel_ns = 0;
  // One pass = memset + a launch per (lanes, K) class on forked streams + rescue plan + the rescue classes and their strict
  // re-runs: captured once per arithmetic mode into a graph and replayed (every argument is fixed at batch creation).
  hipGraphExec_t graph_exec[2] = {nullptr, nullptr};
  // Speculative fp64 pass (one-shot batches that leave most of the chip idle, fast mode): jobs for EVERY read, made on the host with the
  // planner's grouping rule, run next to the fp32 sweep on a forked stream; the results kernel then counts what was needed (run_spec)
  bool spec = false, spec_ran = false;                    // spec_ran: the last pass was a speculative one (its results are fetched accordingly)
  std::vector<PhmmWork> spec_jobs;                        // the classes' job arrays back to back, class c at [spec_off[c], spec_off[c + 1])
  uint32_t spec_off[PHMM_RESCUE_CLASSES + 1] = {0};
  std::vector<uint32_t> spec_counts;                      // PHMM_RESCUE_CLASSES words, uploaded (what the planner's counters would hold)
  DevBuf<PhmmWork> d_spec_jobs;
  DevBuf<uint32_t> d_spec_counts;
  // Rescue probe of a batch that is run again and again (a device-resident batch: bench.py, pipelines).  Whether a pass has anything to
  // rescue depends on the batch's bytes and the arithmetic mode only, so ONE complete pass that flagged nothing settles it for all later
  // ones: from its second pass on the planner reports "something flagged" into a host-visible word, an event marks the end of that pass's
  // tail, and once the host sees the event done and the word still 0 it stops queueing the planner and the (empty) fp64 launches for
  // this batch in this mode -- two dependent launches and their gaps per pass (configs[1]: some 20 us of 275).
  std::vector<hipEvent_t> step_ev;   // accg_phmm_batch_steps_*: [0], [1] around all passes, then a pair per pass around its sweep launches
  int steps_queued = 0;
  int rescue_known[2] = {0, 0};      // per mode: 0 not known yet, 1 never rescues, 2 rescues
  int probe_runs[2] = {0, 0};
  hipEvent_t ev_probe[2] = {nullptr, nullptr};
  uint32_t* probe_flag[2] = {nullptr, nullptr};
  bool probe_armed[2] = {false, false};
  bool timed_by_events = false;  // a ring ticket: the context's ev0 / ev1 bracket its pass (results_finish takes the device time from them)
  bool kernel_copies = false;    // a small batch: upload and results travel by copy kernels on the stream (accg_ctx::kernel_copy_max)
  bool results_late = false;     // ring: the downloads are queued by results_finish, not behind the kernels
  bool results_fetched = false;  // the raw results (and, for kernel copies, the fp64 values) of the last pass are in the staging block
  bool graph_off = false;        // capture failed once, or ACCG_PHMM_GRAPH=0: plain stream launches
};

namespace {

// Wire format (pairhmm/interface/PairHMMHostInterface.cpp:175-206): returns number of records, fills refs
// with offsets relative to `base_off` (position of this blob inside the concatenated device blob).
// The range tests of phmm_dev.h on one read (_i at p + 2 len, _d at p + 3 len, _c at p + 4 len): 5 = five-operation form, 6 = six, 7 = seven.
int phmm_read_form(const uint8_t* p, int len) {
  const HostTables& t = host_tables();
  const uint8_t *qi = p + 2 * (size_t)len, *qd = p + 3 * (size_t)len, *qc = p + 4 * (size_t)len;
  // PHMM_X6_MAX_F: Xs = X / pMX stays within F x max(M), F[r] = 1 + c[r] F[r - 1], c[r] = ph[qc[r]] ph[qi[r - 1]] / ph[qi[r]] =
  // 10^-((qc[r] + qi[r - 1] - qi[r]) / 10).  When that exponent is at least 0.1 everywhere, every c is at most 0.7944 whatever the
  // tables' last bits and F stays below 4.9: no need for the sum itself (the common case: a byte loop the compiler vectorises, where
  // the sum is a chain of dependent divisions -- half of a batch's parse time).
  int slack = 1;
  for (int r = 1; r < len; r++) slack &= (int)((qc[r] & 127) + (qi[r - 1] & 127) >= (qi[r] & 127) + 1);
  if (!slack) {
    double F = 1.0;
    for (int r = 1; r < len; r++) {
      const double c = (double)t.ph_f[qc[r] & 127] * (double)t.ph_f[qi[r - 1] & 127] / (double)t.ph_f[qi[r] & 127];
      F = 1.0 + c * F;
      if (!(F <= (double)PHMM_X6_MAX_F)) return 7;
    }
  }
  // PHMM_X5_*: Ys = Y / pMY and the term / pMM -- the two comparisons on the tables' own floats, tabulated once per quality (pair)
  // ... and, per quality (pair), whether the comparison also holds for EVERY quality (pair) at least as large: a read whose smallest
  // qc and smallest (qi, qd) pass that test passes the per-base test at every base, and three byte minima (loops the compiler
  // vectorises) replace a table lookup per base -- the common case, half of what was left of a region's parse time
  struct Ok { uint8_t c[128], mm[128 * 128], c_from[128], mm_from[128 * 128]; };
  static const Ok ok = [&] {
    Ok o;
    for (int q = 0; q < 128; q++) o.c[q] = t.ph_f[q] <= PHMM_X5_MAX_YY;
    for (int i = 0; i < 128; i++)
      for (int d = 0; d < 128; d++) { const int lo = i < d ? i : d, hi = i < d ? d : i; o.mm[i * 128 + d] = t.m2m_f[((hi * (hi + 1)) >> 1) + lo] >= PHMM_X5_MIN_MM; }
    for (int q = 127; q >= 0; q--) o.c_from[q] = o.c[q] & (q < 127 ? o.c_from[q + 1] : 1);
    for (int i = 127; i >= 0; i--)
      for (int d = 127; d >= 0; d--)
        o.mm_from[i * 128 + d] = o.mm[i * 128 + d] & (i < 127 ? o.mm_from[(i + 1) * 128 + d] : 1) & (d < 127 ? o.mm_from[i * 128 + d + 1] : 1);
    return o;
  }();
  uint8_t min_c = 127, min_i = 127, min_d = 127;
  for (int r = 0; r < len; r++) { const uint8_t c = qc[r] & 127, i = qi[r] & 127, d = qd[r] & 127; min_c = c < min_c ? c : min_c; min_i = i < min_i ? i : min_i; min_d = d < min_d ? d : min_d; }
  if (ok.c_from[min_c] & ok.mm_from[min_i * 128 + min_d]) return 5;
  int all = 1;
  for (int r = 0; r < len; r++) all &= ok.c[qc[r] & 127] & ok.mm[(qi[r] & 127) * 128 + (qd[r] & 127)];
  return all ? 5 : 6;
}

// `deep` (out, or-ed): some read's likelihood may come out within 36 decades of the smallest normal double in the fp64 rescue -- the
// only case in which the fast mode's strict re-run launch (phmm_redo_multi) can have anything to do.  For every other batch the launch
// is skipped, by a bound the host can check: against ANY haplotype the forward sum is at least the path "first base in M, all others
// inserted", summed over the start columns: INIT x dist_min(q[0]) x (1 - ph[qc[0]]) x ph[qi[1]] x prod_{r >= 2} ph[qc[r]] -- i.e.
// log10(result x 2^1020) >= 307.05 - 1.17 - (q[0] + qi[1] + sum_{r >= 2} qc[r]) / 10 when qc[0] >= 1 (tools/check_floor_bound.py holds
// it against the oracle).  A sum of at most 5500 leaves the result above 1e-244; the kernels list a job at 1e-280 (PHMM_F64_TINY).
constexpr uint32_t PHMM_REDO_Q_SUM_MAX = 5500;
int parse_reads(const uint8_t* p, size_t bytes, uint32_t base_off, std::vector<SeqRef>& refs, std::vector<uint8_t>& form, bool& deep) {
  if (bytes < 4) return ACCG_ERR_BAD_WIRE;
  int32_t n; memcpy(&n, p, 4);
  if (n < 0) return ACCG_ERR_BAD_WIRE;
  size_t pos = 4;
  const size_t first = refs.size();
  for (int i = 0; i < n; i++) {                  // the records' lengths, in order (each one's position follows from the ones in front of it)
    if (pos + 4 > bytes) return ACCG_ERR_BAD_WIRE;
    int32_t len; memcpy(&len, p + pos, 4); pos += 4;
    if (len < 0 || pos + 5 * (size_t)len > bytes) return ACCG_ERR_BAD_WIRE;
    if (len == 0) return ACCG_ERR_EMPTY_SEQ;
    if (len > ACCG_PHMM_MAX_READ) return ACCG_ERR_TOO_LONG;
    refs.push_back({base_off + (uint32_t)pos, (uint32_t)len});
    pos += 5 * (size_t)len;
  }
  // base validation and the range tests of the sweep's forms, read by read: independent, on the host's threads for a large region
  // (inside the per-region parallel loop of a multi-region batch this stays serial: no nested teams)
  form.resize(first + (size_t)n);
  int bad = 0, dp = 0;
#pragma omp parallel for schedule(static) num_threads(accg::host_threads()) reduction(| : bad, dp) if (n >= 512 && accg::host_threads() > 1)
  for (int i = 0; i < n; i++) {
    const SeqRef& r = refs[first + (size_t)i];
    const uint8_t* q = p + (r.off - base_off);
    bad |= any_invalid_base(q, r.len);
    form[first + (size_t)i] = (uint8_t)phmm_read_form(q, (int)r.len);
    const size_t L = r.len;
    const uint8_t *bq = q + L, *qi = q + 2 * L, *qc = q + 4 * L;
    uint32_t tq = (uint32_t)(bq[0] & 127) + (L >= 2 ? (uint32_t)(qi[1] & 127) : 0u);
    for (size_t k = 2; k < L; k++) tq += (uint32_t)(qc[k] & 127);
    dp |= (int)((qc[0] & 127) == 0 || tq > PHMM_REDO_Q_SUM_MAX);
  }
  if (bad) return ACCG_ERR_BAD_BASE;
  if (dp) deep = true;
  return n;
}
int parse_haps(const uint8_t* p, size_t bytes, uint32_t base_off, std::vector<SeqRef>& refs, bool& has_n, std::vector<const uint8_t*>& ptrs) {
  if (bytes < 4) return ACCG_ERR_BAD_WIRE;
  int32_t n; memcpy(&n, p, 4);
  if (n < 0) return ACCG_ERR_BAD_WIRE;
  size_t pos = 4;
  for (int i = 0; i < n; i++) {
    if (pos + 4 > bytes) return ACCG_ERR_BAD_WIRE;
    int32_t len; memcpy(&len, p + pos, 4); pos += 4;
    if (len < 0 || pos + (size_t)len > bytes) return ACCG_ERR_BAD_WIRE;
    if (len == 0) return ACCG_ERR_EMPTY_SEQ;
    if (len > ACCG_PHMM_MAX_HAP) return ACCG_ERR_TOO_LONG;
    for (int k = 0; k < len; k++) {
      if (!valid_base_lut(p[pos + k])) return ACCG_ERR_BAD_BASE;
      has_n |= p[pos + k] == 'N';
    }
    refs.push_back({base_off + (uint32_t)pos, (uint32_t)len});
    ptrs.push_back(p + pos);
    pos += (size_t)len;
  }
  return n;
}


// Resident wavefronts per CU for the fp32 kernel at K rows per lane: LDS (160 KiB, granted in 512-byte units) and VGPR (512
// per SIMD lane, in units of 8) limits.  Registers as the build reports them (kernel-resource-usage): 9 K + 23 for the fast
// kernel (column in assembly), 13 K + 23 for the strict one (compiler-scheduled).
// (form 5, the five-operation column: 8 K + 24; wg = wavefronts per workgroup sharing one dist table)
int waves_per_cu(int K, int nchar, int stream_cap, int haps_cap, int lpp, bool strict = false, int form = 6, int wg = 1) {
  const size_t lds = (phmm_lds_bytes(K, 4, nchar, stream_cap, haps_cap, lpp, !strict, false, wg) + 511) / 512 * 512;
  const int by_lds = (int)((160 * 1024) / lds) * wg;
  const int vgpr = ((strict ? 13 * K + 23 : form == 5 ? 8 * K + 24 : 9 * K + 23) + 7) / 8 * 8;
  const int by_vgpr = std::min(8, 512 / vgpr) * 4;
  return std::max(1, std::min(std::min(by_lds, by_vgpr), 32));
}
// Pairs pay where one wavefront per workgroup leaves a CU below sixteen wavefronts for want of LDS and two reach them
bool pairs_pay(int K, int nchar, int stream_cap, int haps_cap, int lpp) {
  static const bool off = [] { const char* e = getenv("ACCG_PHMM_WG"); return e && e[0] == '1'; }();     // A/B knob: 1 = never pair
  return !off && waves_per_cu(K, nchar, stream_cap, haps_cap, lpp, false, 5, 1) < 16 && waves_per_cu(K, nchar, stream_cap, haps_cap, lpp, false, 5, 2) >= 16;
}
// The occupancy a launch is pinned to: 8, 16 or 32 wavefronts per CU, i.e. 2, 4 or 8 per SIMD on every SIMD (see partition()).
int pinned_wpc(int natural) { return natural >= 32 ? 32 : natural >= 16 ? 16 : natural >= 8 ? 8 : natural; }

struct Chunk { uint32_t ids0, n; uint32_t stream_len; };

// A pass's results in the context's pinned staging: [device ticks u64][n_rescued u64][raw f32 x pairs][pad to 8][fp64 x pairs]
constexpr size_t RES_HDR = 2 * sizeof(unsigned long long);
size_t results_off64(uint64_t pairs) { return (RES_HDR + pairs * sizeof(float) + 7) / 8 * 8; }
size_t results_stage_bytes(uint64_t pairs) { return results_off64(pairs) + pairs * sizeof(double) + 64; }

// Greedy runs of haplotypes with at most `budget` stream entries (a single longer haplotype gets a run of its own).
void chunk_region(const accg_phmm_batch& b, const Region& r, uint64_t budget, std::vector<std::pair<uint32_t, uint32_t>>& runs,
                  std::vector<uint32_t>& lens) {
  uint32_t h = 0;
  while (h < r.n_haps) {
    uint32_t h0 = h, n = 0;
    uint64_t len = 1;   // terminal bubble
    while (h < r.n_haps && n < (uint32_t)PHMM_HAPS_MAX) {
      const uint64_t add = b.hp[r.hap0 + h].len + 1;
      if (n > 0 && len + add > budget) break;
      if (len + add > (uint64_t)PHMM_STREAM_MAX) break;   // ACCG_PHMM_MAX_HAP < PHMM_STREAM_MAX, so n == 0 never breaks here
      len += add; n++; h++;
    }
    runs.push_back({h0, n});
    lens.push_back((uint32_t)len);
  }
}

// Cuts every region into jobs = (the reads of one wavefront: up to eight of similar length) x (a run of haplotypes).
// Single-wave workgroups are placed by the hardware dispatcher as slots free up, so the run length is
// chosen to minimise the makespan of a longest-first list schedule on the resident-wave slots of the
// chip (one prologue + 15-step fill per job against the quantisation of jobs over slots); e.g.
// configs[1] comes out as 4096 jobs of 4 haplotypes = exactly one job per slot at 16 waves per CU.
void partition(accg_phmm_batch& b) {
  static const bool trace_p = getenv("ACCG_TRACE") != nullptr;
  const auto tpa = std::chrono::steady_clock::now();
  auto tpb = tpa, tpc = tpa, tpd = tpa;
  const int nchar = b.has_n ? 5 : 4;
  const int n_cu = std::max(b.ctx->n_cu, 1);
  // read groups per region (one group = the reads of one wavefront), by descending read length so that they need the same K
  struct Group { uint32_t read[PHMM_GROUPS]; int K, lpp, form; bool striped; };
  // A/B knobs: ACCG_PHMM_FORM=7|6|5 = the cheapest form allowed (default 5); ACCG_PHMM_X6=0 = the seven-operation form only
  const char* e6 = getenv("ACCG_PHMM_X6");
  const char* ef = getenv("ACCG_PHMM_FORM");
  const int min_form = (e6 && e6[0] == '0') ? 7 : ef ? std::max(5, std::min(7, atoi(ef))) : 5;
  auto form_of = [&](uint32_t rid) { return std::max((int)b.rd_form[rid], min_form); };
  const char* e8 = getenv("ACCG_PHMM_LPP8");                 // A/B knob: largest K run with 8 lanes per read (0 = never)
  // A one-shot batch of a region or two leaves most SIMDs idle and every wavefront runs at its own issue cadence: what counts is the
  // length of a job, and sixteen lanes per read halve the rows per lane (a configs[3] region: sweep 41 -> 30 us).  "Small": even with
  // twice the wavefronts there is at most one per SIMD (reads / 4 wavefronts per haplotype).
  uint64_t jobs16 = 0;
  for (const Region& r : b.regions) jobs16 += (uint64_t)((r.n_reads + 3) / 4) * r.n_haps;
  const bool latency_shapes = b.ctx->oneshot && jobs16 <= 4ull * (uint64_t)std::max(b.ctx->n_cu, 1);
  const int max_k8 = e8 ? atoi(e8) : latency_shapes ? 0 : PHMM_K8_DEFAULT;
  std::vector<std::vector<Group>> groups(b.regions.size());
  b.sorted_reads.assign(b.rd.size(), 0);
  b.regions_dev.assign(b.regions.size(), PhmmRegionDev{0, 0, 0, 0, 0, 0});
  uint64_t kw[4][PHMM_MAX_K + 1] = {{0}};     // haplotype passes per (lanes per read: 8, 16, 32, 64; K)
  uint64_t n_form5 = 0, n_other = 0;          // ... in the five-operation form (whose jobs may go in pairs) / in the others
#pragma omp parallel for schedule(dynamic, 4) num_threads(accg::host_threads()) if (b.regions.size() >= 32 && accg::host_threads() > 1)
  for (size_t ri = 0; ri < b.regions.size(); ri++) {
    const Region& r = b.regions[ri];
    if (r.n_reads == 0 || r.n_haps == 0) continue;
    std::vector<uint32_t> order(r.n_reads);
    std::iota(order.begin(), order.end(), r.read0);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return b.rd[x].len > b.rd[y].len; });
    std::copy(order.begin(), order.end(), b.sorted_reads.begin() + r.read0);
    for (uint32_t i = 0; i < r.n_reads;) {        // the longest read of a wavefront decides lanes per read and K
      Group Q;
      phmm_pick(b.rd[order[i]].len, &Q.lpp, &Q.K, max_k8);
      Q.striped = phmm_striped(b.rd[order[i]].len);            // one read per wavefront either way (64 lanes)
      const uint32_t per = 64 / Q.lpp;
      // reads of at most 15 bases run in the reference's operation order even in fast mode (launch_f32), so they must not share
      // a wavefront with longer ones: a group stops at that boundary
      const bool first_tiny = b.rd[order[i]].len <= 15;
      // ... and a wavefront runs the six- or five-operation form only if all of its reads pass that form's range test: a group stops
      // where the form changes (it takes reads of the first one's kind)
      const int form0 = form_of(order[i]);
      Q.form = Q.striped ? 7 : form0;
      uint32_t take = 0;
      for (uint32_t g = 0; g < PHMM_GROUPS; g++) {
        const bool ok = g < per && i + g < r.n_reads && take == g && (first_tiny || b.rd[order[i + g]].len > 15) &&
                        form_of(order[i + g]) == form0;
        Q.read[g] = ok ? order[i + g] : PHMM_NO_READ;
        take += ok;
      }
      i += take ? take : 1;                                    // (the first read always qualifies for its own group)
      groups[ri].push_back(Q);
    }
  }
  for (size_t ri = 0; ri < b.regions.size(); ri++)
    for (const Group& Q : groups[ri]) {
      kw[Q.lpp == 8 ? 0 : Q.lpp == 16 ? 1 : Q.lpp == 32 ? 2 : 3][Q.K] += (uint64_t)b.regions[ri].n_haps;
      (Q.form == 5 && !Q.striped ? n_form5 : n_other) += (uint64_t)b.regions[ri].n_haps;
    }
  // per-row records of the five-operation sweep: LPP * K per read of a wavefront that runs it (phmm_dev.h: PhmmRowRecs)
  for (const auto& gr : groups)
    for (const Group& Q : gr)
      if (Q.form == 5 && !Q.striped && Q.lpp * Q.K > 16)
        for (int g = 0; g < PHMM_GROUPS; g++)
          if (Q.read[g] != PHMM_NO_READ) b.rd_shape[Q.read[g]] = (uint32_t)Q.K | ((uint32_t)Q.lpp << 8);
  for (size_t i = 0; i < b.rd.size(); i++) {
    b.rd_row0[i] = (uint32_t)b.n_rows;
    b.n_rows += (uint64_t)(b.rd_shape[i] & 255u) * (uint64_t)(b.rd_shape[i] >> 8);
  }
  b.rd_row0[b.rd.size()] = (uint32_t)b.n_rows;
  int K_dom = 1, lpp_dom = 16, n_classes = 0;
  for (int li = 0; li < 4; li++)
    for (int k = 1; k <= PHMM_MAX_K; k++) {
      n_classes += kw[li][k] != 0;
      if (kw[li][k] > kw[lpp_dom == 8 ? 0 : lpp_dom == 16 ? 1 : lpp_dom == 32 ? 2 : 3][K_dom]) { K_dom = k; lpp_dom = 8 << li; }
    }
  n_classes = std::max(n_classes, 1);

  tpb = std::chrono::steady_clock::now();
  // candidate budgets: multiples of the most common haplotype length, plus a geometric ladder
  std::vector<uint64_t> cand;
  if (!b.hp.empty()) {
    std::vector<uint32_t> lens; for (const SeqRef& h : b.hp) lens.push_back(h.len);
    std::nth_element(lens.begin(), lens.begin() + lens.size() / 2, lens.end());
    const uint64_t med = lens[lens.size() / 2] + 1;
    for (uint64_t m = 1; m * med + 1 <= (uint64_t)PHMM_STREAM_MAX && m <= (uint64_t)PHMM_HAPS_MAX; m++) cand.push_back(m * med + 1);
  }
  for (uint64_t L = 256; L <= (uint64_t)PHMM_STREAM_MAX; L = L * 5 / 4) cand.push_back(L);
  cand.push_back(PHMM_STREAM_MAX);
  if (const char* e = getenv("ACCG_PHMM_STREAM_BUDGET")) { cand.clear(); cand.push_back(strtoull(e, nullptr, 10)); }   // tuning knob

  const char* ew = getenv("ACCG_PHMM_WPC");                 // A/B knob: resident wavefronts per CU (8, 16, 32)
  const int force_wpc = ew ? atoi(ew) : 0;
  b.force_wpc = force_wpc;
  const double prologue_steps = 30.0;   // table lookups + dist table + stream build, in units of one sweep step
  uint64_t best_budget = cand[0];
  double best_span = -1;
  bool best_pairs = false;       // the schedule that won was simulated with the dominant class's jobs in pairs
  // Jobs take few distinct costs (chunk length x rows per lane), so the longest-first list schedule is simulated on
  // histograms: job costs in descending order, slot loads as load -> number of slots.  Assigning k equal jobs to the k
  // least-loaded slots of one bucket is what the sequential rule does one job at a time.
  // The jobs of the dominant class go in pairs (one workgroup of two wavefronts sharing the dist table) when that buys occupancy;
  // the schedule is then simulated on pairs and workgroup slots.  Whether it does depends on the stream length, i.e. on the budget:
  // each candidate is simulated singly and, if pairs would pay at its caps, again in pairs.  The candidates are independent of each
  // other: they are evaluated on the host threads this process may use, and the first best one in candidate order is taken.
  const bool dom5 = n_form5 >= n_other;
  struct Eval { double span = -1; bool pays = false; };
  auto evaluate = [&](uint64_t budget, bool sim_pairs) -> Eval {
    std::vector<std::pair<double, uint64_t>> hist;       // (cost, jobs): collected, then sorted by descending cost and merged (a map's
    hist.reserve(b.regions.size() * 8);                  // node per insertion was most of the search's time)
    std::map<double, uint64_t> loads;
    std::vector<std::pair<uint32_t, uint32_t>> runs; std::vector<uint32_t> lens;
    uint32_t cap = 0, hmax = 1;
    uint64_t n_jobs = 0;
    for (size_t ri = 0; ri < b.regions.size(); ri++) {
      if (groups[ri].empty()) continue;
      runs.clear(); lens.clear();
      chunk_region(b, b.regions[ri], budget, runs, lens);
      for (uint32_t len : lens) cap = std::max(cap, len);
      for (const auto& run : runs) hmax = std::max(hmax, run.second);
      int lastK = -1; uint64_t mult = 0;                       // groups are sorted by length: equal K come in runs
      // (pairs: a workgroup of two wavefronts lasts as long as the longer of its two runs)
      auto flush = [&]() {
        if (!mult) return;
        for (size_t c = 0; c < lens.size(); c += sim_pairs ? 2 : 1) {
          const uint32_t len = sim_pairs && c + 1 < lens.size() ? std::max(lens[c], lens[c + 1]) : lens[c];
          hist.emplace_back((len + 15 + prologue_steps) * (7.0 * lastK + 10.0), mult);
        }
      };
      for (const Group& Q : groups[ri]) {
        if (Q.K != lastK) { flush(); lastK = Q.K; mult = 0; }
        mult++;
      }
      flush();
      n_jobs += (uint64_t)groups[ri].size() * (sim_pairs ? (lens.size() + 1) / 2 : lens.size());
    }
    Eval ev;
    if (n_jobs == 0) return ev;
    std::sort(hist.begin(), hist.end(), [](const std::pair<double, uint64_t>& x, const std::pair<double, uint64_t>& y) { return x.first > y.first; });
    {
      size_t w_ = 0;
      for (size_t i = 0; i < hist.size(); i++) {
        if (w_ && hist[w_ - 1].first == hist[i].first) hist[w_ - 1].second += hist[i].second;
        else hist[w_++] = hist[i];
      }
      hist.resize(w_);
    }
    ev.pays = lpp_dom * K_dom > 16 && pairs_pay(K_dom, nchar, (int)((cap + 63) / 64 * 64), (int)hmax, lpp_dom);
    // Resident wavefronts per CU: what registers and LDS allow, or fewer on purpose.  tools/ubench2.hip: the instruction mix of the
    // sweep issues at 1.32 / 1.34 / 1.26 / 1.07 ns per wave-instruction per SIMD with 2 / 3 / 4 / 8 resident wavefronts and at
    // 5 ns for a wavefront that has its SIMD to itself -- a third wavefront per SIMD buys nothing, an odd one per CU unbalances the
    // SIMDs and a lone one at the tail is slow.  So the occupancy is one of 8, 16 or 32 per CU (the launches ask for as much LDS
    // as it takes to get exactly that), and a slot's speed is its SIMD's rate divided by the wavefronts sharing it.
    const int wpc_max = waves_per_cu(K_dom, nchar, (int)((cap + 63) / 64 * 64), (int)hmax, lpp_dom, false, dom5 ? 5 : 6, sim_pairs ? 2 : 1);
    const int wpc = force_wpc > 0 ? std::min(force_wpc, wpc_max) : force_wpc < 0 ? wpc_max : pinned_wpc(wpc_max);
    const int w = std::max(1, wpc / 4);
    const int slots = std::max(1, n_cu * wpc / (sim_pairs ? 2 : 1));
    loads[0.0] = (uint64_t)slots;
    for (const auto& hc : hist) {
      uint64_t left = hc.second;
      while (left) {
        auto lo = loads.begin();
        const uint64_t k = std::min(left, lo->second);
        const double nl = lo->first + hc.first;
        if (k == lo->second) loads.erase(lo); else lo->second -= k;
        loads[nl] += k;
        left -= k;
      }
    }
    // (five-operation column, three VOP3 in five: 1.46 measured at 2 wavefronts per SIMD, VOP3 1.6 / 1.37 / 1.2 / 1.03 and
    // VOP2 1.3 / 1.32 / 1.14 / 1.0 ns at 2 / 3 / 4 / 8)
    const double rate = dom5 ? (w >= 8 ? 1.02 : w >= 4 ? 1.17 : w == 3 ? 1.35 : w == 2 ? 1.46 : 5.0)
                             : (w >= 8 ? 1.07 : w >= 4 ? 1.26 : w == 3 ? 1.34 : w == 2 ? 1.32 : 5.0);
    double span = loads.rbegin()->first * rate * w;
    // the hardware dispatcher is not an ideal list scheduler, and a slot that runs out of jobs early leaves its SIMD partner
    // alone at a quarter of the issue rate: a mild preference for several jobs per slot
    // ... and every (lanes, K) class is a launch of its own with a tail of its own, and the rescue pass inherits the chunking: with
    // the ten classes of a configs[3] mix a 128-region shard ran 18 % faster on jobs a third the size this term used to pick
    // (2.65 against 3.22 ms with the rescue; 1024 regions: the same choice as before)
    span *= 1.0 + 0.05 * (double)n_classes * (double)slots / (double)n_jobs;
    // long streams cost LDS (occupancy of the launches with few rows per lane) and lengthen the tail of every launch: measured
    // on the configs[3] mix +1 % at 2048 entries and +4.5 % at 4096 against 1300 (tools/sweep_c3.sh)
    if (cap > 1300) span *= 1.0 + 0.03 * ((double)cap - 1300.0) / 1024.0;
    ev.span = span;
    return ev;
  };
  // One haplotype per job and every job on a SIMD of its own: nothing to search (the makespan is the longest job, and longer runs
  // only lengthen it).  The blocking call per region lands here.
  {
    uint64_t jobs1 = 0;
    for (size_t ri = 0; ri < b.regions.size(); ri++) jobs1 += (uint64_t)groups[ri].size() * b.regions[ri].n_haps;
    if (jobs1 <= 4ull * (uint64_t)n_cu && !getenv("ACCG_PHMM_STREAM_BUDGET")) { cand.clear(); cand.push_back(1); }
  }
  // The sizing decision of the batch before, if this one looks like it (a stream of tickets cut from one data set: the same number of
  // regions within a quarter, the same dominant class and median haplotype length within a tenth): taken as it is, no search.  The
  // results do not depend on the run length, only the schedule does.  ACCG_PHMM_SIZING_CACHE=0 turns it off.
  accg_ctx::SizingMemo& memo = b.ctx->sizing;
  uint64_t med_hap = 0;
  if (!b.hp.empty()) med_hap = b.hp[b.hp.size() / 2].len;          // (a sample, not the true median: enough to recognise a data set)
  {
    static const bool cache_off = [] { const char* e = getenv("ACCG_PHMM_SIZING_CACHE"); return e && e[0] == '0'; }();
    const uint64_t nr = b.regions.size();
    const bool alike = memo.valid && !cache_off && cand.size() > 1 && !getenv("ACCG_PHMM_STREAM_BUDGET") && memo.K_dom == K_dom && memo.lpp_dom == lpp_dom &&
                       memo.dom5 == dom5 && memo.nchar == nchar && nr * 4 >= memo.regions * 3 && nr * 4 <= memo.regions * 5 &&
                       med_hap * 10 >= memo.med_hap * 9 && med_hap * 10 <= memo.med_hap * 11 && (uint64_t)b.hp.size() * 4 >= memo.haps * 3 &&
                       (uint64_t)b.hp.size() * 4 <= memo.haps * 5;
    if (alike) { cand.clear(); cand.push_back(memo.budget); }
  }
  const bool from_memo = cand.size() == 1 && memo.valid && cand[0] == memo.budget && b.regions.size() > 0 && !(cand[0] == 1);
  std::vector<double> span1(cand.size(), -1.0), span2(cand.size(), -1.0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(accg::host_threads()) if (cand.size() * b.regions.size() >= 512 && accg::host_threads() > 1)
  for (int ci = 0; ci < (int)cand.size(); ci++) {
    if (cand.size() == 1) { span1[ci] = 0.0; continue; }      // a single candidate is taken as it is
    const Eval e1 = evaluate(cand[ci], false);
    span1[ci] = e1.span;
    if (dom5 && e1.pays) span2[ci] = evaluate(cand[ci], true).span;
  }
  for (size_t ci = 0; ci < cand.size(); ci++)
    for (int pr = 0; pr < 2; pr++) {
      const double span = pr ? span2[ci] : span1[ci];
      if (span >= 0 && (best_span < 0 || span < best_span)) { best_span = span; best_budget = cand[ci]; best_pairs = pr != 0; }
    }
  if (from_memo) best_pairs = memo.pairs;
  else if (cand.size() > 1) { memo.valid = true; memo.budget = best_budget; memo.pairs = best_pairs; memo.K_dom = K_dom; memo.lpp_dom = lpp_dom; memo.dom5 = dom5;
                              memo.nchar = nchar; memo.regions = b.regions.size(); memo.med_hap = med_hap; memo.haps = b.hp.size(); }

  tpc = std::chrono::steady_clock::now();
  struct Job { PhmmWork w, w2; int K, lpp, form; bool striped; int wg; uint64_t cost; uint32_t stream_len; };
  std::vector<Job> jobs;
  uint32_t cap_all = 0, hmax_all = 1;
  // Two passes over the regions.  The first, serial and cheap, cuts each region's haplotypes into runs and hands out the places of
  // what the second one writes (run table, haplotype lists, streams, jobs); the second fills them region by region, independently,
  // on the host's threads.
  struct RegPlan { std::vector<std::pair<uint32_t, uint32_t>> runs; std::vector<uint32_t> lens; uint32_t chunk0 = 0, ids0 = 0, job0 = 0, n_jobs = 0; size_t stream0 = 0; };
  std::vector<RegPlan> plan(b.regions.size());
  // (measured and dropped, round 4: rescue items of ONE haplotype each instead of the sweep's runs for batches of up to a few hundred
  // regions -- a 134-region configs[3] shard: planner + rescue 0.481 -> 0.491 ms, 1024 regions 2.52 -> 2.72: no gain at any size)
  {
    uint32_t chunk0 = 0, ids0 = 0; size_t stream0 = 0;
    for (size_t ri = 0; ri < b.regions.size(); ri++) {
      if (groups[ri].empty()) continue;
      const Region& r = b.regions[ri];
      RegPlan& P = plan[ri];
      chunk_region(b, r, best_budget, P.runs, P.lens);
      P.chunk0 = chunk0; P.ids0 = ids0; P.stream0 = stream0;
      for (size_t c = 0; c < P.runs.size(); c++) {
        cap_all = std::max(cap_all, (P.lens[c] + 63) / 64 * 64); hmax_all = std::max(hmax_all, P.runs[c].second);
        ids0 += P.runs[c].second;
        stream0 += (P.lens[c] + PHMM_STREAM_TAIL + 15) / 16 * 16;
        b.rescue_stream_cap = std::max(b.rescue_stream_cap, (int)((P.lens[c] + 63) / 64 * 64));
        b.rescue_haps_cap = std::max(b.rescue_haps_cap, (int)P.runs[c].second);
      }
      chunk0 += (uint32_t)P.runs.size();
      const uint32_t n_rchunks = (uint32_t)P.runs.size();
      b.regions_dev[ri] = {r.read0, r.n_reads, P.chunk0, n_rchunks, r.n_haps, 0};
      {   // upper bound of rescue jobs per class: a group starts with a distinct read of that class
        uint32_t per_class[PHMM_RESCUE_CLASSES] = {0};
        for (uint32_t k = 0; k < r.n_reads; k++) { int c, l, K; phmm_rescue_class(b.rd[r.read0 + k].len, &c, &l, &K); per_class[c]++; }
        for (int c = 0; c < PHMM_RESCUE_CLASSES; c++) b.rescue_bound[c] += (uint64_t)per_class[c] * (uint64_t)((n_rchunks + 1) / 2 * 2);   // (room for pairs)
      }
    }
    b.chunks_dev.resize(chunk0); b.chunk_stream16.resize(chunk0); b.chunk_stream_len.resize(chunk0);
    b.hap_ids.resize(ids0);
    b.streams.assign(stream0, 0);
  }
  // pairs of runs for one workgroup of two wavefronts (same reads, one dist table): five-operation form only, not the short reads
  // that run in the reference's operation order, and only where it buys occupancy at the batch's LDS caps
  auto paired = [&](const Group& Q) { return best_pairs && Q.form == 5 && !Q.striped && Q.lpp * Q.K > 16 && pairs_pay(Q.K, nchar, (int)cap_all, (int)hmax_all, Q.lpp); };
  {
    uint32_t job0 = 0;
    for (size_t ri = 0; ri < b.regions.size(); ri++) {
      RegPlan& P = plan[ri];
      P.job0 = job0;
      for (const Group& Q : groups[ri]) P.n_jobs += (uint32_t)(paired(Q) ? (P.runs.size() + 1) / 2 : P.runs.size());
      job0 += P.n_jobs;
    }
    jobs.resize(job0);
  }
#pragma omp parallel for schedule(dynamic, 1) num_threads(accg::host_threads()) if (b.regions.size() >= 8 && accg::host_threads() > 1)
  for (size_t ri = 0; ri < b.regions.size(); ri++) {
    if (groups[ri].empty()) continue;
    const Region& r = b.regions[ri];
    const RegPlan& P = plan[ri];
    const auto& runs = P.runs; const auto& lens = P.lens;
    std::vector<uint32_t> ids0(runs.size());
    uint32_t idp = P.ids0; size_t sp = P.stream0;
    for (size_t c = 0; c < runs.size(); c++) {
      const auto& run = runs[c];
      ids0[c] = idp;
      b.chunks_dev[P.chunk0 + c] = {idp, run.second};
      for (uint32_t k = 0; k < run.second; k++) b.hap_ids[idp++] = r.hap0 + run.first + k;
      // the run's stream: [marker][codes] per haplotype, a last marker, zeros (phmm_dev.h)
      b.chunk_stream16[P.chunk0 + c] = (uint32_t)(sp / 16);
      b.chunk_stream_len[P.chunk0 + c] = lens[c];
      size_t w_ = sp;
      for (uint32_t k = 0; k < run.second; k++) {
        const SeqRef& h = b.hp[r.hap0 + run.first + k];
        b.streams[w_++] = (uint8_t)nchar;
        const uint8_t* src = b.hp_ptr[r.hap0 + run.first + k];
        {
          struct Lut { uint8_t code[256]; Lut() { for (int i = 0; i < 256; i++) code[i] = i == 'A' ? 0 : i == 'C' ? 1 : i == 'G' ? 2 : i == 'T' ? 3 : 4; } };
          static const Lut lut;
          uint8_t* dst = b.streams.data() + w_;
          for (uint32_t x = 0; x < h.len; x++) dst[x] = lut.code[src[x]];
          w_ += h.len;
        }
      }
      b.streams[w_++] = (uint8_t)nchar;
      sp += (lens[c] + PHMM_STREAM_TAIL + 15) / 16 * 16;
    }
    uint32_t jp = P.job0;
    for (const Group& Q : groups[ri]) {
      PhmmWork w;
      for (int g = 0; g < PHMM_GROUPS; g++) w.read[g] = Q.read[g];
      w.pad_[0] = w.pad_[1] = 0;
      const bool pair = paired(Q);
      for (size_t c = 0; c < runs.size(); c += pair ? 2 : 1) {
        w.hap_off = ids0[c]; w.n_haps = runs[c].second;
        w.pad_[0] = b.chunk_stream16[P.chunk0 + c]; w.pad_[1] = b.chunk_stream_len[P.chunk0 + c];
        PhmmWork w2 = w;
        uint32_t len = lens[c];
        if (pair) {
          if (c + 1 < runs.size()) {
            w2.hap_off = ids0[c + 1]; w2.n_haps = runs[c + 1].second; len = std::max(len, lens[c + 1]);
            w2.pad_[0] = b.chunk_stream16[P.chunk0 + c + 1]; w2.pad_[1] = b.chunk_stream_len[P.chunk0 + c + 1];
          } else { w2.hap_off = 0; w2.n_haps = 0; w2.pad_[0] = w2.pad_[1] = 0; }
        }
        const uint64_t stripes = Q.striped ? (b.rd[Q.read[0]].len + 1024) / 1024 : 1;
        jobs[jp++] = {w, w2, Q.K, Q.lpp, Q.form, Q.striped, pair ? 2 : 1, stripes * (uint64_t)(len + 45) * (uint64_t)(8 * Q.K + 10), len};
      }
    }
  }
  tpd = std::chrono::steady_clock::now();
  // one launch per K; inside a launch the longest jobs go first so the tail is short
  // (sorted through (key, index) pairs: a Job is 120 bytes, the order is decided by a few of them)
  std::vector<std::pair<uint64_t, uint32_t>> order(jobs.size());
  for (size_t i = 0; i < jobs.size(); i++) {
    const Job& x = jobs[i];
    // descending: striped, merge class, lanes, K; ascending form; descending workgroup size and cost; ties in generation order.
    // Merge class: the jobs of the prepared five-operation sweep whose K lies in one window of phmm_launch_f32_multi and that have the
    // same workgroup size end up next to each other, ordered by lanes and K (= by rows per read), and can go out as ONE launch in fast mode.
    const int mc = merge_class(x.K, x.lpp, x.form, x.striped, x.wg);
    // (lanes per read as log2(lanes / 8) in two bits: 8, 16, 32, 64 -> 0..3 -- the six low bits of the lane count itself made 64 a 0 and
    // sent the longest jobs, reads of 513 to 1023 bases, to the back of the list)
    static_assert(PHMM_MAX_K < 32, "K takes five bits of the job sort key");
    const uint64_t lane_code = x.lpp >= 64 ? 3 : x.lpp >= 32 ? 2 : x.lpp >= 16 ? 1 : 0;
    // (inside a class K is fixed, so the order by cost is the order by stripes x (stream length + 45): 15 bits of it, clipped -- only
    // striped reads of several thousand bases against the longest streams reach the clip -- make the key 32 bits, sorted by three
    // stable counting passes instead of a comparison sort: 5 000 jobs of a 32-region ticket 150 -> 20 us)
    const uint64_t stripes = x.striped ? (b.rd[x.w.read[0]].len + 1024) / 1024 : 1;
    const uint32_t len_key = (uint32_t)std::min<uint64_t>(stripes * (uint64_t)(x.stream_len + 45), 32767);
    const uint32_t key = ((uint32_t)(x.striped ? 1 : 0) << 31) | ((uint32_t)(mc & 15) << 27) | ((uint32_t)lane_code << 25) | ((uint32_t)(x.K & 31) << 20) |
                         ((uint32_t)(7 - x.form) << 17) | ((uint32_t)(x.wg & 3) << 15) | len_key;
    order[i] = {(uint64_t)(~key), (uint32_t)i};
  }
  {
    std::vector<std::pair<uint64_t, uint32_t>> tmp(order.size());
    for (int pass = 0; pass < 3; pass++) {
      const int sh = 11 * pass;
      uint32_t cnt[2049] = {0};
      for (const auto& o : order) cnt[((o.first >> sh) & 2047u) + 1]++;
      for (int q = 0; q < 2048; q++) cnt[q + 1] += cnt[q];
      for (const auto& o : order) tmp[cnt[(o.first >> sh) & 2047u]++] = o;
      order.swap(tmp);
    }
  }
  // (the jobs are read through the sorted index: copying them into order first cost as much as the sort)
  b.work.clear();
  b.work.reserve(jobs.size() * 2);
  for (size_t i = 0; i < jobs.size(); i++) {
    const Job& J = jobs[order[i].second];
    if (b.launches.empty() || b.launches.back().K != J.K || b.launches.back().lpp != J.lpp || b.launches.back().form != J.form ||
        b.launches.back().striped != J.striped || b.launches.back().wg != J.wg)
      b.launches.push_back({J.K, J.lpp, J.form, J.striped, J.wg, (uint32_t)b.work.size(), 0, 0, 0, 0, 0});
    KLaunch& L = b.launches.back();
    b.work.push_back(J.w);
    if (J.wg == 2) b.work.push_back(J.w2);
    L.n_work += (uint32_t)J.wg;
    L.cost += J.cost * (uint64_t)J.wg;
    b.any_form5 |= J.form == 5 && !J.striped && J.lpp * J.K > 16;
    L.stream_cap = std::max(L.stream_cap, (int)((J.stream_len + 63) / 64 * 64));
    L.haps_cap = std::max(L.haps_cap, (int)std::max(J.w.n_haps, J.wg == 2 ? J.w2.n_haps : 0u));
  }
  // The fast mode's list: consecutive classes of one merge class become one launch.  configs[3] shards, whole pass, separate launches ->
  // one launch: 64 regions 1.06 -> 0.92 ms, 128 regions 1.75 -> 1.59, 256 regions 3.21 -> 2.92, 512 regions 5.82 -> 5.49, 1024 regions
  // 11.02 -> 10.82 on one box and 10.99 -> 10.95 on another.  (With the rescue's first layout -- two launches per class on the same
  // streams -- the 1024-region pass had measured 3 % SLOWER merged on two boxes although the sweep itself was faster; with the merged
  // rescue windows that is gone.)  ACCG_PHMM_MERGE=0: one launch per class.
  {
    const char* em = getenv("ACCG_PHMM_MERGE");          // read per batch: the tests build the same batch both ways
    const bool merge_off = em && em[0] == '0';
    const int nchar = b.has_n ? 5 : 4;
    int prev_mc = 0;
    for (const KLaunch& l : b.launches) {
      const int mc = merge_off ? 0 : merge_class(l.K, l.lpp, l.form, l.striped, l.wg);
      if (mc && mc == prev_mc && !b.launches_fast.empty()) {
        KLaunch& m = b.launches_fast.back();          // (lanes, K) descend along the work list
        if (!m.k_hi) m.k_hi = m.K;
        m.K = std::min(m.K, l.K); m.k_hi = std::max(m.k_hi, l.K); m.n_work += l.n_work; m.cost += l.cost;
        m.stream_cap = std::max(m.stream_cap, l.stream_cap); m.haps_cap = std::max(m.haps_cap, l.haps_cap);
      } else {
        b.launches_fast.push_back(l);
      }
      prev_mc = mc;
    }
    // LDS request and occupancy of a merged launch: every class in it lays its block out with the launch's stream and haplotype caps
    for (KLaunch& m : b.launches_fast) {
      if (!m.k_hi) continue;
      m.wpc_min = 32;
      for (const KLaunch& l : b.launches) {
        if (l.work0 < m.work0 || l.work0 >= m.work0 + m.n_work) continue;
        m.lds = std::max(m.lds, phmm_lds_bytes(l.K, 4, nchar, m.stream_cap, m.haps_cap, l.lpp, true, false, l.wg));
        m.wpc_min = std::min(m.wpc_min, waves_per_cu(l.K, nchar, m.stream_cap, m.haps_cap, l.lpp, false, 5, l.wg));
      }
    }
  }
  for (std::vector<KLaunch>* list : {&b.launches, &b.launches_fast}) {
    std::vector<KLaunch>& launches = *list;
    // which forked stream takes which launch: ACCG_PHMM_AUX = streams used (default all), ACCG_PHMM_LPT=1: longest first onto the
      // least loaded stream instead of round robin in class order
    const char* ea = getenv("ACCG_PHMM_AUX"); const char* el = getenv("ACCG_PHMM_LPT");
    const int n_aux = std::max(1, std::min((int)accg_ctx::N_AUX, ea ? atoi(ea) : (int)accg_ctx::N_AUX));
    if (el && el[0] == '1') {
      std::vector<size_t> idx(launches.size());
      std::iota(idx.begin(), idx.end(), (size_t)0);
      std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return launches[x].cost > launches[y].cost; });
      std::vector<uint64_t> load((size_t)n_aux, 0);
      for (size_t i : idx) {
        const int q = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        launches[i].aux = q; load[(size_t)q] += launches[i].cost;
      }
    } else {
      for (size_t i = 0; i < launches.size(); i++) launches[i].aux = (int)(i % (size_t)n_aux);
    }
  }
  if (trace_p) {
    auto us = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
    fprintf(stderr, "partition: groups %.0f us, budget search %.0f us, jobs + streams %.0f us, sort + flatten %.0f us (%zu jobs)\n", us(tpa, tpb), us(tpb, tpc), us(tpc, tpd),
            us(tpd, std::chrono::steady_clock::now()), jobs.size());
  }
}

// layout of accg_phmm_batch::d_state (uint32 words)
size_t state_counts(const accg_phmm_batch& b) { return (b.rd.size() + 1) / 2 * 2; }
size_t state_redo(const accg_phmm_batch& b) { return state_counts(b) + PHMM_RESCUE_CLASSES; }     // jobs to redo in the strict form, per class
size_t state_nresc(const accg_phmm_batch& b) { return state_counts(b) + (2 * PHMM_RESCUE_CLASSES + 3) / 2 * 2; }
size_t state_words(const accg_phmm_batch& b) { return state_nresc(b) + 2; }

template <typename T>
PhmmArgs<T> make_args(const accg_phmm_batch& b, T* out, const PhmmTables<T>& tab) {
  PhmmArgs<T> a;
  a.rblob = b.d_rblob.p; a.hblob = b.d_hblob.p; a.rd = b.d_rd.p; a.rd_out = b.d_rd_out.p; a.hp = b.d_hp.p;
  a.hp_local = b.d_hp_local.p; a.hap_desc = b.d_hap_desc.p; a.work = b.d_work.p; a.out = out;
  a.raw = b.d_out.p; a.n_rescued = reinterpret_cast<unsigned long long*>(b.d_state.p + state_nresc(b)); a.tab = tab;
  a.read_flag = b.d_state.p;
  a.nchar = b.has_n ? 5 : 4; a.stream_cap = 0; a.haps_cap = 0; a.job_count = nullptr; a.lds_min = 0;
  a.job_map = nullptr; a.redo_count = nullptr; a.redo_list = nullptr; a.is_redo = 0;
  a.rec = PhmmRowRecs{b.d_rec_coef.p, b.d_rec_dist.p, b.d_rec_misc.p, b.d_rd_row0.p, b.d_rd_shape.p};
  a.streams = b.d_streams.p;
  a.fair = 0; a.zero_words = nullptr; a.n_zero = 0; a.clock_out = nullptr;
  return a;
}

int launch_f32(accg_phmm_batch* b, int mode, hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr, bool no_flags = false) {
  PhmmArgs<float> a = make_args<float>(*b, b->d_out.p, b->ctx->tab_f);
  if (no_flags) a.read_flag = nullptr;            // a speculative pass: no planner will read (and clear) them
  // A pass needs no memset: the read flags are cleared by the planner as it reads them, the rescue job counts and the counter of
  // rescued pairs by block 0 of every sweep launch (the previous pass is through with them, this pass's planner runs behind the sweep),
  // and batch creation left all of them at zero.  ACCG_PHMM_PREPARE_EACH_PASS=1: the per-row records of the five-operation sweep
  // (phmm_prepare_rows; normally written once, at batch creation, like the streams) are rewritten at the start of every pass.
  static const bool each_pass = [] { const char* e = getenv("ACCG_PHMM_PREPARE_EACH_PASS"); return e && e[0] == '1'; }();
  if (each_pass && mode != ACCG_PHMM_STRICT && b->any_form5 && b->n_rows)
    ACCG_HIP(phmm_prepare_rows_launch(a, (uint32_t)b->rd.size(), nullptr, 0, b->ctx->stream));
  a.zero_words = b->d_state.p + state_counts(*b);
  a.n_zero = (int)(state_words(*b) - state_counts(*b));
  a.clock_out = reinterpret_cast<unsigned long long*>(b->d_clock.p);
  if (ev_begin) ACCG_HIP(hipEventRecord(ev_begin, b->ctx->stream));
  const std::vector<KLaunch>& launches = mode == ACCG_PHMM_STRICT ? b->launches : b->launches_fast;
  const bool fork = launches.size() > 1 && !(b->kernel_copies && launches.size() <= 2);   // several rows-per-lane classes: run them side by side
  if (fork) ACCG_HIP(ctx_fork(b->ctx));
  for (const KLaunch& l : launches) {
    a.stream_cap = l.stream_cap; a.haps_cap = l.haps_cap;
    hipStream_t st = fork ? b->ctx->aux[l.aux] : b->ctx->stream;
    const bool strict_l = mode == ACCG_PHMM_STRICT || l.lpp * l.K <= 16;
    const int k_top = l.k_hi ? l.k_hi : l.K;        // registers and LDS of a merged launch are those of its largest K
    // pinned occupancy: the launch asks for as much LDS as leaves exactly 8, 16 or 32 of its wavefronts on a CU
    const int wg = strict_l ? 1 : l.wg;         // a strict launch runs the items of a pair as two wavefronts of their own
    const int natural = l.k_hi ? l.wpc_min : waves_per_cu(k_top, a.nchar, l.stream_cap, l.haps_cap, l.lpp, strict_l, l.form, wg);
    // (the compiled strict column is VOP3-heavy and does gain from a third wavefront per SIMD: any multiple of four for it)
    const int wpc = b->force_wpc > 0 ? std::min(b->force_wpc, natural) : b->force_wpc < 0 ? natural
                    : strict_l ? std::max(natural / 4 * 4, std::min(natural, 4)) : pinned_wpc(natural);
    a.lds_min = (int)((160 * 1024 / std::max(wpc / wg, 1)) / 512 * 512);     // per workgroup
    // Priority steps (phmm_job): for a launch that is about one round of jobs on the chip's slots, where all wavefronts of a SIMD start
    // together and the arbiter's oldest-first rule makes them finish far apart (configs[1]: 0.279 -> 0.259 ms); with several rounds the
    // slots refill as they empty and the steps only cost (a 128-region configs[3] shard: + 2 %).  ACCG_PHMM_FAIR: 0 = never, n = thresholds n.
    static const int fair_knob = [] { const char* e = getenv("ACCG_PHMM_FAIR"); return e ? atoi(e) : 3; }();
    a.fair = (uint64_t)l.n_work <= 2ull * (uint64_t)b->ctx->n_cu * (uint64_t)std::max(wpc, 1) ? fair_knob : 0;
    // Reads of at most 15 bases take the reference's operation order in fast mode too: their log10 is close to 0, where the
    // reference's float `log10f(x) - log10f(2^120)` has a granularity of 3.8e-6 absolute, so a one-ulp difference in x can show
    // as more than 1e-5 relative (a two-base read did, at 5.4e-6; tools/fuzz_phmm.py).  Long reads were suspected as well and
    // cleared: with them contracted the worst case over 1 500 random regions stays at that granularity, 2.4e-6.
    if (l.striped) a.lds_min = 0;                   // one long read per wavefront and a large LDS block: nothing to pin
    static const bool trace_l = getenv("ACCG_TRACE_LAUNCH") != nullptr;
    if (trace_l) fprintf(stderr, "launch K %d..%d lanes %d form %d wg %d striped %d: %u jobs, stream_cap %d haps_cap %d, natural %d wpc %d lds_min %d fair %d aux %d\n",
                         l.K, k_top, l.lpp, l.form, wg, (int)l.striped, l.n_work, l.stream_cap, l.haps_cap, natural, wpc, a.lds_min, a.fair, l.aux);
    if (l.k_hi) ACCG_HIP(phmm_launch_f32_multi(l.K, l.k_hi, l.lds, a, l.work0, l.n_work, st, wg));
    else ACCG_HIP(phmm_launch_f32(l.K, l.lpp, strict_l, strict_l ? 7 : l.form, l.striped, a, l.work0, l.n_work, st, wg));
  }
  if (fork) ACCG_HIP(ctx_join(b->ctx));
  if (ev_end) ACCG_HIP(hipEventRecord(ev_end, b->ctx->stream));
  return ACCG_OK;
}
// on_tail: on the context's tail stream and its forked streams (a pipelined pass, run_direct) instead of the main stream's
bool graphs_wanted();
int launch_rescue(accg_phmm_batch* b, int mode, bool on_tail = false) {
  hipStream_t s = on_tail ? b->ctx->tail : b->ctx->stream;
  hipStream_t* const aux = on_tail ? b->ctx->aux_t : b->ctx->aux;
  // the rescue probe (see accg_phmm_batch::rescue_known); ACCG_PHMM_PROBE=0 turns it off
  const int mi = mode == ACCG_PHMM_STRICT ? 1 : 0;
  static const bool probe_off = [] { const char* e = getenv("ACCG_PHMM_PROBE"); return e && e[0] == '0'; }();
  bool arm = false;
  if (!probe_off && !graphs_wanted()) {
    if (b->rescue_known[mi] == 0 && b->probe_armed[mi] && hipEventQuery(b->ev_probe[mi]) == hipSuccess)
      b->rescue_known[mi] = __atomic_load_n(b->probe_flag[mi], __ATOMIC_ACQUIRE) ? 2 : 1;
    (void)hipGetLastError();                    // (hipErrorNotReady of the query is not an error)
    if (b->rescue_known[mi] == 1) return ACCG_OK;
    if (b->rescue_known[mi] == 0 && !b->probe_armed[mi] && b->probe_runs[mi]++ >= 1) {        // from the second pass in this mode on
      accg_ctx* c = b->ctx;
      if (!c->h_flags) {
        ACCG_HIP(hipHostMalloc((void**)&c->h_flags, 4096, hipHostMallocCoherent));
        memset(c->h_flags, 0, 4096);
      }
      b->probe_flag[mi] = c->h_flags + (c->flag_next++ % 1024u);
      *b->probe_flag[mi] = 0u;
      if (!b->ev_probe[mi]) ACCG_HIP(hipEventCreateWithFlags(&b->ev_probe[mi], hipEventDisableTiming));
      arm = true;
    }
  }
  PhmmPlanArgs p;
  p.host_flag = arm ? b->probe_flag[mi] : nullptr;
  p.regions = b->d_regions.p; p.chunks = b->d_chunks.p; p.sorted_reads = b->d_sorted_reads.p; p.rd = b->d_rd.p;
  p.rd_out = b->d_rd_out.p; p.read_flag = b->d_state.p; p.jobs = b->d_rescue_jobs.p; p.counts = b->d_state.p + state_counts(*b);
  p.flagged = b->d_flagged.p;
  for (int c = 0; c <= PHMM_RESCUE_CLASSES; c++) p.class_off[c] = b->rescue_off[c];
  // Pairs of items that share their dist table (fast mode): worth it when a region's haplotypes make two runs or more on average --
  // with single runs every second wavefront would sit idle on its registers.  ACCG_PHMM_RESCUE_WG=1: never, =2: always.
  // (knobs read per pass: the tests run one batch every way)
  const int wg_knob = [] { const char* e = getenv("ACCG_PHMM_RESCUE_WG"); return e ? atoi(e) : 0; }();
  const bool pairs = mode != ACCG_PHMM_STRICT && wg_knob != 1 && (wg_knob == 2 || b->chunks_dev.size() >= 2 * b->regions_dev.size());
  p.pairs = pairs ? 1u : 0u;
  ACCG_HIP(phmm_rescue_plan_launch(p, (uint32_t)b->regions_dev.size(), s));
  PhmmArgs<double> a = make_args<double>(*b, b->d_out64.p, b->ctx->tab_d);
  a.work = b->d_rescue_jobs.p;
  a.stream_cap = b->rescue_stream_cap; a.haps_cap = b->rescue_haps_cap;
  const bool f5_off = [] { const char* e = getenv("ACCG_PHMM_RESCUE_FORM5"); return e && e[0] == '0'; }();   // A/B knobs
  const bool merge_off = [] { const char* e = getenv("ACCG_PHMM_RESCUE_MERGE"); return e && e[0] == '0'; }();
  // Fast mode, every read in the five-operation form's range: the classes with K <= 8 go out as two launches by register budget
  // (phmm_dev.h: PHMM_RESCUE_MERGED) and one strict re-run launch behind both, instead of two launches per class.
  const bool merged = mode != ACCG_PHMM_STRICT && b->all_form5 && !f5_off && !merge_off;
  // the strict re-run launches: only when some read can reach the range they exist for (parse_reads: `deep`); ACCG_PHMM_REDO_ALWAYS=1: always
  const bool redo = b->redo_possible || [] { const char* e = getenv("ACCG_PHMM_REDO_ALWAYS"); return e && e[0] == '1'; }();
  const int wg = pairs ? 2 : 1;
  uint64_t win_units[2] = {0, 0};
  size_t win_lds[2] = {0, 0}, redo_lds = 0;
  int n_launch = 0;
  for (int c = 0; c < PHMM_RESCUE_CLASSES; c++) {
    if (!b->rescue_bound[c]) continue;
    const int w = merged ? phmm_rescue_window(c) : -1;
    if (w < 0) { n_launch++; continue; }
    int lpp_c, k_c;
    phmm_rescue_shape(c, &lpp_c, &k_c);
    win_units[w] += b->rescue_bound[c] / (uint64_t)wg;
    win_lds[w] = std::max(win_lds[w], phmm_lds_bytes(k_c, 8, a.nchar, a.stream_cap, a.haps_cap, lpp_c, true, false, wg));
    redo_lds = std::max(redo_lds, phmm_lds_bytes(k_c, 8, a.nchar, a.stream_cap, a.haps_cap, lpp_c, false, false, 1));
  }
  n_launch += (win_units[0] != 0) + (win_units[1] != 0);
  // (a small batch -- the blocking call per region -- queues its few, short rescue launches one behind the other: the events of a fork
  // and a join cost it more than the launches overlapping could save)
  const bool fork = n_launch > 1 && !b->kernel_copies;
  if (fork) ACCG_HIP(on_tail ? ctx_fork_tail(b->ctx) : ctx_fork(b->ctx));
  int rr = 0;
  PhmmRescueSet rs;
  for (int c = 0; c <= PHMM_RESCUE_CLASSES; c++) rs.off[c] = b->rescue_off[c];
  rs.counts = b->d_state.p + state_counts(*b);
  for (int w = 0; w < 2; w++) {
    if (!win_units[w]) continue;
    hipStream_t st = fork ? aux[rr++ % accg_ctx::N_AUX] : s;
    a.job_count = nullptr; a.job_map = nullptr; a.is_redo = 0;
    a.redo_count = b->d_state.p + state_redo(*b);          // one list for all merged classes: the first class's counter, the list from slot 0
    a.redo_list = b->d_redo.p;
    static const uint64_t grid_cap = [] { const char* e = getenv("ACCG_PHMM_RESCUE_GRID"); return e && atoi(e) > 0 ? (uint64_t)atoi(e) : (uint64_t)PHMM_RESCUE_GRID_DEFAULT; }();
    ACCG_HIP(phmm_launch_rescue_multi(w, wg, win_lds[w], a, rs, (uint32_t)std::min<uint64_t>(win_units[w], grid_cap), st));
  }
  for (int c = 0; c < PHMM_RESCUE_CLASSES; c++) {
    const uint32_t bound = (uint32_t)b->rescue_bound[c];       // < 2^32: checked at batch creation
    if (!bound || (merged && phmm_rescue_window(c) >= 0)) continue;
    a.job_count = b->d_state.p + state_counts(*b) + c;
    hipStream_t st = fork ? aux[rr++ % accg_ctx::N_AUX] : s;
    int lpp_c, k_c;
    phmm_rescue_shape(c, &lpp_c, &k_c);
    const bool strict = mode == ACCG_PHMM_STRICT;
    a.job_map = nullptr; a.is_redo = 0;
    a.redo_count = strict ? nullptr : b->d_state.p + state_redo(*b) + c;
    a.redo_list = strict ? nullptr : b->d_redo.p + b->rescue_off[c];
    ACCG_HIP(phmm_launch_rescue_f64(k_c, lpp_c, strict, phmm_rescue_striped(c), a, b->rescue_off[c], bound, st, PHMM_RESCUE_GRID_DEFAULT,
                                    b->all_form5 && !f5_off, wg));
    if (!strict && redo) {        // the jobs that launch listed (results next to the denormal range), in the reference's operation order
      PhmmArgs<double> r = a;
      r.job_count = a.redo_count; r.job_map = a.redo_list; r.redo_count = nullptr; r.redo_list = nullptr; r.is_redo = 1;
      ACCG_HIP(phmm_launch_rescue_f64(k_c, lpp_c, true, phmm_rescue_striped(c), r, b->rescue_off[c], bound, st, PHMM_REDO_GRID));
    }
  }
  if (fork) ACCG_HIP(on_tail ? ctx_join_tail(b->ctx) : ctx_join(b->ctx));
  if ((win_units[0] || win_units[1]) && redo) {     // behind both windows: the items they listed, in the reference's operation order
    PhmmArgs<double> r = a;
    r.job_count = nullptr; r.job_map = nullptr; r.is_redo = 1;
    r.redo_count = b->d_state.p + state_redo(*b); r.redo_list = b->d_redo.p;
    ACCG_HIP(phmm_launch_redo_multi(redo_lds, r, rs, PHMM_REDO_GRID, s));
  }
  if (arm) { ACCG_HIP(hipEventRecord(b->ev_probe[mi], s)); b->probe_armed[mi] = true; }
  return ACCG_OK;
}

}  // namespace

// One region's wire blobs, parsed and validated (lengths, bases, the range tests of the sweep's forms); offsets are relative to the
// region's own blobs.  Regions are independent: a multi-region batch parses them on the host's threads, the callers of an
// accg_phmm_mux each parse their own before they queue up.
struct PhmmParsed { std::vector<SeqRef> rd, hp; std::vector<uint8_t> form; std::vector<const uint8_t*> hp_ptr; bool has_n = false, deep = false; int nr = 0, nh = 0; };
static void parse_region(const void* reads_ser, size_t reads_bytes, const void* haps_ser, size_t haps_bytes, PhmmParsed& P) {
  P.nr = parse_reads((const uint8_t*)reads_ser, reads_bytes, 0u, P.rd, P.form, P.deep);
  if (P.nr >= 0) P.nh = parse_haps((const uint8_t*)haps_ser, haps_bytes, 0u, P.hp, P.has_n, P.hp_ptr);
}
static int phmm_batch_create_impl(accg_ctx* ctx, int n_regions, const void* const* reads_ser, const size_t* reads_bytes, const void* const* haps_ser,
                                  const size_t* haps_bytes, const PhmmParsed* const* pre, accg_phmm_batch** out);
extern "C" int accg_phmm_batch_create(accg_ctx* ctx, int n_regions, const void* const* reads_ser,
                                      const size_t* reads_bytes, const void* const* haps_ser,
                                      const size_t* haps_bytes, accg_phmm_batch** out) {
  return phmm_batch_create_impl(ctx, n_regions, reads_ser, reads_bytes, haps_ser, haps_bytes, nullptr, out);
}
// pre (nullable): the regions already parsed (by parse_region over the same blobs)
static int phmm_batch_create_impl(accg_ctx* ctx, int n_regions, const void* const* reads_ser, const size_t* reads_bytes, const void* const* haps_ser,
                                  const size_t* haps_bytes, const PhmmParsed* const* pre, accg_phmm_batch** out) {
  if (!ctx) return ACCG_ERR_NOT_INITIALISED;
  if (!out || n_regions < 0 || (n_regions > 0 && (!reads_ser || !reads_bytes || !haps_ser || !haps_bytes))) return ACCG_ERR_BAD_ARG;
  *out = nullptr;
  ACCG_HIP(hipSetDevice(ctx->device));
  PoolScope pool_scope(&ctx->pool);
  std::unique_ptr<accg_phmm_batch> b(new accg_phmm_batch);
  SyncOnError sync_on_error(ctx->stream);
  b->ctx = ctx;
  const auto tparse0 = std::chrono::steady_clock::now();
  uint64_t roff = 0, hoff = 0;
  for (int i = 0; i < n_regions; i++) { roff += reads_bytes[i]; hoff += haps_bytes[i]; }
  if (roff >= (1ull << 32) || hoff >= (1ull << 32)) return ACCG_ERR_TOO_LONG;   // 32-bit blob offsets
  // Regions are parsed (lengths, base validation, the range tests of the sweep's forms) independently of each other, on the host
  // threads this process may use when there are enough of them, and merged in order.
  std::vector<PhmmParsed> parsed(pre ? (size_t)0 : (size_t)n_regions);
  std::vector<uint64_t> roffs((size_t)n_regions + 1, 0), hoffs((size_t)n_regions + 1, 0);
  for (int i = 0; i < n_regions; i++) { roffs[i + 1] = roffs[i] + reads_bytes[i]; hoffs[i + 1] = hoffs[i] + haps_bytes[i]; }
  if (!pre) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(accg::host_threads()) if (n_regions >= 8 && accg::host_threads() > 1)
    for (int i = 0; i < n_regions; i++) parse_region(reads_ser[i], reads_bytes[i], haps_ser[i], haps_bytes[i], parsed[i]);
  }
  for (int i = 0; i < n_regions; i++) {
    const PhmmParsed& P = pre ? *pre[i] : parsed[i];
    if (P.nr < 0) return P.nr;
    if (P.nh < 0) return P.nh;
    Region r;
    r.read0 = (uint32_t)b->rd.size(); r.hap0 = (uint32_t)b->hp.size(); r.out0 = b->pairs;
    const int nr = P.nr, nh = P.nh;
    // (a parsed region's offsets are relative to its own blobs: shifted to the blobs' places in the batch)
    for (const SeqRef& x : P.rd) b->rd.push_back({x.off + (uint32_t)roffs[i], x.len});
    b->rd_form.insert(b->rd_form.end(), P.form.begin(), P.form.end());
    for (uint8_t f : P.form) b->all_form5 &= f == 5;
    for (const SeqRef& x : P.hp) b->hp.push_back({x.off + (uint32_t)hoffs[i], x.len});
    b->hp_ptr.insert(b->hp_ptr.end(), P.hp_ptr.begin(), P.hp_ptr.end());
    b->has_n |= P.has_n;
    b->redo_possible |= P.deep;
    r.n_reads = (uint32_t)nr; r.n_haps = (uint32_t)nh;
    uint64_t rsum = 0, hsum = 0;
    for (int k = 0; k < nr; k++) rsum += b->rd[r.read0 + k].len;
    for (int k = 0; k < nh; k++) hsum += b->hp[r.hap0 + k].len;
    b->cells += rsum * hsum;
    for (int k = 0; k < nr; k++) {
      uint64_t o = r.out0 + (uint64_t)k * nh;
      if (o + nh >= (1ull << 32)) return ACCG_ERR_TOO_LONG;   // 32-bit output indices
      b->rd_out.push_back((uint32_t)o);
    }
    for (int k = 0; k < nh; k++) b->hp_local.push_back((uint32_t)k);
    b->pairs += (uint64_t)nr * nh;
    b->regions.push_back(r);
  }
  parsed.clear();
  roff = roffs[n_regions]; hoff = hoffs[n_regions];
  b->algo_bytes = roff + hoff + 4 * b->pairs;   // blobs in, one float per pair out
  if (getenv("ACCG_TRACE")) fprintf(stderr, "accg_phmm_batch_create: parse %.0f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tparse0).count());
  b->rd_row0.assign(b->rd.size() + 1, 0); b->rd_shape.assign(b->rd.size(), 0);   // filled by partition()
  const auto tp0 = std::chrono::steady_clock::now();
  partition(*b);
  const auto tp1 = std::chrono::steady_clock::now();
  // Speculative fp64 jobs: only for a one-shot batch (accg_ctx::oneshot) whose fp32 jobs and fp64 jobs TOGETHER leave every wavefront a
  // SIMD of its own, whose reads all pass the five-operation form's range tests (the merged rescue windows), none in a class outside
  // them, and none that could need the strict re-run launch.  ACCG_PHMM_SPEC=0 turns it off.
  {
    static const bool spec_off_env = [] { const char* e = getenv("ACCG_PHMM_SPEC"); return e && e[0] == '0'; }();
    // ... and only when the call before on this context had something to rescue: the fp64 pass over every read is longer than the
    // sweep, so on data that never underflows fp32 the speculation would cost a region a quarter more (100 reads x 10 haplotypes, all
    // related: 0.118 ms without, 0.135-0.158 with); a caller's regions come from one data set, the last one is the best predictor there is
    bool ok = ctx->oneshot && ctx->alone && ctx->spec_hint && !spec_off_env && b->all_form5 && !b->redo_possible && !b->rd.empty();
    uint64_t n_items[PHMM_RESCUE_CLASSES] = {0};
    if (ok) {
      for (size_t ri = 0; ri < b->regions.size() && ok; ri++) {
        const Region& r = b->regions[ri];
        const uint32_t n_chunks = b->regions_dev[ri].n_chunks;
        for (uint32_t i = 0; i < r.n_reads;) {
          int cls, lpp, K;
          phmm_rescue_class(b->rd[b->sorted_reads[r.read0 + i]].len, &cls, &lpp, &K);      // the longest read of a group decides its class (phmm_rescue_plan)
          if (phmm_rescue_window(cls) < 0) { ok = false; break; }
          n_items[cls] += n_chunks;
          i += 64u / (uint32_t)lpp;
        }
      }
      uint64_t total = 0;
      for (int c = 0; c < PHMM_RESCUE_CLASSES; c++) total += n_items[c];
      ok = ok && total > 0 && total + b->work.size() <= 4ull * (uint64_t)std::max(ctx->n_cu, 1);
    }
    if (ok) {
      for (int c = 0; c < PHMM_RESCUE_CLASSES; c++) b->spec_off[c + 1] = b->spec_off[c] + (uint32_t)n_items[c];
      b->spec_jobs.assign(b->spec_off[PHMM_RESCUE_CLASSES], PhmmWork{});
      b->spec_counts.assign(PHMM_RESCUE_CLASSES, 0u);
      for (size_t ri = 0; ri < b->regions.size(); ri++) {
        const Region& r = b->regions[ri];
        const PhmmRegionDev& rdv = b->regions_dev[ri];
        for (uint32_t i = 0; i < r.n_reads;) {
          int cls, lpp, K;
          phmm_rescue_class(b->rd[b->sorted_reads[r.read0 + i]].len, &cls, &lpp, &K);
          const uint32_t per = 64u / (uint32_t)lpp;
          PhmmWork w;
          for (uint32_t g = 0; g < PHMM_GROUPS; g++) w.read[g] = (g < per && i + g < r.n_reads) ? b->sorted_reads[r.read0 + i + g] : PHMM_NO_READ;
          w.pad_[0] = w.pad_[1] = 0;
          for (uint32_t c = 0; c < rdv.n_chunks; c++) {
            w.hap_off = b->chunks_dev[rdv.chunk0 + c].ids0; w.n_haps = b->chunks_dev[rdv.chunk0 + c].n;
            b->spec_jobs[b->spec_off[cls] + b->spec_counts[cls]++] = w;
          }
          i += per;
        }
      }
      b->spec = true;
    }
  }
  hipStream_t s = ctx->stream;
  int st;
  for (int c = 0; c < PHMM_RESCUE_CLASSES; c++) {
    if (b->rescue_bound[c] >= (1ull << 32) || (uint64_t)b->rescue_off[c] + b->rescue_bound[c] >= (1ull << 32)) return ACCG_ERR_TOO_LONG;
    b->rescue_off[c + 1] = b->rescue_off[c] + (uint32_t)b->rescue_bound[c];
  }
  // One device arena: [uploaded tables and blobs][scratch][state | results].  The uploaded part is assembled in the context's
  // pinned staging and goes over in a single copy; the last words of `state` (n_rescued) sit right in front of `out`, so the
  // results come back in a single copy too.
  size_t off = 0;
  auto take = [&](size_t bytes, size_t align = 256) { off = (off + align - 1) / align * align; const size_t o = off; off += bytes; return o; };
  auto vbytes = [](const auto& v) { return v.size() * sizeof(v[0]); };
  const size_t o_rblob = take(roff + 16), o_hblob = take(hoff + 16), o_rd = take(vbytes(b->rd)), o_hp = take(vbytes(b->hp)),
               o_rd_out = take(vbytes(b->rd_out)), o_hp_local = take(vbytes(b->hp_local)), o_hap_ids = take(b->hap_ids.size() * sizeof(PhmmHapDesc)),
               o_work = take(vbytes(b->work)), o_regions = take(vbytes(b->regions_dev)), o_chunks = take(vbytes(b->chunks_dev)),
               o_sorted = take(vbytes(b->sorted_reads)), o_row0 = take(vbytes(b->rd_row0)), o_shape = take(vbytes(b->rd_shape)), o_streams = take(b->streams.size() + 16),
               o_spec_jobs = take(vbytes(b->spec_jobs)), o_spec_counts = take(vbytes(b->spec_counts));
  const size_t upload_bytes = off;
  if (b->n_rows >= (1ull << 32)) return ACCG_ERR_TOO_LONG;
  const size_t n_rec = (size_t)b->n_rows + 1;
  const size_t o_rec_coef = take(n_rec * sizeof(float4)), o_rec_dist = take(n_rec * sizeof(float4)), o_rec_misc = take(n_rec * sizeof(float4));
  const size_t o_clock = take(4 * sizeof(uint64_t));
  const size_t o_flagged = take((b->rd.size() + 1) * sizeof(uint32_t));
  const size_t o_jobs = take(((size_t)b->rescue_off[PHMM_RESCUE_CLASSES] + 1) * sizeof(PhmmWork));
  const size_t o_redo = take(((size_t)b->rescue_off[PHMM_RESCUE_CLASSES] + 1) * sizeof(uint32_t));
  const size_t o_out64 = take((b->pairs + 1) * sizeof(double));
  const size_t sw = state_words(*b);                         // even: the u64 counter at its end is 8-byte aligned
  const size_t o_state = take((sw + (sw & 1)) * sizeof(uint32_t) + (b->pairs + 1) * sizeof(float));
  const size_t o_out = o_state + (sw + (sw & 1)) * sizeof(uint32_t);
  if ((st = b->d_arena.alloc(off + 256)) != ACCG_OK) return st;
  const auto tq1 = std::chrono::steady_clock::now();
  uint8_t* base = b->d_arena.p;
  b->d_rblob.place(base, o_rblob, roff + 16); b->d_hblob.place(base, o_hblob, hoff + 16);
  b->d_rd.place(base, o_rd, b->rd.size()); b->d_hp.place(base, o_hp, b->hp.size());
  b->d_rd_out.place(base, o_rd_out, b->rd_out.size()); b->d_hp_local.place(base, o_hp_local, b->hp_local.size());
  b->d_hap_desc.place(base, o_hap_ids, b->hap_ids.size()); b->d_work.place(base, o_work, b->work.size());
  b->d_regions.place(base, o_regions, b->regions_dev.size()); b->d_chunks.place(base, o_chunks, b->chunks_dev.size());
  b->d_sorted_reads.place(base, o_sorted, b->sorted_reads.size());
  b->d_rd_row0.place(base, o_row0, b->rd_row0.size()); b->d_rd_shape.place(base, o_shape, b->rd_shape.size()); b->d_streams.place(base, o_streams, b->streams.size() + 16);
  b->d_spec_jobs.place(base, o_spec_jobs, b->spec_jobs.size()); b->d_spec_counts.place(base, o_spec_counts, b->spec_counts.size());
  b->d_rec_coef.place(base, o_rec_coef, n_rec); b->d_rec_dist.place(base, o_rec_dist, n_rec); b->d_rec_misc.place(base, o_rec_misc, n_rec);
  b->d_clock.place(base, o_clock, 4);
  b->d_flagged.place(base, o_flagged, b->rd.size() + 1);
  b->d_rescue_jobs.place(base, o_jobs, (size_t)b->rescue_off[PHMM_RESCUE_CLASSES] + 1);
  b->d_redo.place(base, o_redo, (size_t)b->rescue_off[PHMM_RESCUE_CLASSES] + 1);
  b->d_out64.place(base, o_out64, b->pairs + 1);
  b->d_state.place(base, o_state, sw);
  b->d_out.place(base, o_out, b->pairs + 1);
  b->res_ptr = base + (o_out - sizeof(unsigned long long));
  if (state_nresc(*b) * sizeof(uint32_t) + o_state != o_out - sizeof(unsigned long long)) return ACCG_ERR_BAD_ARG;   // layout invariant of the single D2H
  {   // the layout of a second set of pass buffers (allocated if the batch is run a second time): [flagged][jobs][redo][out64][state | out]
    size_t o2 = 0;
    auto take2 = [&](size_t bytes) { o2 = (o2 + 255) / 256 * 256; const size_t o = o2; o2 += bytes; return o; };
    b->off_alt_flagged = take2((b->rd.size() + 1) * sizeof(uint32_t));
    b->off_alt_jobs = take2(((size_t)b->rescue_off[PHMM_RESCUE_CLASSES] + 1) * sizeof(PhmmWork));
    b->off_alt_redo = take2(((size_t)b->rescue_off[PHMM_RESCUE_CLASSES] + 1) * sizeof(uint32_t));
    b->off_alt_out64 = take2((b->pairs + 1) * sizeof(double));
    b->off_alt_state = take2((sw + (sw & 1)) * sizeof(uint32_t) + (b->pairs + 1) * sizeof(float));
    b->off_alt_out = b->off_alt_state + (sw + (sw & 1)) * sizeof(uint32_t);
    b->alt_clear_bytes = o2 - b->off_alt_out64;          // out64, state, out: zero before the first use, like the first set
    b->pass_bytes = o2 + 256;
  }
  void* stage_v = nullptr;
  ACCG_HIP(ctx_stage(ctx, upload_bytes + 16, &stage_v));
  const auto tq2 = std::chrono::steady_clock::now();
  uint8_t* stage = (uint8_t*)stage_v;
#pragma omp parallel for schedule(static) num_threads(accg::host_threads()) if (n_regions >= 32 && accg::host_threads() > 1)
  for (int i = 0; i < n_regions; i++) {
    if (reads_bytes[i]) memcpy(stage + o_rblob + roffs[i], reads_ser[i], reads_bytes[i]);
    if (haps_bytes[i]) memcpy(stage + o_hblob + hoffs[i], haps_ser[i], haps_bytes[i]);
  }
  roff = roffs[n_regions]; hoff = hoffs[n_regions];
  memset(stage + o_rblob + roff, 0, 16); memset(stage + o_hblob + hoff, 0, 16);
  auto put = [&](size_t o, const auto& v) { if (!v.empty()) memcpy(stage + o, v.data(), vbytes(v)); };
  put(o_rd, b->rd); put(o_hp, b->hp); put(o_rd_out, b->rd_out); put(o_hp_local, b->hp_local);
  {
    PhmmHapDesc* hd = reinterpret_cast<PhmmHapDesc*>(stage + o_hap_ids);
    for (size_t i = 0; i < b->hap_ids.size(); i++) { const uint32_t g = b->hap_ids[i]; hd[i] = PhmmHapDesc{b->hp[g].off, b->hp[g].len, b->hp_local[g], g}; }
  }
  put(o_work, b->work); put(o_regions, b->regions_dev); put(o_chunks, b->chunks_dev); put(o_sorted, b->sorted_reads);
  put(o_row0, b->rd_row0); put(o_shape, b->rd_shape); put(o_streams, b->streams); put(o_spec_jobs, b->spec_jobs); put(o_spec_counts, b->spec_counts);
  memset(stage + o_streams + b->streams.size(), 0, 16);
  b->hp_ptr.clear(); b->hp_ptr.shrink_to_fit();            // the caller's blobs are not ours beyond this call
  const auto tq3 = std::chrono::steady_clock::now();
  // a small batch goes over by a copy kernel on its stream (no DMA engine in the chain: util_kernels.hip), which also notes the device's
  // wall clock at the start of the batch's device work (accg_phmm_region reports the device time from it)
  b->kernel_copies = upload_bytes + 16 <= ctx->kernel_copy_max && results_stage_bytes(b->pairs) <= ctx->kernel_copy_max;
  b->spec = b->spec && b->kernel_copies;
  if (upload_bytes && b->kernel_copies) ACCG_HIP(upload_by_kernel(stage, base, upload_bytes, reinterpret_cast<unsigned long long*>(b->d_clock.p) + 2, s));
  else if (upload_bytes) ACCG_HIP(hipMemcpyAsync(base, stage, upload_bytes, hipMemcpyHostToDevice, s));
  const auto tq4 = std::chrono::steady_clock::now();
  // out64, state, out start at zero: cleared by the kernel that writes the per-row records of the five-operation sweep (phmm_dev.h:
  // PhmmRowRecs, from the uploaded reads) when there is one -- a hipMemsetAsync of a megabyte or two costs the host 100 to 150 us
  const size_t clear_bytes = (o_out - o_out64) + (b->pairs + 1) * sizeof(float);       // (o_out64 is 256-byte aligned, sizes are multiples of 4)
  b->spec = b->spec && b->any_form5 && b->n_rows;
  if (b->spec) {
    // A speculative batch (run_spec): the fp32 half -- row records, then the sweep -- goes onto a side stream behind the upload, the
    // fp64 kernels take the main stream right behind the upload.  The fp64 values of every pair get written, so only the state words
    // and the fp32 results are cleared here (the fp64 kernels may be writing theirs at the same time).
    ACCG_HIP(ctx_need_aux(ctx));
    hipStream_t side = ctx->aux[accg_ctx::N_AUX - 1];
    ACCG_HIP(hipEventRecord(ctx->ev_fork, s));
    ACCG_HIP(hipStreamWaitEvent(side, ctx->ev_fork, 0));
    const PhmmArgs<float> pa = make_args<float>(*b, b->d_out.p, ctx->tab_f);
    ACCG_HIP(phmm_prepare_rows_launch(pa, (uint32_t)b->rd.size(), reinterpret_cast<uint32_t*>(base + o_state), (uint32_t)((clear_bytes - (o_state - o_out64)) / 4), side));
    ACCG_HIP(hipEventRecord(ctx->ev_join[accg_ctx::N_AUX - 1], side));      // (for a pass that is NOT speculative, e.g. strict mode: run_direct waits for it)
  } else if (b->any_form5 && b->n_rows && clear_bytes / 4 < (1ull << 32)) {
    const PhmmArgs<float> pa = make_args<float>(*b, b->d_out.p, ctx->tab_f);
    ACCG_HIP(phmm_prepare_rows_launch(pa, (uint32_t)b->rd.size(), reinterpret_cast<uint32_t*>(base + o_out64), (uint32_t)(clear_bytes / 4), s));
  } else {
    ACCG_HIP(hipMemsetAsync(base + o_out64, 0, clear_bytes, s));
  }
  if (!ctx->async_create) ACCG_HIP(hipStreamSynchronize(s));   // the staging buffer is reused by the next call
  if (getenv("ACCG_TRACE")) {
    auto us = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
    fprintf(stderr, "accg_phmm_batch_create: partition %.0f us, arena + upload %.0f us (arena of %zu MiB %.0f, staging %.0f, fill %.0f, copy of %zu KiB queued %.0f, clear + row records queued %.0f)\n",
            us(tp0, tp1), us(tp1, std::chrono::steady_clock::now()), (off + 256) >> 20, us(tp1, tq1), us(tq1, tq2), us(tq2, tq3), upload_bytes >> 10, us(tq3, tq4), us(tq4, std::chrono::steady_clock::now()));
  }
  sync_on_error.dismiss();
  *out = b.release();
  return ACCG_OK;
}

extern "C" uint64_t accg_phmm_batch_pairs(const accg_phmm_batch* b) { return b ? b->pairs : 0; }
extern "C" uint64_t accg_phmm_batch_cells(const accg_phmm_batch* b) { return b ? b->cells : 0; }
extern "C" uint64_t accg_phmm_batch_algorithmic_bytes(const accg_phmm_batch* b) { return b ? b->algo_bytes : 0; }
extern "C" uint64_t accg_phmm_batch_jobs(const accg_phmm_batch* b) { return b ? b->work.size() : 0; }

namespace {
bool graphs_wanted();
// Passes of one batch CAN be pipelined (ACCG_PHMM_PIPELINE=1; off by default): a pass's tail (planner, fp64 rescue, re-runs) goes
// onto the context's tail stream behind its sweep, and the NEXT pass's sweep starts behind this pass's sweep, on the other set of
// pass buffers; whoever needs a pass's results joins the tail first (join_tail).  Measured (tools/ab_step.py): configs[3] 10.89 ->
// 10.75 ms per pass, a 128-region shard 1.547 -> 1.523 -- and configs[1] 0.273 -> 0.321 ms: its sweep pins the CUs' LDS (eight
// workgroups of 20 KB), so the tail's workgroups (22 KB each, empty as they are) cannot start before the next sweep's workgroups
// retire, and the tail of pass i ends up holding back the sweep of pass i + 2.  Not worth 1.5 %: off.
bool pipeline_on() {
  const char* e = getenv("ACCG_PHMM_PIPELINE");          // read per pass: the tests run one batch both ways
  return e && e[0] == '1' && !graphs_wanted();
}
void swap_sets(accg_phmm_batch* b) {
  accg_phmm_batch::PassSet cur;
  cur.out = b->d_out.p; cur.out64 = b->d_out64.p; cur.state = b->d_state.p; cur.redo = b->d_redo.p; cur.flagged = b->d_flagged.p;
  cur.jobs = b->d_rescue_jobs.p; cur.res = b->res_ptr; cur.tail_done = b->tail_done; cur.tail_pending = b->tail_pending;
  b->d_out.p = b->alt.out; b->d_out64.p = b->alt.out64; b->d_state.p = b->alt.state; b->d_redo.p = b->alt.redo; b->d_flagged.p = b->alt.flagged;
  b->d_rescue_jobs.p = b->alt.jobs; b->res_ptr = b->alt.res; b->tail_done = b->alt.tail_done; b->tail_pending = b->alt.tail_pending;
  b->alt = cur;
}
int ensure_alt(accg_phmm_batch* b) {          // the second set of pass buffers, at the batch's second run
  if (b->d_alt.p) return ACCG_OK;
  PoolScope pool_scope(&b->ctx->pool);
  int st = b->d_alt.alloc(b->pass_bytes);
  if (st != ACCG_OK) return st;
  uint8_t* base = b->d_alt.p;
  ACCG_HIP(hipMemsetAsync(base + b->off_alt_out64, 0, b->alt_clear_bytes, b->ctx->stream));
  b->alt.out = (float*)(base + b->off_alt_out); b->alt.out64 = (double*)(base + b->off_alt_out64); b->alt.state = (uint32_t*)(base + b->off_alt_state);
  b->alt.redo = (uint32_t*)(base + b->off_alt_redo); b->alt.flagged = (uint32_t*)(base + b->off_alt_flagged); b->alt.jobs = (PhmmWork*)(base + b->off_alt_jobs);
  b->alt.res = base + b->off_alt_out - sizeof(unsigned long long);
  b->alt.tail_pending = false;
  ACCG_HIP(hipEventCreateWithFlags(&b->alt.tail_done, hipEventDisableTiming));
  return ACCG_OK;
}
// the main stream waits for every tail queued so far (the tail stream is in order: both sets' last tails)
int join_tail(accg_phmm_batch* b) {
  if (b->alt.tail_pending) { ACCG_HIP(hipStreamWaitEvent(b->ctx->stream, b->alt.tail_done, 0)); b->alt.tail_pending = false; }
  if (b->tail_pending) { ACCG_HIP(hipStreamWaitEvent(b->ctx->stream, b->tail_done, 0)); b->tail_pending = false; }
  return ACCG_OK;
}
// A one-shot batch small enough to leave most of the chip idle (accg_phmm_batch::spec): the fp64 values of EVERY pair are computed
// next to the fp32 sweep, on a forked stream, by the merged rescue kernels over jobs the host made at creation -- no planner, no
// fp64 launch behind the sweep.  A region of 2048 pairs: sweep 48 us + planner 7 + rescue 49 one behind the other become ~50 us side by
// side.  The values a caller gets are the same: a pair's fp64 result does not depend on which reads share its wavefront, and the
// results kernel picks (and counts) the pairs below MIN_ACCEPTED exactly as the rescue would have.
int run_spec(accg_phmm_batch* b) {
  accg_ctx* c = b->ctx;
  hipStream_t side = c->aux[accg_ctx::N_AUX - 1];            // (made at creation, where the row records were queued on it behind the upload)
  // the fp64 kernels: main stream, right behind the upload -- they are the longer half (sixteen lanes x K of 5 to 8 in fp64 against the
  // sweep's eight lanes x K up to 13 in fp32) and start without a cross-queue dependency
  PhmmArgs<double> a = make_args<double>(*b, b->d_out64.p, c->tab_d);
  a.work = b->d_spec_jobs.p; a.raw = nullptr; a.n_rescued = nullptr;
  a.stream_cap = b->rescue_stream_cap; a.haps_cap = b->rescue_haps_cap;
  a.job_count = nullptr; a.job_map = nullptr; a.is_redo = 0; a.redo_count = nullptr; a.redo_list = nullptr;
  PhmmRescueSet rs;
  for (int k = 0; k <= PHMM_RESCUE_CLASSES; k++) rs.off[k] = b->spec_off[k];
  rs.counts = b->d_spec_counts.p;
  for (int w = 0; w < 2; w++) {
    uint64_t units = 0; size_t lds = 0;
    for (int k = 0; k < PHMM_RESCUE_CLASSES; k++) {
      if (!b->spec_counts[(size_t)k] || phmm_rescue_window(k) != w) continue;
      int lpp_c, k_c;
      phmm_rescue_shape(k, &lpp_c, &k_c);
      units += b->spec_counts[(size_t)k];
      lds = std::max(lds, phmm_lds_bytes(k_c, 8, a.nchar, a.stream_cap, a.haps_cap, lpp_c, true, false, 1));
    }
    if (units) ACCG_HIP(phmm_launch_rescue_multi(w, 1, lds, a, rs, (uint32_t)units, c->stream));
  }
  // the fp32 sweep: side stream, behind the row records (launch_f32 queues on the context's stream: the side stream plays it for the call)
  std::swap(c->stream, side);
  const int st = launch_f32(b, ACCG_PHMM_FAST, nullptr, nullptr, true);
  std::swap(c->stream, side);
  if (st != ACCG_OK) return st;
  ACCG_HIP(hipEventRecord(c->ev_join[accg_ctx::N_AUX - 1], side));
  ACCG_HIP(hipStreamWaitEvent(c->stream, c->ev_join[accg_ctx::N_AUX - 1], 0));
  return ACCG_OK;
}
int run_direct(accg_phmm_batch* b, int mode, hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr) {
  int st;
  if (b->spec && mode == ACCG_PHMM_FAST && !ev_begin) { b->spec_ran = true; return run_spec(b); }
  b->spec_ran = false;
  if (b->spec) ACCG_HIP(hipStreamWaitEvent(b->ctx->stream, b->ctx->ev_join[accg_ctx::N_AUX - 1], 0));     // the side stream's row records and clears
  if (!pipeline_on()) {
    if ((st = launch_f32(b, mode, ev_begin, ev_end)) != ACCG_OK) return st;
    return launch_rescue(b, mode);
  }
  accg_ctx* c = b->ctx;
  ACCG_HIP(ctx_need_tail(c));
  if (b->runs >= 1) {
    if ((st = ensure_alt(b)) != ACCG_OK) return st;
    swap_sets(b);
  }
  b->runs++;
  if (!b->ev_sweep) ACCG_HIP(hipEventCreateWithFlags(&b->ev_sweep, hipEventDisableTiming));
  if (!b->tail_done) ACCG_HIP(hipEventCreateWithFlags(&b->tail_done, hipEventDisableTiming));
  // this set's previous pass (two passes ago): its tail read the flags and results this sweep is about to overwrite
  if (b->tail_pending) { ACCG_HIP(hipStreamWaitEvent(c->stream, b->tail_done, 0)); b->tail_pending = false; }
  if ((st = launch_f32(b, mode, ev_begin, ev_end)) != ACCG_OK) return st;
  ACCG_HIP(hipEventRecord(b->ev_sweep, c->stream));
  ACCG_HIP(hipStreamWaitEvent(c->tail, b->ev_sweep, 0));
  if ((st = launch_rescue(b, mode, true)) != ACCG_OK) return st;
  ACCG_HIP(hipEventRecord(b->tail_done, c->tail));
  b->tail_pending = true;
  return ACCG_OK;
}
bool graphs_wanted() {
  // Off unless ACCG_PHMM_GRAPH=1.  Measured on MI355X (tools/ab_step.py, round 3): the replayed graph is SLOWER than the plain
  // stream launches it was captured from -- configs[1] 0.367 against 0.362 ms per step, a 128-region configs[3] shard 2.61 against
  // 2.46 ms -- the host is far ahead of the device either way, and the graph's nodes keep the same kernel-to-kernel dependencies.
  static const bool on = [] { const char* e = getenv("ACCG_PHMM_GRAPH"); return e && e[0] == '1'; }();
  return on;
}
// The pass as a graph: the context's stream is captured (thread-local mode: other threads' HIP calls are not affected) while
// the same launch code runs; the forked aux streams join the capture through the fork event and leave it through the join events.
int run_graph(accg_phmm_batch* b, int mode) {
  const int gi = mode == ACCG_PHMM_STRICT ? 1 : 0;
  hipStream_t s = b->ctx->stream;
  if (!b->graph_exec[gi]) {
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); b->graph_off = true; return run_direct(b, mode); }
    const int st = run_direct(b, mode);       // (pipeline_on() is false when graphs are wanted: one stream, one set)
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (st != ACCG_OK || e != hipSuccess || !g) {
      if (g) hipGraphDestroy(g);
      (void)hipGetLastError();
      b->graph_off = true;
      return st != ACCG_OK ? st : run_direct(b, mode);
    }
    const hipError_t ei = hipGraphInstantiate(&b->graph_exec[gi], g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (ei != hipSuccess) { (void)hipGetLastError(); b->graph_exec[gi] = nullptr; b->graph_off = true; return run_direct(b, mode); }
  }
  ACCG_HIP(hipGraphLaunch(b->graph_exec[gi], s));
  return ACCG_OK;
}
int run_pass(accg_phmm_batch* b, int mode) { return (b->graph_off || !graphs_wanted()) ? run_direct(b, mode) : run_graph(b, mode); }
}  // namespace

extern "C" int accg_phmm_batch_run(accg_phmm_batch* b, int mode) {
  if (!b) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  return run_pass(b, mode);
}

// fp64 over every pair (FalconPairHMM::computePairhmmAVX with use_double = true, FalconPairHMM.cpp:82);
// results land in the fp64 buffer and are fetched with accg_phmm_batch_results_f64.
extern "C" int accg_phmm_batch_run_f64(accg_phmm_batch* b) {
  if (!b) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  { const int stj = join_tail(b); if (stj != ACCG_OK) return stj; }
  PhmmArgs<double> a = make_args<double>(*b, b->d_out64.p, b->ctx->tab_d);
  for (const KLaunch& l : b->launches) {
    a.stream_cap = l.stream_cap; a.haps_cap = l.haps_cap;
    ACCG_HIP(phmm_launch_f64(l.K, l.lpp, l.striped, a, l.work0, l.n_work, b->ctx->stream));
  }
  return ACCG_OK;
}
extern "C" int accg_phmm_batch_results_f64(accg_phmm_batch* b, double* out_raw64) {
  if (!b || !out_raw64) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  ACCG_HIP(hipStreamSynchronize(b->ctx->stream));
  if (b->pairs) ACCG_HIP(hipMemcpy(out_raw64, b->d_out64.p, b->pairs * sizeof(double), hipMemcpyDeviceToHost));
  return ACCG_OK;
}

// what: 0 = whole run (fp32 pass + rescue pass), 1 = fp32 pass only (the dominant kernel)
extern "C" int accg_phmm_batch_time2(accg_phmm_batch* b, int mode, int what, int warmup, int iters, float* ms_per_run) {
  if (!b || !ms_per_run || iters <= 0 || warmup < 0) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  accg_ctx* c = b->ctx;
  int st;
  if ((st = join_tail(b)) != ACCG_OK) return st;
  for (int i = 0; i < warmup; i++) {
    if ((st = what == 0 ? run_pass(b, mode) : launch_f32(b, mode)) != ACCG_OK) return st;
  }
  if ((st = join_tail(b)) != ACCG_OK) return st;
  ACCG_HIP(hipEventRecord(c->ev0, c->stream));
  for (int i = 0; i < iters; i++) {
    if ((st = what == 0 ? run_pass(b, mode) : launch_f32(b, mode)) != ACCG_OK) return st;
  }
  if ((st = join_tail(b)) != ACCG_OK) return st;          // the last passes' tails belong to the time
  ACCG_HIP(hipEventRecord(c->ev1, c->stream));
  ACCG_HIP(hipEventSynchronize(c->ev1));
  float ms = 0;
  ACCG_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *ms_per_run = ms / iters;
  b->last_kernel_ns = (uint64_t)((double)ms / iters * 1e6);
  if (what != 0) { if ((st = launch_rescue(b, mode)) != ACCG_OK) return st; }   // leave the buffers consistent
  return ACCG_OK;
}
// `iters` whole passes back to back (plain stream launches, the same kernels in the same order as accg_phmm_batch_run), each with a
// pair of events around its fp32 sweep launches on the stream they are launched on: the dominant kernel timed INSIDE the step,
// in the clock state the steps run in.  kernel_ms = mean time between those events (one kernel for a single-class batch, the
// forked classes otherwise), step_ms = mean time of a whole pass.
// In three parts, so that a caller who brackets the passes with a clock of its own (bench.py's timed region) has nothing but the passes
// inside the bracket: _reserve makes the events (before), _run queues `iters` passes with their events and returns without waiting,
// _times reads the events (after the caller's own wait).
extern "C" int accg_phmm_batch_steps_reserve(accg_phmm_batch* b, int iters) {
  if (!b || iters <= 0 || iters > 4096) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  while (b->step_ev.size() < 2 * (size_t)iters + 2) {
    hipEvent_t e = nullptr;
    ACCG_HIP(hipEventCreate(&e));
    b->step_ev.push_back(e);
  }
  return ACCG_OK;
}
extern "C" int accg_phmm_batch_steps_run(accg_phmm_batch* b, int mode, int iters) {
  int st = accg_phmm_batch_steps_reserve(b, iters);
  if (st != ACCG_OK) return st;
  accg_ctx* c = b->ctx;
  std::vector<hipEvent_t>& ev = b->step_ev;
  if ((st = join_tail(b)) != ACCG_OK) return st;
  ACCG_HIP(hipEventRecord(ev[0], c->stream));
  for (int i = 0; i < iters; i++) {
    if ((st = run_direct(b, mode, ev[2 + 2 * i], ev[3 + 2 * i])) != ACCG_OK) return st;
  }
  if ((st = join_tail(b)) != ACCG_OK) return st;
  ACCG_HIP(hipEventRecord(ev[1], c->stream));
  b->steps_queued = iters;
  return ACCG_OK;
}
extern "C" int accg_phmm_batch_steps_times(accg_phmm_batch* b, float* kernel_ms, float* step_ms) {
  if (!b || !kernel_ms || !step_ms || b->steps_queued <= 0) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  const int iters = b->steps_queued;
  std::vector<hipEvent_t>& ev = b->step_ev;
  ACCG_HIP(hipEventSynchronize(ev[1]));
  float ms = 0;
  ACCG_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
  *step_ms = ms / iters;
  double sum = 0;
  for (int i = 0; i < iters; i++) { float k = 0; ACCG_HIP(hipEventElapsedTime(&k, ev[2 + 2 * i], ev[3 + 2 * i])); sum += k; }
  *kernel_ms = (float)(sum / iters);
  b->last_kernel_ns = (uint64_t)((double)ms / iters * 1e6);
  return ACCG_OK;
}
extern "C" int accg_phmm_batch_time_in_step(accg_phmm_batch* b, int mode, int iters, float* kernel_ms, float* step_ms) {
  if (!b || !kernel_ms || !step_ms || iters <= 0 || iters > 4096) return ACCG_ERR_BAD_ARG;
  const int st = accg_phmm_batch_steps_run(b, mode, iters);
  return st != ACCG_OK ? st : accg_phmm_batch_steps_times(b, kernel_ms, step_ms);
}
// The shader clock the device held while the first wavefront of the last sweep launch ran its job (s_memtime ticks over 100 MHz
// wall-clock ticks, both taken by that wavefront): the clock UNDER the kernel, not that of an idle or lightly loaded card.
extern "C" int accg_phmm_batch_clock_ghz(accg_phmm_batch* b, float* ghz) {
  if (!b || !ghz) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  ACCG_HIP(hipStreamSynchronize(b->ctx->stream));
  uint64_t h[2] = {0, 0};
  ACCG_HIP(hipMemcpy(h, b->d_clock.p, sizeof h, hipMemcpyDeviceToHost));
  int wall_khz = 0;
  ACCG_HIP(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, b->ctx->device));
  if (wall_khz <= 0) wall_khz = 100000;
  *ghz = h[1] ? (float)((double)h[0] / (double)h[1] * (double)wall_khz * 1e-6) : 0.f;
  return ACCG_OK;
}
// The per-row records of the five-operation sweep (phmm_prepare_rows) are written once, at batch creation -- a pure function of the
// reads, like the haplotype streams the host lays out -- so a timed pass over a device-resident batch does not contain them.  This
// times that kernel by itself (mean of `iters` launches on the context's stream), for a line that wants to state it next to its step.
extern "C" int accg_phmm_batch_time_prepare(accg_phmm_batch* b, int iters, float* ms_per_run) {
  if (!b || !ms_per_run || iters <= 0) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  *ms_per_run = 0.f;
  if (!b->any_form5 || !b->n_rows) return ACCG_OK;
  accg_ctx* c = b->ctx;
  { const int stj = join_tail(b); if (stj != ACCG_OK) return stj; }
  const PhmmArgs<float> pa = make_args<float>(*b, b->d_out.p, c->tab_f);
  ACCG_HIP(phmm_prepare_rows_launch(pa, (uint32_t)b->rd.size(), nullptr, 0, c->stream));
  ACCG_HIP(hipEventRecord(c->ev0, c->stream));
  for (int i = 0; i < iters; i++) ACCG_HIP(phmm_prepare_rows_launch(pa, (uint32_t)b->rd.size(), nullptr, 0, c->stream));
  ACCG_HIP(hipEventRecord(c->ev1, c->stream));
  ACCG_HIP(hipEventSynchronize(c->ev1));
  float ms = 0;
  ACCG_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *ms_per_run = ms / iters;
  return ACCG_OK;
}
extern "C" int accg_phmm_batch_time(accg_phmm_batch* b, int mode, int warmup, int iters, float* ms_per_run) {
  return accg_phmm_batch_time2(b, mode, 0, warmup, iters, ms_per_run);
}

namespace {
// results of a pass in two halves for the ring: the device-to-host copies queued behind the kernels (both result arrays, always),
// and the host half -- wait, log10 (same libm as the reference) -- when the caller comes for them
// Where one region's share of a pass's results goes (a batch's results are concatenated in region order)
// (r64: nullable; the fp64 values of the region's pairs for a caller that takes the log10 itself -- filled only when the pass rescued
// something, *resc = the region's rescued pairs)
struct ResultDst { float* raw; double* l10; accg_counters* cnt; uint64_t pairs, cells; double* r64 = nullptr; uint64_t* resc = nullptr; };
// The downloads of a ticket: queued behind its kernels at once (plain ring: one caller thread, nothing else submits meanwhile), or
// -- threaded ring, `late` -- only when the kernels have finished: a copy that sits in a DMA queue waiting for a kernel blocks the
// NEXT copy submitted to that queue, another slot's upload, inside hipMemcpyAsync on that slot's worker (measured: creations of
// 1.7 ms took 7 to 11 ms while the device worked through the tickets ahead).  A small batch (kernel_copies) sends its results by a
// copy kernel instead, which is queued at once either way: it involves no DMA queue.
int results_enqueue(accg_phmm_batch* b, bool late) {
  { const int stj = join_tail(b); if (stj != ACCG_OK) return stj; }
  const size_t n = b->pairs, head = sizeof(unsigned long long) + n * sizeof(float), off64 = results_off64(n);
  void* stage_v = nullptr;
  ACCG_HIP(ctx_stage(b->ctx, results_stage_bytes(n), &stage_v));      // (sized by the submitter before the upload: no reallocation here)
  uint8_t* stage = (uint8_t*)stage_v;
  if (b->kernel_copies) {
    b->results_late = false; b->results_fetched = true;
    if (b->spec_ran) ACCG_HIP(phmm_results_spec_by_kernel(b->d_out.p, b->d_out64.p, n, stage, off64, reinterpret_cast<unsigned long long*>(b->d_clock.p) + 2, b->ctx->stream));
    else ACCG_HIP(phmm_results_by_kernel(b->res_ptr, b->d_out64.p, n, stage, off64, reinterpret_cast<unsigned long long*>(b->d_clock.p) + 2, b->ctx->stream));
    return ACCG_OK;
  }
  b->results_late = late; b->results_fetched = false;
  if (late) return ACCG_OK;
  memset(stage, 0, sizeof(unsigned long long));
  ACCG_HIP(hipMemcpyAsync(stage + sizeof(unsigned long long), b->res_ptr, head, hipMemcpyDeviceToHost, b->ctx->stream));
  if (n) ACCG_HIP(hipMemcpyAsync(stage + off64, b->d_out64.p, n * sizeof(double), hipMemcpyDeviceToHost, b->ctx->stream));
  b->results_fetched = true;
  return ACCG_OK;
}
// waits for the pass, fetches what results_enqueue has not, and hands every region's share to its destination (raw copy, log10 as
// FalconPairHMM.cpp:83-90 / PairHMMWorker.cpp:176-190 take it, counters)
int results_finish(accg_phmm_batch* b, const ResultDst* dst, size_t n_dst) {
  ACCG_HIP(hipStreamSynchronize(b->ctx->stream));
  const size_t n = b->pairs, head = sizeof(unsigned long long) + n * sizeof(float), off64 = results_off64(n);
  uint8_t* stage = (uint8_t*)b->ctx->h_stage;
  if (!b->results_fetched) {
    memset(stage, 0, sizeof(unsigned long long));
    ACCG_HIP(hipMemcpyAsync(stage + sizeof(unsigned long long), b->res_ptr, head, hipMemcpyDeviceToHost, b->ctx->stream));
    ACCG_HIP(hipStreamSynchronize(b->ctx->stream));
  }
  if (b->timed_by_events) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, b->ctx->ev0, b->ctx->ev1) == hipSuccess) b->last_kernel_ns = (uint64_t)((double)ms * 1e6);
    b->timed_by_events = false;
  }
  unsigned long long nresc = 0, ticks = 0;
  memcpy(&ticks, stage, sizeof ticks);
  memcpy(&nresc, stage + sizeof ticks, sizeof nresc);
  if (b->kernel_copies && ticks) b->last_kernel_ns = (uint64_t)((double)ticks * 1e6 / (double)b->ctx->wall_khz);
  b->ctx->spec_hint = nresc != 0;
  const float* raw = (const float*)(stage + RES_HDR);
  const double* r64 = (const double*)(stage + off64);
  bool want_l10 = false;
  for (size_t d = 0; d < n_dst; d++) want_l10 |= dst[d].l10 != nullptr || dst[d].r64 != nullptr;
  if (want_l10 && n && nresc && !b->results_fetched) {              // the fp64 values only when something was rescued
    ACCG_HIP(hipMemcpyAsync(stage + off64, b->d_out64.p, n * sizeof(double), hipMemcpyDeviceToHost, b->ctx->stream));
    ACCG_HIP(hipStreamSynchronize(b->ctx->stream));
  }
  b->results_fetched = true;
  const HostTables& t = host_tables();
  size_t o = 0;
  for (size_t d = 0; d < n_dst; d++) {
    const ResultDst& D = dst[d];
    const size_t m = (size_t)D.pairs;
    if (o + m > n) return ACCG_ERR_BAD_ARG;
    if (D.raw && m) memcpy(D.raw, raw + o, m * sizeof(float));
    if (D.l10 && m) {
      double* out_log10 = D.l10; const float* rw = raw + o; const double* rd = r64 + o;
      // (a log10f per pair is 10 ns: half a millisecond for a ticket of 32 configs[3] regions on one thread)
#pragma omp parallel for schedule(static) num_threads(accg::host_threads()) if (m >= 16384 && accg::host_threads() > 1)
      for (size_t i = 0; i < m; i++)
        out_log10[i] = rw[i] < PHMM_MIN_ACCEPTED ? log10(rd[i]) - t.log10_init_d : (double)(log10f(rw[i]) - t.log10_init_f);
    }
    uint64_t resc = nresc;
    if (n_dst > 1 && (D.cnt || D.resc)) { resc = 0; for (size_t i = 0; i < m; i++) resc += raw[o + i] < PHMM_MIN_ACCEPTED; }      // this region's share
    if (D.r64 && m && nresc) memcpy(D.r64, r64 + o, m * sizeof(double));
    if (D.resc) *D.resc = resc;
    if (D.cnt) { D.cnt->cells = D.cells; D.cnt->pairs = m; D.cnt->kernel_ns = b->last_kernel_ns; D.cnt->rescued = resc; }
    o += m;
  }
  return ACCG_OK;
}
int results_finish(accg_phmm_batch* b, float* out_raw, double* out_log10, accg_counters* cnt) {
  const ResultDst d{out_raw, out_log10, cnt, b->pairs, b->cells};
  return results_finish(b, &d, 1);
}
}  // namespace

extern "C" int accg_phmm_batch_results(accg_phmm_batch* b, float* out_raw, double* out_log10, accg_counters* cnt) {
  if (!b) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  { const int stj = join_tail(b); if (stj != ACCG_OK) return stj; }
  // (a device-resident batch that is run many times: the results are fetched when they are asked for, the fp64 values only when
  // something was rescued)
  void* stage_v = nullptr;
  ACCG_HIP(ctx_stage(b->ctx, results_stage_bytes(b->pairs), &stage_v));
  b->results_late = true; b->results_fetched = false;
  const uint64_t keep_ns = b->last_kernel_ns;
  const bool kc = b->kernel_copies;
  b->kernel_copies = false;            // (the device clock note belongs to creation, not to this pass)
  const int st = results_finish(b, out_raw, out_log10, cnt);
  b->kernel_copies = kc; b->last_kernel_ns = keep_ns;
  return st;
}

extern "C" void accg_phmm_batch_destroy(accg_phmm_batch* b) {
  if (!b) return;
  hipSetDevice(b->ctx->device);
  hipStreamSynchronize(b->ctx->stream);
  if (b->ctx->tail && (b->tail_pending || b->alt.tail_pending || b->runs)) hipStreamSynchronize(b->ctx->tail);
  for (hipGraphExec_t& g : b->graph_exec) if (g) { hipGraphExecDestroy(g); g = nullptr; }
  for (hipEvent_t& e : b->ev_probe) if (e) { hipEventDestroy(e); e = nullptr; }
  for (hipEvent_t e : b->step_ev) if (e) hipEventDestroy(e);
  if (b->ev_sweep) hipEventDestroy(b->ev_sweep);
  if (b->tail_done) hipEventDestroy(b->tail_done);
  if (b->alt.tail_done) hipEventDestroy(b->alt.tail_done);
  b->d_alt.release();
  b->d_arena.release();
  delete b;
}

// One region entirely in fp64 (the reference's compute_fp_avxd per pair / use_double = true): raw likelihood x 2^1020.
extern "C" int accg_phmm_region_f64(accg_ctx* ctx, const void* reads_ser, size_t reads_bytes, const void* haps_ser, size_t haps_bytes,
                                    double* out_raw64) {
  accg_phmm_batch* b = nullptr;
  const void* rs[1] = {reads_ser}; const void* hs[1] = {haps_ser};
  size_t rb[1] = {reads_bytes}, hb[1] = {haps_bytes};
  int st = accg_phmm_batch_create(ctx, 1, rs, rb, hs, hb, &b);
  if (st != ACCG_OK) return st;
  st = accg_phmm_batch_run_f64(b);
  if (st == ACCG_OK) st = accg_phmm_batch_results_f64(b, out_raw64);
  accg_phmm_batch_destroy(b);
  return st;
}

namespace {
// pairs and blob bytes of regions as their headers announce them (the staging is sized from these before anything is queued)
int peek_regions(int n_regions, const void* const* reads_ser, const size_t* reads_bytes, const void* const* haps_ser, const size_t* haps_bytes,
                 uint64_t* pairs, size_t* blob) {
  *pairs = 0; *blob = 0;
  for (int i = 0; i < n_regions; i++) {
    if (reads_bytes[i] < 4 || haps_bytes[i] < 4 || !reads_ser[i] || !haps_ser[i]) return ACCG_ERR_BAD_WIRE;
    int32_t nr = 0, nh = 0;
    memcpy(&nr, reads_ser[i], 4); memcpy(&nh, haps_ser[i], 4);
    if (nr < 0 || nh < 0) return ACCG_ERR_BAD_WIRE;
    *pairs += (uint64_t)nr * (uint64_t)nh; *blob += reads_bytes[i] + haps_bytes[i];
  }
  return ACCG_OK;
}
// One blocking pass over regions handed over together: create (no wait for the upload: the staging block is sized up front and not
// touched again before the results are in), run, results -- ONE wait for the device in the whole call.
int region_call(accg_ctx* ctx, int n_regions, const void* const* rs, const size_t* rb, const void* const* hs, const size_t* hb, int mode,
                ResultDst* dst_in, size_t n_dst, float* out_raw, double* out_log10, accg_counters* cnt, const PhmmParsed* const* pre = nullptr,
                bool alone = true) {
  static const bool trace = getenv("ACCG_TRACE") != nullptr;     // stage timings of the one-shot path on stderr
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  if (!ctx) return ACCG_ERR_NOT_INITIALISED;
  uint64_t pairs = 0; size_t blob = 0;
  int st = peek_regions(n_regions, rs, rb, hs, hb, &pairs, &blob);
  if (st != ACCG_OK) return st;
  ACCG_HIP(hipSetDevice(ctx->device));
  void* stage = nullptr;
  ACCG_HIP(ctx_stage(ctx, std::max(results_stage_bytes(pairs), 3 * blob + ((size_t)1 << 20)), &stage));
  accg_phmm_batch* b = nullptr;
  const bool was_async = ctx->async_create;
  ctx->async_create = true; ctx->oneshot = true; ctx->alone = alone;
  st = phmm_batch_create_impl(ctx, n_regions, rs, rb, hs, hb, pre, &b);
  ctx->async_create = was_async; ctx->oneshot = false;
  if (st != ACCG_OK) return st;
  const auto t1 = clk::now();
  const bool events = !b->kernel_copies;           // a small batch times itself on the device's wall clock (results_finish)
  if (events) hipEventRecord(ctx->ev0, ctx->stream);
  b->graph_off = true;            // a single pass: capturing and instantiating a graph would cost more than it saves
  st = accg_phmm_batch_run(b, mode);
  if (st == ACCG_OK) st = join_tail(b);
  if (events) hipEventRecord(ctx->ev1, ctx->stream);
  if (st == ACCG_OK) st = results_enqueue(b, false);
  auto t2 = clk::now(), t3 = t2;
  if (st == ACCG_OK) {
    if (events && hipEventSynchronize(ctx->ev1) == hipSuccess) {
      float ms = 0; hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
      b->last_kernel_ns = (uint64_t)((double)ms * 1e6);
    }
    ResultDst one{out_raw, out_log10, cnt, b->pairs, b->cells};
    if (!dst_in) { dst_in = &one; n_dst = 1; }
    else if (n_dst == b->regions.size()) {          // one destination per region: its pairs and cells
      for (size_t d = 0; d < n_dst; d++) {
        const Region& r = b->regions[d];
        uint64_t rsum = 0, hsum = 0;
        for (uint32_t k = 0; k < r.n_reads; k++) rsum += b->rd[r.read0 + k].len;
        for (uint32_t k = 0; k < r.n_haps; k++) hsum += b->hp[r.hap0 + k].len;
        dst_in[d].pairs = (uint64_t)r.n_reads * r.n_haps; dst_in[d].cells = rsum * hsum;
      }
    }
    if (trace) { hipStreamSynchronize(ctx->stream); t2 = clk::now(); }
    st = results_finish(b, dst_in, n_dst);
    t3 = clk::now();
  }
  const uint64_t kernel_ns = b->last_kernel_ns;
  accg_phmm_batch_destroy(b);
  if (trace) {
    auto us = [](clk::time_point a, clk::time_point c) { return std::chrono::duration<double, std::micro>(c - a).count(); };
    fprintf(stderr, "accg_phmm_region: %llu pairs: create %.0f us, run+wait %.0f us (device %.0f us), results %.0f us, destroy %.0f us\n",
            (unsigned long long)pairs, us(t0, t1), us(t1, t2), kernel_ns / 1e3, us(t2, t3), us(t3, clk::now()));
  }
  return st;
}
}  // namespace

extern "C" int accg_phmm_region(accg_ctx* ctx, const void* reads_ser, size_t reads_bytes, const void* haps_ser,
                                size_t haps_bytes, int mode, float* out_raw, double* out_log10, accg_counters* cnt) {
  const void* rs[1] = {reads_ser}; const void* hs[1] = {haps_ser};
  size_t rb[1] = {reads_bytes}, hb[1] = {haps_bytes};
  return region_call(ctx, 1, rs, rb, hs, hb, mode, nullptr, 0, out_raw, out_log10, cnt);
}


// ---- ring of regions in flight ------------------------------------------------------------------------------------------------
// The one-shot call above is a latency chain: parse and size the jobs, upload, kernels, download, log10 -- the device idles while the
// host works and the host while the device does.  A ring keeps up to `slots` regions in flight on contexts of their own (stream,
// device block cache, pinned staging): accg_phmm_ring_submit does the host half of region i and queues its upload, kernels and
// downloads without waiting; accg_phmm_ring_wait fetches the results of the oldest one -- while region i computes, region i + 1
// is being parsed and region i - 1 read back.  compute_fpga / FalconPairHMM::computePairhmm keep their blocking signatures on
// top of accg_phmm_region (pairhmm/host/PairHMMFpga.h:16-20); a caller that owns the loop over active regions uses the ring.
// A threaded ring (accg_phmm_ring_create_threaded) moves that host half off the caller too: submit hands the pointers to the slot's
// worker thread and returns, so the host halves of up to `slots` tickets run side by side, each on ONE host thread (a stream of
// configs[3] regions is bound by exactly that: some 60 us of parsing, job sizing and staging per region on one thread against 11 us
// of device time; eight workers keep the device busy).
struct RingSlot {
  std::thread th;                 // lives as long as the ring
  std::mutex mu;
  std::condition_variable cv;
  bool has_work = false, done = false, quit = false;
  int n_regions = 0, mode = 0;
  uint64_t pairs = 0; size_t blob = 0;
  int status = ACCG_OK;
  std::string err;
  std::vector<const void*> rs, hs;
  std::vector<size_t> rb, hb;
  std::vector<float> raw;         // the ticket's results, finished (download waited for, log10 taken) by the worker
  std::vector<double> l10;
  accg_counters cnt{};
  bool busy = false;              // a ticket is outstanding (caller's side)
};
struct accg_phmm_ring {
  std::vector<accg_ctx*> ctx;
  std::vector<accg_phmm_batch*> batch;       // per slot: the region in flight (null: free)
  std::vector<std::unique_ptr<RingSlot>> slot;   // threaded rings
  bool threaded = false;
  uint64_t next_ticket = 0;
};

static int ring_submit_body(accg_phmm_ring* r, size_t slot, int n_regions, const void* const* reads_ser, const size_t* reads_bytes,
                            const void* const* haps_ser, const size_t* haps_bytes, int mode, uint64_t pairs, size_t blob);
static void ring_worker(accg_phmm_ring* r, size_t slot) {
  RingSlot& S = *r->slot[slot];
  // the worker's share of the host's threads: with few slots each worker keeps a small team for the parallel loops of batch creation,
  // with as many slots as threads the workers ARE the parallelism (ACCG_RING_WORKER_THREADS overrides)
  {
    const char* e = getenv("ACCG_RING_WORKER_THREADS");
    const int all = host_threads();
    tls_host_threads = e && atoi(e) > 0 ? atoi(e) : std::max(1, all / (int)r->ctx.size());
    omp_short_blocktime();      // (per thread: see accg_init)
  }
  for (;;) {
    std::unique_lock<std::mutex> lk(S.mu);
    S.cv.wait(lk, [&] { return S.has_work || S.quit; });
    if (S.quit) return;
    S.has_work = false;
    lk.unlock();
    int st = ring_submit_body(r, slot, S.n_regions, S.rs.data(), S.rb.data(), S.hs.data(), S.hb.data(), S.mode, S.pairs, S.blob);
    if (st == ACCG_OK) {        // ... and the device half's end: wait for the downloads, take the log10 -- off the caller's thread too
      accg_phmm_batch* b = r->batch[slot];
      S.raw.resize(b->pairs); S.l10.resize(b->pairs);
      const auto tf0 = std::chrono::steady_clock::now();
      st = results_finish(b, S.raw.data(), S.l10.data(), &S.cnt);
      const auto tf1 = std::chrono::steady_clock::now();
      accg_phmm_batch_destroy(b);
      if (getenv("ACCG_TRACE")) fprintf(stderr, "ring slot %zu: finish %.0f us, destroy %.0f us\n", slot, std::chrono::duration<double, std::micro>(tf1 - tf0).count(),
                                        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tf1).count());
      r->batch[slot] = nullptr;
    }
    lk.lock();
    S.status = st;
    if (st != ACCG_OK) S.err = accg_last_hip_error();
    S.done = true;
    lk.unlock();
    S.cv.notify_all();
  }
}
static int ring_create(accg_ctx* ctx, int slots, bool threaded, accg_phmm_ring** out) {
  if (!ctx) return ACCG_ERR_NOT_INITIALISED;
  if (!out || slots < 1 || slots > 64) return ACCG_ERR_BAD_ARG;
  *out = nullptr;
  std::unique_ptr<accg_phmm_ring> r(new accg_phmm_ring);
  for (int i = 0; i < slots; i++) {
    accg_ctx* c = nullptr;
    const int st = accg_init(ctx->device, &c);
    if (st != ACCG_OK) { for (accg_ctx* x : r->ctx) accg_shutdown(x); return st; }
    c->async_create = true;
    r->ctx.push_back(c);
  }
  r->batch.assign((size_t)slots, nullptr);
  r->threaded = threaded;
  if (threaded) {
    accg_phmm_ring* rp = r.get();
    for (int i = 0; i < slots; i++) r->slot.emplace_back(new RingSlot);
    try {
      for (int i = 0; i < slots; i++) r->slot[(size_t)i]->th = std::thread(ring_worker, rp, (size_t)i);
    } catch (...) {
      accg_phmm_ring_destroy(r.release());
      return ACCG_ERR_BAD_ARG;
    }
  }
  *out = r.release();
  return ACCG_OK;
}
extern "C" int accg_phmm_ring_create(accg_ctx* ctx, int slots, accg_phmm_ring** out) { return ring_create(ctx, slots, false, out); }
extern "C" int accg_phmm_ring_create_threaded(accg_ctx* ctx, int slots, accg_phmm_ring** out) { return ring_create(ctx, slots, true, out); }

// the host half of a ticket and everything it queues: on the caller's thread (plain ring) or on the slot's worker
static int ring_submit_body(accg_phmm_ring* r, size_t slot, int n_regions, const void* const* reads_ser, const size_t* reads_bytes,
                            const void* const* haps_ser, const size_t* haps_bytes, int mode, uint64_t pairs, size_t blob) {
  accg_ctx* c = r->ctx[slot];
  ACCG_HIP(hipSetDevice(c->device));
  // pinned staging large enough for the upload AND the downloads, before anything is queued (it must not move in between)
  void* stage = nullptr;
  ACCG_HIP(ctx_stage(c, std::max(results_stage_bytes(pairs), 3 * blob + ((size_t)1 << 20)), &stage));
  accg_phmm_batch* b = nullptr;
  static const bool trace = getenv("ACCG_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  int st = accg_phmm_batch_create(c, n_regions, reads_ser, reads_bytes, haps_ser, haps_bytes, &b);
  if (st != ACCG_OK) return st;
  const auto t1 = std::chrono::steady_clock::now();
  b->graph_off = true;
  // device time of the ticket (accg_counters::kernel_ns, what the reference prints its GCUPS from: FalconPairHMM.cpp:1214-1220): a small
  // batch times itself on the device's wall clock (results_finish), a large one between two events on the slot's stream
  const bool events = !b->kernel_copies;
  if (events) ACCG_HIP(hipEventRecord(c->ev0, c->stream));
  st = accg_phmm_batch_run(b, mode);
  if (st == ACCG_OK) st = join_tail(b);
  if (events && st == ACCG_OK) { ACCG_HIP(hipEventRecord(c->ev1, c->stream)); b->timed_by_events = true; }
  const auto t2 = std::chrono::steady_clock::now();
  if (st == ACCG_OK) st = results_enqueue(b, r->threaded);
  if (st != ACCG_OK) { accg_phmm_batch_destroy(b); return st; }
  if (trace) {
    auto us = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
    static const auto epoch = std::chrono::steady_clock::now();
    fprintf(stderr, "ring slot %zu: at %.0f us: create %.0f us, run (enqueue) %.0f us, results (enqueue) %.0f us\n", slot, us(epoch, t0), us(t0, t1), us(t1, t2), us(t2, std::chrono::steady_clock::now()));
  }
  r->batch[slot] = b;
  return ACCG_OK;
}

extern "C" int accg_phmm_ring_submit_many(accg_phmm_ring* r, int n_regions, const void* const* reads_ser, const size_t* reads_bytes,
                                          const void* const* haps_ser, const size_t* haps_bytes, int mode, uint64_t* ticket) {
  if (!r || !ticket || n_regions < 1 || !reads_ser || !reads_bytes || !haps_ser || !haps_bytes) return ACCG_ERR_BAD_ARG;
  const size_t slot = (size_t)(r->next_ticket % r->ctx.size());
  // the slot's previous ticket has not been waited for (a threaded slot's batch pointer belongs to its worker while a ticket is out)
  if (r->threaded) { std::lock_guard<std::mutex> lk(r->slot[slot]->mu); if (r->slot[slot]->busy) return ACCG_ERR_BAD_ARG; }
  else if (r->batch[slot] != nullptr) return ACCG_ERR_BAD_ARG;
  uint64_t pairs = 0; size_t blob = 0;
  for (int i = 0; i < n_regions; i++) {
    if (reads_bytes[i] < 4 || haps_bytes[i] < 4 || !reads_ser[i] || !haps_ser[i]) return ACCG_ERR_BAD_WIRE;
    int32_t nr = 0, nh = 0;
    memcpy(&nr, reads_ser[i], 4); memcpy(&nh, haps_ser[i], 4);
    if (nr < 0 || nh < 0) return ACCG_ERR_BAD_WIRE;
    pairs += (uint64_t)nr * (uint64_t)nh; blob += reads_bytes[i] + haps_bytes[i];
  }
  if (!r->threaded) {
    const int st = ring_submit_body(r, slot, n_regions, reads_ser, reads_bytes, haps_ser, haps_bytes, mode, pairs, blob);
    if (st != ACCG_OK) return st;
  } else {
    RingSlot& S = *r->slot[slot];
    {
      std::lock_guard<std::mutex> lk(S.mu);
      S.rs.assign(reads_ser, reads_ser + n_regions); S.hs.assign(haps_ser, haps_ser + n_regions);
      S.rb.assign(reads_bytes, reads_bytes + n_regions); S.hb.assign(haps_bytes, haps_bytes + n_regions);
      S.n_regions = n_regions; S.mode = mode; S.pairs = pairs; S.blob = blob;
      S.status = ACCG_OK; S.err.clear(); S.done = false; S.has_work = true;
      S.busy = true;
    }
    S.cv.notify_all();
  }
  *ticket = r->next_ticket++;
  return ACCG_OK;
}
extern "C" int accg_phmm_ring_submit(accg_phmm_ring* r, const void* reads_ser, size_t reads_bytes, const void* haps_ser, size_t haps_bytes,
                                     int mode, uint64_t* ticket) {
  const void* rs[1] = {reads_ser}; const void* hs[1] = {haps_ser};
  size_t rb[1] = {reads_bytes}, hb[1] = {haps_bytes};
  return accg_phmm_ring_submit_many(r, 1, rs, rb, hs, hb, mode, ticket);
}

extern "C" int accg_phmm_ring_wait(accg_phmm_ring* r, uint64_t ticket, float* out_raw, double* out_log10, accg_counters* cnt) {
  if (!r || ticket >= r->next_ticket || ticket + r->ctx.size() < r->next_ticket) return ACCG_ERR_BAD_ARG;
  const size_t slot = (size_t)(ticket % r->ctx.size());
  if (r->threaded) {
    RingSlot& S = *r->slot[slot];
    {
      std::unique_lock<std::mutex> lk(S.mu);
      if (!S.busy) return ACCG_ERR_BAD_ARG;                   // waited for already
      S.cv.wait(lk, [&] { return S.done; });
      S.busy = false;
    }
    if (S.status != ACCG_OK) { set_error_text(S.err.c_str()); return S.status; }
    if (out_raw && !S.raw.empty()) memcpy(out_raw, S.raw.data(), S.raw.size() * sizeof(float));
    if (out_log10 && !S.l10.empty()) memcpy(out_log10, S.l10.data(), S.l10.size() * sizeof(double));
    if (cnt) *cnt = S.cnt;
    return ACCG_OK;
  }
  accg_phmm_batch* b = r->batch[slot];
  if (!b) return ACCG_ERR_BAD_ARG;                            // waited for already
  ACCG_HIP(hipSetDevice(b->ctx->device));
  const int st = results_finish(b, out_raw, out_log10, cnt);
  accg_phmm_batch_destroy(b);
  r->batch[slot] = nullptr;
  return st;
}

extern "C" void accg_phmm_ring_destroy(accg_phmm_ring* r) {
  if (!r) return;
  for (auto& sp : r->slot) {
    RingSlot& S = *sp;
    { std::unique_lock<std::mutex> lk(S.mu); if (S.busy) S.cv.wait(lk, [&] { return S.done; }); }     // a ticket nobody waited for
    { std::lock_guard<std::mutex> lk(S.mu); S.quit = true; }
    S.cv.notify_all();
    if (S.th.joinable()) S.th.join();
  }
  for (accg_phmm_batch* b : r->batch) if (b) accg_phmm_batch_destroy(b);
  for (accg_ctx* c : r->ctx) accg_shutdown(c);
  delete r;
}

// ---- regions from concurrent blocking callers -------------------------------------------------------------------------------------
// The reference's callers hand over ONE region per blocking call (compute_fpga, pairhmm/host/PairHMMFpga.cpp:125-162;
// FalconPairHMM::computePairhmm, pairhmm/xlnx/host/FalconPairHMM.cpp:1184-1193) and an accelerator manager runs one PairHMM task per
// request, several at a time (pairhmm/task/xlnx/PairHMMTask.cpp:27-143).  One region is 256 wavefront jobs: a twentieth of the chip for
// some 100 us, and the device runs the kernels of different streams one or two at a time (rocprofv3 --kernel-trace of sixteen caller
// threads with a context each: two hardware queues busy, never more than two kernels at once).  So concurrent callers are COMBINED: a
// caller parses its own region, queues it, and -- if one of the mux's few contexts ("lanes") is free -- becomes the leader of everything
// queued so far: one device batch (merged launches, one upload, one results block), whose results it hands back to the callers it took
// along; they take their own log10.  A caller that arrives while all lanes are busy waits for a leader to take it, or for a lane.
// Under load the batches grow by themselves; a lone caller runs its region at once, on its own thread, with no hand-off.
namespace {
struct MuxReq {
  const void* rs; size_t rb; const void* hs; size_t hb; int mode;
  PhmmParsed parsed;
  uint64_t pairs = 0;
  float* raw = nullptr;                 // the caller's out_raw, or this request's own buffer
  std::vector<float> raw_own;
  std::vector<double> r64;              // fp64 values of the pairs (filled when the batch rescued something)
  uint64_t resc = 0;
  accg_counters cnt{};
  int status = ACCG_OK; std::string err;
  bool all5 = true;                     // every read passes the five-operation form's range tests
  bool taken = false, done = false;
  std::condition_variable cv;
};
}  // namespace
struct accg_phmm_mux {
  std::mutex mu;
  std::deque<MuxReq*> q;
  std::vector<accg_ctx*> lanes;
  std::vector<int> free_lanes;
  int max_regions = 64;
  uint64_t max_pairs = 1u << 20;
  uint64_t batches = 0, regions = 0;    // statistics (accg_phmm_mux_stats)
};

extern "C" int accg_phmm_mux_create(int device, int lanes, int max_regions, accg_phmm_mux** out) {
  if (!out || lanes < 1 || lanes > 16 || max_regions < 1) return ACCG_ERR_BAD_ARG;
  *out = nullptr;
  std::unique_ptr<accg_phmm_mux> m(new accg_phmm_mux);
  m->max_regions = max_regions;
  for (int i = 0; i < lanes; i++) {
    accg_ctx* c = nullptr;
    const int st = accg_init(device, &c);
    if (st != ACCG_OK) { for (accg_ctx* x : m->lanes) accg_shutdown(x); return st; }
    m->lanes.push_back(c);
    m->free_lanes.push_back(i);
  }
  *out = m.release();
  return ACCG_OK;
}
extern "C" void accg_phmm_mux_destroy(accg_phmm_mux* m) {
  if (!m) return;
  for (accg_ctx* c : m->lanes) accg_shutdown(c);
  delete m;
}
extern "C" void accg_phmm_mux_stats(accg_phmm_mux* m, uint64_t* batches, uint64_t* regions) {
  if (!m) return;
  std::lock_guard<std::mutex> g(m->mu);
  if (batches) *batches = m->batches;
  if (regions) *regions = m->regions;
}

namespace {
// a leader's batch on its lane; n == 1 or a failed batch: every request on its own, so that an error lands on the region that caused it
void mux_run(accg_phmm_mux* m, int lane, std::vector<MuxReq*>& batch, bool alone) {
  accg_ctx* c = m->lanes[(size_t)lane];
  const size_t n = batch.size();
  // No OpenMP team under a leader: leaders are whatever caller threads come by, and each would become the root of a thread pool of its
  // own (sixteen callers leading in turn: sixteen pools spinning against each other -- a batch of eight regions took 3.4 ms).  The
  // callers ARE the host parallelism here: each has parsed its own region and takes its own log10.
  struct Threads { int prev; explicit Threads(int t) : prev(tls_host_threads) { tls_host_threads = t; } ~Threads() { tls_host_threads = prev; } } threads(1);
  auto run = [&](MuxReq* const* reqs, size_t k) {
    std::vector<const void*> rs(k), hs(k); std::vector<size_t> rb(k), hb(k); std::vector<const PhmmParsed*> pre(k); std::vector<ResultDst> dst(k);
    for (size_t i = 0; i < k; i++) {
      MuxReq& R = *reqs[i];
      rs[i] = R.rs; rb[i] = R.rb; hs[i] = R.hs; hb[i] = R.hb; pre[i] = &R.parsed;
      dst[i] = ResultDst{R.raw, nullptr, &R.cnt, R.pairs, 0, R.r64.data(), &R.resc};
    }
    return region_call(c, (int)k, rs.data(), rb.data(), hs.data(), hb.data(), reqs[0]->mode, dst.data(), k, nullptr, nullptr, nullptr, pre.data(), alone);
  };
  int st = run(batch.data(), n);
  if (st != ACCG_OK && n > 1) {
    for (MuxReq* R : batch) { R->status = run(&R, 1); if (R->status != ACCG_OK) R->err = accg_last_hip_error(); }
    return;
  }
  for (MuxReq* R : batch) { R->status = st; if (st != ACCG_OK) R->err = accg_last_hip_error(); }
}
}  // namespace

extern "C" int accg_phmm_mux_region(accg_phmm_mux* m, const void* reads_ser, size_t reads_bytes, const void* haps_ser, size_t haps_bytes, int mode,
                                    float* out_raw, double* out_log10, accg_counters* counters) {
  if (!m) return ACCG_ERR_NOT_INITIALISED;
  if (!reads_ser || !haps_ser) return ACCG_ERR_BAD_ARG;
  MuxReq me;
  me.rs = reads_ser; me.rb = reads_bytes; me.hs = haps_ser; me.hb = haps_bytes; me.mode = mode;
  // this caller's share of the host work, on its own thread: parse, validate, range tests
  parse_region(reads_ser, reads_bytes, haps_ser, haps_bytes, me.parsed);
  if (me.parsed.nr < 0) return me.parsed.nr;
  if (me.parsed.nh < 0) return me.parsed.nh;
  me.pairs = (uint64_t)me.parsed.nr * (uint64_t)me.parsed.nh;
  for (uint8_t f : me.parsed.form) me.all5 &= f == 5;
  if (out_raw) me.raw = out_raw; else { me.raw_own.resize((size_t)me.pairs); me.raw = me.raw_own.data(); }
  if (out_log10) me.r64.resize((size_t)me.pairs);
  {
    std::unique_lock<std::mutex> lk(m->mu);
    m->q.push_back(&me);
    while (!me.done) {
      if (me.taken || m->free_lanes.empty()) { me.cv.wait(lk); continue; }
      // lead: everything queued in this request's mode, in arrival order, this request included
      // (the lowest free lane: a lone caller always runs on lane 0, whose streams were made first and sit on hardware queues of their own)
      const auto lo = std::min_element(m->free_lanes.begin(), m->free_lanes.end());
      const int lane = *lo;
      m->free_lanes.erase(lo);
      std::vector<MuxReq*> batch;
      uint64_t pairs = 0;
      for (auto it = m->q.begin(); it != m->q.end() && (int)batch.size() < m->max_regions;) {
        MuxReq* R = *it;
        // (one arithmetic mode per batch; and a batch's fp64 rescue runs the five-operation form only if ALL its reads pass that form's
        // range tests, so regions are only combined with regions of the same kind: a region's bits do not depend on its company)
        if (R->mode == me.mode && R->all5 == me.all5 && (batch.empty() || pairs + R->pairs <= m->max_pairs)) {
          R->taken = true; batch.push_back(R); pairs += R->pairs; it = m->q.erase(it);
        } else ++it;
      }
      m->batches++; m->regions += batch.size();
      // (the speculative fp64 pass of a small batch, run_spec, is for a caller that has the device to itself: ten times the fp64 work
      // and a second hardware queue are well spent on an idle chip and wasted on a shared one)
      const bool alone = m->free_lanes.size() + 1 == m->lanes.size() && m->q.empty();
      lk.unlock();
      mux_run(m, lane, batch, alone);
      lk.lock();
      for (MuxReq* R : batch) { R->done = true; if (R != &me) R->cv.notify_one(); }
      m->free_lanes.push_back(lane);
      for (MuxReq* R : m->q) R->cv.notify_one();              // whoever arrived meanwhile: the first to wake up leads the next batch
    }
  }
  if (me.status != ACCG_OK) { set_error_text(me.err.c_str()); return me.status; }
  if (out_log10) {
    const HostTables& t = host_tables();
    const float* rw = me.raw; const double* rd = me.r64.data();
    for (size_t i = 0; i < (size_t)me.pairs; i++)     // FalconPairHMM.cpp:83-90 / PairHMMWorker.cpp:176-190
      out_log10[i] = rw[i] < PHMM_MIN_ACCEPTED ? log10(rd[i]) - t.log10_init_d : (double)(log10f(rw[i]) - t.log10_init_f);
  }
  if (counters) *counters = me.cnt;
  return ACCG_OK;
}

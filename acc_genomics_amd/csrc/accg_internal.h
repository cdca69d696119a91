// Internal host-side declarations shared by the translation units of libaccg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/accg.h"
#include "phmm_dev.h"

namespace accg {

void set_hip_error(hipError_t e, const char* what);
void set_error_text(const char* text);      // same slot as set_hip_error (accg_last_hip_error)

#define ACCG_HIP(call)                                                         \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) { ::accg::set_hip_error(e_, #call); return ACCG_ERR_HIP; } \
  } while (0)

// Host copies of the Context<T> tables (pairhmm/xlnx/host/Context.h) + the derived per-quality
// factors the kernels read.
struct HostTables {
  float ph_f[128], omph_f[128], phd3_f[128], m2m_f[8256], init_f, log10_init_f;
  double ph_d[128], omph_d[128], phd3_d[128], m2m_d[8256], init_d, log10_init_d;
};
const HostTables& host_tables();

// Device memory cache of a context.  The one-shot entry points (accg_phmm_region = one call per active region of the caller)
// build and drop a whole batch per call; hipMalloc/hipFree of its ~17 buffers cost more than the kernels of a small region.
// Blocks are reused in stream order (everything runs on ctx->stream or on streams joined back to it), so a block handed out
// again is only touched after the work that used it before.  Large blocks (> 256 MiB) bypass the cache.
struct DevPool {
  std::mutex mu;
  std::multimap<size_t, void*> free_;
  std::unordered_map<void*, size_t> live_;
  size_t cached = 0;
  hipError_t get(size_t bytes, void** p);
  void put(void* p);
  void drain();
};

// Threads for the host-side post-processing loops: the CPUs this process may really use (cgroup quota when there is one,
// else the affinity mask), ACCG_HOST_THREADS overrides.  OpenMP's default is every CPU it can see, which under a quota
// just gets the process throttled.
int host_threads();

}  // namespace accg

struct accg_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  accg::PhmmTables<float> tab_f{};
  accg::PhmmTables<double> tab_d{};
  void* tab_mem = nullptr;      // one allocation behind both table sets
  char name[128] = {0};
  int n_cu = 0;
  accg::DevPool pool;
  // pinned host staging (grown on demand): one H2D per batch_create, one D2H per results call.  A context is driven by one
  // host thread at a time (it owns one stream), so the staging needs no lock.
  void* h_stage = nullptr;
  size_t h_stage_bytes = 0;
  // a slot of an accg_phmm_ring: batch creation does not wait for its upload (the staging block is the slot's own and is not
  // touched again before the slot's results have been fetched)
  bool async_create = false;
  // set for the duration of a blocking region call (accg_phmm_region, a mux leader's batch): the batch is created for ONE pass whose
  // latency is what counts -- a batch small enough to leave most of the chip idle computes its fp64 values speculatively next to
  // the fp32 sweep instead of behind it (phmm_host.cpp: run_spec)
  // the job-sizing decision of the PairHMM batch created before on this context (phmm_host.cpp: partition)
  struct SizingMemo { bool valid = false, pairs = false, dom5 = false; uint64_t budget = 0, regions = 0, med_hap = 0, haps = 0; int K_dom = 0, lpp_dom = 0, nchar = 0; } sizing;
  // a page of host-visible words for device-written flags (made at first use; phmm_host.cpp: the rescue probe of a batch that is run repeatedly)
  uint32_t* h_flags = nullptr;
  uint32_t flag_next = 0;
  bool oneshot = false;
  bool spec_hint = false;       // the last pass whose results were fetched on this context rescued something (phmm_host.cpp: results_finish)
  bool alone = true;            // ... and no other caller is known to share the device right now (a mux tells: false while other lanes are busy)
  int wall_khz = 100000;        // rate of the device's constant wall clock (wall_clock64): 100 MHz on every gfx9
  // uploads and result blocks up to this size travel by copy kernels on the stream instead of hipMemcpyAsync (ACCG_COPY_KERNEL_MAX, bytes; 0 = never)
  // (2 MiB.  Measured at 512 KiB: a lone configs[1] region call, whose 1.5 MB upload then goes through the DMA engine, 0.53-0.56 against
  // 0.55-0.68 ms -- but sixteen callers through a mux 54 against 37 us per region: their batches of four regions and more fall back
  // to DMA copies, which queue behind each other across the lanes)
  size_t kernel_copy_max = 2u << 20;
  // Independent kernels of one pass (one launch per rows-per-lane class) are spread over these streams, forked from and
  // joined back to `stream`: queued on one stream each launch would wait for the previous one's last wavefront.
  static constexpr int N_AUX = 4;
  hipStream_t aux[N_AUX] = {};
  hipEvent_t ev_fork = nullptr, ev_join[N_AUX] = {};
  // The tail of a PairHMM pass (rescue planner, fp64 rescue, strict re-runs) runs on a stream of its own with its own forked streams,
  // so that the next pass's sweep can start behind this pass's sweep instead of behind its tail (phmm_host.cpp: run_direct).
  hipStream_t tail = nullptr, aux_t[N_AUX] = {};
  hipEvent_t ev_fork_t = nullptr, ev_join_t[N_AUX] = {};
};

namespace accg {
// Declared right after the owning unique_ptr in a *_create function: an early (error) return drains the context's stream before
// that owner's destructor hands device blocks back, so no copy queued by the failed call is still writing into them.
struct SyncOnError {
  hipStream_t s; bool armed = true;
  explicit SyncOnError(hipStream_t st) : s(st) {}
  void dismiss() { armed = false; }
  ~SyncOnError() { if (armed) (void)hipStreamSynchronize(s); }
};
// aux streams wait for everything queued on ctx->stream so far / ctx->stream waits for everything queued on the aux streams
hipError_t ctx_need_aux(accg_ctx* c);         // the forked streams and their events exist from here on (made at first use)
hipError_t ctx_need_tail(accg_ctx* c);
hipError_t ctx_fork(accg_ctx* c);
hipError_t ctx_fork_tail(accg_ctx* c);       // the same for the tail stream and its forked streams
hipError_t ctx_join_tail(accg_ctx* c);
hipError_t ctx_stage(accg_ctx* c, size_t bytes, void** p);   // pinned staging of at least `bytes`
hipError_t ctx_join(accg_ctx* c);
// transfers of a small batch as kernels on its stream (util_kernels.hip); host_pinned / stage_pinned: the context's staging block
hipError_t upload_by_kernel(const void* host_pinned, void* dev, size_t bytes, unsigned long long* tick, hipStream_t s);
hipError_t phmm_results_spec_by_kernel(const float* raw, const double* out64, size_t n, void* stage_pinned, size_t off64_bytes, const unsigned long long* tick, hipStream_t s);
hipError_t phmm_results_by_kernel(const void* res, const double* out64, size_t n, void* stage_pinned, size_t off64_bytes, const unsigned long long* tick, hipStream_t s);
}  // namespace accg

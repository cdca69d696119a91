// Multi-GPU counters: the one collective of the path.
//
// Pairs are independent, so ranks never exchange data; what the reference adds up across its compute dies are the
// throughput figures it prints (cells and kernel time per die, pairhmm/xlnx/host/FalconPairHMM.cpp:1214-1220, after splitting
// the batch in proportion to cell counts, :169-249).  Here one process drives one GPU and the same totals come from an RCCL
// all-reduce over xGMI of uint64[4] {cells, pairs, kernel_ns, rescued} (sum) and of the wall time (max).
//
// librccl is opened at run time: a single-GPU user of libaccg_hip.so needs no RCCL, and a caller that asks for a communicator
// without it gets ACCG_ERR_NO_RCCL instead of a silent single-rank answer.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>
#include "accg_internal.h"

using namespace accg;

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;     // why the library could not be opened
};
Rccl g_rccl;

// nullptr when librccl cannot be opened; the reason is kept for accg_last_hip_error()
Rccl* rccl() {
  Rccl& r = g_rccl;
  static std::once_flag once;
  std::call_once(once, [&r] {
    // ACCG_RCCL_LIB names THE library to use (no search behind it: a caller that points at a particular build must not
    // silently get the system's instead); unset: the usual names
    const char* forced = getenv("ACCG_RCCL_LIB");
    const bool only_forced = forced && *forced;
    const char* names[] = {forced, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      if (only_forced && n != forced) break;
      r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
      r.why = dlerror();
    }
    if (!r.so) return;
    auto sym = [&](const char* s) { void* p = dlsym(r.so, s); if (!p) { r.why = std::string("missing symbol ") + s; } return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) { dlclose(r.so); r.so = nullptr; }
  });
  return r.so ? &r : nullptr;
}

int rccl_fail(Rccl* R, ncclResult_t e, const char* what) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", what, R && R->GetErrorString ? R->GetErrorString(e) : "rccl error");
  set_error_text(buf);
  return ACCG_ERR_RCCL;
}
#define ACCG_NCCL(R, call)                                         \
  do {                                                             \
    ncclResult_t e_ = (call);                                      \
    if (e_ != ncclSuccess) return rccl_fail(R, e_, #call);         \
  } while (0)

}  // namespace

struct accg_comm {
  accg_ctx* ctx = nullptr;
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;     // null for a world of one without ACCG_COMM_FORCE_RCCL
  uint64_t* d_buf = nullptr;     // [4] counters in, [4] counters out, [1] wall in (double), [1] wall out
  uint64_t* h_buf = nullptr;     // pinned mirror
};

// librccl can be opened and has the five entry points this file calls (no device needed): lets every rank of a job find out,
// before any of them enters the collective ncclCommInitRank, whether all of them can.
extern "C" int accg_comm_available(void) {
  if (rccl()) return ACCG_OK;
  set_error_text(("librccl not available: " + g_rccl.why).c_str());
  return ACCG_ERR_NO_RCCL;
}

extern "C" int accg_comm_unique_id(void* id) {
  if (!id) return ACCG_ERR_BAD_ARG;
  Rccl* R = rccl();
  if (!R) { set_error_text(("librccl not available: " + g_rccl.why).c_str()); return ACCG_ERR_NO_RCCL; }
  static_assert(sizeof(ncclUniqueId) == ACCG_COMM_ID_BYTES, "accg.h states the size of ncclUniqueId");
  ncclUniqueId u;
  ACCG_NCCL(R, R->GetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return ACCG_OK;
}

extern "C" int accg_comm_init(accg_ctx* ctx, int rank, int world, const void* id, accg_comm** out) {
  if (!ctx) return ACCG_ERR_NOT_INITIALISED;
  if (!out || world < 1 || rank < 0 || rank >= world) return ACCG_ERR_BAD_ARG;
  *out = nullptr;
  ACCG_HIP(hipSetDevice(ctx->device));
  accg_comm* c = new accg_comm;
  c->ctx = ctx; c->rank = rank; c->world = world;
  const bool force = getenv("ACCG_COMM_FORCE_RCCL") != nullptr;    // rehearses the RCCL calls with a world of one
  int st = ACCG_OK;
  if (world > 1 || force) {
    Rccl* R = rccl();
    if (!R) { set_error_text(("librccl not available: " + g_rccl.why).c_str()); delete c; return ACCG_ERR_NO_RCCL; }
    ncclUniqueId u;
    if (id) memcpy(&u, id, sizeof u);
    else if (world == 1) { ncclResult_t e = R->GetUniqueId(&u); if (e != ncclSuccess) { delete c; return rccl_fail(R, e, "ncclGetUniqueId"); } }
    else { delete c; return ACCG_ERR_BAD_ARG; }
    ncclResult_t e = R->CommInitRank(&c->nccl, world, u, rank);
    if (e != ncclSuccess) { delete c; return rccl_fail(R, e, "ncclCommInitRank"); }
  }
  hipError_t he = hipMalloc((void**)&c->d_buf, 10 * sizeof(uint64_t));
  if (he == hipSuccess) he = hipHostMalloc((void**)&c->h_buf, 10 * sizeof(uint64_t), hipHostMallocDefault);
  if (he != hipSuccess) { set_hip_error(he, "accg_comm_init buffers"); st = ACCG_ERR_HIP; }
  if (st != ACCG_OK) { accg_comm_destroy(c); return st; }
  *out = c;
  return ACCG_OK;
}

extern "C" int accg_comm_rank(const accg_comm* c) { return c ? c->rank : -1; }
extern "C" int accg_comm_world(const accg_comm* c) { return c ? c->world : 0; }
extern "C" int accg_comm_uses_rccl(const accg_comm* c) { return c && c->nccl ? 1 : 0; }

extern "C" int accg_counters_allreduce(accg_comm* c, const accg_counters* mine, double wall_s, accg_counters* total, double* wall_max) {
  if (!c || !mine) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(c->ctx->device));
  hipStream_t s = c->ctx->stream;
  uint64_t* h = c->h_buf;
  accg_counters_pack(mine, h);
  memcpy(h + 8, &wall_s, sizeof(double));
  if (c->nccl) {
    Rccl* R = rccl();
    ACCG_HIP(hipMemcpyAsync(c->d_buf, h, 4 * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    ACCG_HIP(hipMemcpyAsync(c->d_buf + 8, h + 8, sizeof(double), hipMemcpyHostToDevice, s));
    ACCG_NCCL(R, R->AllReduce(c->d_buf, c->d_buf + 4, 4, ncclUint64, ncclSum, c->nccl, s));
    ACCG_NCCL(R, R->AllReduce(c->d_buf + 8, c->d_buf + 9, 1, ncclFloat64, ncclMax, c->nccl, s));
    ACCG_HIP(hipMemcpyAsync(h + 4, c->d_buf + 4, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    ACCG_HIP(hipMemcpyAsync(h + 9, c->d_buf + 9, sizeof(double), hipMemcpyDeviceToHost, s));
    ACCG_HIP(hipStreamSynchronize(s));
  } else {
    ACCG_HIP(hipStreamSynchronize(s));      // a world of one: the barrier meaning (stream drained) still holds
    memcpy(h + 4, h, 4 * sizeof(uint64_t));
    h[9] = h[8];
  }
  if (total) { total->cells = h[4]; total->pairs = h[5]; total->kernel_ns = h[6]; total->rescued = h[7]; }
  if (wall_max) memcpy(wall_max, h + 9, sizeof(double));
  return ACCG_OK;
}

// Everything queued on the context's stream has finished on this rank, and every rank has got here.
extern "C" int accg_comm_barrier(accg_comm* c) {
  if (!c) return ACCG_ERR_BAD_ARG;
  accg_counters z = {0, 0, 0, 0};
  return accg_counters_allreduce(c, &z, 0.0, nullptr, nullptr);
}

extern "C" void accg_comm_destroy(accg_comm* c) {
  if (!c) return;
  hipSetDevice(c->ctx->device);
  hipStreamSynchronize(c->ctx->stream);
  if (c->nccl) { if (Rccl* R = rccl()) R->CommDestroy(c->nccl); }
  if (c->d_buf) hipFree(c->d_buf);
  if (c->h_buf) hipHostFree(c->h_buf);
  delete c;
}

extern "C" int accg_ctx_synchronize(accg_ctx* ctx) {
  if (!ctx) return ACCG_ERR_NOT_INITIALISED;
  ACCG_HIP(hipSetDevice(ctx->device));
  ACCG_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->tail) ACCG_HIP(hipStreamSynchronize(ctx->tail));     // the tails of PairHMM passes (queued behind their sweeps)
  return ACCG_OK;
}

extern "C" int accg_ctx_trim(accg_ctx* ctx) {
  if (!ctx) return ACCG_ERR_NOT_INITIALISED;
  ACCG_HIP(hipSetDevice(ctx->device));
  ACCG_HIP(hipStreamSynchronize(ctx->stream));
  ctx->pool.drain();
  if (ctx->h_stage) { ACCG_HIP(hipHostFree(ctx->h_stage)); ctx->h_stage = nullptr; ctx->h_stage_bytes = 0; }
  return ACCG_OK;
}

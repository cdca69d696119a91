// Host side of the SMEM seeding path: index upload (the role of ocl_init, smem/host/ocl.cpp:169-293) and read batches
// (smem_ocl / ocl_kernel_invoke, :296-418).
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <memory>
#include <vector>
#include "accg_internal.h"
#include "smem_dev.h"

using namespace accg;

struct accg_smem_index {
  accg_ctx* ctx = nullptr;
  uint32_t* d_bwt = nullptr;
  uint4* d_ktab = nullptr;        // prefix table (smem_dev.h), compact layout only
  uint64_t words = 0, primary = 0, L2[5] = {0};
  bool compact = false;
  ~accg_smem_index() { if (d_bwt) hipFree(d_bwt); if (d_ktab) hipFree(d_ktab); }
};

struct accg_smem_batch {
  accg_smem_index* idx = nullptr;
  uint32_t n = 0, stride = 0, max_out = 0, slice = 0;
  uint8_t *d_seq = nullptr, *d_len = nullptr;
  SmemIntv *d_out = nullptr, *d_scratch = nullptr;
  uint32_t* d_queue = nullptr;
  uint32_t *d_seg = nullptr, *d_nseg = nullptr;   // three-kernel path: segment table of the first pass
  uint32_t engine_waves = 0;      // > 0: persistent-wavefront engine (smem_kernel.hip), compact index only
  int32_t* d_num = nullptr;
  SmemIntv* d_out3 = nullptr; int32_t *d_num3 = nullptr, *d_num12 = nullptr;     // third pass on a stream of its own (ACCG_SMEM_PASS3_ASIDE): its intervals, SMEM_PASS3_SLOTS per read
  uint64_t bases = 0;
  uint32_t longest = 0;         // longest read of the batch (sizes the kernels' LDS rows)
  ~accg_smem_batch() { for (void* p : {(void*)d_seq, (void*)d_len, (void*)d_out, (void*)d_num, (void*)d_scratch, (void*)d_queue, (void*)d_seg, (void*)d_nseg, (void*)d_out3, (void*)d_num3, (void*)d_num12}) if (p) hipFree(p); }
};

extern "C" int accg_smem_index_create(accg_ctx* ctx, const uint32_t* bwt, uint64_t bwt_words, const uint64_t* bwt_para,
                                      accg_smem_index** out) {
  if (!ctx) return ACCG_ERR_NOT_INITIALISED;
  if (!out || !bwt || !bwt_para || bwt_words == 0) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(ctx->device));
  // The reference uploads exactly bwt_size words (smem/host/ocl.cpp:214-224) and a BWA index is generally not a whole number
  // of 16-word blocks: a ragged tail is completed with zero words here, the caller's buffer is never read beyond its end.
  std::vector<uint32_t> padded;
  if (bwt_words % 16) {
    padded.assign((bwt_words + 15) / 16 * 16, 0u);
    memcpy(padded.data(), bwt, bwt_words * 4);
    bwt = padded.data();
    bwt_words = padded.size();
  }
  std::unique_ptr<accg_smem_index> x(new accg_smem_index);
  x->ctx = ctx; x->words = bwt_words; x->primary = bwt_para[0];
  for (int c = 0; c < 5; c++) x->L2[c] = bwt_para[1 + c];
  // BWA's block (64 B per 128 symbols: 4 x u64 cumulative counts + 8 x 16 symbols) costs a lookup four 16-byte loads and
  // eight words of popcounts.  While the counts fit 32 bits the index is re-laid-out on upload into half-blocks of the
  // same total size, 32 B per 64 symbols = 4 x u32 counts (cumulated from base 3 down) + the symbols' low and high bits as two
  // 64-bit planes: one sector, two loads, two masked popcounts per lookup.
  // The counts a lookup returns are the same numbers.  ACCG_SMEM_COMPACT=0 keeps the caller's layout.
  const uint64_t n_blocks = bwt_words / 16;
  const char* ec = getenv("ACCG_SMEM_COMPACT");
  x->compact = !(ec && ec[0] == '0') && n_blocks * 128 < (1ull << 32);
  ACCG_HIP(hipMalloc((void**)&x->d_bwt, bwt_words * 4));
  if (x->compact) {
    std::vector<uint32_t> cb(bwt_words);
    uint32_t T[4];
    T[3] = (uint32_t)x->L2[3] + 1u;
    for (int j = 2; j >= 0; j--) T[j] = T[j + 1] + (uint32_t)x->L2[j] + 1u;
#pragma omp parallel for schedule(static) num_threads(accg::host_threads())
    for (int64_t b = 0; b < (int64_t)n_blocks; b++) {
      const uint32_t* src = bwt + b * 16;
      uint32_t* d0 = cb.data() + b * 16;
      uint32_t* d1 = d0 + 8;
      uint32_t c[4];
      for (int s = 0; s < 4; s++) c[s] = src[2 * s];              // low words of the u64 counts (high words are 0 here)
      for (int half = 0; half < 2; half++) {
        uint32_t* d = half ? d1 : d0;
        // cumulated from the top: symbols >= j in front of the half-block, plus T[j] = sum over b >= j of (L2[b] + 1) (smem_kernel.hip, count_ge)
        d[3] = c[3] + T[3]; d[2] = c[2] + c[3] + T[2]; d[1] = c[1] + c[2] + c[3] + T[1]; d[0] = c[0] + c[1] + c[2] + c[3] + T[0];
        uint64_t lo = 0, hi = 0;                                   // bit planes: symbol p of the half-block at bit p
        for (int j = 0; j < 4; j++) {
          const uint32_t v = src[8 + 4 * half + j];                // 16 symbols, first in the top bits
          for (int e = 0; e < 16; e++) {
            const uint32_t sym = (v >> (30 - 2 * e)) & 3u;
            lo |= (uint64_t)(sym & 1u) << (16 * j + e);
            hi |= (uint64_t)(sym >> 1) << (16 * j + e);
            c[sym]++;
          }
        }
        d[4] = (uint32_t)lo; d[5] = (uint32_t)(lo >> 32); d[6] = (uint32_t)hi; d[7] = (uint32_t)(hi >> 32);
      }
    }
    ACCG_HIP(hipMemcpy(x->d_bwt, cb.data(), bwt_words * 4, hipMemcpyHostToDevice));
    const char* ek = getenv("ACCG_SMEM_KTAB");           // A/B knob: 0 = no prefix table
    if (!(ek && ek[0] == '0')) {
      ACCG_HIP(hipMalloc((void**)&x->d_ktab, (size_t)SMEM_KTAB_ENTRIES * sizeof(uint4)));
      SmemArgs a{};
      a.bwt = x->d_bwt; a.primary = x->primary; a.compact = 1u;
      for (int c = 0; c < 5; c++) a.L2[c] = x->L2[c];
      ACCG_HIP(smem_build_ktab(a, x->d_ktab, ctx->stream));
      ACCG_HIP(hipStreamSynchronize(ctx->stream));
    }
  } else {
    ACCG_HIP(hipMemcpy(x->d_bwt, bwt, bwt_words * 4, hipMemcpyHostToDevice));
  }
  *out = x.release();
  return ACCG_OK;
}
extern "C" void accg_smem_index_destroy(accg_smem_index* x) {
  if (!x) return;
  hipSetDevice(x->ctx->device);
  hipStreamSynchronize(x->ctx->stream);
  delete x;
}

extern "C" int accg_smem_batch_create(accg_smem_index* idx, const uint8_t* seq, uint32_t seq_stride, const uint8_t* seq_len,
                                      uint32_t n_reads, uint32_t max_out, accg_smem_batch** out) {
  if (!idx) return ACCG_ERR_NOT_INITIALISED;
  if (!out || (n_reads && (!seq || !seq_len)) || max_out == 0 || seq_stride == 0) return ACCG_ERR_BAD_ARG;
  for (uint32_t i = 0; i < n_reads; i++) if (seq_len[i] > seq_stride) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(idx->ctx->device));
  std::unique_ptr<accg_smem_batch> b(new accg_smem_batch);
  b->idx = idx; b->n = n_reads; b->stride = seq_stride; b->max_out = max_out;
  // Reads of one launch: each holds 768 interval records of scratch (16 B with 32-bit bounds, 32 B otherwise), i.e. 12.9 GB
  // for 2^20 reads -- small change out of 288 GB, and one big launch beats several smaller ones (every launch ends in a tail
  // of the few wavefronts whose reads took longest: 21.5 ms in four launches, 17.5 ms in one, for 2^20 reads).
  uint32_t max_slice = idx->compact ? (1u << 20) : (1u << 19);
  if (const char* e = getenv("ACCG_SMEM_SLICE")) max_slice = std::max<uint32_t>(64, (uint32_t)strtoul(e, nullptr, 10));
  b->slice = std::min<uint32_t>(std::max<uint32_t>(n_reads, 64), max_slice);
  if (idx->compact) {
    const char* ee = getenv("ACCG_SMEM_ENGINE");
    if (ee && ee[0] == '1') {              // off by default: measured 27.7 ms against 17.5 ms of the plain kernel
      uint32_t waves = 4u * 4u * (uint32_t)std::max(idx->ctx->n_cu, 1);      // 4 resident wavefronts per SIMD at ~100 VGPRs
      if (const char* ew = getenv("ACCG_SMEM_ENGINE_WAVES")) waves = std::max<uint32_t>(1, (uint32_t)strtoul(ew, nullptr, 10));
      b->engine_waves = std::min<uint32_t>(waves, (std::max<uint32_t>(n_reads, 1) + 63) / 64);
      b->slice = b->engine_waves * 64;                                       // scratch is per persistent thread
      ACCG_HIP(hipMalloc((void**)&b->d_queue, sizeof(uint32_t)));
    }
  }
  const size_t n1 = std::max<uint32_t>(n_reads, 1);
  ACCG_HIP(hipMalloc((void**)&b->d_seq, n1 * seq_stride));
  ACCG_HIP(hipMalloc((void**)&b->d_len, n1));
  ACCG_HIP(hipMalloc((void**)&b->d_out, n1 * max_out * sizeof(SmemIntv)));
  ACCG_HIP(hipMalloc((void**)&b->d_num, n1 * sizeof(int32_t)));
  ACCG_HIP(hipMalloc((void**)&b->d_scratch, (size_t)b->slice * SMEM_SCRATCH_ENTRIES * (idx->compact ? 16 : sizeof(SmemIntv))));
  {
    const char* es = getenv("ACCG_SMEM_SPLIT");        // A/B knob: 1 = the three-kernel form (smem_kernel.hip; slower)
    if (es && es[0] == '1' && !b->engine_waves) {
      ACCG_HIP(hipMalloc((void**)&b->d_seg, (size_t)b->slice * 256 * sizeof(uint32_t)));
      ACCG_HIP(hipMalloc((void**)&b->d_nseg, (size_t)b->slice * sizeof(uint32_t)));
    }
  }
  if (n_reads) {
    ACCG_HIP(hipMemcpy(b->d_seq, seq, (size_t)n_reads * seq_stride, hipMemcpyHostToDevice));
    ACCG_HIP(hipMemcpy(b->d_len, seq_len, n_reads, hipMemcpyHostToDevice));
    for (uint32_t i = 0; i < n_reads; i++) { b->bases += seq_len[i]; b->longest = std::max<uint32_t>(b->longest, seq_len[i]); }
  }
  *out = b.release();
  return ACCG_OK;
}
extern "C" uint64_t accg_smem_batch_bases(const accg_smem_batch* b) { return b ? b->bases : 0; }

extern "C" int accg_smem_batch_run(accg_smem_batch* b) {
  if (!b) return ACCG_ERR_BAD_ARG;
  accg_smem_index* x = b->idx;
  ACCG_HIP(hipSetDevice(x->ctx->device));
  SmemArgs a;
  a.bwt = x->d_bwt; a.primary = x->primary; a.compact = x->compact ? 1u : 0u; a.ktab = x->d_ktab;
  for (int c = 0; c < 5; c++) a.L2[c] = x->L2[c];
  a.seq = b->d_seq; a.seq_len = b->d_len; a.seq_stride = b->stride; a.n_reads = b->n;
  a.out = b->d_out; a.mem_num = b->d_num; a.max_out = b->max_out; a.scratch = b->d_scratch; a.n_threads = b->slice;
  a.seg = b->d_seg; a.nseg = b->d_nseg;
  a.read_words = ((b->longest + 7) / 8) | 1u;
  // wavefronts per CU (by an LDS request): 20 for the fused kernel (what its 88 VGPRs admit; asking for exactly that measures 1-2 %
  // better than a looser request: 9.19 against 9.35 ms); the split form picks per kernel (smem_launch).  ACCG_SMEM_WPC overrides, 0 = no limit.
  { const char* e = getenv("ACCG_SMEM_WPC"); a.waves_per_cu = e ? (uint32_t)atoi(e) : (b->d_seg ? 0u : 20u); }
  // ACCG_SMEM_COUNT=1: the counting build of the same kernels (measurement runs only: accg_smem_debug_counts)
  static const bool counting = [] { const char* e = getenv("ACCG_SMEM_COUNT"); return e && e[0] == '1'; }();
  if (b->engine_waves) {
    a.queue = b->d_queue;
    ACCG_HIP(hipMemsetAsync(b->d_queue, 0, sizeof(uint32_t), x->ctx->stream));
    ACCG_HIP(counting ? smem_launch_engine_count(a, 0, b->n, b->engine_waves, x->ctx->stream) : smem_launch_engine(a, 0, b->n, b->engine_waves, x->ctx->stream));
    return ACCG_OK;
  }
  a.queue = nullptr;
  a.skip_pass3 = 0;
  // The third pass (LAST-like forward seeds: 28 % of the lookups, flat, memory bound with all lanes busy) beside the fused kernel's first
  // two (latency bound at a fifth of the lanes) instead of behind them in the same threads: it only appends, so it runs on a forked
  // stream into a list of its own and a small kernel appends that list afterwards.  ACCG_SMEM_PASS3_ASIDE=0/1.
  static const int aside_knob = [] { const char* e = getenv("ACCG_SMEM_PASS3_ASIDE"); return e ? atoi(e) : 1; }();
  if (aside_knob >= 1 && !counting && !b->d_seg && b->n) {
    constexpr uint32_t SLOTS = 16;            // a reported seed is at least 20 bases: at most 12 per read of 255
    accg_ctx* c = x->ctx;
    const size_t n1 = b->n;
    if (!b->d_out3) { ACCG_HIP(hipMalloc((void**)&b->d_out3, n1 * SLOTS * sizeof(SmemIntv))); ACCG_HIP(hipMalloc((void**)&b->d_num3, n1 * sizeof(int32_t))); ACCG_HIP(hipMalloc((void**)&b->d_num12, n1 * sizeof(int32_t))); }
    ACCG_HIP(ctx_fork(c));                                  // (behind the previous run's merge, which reads the list)
    hipStream_t side = c->aux[0];
    ACCG_HIP(hipMemsetAsync(b->d_num3, 0, n1 * sizeof(int32_t), side));
    SmemArgs a3 = a;
    a3.out = b->d_out3; a3.mem_num = b->d_num3; a3.max_out = SLOTS; a3.seg = nullptr;
    a.skip_pass3 = 1; a.mem_num = b->d_num12;
    for (uint32_t r0 = 0; r0 < b->n; r0 += b->slice) {
      const uint32_t m = std::min(b->slice, b->n - r0);
      ACCG_HIP(smem_launch(a, r0, m, c->stream));
      ACCG_HIP(smem_launch_pass3(a3, r0, m, side));
    }
    ACCG_HIP(ctx_join(c));
    for (uint32_t r0 = 0; r0 < b->n; r0 += b->slice)
      ACCG_HIP(smem_launch_merge3(b->d_out, b->d_num, b->d_num12, b->max_out, b->d_out3, b->d_num3, SLOTS, r0, std::min(b->slice, b->n - r0), c->stream));
    return ACCG_OK;
  }
  for (uint32_t r0 = 0; r0 < b->n; r0 += b->slice)
    ACCG_HIP(counting ? smem_launch_count(a, r0, std::min(b->slice, b->n - r0), x->ctx->stream) : smem_launch(a, r0, std::min(b->slice, b->n - r0), x->ctx->stream));
  return ACCG_OK;
}
// Lookups the SMEM kernels really performed since the last call (only under ACCG_SMEM_COUNT=1, else zeros): {32-byte index
// sectors fetched, prefix-table entries fetched, bwt_extend calls, 0}.  The reference accounts REQUESTED blocks
// (smem/host/baseline.cpp:28-75); this is what the device fetched.
extern "C" int accg_smem_debug_counts(accg_ctx* ctx, uint64_t out[4]) {
  if (!ctx) return ACCG_ERR_NOT_INITIALISED;
  if (!out) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(ctx->device));
  ACCG_HIP(smem_counts_read(out, true, ctx->stream));
  return ACCG_OK;
}
extern "C" int accg_smem_batch_time(accg_smem_batch* b, int warmup, int iters, float* ms_per_run) {
  if (!b || !ms_per_run || iters <= 0 || warmup < 0) return ACCG_ERR_BAD_ARG;
  accg_ctx* c = b->idx->ctx;
  int st;
  for (int i = 0; i < warmup; i++) if ((st = accg_smem_batch_run(b)) != ACCG_OK) return st;
  ACCG_HIP(hipEventRecord(c->ev0, c->stream));
  for (int i = 0; i < iters; i++) if ((st = accg_smem_batch_run(b)) != ACCG_OK) return st;
  ACCG_HIP(hipEventRecord(c->ev1, c->stream));
  ACCG_HIP(hipEventSynchronize(c->ev1));
  float ms = 0;
  ACCG_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *ms_per_run = ms / iters;
  return ACCG_OK;
}
extern "C" int accg_smem_batch_results(accg_smem_batch* b, void* mem_output, int32_t* mem_num) {
  if (!b) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->idx->ctx->device));
  ACCG_HIP(hipStreamSynchronize(b->idx->ctx->stream));
  if (b->n) {
    if (mem_output) ACCG_HIP(hipMemcpy(mem_output, b->d_out, (size_t)b->n * b->max_out * sizeof(SmemIntv), hipMemcpyDeviceToHost));
    if (mem_num) ACCG_HIP(hipMemcpy(mem_num, b->d_num, (size_t)b->n * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  return ACCG_OK;
}
extern "C" void accg_smem_batch_destroy(accg_smem_batch* b) {
  if (!b) return;
  hipSetDevice(b->idx->ctx->device);
  hipStreamSynchronize(b->idx->ctx->stream);
  delete b;
}

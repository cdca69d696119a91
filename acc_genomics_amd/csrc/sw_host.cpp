// Host side of the HTC Smith-Waterman path: batch upload, pairing/packing of pairs into wavefront jobs,
// launches, results.  Stands in for the pair loop of SWPairwiseAlignmentMultiBatch
// (htc-sw/host/FalconSW_AVX.cpp:304-313) up to and including the end-cell selection (:2314-2339).
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <memory>
#include <numeric>
#include <vector>
#include "accg_internal.h"
#include "sw_dev.h"

using namespace accg;

namespace {
struct SwLaunch { int K, lpp; bool pack16, lane_is_alt; uint32_t work0, n_work; int sweep_cap; };
// device blocks come from the context's cache (accg_internal.h): FalconSWFPGA_run builds and drops a batch per call
template <typename T> int dev_alloc(accg_ctx* c, T** dst, size_t bytes) {
  ACCG_HIP(c->pool.get(bytes ? bytes : 16, (void**)dst));
  return ACCG_OK;
}
template <typename T> int dev_upload(accg_ctx* c, T** dst, const void* src, size_t bytes, hipStream_t s) {
  int st = dev_alloc(c, dst, bytes);
  if (st != ACCG_OK) return st;
  if (bytes) ACCG_HIP(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, s));
  return ACCG_OK;
}
}  // namespace

struct accg_sw_batch {
  accg_ctx* ctx = nullptr;
  int n = 0;
  uint64_t cells = 0, algo_bytes = 0;
  std::vector<SwLaunch> launches;
  uint8_t *d_refs = nullptr, *d_alts = nullptr, *d_strat = nullptr;
  int32_t *d_rl = nullptr, *d_al = nullptr, *d_score = nullptr, *d_p1 = nullptr, *d_p2 = nullptr;
  SwWork* d_work = nullptr;
  SwArgs args{};
  // backtrace mode
  uint4* d_bt = nullptr; uint64_t bt_bytes = 0;
  int32_t *d_cig_n = nullptr, *d_cig_off = nullptr, *d_cig_el = nullptr, *d_cig_packed = nullptr;
  unsigned long long *d_cig_start = nullptr, *d_cig_total = nullptr;
  int max_el = 0;
  ~accg_sw_batch() {            // also reached on the error paths of accg_sw_batch_create
    for (void* p : {(void*)d_refs, (void*)d_alts, (void*)d_strat, (void*)d_rl, (void*)d_al, (void*)d_score,
                    (void*)d_work, (void*)d_bt, (void*)d_cig_total, (void*)d_cig_el, (void*)d_cig_packed})
      if (p) ctx->pool.put(p);
  }
};

extern "C" int accg_sw_batch_create(accg_ctx* ctx, int n, const uint8_t* refs, size_t ref_stride, const int32_t* ref_lens,
                                    const uint8_t* alts, size_t alt_stride, const int32_t* alt_lens,
                                    const uint8_t* strategies, int w_match, int w_mismatch, int w_open, int w_extend,
                                    accg_sw_batch** out) {
  if (!ctx) return ACCG_ERR_NOT_INITIALISED;
  if (!out || n < 0 || (n > 0 && (!refs || !alts || !ref_lens || !alt_lens))) return ACCG_ERR_BAD_ARG;
  *out = nullptr;
  ACCG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<accg_sw_batch> b(new accg_sw_batch);
  SyncOnError sync_on_error(ctx->stream);
  b->ctx = ctx; b->n = n;
  struct Item { uint32_t idx; int K, lpp, ns; bool p16, lia; };
  std::vector<Item> items(n);
  int max_rl = 0, max_al = 0;
  for (int k = 0; k < n; k++) {
    const int rl = ref_lens[k], al = alt_lens[k];
    if (rl <= 0 || al <= 0) return ACCG_ERR_EMPTY_SEQ;
    if (rl > ACCG_SW_MAX_LEN || al > ACCG_SW_MAX_LEN) return ACCG_ERR_TOO_LONG;
    if (strategies && strategies[k] > 3) return ACCG_ERR_BAD_ARG;
    max_rl = std::max(max_rl, rl); max_al = std::max(max_al, al);
    const bool lia = al <= rl;                       // lanes hold the shorter sequence
    const int nl = lia ? al : rl, ns = lia ? rl : al;
    // 16-bit mode only when every reachable matrix value provably fits:
    // highest: nl matches; lowest: a border prefill of ns gaps plus nl mismatches plus one more open.
    const long hi = (long)std::max(w_match, 0) * nl;
    const long lo = (long)std::min(w_open, 0) * 2 + (long)std::min(w_extend, 0) * ns + (long)std::min(w_mismatch, 0) * nl;
    const int lpp = nl <= 255 ? 16 : nl <= 511 ? 32 : 64;
    const int Kk = sw_pick_k(nl, lpp);
    // ... and only up to 16 positions per lane: the decision record keeps one bit per position in each 16-bit half of its planes
    // (K = 20 / 24 exist for sequences over 1024 on 64 lanes; with small-magnitude weights those used to pass the range test, and
    // their CIGARs came out wrong while score and end cell were right -- found by tools/fuzz_sw.py)
    static const bool no16 = [] { const char* e = getenv("ACCG_SW_P16"); return e && e[0] == '0'; }();     // A/B knob: int32 arithmetic for every pair
    const bool p16 = !no16 && hi <= 32000 && lo >= -32000 && std::abs(w_match) < 16000 && std::abs(w_mismatch) < 16000 && Kk <= 16;
    items[k] = {(uint32_t)k, Kk, lpp, ns, p16, lia};
    b->cells += (uint64_t)rl * al;
    b->algo_bytes += (uint64_t)rl + al + 16;
  }
  if ((ref_stride != 0 && (size_t)max_rl > ref_stride) || (alt_stride != 0 && (size_t)max_al > alt_stride)) return ACCG_ERR_BAD_ARG;
  std::stable_sort(items.begin(), items.end(), [](const Item& x, const Item& y) {
    if (x.lia != y.lia) return x.lia > y.lia;
    if (x.p16 != y.p16) return x.p16 > y.p16;
    if (x.lpp != y.lpp) return x.lpp < y.lpp;
    if (x.K != y.K) return x.K > y.K;
    return x.ns > y.ns;
  });
  // groups of one (int32) or two (16-bit, equal sweep length) pairs; four groups per wavefront
  std::vector<SwWork> work;
  size_t i = 0;
  while (i < items.size()) {
    const Item& first = items[i];
    SwLaunch L{first.K, first.lpp, first.p16, first.lia, (uint32_t)work.size(), 0, 0};
    const int gpw = 64 / first.lpp;   // groups per wavefront
    std::vector<std::pair<uint32_t, uint32_t>> groups;
    while (i < items.size() && items[i].K == first.K && items[i].lpp == first.lpp && items[i].p16 == first.p16 && items[i].lia == first.lia) {
      L.sweep_cap = std::max(L.sweep_cap, items[i].ns);
      if (first.p16 && i + 1 < items.size() && items[i + 1].K == first.K && items[i + 1].lpp == first.lpp && items[i + 1].p16 && items[i + 1].lia == first.lia &&
          items[i + 1].ns == items[i].ns) {
        groups.push_back({items[i].idx, items[i + 1].idx}); i += 2;
      } else { groups.push_back({items[i].idx, SW_NO_PAIR}); i += 1; }
    }
    for (size_t gI = 0; gI < groups.size(); gI += gpw) {
      SwWork w;
      for (int g = 0; g < 4; g++) {
        const bool on = g < gpw && gI + g < groups.size();
        w.pair[2 * g] = on ? groups[gI + g].first : SW_NO_PAIR;
        w.pair[2 * g + 1] = on ? groups[gI + g].second : SW_NO_PAIR;
      }
      work.push_back(w);
    }
    L.n_work = (uint32_t)work.size() - L.work0;
    b->launches.push_back(L);
  }
  hipStream_t s = ctx->stream;
  std::vector<uint8_t> strat(std::max(n, 1), 0);
  if (strategies) memcpy(strat.data(), strategies, n);
  const size_t rbytes = ref_stride ? ref_stride * (size_t)n : (size_t)max_rl, abytes = alt_stride ? alt_stride * (size_t)n : (size_t)max_al;
  int st;
  if ((st = dev_upload(ctx, &b->d_refs, refs, rbytes, s)) != ACCG_OK) return st;
  if ((st = dev_upload(ctx, &b->d_alts, alts, abytes, s)) != ACCG_OK) return st;
  if ((st = dev_upload(ctx, &b->d_rl, ref_lens, sizeof(int32_t) * n, s)) != ACCG_OK) return st;
  if ((st = dev_upload(ctx, &b->d_al, alt_lens, sizeof(int32_t) * n, s)) != ACCG_OK) return st;
  if ((st = dev_upload(ctx, &b->d_strat, strat.data(), n, s)) != ACCG_OK) return st;
  if ((st = dev_upload(ctx, &b->d_work, work.data(), sizeof(SwWork) * work.size(), s)) != ACCG_OK) return st;
  // score | p1 | p2 in one block: one copy brings all three back
  if ((st = dev_alloc(ctx, &b->d_score, 3 * sizeof(int32_t) * std::max(n, 1))) != ACCG_OK) return st;
  b->d_p1 = b->d_score + std::max(n, 1); b->d_p2 = b->d_p1 + std::max(n, 1);
  ACCG_HIP(hipStreamSynchronize(s));
  SwArgs& a = b->args;
  a.refs = b->d_refs; a.alts = b->d_alts; a.ref_stride = (uint32_t)ref_stride; a.alt_stride = (uint32_t)alt_stride;
  a.ref_len = b->d_rl; a.alt_len = b->d_al; a.strategy = b->d_strat; a.work = b->d_work;
  a.score = b->d_score; a.p1 = b->d_p1; a.p2 = b->d_p2;
  a.w_match = w_match; a.w_mismatch = w_mismatch; a.w_open = w_open; a.w_extend = w_extend;
  sync_on_error.dismiss();
  *out = b.release();
  return ACCG_OK;
}

extern "C" uint64_t accg_sw_batch_cells(const accg_sw_batch* b) { return b ? b->cells : 0; }
extern "C" uint64_t accg_sw_batch_algorithmic_bytes(const accg_sw_batch* b) { return b ? b->algo_bytes : 0; }

extern "C" int accg_sw_batch_run(accg_sw_batch* b) {
  if (!b) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  for (const SwLaunch& l : b->launches)
    ACCG_HIP(sw_launch(l.K, l.lpp, l.pack16, l.lane_is_alt, false, b->args, l.work0, l.n_work, l.work0, l.sweep_cap, b->ctx->stream));
  return ACCG_OK;
}

// Fill with the per-cell decision record, then the backtrace (calculateCigarOneBatch).  The record is
// 256 B per step per group of two pairs (40 KB per pair at configs[2]); it lives in one scratch buffer of
// at most ACCG_SW_BT_BYTES (default 16 GiB, env) whose two halves successive slices of the job list take in turn.
extern "C" int accg_sw_batch_run_cigar(accg_sw_batch* b, int max_el) {
  if (!b || max_el <= 0) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  hipStream_t s = b->ctx->stream;
  const size_t n1 = (size_t)std::max(b->n, 1);
  if (max_el != b->max_el) {
    ACCG_HIP(hipStreamSynchronize(s));
    if (b->d_cig_el) { b->ctx->pool.put(b->d_cig_el); b->d_cig_el = nullptr; }
    if (b->d_cig_packed) { b->ctx->pool.put(b->d_cig_packed); b->d_cig_packed = nullptr; }
    { int st_ = dev_alloc(b->ctx, &b->d_cig_el, n1 * max_el * 2 * sizeof(int32_t)); if (st_ != ACCG_OK) return st_; }
    { int st_ = dev_alloc(b->ctx, &b->d_cig_packed, n1 * max_el * 2 * sizeof(int32_t)); if (st_ != ACCG_OK) return st_; }
    if (!b->d_cig_total) {     // [total u64][start u64 x n][n_el i32 x n][offset i32 x n]: the per-pair results come back in one copy
      int st_ = dev_alloc(b->ctx, &b->d_cig_total, (1 + n1) * sizeof(unsigned long long) + 2 * n1 * sizeof(int32_t));
      if (st_ != ACCG_OK) return st_;
      b->d_cig_start = b->d_cig_total + 1;
      b->d_cig_n = reinterpret_cast<int32_t*>(b->d_cig_start + n1);
      b->d_cig_off = b->d_cig_n + n1;
    }
    b->max_el = max_el;
  }
  uint64_t limit = 16ull << 30;
  if (const char* e = getenv("ACCG_SW_BT_BYTES")) limit = strtoull(e, nullptr, 10);
  // The scratch is two halves: while the backtrace of one slice walks its record (one thread per pair, dependent loads: latency
  // bound) on an aux stream, the fill of the next slice (packed-VALU bound) writes the other half on the main stream.  A launch
  // is cut into at least SW_BT_MIN_SLICES slices so that there is something to overlap.
  constexpr uint64_t SW_BT_MIN_SLICES = 6;
  auto slice_items = [&](const SwLaunch& l, uint64_t per) {
    const uint64_t by_mem = std::max<uint64_t>(1, (limit / 2) / per);
    const uint64_t by_cut = std::max<uint64_t>(4096, (l.n_work + SW_BT_MIN_SLICES - 1) / SW_BT_MIN_SLICES);
    return std::max<uint64_t>(1, std::min<uint64_t>(l.n_work, std::min(by_mem, by_cut)));
  };
  // the record's layout per launch: lane masks through scalar stores where the shape has them (sw_dev.h: SwArgs::bt_masks; ACCG_SW_MASKS=0: never)
  static const bool masks_off = [] { const char* e = getenv("ACCG_SW_MASKS"); return e && e[0] == '0'; }();
  auto use_masks = [&](const SwLaunch& l) { return !masks_off && l.pack16 && l.lpp == 16; };
  auto item_uint4 = [&](const SwLaunch& l) { return use_masks(l) ? sw_bt_item_uint4_masks(l.sweep_cap, l.lpp, l.K) : sw_bt_item_uint4(l.sweep_cap, l.lpp); };
  uint64_t half = 0;
  for (const SwLaunch& l : b->launches) {
    const uint64_t per = item_uint4(l) * sizeof(uint4);
    half = std::max(half, slice_items(l, per) * per);
  }
  half = (half + 255) & ~(uint64_t)255;
  if (2 * half > b->bt_bytes) {
    ACCG_HIP(hipStreamSynchronize(s));
    if (b->d_bt) b->ctx->pool.put(b->d_bt);
    b->d_bt = nullptr; b->bt_bytes = 0;
    { int st_ = dev_alloc(b->ctx, &b->d_bt, 2 * half); if (st_ != ACCG_OK) return st_; }
    b->bt_bytes = 2 * half;
  }
  SwArgs a = b->args;
  a.cig_n = b->d_cig_n; a.cig_off = b->d_cig_off; a.cig_el = b->d_cig_el; a.max_el = max_el;
  a.cig_packed = b->d_cig_packed; a.cig_start = b->d_cig_start; a.cig_total = b->d_cig_total;
  ACCG_HIP(hipMemsetAsync(b->d_cig_total, 0, sizeof(unsigned long long), s));
  accg_ctx* c = b->ctx;
  ACCG_HIP(ctx_need_aux(c));
  hipStream_t st = c->aux[0];
  { const char* e = getenv("ACCG_SW_TRACE_SAME_STREAM"); if (e && e[0] == '1') st = s; }   // A/B knob: backtrace behind its fill, no overlap
  ACCG_HIP(ctx_fork(c));                                   // the trace stream starts behind everything queued so far (the memset)
  bool used[2] = {false, false};
  uint32_t k = 0;
  for (const SwLaunch& l : b->launches) {
    a.bt_item_stride = item_uint4(l);
    a.bt_masks = use_masks(l) ? 1 : 0;
    const uint64_t per = a.bt_item_stride * sizeof(uint4);
    const uint32_t slice = (uint32_t)slice_items(l, per);
    for (uint32_t off = 0; off < l.n_work; off += slice, k++) {
      const uint32_t n = std::min(slice, l.n_work - off), w0 = l.work0 + off;
      const int h = (int)(k & 1);
      a.bt = reinterpret_cast<uint4*>(reinterpret_cast<char*>(b->d_bt) + (size_t)h * (b->bt_bytes / 2));
      if (used[h]) ACCG_HIP(hipStreamWaitEvent(s, c->ev_join[h], 0));      // the walk over this half's previous record is through
      static const int dbg = [] { const char* e = getenv("ACCG_SW_BT_DEBUG"); return e ? atoi(e) : 0; }();   // measurement aid (tools/exp_sw_bt.py): 1 = no trace kernels, 2 = no record stores either; the CIGARs of such a run are garbage
      if (dbg) { if (dbg == 2) a.bt = nullptr; ACCG_HIP(sw_launch(l.K, l.lpp, l.pack16, l.lane_is_alt, true, a, w0, n, w0, l.sweep_cap, s)); continue; }
      ACCG_HIP(sw_launch(l.K, l.lpp, l.pack16, l.lane_is_alt, true, a, w0, n, w0, l.sweep_cap, s));
      ACCG_HIP(hipEventRecord(c->ev_fork, s));
      ACCG_HIP(hipStreamWaitEvent(st, c->ev_fork, 0));
      ACCG_HIP(sw_trace_launch(l.K, l.lpp, l.pack16, l.lane_is_alt, a, w0, n, w0, l.sweep_cap, st));
      ACCG_HIP(hipEventRecord(c->ev_join[h], st));
      used[h] = true;
    }
  }
  for (int h = 0; h < 2; h++) if (used[h]) ACCG_HIP(hipStreamWaitEvent(s, c->ev_join[h], 0));
  return ACCG_OK;
}

// The CIGARs leave the device packed back to back (a few elements per pair); the fixed-slot layout of the reference's
// Cigar arrays is rebuilt on the host.  One pinned staging block takes [total][starts][n_el][offsets][elements]: two copies
// (the 8-byte total first, to size the block), and the view form hands out pointers into it instead of copying again.
extern "C" int accg_sw_batch_cigars_packed_view(accg_sw_batch* b, const int32_t** n_el, const int32_t** offsets, const uint64_t** starts,
                                                const int32_t** elements, uint64_t* total) {
  if (!b || !b->d_cig_n) return ACCG_ERR_BAD_ARG;
  accg_ctx* c = b->ctx;
  ACCG_HIP(hipSetDevice(c->device));
  const size_t n1 = std::max<size_t>((size_t)b->n, 1);
  const size_t meta = ((1 + n1) * sizeof(unsigned long long) + 2 * n1 * sizeof(int32_t) + 15) & ~(size_t)15;
  void* stage = nullptr;
  ACCG_HIP(ctx_stage(c, meta, &stage));
  ACCG_HIP(hipMemcpyAsync(stage, b->d_cig_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  ACCG_HIP(hipStreamSynchronize(c->stream));
  const unsigned long long tot = *(const unsigned long long*)stage;
  const size_t el_bytes = (size_t)tot * 2 * sizeof(int32_t);
  ACCG_HIP(ctx_stage(c, meta + el_bytes, &stage));        // may move the block: everything is copied after this
  ACCG_HIP(hipMemcpyAsync(stage, b->d_cig_total, (1 + n1) * sizeof(unsigned long long) + 2 * n1 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (el_bytes) ACCG_HIP(hipMemcpyAsync((char*)stage + meta, b->d_cig_packed, el_bytes, hipMemcpyDeviceToHost, c->stream));
  ACCG_HIP(hipStreamSynchronize(c->stream));
  const unsigned long long* m64 = (const unsigned long long*)stage;
  const int32_t* m32 = (const int32_t*)(m64 + 1 + n1);
  if (total) *total = tot;
  if (starts) *starts = (const uint64_t*)(m64 + 1);
  if (n_el) *n_el = m32;
  if (offsets) *offsets = m32 + n1;
  if (elements) *elements = (const int32_t*)((const char*)stage + meta);
  return ACCG_OK;
}

namespace {
void par_copy(void* dst, const void* src, size_t bytes) {          // memcpy over the host threads the process may use
  const int T = accg::host_threads();
  if (bytes < (8u << 20) || T < 2) { memcpy(dst, src, bytes); return; }
#pragma omp parallel for schedule(static) num_threads(T)
  for (int t = 0; t < T; t++) {
    const size_t a = bytes * t / T, e = bytes * (t + 1) / T;
    memcpy((char*)dst + a, (const char*)src + a, e - a);
  }
}
}  // namespace

extern "C" int accg_sw_batch_cigars_packed(accg_sw_batch* b, int32_t* n_el, int32_t* offsets, uint64_t* starts, int32_t* elements,
                                           uint64_t capacity, uint64_t* total) {
  if (!b || !b->d_cig_n) return ACCG_ERR_BAD_ARG;
  const size_t n = (size_t)b->n;
  if (!elements && !n_el && !offsets && !starts) {         // size query: the total alone
    ACCG_HIP(hipSetDevice(b->ctx->device));
    unsigned long long tot = 0;
    ACCG_HIP(hipMemcpyAsync(&tot, b->d_cig_total, sizeof tot, hipMemcpyDeviceToHost, b->ctx->stream));
    ACCG_HIP(hipStreamSynchronize(b->ctx->stream));
    if (total) *total = tot;
    return ACCG_OK;
  }
  const int32_t *v_n, *v_off, *v_el; const uint64_t* v_st; uint64_t tot = 0;
  int st = accg_sw_batch_cigars_packed_view(b, &v_n, &v_off, &v_st, &v_el, &tot);
  if (st != ACCG_OK) return st;
  if (total) *total = tot;
  if (n) {
    if (n_el) par_copy(n_el, v_n, n * sizeof(int32_t));
    if (offsets) par_copy(offsets, v_off, n * sizeof(int32_t));
    if (starts) par_copy(starts, v_st, n * sizeof(uint64_t));
  }
  if (elements) {
    if (capacity < tot) return ACCG_ERR_BAD_ARG;           // *total says how much is needed
    par_copy(elements, v_el, (size_t)tot * 2 * sizeof(int32_t));
  }
  return ACCG_OK;
}

extern "C" int accg_sw_batch_cigars(accg_sw_batch* b, int32_t* n_el, int32_t* offsets, int32_t* elements) {
  if (!b || !b->d_cig_n) return ACCG_ERR_BAD_ARG;
  const size_t n = (size_t)b->n;
  if (!elements) return accg_sw_batch_cigars_packed(b, n_el, offsets, nullptr, nullptr, 0, nullptr);
  const int32_t *cnt, *v_off, *packed; const uint64_t* starts; uint64_t tot = 0;
  int st = accg_sw_batch_cigars_packed_view(b, &cnt, &v_off, &starts, &packed, &tot);
  if (st != ACCG_OK) return st;
  const size_t slot = (size_t)b->max_el * 2;
#pragma omp parallel for schedule(static) num_threads(accg::host_threads())
  for (size_t k = 0; k < n; k++)
    if (cnt[k] > 0) memcpy(elements + k * slot, packed + starts[k] * 2, (size_t)cnt[k] * 2 * sizeof(int32_t));
  if (n && n_el) memcpy(n_el, cnt, n * sizeof(int32_t));
  if (n && offsets) memcpy(offsets, v_off, n * sizeof(int32_t));
  return ACCG_OK;
}

extern "C" int accg_sw_batch_time(accg_sw_batch* b, int warmup, int iters, float* ms_per_run) {
  if (!b || !ms_per_run || iters <= 0 || warmup < 0) return ACCG_ERR_BAD_ARG;
  accg_ctx* c = b->ctx;
  int st;
  for (int i = 0; i < warmup; i++) if ((st = accg_sw_batch_run(b)) != ACCG_OK) return st;
  ACCG_HIP(hipEventRecord(c->ev0, c->stream));
  for (int i = 0; i < iters; i++) if ((st = accg_sw_batch_run(b)) != ACCG_OK) return st;
  ACCG_HIP(hipEventRecord(c->ev1, c->stream));
  ACCG_HIP(hipEventSynchronize(c->ev1));
  float ms = 0;
  ACCG_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *ms_per_run = ms / iters;
  return ACCG_OK;
}

extern "C" int accg_sw_batch_results(accg_sw_batch* b, int32_t* score, int32_t* p1, int32_t* p2) {
  if (!b) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  ACCG_HIP(hipStreamSynchronize(b->ctx->stream));
  const size_t bytes = sizeof(int32_t) * (size_t)b->n;
  if (bytes) {
    void* stage = nullptr;
    ACCG_HIP(ctx_stage(b->ctx, 3 * bytes, &stage));
    ACCG_HIP(hipMemcpyAsync(stage, b->d_score, 3 * bytes, hipMemcpyDeviceToHost, b->ctx->stream));
    ACCG_HIP(hipStreamSynchronize(b->ctx->stream));
    const int32_t* r = (const int32_t*)stage;
    if (score) memcpy(score, r, bytes);
    if (p1) memcpy(p1, r + b->n, bytes);
    if (p2) memcpy(p2, r + 2 * (size_t)b->n, bytes);
  }
  return ACCG_OK;
}

extern "C" void accg_sw_batch_destroy(accg_sw_batch* b) {
  if (!b) return;
  hipSetDevice(b->ctx->device);
  hipStreamSynchronize(b->ctx->stream);
  delete b;
}

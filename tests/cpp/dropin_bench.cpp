// Drives the drop-in entry points the way the reference's callers do -- one region per blocking call, several caller threads at
// once -- and times it.  Built as tests/cpp/libdropin_bench.so; bench.py (`e2e.dropin`) and tests/test_dropin_gpu.py call it through
// ctypes so that the callers are native threads, as in the reference (pairhmm/host/PairHMMFpga.cpp:125-162 is called once per active
// region; an accelerator manager runs one PairHMM task instance per request, several at a time: pairhmm/task/xlnx/PairHMMTask.cpp:27-143).
//
//   what = 0: accg_phmm_region, one accg_ctx per caller thread (include/accg.h)
//   what = 1: the task plugin of libaccg_compat.so per region: create() -> setInput x 3 -> prepare() -> compute() -> output block 0 -> destroy()
//   what = 2: compute_fpga (not re-entrant, like the reference's: threads must be 1)
//   what = 3: FalconPairHMM::computePairhmm, one object per caller thread (final log10 likelihoods; out_raw stays untouched)
//   what = 4: accg_phmm_mux_region on one mux shared by all caller threads (include/accg.h); lanes = ACCG_MUX_LANES or 6
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>
#include "../../acc_genomics_amd/csrc/compat/accg_compat.h"
#include "../../acc_genomics_amd/csrc/compat/accg_task.h"
#include "../../include/accg.h"

namespace {
struct Shared {
  int what, n_regions;
  const void* const* rs; const size_t* rb; const void* const* hs; const size_t* hb;
  const uint64_t* out_off;      // per region: index of its first pair in out_raw / out_log10
  float* out_raw; double* out_log10;
  accg_phmm_mux* mux = nullptr;
  std::atomic<int> next{0};
  std::atomic<int> failed{0};
};
uint64_t pairs_of(const Shared& S, int i) {
  int32_t nr = 0, nh = 0;
  memcpy(&nr, S.rs[i], 4); memcpy(&nh, S.hs[i], 4);
  return (uint64_t)nr * (uint64_t)nh;
}
void to_input(const void* rblob, const void* hblob, pairhmmInput& in) {
  read_t* r = nullptr; hap_t* h = nullptr;
  const int nr = deserialize(rblob, r), nh = deserialize(hblob, h);
  in.reads.resize((size_t)nr); in.haps.resize((size_t)nh);
  for (int i = 0; i < nr; i++) {
    const size_t L = (size_t)r[i].len;
    in.reads[i].bases.assign(r[i]._b, L); in.reads[i]._q.assign(r[i]._q, L); in.reads[i]._i.assign(r[i]._i, L);
    in.reads[i]._d.assign(r[i]._d, L); in.reads[i]._c.assign(r[i]._c, L);
  }
  for (int j = 0; j < nh; j++) in.haps[j].bases.assign(h[j]._b, (size_t)h[j].len);
  free_reads(r, nr); free_haps(h, nh);
}
void caller(Shared* S, accg_ctx* ctx, FalconPairHMM* falcon, const std::vector<pairhmmInput>* inputs) {
  try {
    for (;;) {
      const int i = S->next.fetch_add(1);
      if (i >= S->n_regions) return;
      const uint64_t n = pairs_of(*S, i);
      float* raw = S->out_raw ? S->out_raw + S->out_off[i] : nullptr;
      double* l10 = S->out_log10 ? S->out_log10 + S->out_off[i] : nullptr;
      if (S->what == 0) {
        accg_counters c;
        if (accg_phmm_region(ctx, S->rs[i], S->rb[i], S->hs[i], S->hb[i], ACCG_PHMM_FAST, raw, l10, &c) != ACCG_OK) { S->failed++; return; }
      } else if (S->what == 4) {
        accg_counters c;
        if (accg_phmm_mux_region(S->mux, S->rs[i], S->rb[i], S->hs[i], S->hb[i], ACCG_PHMM_FAST, raw, l10, &c) != ACCG_OK) { S->failed++; return; }
      } else if (S->what == 1) {
        task_host::Task* t = create();
        const uint64_t num_cell = 0;
        t->setInput(0, &num_cell, 8); t->setInput(1, S->rs[i], S->rb[i]); t->setInput(2, S->hs[i], S->hb[i]);
        t->prepare();
        t->compute();
        const std::vector<float>& o = t->getOutputBlock(0);
        if (o.size() != n) { S->failed++; destroy(t); return; }
        if (raw) memcpy(raw, o.data(), n * sizeof(float));
        destroy(t);
      } else if (S->what == 2) {
        std::string r((const char*)S->rs[i], S->rb[i]), h((const char*)S->hs[i], S->hb[i]);
        const float* o = compute_fpga("", r, h, 0);
        if (!o) { S->failed++; return; }
        if (raw) memcpy(raw, o, n * sizeof(float));
      } else {
        pairhmmOutput out; bool used = false;
        pairhmmInput in = (*inputs)[(size_t)i];        // (the caller owns a mutable input in the reference's test main too)
        falcon->computePairhmm(&in, &out, used);
        if (!used || out.likelihoodData.size() != n) { S->failed++; return; }
        if (l10) memcpy(l10, out.likelihoodData.data(), n * sizeof(double));
      }
    }
  } catch (...) { S->failed++; }
}
}  // namespace

// Returns 0 and the wall time of the fastest of `passes` passes over all regions (seconds) in *best_s; -1 on any failure.
// out_raw / out_log10 (nullable): every region's results at out_off[region], from the last pass.
extern "C" int dropin_bench(int what, int threads, int n_regions, const void* const* rs, const size_t* rb, const void* const* hs, const size_t* hb,
                            const uint64_t* out_off, int passes, float* out_raw, double* out_log10, double* best_s) {
  if (what < 0 || what > 4 || threads < 1 || threads > 256 || n_regions < 1 || passes < 1 || !best_s || (what == 2 && threads != 1)) return -1;
  std::vector<accg_ctx*> ctxs;
  std::vector<FalconPairHMM*> falcons;
  std::vector<pairhmmInput> inputs;
  int rc = 0;
  try {
    if (what == 0) for (int t = 0; t < threads; t++) { accg_ctx* c = nullptr; if (accg_init(0, &c) != ACCG_OK) { rc = -1; break; } ctxs.push_back(c); }
    if (what == 3) {
      for (int t = 0; t < threads; t++) falcons.push_back(new FalconPairHMM());
      inputs.resize((size_t)n_regions);
      for (int i = 0; i < n_regions; i++) to_input(rs[i], hs[i], inputs[(size_t)i]);
    }
  } catch (...) { rc = -1; }
  accg_phmm_mux* mux = nullptr;
  if (what == 4 && rc == 0) {
    const char* l = getenv("ACCG_MUX_LANES");
    if (accg_phmm_mux_create(0, l && atoi(l) > 0 ? atoi(l) : 6, 64, &mux) != ACCG_OK) rc = -1;
  }
  double best = -1;
  for (int p = 0; p < passes + 1 && rc == 0; p++) {       // (one untimed pass first)
    Shared S;
    S.mux = mux; S.what = what; S.n_regions = n_regions; S.rs = rs; S.rb = rb; S.hs = hs; S.hb = hb; S.out_off = out_off; S.out_raw = out_raw; S.out_log10 = out_log10;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(caller, &S, what == 0 ? ctxs[(size_t)t] : nullptr, what == 3 ? falcons[(size_t)t] : nullptr, &inputs);
    caller(&S, what == 0 ? ctxs[0] : nullptr, what == 3 ? falcons[0] : nullptr, &inputs);
    for (auto& x : th) x.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (S.failed.load()) rc = -1;
    if (p > 0 && (best < 0 || s < best)) best = s;
  }
  if (mux) {
    uint64_t nb = 0, nr = 0;
    accg_phmm_mux_stats(mux, &nb, &nr);
    if (getenv("ACCG_TRACE_MUX")) fprintf(stderr, "mux: %llu regions in %llu batches\n", (unsigned long long)nr, (unsigned long long)nb);
    accg_phmm_mux_destroy(mux);
  }
  for (accg_ctx* c : ctxs) accg_shutdown(c);
  for (FalconPairHMM* f : falcons) delete f;
  *best_s = best;
  return rc;
}

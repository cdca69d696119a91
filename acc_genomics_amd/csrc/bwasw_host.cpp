// Host side of the BWA-MEM seed-extension path: the role of the FPGA host's task packing (bwa-sw/sdaccel: seed_param /
// seed_seq streams into seed_proc, smithwaterman.cpp:564-584) and of reading back the five result words (:666-670).
#include <string.h>
#include <algorithm>
#include <memory>
#include <numeric>
#include <vector>
#include "accg_internal.h"
#include "bwasw_dev.h"

using namespace accg;

struct accg_bwasw_batch {
  accg_ctx* ctx = nullptr;
  uint32_t n = 0;
  uint8_t* d_blob = nullptr;
  BwaswSeed* d_seeds = nullptr;
  BwaswWork* d_work = nullptr;
  int16_t* d_out = nullptr;
  std::vector<uint16_t> seed_index;                      // echoed into the packed result words
  struct Launch { int K, side; uint32_t work_off, n_work; };
  std::vector<Launch> launches;
  uint64_t cells = 0;
  ~accg_bwasw_batch() { for (void* p : {(void*)d_blob, (void*)d_seeds, (void*)d_work, (void*)d_out}) if (p) ctx->pool.put(p); }
};

extern "C" int accg_bwasw_batch_create(accg_ctx* ctx, uint32_t n_seeds, const uint8_t* seqs, const uint32_t* seq_off,
                                       const uint16_t* params, accg_bwasw_batch** out) {
  if (!ctx) return ACCG_ERR_NOT_INITIALISED;
  if (!out || (n_seeds && (!seqs || !seq_off || !params))) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<accg_bwasw_batch> b(new accg_bwasw_batch);
  SyncOnError sync_on_error(ctx->stream);
  b->ctx = ctx; b->n = n_seeds;
  std::vector<BwaswSeed> seeds(n_seeds);
  b->seed_index.resize(n_seeds);
  size_t blob_bytes = 0;
  for (uint32_t i = 0; i < n_seeds; i++) {
    const uint16_t* p = params + (size_t)i * 7;
    // device limits: uint8_t qlen with a `j <= qlen` loop, uint11_t tlen, seq_mem[2048], ap_int<10> prev (smithwaterman.cpp:75-135, :520-556)
    if (p[0] > ACCG_BWASW_MAX_QLEN || p[2] > ACCG_BWASW_MAX_QLEN) return ACCG_ERR_TOO_LONG;
    if (p[1] > ACCG_BWASW_MAX_TLEN || p[3] > ACCG_BWASW_MAX_TLEN) return ACCG_ERR_TOO_LONG;
    if ((uint32_t)p[0] + p[1] + p[2] + p[3] > 2048u) return ACCG_ERR_TOO_LONG;
    if ((uint32_t)p[4] + p[0] + p[2] > 511u) return ACCG_ERR_TOO_LONG;
    BwaswSeed& s = seeds[i];
    s.qlen[0] = p[0]; s.tlen[0] = p[1]; s.qlen[1] = p[2]; s.tlen[1] = p[3]; s.seed_len = p[4]; s.seed_qbeg = p[5]; s.pad_ = 0;
    b->seed_index[i] = p[6];
    s.q_off[0] = (uint32_t)blob_bytes;               blob_bytes += p[0];
    s.q_off[1] = (uint32_t)blob_bytes;               blob_bytes += p[2];
    blob_bytes = (blob_bytes + 3) & ~(size_t)3;
    s.t_off[0] = (uint32_t)blob_bytes;               blob_bytes += (p[1] + 3u) & ~3u;
    s.t_off[1] = (uint32_t)blob_bytes;               blob_bytes += (p[3] + 3u) & ~3u;
    if (blob_bytes > 0xFFFFFFF0ull) return ACCG_ERR_TOO_LONG;
    b->cells += (uint64_t)p[0] * p[1] + (uint64_t)p[2] * p[3];
  }
  // the blob is assembled straight in the context's pinned staging (all host threads), then goes over in one copy
  const size_t blob_size = blob_bytes + 16;
  void* stage_v = nullptr;
  ACCG_HIP(ctx_stage(ctx, blob_size, &stage_v));
  uint8_t* blob = (uint8_t*)stage_v;
#pragma omp parallel for schedule(static, 1024) num_threads(accg::host_threads())
  for (int64_t i = 0; i < (int64_t)n_seeds; i++) {
    const BwaswSeed& s = seeds[i];
    const uint8_t* src = seqs + seq_off[i];           // [left query][right query][left target][right target]
    const uint32_t len[4] = {s.qlen[0], s.qlen[1], s.tlen[0], s.tlen[1]};
    const uint32_t dst[4] = {s.q_off[0], s.q_off[1], s.t_off[0], s.t_off[1]};
    const uint32_t nxt = i + 1 < (int64_t)n_seeds ? seeds[i + 1].q_off[0] : (uint32_t)blob_size;
    for (int part = 0; part < 4; part++) {
      uint8_t* d = blob + dst[part];
      for (uint32_t k = 0; k < len[part]; k++) { const uint8_t c = src[k]; d[k] = c > 3 ? 4 : c; }
      const uint32_t end = part < 3 ? dst[part + 1] : nxt;      // alignment padding reads as N
      for (uint32_t k = dst[part] + len[part]; k < end; k++) blob[k] = 4;
      src += len[part];
    }
  }
  // one side of four seeds per wavefront: same K, similar target lengths (a wavefront runs as long as its longest member);
  // every left pass is queued before the right passes that read its records.  Order = (K descending, tlen descending, index
  // ascending): a counting sort on the 16-bit key, stable in the index.
  std::vector<BwaswWork> work;
  work.reserve((size_t)n_seeds / 2 + 64);
  std::vector<uint32_t> order(n_seeds), bucket(17u * 2048u + 1u);
  for (int side = 0; side < 2; side++) {
    auto kclass = [&](uint32_t x) { return (seeds[x].qlen[side] + 1 + 15) / 16; };
    auto key = [&](uint32_t x) { return (uint32_t)(16 - kclass(x)) * 2048u + (2047u - seeds[x].tlen[side]); };   // ascending key
    std::fill(bucket.begin(), bucket.end(), 0u);
    for (uint32_t x = 0; x < n_seeds; x++) bucket[key(x) + 1]++;
    for (size_t k = 1; k < bucket.size(); k++) bucket[k] += bucket[k - 1];
    for (uint32_t x = 0; x < n_seeds; x++) order[bucket[key(x)]++] = x;
    for (uint32_t i = 0; i < n_seeds;) {
      const int K = kclass(order[i]);
      accg_bwasw_batch::Launch L{K, side, (uint32_t)work.size(), 0};
      while (i < n_seeds && kclass(order[i]) == K) {
        BwaswWork w;
        for (int g = 0; g < 4; g++) w.seed[g] = (i < n_seeds && kclass(order[i]) == K) ? order[i++] : BWASW_EMPTY;
        work.push_back(w);
        L.n_work++;
      }
      b->launches.push_back(L);
    }
  }
  hipStream_t st = ctx->stream;
  ACCG_HIP(ctx->pool.get(blob_size, (void**)&b->d_blob));
  ACCG_HIP(ctx->pool.get(std::max<size_t>(n_seeds, 1) * sizeof(BwaswSeed), (void**)&b->d_seeds));
  ACCG_HIP(ctx->pool.get(std::max<size_t>(work.size(), 1) * sizeof(BwaswWork), (void**)&b->d_work));
  ACCG_HIP(ctx->pool.get(std::max<size_t>(n_seeds, 1) * 8 * sizeof(int16_t), (void**)&b->d_out));
  ACCG_HIP(hipMemcpyAsync(b->d_blob, blob, blob_size, hipMemcpyHostToDevice, st));
  if (n_seeds) {
    ACCG_HIP(hipMemcpyAsync(b->d_seeds, seeds.data(), n_seeds * sizeof(BwaswSeed), hipMemcpyHostToDevice, st));
    ACCG_HIP(hipMemcpyAsync(b->d_work, work.data(), work.size() * sizeof(BwaswWork), hipMemcpyHostToDevice, st));
  }
  ACCG_HIP(hipStreamSynchronize(st));                // the staging and the vectors go away / get reused
  sync_on_error.dismiss();
  *out = b.release();
  return ACCG_OK;
}
extern "C" uint64_t accg_bwasw_batch_cells(const accg_bwasw_batch* b) { return b ? b->cells : 0; }

extern "C" int accg_bwasw_batch_run(accg_bwasw_batch* b) {
  if (!b) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  BwaswArgs a;
  a.blob = b->d_blob; a.seeds = b->d_seeds; a.out = b->d_out;
  // The K classes of one side are independent kernels whose wavefronts run for hundreds of microseconds: they are spread over
  // the context's forked streams; the right-side pass forks after the left-side pass has joined.
  for (int side = 0; side < 2; side++) {
    ACCG_HIP(ctx_fork(b->ctx));
    int rr = 0;
    for (const auto& L : b->launches) {
      if (L.side != side) continue;
      a.work = b->d_work + L.work_off;
      ACCG_HIP(bwasw_launch(L.K, L.side, a, L.n_work, b->ctx->aux[rr++ % accg_ctx::N_AUX]));
    }
    ACCG_HIP(ctx_join(b->ctx));
  }
  return ACCG_OK;
}
extern "C" int accg_bwasw_batch_time(accg_bwasw_batch* b, int warmup, int iters, float* ms_per_run) {
  if (!b || !ms_per_run || iters <= 0 || warmup < 0) return ACCG_ERR_BAD_ARG;
  accg_ctx* c = b->ctx;
  int st;
  for (int i = 0; i < warmup; i++) if ((st = accg_bwasw_batch_run(b)) != ACCG_OK) return st;
  ACCG_HIP(hipEventRecord(c->ev0, c->stream));
  for (int i = 0; i < iters; i++) if ((st = accg_bwasw_batch_run(b)) != ACCG_OK) return st;
  ACCG_HIP(hipEventRecord(c->ev1, c->stream));
  ACCG_HIP(hipEventSynchronize(c->ev1));
  float ms = 0;
  ACCG_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *ms_per_run = ms / iters;
  return ACCG_OK;
}
extern "C" int accg_bwasw_batch_results(accg_bwasw_batch* b, int16_t* fields, int32_t* words) {
  if (!b) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(b->ctx->device));
  ACCG_HIP(hipStreamSynchronize(b->ctx->stream));
  if (!b->n) return ACCG_OK;
  std::vector<int16_t> raw((size_t)b->n * 8);
  ACCG_HIP(hipMemcpy(raw.data(), b->d_out, raw.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < b->n; i++) {
    const int16_t* r = raw.data() + (size_t)i * 8;
    if (fields) memcpy(fields + (size_t)i * 7, r, 7 * sizeof(int16_t));
    if (words) {                                      // the match stream of smithwaterman.cpp:666-670
      int32_t* w = words + (size_t)i * 5;
      w[0] = b->seed_index[i];
      w[1] = (int32_t)(((uint32_t)(uint16_t)r[0]) | ((uint32_t)(uint16_t)r[1] << 16));
      w[2] = (int32_t)(((uint32_t)(uint16_t)r[2]) | ((uint32_t)(uint16_t)r[3] << 16));
      w[3] = (int32_t)(((uint32_t)(uint16_t)r[4]) | ((uint32_t)(uint16_t)r[5] << 16));
      w[4] = (int32_t)(uint16_t)r[6];
    }
  }
  return ACCG_OK;
}
extern "C" void accg_bwasw_batch_destroy(accg_bwasw_batch* b) {
  if (!b) return;
  hipSetDevice(b->ctx->device);
  hipStreamSynchronize(b->ctx->stream);
  delete b;
}

// ---- the FPGA kernel's own buffers ------------------------------------------------------------------------------------
// sw_top(input, output, pac_input, size) (bwa-sw/sdaccel/smithwaterman.cpp:1046-1054) takes the host's int stream as it is:
// per read  [index of the next read's record][read_len][ceil(read_len/8) ints, 8 bases each, 4 bits, first base on top]
//           [chain_num] then per chain [rmax0 lo][rmax0 hi][rmax1 lo][rmax1 hi][seed_num] and per seed
//           [seed_index][rbeg lo][rbeg hi][qbeg][seed_len]                       (data_parse :311-458, main_cl.cpp:73-90),
// and the 2-bit packed reference (16 bases per int, base k at bits 2k; read_proc :727-757).  The sequences of a seed are cut
// out as read_proc does (:775-806): left query = read[qbeg-1 .. 0], right query = read[qbeg+len ..], left target =
// ref[rbeg-1 .. rmax0], right target = ref[rbeg+len .. rmax1).  Results: five ints per seed in input order.
extern "C" int accg_bwasw_records(accg_ctx* ctx, const int32_t* input, int64_t size, const uint32_t* pac, uint64_t pac_words,
                                  int32_t* results, int64_t results_cap, int64_t* n_tasks) {
  if (!ctx) return ACCG_ERR_NOT_INITIALISED;
  if (!input || size < 0 || !pac || !n_tasks) return ACCG_ERR_BAD_ARG;
  std::vector<uint8_t> seqs;
  std::vector<uint32_t> off;
  std::vector<uint16_t> par;
  std::vector<uint8_t> read, ref;
  int64_t k = 0;
  auto need = [&](int64_t m) { return k + m <= size; };
  while (k < size) {
    if (!need(2)) return ACCG_ERR_BAD_WIRE;
    k++;                                                   // index of the next record (the device's readEndIndex)
    const int64_t read_len = input[k++];
    if (read_len < 0 || read_len > 511 || !need((read_len + 7) / 8 + 1)) return ACCG_ERR_BAD_WIRE;   // read_seq[512]
    read.resize((size_t)read_len);
    for (int64_t i = 0; i < read_len; i++) read[(size_t)i] = (uint8_t)(((uint32_t)input[k + i / 8] >> (28 - 4 * (i & 7))) & 0xF);
    k += (read_len + 7) / 8;
    const int64_t chain_num = input[k++];
    if (chain_num < 0) return ACCG_ERR_BAD_WIRE;
    for (int64_t c = 0; c < chain_num; c++) {
      if (!need(5)) return ACCG_ERR_BAD_WIRE;
      const uint64_t rmax0 = (uint32_t)input[k] | ((uint64_t)(uint32_t)input[k + 1] << 32);
      const uint64_t rmax1 = (uint32_t)input[k + 2] | ((uint64_t)(uint32_t)input[k + 3] << 32);
      k += 4;
      const int64_t seed_num = input[k++];
      if (seed_num < 0 || rmax1 < rmax0 || rmax1 - rmax0 > 2048 || (rmax1 >> 4) >= pac_words || !need(5 * seed_num)) return ACCG_ERR_BAD_WIRE;
      ref.resize((size_t)(rmax1 - rmax0));                 // chain_rseqs[2048]
      for (uint64_t p = rmax0; p < rmax1; p++) ref[(size_t)(p - rmax0)] = (uint8_t)((pac[p >> 4] >> ((p & 15) << 1)) & 3);
      for (int64_t sd = 0; sd < seed_num; sd++) {
        const int32_t seed_index = input[k];
        const uint64_t rbeg = (uint32_t)input[k + 1] | ((uint64_t)(uint32_t)input[k + 2] << 32);
        const int64_t qbeg = input[k + 3], seed_len = input[k + 4];
        k += 5;
        if (qbeg < 0 || seed_len < 0 || qbeg + seed_len > read_len || rbeg < rmax0 || rbeg + (uint64_t)seed_len > rmax1) return ACCG_ERR_BAD_WIRE;
        const int64_t lq = qbeg, rq = read_len - qbeg - seed_len, lr = (int64_t)(rbeg - rmax0), rr = (int64_t)(rmax1 - rbeg) - seed_len;
        if (lq > 0xFFFF || lr > 0xFFFF || rq > 0xFFFF || rr > 0xFFFF) return ACCG_ERR_TOO_LONG;
        off.push_back((uint32_t)seqs.size());
        for (int64_t i = 0; i < lq; i++) seqs.push_back(read[(size_t)(lq - 1 - i)]);
        for (int64_t i = 0; i < rq; i++) seqs.push_back(read[(size_t)(qbeg + seed_len + i)]);
        for (int64_t i = 0; i < lr; i++) seqs.push_back(ref[(size_t)(lr - 1 - i)]);
        for (int64_t i = 0; i < rr; i++) seqs.push_back(ref[(size_t)(lr + seed_len + i)]);
        const uint16_t p7[7] = {(uint16_t)lq, (uint16_t)lr, (uint16_t)rq, (uint16_t)rr, (uint16_t)seed_len, (uint16_t)qbeg, (uint16_t)seed_index};
        par.insert(par.end(), p7, p7 + 7);
      }
    }
  }
  *n_tasks = (int64_t)off.size();
  if (!results) return ACCG_OK;                            // size query (main_cl.cpp:73-90 counts the tasks the same way)
  if (results_cap < *n_tasks * 5) return ACCG_ERR_BAD_ARG;
  seqs.resize(seqs.size() + 8, 4);
  accg_bwasw_batch* b = nullptr;
  int st = accg_bwasw_batch_create(ctx, (uint32_t)off.size(), seqs.data(), off.data(), par.data(), &b);
  if (st != ACCG_OK) return st;
  st = accg_bwasw_batch_run(b);
  if (st == ACCG_OK) st = accg_bwasw_batch_results(b, nullptr, results);
  accg_bwasw_batch_destroy(b);
  return st;
}

// FM-index construction on the device: suffix array by prefix doubling (rocPRIM radix sort), BWT, BWA's block layout.
//
// The reference never builds an index: it loads one made by `bwa index` through libbwa (smem/main.cpp:434, bwa_idx_load) and
// hands the kernel the raw block array plus {primary, L2[0..4], blocks} (smem/main.cpp:217-222).  A synthetic genome
// (BASELINE.json configs[4]) therefore needs a constructor of its own; this is it.  What it produces is
// BWA's layout as smem/host/baseline.cpp:26-37 reads it: text = genome ++ reverse complement over {0,1,2,3}; BWT of text$
// with the sentinel row removed and its position kept as `primary`; per 128 symbols 4 x uint64 counts of A,C,G,T in front of
// the block, then 8 x uint32 of 16 two-bit symbols, first symbol in the top bits.
//
// Setup code, not the hot path: a full radix sort of (rank[i], rank[i+h]) keys per doubling round -- a random genome is done
// after the first or second round (21 bases already tell 1.3e8 suffixes apart), repeats just take more rounds.
#include <string.h>
#include <cstring>
#include <algorithm>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "accg_internal.h"

using namespace accg;

namespace {

constexpr int SYM_PER_KEY = 21;      // 3 bits per symbol (0 = past the end, 1..4 = A C G T) in a 63-bit key

__global__ void k_text(const uint8_t* g, uint64_t G, uint8_t* text) {      // text[0..2G) = codes + 1, text[2G] = 0 (sentinel)
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < G) { const uint8_t c = g[i] & 3; text[i] = c + 1; text[2 * G - 1 - i] = (3 - c) + 1; }
  if (i == 0) text[2 * G] = 0;
}
__global__ void k_first_keys(const uint8_t* text, uint64_t N, uint64_t* key, uint32_t* idx) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  uint64_t k = 0;
  for (int j = 0; j < SYM_PER_KEY; j++) k = (k << 3) | (i + j < N ? text[i + j] : 0);
  key[i] = k; idx[i] = (uint32_t)i;
}
__global__ void k_flags(const uint64_t* key, uint64_t N, uint32_t* flag) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < N) flag[j] = (j == 0 || key[j] != key[j - 1]) ? 1u : 0u;
}
__global__ void k_scatter_rank(const uint32_t* sa, const uint32_t* rank_sorted, uint64_t N, uint32_t* rank) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < N) rank[sa[j]] = rank_sorted[j];
}
__global__ void k_next_keys(const uint32_t* rank, uint64_t N, uint64_t h, uint64_t* key, uint32_t* idx) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  key[i] = ((uint64_t)rank[i] << 32) | (i + h < N ? rank[i + h] : 0u);     // ranks start at 1: 0 = past the end
  idx[i] = (uint32_t)i;
}
__global__ void k_bwt(const uint8_t* text, const uint32_t* sa, uint64_t N, uint8_t* b, unsigned long long* primary) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const uint32_t p = sa[j];
  if (p == 0) { *primary = j; b[j] = 0xFF; } else b[j] = text[p - 1] - 1;
}

struct Bufs {
  void* p[8] = {};
  ~Bufs() { for (void* q : p) if (q) (void)hipFree(q); }
};

}  // namespace

extern "C" uint64_t accg_smem_index_words(uint64_t n_genome) { return (2 * n_genome + 127) / 128 * 16; }

extern "C" int accg_smem_index_build(accg_ctx* ctx, const uint8_t* genome_codes, uint64_t n_genome, uint32_t* bwt_out,
                                     uint64_t bwt_words_cap, uint64_t* bwt_para) {
  if (!ctx) return ACCG_ERR_NOT_INITIALISED;
  if (!genome_codes || !bwt_out || !bwt_para || n_genome == 0) return ACCG_ERR_BAD_ARG;
  if (n_genome > (1ull << 30)) return ACCG_ERR_TOO_LONG;                 // suffix indices and ranks are 32-bit
  const uint64_t n = 2 * n_genome, N = n + 1, words = accg_smem_index_words(n_genome);
  if (bwt_words_cap < words) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  Bufs B;
  uint8_t *d_g, *d_text, *d_b; uint64_t *d_key0, *d_key1; uint32_t *d_idx0, *d_idx1, *d_rank;
  ACCG_HIP(hipMalloc(&B.p[0], n_genome)); d_g = (uint8_t*)B.p[0];
  ACCG_HIP(hipMalloc(&B.p[1], N + 64)); d_text = (uint8_t*)B.p[1];
  ACCG_HIP(hipMalloc(&B.p[2], N * 8)); d_key0 = (uint64_t*)B.p[2];
  ACCG_HIP(hipMalloc(&B.p[3], N * 8)); d_key1 = (uint64_t*)B.p[3];
  ACCG_HIP(hipMalloc(&B.p[4], N * 4)); d_idx0 = (uint32_t*)B.p[4];
  ACCG_HIP(hipMalloc(&B.p[5], N * 4)); d_idx1 = (uint32_t*)B.p[5];
  ACCG_HIP(hipMalloc(&B.p[6], N * 4 + 16)); d_rank = (uint32_t*)B.p[6];
  size_t t_sort = 0, t_scan = 0;
  ACCG_HIP(rocprim::radix_sort_pairs(nullptr, t_sort, d_key0, d_key1, d_idx0, d_idx1, N, 0, 64, s));
  ACCG_HIP(rocprim::inclusive_scan(nullptr, t_scan, d_idx0, d_idx0, N, rocprim::plus<uint32_t>(), s));
  const size_t t_bytes = std::max(t_sort, t_scan);
  ACCG_HIP(hipMalloc(&B.p[7], t_bytes + 16 + sizeof(unsigned long long)));
  void* d_tmp = B.p[7];
  const unsigned T = 256;
  const unsigned gN = (unsigned)((N + T - 1) / T), gG = (unsigned)((n_genome + T - 1) / T);
  ACCG_HIP(hipMemcpyAsync(d_g, genome_codes, n_genome, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_text, dim3(gG), dim3(T), 0, s, d_g, n_genome, d_text);
  hipLaunchKernelGGL(k_first_keys, dim3(gN), dim3(T), 0, s, d_text, N, d_key0, d_idx0);
  uint64_t h = SYM_PER_KEY;
  int end_bit = 63;
  for (int round = 0;; round++) {
    size_t tb = t_bytes;
    ACCG_HIP(rocprim::radix_sort_pairs(d_tmp, tb, d_key0, d_key1, d_idx0, d_idx1, N, 0, end_bit, s));
    // d_key1 sorted, d_idx1 = suffix order so far.  New ranks: 1 + number of key changes in front.
    hipLaunchKernelGGL(k_flags, dim3(gN), dim3(T), 0, s, d_key1, N, d_idx0);
    tb = t_bytes;
    ACCG_HIP(rocprim::inclusive_scan(d_tmp, tb, d_idx0, d_idx0, N, rocprim::plus<uint32_t>(), s));
    uint32_t distinct = 0;
    ACCG_HIP(hipMemcpyAsync(&distinct, d_idx0 + (N - 1), 4, hipMemcpyDeviceToHost, s));
    ACCG_HIP(hipStreamSynchronize(s));
    if (distinct == N) break;
    if (round > 40) return ACCG_ERR_BAD_ARG;           // cannot happen: h doubles past N long before
    hipLaunchKernelGGL(k_scatter_rank, dim3(gN), dim3(T), 0, s, d_idx1, d_idx0, N, d_rank);
    hipLaunchKernelGGL(k_next_keys, dim3(gN), dim3(T), 0, s, d_rank, N, h, d_key0, d_idx0);
    h *= 2;
    end_bit = 64;
  }
  ACCG_HIP(hipGetLastError());
  // BWT bytes in suffix order (one per row, the row of suffix 0 marked), packed on the host
  d_b = (uint8_t*)d_key0;
  unsigned long long* d_primary = (unsigned long long*)((uint8_t*)d_tmp + (t_bytes + 15) / 16 * 16);
  hipLaunchKernelGGL(k_bwt, dim3(gN), dim3(T), 0, s, d_text, d_idx1, N, d_b, d_primary);
  std::vector<uint8_t> b(N);
  unsigned long long primary = 0;
  ACCG_HIP(hipMemcpyAsync(b.data(), d_b, N, hipMemcpyDeviceToHost, s));
  ACCG_HIP(hipMemcpyAsync(&primary, d_primary, sizeof primary, hipMemcpyDeviceToHost, s));
  ACCG_HIP(hipStreamSynchronize(s));
  if (primary >= N || b[primary] != 0xFF) return ACCG_ERR_HIP;
  const uint64_t nblk = words / 16;
  auto sym = [&](uint64_t k) -> uint32_t { return k < n ? b[k < primary ? k : k + 1] : 0u; };   // row `primary` removed
  std::vector<uint64_t> cnt((nblk + 1) * 4, 0);
#pragma omp parallel for schedule(static) num_threads(accg::host_threads())
  for (int64_t blk = 0; blk < (int64_t)nblk; blk++) {
    uint64_t c[4] = {0, 0, 0, 0};
    uint32_t* w = bwt_out + blk * 16 + 8;
    for (int j = 0; j < 8; j++) {
      uint32_t v = 0;
      for (int e = 0; e < 16; e++) {
        const uint64_t k = (uint64_t)blk * 128 + j * 16 + e;
        const uint32_t x = sym(k);
        v |= x << (30 - 2 * e);
        if (k < n) c[x]++;
      }
      w[j] = v;
    }
    for (int x = 0; x < 4; x++) cnt[(blk + 1) * 4 + x] = c[x];
  }
  for (uint64_t blk = 0; blk < nblk; blk++) {
    for (int x = 0; x < 4; x++) {
      const uint64_t before = cnt[blk * 4 + x];
      memcpy(bwt_out + blk * 16 + 2 * x, &before, 8);
      cnt[(blk + 1) * 4 + x] += before;
    }
  }
  bwt_para[0] = primary;
  bwt_para[1] = 0;
  for (int x = 0; x < 4; x++) bwt_para[2 + x] = bwt_para[1 + x] + cnt[nblk * 4 + x];   // L2[c+1] = symbols <= c
  bwt_para[6] = nblk;
  return ACCG_OK;
}

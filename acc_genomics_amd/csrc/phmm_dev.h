// Device/host shared declarations for the PairHMM kernels (internal; the public ABI is include/accg.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace accg {

constexpr int PHMM_LANES = 16;          // lanes per read (one DPP row)
constexpr int PHMM_GROUPS = 4;          // reads per wavefront
constexpr int PHMM_MAX_K = 16;          // rows per lane
constexpr int PHMM_STREAM_MAX = 4096;   // haplotype stream entries per work item (bubbles included)
constexpr int PHMM_HAPS_MAX = 48;       // haplotypes per work item
constexpr int PHMM_NCHAR = 5;           // A C G T N

// Dynamic LDS of one wavefront, in bytes, as laid out by phmm_kernel:
//   [dist table: NCHAR x QT x 64 lanes x 16 B][stream: u16 x (15 + cap + 48)][y0: T x (HAPS_MAX+1)]
//   [hcol: u32 x (HAPS_MAX+1)][bpos: u32 x (HAPS_MAX+2)]
constexpr int phmm_qt(int K, int elem_bytes) { return (K * elem_bytes + 15) / 16; }
constexpr size_t phmm_lds_bytes(int K, int elem_bytes, int stream_cap) {
  size_t tab = (size_t)PHMM_NCHAR * phmm_qt(K, elem_bytes) * 1024;
  size_t str = (((size_t)(15 + stream_cap + 48) * 2) + 15) / 16 * 16;
  size_t y0 = ((size_t)(PHMM_HAPS_MAX + 1) * elem_bytes + 15) / 16 * 16;
  size_t idx = ((size_t)(2 * PHMM_HAPS_MAX + 3) * 4 + 15) / 16 * 16;
  return tab + str + y0 + idx;
}
constexpr uint32_t PHMM_NO_READ = 0xFFFFFFFFu;

// One wavefront's job: up to four reads against a list of haplotypes.
struct PhmmWork {
  uint32_t read[PHMM_GROUPS];  // global read index or PHMM_NO_READ
  uint32_t hap_off;            // first entry of this job in PhmmArgs::hap_ids
  uint32_t n_haps;
  uint32_t pad_[2];
};

struct SeqRef { uint32_t off, len; };   // byte offset into the blob, length

// Tables of pairhmm/xlnx/host/Context.h, precomputed on the host so that the device sees the
// same bits the CPU path multiplies with.
template <typename T>
struct PhmmTables {
  const T* ph;     // ph2pr[q] = 10^(-q/10)            Context.h:105-107,145-147
  const T* omph;   // 1 - ph2pr[q]                     baseline_impl.cpp:54, avx-pairhmm-template.h:156
  const T* phd3;   // ph2pr[q] / 3                     baseline_impl.cpp:83, avx-pairhmm-template.h:158
  const T* m2m;    // matchToMatchProb (triangular)    Context.h:50-61
  T init;          // INITIAL_CONSTANT 2^120 | 2^1020  Context.h:109,149
};

template <typename T>
struct PhmmArgs {
  const uint8_t* rblob;       // concatenated wire-format read blobs
  const uint8_t* hblob;       // concatenated wire-format hap blobs
  const SeqRef* rd;           // per read: offset of _b inside rblob, length (fields follow at +len each)
  const uint32_t* rd_out;     // per read: index of out[read][hap 0]
  const SeqRef* hp;           // per hap
  const uint32_t* hp_local;   // per hap: column inside its region's output row
  const uint32_t* hap_ids;    // job hap lists
  const PhmmWork* work;
  T* out;
  const float* raw;           // rescue pass only: the fp32 results that decide which pairs are redone
  unsigned long long* n_rescued;  // rescue pass only: count of (read, hap) pairs below the threshold
  PhmmTables<T> tab;
};

// Launchers (phmm_kernel.hip). K = rows per lane, 1..PHMM_MAX_K.
// stream_cap = largest haplotype stream (entries, bubbles included) among the jobs of this launch.
hipError_t phmm_launch_f32(int K, bool strict, const PhmmArgs<float>& a, uint32_t work_base, uint32_t n_work, int stream_cap, hipStream_t s);
// fp64 rescue pass: same jobs as the fp32 pass; a wavefront redoes only the haplotypes for which one of
// its reads came out below MIN_ACCEPTED (host_type.h:21), and exits at once when there is none.
hipError_t phmm_launch_rescue_f64(int K, const PhmmArgs<double>& a, uint32_t work_base, uint32_t n_work, int stream_cap, hipStream_t s);
// fp64 over every pair of the jobs (tests, and FalconPairHMM's use_double=true path).
hipError_t phmm_launch_f64(int K, const PhmmArgs<double>& a, uint32_t work_base, uint32_t n_work, int stream_cap, hipStream_t s);
constexpr float PHMM_MIN_ACCEPTED = 1e-28f;   // host_type.h:21

}  // namespace accg

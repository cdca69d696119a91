// Device/host shared declarations for the PairHMM kernels (internal; the public ABI is include/accg.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace accg {

constexpr int PHMM_GROUPS = 8;          // reads per wavefront (8 lanes per read up to 127 bp, 16 up to 255, 32 up to 511, 64 up to 1023)
constexpr int PHMM_MAX_K = 16;          // rows per lane
constexpr int PHMM_STREAM_MAX = 4096;   // haplotype stream entries per work item (bubbles included)
constexpr int PHMM_HAPS_MAX = 48;       // haplotypes per work item

// Dynamic LDS of one wavefront, in bytes, as laid out by phmm_kernel:
//   [dist table: nchar x QT x 64 lanes x 16 B][y0: T x (haps_cap+1)][hcol: u32 x (haps_cap+1)]
//   [bpos: u32 x (haps_cap+2)][stream: u8 x (lpp-1 + stream_cap + lpp+20)]
// nchar = 4 (A C G T) or 5 (+N) -- the N slab is only carried when some haplotype of the batch has an N.
constexpr int phmm_qt(int K, int elem_bytes) { return (K * elem_bytes + 15) / 16; }
constexpr size_t phmm_align16(size_t x) { return (x + 15) / 16 * 16; }
// Bytes of one base's slab of the dist table.  General layout: QT 16-byte vectors per lane, quad q at q * 1024 + lane * 16.
// fp32 fast kernels (`compact`): the K % 4 rows behind the last full quad take 4 (one row) or 8 (two rows) bytes per lane
// instead of 16 -- at K = 13 that is 13 KB instead of 16 KB per wavefront, i.e. 11 instead of 9 resident wavefronts per CU.
constexpr int phmm_tail_stride(int K, int elem_bytes = 4) {
  return elem_bytes == 8 ? (K % 2 ? 8 : 0) : (K % 4 == 1 ? 4 : K % 4 == 2 ? 8 : K % 4 == 3 ? 16 : 0);
}
constexpr int phmm_slab_bytes(int K, int elem_bytes, bool compact) {
  return compact ? (K / (16 / elem_bytes)) * 1024 + 64 * phmm_tail_stride(K, elem_bytes) : phmm_qt(K, elem_bytes) * 1024;
}
// Kernels whose column is written in assembly (phmm_kernel_impl.h) and use the compact slabs: the fast (contracted) arithmetic in fp32
// for every K, in fp64 (the rescue pass) up to K = 10 -- beyond that an fp64 lane's 9 K register pairs do not fit the 256
// architectural VGPRs an inline-assembly operand can live in.
constexpr int PHMM_ASM_MAX_K_F64 = 10;
constexpr bool phmm_is_compact(int elem_bytes, bool strict, int K = 1) { return !strict && (elem_bytes == 4 || K <= PHMM_ASM_MAX_K_F64); }
// striped (reads longer than 64 x 16 - 1 bases): two carry arrays of one value per stream position behind the stream
// ... per wavefront: everything but the dist table ([y0][hcol][bpos][stream], and the carry arrays of a striped read)
// (+ 64 bytes at its end: {output row, read index} of the wavefront's up to eight reads, which only the off-the-hot-path code of the
// sweep needs -- held there they take no registers in the sweep)
constexpr size_t PHMM_STASH_BYTES = 64;
// stream region: [pad so that entry lpp-1 is 16-byte aligned: < 16][lpp-1 leading entries][stream_cap][lpp + 20 drain entries, copied in
// 16-byte units: + 16]
constexpr size_t PHMM_STREAM_SLACK = 24 + 16 + 16 + 8;
constexpr size_t phmm_wave_area_bytes(int elem_bytes, int stream_cap, int haps_cap, int lpp = 16, bool striped = false) {
  return phmm_align16((size_t)(haps_cap + 1) * elem_bytes) + phmm_align16((size_t)(2 * haps_cap + 3) * 4) +
         phmm_align16((size_t)2 * lpp + stream_cap + PHMM_STREAM_SLACK) + (striped ? phmm_align16((size_t)2 * elem_bytes * (stream_cap + 2 * lpp + 24)) : 0) +
         PHMM_STASH_BYTES;
}
// wg = wavefronts per workgroup: 1, or 2 that hold the SAME reads against different runs of haplotypes and share one dist table
// (the table depends on the reads only): [dist table][wave 0's area][wave 1's area]
constexpr size_t phmm_lds_bytes(int K, int elem_bytes, int nchar, int stream_cap, int haps_cap, int lpp = 16, bool compact = false,
                                bool striped = false, int wg = 1) {
  return (size_t)nchar * phmm_slab_bytes(K, elem_bytes, compact && !striped) +
         (size_t)wg * phmm_wave_area_bytes(elem_bytes, stream_cap, haps_cap, lpp, striped);
}
constexpr uint32_t PHMM_NO_READ = 0xFFFFFFFFu;

// One wavefront's job: up to eight reads against a list of haplotypes.
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

struct PhmmHapDesc { uint32_t off, len, col, id; };
// Streams laid out at batch creation (phmm_host.cpp) for the kernels that do not build theirs: one per run of haplotypes,
// [marker][base codes of haplotype 0][marker][...]...[marker][zeros], marker = nchar, codes A C G T N = 0..4; PhmmWork::pad_[0] = its
// offset in 16-byte units, pad_[1] = its length up to and including the last marker.  Zeros behind it: at least PHMM_STREAM_TAIL.
constexpr uint32_t PHMM_STREAM_TAIL = 64 + 20 + 16;
// Per flat row of a read's wavefront slot, written by phmm_prepare_rows at the start of every pass for the five-operation sweep
// (three arrays of 16-byte records): what the sweep's registers and dist table hold for that row, so that a job's prologue is straight
// loads -- no table lookups, no divisions, no selects.  A read that runs on LPP lanes x K rows has LPP * K records, clones of row 0
// included, in the order the lanes load them: flat row f = lane * K + k at row0[read] + k * LPP + lane (for a given k the lanes
// of a read fetch LPP consecutive records).  shape[read] = K | LPP << 8, 0 for a read whose wavefront does not run that sweep.
//   coef = {a, b, pYY, cx}: a = (pMX[r] pGM[r+1]) / pMM[r+1], b = (pMY[r] pGM[r+1]) / pMM[r+1] (0 for the last row),
//                           cx = (pXX[r] pMX[r-1]) / pMX[r] (0 for the first row); a clone: {0, 0, 1, 0}, the last clone's b = (1 pGM[1]) / pMM[1]
//   dist = dist(r, A / C / G / T) x pMM[r]; a clone: 0
//   misc = {dist(r, N) x pMM[r], pMX[r], 0, 0}; a clone: 0
struct PhmmRowRecs { float4* coef; float4* dist; float4* misc; const uint32_t* row0; const uint32_t* shape; };   // offset and length of the bases in hblob, column in the region's output row, global hap index

constexpr uint32_t PHMM_RESCUE_GRID_DEFAULT = 4096;
template <typename T>
struct PhmmArgs {
  const uint8_t* rblob;       // concatenated wire-format read blobs
  const uint8_t* hblob;       // concatenated wire-format hap blobs
  const SeqRef* rd;           // per read: offset of _b inside rblob, length (fields follow at +len each)
  const uint32_t* rd_out;     // per read: index of out[read][hap 0]
  const SeqRef* hp;           // per hap
  const uint32_t* hp_local;   // per hap: column inside its region's output row
  const PhmmHapDesc* hap_desc;  // job hap lists, one descriptor per entry (what hp / hp_local hold for that haplotype: one level of loads less per job)
  const PhmmWork* work;
  T* out;
  const float* raw;           // rescue pass only: the fp32 results that decide which pairs are redone
  unsigned long long* n_rescued;  // rescue pass only: count of (read, hap) pairs below the threshold
  const uint32_t* job_count;      // rescue pass only: number of valid jobs (device-side); blocks beyond it exit
  const uint32_t* job_map;        // rescue pass, strict re-run of the jobs a fast launch listed: job index = job_map[i] (else null)
  uint32_t* redo_count;           // rescue pass, fast mode: jobs that produced a result below PHMM_F64_TINY are appended to
  uint32_t* redo_list;            //   redo_list (their index), counted in redo_count, and re-run in the strict form by a second launch
  int is_redo;                    // that second launch: do not count the rescued pairs again
  uint32_t* read_flag;            // fp32 pass: set to 1 for a read with a result below MIN_ACCEPTED (nullable)
  PhmmTables<T> tab;
  int nchar;                  // 4 or 5 slabs in the dist table
  int stream_cap, haps_cap;   // LDS capacities of this launch (entries / haplotypes per job)
  int lds_min;                // host side only: ask for at least this much dynamic LDS (pins the resident wavefronts per CU)
  uint32_t* zero_words;       // fp32 pass: block 0 of every launch zeroes these n_zero words (rescue job counts, redo counts, the counter of
  int n_zero;                 //   rescued pairs) -- the previous pass is through with them, this pass's planner comes after the sweep (nullable)
  unsigned long long* clock_out;  // fp32 pass (nullable): the first wavefront of a launch leaves {shader-clock ticks, 100 MHz wall-clock ticks} of its job here
  int fair;                   // assembly sweep: 1 = a wavefront lowers its issue priority as it advances (see phmm_job), 0 = never
  PhmmRowRecs rec;            // five-operation sweep: per-row records (phmm_prepare_rows) ...
  const uint8_t* streams;     // ... and the streams laid out at batch creation
};
// one block per read: fills PhmmArgs::rec for reads [0, n_reads); state (nullable): words to zero on the way
hipError_t phmm_prepare_rows_launch(const PhmmArgs<float>& a, uint32_t n_reads, uint32_t* state, uint32_t state_words, hipStream_t s);

// ---- fp64 rescue planning (device side, no host round trip) -----------------------------------------
// Reads that underflowed in fp32 against at least one haplotype are regrouped into new wavefront jobs so that
// the fp64 pass only carries those reads (FalconPairHMM.cpp:636-652 redoes exactly the underflowed pairs).
// (lanes per read, K) of the fp64 rescue kernels by rows (= read length + 1).  Up to 160 rows: 16 lanes per read and K up to 10 -- the
// per-column work of a lane (hand-off, stream and table reads, loop) is spread over K rows, and 32 lanes x K <= 5 measured no faster
// although it doubles the resident wavefronts (configs[3]: 3.0 against 2.9 ms).  Beyond that an fp64 lane runs out of registers (K = 12:
// 412 with spills, one wavefront per SIMD), so longer reads are spread over 32 or 64 lanes with K between 5 and 8.
constexpr int PHMM_RESCUE_CLASSES = 16;      // the last two: (64,16) up to 1024 rows, and (64,16) in stripes for reads of 1024 bases and more
constexpr int phmm_rescue_lpp(int cls) {
  constexpr int lp[PHMM_RESCUE_CLASSES] = {16, 16, 16, 16, 16, 16, 32, 32, 32, 32, 64, 64, 64, 64, 64, 64};
  return lp[cls];
}
constexpr int phmm_rescue_k(int cls) {
  constexpr int ks[PHMM_RESCUE_CLASSES] = {2, 4, 5, 6, 7, 8, 5, 6, 7, 8, 5, 6, 7, 8, 16, 16};
  return ks[cls];
}
__host__ __device__ inline void phmm_rescue_shape(int cls, int* lpp, int* K) { *lpp = phmm_rescue_lpp(cls); *K = phmm_rescue_k(cls); }
// Merged rescue launches (fast mode, five-operation form): the classes with K <= 8 go out as two launches by register budget --
// window 0: K <= 5 (at most 124 registers, four wavefronts per SIMD), window 1: K = 6..8 (at most 168, three) -- each walking its
// classes' job arrays one after the other, the longest rows first.  PHMM_RESCUE_MERGED = classes covered (the two (64,16) ones keep
// launches of their own).
constexpr int PHMM_RESCUE_MERGED = PHMM_RESCUE_CLASSES - 2;
constexpr int PHMM_RESCUE_WIN0_N = 5, PHMM_RESCUE_WIN1_N = 9;
constexpr int phmm_rescue_win_class(int win, int i) {
  constexpr int w0[PHMM_RESCUE_WIN0_N] = {10, 6, 2, 1, 0};
  constexpr int w1[PHMM_RESCUE_WIN1_N] = {13, 12, 11, 9, 8, 7, 5, 4, 3};
  return win == 0 ? w0[i] : w1[i];
}
constexpr int phmm_rescue_window(int cls) { return cls >= PHMM_RESCUE_MERGED ? -1 : phmm_rescue_k(cls) <= 5 ? 0 : 1; }
struct PhmmRescueSet {
  uint32_t off[PHMM_RESCUE_CLASSES + 1];   // class c's items at [off[c], off[c + 1]) of PhmmArgs::work
  const uint32_t* counts;                  // items the planner wrote per class
};
__host__ __device__ inline bool phmm_rescue_striped(int cls) { return cls == PHMM_RESCUE_CLASSES - 1; }
__host__ __device__ inline void phmm_rescue_class(uint32_t len, int* cls, int* lpp, int* K) {
  const uint32_t rows = len + 1;
  int c = PHMM_RESCUE_CLASSES - 1;
  if (rows <= 1024) {
    c = 0;
    for (;;) { phmm_rescue_shape(c, lpp, K); if ((uint32_t)(*lpp * *K) >= rows) break; c++; }
  }
  *cls = c;
  phmm_rescue_shape(c, lpp, K);
}
struct PhmmRegionDev { uint32_t read0, n_reads, chunk0, n_chunks, n_haps, pad_; };
struct PhmmChunkDev { uint32_t ids0, n; };
struct PhmmPlanArgs {
  const PhmmRegionDev* regions;
  const PhmmChunkDev* chunks;
  const uint32_t* sorted_reads;   // per region: its reads by descending length (global ids), at [read0, read0 + n_reads)
  const SeqRef* rd;
  const uint32_t* rd_out;
  uint32_t* read_flag;            // set by the fp32 pass; the planner clears the flags it has read (the next pass starts clean without a memset)
  PhmmWork* jobs;                 // the classes' job arrays back to back: class c at [class_off[c], class_off[c + 1])
  uint32_t* counts;               // jobs written per class
  uint32_t* flagged;              // scratch, one slot per read
  uint32_t class_off[PHMM_RESCUE_CLASSES + 1];
  uint32_t* host_flag;            // nullable, host-visible: set to 1 when some read of the batch is flagged (phmm_host.cpp: rescue probe)
  uint32_t pairs;                 // 1: every group's items come in pairs for workgroups of two wavefronts (an odd last one is followed by an empty item)
};
hipError_t phmm_rescue_plan_launch(const PhmmPlanArgs& p, uint32_t n_regions, hipStream_t s);

// Launchers (phmm_kernel_impl.h). K = rows per lane, 1..PHMM_MAX_K.
// a.stream_cap / a.haps_cap = largest haplotype stream (entries, bubbles included) / haplotype count among the jobs of this launch.
constexpr int PHMM_K8_DEFAULT = 13;   // 8 lanes per read while the rows fit K <= 13 (beyond that the 2-wave occupancy costs more than it saves)
void phmm_pick(uint32_t read_len, int* lpp, int* K, int max_k8 = 0);        // reads of 1024 bases and more: (64, 16), swept in stripes
inline bool phmm_striped(uint32_t read_len) { return read_len + 1 > 1024; }
// form (fast mode only): 7, 6 or 5 operations per cell; a wavefront runs the six- / five-operation form only if all of its reads pass
// the form's range test (phmm_host.cpp: phmm_read_form)
// wg = 2 (fast mode, not striped): consecutive pairs of work items hold the same reads and run as one workgroup of two wavefronts
// sharing the dist table in LDS (n_work even; a pair's second item may be empty, n_haps = 0)
hipError_t phmm_launch_f32(int K, int lpp, bool strict, int form, bool striped, const PhmmArgs<float>& a, uint32_t work_base, uint32_t n_work, hipStream_t s,
                           int wg = 1);
// The six-operation form keeps X divided by the row's pMX: Xs[r] = M[r-1] + c[r] Xs[r-1], c[r] = pXX[r] pMX[r-1] / pMX[r].
// Xs is bounded by max(M) * F, F[r] = 1 + c[r] F[r-1]; M never exceeds INIT / H <= 2^120, so F <= 32 leaves a factor of 8 to
// FLT_MAX.  Reads whose insertion qualities jump by more than ~7 dB from one base to the next push F up and stay in the
// seven-operation form.
// ... and the prepared five-operation sweep for several (lanes, K) classes in one launch: K in one of the windows {2..5}, {6..13},
// 8 or 16 lanes per read, one workgroup size; lds_bytes = the largest phmm_lds_bytes of the classes
hipError_t phmm_launch_f32_multi(int k_lo, int k_hi, size_t lds_bytes, const PhmmArgs<float>& a, uint32_t work_base, uint32_t n_work, hipStream_t s, int wg);
inline int phmm_multi_window(int K) { return K >= 6 && K <= 13 ? 2 : K >= 2 && K <= 5 ? 1 : 0; }   // 0: no merged launch for this K
constexpr float PHMM_X6_MAX_F = 32.f;
// The five-operation form additionally keeps Y divided by the row's pMY (Ys <= max(M) / (1 - pYY)) and the diagonal term divided
// by the consumer row's pMM: every pYY <= 31/32 and every pMM >= 1/16 keep both within a factor 32 / 16 of the unscaled values.
constexpr float PHMM_X5_MAX_YY = 0.96875f, PHMM_X5_MIN_MM = 0.0625f;
// fp64 rescue pass: same jobs as the fp32 pass; a wavefront redoes only the haplotypes for which one of
// its reads came out below MIN_ACCEPTED (host_type.h:21), and exits at once when there is none.
// strict: the operation order of compute_full_prob_baseline<double> (bit-exact with it); else the 7-op contraction with a redo in
// that order of every job that produced a result below PHMM_F64_TINY
hipError_t phmm_launch_rescue_f64(int K, int lpp, bool strict, bool striped, const PhmmArgs<double>& a, uint32_t work_base, uint32_t n_work, hipStream_t s,
                                  uint32_t grid_cap = PHMM_RESCUE_GRID_DEFAULT, bool form5 = false, int wg = 1);
// one window of merged classes (a.work = the whole job array, a.redo_count / a.redo_list = ONE list of absolute item indices for all merged
// classes); lds_bytes = the largest request among the window's classes that have jobs
hipError_t phmm_launch_rescue_multi(int window, int wg, size_t lds_bytes, const PhmmArgs<double>& a, const PhmmRescueSet& rs, uint32_t grid, hipStream_t s);
// the strict re-run of the items that list names, whatever their class (a.redo_count / a.redo_list as above)
hipError_t phmm_launch_redo_multi(size_t lds_bytes, const PhmmArgs<double>& a, const PhmmRescueSet& rs, uint32_t grid, hipStream_t s);
constexpr double PHMM_F64_TINY = 1e-280;      // x 2^1020 scaling included: 28 decades above the smallest normal double
// fp64 over every pair of the jobs (tests, and FalconPairHMM's use_double=true path).
hipError_t phmm_launch_f64(int K, int lpp, bool striped, const PhmmArgs<double>& a, uint32_t work_base, uint32_t n_work, hipStream_t s);
constexpr float PHMM_MIN_ACCEPTED = 1e-28f;   // host_type.h:21
constexpr int PHMM_RESCUE_GRID = 4096;        // wavefronts per rescue launch (they stride over the device-side job count)
constexpr int PHMM_REDO_GRID = 256;           // ... per strict re-run launch (nearly always nothing to do)

}  // namespace accg

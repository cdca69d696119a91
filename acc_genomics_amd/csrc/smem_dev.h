// Device/host shared declarations for the SMEM seeding kernel (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace accg {

struct alignas(16) SmemIntv { uint64_t x0, x1, x2, info; };   // bwtintv_t: {x[0], x[1], x[2], info}
// curr entries per read in the scratch: one bwt_smem1a_new call needs at most 256 (the fused kernel reuses them call by call); the
// three-kernel path keeps the lists of all calls of the first pass, whose entries add up to at most twice the read length
constexpr int SMEM_CURR_CAP = 512;
constexpr int SMEM_SCRATCH_ENTRIES = SMEM_CURR_CAP + 256;

// Prefix table: what SMEM_KTAB_L forward bwt_extend calls from a single base arrive at, for every string of that many bases
// (4^10 x 16 B = 16.8 MB for the last level, 22.4 MB in all).  A forward extension that starts on SMEM_KTAB_L clean bases takes
// its first intervals from here -- independent loads, one memory latency -- instead of 2 (SMEM_KTAB_L - 1) dependent lookups.
constexpr int SMEM_KTAB_L = 10;
__host__ __device__ constexpr uint32_t smem_ktab_off(int L) { return (uint32_t)(((1ull << (2 * L)) - 4) / 3); }   // entries in front of level L
constexpr uint32_t SMEM_KTAB_ENTRIES = (uint32_t)(((1ull << (2 * (SMEM_KTAB_L + 1))) - 4) / 3);

struct SmemArgs {
  const uint32_t* bwt;          // BWA block layout, 64 B per 128 symbols (smem/host/baseline.cpp:26-37)
  uint64_t primary, L2[5];
  const uint4* ktab;            // compact index only, may be null: bi-intervals {x0, x1, x2, 0} of every string of 1 .. SMEM_KTAB_L bases
                                //   (level L at smem_ktab_off(L), entry = the bases read as a base-4 number, first base most significant)
  uint32_t compact;             // 1: re-laid-out index, 32 B per 64 symbols = 4 x u32 cumulative counts + 4 x 16 symbols (smem_host.cpp)
  const uint8_t* seq;           // n_reads x seq_stride codes (0-3, >= 4 ambiguous)
  const uint8_t* seq_len;
  uint32_t seq_stride, n_reads;
  uint32_t read_words;          // LDS row per lane for its read: (longest read of the batch + 7) / 8 words, made odd
  SmemIntv* out;                // n_reads x max_out
  int32_t* mem_num;             // uncapped interval count per read
  uint32_t max_out;
  SmemIntv* scratch;            // SMEM_SCRATCH_ENTRIES intervals per thread (curr lists, then the back list), interleaved: element e of thread t at [e * n_threads + t]
  uint32_t n_threads;
  uint32_t* seg;                // three-kernel path: segment table of the first pass, 256 words per thread, interleaved the same way; null = the one-kernel form (default)
  uint32_t* nseg;               //   ... segments per thread
  uint32_t waves_per_cu;        // 0 = whatever fits; else an LDS request that admits this many wavefronts per CU
  uint32_t* queue;              // engine variant: next read of the launch (zeroed before it)
  uint32_t skip_pass3;          // fused kernel: leave the LAST-like third pass to smem_launch_pass3 on another stream (smem_host.cpp)
};

hipError_t smem_launch(const SmemArgs& a, uint32_t read_base, uint32_t n_reads, hipStream_t s);
// the third pass by itself (flat: one bwt_extend per iteration, 63 of 64 lanes busy), appending at a.mem_num[read] into a.out; and the
// kernel that appends such a separate list behind the first two passes' intervals (out, their counts num12; the final count goes to num),
// capped like every push
hipError_t smem_launch_pass3(const SmemArgs& a, uint32_t read_base, uint32_t n_reads, hipStream_t s);
hipError_t smem_launch_merge3(SmemIntv* out, int32_t* num, const int32_t* num12, uint32_t max_out, const SmemIntv* out3, const int32_t* num3, uint32_t max3,
                              uint32_t read_base, uint32_t n_reads, hipStream_t s);
// fills ktab (SMEM_KTAB_ENTRIES x uint4) level by level with the kernel's own bwt_extend; a.ktab is ignored
hipError_t smem_build_ktab(const SmemArgs& a, uint4* ktab, hipStream_t s);
// persistent wavefronts that take reads from a queue; compact (32-bit) index only; scratch for n_waves x 64 threads
hipError_t smem_launch_engine(const SmemArgs& a, uint32_t read_base, uint32_t n_reads, uint32_t n_waves, hipStream_t s);

// the -DSMEM_COUNT build of the same kernels (smem_kernel.hip): performed lookups counted into device counters
hipError_t smem_launch_count(const SmemArgs& a, uint32_t read_base, uint32_t n_reads, hipStream_t s);
hipError_t smem_launch_engine_count(const SmemArgs& a, uint32_t read_base, uint32_t n_reads, uint32_t n_waves, hipStream_t s);
// {index sectors (32-byte half-blocks) fetched, prefix-table entries fetched, bwt_extend calls, 0} since the last reset
hipError_t smem_counts_read(uint64_t out[4], bool reset, hipStream_t s);

}  // namespace accg

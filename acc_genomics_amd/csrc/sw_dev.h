// Device/host shared declarations for the HTC Smith-Waterman kernels (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace accg {

constexpr uint32_t SW_NO_PAIR = 0xFFFFFFFFu;

// One wavefront's job: 64/lpp groups of lpp lanes x two packed pairs (lo/hi 16-bit halves).
struct SwWork {
  uint32_t pair[8];     // pair[2*g + half]; SW_NO_PAIR = empty (hi empty: int32 mode or no partner)
};

struct SwArgs {
  const uint8_t* refs; const uint8_t* alts;   // strided matrices
  uint32_t ref_stride, alt_stride;
  const int32_t* ref_len; const int32_t* alt_len;
  const uint8_t* strategy;                    // per pair, htc-sw/host/common.h:15-18
  const SwWork* work;
  int32_t* score; int32_t* p1; int32_t* p2;   // per pair
  int w_match, w_mismatch, w_open, w_extend;
  // backtrace mode only:
  uint4* bt;                 // decision bit planes, one uint4 per (step, lane): see sw_kernel.hip
  uint64_t bt_item_stride;   // uint4 elements per wavefront job
  // Record layout.  0: one uint4 of four bit planes per (step, lane), formed arithmetically (every shape).  1 (16 lanes per pair, packed
  // int16): the decisions as 64-bit LANE MASKS straight out of v_cmp_lt_i16_sdwa -- per (step t, row k) eight masks of 8 bytes,
  // [half][x, y, z, w], bit = lane of the wavefront -- sent to memory by scalar stores: at (t * K + k) * 4 uint4 of the job's block.
  int bt_masks;
  int32_t* cig_n;            // per pair: number of elements, or -(needed) when max_el was too small, or -1 (no alignment)
  int32_t* cig_off;          // per pair: alignment_offset
  int32_t* cig_el;           // per pair: max_el x {length, state} (device-side slots, backtrace order)
  int32_t* cig_packed;       // all CIGARs back to back in alignment order: pair k's elements start at cig_start[k]
  unsigned long long* cig_start;
  unsigned long long* cig_total;   // elements allocated so far (one wave-aggregated atomic per wavefront)
  int32_t max_el;
};

// lane_is_alt: the lanes hold the alternate/read and the sweep runs over the reference window (else the converse).
// pack16: two pairs per group in 16-bit halves (scores must fit int16), else one pair per group in int32.
// with_bt: also record the per-cell decisions needed by the backtrace (a.bt); job i of the launch uses
// a.bt + (i - bt_first) * a.bt_item_stride.
// lpp = lanes per pair group: 16 (lane sequence <= 255), 32 (<= 511) or 64 (<= 1535).
hipError_t sw_launch(int K, int lpp, bool pack16, bool lane_is_alt, bool with_bt, const SwArgs& a, uint32_t work_base, uint32_t n_work,
                     uint32_t bt_first, int sweep_cap, hipStream_t s);
int sw_pick_k(int lane_seq_len, int lpp);
// the backtrace proper (calculateCigarOneBatch, FalconSW_AVX.cpp:2303-2419): one thread per pair
hipError_t sw_trace_launch(int K, int lpp, bool pack16, bool lane_is_alt, const SwArgs& a, uint32_t work_base, uint32_t n_work,
                           uint32_t bt_first, int sweep_cap, hipStream_t s);
size_t sw_lds_bytes(int sweep_cap);
__host__ __device__ inline int sw_group_stride(int sweep_cap) { return ((sweep_cap + 1 + 15) / 32) * 32 + 16; }   // >= cap + 1, = 16 mod 32
inline uint64_t sw_bt_item_uint4(int sweep_cap, int lpp) { return (uint64_t)64 * (sweep_cap + lpp); }
inline uint64_t sw_bt_item_uint4_masks(int sweep_cap, int lpp, int K) { return (uint64_t)4 * K * (sweep_cap + lpp + 1); }

}  // namespace accg

// Device/host shared declarations for the HTC Smith-Waterman kernels (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace accg {

constexpr int SW_MAX_K = 16;             // lane-sequence positions per lane
constexpr int SW_MAX_LANE_SEQ = 16 * SW_MAX_K - 1;   // 255 (one position reserved for the border)
constexpr uint32_t SW_NO_PAIR = 0xFFFFFFFFu;

// One wavefront's job: four groups (DPP rows) x two packed pairs (lo/hi 16-bit halves).
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
};

// lane_is_alt: the lanes hold the alternate/read and the sweep runs over the reference window (else the converse).
// pack16: two pairs per group in 16-bit halves (scores must fit int16), else one pair per group in int32.
hipError_t sw_launch(int K, bool pack16, bool lane_is_alt, const SwArgs& a, uint32_t work_base, uint32_t n_work,
                     int sweep_cap, hipStream_t s);
size_t sw_lds_bytes(int sweep_cap);

}  // namespace accg

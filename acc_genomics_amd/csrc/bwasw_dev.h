// Device/host shared declarations for the BWA-MEM seed-extension kernel (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace accg {

constexpr uint32_t BWASW_EMPTY = 0xFFFFFFFFu;
constexpr int BWASW_MAX_K = 16;          // 16 lanes x K entries cover eh[0..qlen], qlen <= 254

struct alignas(16) BwaswSeed {           // one seed = left extension then right extension (seed_proc, smithwaterman.cpp:586-670)
  uint32_t q_off[2];                     // query codes of each side in the blob
  uint32_t t_off[2];                     // target codes of each side, 4-byte aligned
  uint16_t qlen[2], tlen[2];
  uint16_t seed_len, seed_qbeg;
  uint32_t pad_;
};

struct BwaswWork { uint32_t seed[4]; };  // one wavefront = one side of four seeds, 16 lanes each

struct BwaswArgs {
  const uint8_t* blob;                   // codes 0-3, 4 = N; padded so a 4-byte read at any target offset stays inside
  const BwaswSeed* seeds;
  const BwaswWork* work;
  int16_t* out;                          // n_seeds x 8: qBeg, qEnd, rBeg, rEnd, score, trueScore, width, 0
};

hipError_t bwasw_launch(int K, int side, const BwaswArgs& a, uint32_t n_work, hipStream_t s);

}  // namespace accg

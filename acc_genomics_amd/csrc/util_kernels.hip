// Small device utilities of the context (not part of any hot path).
#include <algorithm>
#include "accg_internal.h"

namespace {
// Every wavefront spins on dependent fp32 fmas for `spin_ticks` of the constant-rate wall clock; wavefront 0 of block 0 notes the
// shader-clock counter (s_memtime) and the wall clock (s_memrealtime) when it starts and when it is done.
__global__ __launch_bounds__(256) void clock_probe(unsigned long long spin_ticks, unsigned long long* out, float* sink) {
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  float a = (float)threadIdx.x, b = 1.0000001f, c = 1e-9f;
  unsigned long long w1 = w0;
  while (w1 - w0 < spin_ticks) {
#pragma unroll
    for (int i = 0; i < 64; i++) a = __builtin_fmaf(a, b, c);
    w1 = wall_clock64();
  }
  const unsigned long long c1 = clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  if (a == 123.456f) *sink = a;      // keeps the fma chain alive
}
// Small transfers as kernels on the stream of the work they belong to: a hipMemcpyAsync between pinned host memory and the device goes
// through a DMA engine and costs the stream a cross-engine dependency each way (tools/ubench_latency.hip: upload of 80 KB + two kernels +
// 24 KB back + wait = 29.7 us with copies, 20.9 us with copy kernels) -- for a region of a few thousand pairs that is a tenth of the call.
__global__ __launch_bounds__(256) void k_upload16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, unsigned long long* tick) {
  if (tick && blockIdx.x == 0 && threadIdx.x == 0) *tick = wall_clock64();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// results of a PairHMM pass into pinned host memory: [ticks since *tick u64][n_rescued u64][raw f32 x n][pad to 8][fp64 x n, only when
// something was rescued]; res = the device's [n_rescued u64][raw f32 x n] block
__global__ __launch_bounds__(256) void k_phmm_results(const uint32_t* __restrict__ res, const uint32_t* __restrict__ out64, size_t n, uint32_t* __restrict__ stage,
                                                      size_t off64_words, const unsigned long long* tick) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned long long t = tick ? wall_clock64() - *tick : 0ull;
    stage[0] = (uint32_t)t; stage[1] = (uint32_t)(t >> 32);
  }
  const size_t head = 2 + n, stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < head; i += stride) stage[2 + i] = res[i];
  if ((res[0] | res[1]) != 0u)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 2 * n; i += stride) stage[off64_words + i] = out64[i];
}
// the same for a batch whose fp64 values were computed speculatively for every pair (no planner, nothing counted on the way): ONE block
// counts the pairs below MIN_ACCEPTED (1e-28f, host_type.h:21) while it copies the fp32 results, and sends the fp64 values if there are any
__global__ __launch_bounds__(256) void k_phmm_results_spec(const float* __restrict__ raw, const uint32_t* __restrict__ out64, size_t n, uint32_t* __restrict__ stage,
                                                            size_t off64_words, const unsigned long long* tick) {
  __shared__ unsigned s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  unsigned mine = 0;
  for (size_t i = threadIdx.x; i < n; i += 256) { const float v = raw[i]; stage[4 + i] = __float_as_uint(v); mine += v < 1e-28f; }
  if (mine) atomicAdd(&s_cnt, mine);
  __syncthreads();
  const unsigned cnt = s_cnt;
  if (cnt) for (size_t i = threadIdx.x; i < 2 * n; i += 256) stage[off64_words + i] = out64[i];
  if (threadIdx.x == 0) {
    const unsigned long long t = tick ? wall_clock64() - *tick : 0ull;
    stage[0] = (uint32_t)t; stage[1] = (uint32_t)(t >> 32); stage[2] = cnt; stage[3] = 0u;
  }
}
}  // namespace

namespace accg {
hipError_t phmm_results_spec_by_kernel(const float* raw, const double* out64, size_t n, void* stage_pinned, size_t off64_bytes, const unsigned long long* tick, hipStream_t s) {
  hipLaunchKernelGGL(k_phmm_results_spec, dim3(1), dim3(256), 0, s, raw, (const uint32_t*)out64, n, (uint32_t*)stage_pinned, off64_bytes / 4, tick);
  return hipGetLastError();
}
hipError_t upload_by_kernel(const void* host_pinned, void* dev, size_t bytes, unsigned long long* tick, hipStream_t s) {
  const size_t n16 = (bytes + 15) / 16;          // (both blocks are padded to 16 bytes by their owners)
  const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>((n16 + 255) / 256, 1), 256);
  hipLaunchKernelGGL(k_upload16, dim3(grid), dim3(256), 0, s, (const uint4*)host_pinned, (uint4*)dev, n16, tick);
  return hipGetLastError();
}
hipError_t phmm_results_by_kernel(const void* res, const double* out64, size_t n, void* stage_pinned, size_t off64_bytes, const unsigned long long* tick, hipStream_t s) {
  const unsigned grid = (unsigned)std::min<size_t>(std::max<size_t>((2 * n + 255) / 256, 1), 128);
  hipLaunchKernelGGL(k_phmm_results, dim3(grid), dim3(256), 0, s, (const uint32_t*)res, (const uint32_t*)out64, n, (uint32_t*)stage_pinned, off64_bytes / 4, tick);
  return hipGetLastError();
}
}  // namespace accg

extern "C" int accg_ctx_clock_ghz(accg_ctx* ctx, float* ghz) {
  if (!ctx) return ACCG_ERR_NOT_INITIALISED;
  if (!ghz) return ACCG_ERR_BAD_ARG;
  ACCG_HIP(hipSetDevice(ctx->device));
  int wall_khz = 0;
  ACCG_HIP(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, ctx->device));
  if (wall_khz <= 0) wall_khz = 100000;                       // 100 MHz on every gfx9
  void* d = nullptr;
  ACCG_HIP(ctx->pool.get(64, &d));
  struct Put { accg::DevPool& p; void* d; ~Put() { p.put(d); } } put{ctx->pool, d};
  unsigned long long* out = (unsigned long long*)d;
  const unsigned long long ticks = (unsigned long long)wall_khz * 3 / 10;      // 0.3 ms
  hipLaunchKernelGGL(clock_probe, dim3(ctx->n_cu * 8), dim3(256), 0, ctx->stream, ticks, out, (float*)(out + 4));
  ACCG_HIP(hipGetLastError());
  unsigned long long h[2] = {0, 0};
  ACCG_HIP(hipMemcpyAsync(h, out, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
  ACCG_HIP(hipStreamSynchronize(ctx->stream));
  if (h[1] == 0) return ACCG_ERR_HIP;
  *ghz = (float)((double)h[0] / (double)h[1] * (double)wall_khz * 1e-6);
  return ACCG_OK;
}

// Small device utilities of the context (not part of any hot path).
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
}  // namespace

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

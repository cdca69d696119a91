// Random 32-byte-sector read rate out of a 64 MB table (the SMEM index size of configs[4]): every thread runs a chain of
// dependent lookups, two independent 32-byte reads per step (what bwt_extend does over the half-block index), all 64 lanes
// of a wavefront active.  16384 wavefronts x 170 steps (the shape of the SMEM forward kernel on configs[4]); the resident
// wavefronts per CU are held down with an LDS request.  Prints sectors per second for several occupancies, for two sectors
// per step and for one.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_random.hip -o tools/ubench_random
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int TWO>
__global__ __launch_bounds__(64) void chase(const uint4* __restrict__ tab, uint32_t mask, int steps, uint32_t* out) {
  extern __shared__ uint32_t pad[];
  uint32_t a = blockIdx.x * 64 + threadIdx.x, b = a * 2654435761u + 12345u;
  uint32_t acc = 0;
  for (int s = 0; s < steps; s++) {
    const uint4* pa = tab + 2 * (size_t)(a & mask);
    const uint4* pb = tab + 2 * (size_t)(b & mask);
    const uint4 a0 = pa[0], a1 = pa[1];
    uint4 b0 = a0, b1 = a1;
    if (TWO) { b0 = pb[0]; b1 = pb[1]; }
    acc += a0.x ^ a1.y ^ b0.z ^ b1.w;
    a = a * 1664525u + 1013904223u + a0.x;     // the next addresses depend on what was read
    b = b * 22695477u + 1u + b1.w;
  }
  if (acc == 0x12345678u) pad[threadIdx.x] = acc;
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main() {
  const size_t bytes = 64u << 20, n32 = bytes / 32;
  uint4* tab; uint32_t* out;
  hipMalloc(&tab, bytes);
  std::vector<uint32_t> h(bytes / 4);
  for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)rand();
  hipMemcpy(tab, h.data(), bytes, hipMemcpyHostToDevice);
  const int w = 16384, steps = 170;
  hipMalloc(&out, w * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int per_cu[] = {4, 8, 12, 16, 20, 24, 28, 32};
  for (int two = 1; two >= 0; two--)
    for (int wpc : per_cu) {
      const size_t lds = (160 * 1024 / wpc) & ~255u;
      auto k = two ? chase<1> : chase<0>;
      hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(k, dim3(w), dim3(64), lds, 0, tab, (uint32_t)(n32 - 1), steps, out);
      hipEventRecord(e0);
      for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(w), dim3(64), lds, 0, tab, (uint32_t)(n32 - 1), steps, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
      const double sectors = (double)w * 64 * steps * (two ? 2 : 1);
      printf("%d sector(s)/step, %2d waves per CU: %.3f ms, %.1f G sectors/s, %.2f TB/s\n", two ? 2 : 1, wpc, ms, sectors / ms / 1e6, sectors * 32 / ms / 1e9);
    }
  return 0;
}

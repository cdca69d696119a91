// Random 32-byte-sector read rate out of a 64 MB table (the SMEM index size of configs[4]): every thread runs a chain of
// dependent lookups, two independent 32-byte reads per step (what bwt_extend does over the half-block index), all 64 lanes
// of a wavefront active.  16384 wavefronts x 170 steps (the shape of the SMEM forward kernel on configs[4]); the resident
// wavefronts per CU are held down with an LDS request.  Prints sectors per second for several occupancies, for two sectors
// per step and for one.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_random.hip -o tools/ubench_random
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

template <int TWO>
__global__ __launch_bounds__(64) void chase(const uint4* __restrict__ tab, uint64_t mask, int steps, uint32_t* out, int active, uint4* wbuf, uint64_t wmask, int rstream) {
  extern __shared__ uint32_t pad[];
  uint32_t a = blockIdx.x * 64 + threadIdx.x, b = a * 2654435761u + 12345u;
  uint32_t acc = 0;
  if ((int)threadIdx.x >= active) steps = 0;        // lanes beyond `active` idle: the shape of a divergent wavefront's loads
  for (int s = 0; s < steps; s++) {
    // (a table of more than 2^32 sectors would need wider state; up to 128 GB the 32-bit state times a stride covers it)
    const uint4* pa = tab + 2 * (size_t)(((uint64_t)a * 2654435761ull) & mask);
    const uint4* pb = tab + 2 * (size_t)(((uint64_t)b * 2246822519ull) & mask);
    const uint4 a0 = pa[0], a1 = pa[1];
    uint4 b0 = a0, b1 = a1;
    if (TWO) { b0 = pb[0]; b1 = pb[1]; }
    acc += a0.x ^ a1.y ^ b0.z ^ b1.w;
    // a record of 16 bytes per lane and step into a region that wraps (the SMEM kernel's list pushes: what does a write stream do to the reads' cache?)
    if (wbuf) {
      const uint64_t wi = (((uint64_t)blockIdx.x * steps + s) * 64 + threadIdx.x) & wmask;
      if (rstream) { const uint4 r = wbuf[(wi + (wmask + 1) / 2) & wmask]; acc += r.x; }      // ... and a record read from the other half of the region (list reads)
      wbuf[wi] = make_uint4(acc, a, b, s);
    }
    a = a * 1664525u + 1013904223u + a0.x;     // the next addresses depend on what was read
    b = b * 22695477u + 1u + b1.w;
  }
  if (acc == 0x12345678u) pad[threadIdx.x] = acc;
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  // usage: ubench_random [table MB = 64] [lanes active per wavefront = 64]   (a table far beyond the 256 MB Infinity Cache shows what HBM
  // itself serves at random; the SMEM kernel's loads have ~11 active lanes)
  const size_t mb = argc > 1 ? strtoull(argv[1], nullptr, 10) : 64;
  const int active = argc > 2 ? atoi(argv[2]) : 64;
  const size_t wmb = argc > 3 ? strtoull(argv[3], nullptr, 10) : 0;       // power of two
  const int rstream = argc > 4 ? atoi(argv[4]) : 0;
  uint4* wbuf = nullptr; uint64_t wmask = 0;
  if (wmb) { if (hipMalloc(&wbuf, wmb << 20) != hipSuccess) { printf("hipMalloc of the write region failed\n"); return 1; } wmask = ((uint64_t)wmb << 20) / 16 - 1; }
  const size_t bytes = mb << 20, n32 = bytes / 32;
  uint4* tab; uint32_t* out;
  if (hipMalloc(&tab, bytes) != hipSuccess) { printf("hipMalloc of %zu MB failed\n", mb); return 1; }
  {
    std::vector<uint32_t> h((64u << 20) / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)rand();
    for (size_t o = 0; o < bytes; o += 64u << 20) hipMemcpy((char*)tab + o, h.data(), std::min<size_t>(64u << 20, bytes - o), hipMemcpyHostToDevice);
  }
  const int w = 16384, steps = 170;
  hipMalloc(&out, w * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("table of %zu MB, %d of 64 lanes active, write region %zu MB%s\n", mb, active, wmb, rstream ? " (and a record read from it per step)" : "");
  const int per_cu[] = {4, 8, 20, 32};
  for (int two = 1; two >= 0; two--)
    for (int wpc : per_cu) {
      const size_t lds = (160 * 1024 / wpc) & ~255u;
      auto k = two ? chase<1> : chase<0>;
      hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(k, dim3(w), dim3(64), lds, 0, tab, (uint64_t)(n32 - 1), steps, out, active, wbuf, wmask, rstream);
      hipEventRecord(e0);
      for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(w), dim3(64), lds, 0, tab, (uint64_t)(n32 - 1), steps, out, active, wbuf, wmask, rstream);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
      const double sectors = (double)w * active * steps * (two ? 2 : 1);
      printf("%d sector(s)/step, %2d waves per CU: %.3f ms, %.1f G sectors/s, %.2f TB/s of 32-byte sectors (%.2f TB/s if every one is a 64-byte fetch)\n", two ? 2 : 1, wpc, ms,
             sectors / ms / 1e6, sectors * 32 / ms / 1e9, sectors * 64 / ms / 1e9);
    }
  return 0;
}

// Host-side latency of the small-batch submission patterns a blocking region call can be built from (MI355X, one stream):
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_latency.hip -o tools/ubench_latency && ./tools/ubench_latency
// Each pattern = upload of 80 KB, two small kernels, results of 24 KB back, wait; median of 200 repetitions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void work(const uint32_t* in, uint32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * 3u + 1u;
}
__global__ void copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// results to host memory, then a flag the host polls (system-scope release behind the data)
__global__ void copy_flag(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, unsigned* done_count, volatile uint32_t* flag, uint32_t seq) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned k = atomicAdd(done_count, 1u);
    if (k == gridDim.x - 1) { *done_count = 0; __threadfence_system(); *flag = seq; }
  }
}

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

int main() {
  const size_t UP = 80 << 10, DOWN = 24 << 10;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint8_t *h_up, *h_down; uint32_t* h_flag;
  CK(hipHostMalloc(&h_up, UP, hipHostMallocDefault)); CK(hipHostMalloc(&h_down, DOWN, hipHostMallocDefault));
  CK(hipHostMalloc(&h_flag, 64, hipHostMallocCoherent));
  *h_flag = 0;
  uint8_t *d_in, *d_out; unsigned* d_cnt;
  CK(hipMalloc(&d_in, UP)); CK(hipMalloc(&d_out, UP)); CK(hipMalloc(&d_cnt, 64)); CK(hipMemset(d_cnt, 0, 64));
  memset(h_up, 1, UP);
  const int n = (int)(UP / 4);
  auto k2 = [&]() {
    hipLaunchKernelGGL(work, dim3((n + 255) / 256), dim3(256), 0, s, (const uint32_t*)d_in, (uint32_t*)d_out, n);
    hipLaunchKernelGGL(work, dim3((n + 255) / 256), dim3(256), 0, s, (const uint32_t*)d_out, (uint32_t*)d_in, n);
  };
  struct Pat { const char* name; int up, down; };
  // up: 0 hipMemcpyAsync, 1 copy kernel from pinned; down: 0 sync + hipMemcpyAsync + sync, 1 hipMemcpyAsync queued + one sync, 2 copy kernel to pinned + one sync,
  //   3 copy kernel + flag polled by the host
  const Pat pats[] = {{"memcpyAsync up | sync, memcpyAsync down, sync", 0, 0}, {"memcpyAsync up | memcpyAsync down queued, one sync", 0, 1},
                      {"copy kernel up | memcpyAsync down queued, one sync", 1, 1}, {"copy kernel up | copy kernel down, one sync", 1, 2},
                      {"copy kernel up | copy kernel down + host-polled flag", 1, 3}, {"memcpyAsync up | copy kernel down + host-polled flag", 0, 3}};
  uint32_t seq = 0;
  for (const Pat& p : pats) {
    std::vector<double> tot, t_up, t_k, t_dn;
    for (int rep = 0; rep < 220; rep++) {
      const auto t0 = clk::now();
      if (p.up == 0) CK(hipMemcpyAsync(d_in, h_up, UP, hipMemcpyHostToDevice, s));
      else hipLaunchKernelGGL(copy16, dim3(16), dim3(256), 0, s, (const uint4*)h_up, (uint4*)d_in, UP / 16);
      const auto t1 = clk::now();
      k2();
      const auto t2 = clk::now();
      if (p.down == 0) { CK(hipStreamSynchronize(s)); CK(hipMemcpyAsync(h_down, d_in, DOWN, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
      else if (p.down == 1) { CK(hipMemcpyAsync(h_down, d_in, DOWN, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
      else if (p.down == 2) { hipLaunchKernelGGL(copy16, dim3(8), dim3(256), 0, s, (const uint4*)d_in, (uint4*)h_down, DOWN / 16); CK(hipStreamSynchronize(s)); }
      else {
        ++seq;
        hipLaunchKernelGGL(copy_flag, dim3(8), dim3(256), 0, s, (const uint4*)d_in, (uint4*)h_down, DOWN / 16, d_cnt, (volatile uint32_t*)h_flag, seq);
        while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq) {}
      }
      const auto t3 = clk::now();
      if (rep >= 20) { tot.push_back(us(t0, t3)); t_up.push_back(us(t0, t1)); t_k.push_back(us(t1, t2)); t_dn.push_back(us(t2, t3)); }
    }
    CK(hipStreamSynchronize(s));
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("%-58s total %6.1f us  (queue upload %5.1f, two launches %5.1f, download + wait %5.1f)\n", p.name, med(tot), med(t_up), med(t_k), med(t_dn));
  }
  // plain launch + sync latency, for scale
  {
    std::vector<double> v;
    for (int rep = 0; rep < 220; rep++) {
      const auto t0 = clk::now();
      hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, (const uint32_t*)d_in, (uint32_t*)d_out, 64);
      CK(hipStreamSynchronize(s));
      if (rep >= 20) v.push_back(us(t0, clk::now()));
    }
    std::sort(v.begin(), v.end());
    printf("one empty launch + hipStreamSynchronize: %.1f us\n", v[v.size() / 2]);
  }
  return 0;
}

// VALU issue-rate microbenchmark for gfx950 (design input for the PairHMM / SW kernels).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o gpurun_out/ubench ; run on the GPU box.
// Prints wave-instructions per SIMD-cycle-equivalent (normalised by the measured v_fma rate is up to the reader).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float c = 0.999f, d = 0.001f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pc = {c, c}, pd = {d, d};
  int cls = 0x60;
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {        // 8 independent v_fma_f32, x2
      asm volatile(
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    } else if (MODE == 1) { // 4 independent v_pk_fma_f32, x4 (16 instr = 32 lane-fmas)
      asm volatile(
          "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
          "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
          "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
          "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
          : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc), "v"(pd));
    } else if (MODE == 2) { // v_mul_f32 with DPP row_shr:1 operand
      asm volatile(
          "v_mul_f32_dpp %0, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %1, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mul_f32_dpp %2, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %3, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mul_f32_dpp %4, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %5, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mul_f32_dpp %6, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %7, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mul_f32_dpp %0, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %1, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mul_f32_dpp %2, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %3, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mul_f32_dpp %4, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %5, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mul_f32_dpp %6, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32_dpp %7, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    } else if (MODE == 3) { // v_cmp_class_f16 + v_cndmask pairs (8 pairs = 16 instr)
      asm volatile(
          "v_cmp_class_f16 vcc, %0, %10\n v_cndmask_b32 %0, %8, %9, vcc\n v_cmp_class_f16 vcc, %1, %10\n v_cndmask_b32 %1, %8, %9, vcc\n"
          "v_cmp_class_f16 vcc, %2, %10\n v_cndmask_b32 %2, %8, %9, vcc\n v_cmp_class_f16 vcc, %3, %10\n v_cndmask_b32 %3, %8, %9, vcc\n"
          "v_cmp_class_f16 vcc, %4, %10\n v_cndmask_b32 %4, %8, %9, vcc\n v_cmp_class_f16 vcc, %5, %10\n v_cndmask_b32 %5, %8, %9, vcc\n"
          "v_cmp_class_f16 vcc, %6, %10\n v_cndmask_b32 %6, %8, %9, vcc\n v_cmp_class_f16 vcc, %7, %10\n v_cndmask_b32 %7, %8, %9, vcc\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d), "v"(cls) : "vcc");
    } else if (MODE == 4) { // v_mul_f32 (VOP2), 16 independent-ish
      asm volatile(
          "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
          "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
          "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
          "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    } else if (MODE == 5) { // v_pk_add_i16 with clamp + v_pk_max_i16 (SW kernel mix)
      asm volatile(
          "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_max_i16 %1, %1, %0\n v_pk_add_i16 %2, %2, %8 clamp\n v_pk_max_i16 %3, %3, %2\n"
          "v_pk_add_i16 %4, %4, %8 clamp\n v_pk_max_i16 %5, %5, %4\n v_pk_add_i16 %6, %6, %8 clamp\n v_pk_max_i16 %7, %7, %6\n"
          "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_max_i16 %1, %1, %0\n v_pk_add_i16 %2, %2, %8 clamp\n v_pk_max_i16 %3, %3, %2\n"
          "v_pk_add_i16 %4, %4, %8 clamp\n v_pk_max_i16 %5, %5, %4\n v_pk_add_i16 %6, %6, %8 clamp\n v_pk_max_i16 %7, %7, %6\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    } else if (MODE == 6) { // v_add_u32 + v_max_i32 + v_max3_i32 (int32 SW mix)
      asm volatile(
          "v_add_u32 %0, %0, %8\n v_max_i32 %1, %1, %0\n v_max3_i32 %2, %2, %0, %1\n v_add_u32 %3, %3, %8\n"
          "v_max_i32 %4, %4, %3\n v_max3_i32 %5, %5, %3, %4\n v_add_u32 %6, %6, %8\n v_max_i32 %7, %7, %6\n"
          "v_add_u32 %0, %0, %8\n v_max_i32 %1, %1, %0\n v_max3_i32 %2, %2, %0, %1\n v_add_u32 %3, %3, %8\n"
          "v_max_i32 %4, %4, %3\n v_max3_i32 %5, %5, %3, %4\n v_add_u32 %6, %6, %8\n v_max_i32 %7, %7, %6\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    } else if (MODE == 7) { // v_fma_f64, 4 independent x4
      asm volatile("; f64 handled below" ::);
    }
  }
  if (MODE == 1) { a0 = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y; a1 = a2 = a3 = a4 = a5 = a6 = a7 = 0; }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ __launch_bounds__(64) void k64(double* out, int iters) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, c = 0.999, d = 0.001;
  for (int i = 0; i < iters; i++) {
    asm volatile(
        "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
        "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
        "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
        "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(d));
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}

int main() {
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f32_dpp(row_shr:1)", "v_cmp_class_f16+v_cndmask", "v_mul_f32", "v_pk_add_i16 clamp+v_pk_max_i16", "v_add_u32/v_max_i32/v_max3_i32"};
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount;
  printf("device %s CUs %d clock %d kHz\n", p.name, cus, p.clockRate);
  float* out; hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8 * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int w = 1; w <= 8; w *= 2) {
    int blocks = cus * 4 * w;
    for (int m = 0; m < 8; m++) {
      float best = 1e9;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        switch (m) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, iters); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, iters); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, iters); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, iters); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, iters); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, out, iters); break;
          case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(64), 0, 0, out, iters); break;
          case 7: hipLaunchKernelGGL(k64, dim3(blocks), dim3(64), 0, 0, (double*)out, iters); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      // wave-instructions issued per SIMD = w * iters * 16 ; time per wave-instr per SIMD in ns:
      double ns_per_instr = (double)best * 1e6 / ((double)w * iters * 16);
      printf("waves/SIMD %d  %-34s %8.3f ms  %6.3f ns per wave-instr per SIMD (=%5.2f cyc @2.4GHz)\n", w,
             m < 7 ? names[m] : "v_fma_f64", best, ns_per_instr, ns_per_instr * 2.4);
    }
  }
  return 0;
}

// Can decision bits leave the SIMD without VALU work?  v_cmp_*_i16_sdwa writes a 64-bit lane mask to an SGPR pair; s_store_dwordx4 sends
// SGPRs to memory on the scalar-memory pipe.  This measures, per "step" of 10 rows (the shape of the HTC-SW record at configs[2]):
//   A: 120 packed-int16 ops (the fill's ballast)
//   B: A + 80 v_cmp_lt_i16_sdwa (8 per row)
//   C: B + 40 s_store_dwordx4 (640 bytes of masks per step and wavefront)
//   D: A + the arithmetic extraction used today (per row 4 x {v_pk_sub_i16 clamp, v_lshrrev_b32, v_and_or_b32}) + one global_store_dwordx4
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_sstore.hip -o tools/ubench_sstore && ./tools/ubench_sstore
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint4* rec, int steps) {
  typedef short s2 __attribute__((ext_vector_type(2)));
  int a0 = threadIdx.x * 3 + 1, a1 = threadIdx.x * 5 + 2, a2 = threadIdx.x * 7 + 3, a3 = threadIdx.x * 11 + 4;
  int b0 = 0x00010001, b1 = 0x00020002;
  unsigned p0 = 0, p1 = 0, p2 = 0, p3 = 0;
  uint64_t base = (uint64_t)(rec + (size_t)blockIdx.x * 64 * 48);     // a wave's own 48 KB window, reused (the question is the pipe, not HBM)
  uint4* vrec = rec + (size_t)blockIdx.x * 64 * 48 + threadIdx.x;
  for (int t = 0; t < steps; t++) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
      // ballast: 12 packed ops with a dependency pattern like the fill's
      asm volatile(
          "v_pk_add_i16 %0, %0, %4 clamp\n\tv_pk_max_i16 %1, %1, %0\n\tv_pk_add_i16 %2, %2, %5 clamp\n\tv_pk_max_i16 %3, %3, %2\n\t"
          "v_pk_add_i16 %0, %0, %5 clamp\n\tv_pk_max_i16 %1, %1, %3\n\tv_pk_add_i16 %2, %2, %4 clamp\n\tv_pk_max_i16 %3, %3, %1\n\t"
          "v_pk_add_i16 %0, %0, %4 clamp\n\tv_pk_max_i16 %1, %1, %2\n\tv_pk_add_i16 %2, %2, %5 clamp\n\tv_pk_max_i16 %3, %3, %0\n\t"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
      if (MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the previous row's stores have read their SGPRs
      if (MODE == 1 || MODE == 2 || MODE == 4) {
        // eight compares -> s[40:55]; two rows make one batch of 4 x dwordx4 stores... here: stored per row as 4 x dwordx4 (64 B)
        asm volatile(
            "v_cmp_lt_i16_sdwa s[40:41], %0, %1 src0_sel:WORD_0 src1_sel:WORD_0\n\tv_cmp_lt_i16_sdwa s[42:43], %0, %1 src0_sel:WORD_1 src1_sel:WORD_1\n\t"
            "v_cmp_lt_i16_sdwa s[44:45], %1, %2 src0_sel:WORD_0 src1_sel:WORD_0\n\tv_cmp_lt_i16_sdwa s[46:47], %1, %2 src0_sel:WORD_1 src1_sel:WORD_1\n\t"
            "v_cmp_lt_i16_sdwa s[48:49], %2, %3 src0_sel:WORD_0 src1_sel:WORD_0\n\tv_cmp_lt_i16_sdwa s[50:51], %2, %3 src0_sel:WORD_1 src1_sel:WORD_1\n\t"
            "v_cmp_lt_i16_sdwa s[52:53], %3, %0 src0_sel:WORD_0 src1_sel:WORD_0\n\tv_cmp_lt_i16_sdwa s[54:55], %3, %0 src0_sel:WORD_1 src1_sel:WORD_1\n\t"
            : : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
            : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
        if (MODE == 2 || MODE == 4) {
          const uint64_t addr = base + (uint64_t)((t % 48) * 640 + r * 64);
          asm volatile(
              "s_store_dwordx4 s[40:43], %0, 0x0\n\ts_store_dwordx4 s[44:47], %0, 0x10\n\ts_store_dwordx4 s[48:51], %0, 0x20\n\ts_store_dwordx4 s[52:55], %0, 0x30\n\t"
              : : "s"(addr) : "memory");
        } else {
          unsigned x;
          asm volatile("s_xor_b32 %0, s40, s55" : "=s"(x));
          p0 ^= x;
        }
      }
      if (MODE == 5 || MODE == 6) {      // what a compare costs by encoding: 8 x VOP3 v_cmp_lt_i16 (low halves only) / 8 x v_cmp_lt_i32, no stores
        if (MODE == 5)
          asm volatile(
              "v_cmp_lt_i16_e64 s[40:41], %0, %1\n\tv_cmp_lt_i16_e64 s[42:43], %1, %2\n\tv_cmp_lt_i16_e64 s[44:45], %2, %3\n\tv_cmp_lt_i16_e64 s[46:47], %3, %0\n\t"
              "v_cmp_lt_i16_e64 s[48:49], %0, %2\n\tv_cmp_lt_i16_e64 s[50:51], %1, %3\n\tv_cmp_lt_i16_e64 s[52:53], %2, %0\n\tv_cmp_lt_i16_e64 s[54:55], %3, %1\n\t"
              : : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
              : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
        else
          asm volatile(
              "v_cmp_lt_i32_e64 s[40:41], %0, %1\n\tv_cmp_lt_i32_e64 s[42:43], %1, %2\n\tv_cmp_lt_i32_e64 s[44:45], %2, %3\n\tv_cmp_lt_i32_e64 s[46:47], %3, %0\n\t"
              "v_cmp_lt_i32_e64 s[48:49], %0, %2\n\tv_cmp_lt_i32_e64 s[50:51], %1, %3\n\tv_cmp_lt_i32_e64 s[52:53], %2, %0\n\tv_cmp_lt_i32_e64 s[54:55], %3, %1\n\t"
              : : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
              : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
        unsigned x;
        asm volatile("s_xor_b32 %0, s40, s55" : "=s"(x));
        p0 ^= x;
      }
      if (MODE == 3) {
        s2 d0 = __builtin_elementwise_sub_sat(__builtin_bit_cast(s2, a0), __builtin_bit_cast(s2, a1));
        s2 d1 = __builtin_elementwise_sub_sat(__builtin_bit_cast(s2, a1), __builtin_bit_cast(s2, a2));
        s2 d2 = __builtin_elementwise_sub_sat(__builtin_bit_cast(s2, a2), __builtin_bit_cast(s2, a3));
        s2 d3 = __builtin_elementwise_sub_sat(__builtin_bit_cast(s2, a3), __builtin_bit_cast(s2, a0));
        p0 = (p0 >> 1) | (__builtin_bit_cast(unsigned, d0) & 0x80008000u);
        p1 = (p1 >> 1) | (__builtin_bit_cast(unsigned, d1) & 0x80008000u);
        p2 = (p2 >> 1) | (__builtin_bit_cast(unsigned, d2) & 0x80008000u);
        p3 = (p3 >> 1) | (__builtin_bit_cast(unsigned, d3) & 0x80008000u);
      }
    }
    if (MODE == 3) vrec[(t % 48) * 64] = make_uint4(p0, p1, p2, p3);
  }
  if (MODE == 2 || MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
  out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ p0 ^ p1 ^ p2 ^ p3;
}

template <int MODE>
static int run(const char* name, uint32_t* out, uint4* rec, int steps, int waves) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, out, rec, steps);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, out, rec, steps);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-70s %7.3f ms  = %6.1f ns per step and wavefront-slot (x 4 per SIMD)\n", name, ms, ms * 1e6 / steps / (waves / (1024.0 * 4)) / 4);
  return 0;
}

int main() {
  const int waves = 1024 * 4 * 2, steps = 2000;     // four wavefronts per SIMD, two rounds
  uint32_t* out; uint4* rec;
  CK(hipMalloc(&out, (size_t)waves * 64 * 4)); CK(hipMalloc(&rec, (size_t)waves * 64 * 48 * 16));
  if (run<0>("A: 120 packed ops per step", out, rec, steps, waves)) return 1;
  if (run<1>("B: A + 80 v_cmp_lt_i16_sdwa -> SGPR masks", out, rec, steps, waves)) return 1;
  if (run<2>("C: B + 40 s_store_dwordx4 (640 B per step)", out, rec, steps, waves)) return 1;
  if (run<3>("D: A + arithmetic extraction (40 x 3 ops) + global_store_dwordx4", out, rec, steps, waves)) return 1;
  if (run<5>("F: A + 80 v_cmp_lt_i16_e64 (VOP3, low halves) -> SGPR masks", out, rec, steps, waves)) return 1;
  if (run<6>("G: A + 80 v_cmp_lt_i32_e64 -> SGPR masks", out, rec, steps, waves)) return 1;
  if (run<4>("E: C with s_waitcnt lgkmcnt(0) in front of every row's compares", out, rec, steps, waves)) return 1;
  // what E left in memory against what C left (same inputs, same addresses): a difference = C overwrote SGPRs a store had not read yet
  {
    const size_t words = (size_t)48 * 640 / 4;      // the first wavefront's window
    uint32_t *hc = new uint32_t[words], *he = new uint32_t[words];
    hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, out, rec, 96);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(hc, rec, words * 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k<4>, dim3(waves), dim3(64), 0, 0, out, rec, 96);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(he, rec, words * 4, hipMemcpyDeviceToHost));
    size_t diff = 0, nz = 0;
    for (size_t i = 0; i < words; i++) { diff += hc[i] != he[i]; nz += he[i] != 0; }
    printf("record words differing between C (no wait) and E (wait): %zu of %zu (%zu non-zero)\n", diff, words, nz);
  }
  return 0;
}

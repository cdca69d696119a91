// Issue cost of two PairHMM rows of the five-operation sweep, singles against packed (v_pk_fma_f32 for the three row-local
// operations), with the sweep's own dependencies, at 1..8 wavefronts per SIMD.  Explicit registers.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench3.hip -o tools/ubench3
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"
// registers: M: v20,v21 (rows a,b)  Xs: v22,v23  Ys: v24,v25  a: v26,v27  b: v28,v29  yy: v30,v31  cx: v32,v33  d: v34,v35  t: v36,v37  Xp: v38 Mp: v39 tc: v40
// second pair of rows: +20 (v40.. reused carefully): M v42,v43 Xs v44,v45 Ys v46,v47 t v48,v49 ; coefficients shared with the first pair
#define SINGLES(M0,M1,X0,X1,Y0,Y1,T0,T1,XP,MP,TC) \
  "v_fma_f32 " T0 ", " X0 ", v26, " M0 "\n v_fma_f32 " X0 ", " XP ", v32, " MP "\n v_fmac_f32 " T0 ", " Y0 ", v28\n v_fma_f32 " Y0 ", " Y0 ", v30, " M0 "\n v_mul_f32 " M0 ", v34, " TC "\n" \
  "v_fma_f32 " T1 ", " X1 ", v27, " M1 "\n v_fma_f32 " X1 ", " X0 ", v33, " M0 "\n v_fmac_f32 " T1 ", " Y1 ", v29\n v_fma_f32 " Y1 ", " Y1 ", v31, " M1 "\n v_mul_f32 " M1 ", v35, " T0 "\n"
#define PACKED(MP_,M0,M1,XP_,X0,X1,YP_,TP_,T0,XPREV,MPREV,TC) \
  "v_pk_fma_f32 " TP_ ", " XP_ ", v[26:27], " MP_ "\n v_fma_f32 " X0 ", " XPREV ", v32, " MPREV "\n v_pk_fma_f32 " TP_ ", " YP_ ", v[28:29], " TP_ "\n v_pk_fma_f32 " YP_ ", " YP_ ", v[30:31], " MP_ "\n" \
  "v_mul_f32 " M0 ", v34, " TC "\n v_fma_f32 " X1 ", " X0 ", v33, " M0 "\n v_mul_f32 " M1 ", v35, " T0 "\n"
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
  asm volatile("v_mov_b32 v20, 0.5\n v_mov_b32 v21, 0.5\n v_mov_b32 v22, 0.5\n v_mov_b32 v23, 0.5\n v_mov_b32 v24, 0.5\n v_mov_b32 v25, 0.5\n v_mov_b32 v26, 0.5\n v_mov_b32 v27, 0.5\n"
               "v_mov_b32 v28, 0.5\n v_mov_b32 v29, 0.5\n v_mov_b32 v30, 0.5\n v_mov_b32 v31, 0.5\n v_mov_b32 v32, 0.5\n v_mov_b32 v33, 0.5\n v_mov_b32 v34, 0.5\n v_mov_b32 v35, 0.5\n"
               "v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0.5\n v_mov_b32 v43, 0.5\n"
               "v_mov_b32 v44, 0.5\n v_mov_b32 v45, 0.5\n v_mov_b32 v46, 0.5\n v_mov_b32 v47, 0.5\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n" ::: CLOB);
  for (int i = 0; i < iters; i++) {
    if (MODE == 0)        // 4 rows as singles: 20 instructions
      asm volatile(SINGLES("v20","v21","v22","v23","v24","v25","v36","v37","v38","v39","v40") SINGLES("v42","v43","v44","v45","v46","v47","v48","v49","v23","v21","v37") ::: CLOB);
    else                  // 4 rows packed: 14 instructions
      asm volatile(PACKED("v[20:21]","v20","v21","v[22:23]","v22","v23","v[24:25]","v[36:37]","v36","v38","v39","v40")
                   PACKED("v[42:43]","v42","v43","v[44:45]","v44","v45","v[46:47]","v[48:49]","v48","v23","v21","v37") ::: CLOB);
  }
  float r;
  asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v42\n v_add_f32 %0, %0, v49" : "=v"(r)::CLOB);
  out[blockIdx.x * 64 + threadIdx.x] = r;
}
template <int M> void run(const char* name, float* out, int cus, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-40s", name);
  for (int w : {1, 2, 3, 4, 8}) {
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<M>, dim3(cus * 4 * w), dim3(64), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("  w%d %7.2f", w, (double)best * 1e6 / ((double)w * iters * 4));     // ns of SIMD time per row
  }
  printf("   ns of SIMD time per row\n");
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  float* out; hipMalloc(&out, sizeof(float) * 64 * p.multiProcessorCount * 4 * 8);
  run<0>("five-op rows, single instructions", out, p.multiProcessorCount, 20000);
  run<1>("five-op rows, row-local ops packed", out, p.multiProcessorCount, 20000);
  return 0;
}

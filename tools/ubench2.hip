// VALU issue microbenchmark 2 (gfx950): does the register bank of the operands, the distance between dependent instructions or
// the encoding (VOP2 / VOP3 / DPP) change the issue rate of fp32 multiply-adds?  Explicit physical registers.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o tools/ubench2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51"
// 16 instructions per block
#define X16(a) a a a a a a a a a a a a a a a a
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters) {
  asm volatile("v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n v_mov_b32 v24, 0.5\n v_mov_b32 v25, 0.5\n v_mov_b32 v26, 0.5\n v_mov_b32 v27, 0.5\n"
               "v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n"
               "v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"
               "v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n" ::: CLOB);
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {   // v_fma, sources in three different banks (21, 22, 23 -> 1, 2, 3), 16 independent destinations
      asm volatile("v_fma_f32 v28, v21, v22, v23\n v_fma_f32 v29, v21, v22, v23\n v_fma_f32 v30, v21, v22, v23\n v_fma_f32 v31, v21, v22, v23\n"
                   "v_fma_f32 v32, v21, v22, v23\n v_fma_f32 v33, v21, v22, v23\n v_fma_f32 v34, v21, v22, v23\n v_fma_f32 v35, v21, v22, v23\n"
                   "v_fma_f32 v36, v21, v22, v23\n v_fma_f32 v37, v21, v22, v23\n v_fma_f32 v38, v21, v22, v23\n v_fma_f32 v39, v21, v22, v23\n"
                   "v_fma_f32 v40, v21, v22, v23\n v_fma_f32 v41, v21, v22, v23\n v_fma_f32 v42, v21, v22, v23\n v_fma_f32 v43, v21, v22, v23\n" ::: CLOB);
    } else if (MODE == 1) {   // all three sources in the same bank (20, 24, 44 -> 0, 0, 0)
      asm volatile("v_fma_f32 v28, v20, v24, v44\n v_fma_f32 v29, v20, v24, v44\n v_fma_f32 v30, v20, v24, v44\n v_fma_f32 v31, v20, v24, v44\n"
                   "v_fma_f32 v32, v20, v24, v44\n v_fma_f32 v33, v20, v24, v44\n v_fma_f32 v34, v20, v24, v44\n v_fma_f32 v35, v20, v24, v44\n"
                   "v_fma_f32 v36, v20, v24, v44\n v_fma_f32 v37, v20, v24, v44\n v_fma_f32 v38, v20, v24, v44\n v_fma_f32 v39, v20, v24, v44\n"
                   "v_fma_f32 v40, v20, v24, v44\n v_fma_f32 v41, v20, v24, v44\n v_fma_f32 v42, v20, v24, v44\n v_fma_f32 v43, v20, v24, v44\n" ::: CLOB);
    } else if (MODE == 2) {   // two sources share a bank (20, 24 -> 0, 0; 21 -> 1)
      asm volatile("v_fma_f32 v28, v20, v24, v21\n v_fma_f32 v29, v20, v24, v21\n v_fma_f32 v30, v20, v24, v21\n v_fma_f32 v31, v20, v24, v21\n"
                   "v_fma_f32 v32, v20, v24, v21\n v_fma_f32 v33, v20, v24, v21\n v_fma_f32 v34, v20, v24, v21\n v_fma_f32 v35, v20, v24, v21\n"
                   "v_fma_f32 v36, v20, v24, v21\n v_fma_f32 v37, v20, v24, v21\n v_fma_f32 v38, v20, v24, v21\n v_fma_f32 v39, v20, v24, v21\n"
                   "v_fma_f32 v40, v20, v24, v21\n v_fma_f32 v41, v20, v24, v21\n v_fma_f32 v42, v20, v24, v21\n v_fma_f32 v43, v20, v24, v21\n" ::: CLOB);
    } else if (MODE == 3) {   // VOP2 v_fmac, independent, distinct banks (dst 28.., srcs 21, 22)
      asm volatile("v_fmac_f32 v28, v21, v22\n v_fmac_f32 v29, v21, v22\n v_fmac_f32 v30, v21, v22\n v_fmac_f32 v31, v21, v22\n"
                   "v_fmac_f32 v32, v21, v22\n v_fmac_f32 v33, v21, v22\n v_fmac_f32 v34, v21, v22\n v_fmac_f32 v35, v21, v22\n"
                   "v_fmac_f32 v36, v21, v22\n v_fmac_f32 v37, v21, v22\n v_fmac_f32 v38, v21, v22\n v_fmac_f32 v39, v21, v22\n"
                   "v_fmac_f32 v40, v21, v22\n v_fmac_f32 v41, v21, v22\n v_fmac_f32 v42, v21, v22\n v_fmac_f32 v43, v21, v22\n" ::: CLOB);
    } else if (MODE == 4) {   // VOP2 v_mul, independent
      asm volatile("v_mul_f32 v28, v21, v22\n v_mul_f32 v29, v21, v22\n v_mul_f32 v30, v21, v22\n v_mul_f32 v31, v21, v22\n"
                   "v_mul_f32 v32, v21, v22\n v_mul_f32 v33, v21, v22\n v_mul_f32 v34, v21, v22\n v_mul_f32 v35, v21, v22\n"
                   "v_mul_f32 v36, v21, v22\n v_mul_f32 v37, v21, v22\n v_mul_f32 v38, v21, v22\n v_mul_f32 v39, v21, v22\n"
                   "v_mul_f32 v40, v21, v22\n v_mul_f32 v41, v21, v22\n v_mul_f32 v42, v21, v22\n v_mul_f32 v43, v21, v22\n" ::: CLOB);
    } else if (MODE == 5) {   // dependent chain, distance 1 (every instruction needs the previous one)
      asm volatile(X16("v_fma_f32 v28, v28, v21, v22\n") ::: CLOB);
    } else if (MODE == 6) {   // two interleaved chains: distance 2
      asm volatile("v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n"
                   "v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n"
                   "v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n"
                   "v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n" ::: CLOB);
    } else if (MODE == 7) {   // three chains: distance 3
      asm volatile("v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n v_fma_f32 v30, v30, v21, v22\n v_fma_f32 v28, v28, v21, v22\n"
                   "v_fma_f32 v29, v29, v21, v22\n v_fma_f32 v30, v30, v21, v22\n v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n"
                   "v_fma_f32 v30, v30, v21, v22\n v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n v_fma_f32 v30, v30, v21, v22\n"
                   "v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n v_fma_f32 v30, v30, v21, v22\n v_fma_f32 v28, v28, v21, v22\n" ::: CLOB);
    } else if (MODE == 8) {   // four chains: distance 4
      asm volatile("v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n v_fma_f32 v30, v30, v21, v22\n v_fma_f32 v31, v31, v21, v22\n"
                   "v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n v_fma_f32 v30, v30, v21, v22\n v_fma_f32 v31, v31, v21, v22\n"
                   "v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n v_fma_f32 v30, v30, v21, v22\n v_fma_f32 v31, v31, v21, v22\n"
                   "v_fma_f32 v28, v28, v21, v22\n v_fma_f32 v29, v29, v21, v22\n v_fma_f32 v30, v30, v21, v22\n v_fma_f32 v31, v31, v21, v22\n" ::: CLOB);
    } else if (MODE == 9) {   // the PairHMM row as written in column_rows (7 instructions x 2 rows + 2 fillers), registers as an allocator might pick
      asm volatile("v_fma_f32 v50, v30, v40, v32\n v_mul_f32 v32, v32, v41\n v_mul_f32 v30, v29, v41\n v_fmac_f32 v50, v31, v42\n v_fmac_f32 v32, v31, v43\n v_fmac_f32 v30, v28, v44\n v_mul_f32 v31, v45, v51\n"
                   "v_fma_f32 v51, v33, v46, v35\n v_mul_f32 v35, v35, v47\n v_mul_f32 v33, v30, v47\n v_fmac_f32 v51, v34, v48\n v_fmac_f32 v35, v34, v49\n v_fmac_f32 v33, v31, v20\n v_mul_f32 v34, v21, v50\n"
                   "v_mul_f32 v36, v21, v22\n v_mul_f32 v37, v21, v22\n" ::: CLOB);
    } else if (MODE == 10) {  // 16 x v_mul_f32_dpp row_shr:1
      asm volatile(X16("v_mul_f32_dpp v28, v21, v22 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") ::: CLOB);
    } else if (MODE == 12) {  // v_pk_fma_f32, 8 independent destinations x 2 (sources in different bank pairs)
      asm volatile("v_pk_fma_f32 v[28:29], v[20:21], v[22:23], v[24:25]\n v_pk_fma_f32 v[30:31], v[20:21], v[22:23], v[24:25]\n v_pk_fma_f32 v[32:33], v[20:21], v[22:23], v[24:25]\n v_pk_fma_f32 v[34:35], v[20:21], v[22:23], v[24:25]\n"
                   "v_pk_fma_f32 v[36:37], v[20:21], v[22:23], v[24:25]\n v_pk_fma_f32 v[38:39], v[20:21], v[22:23], v[24:25]\n v_pk_fma_f32 v[40:41], v[20:21], v[22:23], v[24:25]\n v_pk_fma_f32 v[42:43], v[20:21], v[22:23], v[24:25]\n"
                   "v_pk_fma_f32 v[28:29], v[20:21], v[22:23], v[24:25]\n v_pk_fma_f32 v[30:31], v[20:21], v[22:23], v[24:25]\n v_pk_fma_f32 v[32:33], v[20:21], v[22:23], v[24:25]\n v_pk_fma_f32 v[34:35], v[20:21], v[22:23], v[24:25]\n"
                   "v_pk_fma_f32 v[36:37], v[20:21], v[22:23], v[24:25]\n v_pk_fma_f32 v[38:39], v[20:21], v[22:23], v[24:25]\n v_pk_fma_f32 v[40:41], v[20:21], v[22:23], v[24:25]\n v_pk_fma_f32 v[42:43], v[20:21], v[22:23], v[24:25]\n" ::: CLOB);
    } else if (MODE == 13) {  // v_pk_mul_f32
      asm volatile("v_pk_mul_f32 v[28:29], v[20:21], v[22:23]\n v_pk_mul_f32 v[30:31], v[20:21], v[22:23]\n v_pk_mul_f32 v[32:33], v[20:21], v[22:23]\n v_pk_mul_f32 v[34:35], v[20:21], v[22:23]\n"
                   "v_pk_mul_f32 v[36:37], v[20:21], v[22:23]\n v_pk_mul_f32 v[38:39], v[20:21], v[22:23]\n v_pk_mul_f32 v[40:41], v[20:21], v[22:23]\n v_pk_mul_f32 v[42:43], v[20:21], v[22:23]\n"
                   "v_pk_mul_f32 v[28:29], v[20:21], v[22:23]\n v_pk_mul_f32 v[30:31], v[20:21], v[22:23]\n v_pk_mul_f32 v[32:33], v[20:21], v[22:23]\n v_pk_mul_f32 v[34:35], v[20:21], v[22:23]\n"
                   "v_pk_mul_f32 v[36:37], v[20:21], v[22:23]\n v_pk_mul_f32 v[38:39], v[20:21], v[22:23]\n v_pk_mul_f32 v[40:41], v[20:21], v[22:23]\n v_pk_mul_f32 v[42:43], v[20:21], v[22:23]\n" ::: CLOB);
    } else if (MODE == 14) {  // packed PairHMM rows: per 2 rows 4 packed + 6 single; 3 x (2 rows) + 2 packed fillers = 32 instructions, counted as 2 x 16
      asm volatile("v_pk_fma_f32 v[50:51], v[30:31], v[40:41], v[32:33]\n v_pk_mul_f32 v[32:33], v[32:33], v[42:43]\n v_pk_fma_f32 v[50:51], v[34:35], v[44:45], v[50:51]\n v_mul_f32 v30, v29, v42\n v_pk_fma_f32 v[32:33], v[34:35], v[46:47], v[32:33]\n"
                   "v_fmac_f32 v30, v28, v48\n v_mul_f32 v34, v20, v49\n v_mul_f32 v31, v30, v43\n v_mul_f32 v35, v21, v50\n v_fmac_f32 v31, v34, v49\n"
                   "v_pk_fma_f32 v[48:49], v[36:37], v[40:41], v[38:39]\n v_pk_mul_f32 v[38:39], v[38:39], v[42:43]\n v_pk_fma_f32 v[48:49], v[24:25], v[44:45], v[48:49]\n v_mul_f32 v36, v31, v42\n v_pk_fma_f32 v[38:39], v[24:25], v[46:47], v[38:39]\n"
                   "v_fmac_f32 v36, v35, v40\n v_mul_f32 v24, v22, v51\n v_mul_f32 v37, v36, v43\n v_mul_f32 v25, v23, v48\n v_fmac_f32 v37, v24, v41\n"
                   "v_pk_fma_f32 v[50:51], v[30:31], v[40:41], v[32:33]\n v_pk_mul_f32 v[32:33], v[32:33], v[42:43]\n v_pk_fma_f32 v[50:51], v[34:35], v[44:45], v[50:51]\n v_mul_f32 v30, v37, v42\n v_pk_fma_f32 v[32:33], v[34:35], v[46:47], v[32:33]\n"
                   "v_fmac_f32 v30, v25, v48\n v_mul_f32 v34, v20, v49\n v_mul_f32 v31, v30, v43\n v_mul_f32 v35, v21, v50\n v_fmac_f32 v31, v34, v49\n"
                   "v_pk_mul_f32 v[26:27], v[20:21], v[22:23]\n v_pk_mul_f32 v[26:27], v[20:21], v[22:23]\n" ::: CLOB);
    } else if (MODE == 11) {  // s_waitcnt / s_nop mixed in: 12 fma + 2 s_waitcnt lgkmcnt(4) + 2 s_nop (counted as 16)
      asm volatile("v_fma_f32 v28, v21, v22, v23\n v_fma_f32 v29, v21, v22, v23\n v_fma_f32 v30, v21, v22, v23\n s_waitcnt lgkmcnt(4)\n v_fma_f32 v31, v21, v22, v23\n"
                   "v_fma_f32 v32, v21, v22, v23\n v_fma_f32 v33, v21, v22, v23\n s_nop 0\n v_fma_f32 v34, v21, v22, v23\n v_fma_f32 v35, v21, v22, v23\n"
                   "v_fma_f32 v36, v21, v22, v23\n s_waitcnt lgkmcnt(4)\n v_fma_f32 v37, v21, v22, v23\n v_fma_f32 v38, v21, v22, v23\n s_nop 0\n v_fma_f32 v39, v21, v22, v23\n" ::: CLOB);
    }
  }
  float r;
  asm volatile("v_add_f32 %0, v28, v29\n v_add_f32 %0, %0, v30\n v_add_f32 %0, %0, v50" : "=v"(r)::CLOB);
  out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int M> void run(const char* name, float* out, int cus, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-44s", name);
  for (int w : {1, 2, 3, 4, 8}) {
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<M>, dim3(cus * 4 * w), dim3(64), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("  w%d %6.3f", w, (double)best * 1e6 / ((double)w * iters * 16));
  }
  printf("   ns per wave-instruction per SIMD\n");
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount;
  printf("device %s CUs %d\n", p.name, cus);
  float* out; hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8);
  const int it = 20000;
  run<0>("v_fma sources in 3 banks", out, cus, it);
  run<1>("v_fma sources in 1 bank", out, cus, it);
  run<2>("v_fma two sources share a bank", out, cus, it);
  run<3>("v_fmac (VOP2) independent", out, cus, it);
  run<4>("v_mul (VOP2) independent", out, cus, it);
  run<5>("v_fma dependent, distance 1", out, cus, it);
  run<6>("v_fma dependent, distance 2", out, cus, it);
  run<7>("v_fma dependent, distance 3", out, cus, it);
  run<8>("v_fma dependent, distance 4", out, cus, it);
  run<9>("PairHMM row sequence (2 rows + 2 fillers)", out, cus, it);
  run<10>("v_mul_f32_dpp row_shr:1", out, cus, it);
  run<11>("12 fma + 2 s_waitcnt + 2 s_nop", out, cus, it);
  run<12>("v_pk_fma_f32 independent", out, cus, it);
  run<13>("v_pk_mul_f32 independent", out, cus, it);
  run<14>("packed PairHMM rows (32 instr = 6.4 rows; x0.5)", out, cus, it);
  return 0;
}

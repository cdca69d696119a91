// fp64 instantiations of the PairHMM kernel (full-fp64 runs, the rescue pass in both modes) and the rescue planner.
#include "phmm_kernel_impl.h"
namespace accg {
hipError_t phmm_rescue_plan_launch(const PhmmPlanArgs& p, uint32_t n_regions, hipStream_t s) {
  if (n_regions == 0) return hipSuccess;
  hipLaunchKernelGGL(phmm_rescue_plan, dim3(n_regions), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t phmm_launch_f64(int K, int lpp, bool striped, const PhmmArgs<double>& a, uint32_t wb, uint32_t n, hipStream_t s) {
  return launch<double, true, false>(K, lpp, a, wb, n, s, striped);
}
hipError_t phmm_launch_rescue_f64(int K, int lpp, bool strict, bool striped, const PhmmArgs<double>& a, uint32_t wb, uint32_t n, hipStream_t s,
                                  uint32_t grid_cap, bool form5, int wg) {
  // wg = 2 (fast mode, K <= 8): the items come in pairs that share their dist table (phmm_kernel)
  if (wg == 2 && !strict && !striped && K <= 8)
    return form5 ? launch<double, false, true, 5, 2>(K, lpp, a, wb, n, s, striped, grid_cap) : launch<double, false, true, 0, 2>(K, lpp, a, wb, n, s, striped, grid_cap);
  // form5 (fast mode, a batch all of whose reads pass the five-operation form's range tests, classes with the column in assembly):
  // five fp64 operations per cell instead of seven
  if (form5 && !strict && !striped && K <= PHMM_ASM_MAX_K_F64) return launch<double, false, true, 5>(K, lpp, a, wb, n, s, striped, grid_cap);
  // strict: the reference's operation order throughout.  Otherwise the 7-op contraction of the fast mode, which is within 1e-8 of
  // it -- except where the fp64 likelihood x 2^1020 comes within ~1e28 of the smallest normal double: there, which values get
  // flushed (x86 FTZ, matched on the device) depends on the last bits of every intermediate, and a contracted result landed
  // 2.6e-5 away from compute_fp_avxd on log10 (the reference's own scalar baseline built with -mfma deviates by exactly as
  // much; found by tools/fuzz_phmm.py).  A job that produces such a result is redone in the reference's order by the same
  // wavefront (phmm_kernel), so the fast mode is bit-equal to the strict one for those pairs.
  return strict ? launch<double, true, true>(K, lpp, a, wb, n, s, striped, grid_cap) : launch<double, false, true>(K, lpp, a, wb, n, s, striped, grid_cap);
}

}  // namespace accg

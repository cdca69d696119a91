// fp32 fast-mode instantiations of the PairHMM kernel (column in assembly; seven- and six-operation forms).
#include "phmm_kernel_impl.h"
namespace accg {
hipError_t phmm_launch_f32_fast(int K, int lpp, bool x6, const PhmmArgs<float>& a, uint32_t wb, uint32_t n, hipStream_t s) {
#ifdef PHMM_TIMING
  hipError_t e = x6 ? launch<float, false, false, true>(K, lpp, a, wb, n, s) : launch<float, false, false, false>(K, lpp, a, wb, n, s);
  static int calls = 0;
  if (++calls % 100 == 0) hipLaunchKernelGGL(phmm_timing_print, dim3(1), dim3(1), 0, s);
  return e;
#else
  return x6 ? launch<float, false, false, true>(K, lpp, a, wb, n, s) : launch<float, false, false, false>(K, lpp, a, wb, n, s);
#endif
}
}  // namespace accg

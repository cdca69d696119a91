// fp32 fast-mode instantiations of the PairHMM kernel (column in assembly; seven-, six- and five-operation forms).
#include "phmm_kernel_impl.h"
namespace accg {
hipError_t phmm_launch_f32_fast(int K, int lpp, int form, const PhmmArgs<float>& a, uint32_t wb, uint32_t n, hipStream_t s) {
  hipError_t e = form == 5 ? launch<float, false, false, 5>(K, lpp, a, wb, n, s)
               : form == 6 ? launch<float, false, false, 6>(K, lpp, a, wb, n, s) : launch<float, false, false, 0>(K, lpp, a, wb, n, s);
#ifdef PHMM_TIMING
  static int calls = 0;
  if (++calls % 100 == 0) hipLaunchKernelGGL(phmm_timing_print, dim3(1), dim3(1), 0, s);
#endif
  return e;
}
}  // namespace accg

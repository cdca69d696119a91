// fp32 fast-mode instantiations of the PairHMM kernel (column in assembly; seven-, six- and five-operation forms; the
// five-operation form also as workgroups of two wavefronts sharing the dist table).
#include "phmm_kernel_impl.h"
namespace accg {
hipError_t phmm_launch_f32_fast(int K, int lpp, int form, const PhmmArgs<float>& a, uint32_t wb, uint32_t n, hipStream_t s, int wg) {
  if (wg == 2 && form != 5) return hipErrorInvalidValue;        // pairs are only built for the five-operation form (phmm_host.cpp)
  hipError_t e = wg == 2    ? launch<float, false, false, 5, 2>(K, lpp, a, wb, n, s)
               : form == 5 ? launch<float, false, false, 5>(K, lpp, a, wb, n, s)
               : form == 6 ? launch<float, false, false, 6>(K, lpp, a, wb, n, s) : launch<float, false, false, 0>(K, lpp, a, wb, n, s);
#ifdef PHMM_TIMING
  static int calls = 0;
  if (++calls % 100 == 99) hipLaunchKernelGGL(phmm_timing_reset_wall, dim3(1), dim3(1), 0, s);     // the 100th launch alone in the wall-clock spread
  if (calls % 100 == 0) hipLaunchKernelGGL(phmm_timing_print, dim3(1), dim3(1), 0, s);
#endif
  return e;
}
// windows of K that one merged launch covers (phmm_kernel_multi): {2..5}, {6..13}, both lane counts
hipError_t phmm_launch_f32_multi(int k_lo, int k_hi, size_t lds_bytes, const PhmmArgs<float>& a, uint32_t wb, uint32_t n, hipStream_t s, int wg) {
  return wg == 2 ? launch_multi<2>(k_lo, k_hi, lds_bytes, a, wb, n, s) : launch_multi<1>(k_lo, k_hi, lds_bytes, a, wb, n, s);
}
hipError_t phmm_prepare_rows_launch(const PhmmArgs<float>& a, uint32_t n_reads, uint32_t* state, uint32_t state_words, hipStream_t s) {
  if (n_reads == 0) return hipSuccess;
  hipLaunchKernelGGL(phmm_prepare_rows, dim3(n_reads), dim3(128), 0, s, a, n_reads, state, state_words);
  return hipGetLastError();
}
}  // namespace accg

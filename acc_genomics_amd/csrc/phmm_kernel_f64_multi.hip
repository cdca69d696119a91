// The merged launches of the fp64 rescue pass (phmm_dev.h: PHMM_RESCUE_MERGED): a translation unit of their own for the parallel build.
#include "phmm_kernel_impl.h"
namespace accg {
hipError_t phmm_launch_rescue_multi(int window, int wg, size_t lds_bytes, const PhmmArgs<double>& a, const PhmmRescueSet& rs, uint32_t grid, hipStream_t s) {
  return launch_rescue_multi(window, wg, lds_bytes, a, rs, grid, s);
}
hipError_t phmm_launch_redo_multi(size_t lds_bytes, const PhmmArgs<double>& a, const PhmmRescueSet& rs, uint32_t grid, hipStream_t s) {
  return launch_redo_multi(lds_bytes, a, rs, grid, s);
}

}  // namespace accg

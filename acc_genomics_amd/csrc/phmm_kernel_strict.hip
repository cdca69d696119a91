// fp32 strict-mode instantiations of the PairHMM kernel (the reference's operation order, compiled column), the striped fp32
// kernels for reads of 1024 bases and more, and the fp32 dispatcher.
#include "phmm_kernel_impl.h"
namespace accg {
hipError_t phmm_launch_f32_fast(int K, int lpp, int form, const PhmmArgs<float>& a, uint32_t wb, uint32_t n, hipStream_t s, int wg);
hipError_t phmm_launch_f32(int K, int lpp, bool strict, int form, bool striped, const PhmmArgs<float>& a, uint32_t wb, uint32_t n, hipStream_t s, int wg) {
  if (striped) return strict ? launch<float, true, false>(K, lpp, a, wb, n, s, true) : launch<float, false, false>(K, lpp, a, wb, n, s, true);
  if (strict) return launch<float, true, false>(K, lpp, a, wb, n, s);
  return phmm_launch_f32_fast(K, lpp, form, a, wb, n, s, wg);    // (strict launches run a pair's two items as two independent wavefronts)
}

// (lanes per read, rows per lane) for a read of `len` bases; K = 0: longer than the kernels support
void phmm_pick(uint32_t len, int* lpp, int* K, int max_k8) {
  const uint32_t rows = len + 1;            // one row reserved as "row 0"
  // 8 lanes per read: twice the rows per lane, so the per-step overhead (hand-off, stream and table reads) is spread
  // over twice the cells, and the padding to a multiple of the lane count halves
  if (rows <= 8u * (uint32_t)(max_k8 < PHMM_MAX_K ? max_k8 : PHMM_MAX_K)) { *lpp = 8; *K = (int)((rows + 7) / 8); return; }
  if (rows <= 256) { *lpp = 16; *K = (int)((rows + 15) / 16); return; }
  static const int ks[] = {9, 10, 12, 14, 16};
  for (int l : {32, 64}) {
    const int need = (int)((rows + l - 1) / l);
    for (int k : ks) if (k >= need) { *lpp = l; *K = k; return; }
  }
  *lpp = 64; *K = 16;          // 1024 rows and more: striped (phmm_striped)
}

}  // namespace accg

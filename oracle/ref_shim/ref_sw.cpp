// TEST INFRASTRUCTURE ONLY -- driver that compiles the reference's own HTC Smith-Waterman CPU path in place.
//
// Reference sources compiled where they lie under /root/reference (never copied):
//   htc-sw/host/FalconSW_AVX.cpp        -> SWPairwiseAlignmentMultiBatch (:304), calculateMatrixOneBatch (:1693),
//                                           calculateMatrixRowWiseSIMDUnroll4x (:967), calculateCigarOneBatch (:2303)
//   htc-sw/intel_avx/PairWiseSW.h       -> smithWatermanBackTrack (:41), getCIGAR (:243), runSWOnePairBT_avx2 (:440)
//                                           (included here through intel_avx/avx2-smithwaterman.h, exactly as
//                                            intel_avx/avx2_impl.cc:3 does)
// Output: oracle/_ref/libaccg_ref_sw.so, used by tests/ and bench.py's cpu_baseline leg only.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "intel_avx/avx2_impl.h"            // -> smithwaterman_common.h -> host/common.h, as avx2_impl.cc:2 does
#include "intel_avx/avx2-smithwaterman.h"   // smithWatermanBackTrack / getCIGAR / runSWOnePairBT_avx2 (avx2_impl.cc:4)

// Globals and two helpers that the reference keeps in its test driver htc-sw/host/sw_host.cpp:9-26,92-103
// (a file with a main()); FalconSW_AVX.cpp:25-35 declares them extern.
long calMatrix_C_time = 0, calCigar_C_time = 0, malloc_time = 0, SW_complexity = 0;
int addCigarElement(struct Cigar* cigar, int length, int state) {
  if (cigar->CigarElementNum < 0) return -1;
  if (length > 0) {
    cigar->cigarElements[cigar->CigarElementNum].length = length;
    cigar->cigarElements[cigar->CigarElementNum].state = state;
    cigar->CigarElementNum++;
  }
  return 0;
}
struct timespec diff_time(struct timespec start, struct timespec end) {
  struct timespec t;
  if ((end.tv_nsec - start.tv_nsec) < 0) { t.tv_sec = end.tv_sec - start.tv_sec - 1; t.tv_nsec = 1000000000 + end.tv_nsec - start.tv_nsec; }
  else { t.tv_sec = end.tv_sec - start.tv_sec; t.tv_nsec = end.tv_nsec - start.tv_nsec; }
  return t;
}
int calculateMatrixRowWiseOpt(char*, char*, int**, int, int, int**, int, int, int, int, int, int);

extern "C" {

int ref_sw_cigar_struct_bytes() { return (int)sizeof(struct Cigar); }

// One ref x B alts through SWPairwiseAlignmentMultiBatch (FalconSW_AVX.cpp:304). alts is B rows of
// MAX_SEQ_LENGTH bytes. cig_len/cig_state are [B][max_el]; n_el[B]; offs[B].
int ref_sw_multibatch(const char* ref, int refLen, const char* alts, int B, const int* altLens, int strategy,
                      int option, int max_el, int* n_el, int* cig_len, int* cig_state, int* offs) {
  struct Cigar* cg = (struct Cigar*)calloc(B, sizeof(struct Cigar));
  int rc = SWPairwiseAlignmentMultiBatch((char*)ref, refLen, (char(*)[MAX_SEQ_LENGTH])alts, B, (int*)altLens, cg,
                                         offs, strategy, option);
  for (int b = 0; b < B && rc == 0; b++) {
    n_el[b] = cg[b].CigarElementNum;
    for (int e = 0; e < cg[b].CigarElementNum && e < max_el; e++) {
      cig_len[(size_t)b * max_el + e] = cg[b].cigarElements[e].length;
      cig_state[(size_t)b * max_el + e] = cg[b].cigarElements[e].state;
    }
  }
  free(cg);
  return rc;
}

// Fill only: sw / btrack are (refLen+1) x (altLen+1) row-major, zero-initialised here as
// SWPairwiseAlignmentOneBatch does (FalconSW_AVX.cpp:331-355). option 0 = SIMD default, 1 = scalar.
int ref_sw_matrix(const char* ref, const char* alt, int refLen, int altLen, int strategy, int option, int* sw,
                  int* btrack) {
  int n = refLen + 1, m = altLen + 1;
  int** swp = (int**)malloc(n * sizeof(int*));
  int** btp = (int**)malloc(n * sizeof(int*));
  memset(sw, 0, (size_t)n * m * sizeof(int));
  memset(btrack, 0, (size_t)n * m * sizeof(int));
  for (int i = 0; i < n; i++) { swp[i] = sw + (size_t)i * m; btp[i] = btrack + (size_t)i * m; }
  // the SIMD variant reads a few bytes past the sequences; give it padded copies
  char* r = (char*)calloc(refLen + 64, 1); char* a = (char*)calloc(altLen + 64, 1);
  memcpy(r, ref, refLen); memcpy(a, alt, altLen);
  int rc;
  if (option == 0) rc = calculateMatrixRowWiseSIMDUnroll4x(r, a, swp, m, n, btp, strategy, 0, W_MATCH, W_MISMATCH, W_OPEN, W_EXTEND);
  else rc = calculateMatrixOneBatch(r, a, swp, m, n, btp, strategy, 0);
  free(r); free(a); free(swp); free(btp);
  return rc;
}

// Backtrace only, on a caller-supplied matrix (calculateCigarOneBatch, FalconSW_AVX.cpp:2303).
int ref_sw_cigar_from_matrix(int* sw, int* btrack, int refLen, int altLen, int strategy, int max_el, int* n_el,
                             int* cig_len, int* cig_state, int* off) {
  int n = refLen + 1, m = altLen + 1;
  int** swp = (int**)malloc(n * sizeof(int*));
  int** btp = (int**)malloc(n * sizeof(int*));
  for (int i = 0; i < n; i++) { swp[i] = sw + (size_t)i * m; btp[i] = btrack + (size_t)i * m; }
  struct Cigar* cg = (struct Cigar*)calloc(1, sizeof(struct Cigar));
  int rc = calculateCigarOneBatch(swp, btp, n, m, strategy, cg, off);
  *n_el = cg->CigarElementNum;
  for (int e = 0; e < cg->CigarElementNum && e < max_el; e++) { cig_len[e] = cg->cigarElements[e].length; cig_state[e] = cg->cigarElements[e].state; }
  free(cg); free(swp); free(btp);
  return rc;
}

// intel_avx pair entry (runSWOnePairBT_avx2, PairWiseSW.h:440-470) -- the "intel_avx" CPU baseline.
int ref_sw_gkl_pair(int match, int mismatch, int open, int extend, const unsigned char* seq1,
                    const unsigned char* seq2, int len1, int len2, int strategy, int max_el, int* n_el,
                    int* cig_len, int* cig_state) {
  struct Cigar* cg = (struct Cigar*)calloc(1, sizeof(struct Cigar));
  int off = runSWOnePairBT_avx2(match, mismatch, open, extend, (uint8_t*)seq1, (uint8_t*)seq2, len1, len2,
                                (int8_t)strategy, cg);
  *n_el = cg->CigarElementNum;
  for (int e = 0; e < cg->CigarElementNum && e < max_el; e++) { cig_len[e] = cg->cigarElements[e].length; cig_state[e] = cg->cigarElements[e].state; }
  free(cg);
  return off;
}

// Same fill as runSWOnePairBT_avx2 but stopping after smithWatermanBackTrack (PairWiseSW.h:41) so that
// p.score / p.max_i / p.max_j -- the quantities BASELINE.json asks to be bit-exact -- can be read.
void ref_sw_gkl_score(int match, int mismatch, int open, int extend, const unsigned char* seq1,
                      const unsigned char* seq2, int len1, int len2, int strategy, int* score, int* max_i,
                      int* max_j) {
  int32_t* E_ = (int32_t*)_mm_malloc((6 * (MAX_SEQ_LEN + AVX_LENGTH)) * sizeof(int32_t), 64);
  int16_t* bt = (int16_t*)_mm_malloc(((size_t)2 * MAX_SEQ_LEN * MAX_SEQ_LEN + 2 * AVX_LENGTH) * sizeof(int16_t), 64);
  SeqPair p;
  p.seq1 = (uint8_t*)seq1; p.seq2 = (uint8_t*)seq2; p.len1 = len1; p.len2 = len2; p.overhangStrategy = (int8_t)strategy;
  p.btrack = bt; p.cigar = NULL;
  smithWatermanBackTrack(&p, match, mismatch, open, extend, E_, 0);
  *score = p.score; *max_i = p.max_i; *max_j = p.max_j;
  _mm_free(E_); _mm_free(bt);
}

// Throughput helpers for bench.py's cpu_baseline leg: run `n` independent pairs laid out as fixed-stride rows.
void ref_sw_gkl_many(const unsigned char* refs, int ref_stride, const int* refLens, const unsigned char* alts,
                     int alt_stride, const int* altLens, int n, int strategy, int* offs) {
  struct Cigar* cg = (struct Cigar*)calloc(1, sizeof(struct Cigar));
  for (int k = 0; k < n; k++)
    offs[k] = runSWOnePairBT_avx2(W_MATCH, W_MISMATCH, W_OPEN, W_EXTEND, (uint8_t*)(refs + (size_t)k * ref_stride),
                                  (uint8_t*)(alts + (size_t)k * alt_stride), refLens[k], altLens[k], (int8_t)strategy, cg);
  free(cg);
}

}  // extern "C"

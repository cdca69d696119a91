/* TEST INFRASTRUCTURE ONLY -- see oracle.h.  BWA-MEM seed extension (banded, ksw_extend2-style), CPU restatement.
 *
 * PARITY UNPINNED: the reference for this path is FPGA device code, bwa-sw/sdaccel/smithwaterman.cpp (sw_extend :75-273,
 * seed_proc :511-672).  It needs Xilinx's ap_int.h / hls_stream.h, its host needs libbwa, and its golden files are on S3
 * (bwa-sw/intel/aocl/testdata/get-data.sh:2): nothing of it can be built or replayed here.  This file restates the two
 * functions with plain ints (the ap_int widths are wide enough for reads <= 255 bp, see the notes inline); what pins it is an
 * independent re-implementation of upstream BWA's published ksw_extend2 recurrence in tests/test_bwasw_oracle.py
 * (score, end points, global score, max offset on random inputs).
 *
 * Scoring is fixed as in the device code (:28-35): match 1, mismatch -4, N -1, gap open 6, extend 1, pen_clip 5, w 100. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

enum { O_DEL = 6, E_DEL = 1, O_INS = 6, E_INS = 1, PEN_CLIP = 5, W_IN = 100 };

static int mat(int a, int b) { return (a > 3 || b > 3) ? -1 : (a == b ? 1 : -4); }   /* my_mat, :112 */

typedef struct { int max, max_i, max_j, max_ie, gscore, max_off; } ext_state;

/* sw_extend (:75-273).  q = query codes (qlen), t = target codes (tlen).  eh_h / eh_e persist across the two band tries,
 * exactly as the device arrays do.  Returns the band actually used (aw_tmp). */
static int sw_extend(const uint8_t* q, int qlen, const uint8_t* t, int tlen, int h0, int* regScore, int max_ins, int max_del,
                     int* qle, int* tle, int* gtle, int* gscore_out, int* maxoff) {
  const int oe_del = O_DEL + E_DEL, oe_ins = O_INS + E_INS;
  int eh_h[258], eh_e[258];
  ext_state s = {h0, -1, -1, -1, -1, 0};
  for (int j = 0; j <= qlen; j++) eh_h[j] = eh_e[j] = 0;
  int k = 0, stop = 0, aw_tmp = W_IN;
  while (k < 2 && !stop) {
    const int prev = *regScore;
    aw_tmp = (W_IN << k) & 0xFF;                        /* uint8_t aw_tmp, :138 */
    int aw1 = aw_tmp < max_ins ? aw_tmp : max_ins;
    aw1 = aw1 < max_del ? aw1 : max_del;
    int beg = 0, end = qlen;
    int tmp_eme = h0 - oe_ins; if (tmp_eme < 0) tmp_eme = 0;
    int h1_init = h0 - O_DEL;
    for (int i = 0; i < tlen; i++) {
      int f = 0, m = 0, mj = -1, h1;
      if (beg < i - aw1) beg = i - aw1;
      if (end > i + aw1 + 1) end = i + aw1 + 1;
      if (end > qlen) end = qlen;
      if (beg == 0) { h1_init -= E_DEL; h1 = h1_init < 0 ? 0 : h1_init; } else h1 = 0;
      int backw = 0, forw = 0, forw_done = 0, j;
      for (j = beg; j < end; j++) {
        int h, e, M;
        if (i == 0) {                                   /* first row comes from h0, not from the arrays (:175-191) */
          e = 0;
          if (j == 0) h = h0;
          else if (j == 1) h = tmp_eme;
          else { tmp_eme -= E_INS; h = tmp_eme > 0 ? tmp_eme : 0; }
          M = h;
        } else { e = eh_e[j]; M = eh_h[j]; }
        const int h1_reg = h1;
        M = M ? M + mat(t[i], q[j]) : 0;
        h = M > e ? M : e;
        h = h > f ? h : f;
        h1 = h;
        int x = M - oe_del; if (x < 0) x = 0;
        e -= E_DEL; if (e < x) e = x;
        x = M - oe_ins; if (x < 0) x = 0;
        f -= E_INS; if (f < x) f = x;
        eh_e[j] = e;
        eh_h[j] = h1_reg;
        if (m <= h) { mj = j; m = h; }
        if (!forw_done) { if (h1_reg == 0 && e == 0) forw++; else forw_done = 1; }
        if (h1_reg == 0 && e == 0) backw++; else backw = 0;
      }
      eh_h[end] = h1; eh_e[end] = 0;
      if (h1 == 0) backw++; else backw = 0;
      if (j == qlen && s.gscore <= h1) { s.max_ie = i; s.gscore = h1; }
      if (m == 0) break;
      if (m > s.max) {
        s.max = m; s.max_i = i; s.max_j = mj;
        const int d = mj > i ? mj - i : i - mj;
        if (s.max_off < d) s.max_off = d;
      }
      beg += forw;
      end = end - backw + 2 < qlen ? end - backw + 2 : qlen;
    }
    *qle = s.max_j + 1; *tle = s.max_i + 1; *gtle = s.max_ie + 1; *gscore_out = s.gscore; *maxoff = s.max_off;
    *regScore = s.max;
    if (s.max == prev || s.max_off < (aw_tmp >> 1) + (aw_tmp >> 2)) stop = 1;
    k++;
  }
  return aw_tmp;
}

/* seed_proc (:586-670) for one seed: seq = [left query][right query][left target][right target] (codes 0-4),
 * par = {leftQlen, leftRlen, rightQlen, rightRlen, seed_len, seed_qbeg, seed_index}; out = {qBeg, qEnd, rBeg, rEnd, score,
 * trueScore, width} (the five words the device streams out pack exactly these, :666-670). */
static void seed_one(const uint8_t* seq, const uint16_t* par, int16_t out[7]) {
  const int qlen[2] = {par[0], par[2]}, tlen[2] = {par[1], par[3]};
  const int seed_len = par[4], seed_qbeg = par[5];
  int regScore = seed_len, aw[2] = {W_IN, W_IN};
  int qBeg = 0, qEnd = qlen[1], rBeg = 0, rEnd = 0, trueScore = regScore, score = 0;
  int qle = -1, tle = -1, gtle = -1, gscore = -1, maxoff = -1;
  const uint8_t* qs = seq;
  const uint8_t* ts = seq + qlen[0] + qlen[1];
  for (int i = 0; i < 2; i++) {
    const int sc0 = regScore;
    const int h0 = i == 0 ? seed_len : sc0;
    aw[i] = sw_extend(qs, qlen[i], ts, tlen[i], h0, &regScore, qlen[i], qlen[i], &qle, &tle, &gtle, &gscore, &maxoff);
    score = regScore;
    if (gscore <= 0 || gscore <= regScore - PEN_CLIP) {
      if (i == 0) { qBeg = seed_qbeg - qle; rBeg = -tle; trueScore = regScore; }
      else { qEnd = qle; rEnd = tle; trueScore += regScore - sc0; }
    } else {
      if (i == 0) { qBeg = 0; rBeg = -gtle; trueScore = gscore; }
      else { qEnd = qlen[1]; rEnd = gtle; trueScore += gscore - sc0; }
    }
    qs += qlen[i]; ts += tlen[i];
  }
  out[0] = (int16_t)qBeg; out[1] = (int16_t)qEnd; out[2] = (int16_t)rBeg; out[3] = (int16_t)rEnd;
  out[4] = (int16_t)score; out[5] = (int16_t)trueScore; out[6] = (int16_t)(aw[0] > aw[1] ? aw[0] : aw[1]);
}

void orc_bwasw_batch(const uint8_t* seqs, const uint32_t* seq_off, const uint16_t* params /* n x 7 */, int n, int16_t* out /* n x 7 */,
                     int n_threads) {
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads)
  for (int k = 0; k < n; k++) seed_one(seqs + seq_off[k], params + (size_t)k * 7, out + (size_t)k * 7);
}

/* one extension exposed for the brute-force pin: returns {max, qle, tle, gtle, gscore, max_off, band} */
void orc_bwasw_extend(const uint8_t* q, int qlen, const uint8_t* t, int tlen, int h0, int out[7]) {
  int regScore = h0, qle, tle, gtle, gscore, maxoff;
  out[6] = sw_extend(q, qlen, t, tlen, h0, &regScore, qlen, qlen, &qle, &tle, &gtle, &gscore, &maxoff);
  out[0] = regScore; out[1] = qle; out[2] = tle; out[3] = gtle; out[4] = gscore; out[5] = maxoff;
}

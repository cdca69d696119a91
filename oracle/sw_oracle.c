/* TEST INFRASTRUCTURE ONLY -- see oracle.h.  HTC (GATK SWPairwiseAlignment) Smith-Waterman, CPU restatement.
 * Follows the scalar definition htc-sw/host/FalconSW_AVX.cpp:1693-1823 (calculateMatrixOneBatch) and
 * the end-cell/backtrace rules of :2303-2419 (calculateCigarOneBatch); weights and state codes from
 * htc-sw/host/common.h:13-26.  Rows index the reference window, columns the alternate/read. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define NEG_INIT (-1073741824) /* FalconSW_AVX.cpp:1700 lowInitValue */
#define ST_M 0
#define ST_I 1
#define ST_D 2
#define ST_S 4

void orc_sw_fill(const char* ref, const char* alt, int refLen, int altLen, int strategy, int w_match,
                 int w_mismatch, int w_open, int w_extend, int* sw, int* bt) {
  const int n = refLen + 1, m = altLen + 1;
  memset(sw, 0, sizeof(int) * (size_t)n * m);
  memset(bt, 0, sizeof(int) * (size_t)n * m);
  int* gap_v = (int*)malloc(sizeof(int) * (size_t)(m + 1));  /* best vertical gap ending in column j */
  int* len_v = (int*)malloc(sizeof(int) * (size_t)(m + 1));
  for (int j = 0; j <= m; j++) { gap_v[j] = NEG_INIT; len_v[j] = 0; }
  if (strategy == ORC_SW_INDEL || strategy == ORC_SW_LEADING_INDEL) { /* :1732-1746 */
    for (int j = 1; j < m; j++) sw[j] = w_open + (j - 1) * w_extend;
    for (int i = 1; i < n; i++) sw[(size_t)i * m] = w_open + (i - 1) * w_extend;
  }
  for (int i = 1; i < n; i++) {
    const int* up = sw + (size_t)(i - 1) * m;
    int* cur = sw + (size_t)i * m;
    int* b = bt + (size_t)i * m;
    int gap_h = NEG_INIT, len_h = 0; /* per-row horizontal gap state, :1725-1728 */
    const char a = ref[i - 1];
    for (int j = 1; j < m; j++) {
      int diag = up[j - 1] + (a == alt[j - 1] ? w_match : w_mismatch);
      int open_v = up[j] + w_open;
      gap_v[j] += w_extend;
      if (open_v > gap_v[j]) { gap_v[j] = open_v; len_v[j] = 1; } else len_v[j]++;     /* :1772-1779 */
      int open_h = cur[j - 1] + w_open;
      gap_h += w_extend;
      if (open_h > gap_h) { gap_h = open_h; len_h = 1; } else len_h++;                 /* :1784-1793 */
      int down = gap_v[j], right = gap_h, v, t;
      if (diag >= down && diag >= right) { v = diag; t = 0; }                          /* :1797-1810 */
      else if (right >= down) { v = right; t = -len_h; }
      else { v = down; t = len_v[j]; }
      cur[j] = v > -100000000 ? v : -100000000; /* MATRIX_MIN_CUTOFF, never binds */
      b[j] = t;
    }
  }
  free(gap_v); free(len_v);
}

static int iabs(int x) { return x < 0 ? -x : x; }

void orc_sw_endcell(const int* sw, int refLen, int altLen, int strategy, int* p1o, int* p2o, int* scoreo,
                    int* seglen) {
  const int m = altLen + 1;
  int p1 = 0, p2 = 0, best = (int)0x80000000, seg = 0;
  if (strategy == ORC_SW_INDEL) { p1 = refLen; p2 = altLen; }                          /* :2314-2317 */
  else {
    p2 = altLen;
    for (int i = 1; i <= refLen; i++) {                                                /* :2320-2326, ties -> larger i */
      int s = sw[(size_t)i * m + altLen];
      if (s >= best) { p1 = i; best = s; }
    }
    if (strategy != ORC_SW_LEADING_INDEL)
      for (int j = 1; j <= altLen; j++) {                                              /* :2328-2337 */
        int s = sw[(size_t)refLen * m + j];
        if (s > best || (s == best && iabs(refLen - j) < iabs(p1 - p2))) { p1 = refLen; p2 = j; best = s; seg = altLen - j; }
      }
  }
  *p1o = p1; *p2o = p2; *seglen = seg;
  *scoreo = sw[(size_t)p1 * m + p2];
}

typedef struct { int n, cap; int* len; int* st; } elist;
static void push(elist* e, int len, int st) { /* addCigarElement semantics, sw_host.cpp:17-26 */
  if (len <= 0) return;
  if (e->n < e->cap) { e->len[e->n] = len; e->st[e->n] = st; }
  e->n++;
}

int orc_sw_cigar(const int* sw, const int* bt, int refLen, int altLen, int strategy, int max_el, int* cig_len,
                 int* cig_state, int* alignment_offset) {
  const int m = altLen + 1;
  int p1, p2, score, seg;
  orc_sw_endcell(sw, refLen, altLen, strategy, &p1, &p2, &score, &seg);
  elist e = {0, max_el, cig_len, cig_state};
  if (seg > 0 && strategy == ORC_SW_SOFTCLIP) { push(&e, seg, ST_S); seg = 0; }        /* :2342-2345 */
  int state = ST_M;
  do {                                                                                 /* :2351-2377 */
    int t = bt[(size_t)p1 * m + p2], ns, step = 1;
    if (t > 0) { ns = ST_D; step = t; } else if (t < 0) { ns = ST_I; step = -t; } else ns = ST_M;
    if (ns == ST_M) { p1--; p2--; } else if (ns == ST_I) p2 -= step; else p1 -= step;
    if (ns == state) seg += step;
    else { push(&e, seg, state); seg = step; state = ns; }
  } while (p1 > 0 && p2 > 0);
  if (strategy == ORC_SW_SOFTCLIP) {                                                   /* :2379-2401 */
    push(&e, seg, state);
    if (p2 > 0) push(&e, p2, ST_S);
    *alignment_offset = p1;
  } else if (strategy == ORC_SW_IGNORE) {
    push(&e, seg + p2, state);
    *alignment_offset = p1 - p2;
  } else {
    push(&e, seg, state);
    if (p1 > 0) push(&e, p1, ST_D); else if (p2 > 0) push(&e, p2, ST_I);
    *alignment_offset = 0;
  }
  if (e.n <= 0) return -1;
  int k = e.n < max_el ? e.n : max_el;
  for (int a = 0, b = k - 1; a < b; a++, b--) {                                        /* :2408-2417 reverse */
    int t = cig_len[a]; cig_len[a] = cig_len[b]; cig_len[b] = t;
    t = cig_state[a]; cig_state[a] = cig_state[b]; cig_state[b] = t;
  }
  return e.n;
}

int orc_sw_pair(const char* ref, const char* alt, int refLen, int altLen, int strategy, int w_match, int w_mismatch,
                int w_open, int w_extend, int* score, int* p1, int* p2, int max_el, int* cig_len, int* cig_state,
                int* alignment_offset) {
  size_t cells = (size_t)(refLen + 1) * (altLen + 1);
  int* sw = (int*)malloc(sizeof(int) * cells);
  int* bt = (int*)malloc(sizeof(int) * cells);
  orc_sw_fill(ref, alt, refLen, altLen, strategy, w_match, w_mismatch, w_open, w_extend, sw, bt);
  int seg;
  orc_sw_endcell(sw, refLen, altLen, strategy, p1, p2, score, &seg);
  int n = orc_sw_cigar(sw, bt, refLen, altLen, strategy, max_el, cig_len, cig_state, alignment_offset);
  free(sw); free(bt);
  return n;
}

/* Score + end cell with two rolling rows; same recurrence as orc_sw_fill, last column / last row
 * recorded on the fly.  Used for large parity sweeps and as the "port" CPU timing. */
static void score_one(const char* ref, const char* alt, int refLen, int altLen, int strategy, int w_match,
                      int w_mismatch, int w_open, int w_extend, int* score, int* p1o, int* p2o) {
  const int m = altLen + 1;
  int* rows = (int*)malloc(sizeof(int) * (size_t)(3 * m + refLen + 2));
  int *up = rows, *cur = rows + m, *gap_v = rows + 2 * m, *lastcol = rows + 3 * m; /* lastcol[i], i=0..refLen */
  const int prefill = (strategy == ORC_SW_INDEL || strategy == ORC_SW_LEADING_INDEL);
  up[0] = 0;
  for (int j = 1; j < m; j++) up[j] = prefill ? w_open + (j - 1) * w_extend : 0;
  for (int j = 0; j < m; j++) gap_v[j] = NEG_INIT;
  lastcol[0] = up[altLen];
  for (int i = 1; i <= refLen; i++) {
    cur[0] = prefill ? w_open + (i - 1) * w_extend : 0;
    int gap_h = NEG_INIT;
    const char a = ref[i - 1];
    for (int j = 1; j < m; j++) {
      int diag = up[j - 1] + (a == alt[j - 1] ? w_match : w_mismatch);
      int ov = up[j] + w_open, ev = gap_v[j] + w_extend;
      gap_v[j] = ov > ev ? ov : ev;
      int oh = cur[j - 1] + w_open, eh = gap_h + w_extend;
      gap_h = oh > eh ? oh : eh;
      int v = diag;
      if (gap_h > v) v = gap_h;
      if (gap_v[j] > v) v = gap_v[j];
      cur[j] = v;
    }
    lastcol[i] = cur[altLen];
    int* t = up; up = cur; cur = t;
  }
  /* `up` now holds row refLen */
  int p1 = 0, p2 = 0, best = (int)0x80000000;
  if (strategy == ORC_SW_INDEL) { p1 = refLen; p2 = altLen; best = up[altLen]; }
  else {
    p2 = altLen;
    for (int i = 1; i <= refLen; i++) if (lastcol[i] >= best) { p1 = i; best = lastcol[i]; }
    if (strategy != ORC_SW_LEADING_INDEL)
      for (int j = 1; j <= altLen; j++) {
        int s = up[j];
        if (s > best || (s == best && iabs(refLen - j) < iabs(p1 - p2))) { p1 = refLen; p2 = j; best = s; }
      }
  }
  *score = best; *p1o = p1; *p2o = p2;
  free(rows);
}

void orc_sw_score_many(const char* refs, int ref_stride, const int* refLens, const char* alts, int alt_stride,
                       const int* altLens, int n, int strategy, int w_match, int w_mismatch, int w_open,
                       int w_extend, int* score, int* p1, int* p2, int n_threads) {
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads)
  for (int k = 0; k < n; k++)
    score_one(refs + (size_t)k * ref_stride, alts + (size_t)k * alt_stride, refLens[k], altLens[k], strategy, w_match,
              w_mismatch, w_open, w_extend, &score[k], &p1[k], &p2[k]);
}

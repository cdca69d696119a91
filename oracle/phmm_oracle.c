/* TEST INFRASTRUCTURE ONLY -- see oracle.h.  PairHMM forward algorithm, CPU restatement.
 * Follows pairhmm/xlnx/host/baseline_impl.cpp:8-104 (recurrence), Context.h:42-175 (tables) and
 * FalconPairHMM.cpp:69-95 (pair loop + fp64 rescue + log10).  Written from the algorithm's
 * definition, with rolling rows instead of the reference's full matrices. */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <xmmintrin.h>

#define MAXQ 254
#define JTOL 8.0
#define JSTEP 0.0001
#define JSIZE ((int)(JTOL / JSTEP) + 1)

static float  g_ph_f[128], g_m2m_f[ORC_M2M_SIZE], g_init_f, g_l10init_f;
static double g_ph_d[128], g_m2m_d[ORC_M2M_SIZE], g_init_d, g_l10init_d;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

/* Context.h:67-90 in NUMBER=float: every intermediate is a float. */
static float jac_sum_f(const float* jt, float small, float big) {
  if (small > big) { float t = big; big = small; small = t; }
  float diff = big - small;
  if (diff >= (float)JTOL) return big;
  float x = diff * (float)(1.0 / JSTEP);
  int ind = (x > 0.0f) ? (int)(x + 0.5f) : (int)(x - 0.5f);
  return big + jt[ind];
}
static double jac_sum_d(const double* jt, double small, double big) {
  if (small > big) { double t = big; big = small; small = t; }
  double diff = big - small;
  if (diff >= JTOL) return big;
  double x = diff * (1.0 / JSTEP);
  int ind = (x > 0.0) ? (int)(x + 0.5) : (int)(x - 0.5);
  return big + jt[ind];
}

static void init_tables(void) {
  /* Context.h:42-47: table entries computed in double, stored as NUMBER */
  float* jf = (float*)malloc(sizeof(float) * JSIZE);
  double* jd = (double*)malloc(sizeof(double) * JSIZE);
  for (int k = 0; k < JSIZE; k++) {
    double v = log10(1.0 + pow(10.0, -((double)k) * JSTEP));
    jd[k] = v; jf[k] = (float)v;
  }
  /* Context.h:50-61: triangular matchToMatch table, offset(i) = i(i+1)/2 */
  const double inv_ln10 = 1.0 / log(10);
  for (int i = 0, off = 0; i <= MAXQ; off += ++i)
    for (int j = 0; j <= i; j++) {
      double sf = jac_sum_f(jf, (float)(-0.1 * i), (float)(-0.1 * j));
      double sd = jac_sum_d(jd, -0.1 * i, -0.1 * j);
      double pf = pow(10, sf), pd = pow(10, sd);
      g_m2m_f[off + j] = (float)pow(10, log1p(-(pf < 1.0 ? pf : 1.0)) * inv_ln10);
      g_m2m_d[off + j] = pow(10, log1p(-(pd < 1.0 ? pd : 1.0)) * inv_ln10);
    }
  for (int x = 0; x < 128; x++) {
    g_ph_f[x] = powf(10.f, -((float)x) / 10.f);   /* Context.h:145-147 */
    g_ph_d[x] = pow(10.0, -((double)x) / 10.0);   /* Context.h:105-107 */
  }
  g_init_f = ldexpf(1.f, 120); g_l10init_f = log10f(g_init_f);   /* Context.h:149-150 */
  g_init_d = ldexp(1.0, 1020); g_l10init_d = log10(g_init_d);    /* Context.h:109-110 */
  free(jf); free(jd);
}
static void ensure(void) { pthread_once(&g_once, init_tables); }

void orc_phmm_tables_f(float* ph, float* m2m, float* ic, float* l10) {
  ensure(); memcpy(ph, g_ph_f, sizeof g_ph_f); memcpy(m2m, g_m2m_f, sizeof g_m2m_f); *ic = g_init_f; *l10 = g_l10init_f;
}
void orc_phmm_tables_d(double* ph, double* m2m, double* ic, double* l10) {
  ensure(); memcpy(ph, g_ph_d, sizeof g_ph_d); memcpy(m2m, g_m2m_d, sizeof g_m2m_d); *ic = g_init_d; *l10 = g_l10init_d;
}

static inline int tri(int a, int b) { /* Context.h:163-174: index by (max,min) */
  int lo = a < b ? a : b, hi = a < b ? b : a;
  return ((hi * (hi + 1)) >> 1) + lo;
}

/* The forward pass is written once as a macro body so that the float and double versions are the
 * same text.  Row r keeps three rolling arrays over columns 0..H:
 *   M[r][c] = dist(r,c) * ((M[r-1][c-1]*pMM + X[r-1][c-1]*pGM) + Y[r-1][c-1]*pGM)   baseline_impl.cpp:84
 *   X[r][c] = M[r-1][c]*pMX + X[r-1][c]*pXX                                           :85
 *   Y[r][c] = M[r][c-1]*pMY + Y[r][c-1]*pYY                                           :86          */
#define FORWARD_BODY(T, PH, M2M, INIT, ONE, THREE, ZERO)                                               \
  ensure();                                                                                            \
  int R = rslen, H = haplen;                                                                           \
  T* buf = (T*)malloc(sizeof(T) * 6 * (size_t)(H + 1));                                                \
  T *Mp = buf, *Xp = Mp + H + 1, *Yp = Xp + H + 1, *Mc = Yp + H + 1, *Xc = Mc + H + 1, *Yc = Xc + H + 1; \
  for (int c = 0; c <= H; c++) { Mp[c] = ZERO; Xp[c] = ZERO; Yp[c] = INIT / (T)H; } /* :60-64 */        \
  for (int r = 1; r <= R; r++) {                                                                       \
    int qi_ = qi[r - 1] & 127, qd_ = qd[r - 1] & 127, qc_ = qc[r - 1] & 127, qq_ = q[r - 1] & 127;      \
    T pMM = M2M[tri(qi_, qd_)], pGM = ONE - PH[qc_], pMX = PH[qi_], pXX = PH[qc_], pMY = PH[qd_],       \
      pYY = PH[qc_];                                                               /* :50-59 */        \
    T dmis = PH[qq_] / THREE, dmat = ONE - PH[qq_];                                /* :76-83 */        \
    char rb = rs[r - 1];                                                                               \
    Mc[0] = ZERO; Xc[0] = Xp[0] * pXX; Yc[0] = ZERO;                               /* :66-70 */        \
    for (int c = 1; c <= H; c++) {                                                                     \
      char hb = hap[c - 1];                                                                            \
      T dist = (rb == hb || rb == 'N' || hb == 'N') ? dmat : dmis;                                     \
      T a = Mp[c - 1] * pMM + Xp[c - 1] * pGM;                                                         \
      a = a + Yp[c - 1] * pGM;                                                                         \
      Mc[c] = dist * a;                                                                                \
      Xc[c] = Mp[c] * pMX + Xp[c] * pXX;                                                               \
      Yc[c] = Mc[c - 1] * pMY + Yc[c - 1] * pYY;                                                       \
    }                                                                                                  \
    T* t;                                                                                              \
    t = Mp; Mp = Mc; Mc = t; t = Xp; Xp = Xc; Xc = t; t = Yp; Yp = Yc; Yc = t;                         \
  }                                                                                                    \
  T res;                                                                                               \
  if (sum_order == 0) {                                                                                \
    res = ZERO;                                                                                        \
    for (int c = 0; c <= H; c++) res += Mp[c] + Xp[c];                             /* :90-92 */        \
  } else {                                                                                             \
    T sm = ZERO, sx = ZERO;                                                                            \
    for (int c = 1; c <= H; c++) { sm += Mp[c]; sx += Xp[c]; }                                         \
    res = sm + sx;                                                                                     \
  }                                                                                                    \
  free(buf);                                                                                           \
  return res;


static float fwd_f(int rslen, int haplen, const char* rs, const char* q, const char* qi, const char* qd,
                   const char* qc, const char* hap, int sum_order) {
  FORWARD_BODY(float, g_ph_f, g_m2m_f, g_init_f, 1.0f, 3.0f, 0.0f)
}
static double fwd_d(int rslen, int haplen, const char* rs, const char* q, const char* qi, const char* qd,
                    const char* qc, const char* hap, int sum_order) {
  FORWARD_BODY(double, g_ph_d, g_m2m_d, g_init_d, 1.0, 3.0, 0.0)
}

/* The reference runs with SSE flush-to-zero on (FalconPairHMM.cpp:850, host/main.cpp:248). */
#define FTZ_BEGIN unsigned csr_ = _mm_getcsr(); _mm_setcsr(csr_ | 0x8000u)
#define FTZ_END _mm_setcsr(csr_)

float orc_phmm_forward_f32(int rslen, int haplen, const char* rs, const char* q, const char* qi, const char* qd,
                           const char* qc, const char* hap, int sum_order) {
  FTZ_BEGIN; float r = fwd_f(rslen, haplen, rs, q, qi, qd, qc, hap, sum_order); FTZ_END; return r;
}
double orc_phmm_forward_f64(int rslen, int haplen, const char* rs, const char* q, const char* qi, const char* qd,
                            const char* qc, const char* hap, int sum_order) {
  FTZ_BEGIN; double r = fwd_d(rslen, haplen, rs, q, qi, qd, qc, hap, sum_order); FTZ_END; return r;
}

/* Same recurrence in the arithmetic of the GPU fast mode (one rounding per fma); Yt = Y * pGM[r+1] is
 * what the kernel keeps per row, so that the diagonal term is two fmas:
 *   M[r][c]  = dist * fma(M[r-1][c-1], pMM[r], fma(X[r-1][c-1], pGM[r], Yt[r-1][c-1]))
 *   X[r][c]  = fma(M[r-1][c], pMX[r], X[r-1][c] * pXX[r])
 *   Yt[r][c] = fma(M[r][c-1], pMY[r]*pGM[r+1], Yt[r][c-1] * pXX[r])        (pGM[R+1] := 0, Yt[0][c] = (INIT/H)*pGM[1])
 * and the scalar summation order.  Not a reference function: it exists so that tests can separate
 * "GPU differs from its own arithmetic model" (a bug) from "model differs from the reference by
 * rounding" (bounded, checked against the 1e-5 budget). */
float orc_phmm_forward_f32_fma(int rslen, int haplen, const char* rs, const char* q, const char* qi,
                               const char* qd, const char* qc, const char* hap) {
  ensure();
  FTZ_BEGIN;
  int R = rslen, H = haplen;
  float* buf = (float*)malloc(sizeof(float) * 6 * (size_t)(H + 1));
  float *Mp = buf, *Xp = Mp + H + 1, *Yp = Xp + H + 1, *Mc = Yp + H + 1, *Xc = Mc + H + 1, *Yc = Xc + H + 1;
  float g1 = 1.0f - g_ph_f[qc[0] & 127];
  for (int c = 0; c <= H; c++) { Mp[c] = 0.f; Xp[c] = 0.f; Yp[c] = (g_init_f / (float)H) * g1; }
  for (int r = 1; r <= R; r++) {
    int qi_ = qi[r - 1] & 127, qd_ = qd[r - 1] & 127, qc_ = qc[r - 1] & 127, qq_ = q[r - 1] & 127;
    float pMM = g_m2m_f[tri(qi_, qd_)], pGM = 1.0f - g_ph_f[qc_], pMX = g_ph_f[qi_], pXX = g_ph_f[qc_], pMY = g_ph_f[qd_];
    float gnext = r < R ? 1.0f - g_ph_f[qc[r] & 127] : 0.0f;
    float pMYg = pMY * gnext;
    float dmis = g_ph_f[qq_] / 3.0f, dmat = 1.0f - g_ph_f[qq_];
    char rb = rs[r - 1];
    Mc[0] = 0.f; Xc[0] = 0.f; Yc[0] = 0.f;
    for (int c = 1; c <= H; c++) {
      char hb = hap[c - 1];
      float dist = (rb == hb || rb == 'N' || hb == 'N') ? dmat : dmis;
      float a = fmaf(Mp[c - 1], pMM, fmaf(Xp[c - 1], pGM, Yp[c - 1]));
      Mc[c] = dist * a;
      Xc[c] = fmaf(Mp[c], pMX, Xp[c] * pXX);
      Yc[c] = fmaf(Mc[c - 1], pMYg, Yc[c - 1] * pXX);
    }
    float* t;
    t = Mp; Mp = Mc; Mc = t; t = Xp; Xp = Xc; Xc = t; t = Yp; Yp = Yc; Yc = t;
  }
  float res = 0.f;
  for (int c = 1; c <= H; c++) res += Mp[c] + Xp[c];
  free(buf);
  FTZ_END;
  return res;
}


/* The six-operation form of the GPU fast mode: X kept divided by the row's pMX,
 *   Xs[r][c] = fma(Xs[r-1][c], (pXX[r] * pMX[r-1]) / pMX[r], M[r-1][c])          (coefficient 0 for r = 1: X[0][c] = 0)
 *   M[r][c]  = dist * fma(M[r-1][c-1], pMM[r], fma(Xs[r-1][c-1], pGM[r] * pMX[r-1], Yt[r-1][c-1]))
 *   result   = sum_c fma(Xs[R][c], pMX[R], M[R][c])
 * Yt as in orc_phmm_forward_f32_fma.  The kernel only takes this form for reads that pass orc_phmm_x6_eligible (the bound
 * that keeps Xs clear of FLT_MAX, phmm_dev.h).  Not a reference function: the GPU's arithmetic model, as the one above. */
int orc_phmm_x6_eligible(int rslen, const char* qi, const char* qc) {
  ensure();
  double F = 1.0;
  for (int r = 1; r < rslen; r++) {
    double c = (double)g_ph_f[qc[r] & 127] * (double)g_ph_f[qi[r - 1] & 127] / (double)g_ph_f[qi[r] & 127];
    F = 1.0 + c * F;
    if (!(F <= 32.0)) return 0;
  }
  return 1;
}
float orc_phmm_forward_f32_fma6(int rslen, int haplen, const char* rs, const char* q, const char* qi,
                                const char* qd, const char* qc, const char* hap) {
  ensure();
  FTZ_BEGIN;
  int R = rslen, H = haplen;
  float* buf = (float*)malloc(sizeof(float) * 6 * (size_t)(H + 1));
  float *Mp = buf, *Xp = Mp + H + 1, *Yp = Xp + H + 1, *Mc = Yp + H + 1, *Xc = Mc + H + 1, *Yc = Xc + H + 1;
  float g1 = 1.0f - g_ph_f[qc[0] & 127];
  for (int c = 0; c <= H; c++) { Mp[c] = 0.f; Xp[c] = 0.f; Yp[c] = (g_init_f / (float)H) * g1; }
  float pMXprev = 0.f;
  for (int r = 1; r <= R; r++) {
    int qi_ = qi[r - 1] & 127, qd_ = qd[r - 1] & 127, qc_ = qc[r - 1] & 127, qq_ = q[r - 1] & 127;
    float pMM = g_m2m_f[tri(qi_, qd_)], pGM = 1.0f - g_ph_f[qc_], pMX = g_ph_f[qi_], pXX = g_ph_f[qc_], pMY = g_ph_f[qd_];
    float gnext = r < R ? 1.0f - g_ph_f[qc[r] & 127] : 0.0f;
    float pMYg = pMY * gnext;
    float cXt = pGM * pMXprev;                       /* what the diagonal term multiplies Xs of the row above with */
    float cXc = (pXX * pMXprev) / pMX;               /* chain coefficient; 0 for the first row */
    float dmis = g_ph_f[qq_] / 3.0f, dmat = 1.0f - g_ph_f[qq_];
    char rb = rs[r - 1];
    Mc[0] = 0.f; Xc[0] = 0.f; Yc[0] = 0.f;
    for (int c = 1; c <= H; c++) {
      char hb = hap[c - 1];
      float dist = (rb == hb || rb == 'N' || hb == 'N') ? dmat : dmis;
      float a = fmaf(Mp[c - 1], pMM, fmaf(Xp[c - 1], cXt, Yp[c - 1]));
      Mc[c] = dist * a;
      Xc[c] = fmaf(Xp[c], cXc, Mp[c]);
      Yc[c] = fmaf(Mc[c - 1], pMYg, Yc[c - 1] * pXX);
    }
    float* t;
    t = Mp; Mp = Mc; Mc = t; t = Xp; Xp = Xc; Xc = t; t = Yp; Yp = Yc; Yc = t;
    pMXprev = pMX;
  }
  float res = 0.f;
  for (int c = 1; c <= H; c++) res += fmaf(Xp[c], pMXprev, Mp[c]);
  free(buf);
  FTZ_END;
  return res;
}

/* The five-operation form of the GPU fast mode: on top of Xs = X / pMX[r], Y is kept divided by the row's pMY and the diagonal
 * term divided by the consumer row's pMM, which moves into that row's emission values:
 *   Ys[r][c] = fma(Ys[r][c-1], pYY[r], M[r][c-1])                                 (Ys[0][c] = INIT/H: row 0 is scaled by 1)
 *   T[r][c]  = fma(Ys[r][c], b[r], fma(Xs[r][c], a[r], M[r][c]))                 a[r] = (pMX[r] * pGM[r+1]) / pMM[r+1],
 *                                                                                 b[r] = (sY[r]  * pGM[r+1]) / pMM[r+1], sY[0] = 1, sY[r] = pMY[r]
 *   M[r][c]  = (dist(r,c) * pMM[r]) * T[r-1][c-1]
 *   Xs and the result as in orc_phmm_forward_f32_fma6.
 * Restates baseline_impl.cpp:84-86 / avx-pairhmm-template.h:183-198 with 5 instead of 12 operations per cell.  The kernel only takes
 * this form for reads that pass orc_phmm_x5_eligible: the x6 bound, every pMM >= 1/16 (a, b and T stay within a factor 16 of the
 * unscaled values) and every pYY <= 31/32 (Ys <= 32 max M).  Not a reference function: the GPU's arithmetic model. */
int orc_phmm_x5_eligible(int rslen, const char* qi, const char* qd, const char* qc) {
  ensure();
  if (!orc_phmm_x6_eligible(rslen, qi, qc)) return 0;
  for (int r = 0; r < rslen; r++) {
    if (!(g_m2m_f[tri(qi[r] & 127, qd[r] & 127)] >= 0.0625f)) return 0;
    if (!(g_ph_f[qc[r] & 127] <= 0.96875f)) return 0;
  }
  return 1;
}
float orc_phmm_forward_f32_fma5(int rslen, int haplen, const char* rs, const char* q, const char* qi,
                                const char* qd, const char* qc, const char* hap) {
  ensure();
  FTZ_BEGIN;
  int R = rslen, H = haplen;
  /* T of the row above (previous column order), Xs of the row above, this row's M / Xs / T */
  float* buf = (float*)malloc(sizeof(float) * 6 * (size_t)(H + 1));
  float *Mp = buf, *Xp = Mp + H + 1, *Tp = Xp + H + 1, *Mc = Tp + H + 1, *Xc = Mc + H + 1, *Tc = Xc + H + 1;
  /* row 0: M = X = 0, Ys = INIT/H, T[0][c] = fma(Ys, b[0], fma(0, a[0], 0)) with b[0] = (1 * pGM[1]) / pMM[1] */
  {
    float g1 = 1.0f - g_ph_f[qc[0] & 127], m1 = g_m2m_f[tri(qi[0] & 127, qd[0] & 127)];
    float b0 = (1.0f * g1) / m1, y0 = g_init_f / (float)H;
    for (int c = 0; c <= H; c++) { Mp[c] = 0.f; Xp[c] = 0.f; Tp[c] = fmaf(y0, b0, fmaf(0.f, 0.f, 0.f)); }
  }
  float pMXprev = 0.f;
  for (int r = 1; r <= R; r++) {
    int qi_ = qi[r - 1] & 127, qd_ = qd[r - 1] & 127, qc_ = qc[r - 1] & 127, qq_ = q[r - 1] & 127;
    float pMM = g_m2m_f[tri(qi_, qd_)], pMX = g_ph_f[qi_], pYY = g_ph_f[qc_], pMY = g_ph_f[qd_];
    float gnext = 0.f, mnext = 0.f;
    if (r < R) { gnext = 1.0f - g_ph_f[qc[r] & 127]; mnext = g_m2m_f[tri(qi[r] & 127, qd[r] & 127)]; }
    float a = r < R ? (pMX * gnext) / mnext : 0.f, b = r < R ? (pMY * gnext) / mnext : 0.f;
    float cXc = (pYY * pMXprev) / pMX;               /* chain coefficient of Xs; 0 for the first row (pXX == pYY) */
    float dmis = (g_ph_f[qq_] / 3.0f) * pMM, dmat = (1.0f - g_ph_f[qq_]) * pMM;
    char rb = rs[r - 1];
    Mc[0] = 0.f; Xc[0] = 0.f;
    float ys = 0.f;                                  /* Ys[r][0] = 0 */
    Tc[0] = 0.f;                                     /* never used: column 0 of the row below is the border */
    for (int c = 1; c <= H; c++) {
      char hb = hap[c - 1];
      float dist = (rb == hb || rb == 'N' || hb == 'N') ? dmat : dmis;
      Mc[c] = dist * Tp[c - 1];
      Xc[c] = fmaf(Xp[c], cXc, Mp[c]);
      ys = fmaf(ys, pYY, Mc[c - 1]);                 /* Ys[r][c] from Ys[r][c-1] and M[r][c-1] */
      Tc[c] = fmaf(ys, b, fmaf(Xc[c], a, Mc[c]));
    }
    /* T[r][0] as the kernel computes it at a bubble: M = Xs = Ys = 0 -> 0; Tp[c-1] with c = 1 reads it */
    float* t;
    t = Mp; Mp = Mc; Mc = t; t = Xp; Xp = Xc; Xc = t; t = Tp; Tp = Tc; Tc = t;
    pMXprev = pMX;
  }
  float res = 0.f;
  for (int c = 1; c <= H; c++) res += fmaf(Xp[c], pMXprev, Mp[c]);
  free(buf);
  FTZ_END;
  return res;
}

double orc_phmm_finish(float raw, int rslen, int haplen, const char* rs, const char* q, const char* qi,
                       const char* qd, const char* qc, const char* hap, int* rescued) {
  ensure();
  if (raw < 1e-28f) { /* MIN_ACCEPTED, host_type.h:21; FalconPairHMM.cpp:84-87 */
    if (rescued) (*rescued)++;
    double d = orc_phmm_forward_f64(rslen, haplen, rs, q, qi, qd, qc, hap, 0);
    return log10(d) - g_l10init_d;
  }
  return (double)(log10f(raw) - g_l10init_f); /* FalconPairHMM.cpp:89 */
}

int orc_phmm_region(int n_reads, const int* rlen, const char* const* rs, const char* const* q, const char* const* qi,
                    const char* const* qd, const char* const* qc, int n_haps, const int* hlen,
                    const char* const* hap, float* out_raw, double* out_log10, int n_threads) {
  ensure();
  int rescued = 0;
  long total = (long)n_reads * n_haps;
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads) reduction(+ : rescued)
  for (long k = 0; k < total; k++) {
    int a = (int)(k / n_haps), b = (int)(k % n_haps);
    float f = orc_phmm_forward_f32(rlen[a], hlen[b], rs[a], q[a], qi[a], qd[a], qc[a], hap[b], 0);
    int resc = 0;
    double r = orc_phmm_finish(f, rlen[a], hlen[b], rs[a], q[a], qi[a], qd[a], qc[a], hap[b], &resc);
    rescued += resc;
    if (out_raw) out_raw[k] = f;
    if (out_log10) out_log10[k] = r;
  }
  return rescued;
}

/* P8 wire format, PairHMMHostInterface.cpp:175-206 */
int64_t orc_phmm_serialize_reads(void* buf, int n, const int* len, const char* const* b, const char* const* q,
                                 const char* const* qi, const char* const* qd, const char* const* qc) {
  char* p = (char*)buf; int32_t v = n;
  memcpy(p, &v, 4); p += 4;
  for (int k = 0; k < n; k++) {
    v = len[k]; memcpy(p, &v, 4); p += 4;
    const char* f[5] = {b[k], q[k], qi[k], qd[k], qc[k]};
    for (int j = 0; j < 5; j++) { memcpy(p, f[j], (size_t)len[k]); p += len[k]; }
  }
  return p - (char*)buf;
}
int64_t orc_phmm_serialize_haps(void* buf, int n, const int* len, const char* const* b) {
  char* p = (char*)buf; int32_t v = n;
  memcpy(p, &v, 4); p += 4;
  for (int k = 0; k < n; k++) { v = len[k]; memcpy(p, &v, 4); p += 4; memcpy(p, b[k], (size_t)len[k]); p += len[k]; }
  return p - (char*)buf;
}

// PairHMM forward algorithm for gfx950 (MI355X), hand-written HIP: kernel templates, shared by the translation units that
// instantiate them (phmm_kernel_fast.hip, phmm_kernel_strict.hip, phmm_kernel_f64.hip -- three, so that they build in parallel).
//
// What it computes: compute_full_prob_baseline<T> of the reference
// (pairhmm/xlnx/host/baseline_impl.cpp:8-104; AVX twin avx-pairhmm-template.h:210-346), i.e. for one
// read (rows r = 1..R) against one haplotype (columns c = 1..H)
//     M[r][c] = dist(r,c) * ((M[r-1][c-1]*pMM[r] + X[r-1][c-1]*pGM[r]) + Y[r-1][c-1]*pGM[r])
//     X[r][c] = M[r-1][c]*pMX[r] + X[r-1][c]*pXX[r]
//     Y[r][c] = M[r][c-1]*pMY[r] + Y[r][c-1]*pYY[r]          (pYY[r] == pXX[r], baseline_impl.cpp:56,58)
//     result  = sum_c (M[R][c] + X[R][c]),   Y[0][c] = INIT/H, everything else on the border 0.
//
// How it is mapped (a design of its own, neither the FPGA PE array nor the AVX stripes):
//   * one wavefront = four reads of 16 lanes (reads <= 103 bp: eight reads of 8 lanes; > 255 bp: two reads on 32 lanes
//     each, > 511 bp: one read on all 64);
//     read g lives in DPP row g (16 lanes); lane l of the row owns K
//     consecutive read rows in registers (K = ceil((R+1)/16) is a template parameter, so all row
//     state is register-resident and indexed at compile time).  Rows are right-aligned: the last
//     read row is always (lane 15, k = K-1); the rows in front are clones of "row 0"
//     (M = X = 0, Y = INIT/H), which makes the top border fall out of the same recurrence.
//   * the reads of a wavefront sweep the SAME haplotype stream, one column per step, skewed one column per
//     lane (lane l is at column t-l), so the inter-lane hand-off is exactly one DPP row_shr:1 of
//     two values per step (the pre-multiplied diagonal term and the X of the row below) - no LDS,
//     no bpermute.  Inside a lane the K rows are chained through registers.
//   * all haplotypes of the job are concatenated into one stream in LDS (one "bubble" entry in
//     front of each haplotype plays column 0), so the 15-step pipeline fill/drain is paid once per
//     job, not once per pair.
//   * the emission probability dist(r,c) takes one of two per-row values depending on the hap base.
//     v_cmp/v_cndmask run at about 0.6x the rate of fp32 mul/fma on gfx950 (tools/ubench.hip), so the
//     selection is done by the LDS instead: a per-wave table T[base][row quad][lane] of 16-byte
//     vectors holds dist for every (hap base, row); a stream entry is the byte offset of its base's
//     slab, and one ds_read_b128 per 4 rows (2 rows in fp64) fetches the step's values, conflict-free
//     (slabs are multiples of 1 KiB apart, so the bank is decided by the lane alone).  Both LDS reads
//     are software-pipelined one step ahead of their use.
//   * no MFMA: the recurrence is a chain of fp32 mul/fma along the anti-diagonal, not a contraction.
#pragma once
#include "phmm_dev.h"
#include <utility>

namespace accg {
namespace {

enum { CH_A = 0, CH_C = 1, CH_G = 2, CH_T = 3, CH_N = 4 };
__device__ __forceinline__ int char_index(uint8_t b) {   // bases are validated on the host (ACCG_ERR_BAD_BASE)
  return b == 'A' ? CH_A : b == 'C' ? CH_C : b == 'G' ? CH_G : b == 'T' ? CH_T : CH_N;
}

// lane l <- lane l-1 inside a group of LPP lanes; the first lane of a group receives 0.
// LPP = 16: DPP row_shr:1.  LPP = 32 / 64: wave_shr:1 (lane 32's source is lane 31, the last lane of the
// other group, whose a_out / x_out are 0 by construction when LPP = 32 - see the lane-constant setup).
// LPP = 8: row_shr:1 again; lane 8's source is lane 7, the last lane of the other group of the row, which hands
// over zeros for the same reason.
template <int LPP>
__device__ __forceinline__ int shr1_bits(int v) {
  return LPP <= 16 ? __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true) : __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true);
}
template <int LPP>
__device__ __forceinline__ float group_shr1(float v) { return __builtin_bit_cast(float, shr1_bits<LPP>(__builtin_bit_cast(int, v))); }
template <int LPP>
__device__ __forceinline__ double group_shr1(double v) {
  long long b = __builtin_bit_cast(long long, v);
  int lo = shr1_bits<LPP>((int)b), hi = shr1_bits<LPP>((int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// diagonal term handed to the row below:  M*pMM + X*pGM + Y*pGM.
// Fast mode keeps Y pre-multiplied by the consumer row's pGM (Yt = Y*pGM', updated with the coefficient
// pMY*pGM' that is folded into a register once per job), which makes the term two fmas.
template <bool STRICT, typename T>
__device__ __forceinline__ T diag_term(T m, T x, T y, T pmm, T pgm) {
  if (STRICT) { T a = m * pmm + x * pgm; return a + y * pgm; }   // baseline_impl.cpp:84 order
  return fma_(m, pmm, fma_(x, pgm, y));
}
template <bool STRICT, typename T>
__device__ __forceinline__ T mul_add2(T a, T b, T c, T d) {      // a*b + c*d
  if (STRICT) return a * b + c * d;                              // baseline_impl.cpp:85-86
  return fma_(a, b, c * d);
}

template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef float type __attribute__((ext_vector_type(4))); enum { N = 4 }; };
template <> struct Vec16<double> { typedef double type __attribute__((ext_vector_type(2))); enum { N = 2 }; };

template <typename T, int K>
struct Rows {
  T M[K], X[K], Y[K];     // fast mode: Y[k] holds Y * (pGM of the row below)
  T pMM[K], pGM[K], pMX[K], pXX[K], pMY[K];   // fast mode: pMY[k] holds pMY * (pGM of the row below)
  T nMM, nGM, nMX, nXX;   // row-0 coefficients of the lane to the right (lane 15: nMX = nXX = 1, so x_out = M + X)
  T a_out, x_out;         // what this lane hands to the right at the next step
  T acc;                  // running sum of M+X of the last read row (meaningful in lane 15)
  T xl, gclone;           // six-operation form only: pMX of the lane's last row; pGM of the first read row behind this lane's clones
  int npad;               // local rows k < npad are clones of row 0 (they form a prefix of the lane's rows)
};

// dist values of one step for this lane's K rows: QT vector reads, slab offset `off` (bytes).
template <typename T, int K>
__device__ __forceinline__ void load_dist(const unsigned char* tab_lane, unsigned off, T (&d)[K]) {
  typedef typename Vec16<T>::type V;
  constexpr int N = Vec16<T>::N, QT = (K + N - 1) / N;
#pragma unroll
  for (int q = 0; q < QT; q++) {
    V v = *reinterpret_cast<const V*>(tab_lane + off + q * 1024);
#pragma unroll
    for (int e = 0; e < N; e++)
      if (q * N + e < K) d[q * N + e] = v[e];
  }
}

// One column for every lane.  d[k] = dist of local row k against this lane's column.
template <bool STRICT, int LPP, typename T, int K>
__device__ __forceinline__ void column(Rows<T, K>& s, const T (&d)[K], bool carry_in = false, T ca = T(0), T cx = T(0)) {
  T a_in = group_shr1<LPP>(s.a_out);
  T x_in = group_shr1<LPP>(s.x_out);
  if (carry_in) { a_in = ca; x_in = cx; }     // striped reads: lane 0 continues the row above, handed over by the previous stripe
  // pass 1: everything that reads the previous column's state
  s.a_out = diag_term<STRICT>(s.M[K - 1], s.X[K - 1], s.Y[K - 1], s.nMM, s.nGM);
  T Mn[K], Yn[K];
#pragma unroll
  for (int k = K - 1; k >= 0; k--) {
    T a = (k == 0) ? a_in : diag_term<STRICT>(s.M[k - 1], s.X[k - 1], s.Y[k - 1], s.pMM[k], s.pGM[k]);
    Mn[k] = d[k] * a;
    Yn[k] = mul_add2<STRICT>(s.M[k], s.pMY[k], s.Y[k], s.pXX[k]);
  }
  // pass 2: the X chain runs down the rows of the *current* column
  T xk = x_in;
#pragma unroll
  for (int k = 0; k < K; k++) {
    if (k > 0) xk = mul_add2<STRICT>(Mn[k - 1], s.pMX[k], xk, s.pXX[k]);
    s.X[k] = xk;
  }
#pragma unroll
  for (int k = 0; k < K; k++) { s.M[k] = Mn[k]; s.Y[k] = Yn[k]; }
  s.x_out = mul_add2<STRICT>(s.M[K - 1], s.nMX, s.X[K - 1], s.nXX);
  // last lane of a group: M + X of the last read row (baseline_impl.cpp:91).  With one or four groups per wave it is
  // x_out itself (nMX = nXX = 1, nobody consumes it); with two groups lane 31's x_out must stay 0 for lane 32.
  s.acc = s.acc + ((LPP == 32 || LPP == 8) ? s.M[K - 1] + s.X[K - 1] : s.x_out);
}

// ---- fp32 fast mode: the same column, written row by row in gfx950 assembly -----------------------------------
// Why: left to the compiler the column above needs 13 K + 23 registers (interleaved rows, two copies of the dist values,
// Mn / Yn temporaries): 192 at K = 13 = two waves per SIMD, and a wave alone issues at most one VALU instruction per
// four cycles, so two of them cannot cover each other's LDS waits (measured 0.58 of the issue roof).  One ascending pass,
// every state register updated in place, needs 9 K + ~25:
//   row k:  tn   = fma(M[k], pMM[k+1], fma(X[k], pGM[k+1], Y[k]))     the diagonal term row k+1 will multiply (old state)
//           Y[k] = fma(M[k], pMY'[k], Y[k] * pXX[k])                   old M[k]
//           X[k] = fma(M[k-1], pMX[k], X[k-1] * pXX[k])                new state of row k-1 (row 0: x_in straight from the DPP)
//           M[k] = d[k] * tc                                           tc = the tn of row k-1 (row 0: a_in, DPP fused into the mul)
// The arithmetic (operands and their order in every fma) is exactly that of column<false>: results are bit-identical.
// The dist values of the next step are re-loaded into the same registers quad by quad as soon as a quad's rows are done.
// LDS traffic of the fp32 fast sweep, issued and awaited by hand.  Per step and wave: one byte of the haplotype stream (the
// slab offset two steps ahead) and the QT quads of dist values of the next step, each re-loaded into its own registers as
// soon as its rows are done.  The compiler's s_waitcnt insertion merges the loop's back edge with the bubble block and ends up
// waiting for the youngest load at the top of every step; counted by hand every wait is for a load issued a whole step earlier.
// Order per step: U (stream byte), L0 .. L(QT-1); LDS operations return in order, so before any of them is consumed exactly
// QT younger ones may still be in flight: s_waitcnt lgkmcnt(QT) everywhere.  (No scalar memory load may sit in this loop: SMEM
// shares the counter and returns out of order.  tools/check_phmm_asm.py looks for one in the built code object.)
template <int N> __device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N)); }
// The same as an instruction the compiler knows (vmcnt 63, expcnt 7, lgkmcnt N in the gfx9 encoding).  Why it matters: the
// hazard recognizer gives an inline-asm statement zero wait states and assumes the worst of what it writes (gfx950's forwarding
// hazard of sub-dword writes), so between two asm statements of which the second reads a register of the first it puts an
// s_nop -- one issue slot, and with two wavefronts per SIMD issue slots are what the sweep is short of.  A real instruction in
// between (this one, at every quad boundary) makes that s_nop unnecessary; rows are issued two per asm statement for the same
// reason (12 -> 3 s_nop per column at K = 13).
template <int N> __device__ __forceinline__ void lgkm_wait_visible() { __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8)); }
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)p; }   // low half of a flat LDS address = the LDS offset

template <int REM> struct TailQuad;        // the last quad of a K that is not a multiple of four is loaded with its exact width
template <> struct TailQuad<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct TailQuad<3> { typedef float type __attribute__((ext_vector_type(3))); };
template <> struct TailQuad<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct TailQuad<1> { typedef float type; };

template <int REM, int OFF>
__device__ __forceinline__ void lds_load_quad(typename TailQuad<REM>::type& v, unsigned addr) {
  if constexpr (REM == 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  else if constexpr (REM == 3) asm volatile("ds_read_b96 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  else if constexpr (REM == 2) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int REM> __device__ __forceinline__ float quad_elem(const typename TailQuad<REM>::type& v, int e) { if constexpr (REM == 1) return v; else return v[e]; }

// dist values of one step for the K rows of a lane: full quads + one tail of K % 4 values
template <int K>
struct DistRegs {
  static constexpr int FULL = K / 4, REM = K % 4, QT = (K + 3) / 4;
  typename TailQuad<4>::type q[FULL > 0 ? FULL : 1];
  typename TailQuad<REM ? REM : 4>::type tail;
  __device__ __forceinline__ float get(int k) const { return k / 4 < FULL ? q[k / 4][k % 4] : quad_elem<REM ? REM : 4>(tail, k % 4); }
  // addr = slab + lane * 16; the tail's rows sit at FULL * 1024 + lane * phmm_tail_stride(K): tail_adj = lane * (stride - 16)
  // A use of every register (no instruction): behind the sweep loop, so that the registers count as live on the loop's exit path
  // too -- the loads of the step that is never run are still in flight there, and a register the compiler took for dead would be
  // handed to other code while its load is on the way.
  __device__ __forceinline__ void keep() const {
#pragma unroll
    for (int i = 0; i < (FULL > 0 ? FULL : 0); i++) asm volatile("" ::"v"(q[i]));
    if constexpr (REM != 0) asm volatile("" ::"v"(tail));
  }
  template <int Q> __device__ __forceinline__ void load(unsigned addr, unsigned tail_adj) {
    if constexpr (Q < FULL) lds_load_quad<4, Q * 1024>(q[Q], addr);
    else if constexpr (REM == 3 || REM == 0) lds_load_quad<REM ? REM : 4, Q * 1024>(tail, addr);
    else lds_load_quad<REM, Q * 1024>(tail, addr + tail_adj);
  }
};

template <int K, int... Q>
__device__ __forceinline__ void load_all_quads(DistRegs<K>& dq, unsigned addr, unsigned tail_adj, std::integer_sequence<int, Q...>) { (dq.template load<Q>(addr, tail_adj), ...); }

#define ACCG_DPP_ROW "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define ACCG_DPP_WAVE "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"

template <int LPP, int K, bool X6, int Q>
__device__ __forceinline__ void column_rows(Rows<float, K>& s, DistRegs<K>& dq, unsigned addr_next, unsigned tail_adj, float& tc, float& a_new) {
  if constexpr (Q < DistRegs<K>::QT) {
    // this quad's values (loaded during the previous step) have landed: awaited at the top of the column for quad 0, and for
    // every later quad BEFORE the reload of the quad in front of it is issued (one load less in flight at that point: QT - 1), so
    // that the wait -- an instruction the compiler knows -- stands between the rows' asm statement and the load's
    // (quad 0: awaited together with the stream byte at the top of the step, in the sweep loop)
    constexpr int K0 = 4 * Q, K1 = (4 * Q + 4 < K) ? 4 * Q + 4 : K;     // rows of this quad
    if constexpr (X6 && K1 - K0 == 4) {
      // a whole quad of rows in ONE asm statement (see lgkm_wait_visible: every boundary between two asm statements costs an issue
      // slot, an s_nop or a wait; a quad boundary needs its wait anyway)
#define ACCG_ROW6(n, xp, mp, tcin)                                                       \
  "v_fma_f32 %[t" #n "], %[X" #n "], %[gn" #n "], %[Y" #n "]\n\t"                        \
  "v_mul_f32 %[Y" #n "], %[Y" #n "], %[xx" #n "]\n\t"                                    \
  "v_fma_f32 %[X" #n "], %[" xp "], %[mx" #n "], %[" mp "]\n\t"                          \
  "v_fmac_f32 %[t" #n "], %[M" #n "], %[mn" #n "]\n\t"                                   \
  "v_fmac_f32 %[Y" #n "], %[M" #n "], %[my" #n "]\n\t"                                   \
  "v_mul_f32 %[M" #n "], %[d" #n "], %[" tcin "]\n\t"
#define ACCG_ROW0(dpp)                                                                   \
  "v_fma_f32 %[t0], %[X0], %[gn0], %[Y0]\n\t"                                           \
  "v_mul_f32 %[Y0], %[Y0], %[xx0]\n\t"                                                  \
  "v_fmac_f32 %[t0], %[M0], %[mn0]\n\t"                                                 \
  "v_fmac_f32 %[Y0], %[M0], %[my0]\n\t"                                                 \
  "v_mov_b32_dpp %[X0], %[xo] " dpp "\n\t"                                              \
  "v_mul_f32_dpp %[M0], %[ao], %[d0] " dpp "\n\t"
#define ACCG_QUAD_OUT                                                                                                        \
  [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [X0] "+v"(s.X[K0]), [Y0] "+v"(s.Y[K0]), [M0] "+v"(s.M[K0]),   \
      [X1] "+v"(s.X[K0 + 1]), [Y1] "+v"(s.Y[K0 + 1]), [M1] "+v"(s.M[K0 + 1]), [X2] "+v"(s.X[K0 + 2]), [Y2] "+v"(s.Y[K0 + 2]),   \
      [M2] "+v"(s.M[K0 + 2]), [X3] "+v"(s.X[K0 + 3]), [Y3] "+v"(s.Y[K0 + 3]), [M3] "+v"(s.M[K0 + 3])
#define ACCG_QUAD_IN                                                                                                         \
  [gn0] "v"(s.pGM[K0 + 1]), [gn1] "v"(s.pGM[K0 + 2]), [gn2] "v"(s.pGM[K0 + 3]), [gn3] "v"(gn3), [mn0] "v"(s.pMM[K0 + 1]),      \
      [mn1] "v"(s.pMM[K0 + 2]), [mn2] "v"(s.pMM[K0 + 3]), [mn3] "v"(mn3), [xx0] "v"(s.pXX[K0]), [xx1] "v"(s.pXX[K0 + 1]),       \
      [xx2] "v"(s.pXX[K0 + 2]), [xx3] "v"(s.pXX[K0 + 3]), [my0] "v"(s.pMY[K0]), [my1] "v"(s.pMY[K0 + 1]), [my2] "v"(s.pMY[K0 + 2]), \
      [my3] "v"(s.pMY[K0 + 3]), [mx1] "v"(s.pMX[K0 + 1]), [mx2] "v"(s.pMX[K0 + 2]), [mx3] "v"(s.pMX[K0 + 3]), [d0] "v"(dq.get(K0)), \
      [d1] "v"(dq.get(K0 + 1)), [d2] "v"(dq.get(K0 + 2)), [d3] "v"(dq.get(K0 + 3))
      const float gn3 = (K0 + 4 < K) ? s.pGM[K0 + 4] : s.nGM, mn3 = (K0 + 4 < K) ? s.pMM[K0 + 4] : s.nMM;
      float t0, t1, t2, t3;
      if constexpr (Q == 0) {
        if (LPP <= 16)
          asm volatile(ACCG_ROW0(ACCG_DPP_ROW) ACCG_ROW6(1, "X0", "M0", "t0") ACCG_ROW6(2, "X1", "M1", "t1") ACCG_ROW6(3, "X2", "M2", "t2")
                       : ACCG_QUAD_OUT : ACCG_QUAD_IN, [xo] "v"(s.x_out), [ao] "v"(s.a_out));
        else
          asm volatile(ACCG_ROW0(ACCG_DPP_WAVE) ACCG_ROW6(1, "X0", "M0", "t0") ACCG_ROW6(2, "X1", "M1", "t1") ACCG_ROW6(3, "X2", "M2", "t2")
                       : ACCG_QUAD_OUT : ACCG_QUAD_IN, [xo] "v"(s.x_out), [ao] "v"(s.a_out));
      } else {
        asm volatile(ACCG_ROW6(0, "Xp", "Mp", "tc") ACCG_ROW6(1, "X0", "M0", "t0") ACCG_ROW6(2, "X1", "M1", "t1") ACCG_ROW6(3, "X2", "M2", "t2")
                     : ACCG_QUAD_OUT : ACCG_QUAD_IN, [mx0] "v"(s.pMX[K0]), [Xp] "v"(s.X[K0 - 1]), [Mp] "v"(s.M[K0 - 1]), [tc] "v"(tc));
      }
#undef ACCG_ROW6
#undef ACCG_ROW0
#undef ACCG_QUAD_OUT
#undef ACCG_QUAD_IN
      if (K0 + 4 < K) tc = t3; else a_new = t3;
    } else {
#pragma unroll
    for (int k = K0; k < K1; k++) {
      if constexpr (X6) {
        // two rows per asm statement (see lgkm_wait_visible): rows (k, k + 1) of the same quad, k even inside the quad
        if ((k - K0) % 2 == 1) continue;                                   // done as the second row of the pair in front of it
        if (k + 1 < K1) {
          const float d0 = dq.get(k), d1 = dq.get(k + 1);
          const float gn0 = s.pGM[k + 1], mn0 = s.pMM[k + 1];
          const float gn1 = (k + 2 < K) ? s.pGM[k + 2] : s.nGM, mn1 = (k + 2 < K) ? s.pMM[k + 2] : s.nMM;
          float t0, t1;
          if (k == 0) {
            if (LPP <= 16) {
              asm volatile(
                  "v_fma_f32 %[t0], %[X0], %[gn0], %[Y0]\n\t"
                  "v_mul_f32 %[Y0], %[Y0], %[xx0]\n\t"
                  "v_fmac_f32 %[t0], %[M0], %[mn0]\n\t"
                  "v_fmac_f32 %[Y0], %[M0], %[my0]\n\t"
                  "v_mov_b32_dpp %[X0], %[xo] " ACCG_DPP_ROW "\n\t"
                  "v_mul_f32_dpp %[M0], %[ao], %[d0] " ACCG_DPP_ROW "\n\t"
                  "v_fma_f32 %[t1], %[X1], %[gn1], %[Y1]\n\t"
                  "v_mul_f32 %[Y1], %[Y1], %[xx1]\n\t"
                  "v_fma_f32 %[X1], %[X0], %[mx1], %[M0]\n\t"
                  "v_fmac_f32 %[t1], %[M1], %[mn1]\n\t"
                  "v_fmac_f32 %[Y1], %[M1], %[my1]\n\t"
                  "v_mul_f32 %[M1], %[d1], %[t0]"
                  : [t0] "=&v"(t0), [t1] "=&v"(t1), [X0] "+v"(s.X[0]), [Y0] "+v"(s.Y[0]), [M0] "+v"(s.M[0]), [X1] "+v"(s.X[1]), [Y1] "+v"(s.Y[1]),
                    [M1] "+v"(s.M[1])
                  : [gn0] "v"(gn0), [xx0] "v"(s.pXX[0]), [mn0] "v"(mn0), [my0] "v"(s.pMY[0]), [xo] "v"(s.x_out), [ao] "v"(s.a_out), [d0] "v"(d0),
                    [gn1] "v"(gn1), [xx1] "v"(s.pXX[1]), [mn1] "v"(mn1), [my1] "v"(s.pMY[1]), [mx1] "v"(s.pMX[1]), [d1] "v"(d1));
            } else {
              asm volatile(
                  "v_fma_f32 %[t0], %[X0], %[gn0], %[Y0]\n\t"
                  "v_mul_f32 %[Y0], %[Y0], %[xx0]\n\t"
                  "v_fmac_f32 %[t0], %[M0], %[mn0]\n\t"
                  "v_fmac_f32 %[Y0], %[M0], %[my0]\n\t"
                  "v_mov_b32_dpp %[X0], %[xo] " ACCG_DPP_WAVE "\n\t"
                  "v_mul_f32_dpp %[M0], %[ao], %[d0] " ACCG_DPP_WAVE "\n\t"
                  "v_fma_f32 %[t1], %[X1], %[gn1], %[Y1]\n\t"
                  "v_mul_f32 %[Y1], %[Y1], %[xx1]\n\t"
                  "v_fma_f32 %[X1], %[X0], %[mx1], %[M0]\n\t"
                  "v_fmac_f32 %[t1], %[M1], %[mn1]\n\t"
                  "v_fmac_f32 %[Y1], %[M1], %[my1]\n\t"
                  "v_mul_f32 %[M1], %[d1], %[t0]"
                  : [t0] "=&v"(t0), [t1] "=&v"(t1), [X0] "+v"(s.X[0]), [Y0] "+v"(s.Y[0]), [M0] "+v"(s.M[0]), [X1] "+v"(s.X[1]), [Y1] "+v"(s.Y[1]),
                    [M1] "+v"(s.M[1])
                  : [gn0] "v"(gn0), [xx0] "v"(s.pXX[0]), [mn0] "v"(mn0), [my0] "v"(s.pMY[0]), [xo] "v"(s.x_out), [ao] "v"(s.a_out), [d0] "v"(d0),
                    [gn1] "v"(gn1), [xx1] "v"(s.pXX[1]), [mn1] "v"(mn1), [my1] "v"(s.pMY[1]), [mx1] "v"(s.pMX[1]), [d1] "v"(d1));
            }
          } else {
            asm volatile(
                "v_fma_f32 %[t0], %[X0], %[gn0], %[Y0]\n\t"
                "v_mul_f32 %[Y0], %[Y0], %[xx0]\n\t"
                "v_fma_f32 %[X0], %[Xp], %[mx0], %[Mp]\n\t"
                "v_fmac_f32 %[t0], %[M0], %[mn0]\n\t"
                "v_fmac_f32 %[Y0], %[M0], %[my0]\n\t"
                "v_mul_f32 %[M0], %[d0], %[tc]\n\t"
                "v_fma_f32 %[t1], %[X1], %[gn1], %[Y1]\n\t"
                "v_mul_f32 %[Y1], %[Y1], %[xx1]\n\t"
                "v_fma_f32 %[X1], %[X0], %[mx1], %[M0]\n\t"
                "v_fmac_f32 %[t1], %[M1], %[mn1]\n\t"
                "v_fmac_f32 %[Y1], %[M1], %[my1]\n\t"
                "v_mul_f32 %[M1], %[d1], %[t0]"
                : [t0] "=&v"(t0), [t1] "=&v"(t1), [X0] "+v"(s.X[k]), [Y0] "+v"(s.Y[k]), [M0] "+v"(s.M[k]), [X1] "+v"(s.X[k + 1]),
                  [Y1] "+v"(s.Y[k + 1]), [M1] "+v"(s.M[k + 1])
                : [gn0] "v"(gn0), [xx0] "v"(s.pXX[k]), [mn0] "v"(mn0), [my0] "v"(s.pMY[k]), [mx0] "v"(s.pMX[k]), [d0] "v"(d0),
                  [Xp] "v"(s.X[k - 1]), [Mp] "v"(s.M[k - 1]), [tc] "v"(tc),
                  [gn1] "v"(gn1), [xx1] "v"(s.pXX[k + 1]), [mn1] "v"(mn1), [my1] "v"(s.pMY[k + 1]), [mx1] "v"(s.pMX[k + 1]), [d1] "v"(d1));
          }
          if (k + 2 < K) tc = t1; else a_new = t1;
          continue;
        }
      }
      const float dk = dq.get(k);
      const float gn = (k + 1 < K) ? s.pGM[k + 1] : s.nGM;
      const float mn = (k + 1 < K) ? s.pMM[k + 1] : s.nMM;
      float tn;
      if (k == 0) {
        // X[0] = x_in and M[0] = d[0] * a_in: the two lane shifts ride on the instructions that consume them
        if (LPP <= 16) {
          asm volatile(
              "v_fma_f32 %[tn], %[X], %[gn], %[Y]\n\t"
              "v_mul_f32 %[Y], %[Y], %[xx]\n\t"
              "v_fmac_f32 %[tn], %[M], %[mn]\n\t"
              "v_fmac_f32 %[Y], %[M], %[my]\n\t"
              "v_mov_b32_dpp %[X], %[xo] " ACCG_DPP_ROW "\n\t"
              "v_mul_f32_dpp %[M], %[ao], %[d] " ACCG_DPP_ROW
              : [tn] "=&v"(tn), [X] "+v"(s.X[0]), [Y] "+v"(s.Y[0]), [M] "+v"(s.M[0])
              : [gn] "v"(gn), [xx] "v"(s.pXX[0]), [mn] "v"(mn), [my] "v"(s.pMY[0]), [xo] "v"(s.x_out), [ao] "v"(s.a_out), [d] "v"(dk));
        } else {
          asm volatile(
              "v_fma_f32 %[tn], %[X], %[gn], %[Y]\n\t"
              "v_mul_f32 %[Y], %[Y], %[xx]\n\t"
              "v_fmac_f32 %[tn], %[M], %[mn]\n\t"
              "v_fmac_f32 %[Y], %[M], %[my]\n\t"
              "v_mov_b32_dpp %[X], %[xo] " ACCG_DPP_WAVE "\n\t"
              "v_mul_f32_dpp %[M], %[ao], %[d] " ACCG_DPP_WAVE
              : [tn] "=&v"(tn), [X] "+v"(s.X[0]), [Y] "+v"(s.Y[0]), [M] "+v"(s.M[0])
              : [gn] "v"(gn), [xx] "v"(s.pXX[0]), [mn] "v"(mn), [my] "v"(s.pMY[0]), [xo] "v"(s.x_out), [ao] "v"(s.a_out), [d] "v"(dk));
        }
      } else if (X6) {
        // six-operation form: X is kept divided by the row's pMX (Xs = X / pMX[k]), so its update is one fma,
        // Xs[k] = M[k-1] + (pXX[k] pMX[k-1] / pMX[k]) Xs[k-1], and the factor comes back inside the coefficient the diagonal
        // term multiplies it with anyway (the pGM slot holds pGM[k+1] pMX[k], the pMX slot the chain coefficient)
        asm volatile(
            "v_fma_f32 %[tn], %[X], %[gn], %[Y]\n\t"
            "v_mul_f32 %[Y], %[Y], %[xx]\n\t"
            "v_fma_f32 %[X], %[Xp], %[mx], %[Mp]\n\t"
            "v_fmac_f32 %[tn], %[M], %[mn]\n\t"
            "v_fmac_f32 %[Y], %[M], %[my]\n\t"
            "v_mul_f32 %[M], %[d], %[tc]"
            : [tn] "=&v"(tn), [X] "+v"(s.X[k]), [Y] "+v"(s.Y[k]), [M] "+v"(s.M[k])
            : [gn] "v"(gn), [xx] "v"(s.pXX[k]), [mn] "v"(mn), [my] "v"(s.pMY[k]), [Xp] "v"(s.X[k - 1]), [Mp] "v"(s.M[k - 1]),
              [mx] "v"(s.pMX[k]), [d] "v"(dk), [tc] "v"(tc));
      } else {
        // every result is consumed at least three instructions after it is produced (the X product of this row is the
        // fourth instruction after the X of the row above, the M of this row the sixth before its use by the row below)
        asm volatile(
            "v_fma_f32 %[tn], %[X], %[gn], %[Y]\n\t"
            "v_mul_f32 %[Y], %[Y], %[xx]\n\t"
            "v_mul_f32 %[X], %[Xp], %[xx]\n\t"
            "v_fmac_f32 %[tn], %[M], %[mn]\n\t"
            "v_fmac_f32 %[Y], %[M], %[my]\n\t"
            "v_fmac_f32 %[X], %[Mp], %[mx]\n\t"
            "v_mul_f32 %[M], %[d], %[tc]"
            : [tn] "=&v"(tn), [X] "+v"(s.X[k]), [Y] "+v"(s.Y[k]), [M] "+v"(s.M[k])
            : [gn] "v"(gn), [xx] "v"(s.pXX[k]), [mn] "v"(mn), [my] "v"(s.pMY[k]), [Xp] "v"(s.X[k - 1]), [Mp] "v"(s.M[k - 1]),
              [mx] "v"(s.pMX[k]), [d] "v"(dk), [tc] "v"(tc));
      }
      if (k + 1 < K) tc = tn; else a_new = tn;
    }
    }
    if constexpr (Q + 1 < DistRegs<K>::QT) lgkm_wait_visible<DistRegs<K>::QT - 1>();
    dq.template load<Q>(addr_next, tail_adj);          // the same registers, for the next step
    column_rows<LPP, K, X6, Q + 1>(s, dq, addr_next, tail_adj, tc, a_new);
  }
}

// ---- five-operation form (round 3) --------------------------------------------------------------------------------------------
// On top of Xs = X / pMX (six-operation form) Y is kept divided by the row's pMY, Ys[c] = fma(Ys[c-1], pYY, M[c-1]) -- one operation
// instead of two -- and the diagonal term divided by the CONSUMER row's pMM, so that M enters it with coefficient 1:
//     T[k] = fma(Ys[k], b[k], fma(Xs[k], a[k], M[k])),   a[k] = (pMX[k] pGM[k+1]) / pMM[k+1],   b[k] = (sY[k] pGM[k+1]) / pMM[k+1]
// (sY = pMY of a read row, 1 for a clone of row 0, whose Ys is Y = INIT/H itself); the consumer's pMM comes back through its emission
// values: the LDS table of such a wavefront holds dist * pMM (free: it is per (base, row)).  Restates baseline_impl.cpp:84-86 /
// avx-pairhmm-template.h:183-198; oracle model orc_phmm_forward_f32_fma5, bit-identical.  Per row: 3 VOP3 fmas, one fmac, one mul;
// 8 K + ~23 registers (the pMY slots are not referenced and cost nothing).  Coefficient slots: pGM[k] = a[k-1], pMM[k] = b[k-1],
// pMX[k] = chain coefficient of Xs (as in the six-operation form), pXX[k] = pYY[k]; nGM / nMM / nXX the same for row 0 of the lane
// to the right.  Reads must pass the host's range test (phmm_read_form): the six-operation bound, every pMM >= 1/16, every pYY <= 31/32.
#define ACCG_R5(n, xp, mp, tcin)                                  \
  "v_fma_f32 %[t" #n "], %[X" #n "], %[gn" #n "], %[M" #n "]\n\t" \
  "v_fma_f32 %[X" #n "], %[" xp "], %[mx" #n "], %[" mp "]\n\t"   \
  "v_fmac_f32 %[t" #n "], %[Y" #n "], %[mn" #n "]\n\t"            \
  "v_fma_f32 %[Y" #n "], %[Y" #n "], %[xx" #n "], %[M" #n "]\n\t" \
  "v_mul_f32 %[M" #n "], %[d" #n "], %[" tcin "]\n\t"
#define ACCG_R5_DPP(dpp)                             \
  "v_fma_f32 %[t0], %[X0], %[gn0], %[M0]\n\t"        \
  "v_fmac_f32 %[t0], %[Y0], %[mn0]\n\t"              \
  "v_fma_f32 %[Y0], %[Y0], %[xx0], %[M0]\n\t"        \
  "v_mov_b32_dpp %[X0], %[xo] " dpp "\n\t"           \
  "v_mul_f32_dpp %[M0], %[ao], %[d0] " dpp "\n\t"
#define ACCG_A5_FIRST1(dpp) ACCG_R5_DPP(dpp)
#define ACCG_A5_FIRST2(dpp) ACCG_A5_FIRST1(dpp) ACCG_R5(1, "X0", "M0", "t0")
#define ACCG_A5_FIRST3(dpp) ACCG_A5_FIRST2(dpp) ACCG_R5(2, "X1", "M1", "t1")
#define ACCG_A5_FIRST4(dpp) ACCG_A5_FIRST3(dpp) ACCG_R5(3, "X2", "M2", "t2")
#define ACCG_A5_NEXT1 ACCG_R5(0, "Xp", "Mp", "tc")
#define ACCG_A5_NEXT2 ACCG_A5_NEXT1 ACCG_R5(1, "X0", "M0", "t0")
#define ACCG_A5_NEXT3 ACCG_A5_NEXT2 ACCG_R5(2, "X1", "M1", "t1")
#define ACCG_A5_NEXT4 ACCG_A5_NEXT3 ACCG_R5(3, "X2", "M2", "t2")
#define ACCG_O5_1 [t0] "=&v"(t0), [X0] "+v"(s.X[K0]), [Y0] "+v"(s.Y[K0]), [M0] "+v"(s.M[K0])
#define ACCG_O5_2 ACCG_O5_1, [t1] "=&v"(t1), [X1] "+v"(s.X[K0 + 1]), [Y1] "+v"(s.Y[K0 + 1]), [M1] "+v"(s.M[K0 + 1])
#define ACCG_O5_3 ACCG_O5_2, [t2] "=&v"(t2), [X2] "+v"(s.X[K0 + 2]), [Y2] "+v"(s.Y[K0 + 2]), [M2] "+v"(s.M[K0 + 2])
#define ACCG_O5_4 ACCG_O5_3, [t3] "=&v"(t3), [X3] "+v"(s.X[K0 + 3]), [Y3] "+v"(s.Y[K0 + 3]), [M3] "+v"(s.M[K0 + 3])
#define ACCG_I5_1 [gn0] "v"(gn0), [mn0] "v"(mn0), [xx0] "v"(s.pXX[K0]), [d0] "v"(dq.get(K0))
#define ACCG_I5_2 ACCG_I5_1, [gn1] "v"(gn1), [mn1] "v"(mn1), [xx1] "v"(s.pXX[K0 + 1]), [d1] "v"(dq.get(K0 + 1)), [mx1] "v"(s.pMX[K0 + 1])
#define ACCG_I5_3 ACCG_I5_2, [gn2] "v"(gn2), [mn2] "v"(mn2), [xx2] "v"(s.pXX[K0 + 2]), [d2] "v"(dq.get(K0 + 2)), [mx2] "v"(s.pMX[K0 + 2])
#define ACCG_I5_4 ACCG_I5_3, [gn3] "v"(gn3), [mn3] "v"(mn3), [xx3] "v"(s.pXX[K0 + 3]), [d3] "v"(dq.get(K0 + 3)), [mx3] "v"(s.pMX[K0 + 3])
#define ACCG_I5_FIRST [xo] "v"(s.x_out), [ao] "v"(s.a_out)
#define ACCG_I5_NEXT [mx0] "v"(s.pMX[K0]), [Xp] "v"(s.X[K0 - 1]), [Mp] "v"(s.M[K0 - 1]), [tc] "v"(tc)
template <int LPP, int K, int Q>
__device__ __forceinline__ void column_rows5(Rows<float, K>& s, DistRegs<K>& dq, unsigned addr_next, unsigned tail_adj, float& tc, float& a_new) {
  if constexpr (Q < DistRegs<K>::QT) {
    // all rows of a quad (1 to 4 of them: the last quad of a K that is not a multiple of four is shorter) in ONE asm statement, see
    // column_rows; waits and reloads exactly as there
    constexpr int K0 = 4 * Q, K1 = (4 * Q + 4 < K) ? 4 * Q + 4 : K, NR = K1 - K0;
    // what row K0 + i's Xs / Ys are multiplied with in the term it hands to the row below (the lane's last row: to the lane on the right)
    const float gn0 = (K0 + 1 < K) ? s.pGM[(K0 + 1 < K) ? K0 + 1 : 0] : s.nGM, mn0 = (K0 + 1 < K) ? s.pMM[(K0 + 1 < K) ? K0 + 1 : 0] : s.nMM;
    const float gn1 = (K0 + 2 < K) ? s.pGM[(K0 + 2 < K) ? K0 + 2 : 0] : s.nGM, mn1 = (K0 + 2 < K) ? s.pMM[(K0 + 2 < K) ? K0 + 2 : 0] : s.nMM;
    const float gn2 = (K0 + 3 < K) ? s.pGM[(K0 + 3 < K) ? K0 + 3 : 0] : s.nGM, mn2 = (K0 + 3 < K) ? s.pMM[(K0 + 3 < K) ? K0 + 3 : 0] : s.nMM;
    const float gn3 = (K0 + 4 < K) ? s.pGM[(K0 + 4 < K) ? K0 + 4 : 0] : s.nGM, mn3 = (K0 + 4 < K) ? s.pMM[(K0 + 4 < K) ? K0 + 4 : 0] : s.nMM;
    float t0, t1, t2, t3;
    if constexpr (Q == 0) {
      if constexpr (LPP <= 16) {
        if constexpr (NR == 4) asm volatile(ACCG_A5_FIRST4(ACCG_DPP_ROW) : ACCG_O5_4 : ACCG_I5_4, ACCG_I5_FIRST);
        else if constexpr (NR == 3) asm volatile(ACCG_A5_FIRST3(ACCG_DPP_ROW) : ACCG_O5_3 : ACCG_I5_3, ACCG_I5_FIRST);
        else if constexpr (NR == 2) asm volatile(ACCG_A5_FIRST2(ACCG_DPP_ROW) : ACCG_O5_2 : ACCG_I5_2, ACCG_I5_FIRST);
        else asm volatile(ACCG_A5_FIRST1(ACCG_DPP_ROW) : ACCG_O5_1 : ACCG_I5_1, ACCG_I5_FIRST);
      } else {
        if constexpr (NR == 4) asm volatile(ACCG_A5_FIRST4(ACCG_DPP_WAVE) : ACCG_O5_4 : ACCG_I5_4, ACCG_I5_FIRST);
        else if constexpr (NR == 3) asm volatile(ACCG_A5_FIRST3(ACCG_DPP_WAVE) : ACCG_O5_3 : ACCG_I5_3, ACCG_I5_FIRST);
        else if constexpr (NR == 2) asm volatile(ACCG_A5_FIRST2(ACCG_DPP_WAVE) : ACCG_O5_2 : ACCG_I5_2, ACCG_I5_FIRST);
        else asm volatile(ACCG_A5_FIRST1(ACCG_DPP_WAVE) : ACCG_O5_1 : ACCG_I5_1, ACCG_I5_FIRST);
      }
    } else {
      if constexpr (NR == 4) asm volatile(ACCG_A5_NEXT4 : ACCG_O5_4 : ACCG_I5_4, ACCG_I5_NEXT);
      else if constexpr (NR == 3) asm volatile(ACCG_A5_NEXT3 : ACCG_O5_3 : ACCG_I5_3, ACCG_I5_NEXT);
      else if constexpr (NR == 2) asm volatile(ACCG_A5_NEXT2 : ACCG_O5_2 : ACCG_I5_2, ACCG_I5_NEXT);
      else asm volatile(ACCG_A5_NEXT1 : ACCG_O5_1 : ACCG_I5_1, ACCG_I5_NEXT);
    }
    const float tl = NR == 4 ? t3 : NR == 3 ? t2 : NR == 2 ? t1 : t0;
    (void)t0; (void)t1; (void)t2; (void)t3; (void)gn1; (void)gn2; (void)gn3; (void)mn1; (void)mn2; (void)mn3;
    if (K1 < K) tc = tl; else a_new = tl;
    if constexpr (Q + 1 < DistRegs<K>::QT) lgkm_wait_visible<DistRegs<K>::QT - 1>();
    dq.template load<Q>(addr_next, tail_adj);          // the same registers, for the next step
    column_rows5<LPP, K, Q + 1>(s, dq, addr_next, tail_adj, tc, a_new);
  }
}
#undef ACCG_R5
#undef ACCG_R5_DPP

// One column for every lane; returns what the group's last lane adds to its running sum of the last read row.
// XF: 0 = seven-operation form, 6 = six-operation form, 5 = five-operation form.
template <int LPP, int K, int XF>
__device__ __forceinline__ float column_f32_asm(Rows<float, K>& s, DistRegs<K>& dq, unsigned addr_next, unsigned tail_adj) {
  float tc = 0.f, a_new = 0.f;
  constexpr bool X6 = XF == 6, X5 = XF == 5;
  if constexpr (X5) column_rows5<LPP, K, 0>(s, dq, addr_next, tail_adj, tc, a_new);
  else column_rows<LPP, K, X6, 0>(s, dq, addr_next, tail_adj, tc, a_new);
  s.a_out = a_new;
  if (X6 || X5) {
    // Xs of row 0 of the lane to the right: the same single fma as inside a lane (nMX is 1, or 0 towards another read's lanes and
    // in a group's last lane: the product with it is exact); the last read row's M + X = M + pMX * Xs likewise
    // (nMX is 1 in every lane but a group's last, whose hand-off nobody needs: what it passes to the first lane of the next read --
    // its M -- lands in the X of a clone of row 0, which is multiplied by the clone's zero coefficients only.  So no product with nMX.)
    // (in assembly: from C the compiler makes this fma a copy of M and an fmac into the copy -- one more issue slot per step)
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(s.x_out) : "v"(s.X[K - 1]), "v"(s.nXX), "v"(s.M[K - 1]));
    float contrib;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(contrib) : "v"(s.X[K - 1]), "v"(s.xl), "v"(s.M[K - 1]));
    return contrib;
  }
  s.x_out = fma_(s.M[K - 1], s.nMX, s.X[K - 1] * s.nXX);
  return (LPP == 32 || LPP == 8) ? s.M[K - 1] + s.X[K - 1] : s.x_out;   // 16 / 64 lanes: the last lane has nMX = nXX = 1, x_out = M + X
}

// ---- the same in fp64 (rescue pass, K <= PHMM_ASM_MAX_K_F64): register pairs, v_fma_f64 / v_mul_f64, seven operations per cell
// in the order of column<false> (bit-identical results).  Compiled by hipcc the fp64 column needs 26 K + 26 registers: from K = 9
// on that is more than the 256 architectural VGPRs, the rest goes through AGPR copies and one wavefront per SIMD is left; in place
// it is 18 K + ~40, two wavefronts per SIMD at K = 10.  The lane shifts of 64-bit values stay with the compiler (two DPP moves each).
typedef double d2v __attribute__((ext_vector_type(2)));
template <int K>
struct DistRegsD {
  static constexpr int FULL = K / 2, REM = K % 2, QT = (K + 1) / 2;
  d2v q[FULL > 0 ? FULL : 1];
  double tail;
  __device__ __forceinline__ double get(int k) const { return k / 2 < FULL ? q[k / 2][k % 2] : tail; }
  __device__ __forceinline__ void keep() const {         // see DistRegs::keep
#pragma unroll
    for (int i = 0; i < (FULL > 0 ? FULL : 0); i++) asm volatile("" ::"v"(q[i]));
    if constexpr (K % 2 != 0) asm volatile("" ::"v"(tail));
  }
  template <int Q> __device__ __forceinline__ void load(unsigned addr, unsigned tail_adj) {
    if constexpr (Q < FULL) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[Q]) : "v"(addr), "n"(Q * 1024));
    else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(tail) : "v"(addr + tail_adj), "n"(Q * 1024));
  }
};
template <int K, int... Q>
__device__ __forceinline__ void load_all_quads(DistRegsD<K>& dq, unsigned addr, unsigned tail_adj, std::integer_sequence<int, Q...>) { (dq.template load<Q>(addr, tail_adj), ...); }

template <int LPP, int K, int Q>
__device__ __forceinline__ void column_rows_f64(Rows<double, K>& s, DistRegsD<K>& dq, unsigned addr_next, unsigned tail_adj, double a_in, double x_in,
                                                double& tc, double& a_new) {
  if constexpr (Q < DistRegsD<K>::QT) {
    lgkm_wait<DistRegsD<K>::QT>();
#pragma unroll
    for (int k = 2 * Q; k < 2 * Q + 2 && k < K; k++) {
      const double dk = dq.get(k);
      const double gn = (k + 1 < K) ? s.pGM[k + 1] : s.nGM;
      const double mn = (k + 1 < K) ? s.pMM[k + 1] : s.nMM;
      double tn;
      if (k == 0) {
        asm volatile(
            "v_fma_f64 %[tn], %[X], %[gn], %[Y]\n\t"
            "v_mul_f64 %[Y], %[Y], %[xx]\n\t"
            "v_fma_f64 %[tn], %[M], %[mn], %[tn]\n\t"
            "v_fma_f64 %[Y], %[M], %[my], %[Y]\n\t"
            "v_mul_f64 %[M], %[d], %[ai]"
            : [tn] "=&v"(tn), [Y] "+v"(s.Y[0]), [M] "+v"(s.M[0])
            : [X] "v"(s.X[0]), [gn] "v"(gn), [xx] "v"(s.pXX[0]), [mn] "v"(mn), [my] "v"(s.pMY[0]), [d] "v"(dk), [ai] "v"(a_in));
        s.X[0] = x_in;
      } else {
        asm volatile(
            "v_fma_f64 %[tn], %[X], %[gn], %[Y]\n\t"
            "v_mul_f64 %[Y], %[Y], %[xx]\n\t"
            "v_mul_f64 %[X], %[Xp], %[xx]\n\t"
            "v_fma_f64 %[tn], %[M], %[mn], %[tn]\n\t"
            "v_fma_f64 %[Y], %[M], %[my], %[Y]\n\t"
            "v_fma_f64 %[X], %[Mp], %[mx], %[X]\n\t"
            "v_mul_f64 %[M], %[d], %[tc]"
            : [tn] "=&v"(tn), [X] "+v"(s.X[k]), [Y] "+v"(s.Y[k]), [M] "+v"(s.M[k])
            : [gn] "v"(gn), [xx] "v"(s.pXX[k]), [mn] "v"(mn), [my] "v"(s.pMY[k]), [Xp] "v"(s.X[k - 1]), [Mp] "v"(s.M[k - 1]),
              [mx] "v"(s.pMX[k]), [d] "v"(dk), [tc] "v"(tc));
      }
      if (k + 1 < K) tc = tn; else a_new = tn;
    }
    dq.template load<Q>(addr_next, tail_adj);
    column_rows_f64<LPP, K, Q + 1>(s, dq, addr_next, tail_adj, a_in, x_in, tc, a_new);
  }
}
template <int LPP, int K>
__device__ __forceinline__ double column_f64_asm(Rows<double, K>& s, DistRegsD<K>& dq, unsigned addr_next, unsigned tail_adj) {
  const double a_in = group_shr1<LPP>(s.a_out), x_in = group_shr1<LPP>(s.x_out);
  double tc = 0.0, a_new = 0.0;
  column_rows_f64<LPP, K, 0>(s, dq, addr_next, tail_adj, a_in, x_in, tc, a_new);
  s.a_out = a_new;
  s.x_out = fma_(s.M[K - 1], s.nMX, s.X[K - 1] * s.nXX);
  return (LPP == 32 || LPP == 8) ? s.M[K - 1] + s.X[K - 1] : s.x_out;
}

// ... and the five-operation form in fp64 (the rescue pass of a batch all of whose reads pass the form's range tests): the same
// arithmetic as column_rows5 in register pairs -- five operations of the half-rate fp64 pipe per cell instead of seven.
template <int LPP, int K, int Q>
__device__ __forceinline__ void column_rows5_f64(Rows<double, K>& s, DistRegsD<K>& dq, unsigned addr_next, unsigned tail_adj, double a_in, double x_in,
                                                 double& tc, double& a_new) {
  if constexpr (Q < DistRegsD<K>::QT) {
    lgkm_wait<DistRegsD<K>::QT>();
#pragma unroll
    for (int k = 2 * Q; k < 2 * Q + 2 && k < K; k++) {
      const double dk = dq.get(k);
      const double gn = (k + 1 < K) ? s.pGM[k + 1] : s.nGM;      // a[k]: what this row's Xs enters the term for the row below with
      const double mn = (k + 1 < K) ? s.pMM[k + 1] : s.nMM;      // b[k]: ... its Ys
      double tn;
      if (k == 0) {
        asm volatile(
            "v_fma_f64 %[tn], %[X], %[gn], %[M]\n\t"
            "v_fma_f64 %[tn], %[Y], %[mn], %[tn]\n\t"
            "v_fma_f64 %[Y], %[Y], %[xx], %[M]\n\t"
            "v_mul_f64 %[M], %[d], %[ai]"
            : [tn] "=&v"(tn), [Y] "+v"(s.Y[0]), [M] "+v"(s.M[0])
            : [X] "v"(s.X[0]), [gn] "v"(gn), [mn] "v"(mn), [xx] "v"(s.pXX[0]), [d] "v"(dk), [ai] "v"(a_in));
        s.X[0] = x_in;
      } else {
        asm volatile(
            "v_fma_f64 %[tn], %[X], %[gn], %[M]\n\t"
            "v_fma_f64 %[X], %[Xp], %[mx], %[Mp]\n\t"
            "v_fma_f64 %[tn], %[Y], %[mn], %[tn]\n\t"
            "v_fma_f64 %[Y], %[Y], %[xx], %[M]\n\t"
            "v_mul_f64 %[M], %[d], %[tc]"
            : [tn] "=&v"(tn), [X] "+v"(s.X[k]), [Y] "+v"(s.Y[k]), [M] "+v"(s.M[k])
            : [gn] "v"(gn), [mn] "v"(mn), [xx] "v"(s.pXX[k]), [Xp] "v"(s.X[k - 1]), [Mp] "v"(s.M[k - 1]), [mx] "v"(s.pMX[k]), [d] "v"(dk), [tc] "v"(tc));
      }
      if (k + 1 < K) tc = tn; else a_new = tn;
    }
    dq.template load<Q>(addr_next, tail_adj);
    column_rows5_f64<LPP, K, Q + 1>(s, dq, addr_next, tail_adj, a_in, x_in, tc, a_new);
  }
}
template <int LPP, int K>
__device__ __forceinline__ double column_f64_asm5(Rows<double, K>& s, DistRegsD<K>& dq, unsigned addr_next, unsigned tail_adj) {
  const double a_in = group_shr1<LPP>(s.a_out), x_in = group_shr1<LPP>(s.x_out);
  double tc = 0.0, a_new = 0.0;
  column_rows5_f64<LPP, K, 0>(s, dq, addr_next, tail_adj, a_in, x_in, tc, a_new);
  s.a_out = a_new;
  s.x_out = fma_(s.X[K - 1], s.nXX, s.M[K - 1]);           // Xs of row 0 of the lane to the right
  return fma_(s.X[K - 1], s.xl, s.M[K - 1]);               // M + X of the lane's last row
}

// what the sweep needs to know about the assembly column of a value type
template <typename T, int K> struct AsmCol;
template <int K> struct AsmCol<float, K> {
  typedef DistRegs<K> Regs;
  template <int LPP, int XF> static __device__ __forceinline__ float column(Rows<float, K>& s, Regs& dq, unsigned a, unsigned t) { return column_f32_asm<LPP, K, XF>(s, dq, a, t); }
};
template <int K> struct AsmCol<double, K> {
  typedef DistRegsD<K> Regs;
  template <int LPP, int XF> static __device__ __forceinline__ double column(Rows<double, K>& s, Regs& dq, unsigned a, unsigned t) {
    if constexpr (XF == 5) return column_f64_asm5<LPP, K>(s, dq, a, t);
    else return column_f64_asm<LPP, K>(s, dq, a, t);
  }
};

// One wavefront per workgroup (measured: 256-thread workgroups of four independent jobs change nothing and
// would cap the fp64 rescue kernel's LDS).
#ifdef PHMM_TIMING
static __device__ unsigned long long g_phmm_t[8];
static __device__ unsigned long long g_phmm_w[4] = {~0ull, 0ull, ~0ull, 0ull};   // wall clock: first / last job start, first / last job end (last launch)
// where the jobs of the last launch ran: wavefronts per (XCD, shader engine, CU, SIMD) and their durations
static __device__ unsigned g_phmm_simd_n[32768];
static __device__ unsigned long long g_phmm_simd_t[32768];
static __global__ void phmm_timing_reset_wall() {
  g_phmm_w[0] = g_phmm_w[2] = ~0ull; g_phmm_w[1] = g_phmm_w[3] = 0;
  for (int i = 0; i < 32768; i++) { g_phmm_simd_n[i] = 0; g_phmm_simd_t[i] = 0; }
}
static __global__ void phmm_timing_print() {
  const double n = (double)g_phmm_t[0];
  printf("phmm timing per job (s_memtime ticks): prologue %.0f = stream %.0f + row loads %.0f + dist table %.0f + rest %.0f; sweep %.0f; shader clock over the job %.3f GHz\n", g_phmm_t[1] / n,
         g_phmm_t[3] / n, g_phmm_t[4] / n, g_phmm_t[5] / n, g_phmm_t[6] / n, g_phmm_t[2] / n, (double)(g_phmm_t[1] + g_phmm_t[2]) / (double)g_phmm_t[7] * 0.1);
  printf("  last launch (wall clock, us): job starts spread over %.1f, first start to first end %.1f, to last end %.1f\n", (g_phmm_w[1] - g_phmm_w[0]) * 0.01,
         (g_phmm_w[2] - g_phmm_w[0]) * 0.01, (g_phmm_w[3] - g_phmm_w[0]) * 0.01);
  {
    unsigned hist[12] = {0}; double tsum[12] = {0}; unsigned cu_hist[40] = {0};
    for (int cu = 0; cu < 32768 / 4; cu++) {
      unsigned tot = 0;
      for (int sd = 0; sd < 4; sd++) {
        const int i = (cu >> 8) * 1024 + ((cu & 255) << 2 | sd);       // key = xcc[14:12] : hw_id[15:6] : simd -- see the recording side
        (void)i;
      }
      (void)tot;
    }
    for (int i = 0; i < 32768; i++) {
      const unsigned n = g_phmm_simd_n[i];
      if (n) { hist[n < 11 ? n : 11]++; tsum[n < 11 ? n : 11] += (double)g_phmm_simd_t[i] / n; }
    }
    for (int i = 0; i < 32768; i += 4) { const unsigned t = g_phmm_simd_n[i] + g_phmm_simd_n[i + 1] + g_phmm_simd_n[i + 2] + g_phmm_simd_n[i + 3]; if (t) cu_hist[t < 39 ? t : 39]++; }
    printf("  SIMDs by wavefronts of the last launch they ran (count: mean job us):");
    for (int n = 1; n < 12; n++) if (hist[n]) printf("  %d waves: %u SIMDs, %.1f us", n, hist[n], tsum[n] / hist[n] * 0.01);
    printf("\n  CUs by wavefronts:");
    for (int n = 1; n < 40; n++) if (cu_hist[n]) printf("  %d: %u", n, cu_hist[n]);
    printf("\n");
  }
  for (int i = 0; i < 8; i++) g_phmm_t[i] = 0;
  g_phmm_w[0] = g_phmm_w[2] = ~0ull; g_phmm_w[1] = g_phmm_w[3] = 0;
}
#endif
// W = wavefronts per workgroup: 1, or 2 whose jobs (work items 2 b and 2 b + 1 of the launch) list the SAME reads against different
// runs of haplotypes.  The dist table depends on the reads only, so the two share one (each writes half of its slabs) and only the
// stream and its bookkeeping are per wavefront: 13.3 + 2 x 1.4 KB for two wavefronts at K = 13 instead of 2 x 14.7 -- sixteen
// wavefronts on a CU instead of ten, i.e. four per SIMD, where a VOP3 instruction issues in 1.2 ns instead of 1.6
// (profiles/r02_ubench2.txt; the five-operation column is three VOP3 fmas in five).
template <typename T, int K, int LPP, bool STRICT, bool RESCUE, int XF = 0, bool STRIPED = false, int W = 1>
__device__ __forceinline__ bool phmm_job(const PhmmArgs<T>& a, uint32_t work_base, const uint32_t job, const bool count_rescued = true) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool X6 = XF == 6, X5 = XF == 5;     // six- / five-operation form of the fp32 fast sweep (0: seven operations)
  // the five-operation fp32 pass runs on prepared inputs: per-row records (phmm_prepare_rows) and streams laid out at batch creation
  constexpr bool PRE = X5 && !RESCUE && !STRIPED && sizeof(T) == 4;
  constexpr int VN = Vec16<T>::N, QT = (K + VN - 1) / VN;
  constexpr bool COMPACT = phmm_is_compact((int)sizeof(T), STRICT, K) && !STRIPED;
  constexpr unsigned SLAB = phmm_slab_bytes(K, (int)sizeof(T), COMPACT);   // bytes between two bases' tables
  static_assert(W == 1 || !STRIPED, "shared-table workgroups: not for reads swept in stripes");
  unsigned char* tab = smem;
  const int wave = W > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  T* y0s = reinterpret_cast<T*>(smem + a.nchar * SLAB + (W > 1 ? wave * (int)phmm_wave_area_bytes((int)sizeof(T), a.stream_cap, a.haps_cap, LPP, STRIPED) : 0));
  uint32_t* hcol = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(y0s) + phmm_align16(sizeof(T) * (a.haps_cap + 1)));
  uint32_t* bpos = hcol + a.haps_cap + 1;
  // (entry LPP - 1, where the first haplotype's bubble sits, is 16-byte aligned: a stream laid out at batch creation is copied in in 16-byte units)
  uint8_t* stream = reinterpret_cast<uint8_t*>(hcol) + phmm_align16((size_t)(2 * a.haps_cap + 3) * 4) + ((16 - (LPP - 1) % 16) % 16);
  uint2* stash = reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(y0s) + phmm_wave_area_bytes((int)sizeof(T), a.stream_cap, a.haps_cap, LPP, STRIPED) - PHMM_STASH_BYTES);
  // striped reads (more rows than a wavefront holds): what the last lane hands "to the right", per stream position, for the next stripe
  T* carry_a = reinterpret_cast<T*>(reinterpret_cast<uint8_t*>(hcol) + phmm_align16((size_t)(2 * a.haps_cap + 3) * 4) + phmm_align16((size_t)2 * LPP + a.stream_cap + PHMM_STREAM_SLACK));
  T* carry_x = carry_a + (a.stream_cap + 2 * LPP + 24);

  const int lane = W > 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
#ifdef PHMM_TIMING
  const unsigned long long tm0 = __builtin_amdgcn_s_memtime(), tw0 = wall_clock64();     // (wall clock: 100 MHz)
#endif
  constexpr int NG = 64 / LPP;              // reads per wavefront
  const int g = lane / LPP, l = lane % LPP;
  // (field-wise loads: indexing a by-value copy of the struct with g would put it in scratch)
  const PhmmWork* wp = a.work + (work_base + job);
  const int n_list = __builtin_amdgcn_readfirstlane((int)wp->n_haps);
  const uint32_t hap_off = __builtin_amdgcn_readfirstlane(wp->hap_off);
  if (W == 1 && n_list == 0) return false;     // the empty second item of an odd pair (phmm_host.cpp), met by a kernel that runs items singly

  const uint32_t ridx = wp->read[g < NG ? g : 0];
  const bool have = ridx != PHMM_NO_READ;
  SeqRef rr = {0u, 0u};
  uint32_t out_base = 0;
  uint32_t rec0 = 0;               // PRE: the read's first row in the per-row records
  if (have) { rr = a.rd[ridx]; out_base = a.rd_out[ridx]; if constexpr (PRE) rec0 = a.rec.row0[ridx]; }
  if (g < NG && l == LPP - 1) stash[g] = make_uint2(out_base, ridx);      // for the off-the-hot-path code of the assembly sweep

  // ---- haplotype stream: [15 pad] bubble hap0 bubble hap1 ... bubble(terminal) [pad] ----------
  // an entry is the byte offset of its base's slab in the dist table; bubbles/padding use slab 0
  // (their dist value is never kept: M is forced to 0 on a bubble, and all state is 0 in the padding)
  int pos = 0, n_haps = 0;
  unsigned n_flag = 0;
  for (int i = lane; i < LPP - 1; i += 64) stream[i] = 0;
  if constexpr (PRE) {
    // The stream as laid out at batch creation, 16 bytes per lane and round: one memory latency (built here, from the haplotypes'
    // bases, it was four dependent levels of loads: 18 000 cycles of a job whose sweep takes 335 000 at four wavefronts per SIMD).
    const uint32_t s_off16 = __builtin_amdgcn_readfirstlane(wp->pad_[0]), s_len = __builtin_amdgcn_readfirstlane(wp->pad_[1]);
    const uint4* src = reinterpret_cast<const uint4*>(a.streams) + s_off16;
    uint4* dst = reinterpret_cast<uint4*>(stream + LPP - 1);
    const int n16 = (int)((s_len + LPP + 20 + 15) / 16);
    uint32_t col_l = 0, hlen_l = 0;
    if (lane < n_list) { const PhmmHapDesc h_ = a.hap_desc[hap_off + lane]; col_l = h_.col; hlen_l = h_.len; }
    for (int i = lane; i < n16; i += 64) dst[i] = src[i];
    // bubble positions = exclusive prefix sums of (length + 1) over the job's haplotypes, one per lane
    const unsigned v = lane < n_list ? hlen_l + 1u : 0u;
    unsigned incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned u = __shfl_up(incl, d); if (lane >= d) incl += u; }
    if (lane < n_list) { bpos[lane] = incl - v; y0s[lane] = a.tab.init / (T)(int)hlen_l; hcol[lane] = col_l; }   // baseline_impl.cpp:63
    pos = (int)__builtin_amdgcn_readlane(incl, 63);
    n_haps = n_list;
  } else {
  // the haplotypes' descriptors, one per lane (a job lists at most PHMM_HAPS_MAX = 48): two latencies for all of them instead
  // of two per haplotype
  uint32_t col_l = 0, hoff_l = 0, hlen_l = 0;
  if (lane < n_list) {
    const PhmmHapDesc h_ = a.hap_desc[hap_off + lane];
    col_l = h_.col; hoff_l = h_.off; hlen_l = h_.len;
  }
  for (int j = 0; j < n_list; j++) {
    const uint32_t col = __builtin_amdgcn_readlane(col_l, j);
    // keep this haplotype only if one of the wavefront's reads underflowed in fp32 against it (a.raw == nullptr: a speculative fp64
    // pass NEXT TO the fp32 sweep of a small batch, phmm_host.cpp run_spec -- every haplotype, nothing counted here)
    if (RESCUE && a.raw) {
      const bool under = have && g < NG && l == 0 && a.raw[out_base + col] < PHMM_MIN_ACCEPTED;
      const unsigned long long m = __ballot(under);
      if (m == 0) continue;
      n_flag += (unsigned)__popcll(m);
    }
    const SeqRef hr = {(uint32_t)__builtin_amdgcn_readlane(hoff_l, j), (uint32_t)__builtin_amdgcn_readlane(hlen_l, j)};
    if (lane == 0) {
      stream[LPP - 1 + pos] = (uint8_t)a.nchar;            // a bubble: the slab index one beyond the tables (its values are never kept)
      bpos[n_haps] = pos;
      y0s[n_haps] = a.tab.init / (T)(int)hr.len;     // baseline_impl.cpp:63
      hcol[n_haps] = col;
    }
    // the bases, eight byte loads per lane in flight before the first one is used (512 bases per round trip: one lane-strided
    // byte at a time, a 300-base haplotype cost five memory latencies in a row -- 19 of them per configs[1] job, 15 us of its 22 us prologue)
    for (int i0 = 0; i0 < (int)hr.len; i0 += 512) {
      uint8_t hb[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = i0 + u * 64 + lane; hb[u] = i < (int)hr.len ? a.hblob[hr.off + i] : (uint8_t)0; }
#pragma unroll
      for (int u = 0; u < 8; u++) { const int i = i0 + u * 64 + lane; if (i < (int)hr.len) stream[LPP - 1 + pos + 1 + i] = (uint8_t)char_index(hb[u]); }
    }
    pos += (int)hr.len + 1;
    n_haps++;
  }
  }
  if (RESCUE) {
    if (W == 1 && n_haps == 0) return false;      // (in a pair this wavefront still writes its share of the table: it leaves behind the barrier)
    if (lane == 0 && count_rescued && n_flag) atomicAdd(a.n_rescued, (unsigned long long)n_flag);
  }
  if (lane == 0) { bpos[n_haps] = pos; bpos[n_haps + 1] = 0x7FFFFFFF; y0s[n_haps] = T(0); hcol[n_haps] = 0; }
  if (!PRE) for (int i = lane; i < LPP + 20; i += 64) stream[LPP - 1 + pos + i] = i == 0 ? (uint8_t)a.nchar : (uint8_t)0;   // terminal bubble + drain + prefetch slack
  const int t_end = __builtin_amdgcn_readfirstlane(pos + LPP);         // the last lane passes the terminal bubble at pos+LPP-1
#ifdef PHMM_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned long long tmA = __builtin_amdgcn_s_memtime();
#endif

  // ---- per-row constants (registers) and the dist table (LDS) ---------------------------------
  // A read of more than LPP * K - 1 bases (STRIPED, 64 lanes x 16 rows) is swept in stripes of LPP * K rows, top to bottom
  // (the reference's CPU code does the same with its 8- or 4-row stripes and shiftOutM/X/Y, avx-pairhmm-template.h:224,265-297):
  // stripe 0 holds the clone of row 0 and the first R - (stripes - 1) * LPP * K read rows, right-aligned as always; every further
  // stripe is full.  The last lane's hand-off values go to carry_a / carry_x indexed by stream position, lane 0 of the next stripe
  // takes them from there; only the last stripe sums up the last read row.
  Rows<T, K> s;
  const int R = (int)rr.len;
  const uint8_t* rb = a.rblob + rr.off;
  constexpr int SROWS = LPP * K;
  const int n_stripes = STRIPED ? __builtin_amdgcn_readfirstlane((R + SROWS) / SROWS) : 1;
  const int rows0 = R - (n_stripes - 1) * SROWS;                 // read rows in stripe 0 (0 .. SROWS - 1)
  bool tiny = false;          // contracted fp64 rescue: a result close enough to the denormal range for the flush pattern to matter
  for (int stripe = 0; stripe < n_stripes; stripe++) {
  const bool last_stripe = stripe == n_stripes - 1;
  const int pad = stripe == 0 ? SROWS - rows0 : 0;              // stripe 0: >= 1 by construction of the job
  const int roff = stripe == 0 ? 0 : rows0 + (stripe - 1) * SROWS;   // read row (0-based) of the stripe's first non-clone flat row
  s.npad = pad - l * K;                     // clones are the first `pad` flat rows
  typedef typename Vec16<T>::type V;
#ifdef PHMM_TIMING
  unsigned long long tmB = tmA, tmC = tmA;
#endif
  if constexpr (PRE) {
    // Prepared rows: the K coefficient records of the lane's rows (their components ARE the sweep's registers), the chain
    // coefficient of the row behind them, pMX of the last one and the first row's clone coefficient, then the dist records into
    // the table -- one memory latency behind the read's descriptor, no table lookups, no divisions (phmm_prepare_rows did them
    // once per read and pass; here they cost every job 20 000 cycles of gathers and every CU 2 700 gather instructions at once).
    // (record of local row k of this lane: rec0 + k * LPP + l; clones of row 0 have records of their own)
    float4 co[K], cn = make_float4(0.f, 0.f, 0.f, 0.f), ml = cn;
#pragma unroll
    for (int k = 0; k < K; k++) co[k] = cn;
    if (have) {
      const float4* rc = a.rec.coef + rec0 + l;
#pragma unroll
      for (int k = 0; k < K; k++) co[k] = rc[k * LPP];
      if (l < LPP - 1) cn = rc[1];                          // row 0 of the lane to the right
      ml = a.rec.misc[rec0 + (K - 1) * LPP + l];            // pMX of this lane's last row
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      s.M[k] = T(0); s.X[k] = T(0); s.Y[k] = T(0);
      s.pXX[k] = co[k].z;                                    // pYY (a clone: 1, it keeps its Ys = INIT/H)
      s.pMX[k] = co[k].w;                                    // chain coefficient of Xs
      s.pMY[k] = T(0);
      if (k + 1 < K) { s.pGM[k + 1] = co[k].x; s.pMM[k + 1] = co[k].y; }     // what this row's Xs / Ys enter the term for the row below with
    }
    s.pGM[0] = T(0); s.pMM[0] = T(0);
    s.nGM = co[K - 1].x; s.nMM = co[K - 1].y;
    s.xl = ml.y;
    s.nXX = cn.w;
    s.nMX = l == LPP - 1 ? T(0) : T(1);
    s.gclone = T(0);
#ifdef PHMM_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    tmB = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int q = 0; q < QT; q++) {
      float4 dd[VN];
      float dn5[VN];
#pragma unroll
      for (int e = 0; e < VN; e++) {
        const int k = q * VN + e;
        dd[e] = make_float4(0.f, 0.f, 0.f, 0.f); dn5[e] = 0.f;
        if (k < K && have) {
          dd[e] = a.rec.dist[rec0 + k * LPP + l];
          if (a.nchar > 4) dn5[e] = a.rec.misc[rec0 + k * LPP + l].x;
        }
      }
#pragma unroll
      for (int c = 0; c < 5; c++) {
        if (c >= a.nchar) break;
        if (W > 1 && (c % W) != wave) continue;       // the workgroup's wavefronts hold the same reads: each writes its share of the slabs
        V v;
#pragma unroll
        for (int e = 0; e < VN; e++) v[e] = c == 0 ? dd[e].x : c == 1 ? dd[e].y : c == 2 ? dd[e].z : c == 3 ? dd[e].w : dn5[e];
        constexpr int TS = phmm_tail_stride(K, (int)sizeof(T));
        if (q < K / VN || TS == 16) *reinterpret_cast<V*>(tab + c * SLAB + q * 1024 + lane * 16) = v;
        else if (TS == 8) { typedef float V2 __attribute__((ext_vector_type(2))); *reinterpret_cast<V2*>(tab + c * SLAB + q * 1024 + lane * 8) = V2{(float)v[0], (float)v[1]}; }
        else *reinterpret_cast<T*>(tab + c * SLAB + q * 1024 + lane * TS) = v[0];
      }
    }
#ifdef PHMM_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    tmC = __builtin_amdgcn_s_memtime();
#endif
  } else {
  // Two batches of loads for the lane's K rows and row 0 of the lane to the right (index K): the five bytes of every row, then
  // the table entries they select -- two memory latencies per job.  (With the loads inside "is this a read row?" branches,
  // one row at a time, a job spent 26 dependent latencies here: a quarter of a wavefront's time on configs[1].)  Rows that are
  // clones of row 0, or past the read, load a clamped row and are overruled below.
  int vq[K + 1], vi[K + 1], vd[K + 1], vc[K + 1], vb[K + 1];
  const int r_hi = R > 0 ? R - 1 : 0;
#pragma unroll
  for (int k = 0; k <= K; k++) {
    int r = roff + l * K + k - pad;
    r = r < 0 ? 0 : r > r_hi ? r_hi : r;
    vb[k] = rb[r]; vq[k] = rb[R + r] & 127; vi[k] = rb[2 * R + r] & 127; vd[k] = rb[3 * R + r] & 127; vc[k] = rb[4 * R + r] & 127;
  }
  T tMM[K + 1], tGM[K + 1], tMX[K + 1], tXX[K + 1], tMY[K], tdM[K], tdX[K];
#pragma unroll
  for (int k = 0; k <= K; k++) {
    const int lo = vi[k] < vd[k] ? vi[k] : vd[k], hi = vi[k] < vd[k] ? vd[k] : vi[k];
    tMM[k] = a.tab.m2m[((hi * (hi + 1)) >> 1) + lo];   // Context.h:163-174
    tGM[k] = a.tab.omph[vc[k]];                          // baseline_impl.cpp:54
    tMX[k] = a.tab.ph[vi[k]];
    tXX[k] = a.tab.ph[vc[k]];
    if (k < K) {
      tMY[k] = a.tab.ph[vd[k]];
      tdM[k] = a.tab.omph[vq[k]];                        // baseline_impl.cpp:79-81
      tdX[k] = a.tab.phd3[vq[k]];                        // baseline_impl.cpp:83
    }
  }
#ifdef PHMM_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  tmB = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
  for (int q = 0; q < QT; q++) {
    T dM[VN], dX[VN];
    int rbase[VN];
#pragma unroll
    for (int e = 0; e < VN; e++) {
      const int k = q * VN + e;
      dM[e] = T(0); dX[e] = T(0); rbase[e] = CH_A;
      if (k < K) {
        const int r = roff + l * K + k - pad;   // 0-based read row, < roff (only in stripe 0: < 0): clone of row 0
        const bool real = r >= roff;
        s.M[k] = T(0); s.X[k] = T(0); s.Y[k] = T(0);
        // clone of row 0: M stays 0 (dist = 0), X stays 0 (0*0 + 0*1), Y keeps INIT/H (0*0 + Y*1)
        s.pMM[k] = real ? tMM[k] : T(0); s.pGM[k] = real ? tGM[k] : T(0); s.pMX[k] = real ? tMX[k] : T(0);
        s.pXX[k] = real ? tXX[k] : T(1); s.pMY[k] = real ? tMY[k] : T(0);
        dM[e] = real ? tdM[k] : T(0); dX[e] = real ? tdX[k] : T(0);
        if constexpr (X5) { dM[e] = dM[e] * s.pMM[k]; dX[e] = dX[e] * s.pMM[k]; }   // five-operation form: the row's pMM rides on its emission values
        rbase[e] = real ? char_index((uint8_t)vb[k]) : CH_A;
      }
    }
    // dist table: one 16-byte vector per (hap base, row quad, lane)
#pragma unroll
    for (int c = 0; c < 5; c++) {
      if (c >= a.nchar) break;
      if (W > 1 && (c % W) != wave) continue;       // the workgroup's wavefronts hold the same reads: each writes its share of the slabs
      V v;
#pragma unroll
      for (int e = 0; e < VN; e++)   // rs == hap || rs == 'N' || hap == 'N'   (baseline_impl.cpp:80)
        v[e] = (c == CH_N || rbase[e] == CH_N || rbase[e] == c) ? dM[e] : dX[e];
      if constexpr (COMPACT) {
        // full vectors at q * 1024 + lane * 16; the K % VN rows behind them packed at the tail stride (phmm_tail_stride)
        constexpr int TS = phmm_tail_stride(K, (int)sizeof(T));
        if (q < K / VN || TS == 16) *reinterpret_cast<V*>(tab + c * SLAB + q * 1024 + lane * 16) = v;
        else if (TS == 8 && sizeof(T) == 4) { typedef float V2 __attribute__((ext_vector_type(2))); *reinterpret_cast<V2*>(tab + c * SLAB + q * 1024 + lane * 8) = V2{(float)v[0], (float)v[1]}; }
        else *reinterpret_cast<T*>(tab + c * SLAB + q * 1024 + lane * TS) = v[0];
      } else {
        *reinterpret_cast<V*>(tab + c * SLAB + q * 1024 + lane * 16) = v;
      }
    }
  }
#ifdef PHMM_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  tmC = __builtin_amdgcn_s_memtime();
#endif
  {
    const int r = roff + (l + 1) * K - pad; // row 0 of the lane to the right (of the next stripe, for the last lane of a stripe that has one)
    if ((l < LPP - 1 || !last_stripe) && r >= roff) { s.nMM = tMM[K]; s.nGM = tGM[K]; s.nMX = tMX[K]; s.nXX = tXX[K]; }
    else if (l == LPP - 1 && LPP != 32 && LPP != 8) { s.nMM = T(0); s.nGM = T(0); s.nMX = T(1); s.nXX = T(1); }
    else { s.nMM = T(0); s.nGM = T(0); s.nMX = T(0); s.nXX = T(0); }
  }
  if (!STRICT && !X5) {
#pragma unroll
    for (int k = 0; k < K; k++) s.pMY[k] = s.pMY[k] * (k + 1 < K ? s.pGM[k + 1] : s.nGM);
  }
  s.xl = T(0); s.gclone = T(0);
  if constexpr (X6) {
    // Six-operation form (see column_rows): from here on the pGM slots hold pGM[k] * pMX[k-1] (what the diagonal term of row k
    // multiplies Xs[k-1] with), the pMX slots the chain coefficient pXX[k] * pMX[k-1] / pMX[k] (0 behind a clone of row 0, whose
    // pMX is 0: X starts at 0 there), nGM / nXX the same two for row 0 of the lane to the right, nMX the factor of M in its Xs
    // (1; 0 towards another read's lanes).  The pGM of the first read row is still needed as such: the clones in front of it
    // carry Y = INIT/H * that pGM (the reset at every bubble).
#pragma unroll
    for (int k = 0; k < K; k++) if (k == s.npad) s.gclone = s.pGM[k];
    if (s.npad >= K) s.gclone = s.nGM;
    s.xl = s.pMX[K - 1];
    const T nmx_true = s.nMX;
    if (l == LPP - 1) {
      s.nXX = T(0); s.nMX = T(0);                                 // nothing to hand over; the running sum takes M + pMX * Xs directly
    } else {
      s.nXX = nmx_true != T(0) ? (s.nXX * s.xl) / nmx_true : T(0);
      s.nMX = T(1);
    }
    s.nGM = s.nGM * s.xl;
#pragma unroll
    for (int k = K - 1; k >= 1; k--) {
      s.pGM[k] = s.pGM[k] * s.pMX[k - 1];
      s.pMX[k] = s.pMX[k] != T(0) ? (s.pXX[k] * s.pMX[k - 1]) / s.pMX[k] : T(0);
    }
  }
  if constexpr (X5) {
    // Five-operation form (see column_rows5).  From here on: pGM[k] = a[k-1], pMM[k] = b[k-1] (what the term row k-1 hands down
    // multiplies its Xs / Ys with), pMX[k] = the chain coefficient of Xs, nGM / nMM / nXX the same three for row 0 of the lane to the
    // right; a consumer with pMM = 0 is a clone of row 0 (an eligible read has none, phmm_read_form) or nobody: coefficients 0.
    s.xl = s.pMX[K - 1];
    {
      const T sY = (K - 1 < s.npad) ? T(1) : s.pMY[K - 1];
      const T g = s.nGM, m = s.nMM, nmx_true = s.nMX;
      s.nGM = m != T(0) ? (s.xl * g) / m : T(0);
      s.nMM = m != T(0) ? (sY * g) / m : T(0);
      if (l == LPP - 1) { s.nXX = T(0); s.nMX = T(0); }
      else { s.nXX = nmx_true != T(0) ? (s.nXX * s.xl) / nmx_true : T(0); s.nMX = T(1); }
    }
#pragma unroll
    for (int k = K - 1; k >= 1; k--) {
      const T sY = (k - 1 < s.npad) ? T(1) : s.pMY[k - 1];
      const T g = s.pGM[k], m = s.pMM[k];
      s.pGM[k] = m != T(0) ? (s.pMX[k - 1] * g) / m : T(0);
      s.pMM[k] = m != T(0) ? (sY * g) / m : T(0);
      s.pMX[k] = s.pMX[k] != T(0) ? (s.pXX[k] * s.pMX[k - 1]) / s.pMX[k] : T(0);
    }
  }
  }
  s.a_out = T(0); s.x_out = T(0); s.acc = T(0);
  if constexpr (W > 1) {
    __syncthreads();                                        // the other wavefront's slabs of the table
    if (n_list == 0 || (RESCUE && n_haps == 0)) return false;   // the empty second job of an odd pair / nothing to rescue here: it has done its share of the table
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // this wave's LDS writes (table, stream) before its own reads
    __builtin_amdgcn_wave_barrier();
  }

  // ---- sweep ----------------------------------------------------------------------------------
  const uint8_t* hs = stream + LPP - 1 - l;           // hs[t] = this lane's column at step t (base index 0..4)
  const unsigned char* tab_lane = tab + lane * 16;
  int t = 0, jn = 0, jl = -1;
  int nb = 0;                 // stream position of the next bubble lane 0 will meet
  unsigned long long rm = 0;  // bit i: lane i of every group is on a bubble this step
  constexpr bool ASMCOL = COMPACT;   // fast mode in fp32, and in fp64 up to K = 10: the column in assembly, dist values single-buffered
  unsigned o1 = hs[1];        // base index of step t+1 (loaded two steps ahead)
  if constexpr (ASMCOL) {
    // One loop, one call site of the column: the state registers then have a single life range set (a second copy of the
    // column for bubble-free runs made the register allocator keep two sets and shuffle between them).  Bubble bookkeeping is
    // scalar (rm, nb, jn live in SGPRs), so a bubble-free step pays a few SALU instructions for it.
    typename AsmCol<T, K>::Regs dq;   // dist of step t; re-loaded for step t+1 quad by quad inside the column
    constexpr int NLD = AsmCol<T, K>::Regs::QT;      // LDS loads of dist values per step
    const unsigned tab_a = lds_addr(tab_lane), hs_a = lds_addr(hs);
    const unsigned slab_s = __builtin_amdgcn_readfirstlane((int)SLAB);
    const unsigned tail_adj = (unsigned)lane * (unsigned)(phmm_tail_stride(K, (int)sizeof(T)) - 16);     // wraps: lane * stride - lane * 16
    unsigned o1n, addr_next;  // stream byte in flight; LDS address of the next step's slab for this lane
    // Which lanes are on a bubble (column 0 of their next haplotype) this step: read off the stream byte itself -- a bubble carries
    // the marker nchar -- by the v_cmp that rides along with the slab address of the byte, one step ahead.  (Kept as scalar
    // bookkeeping -- a mask shifted every step, a compare with the next bubble's position -- this cost six SALU instructions and
    // two branches per step, and issue slots are what the sweep is short of.)
    unsigned any_b, any_next;   // does any lane meet a bubble this step / the next one
    const unsigned nchar_s = __builtin_amdgcn_readfirstlane((int)a.nchar);
    constexpr int U = 8;
    const int t_stop = (t_end + U - 1) / U * U;
    // Issue priority.  The SIMD's arbiter serves the oldest wavefront first: of four wavefronts that start a sweep together the oldest
    // runs at its own full rate and finishes at 0.4 of the time the youngest needs (measured, -DPHMM_TIMING: 132 against 318 us on
    // configs[1]), and what is left runs two, then one to a SIMD, at a third of the issue rate.  So a wavefront starts at priority 3 and
    // steps down at every quarter of its sweep: whoever is ahead yields, and the four stay within a quarter of each other.
    // (a.fair picks the three thresholds, in 32nds of the sweep)
    const int f1 = a.fair == 1 ? 8 : a.fair == 2 ? 16 : a.fair == 3 ? 20 : a.fair == 4 ? 24 : a.fair == 5 ? 28 : 16;
    const int f2 = a.fair == 1 ? 16 : a.fair == 2 ? 24 : a.fair == 3 ? 26 : a.fair == 4 ? 28 : a.fair == 5 ? 30 : 22;
    const int f3 = a.fair == 1 ? 24 : a.fair == 2 ? 28 : a.fair == 3 ? 30 : a.fair == 4 ? 30 : a.fair == 5 ? 31 : 28;
    const int q1 = a.fair ? (t_stop * f1 / 32 + U - 1) / U * U : -1, q2 = a.fair ? (t_stop * f2 / 32 + U - 1) / U * U : -1, q3 = a.fair ? (t_stop * f3 / 32 + U - 1) / U * U : -1;
    if (a.fair) __builtin_amdgcn_s_setprio(3);
    // (all of this in front of the first hand-issued load: between those loads and the loop there must be no compiled code, which
    // takes their destination registers for written -- the empty statement pins the three thresholds, as scalars, to this point)
    asm volatile("" ::"s"(q1), "s"(q2), "s"(q3));
    {
      // step 0's values (its slab index read and awaited on the spot), then the steady-state order: U, L0 .. L(QT-1)
      unsigned b0;
      asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b0) : "v"(hs_a));
      asm volatile("v_cmp_eq_u32 vcc, %2, %1\n\ts_or_b32 %0, vcc_lo, vcc_hi" : "=s"(any_b) : "v"(b0), "s"(nchar_s) : "vcc", "scc");
      asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(addr_next) : "v"(b0), "s"(slab_s), "v"(tab_a));
      asm volatile("ds_read_u8 %0, %1 offset:1" : "=v"(o1n) : "v"(hs_a));
      load_all_quads<K>(dq, addr_next, tail_adj, std::make_integer_sequence<int, NLD>{});
    }
    // U steps per loop iteration, in one straight line (a taken branch costs a wave some 30 cycles: with a loop back edge, a
    // bubble test and a "next bubble" test per step the single-step form of this loop lost 15 %).  The trip count is rounded up:
    // the steps past t_end run over the stream's padding after the terminal bubble and write nothing.
#ifdef PHMM_TIMING
    const unsigned long long tm1 = __builtin_amdgcn_s_memtime();
#endif
    unsigned hs_cur = hs_a;            // LDS address of this lane's stream byte of step t: the ONE register the stream takes in the sweep
    while (t < t_stop) {
      if (__builtin_expect(t == q1, 0)) __builtin_amdgcn_s_setprio(2);
      if (__builtin_expect(t == q2, 0)) __builtin_amdgcn_s_setprio(1);
      if (__builtin_expect(t == q3, 0)) __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int u = 0; u < U; u++) {
        // the stream byte issued one step ago has landed: address of the next step's slab and its bubble flag, next byte on its way
        // (two statements: with a vector and a scalar output in one the compiler no longer takes the scalar for uniform and tests it
        // with a v_cmp and an EXEC mask)
        // (the wait also covers the first quad of dist values, the load right behind the byte: NLD - 1 younger ones may be in flight)
        asm volatile("s_waitcnt lgkmcnt(%2)\n\tv_cmp_eq_u32 vcc, %3, %1\n\ts_or_b32 %0, vcc_lo, vcc_hi" : "=s"(any_next) : "v"(o1n), "n"(NLD - 1), "s"(nchar_s) : "vcc", "scc");
        asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(addr_next) : "v"(o1n), "s"(slab_s), "v"(tab_a));
        asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(o1n) : "v"(hs_cur), "n"(u + 2));
        const T contrib = AsmCol<T, K>::template column<LPP, XF>(s, dq, addr_next, tail_adj);
        const T acc_done = s.acc;            // the running sum up to the step before: what a bubble lane reports
        s.acc = s.acc + contrib;             // (added before the branch: a bubble lane overwrites the sum, there is no second value of contrib to merge)
        if (__builtin_expect(any_b != 0, 0)) {
          // Off the hot path: first let every load this wave has in flight land.  The code below is the compiler's, which takes
          // the asm loads' destination registers for written the moment the asm statement is through -- on the loop's exit path,
          // where they are dead, it reuses one as an address register while the load is still on its way (tools/check_phmm_asm.py).
          // (the step index and the lane's haplotype counter pass through the statement, so that nothing below -- all of it hangs on
          // one of the two -- is scheduled in front of the wait)
          // (... and the lane's own stream byte of this step is fetched again by the same statement, from the sweep's own address
          // register: read through the `hs` pointer it cost two more registers, live across the whole loop)
          unsigned mine;
          asm volatile("; bubble step\n\ts_waitcnt lgkmcnt(0)\n\tds_read_u8 %0, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(mine), "+v"(jl) : "v"(hs_cur), "n"(u));
          // Some lane (one per group) is on a bubble = column 0 of its next haplotype.  Everybody ran the ordinary column; that
          // lane now overwrites its state with the column-0 border (M = X = 0, Y = 0, clones of row 0: Y = INIT/H;
          // baseline_impl.cpp:60-70) under EXEC.  Which lane: the one whose own stream byte of this step is the marker (read again
          // here, off the hot path).
          if (mine == nchar_s) {
            if (l == LPP - 1 && jl >= 0 && have) {                                       // haplotype jl is complete
              // The read's output row and index come from the LDS stash, the zero of the 64-bit index and the flag's 1 are made right
              // here (opaque to the compiler, which otherwise keeps each of them, and the flag's address, in a register across the
              // whole sweep: six registers of a K = 13 wavefront that needs to stay within 128 to run four per SIMD).
              unsigned lid, zero, one;
              asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 1" : "=&v"(lid), "=v"(zero), "=v"(one));
              const uint2 st = stash[lid / LPP];
              const unsigned long long oi = ((unsigned long long)zero << 32) | (unsigned)(st.x + hcol[jl]);
              a.out[oi] = acc_done;
              if (RESCUE && acc_done < (T)PHMM_F64_TINY) tiny = true;
              if (!RESCUE && sizeof(T) == 4 && a.read_flag && acc_done < (T)PHMM_MIN_ACCEPTED) a.read_flag[((unsigned long long)zero << 32) | st.y] = one;
            }
            jl++;
            const T y0 = y0s[jl];
#pragma unroll
            for (int k = 0; k < K; k++) {
              s.M[k] = T(0); s.X[k] = T(0);
              // (five-operation form: a clone's Ys is INIT/H itself; else Y rides pre-multiplied by the consumer row's pGM)
              s.Y[k] = (k < s.npad) ? (X5 ? y0 : y0 * (X6 ? s.gclone : (k + 1 < K ? s.pGM[k + 1] : s.nGM))) : T(0);
            }
            s.x_out = T(0);
            s.acc = T(0);
          }
        }
        any_b = any_next;
      }
      t += U;
      hs_cur += U;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(o1n));     // the loads issued for the step behind the last one (see DistRegs::keep)
    dq.keep();
#ifdef PHMM_TIMING
    if (lane == 0) { const unsigned long long tm2 = __builtin_amdgcn_s_memtime(); atomicAdd(&g_phmm_t[0], 1ull); atomicAdd(&g_phmm_t[1], tm1 - tm0); atomicAdd(&g_phmm_t[2], tm2 - tm1); atomicAdd(&g_phmm_t[3], tmA - tm0); atomicAdd(&g_phmm_t[4], tmB - tmA); atomicAdd(&g_phmm_t[5], tmC - tmB); atomicAdd(&g_phmm_t[6], tm1 - tmC); const unsigned long long tw1 = wall_clock64(); atomicAdd(&g_phmm_t[7], tw1 - tw0); atomicMin(&g_phmm_w[0], tw0); atomicMax(&g_phmm_w[1], tw0); atomicMin(&g_phmm_w[2], tw1); atomicMax(&g_phmm_w[3], tw1);
      unsigned hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
      const unsigned key = ((xcc & 7u) << 12) | (((hw >> 8) & 0xFFu) << 4) | (((hw >> 6) & 3u) << 2) | ((hw >> 4) & 3u);   // xcc : se,sh,cu : pipe : simd  -> simd in the low 2 bits
      atomicAdd(&g_phmm_simd_n[key & 32767u], 1u); atomicAdd(&g_phmm_simd_t[key & 32767u], tw1 - tw0); }
#endif
    return __any(tiny);
  }
  T dn[K];                    // dist of step t (loaded one step ahead)
  load_dist<T, K>(tab_lane, hs[0] * SLAB, dn);
  constexpr int U = 8;
  while (t < t_end) {
    if (!STRIPED && rm == 0 && nb - t >= U) {          // no lane is on a bubble for the next U steps
#pragma unroll
      for (int u = 0; u < U; u++) {
        T d[K];
#pragma unroll
        for (int k = 0; k < K; k++) d[k] = dn[k];
        load_dist<T, K>(tab_lane, o1 * SLAB, dn);
        o1 = hs[t + u + 2];
        column<STRICT, LPP>(s, d);
      }
      t += U;
      continue;
    }
    const bool nbit = (t == nb);
    if (nbit) { jn++; nb = __builtin_amdgcn_readfirstlane((int)bpos[jn]); }
    rm = ((rm << 1) | (nbit ? 1ull : 0ull)) & (LPP == 64 ? ~0ull : ((1ull << (LPP & 63)) - 1));
    T d[K];
#pragma unroll
    for (int k = 0; k < K; k++) d[k] = dn[k];
    load_dist<T, K>(tab_lane, o1 * SLAB, dn);
    o1 = hs[t + 2];
    T ca = T(0), cx = T(0);
    const bool carry_in = STRIPED && stripe > 0 && l == 0;
    if (carry_in) { ca = carry_a[t]; cx = carry_x[t]; }
    if (!STRIPED && rm == 0) { column<STRICT, LPP>(s, d); t++; continue; }     // the last few columns in front of a bubble
    // Some lane (one per row of 16) is on a bubble = column 0 of its next haplotype.  Everybody runs the
    // ordinary column; that lane then overwrites its state with the column-0 border
    // (M = X = 0, Y = 0, clones of row 0: Y = INIT/H; baseline_impl.cpp:60-70) under EXEC.
    const bool rst = (rm >> l) & 1ull;
    T acc_done = s.acc;
    column<STRICT, LPP>(s, d, carry_in, ca, cx);
    if (rst) {
      if (l == LPP - 1 && jl >= 0 && have && last_stripe) {                          // haplotype jl is complete
        a.out[out_base + hcol[jl]] = acc_done;
        if (RESCUE && !STRICT && acc_done < (T)PHMM_F64_TINY) tiny = true;
        if (!RESCUE && sizeof(T) == 4 && a.read_flag && acc_done < (T)PHMM_MIN_ACCEPTED) a.read_flag[ridx] = 1u;
      }
      jl++;
      const T y0 = y0s[jl];
#pragma unroll
      for (int k = 0; k < K; k++) {
        s.M[k] = T(0); s.X[k] = T(0);
        const T yk = STRICT ? y0 : y0 * (k + 1 < K ? s.pGM[k + 1] : s.nGM);
        s.Y[k] = (k < s.npad) ? yk : T(0);
      }
      s.x_out = T(0);
      s.acc = T(0);
    }
    if (STRIPED && !last_stripe && l == LPP - 1 && t >= LPP - 1) { carry_a[t - (LPP - 1)] = s.a_out; carry_x[t - (LPP - 1)] = s.x_out; }
    t++;
  }
  if (STRIPED && !last_stripe) {            // the next stripe rebuilds the dist table this one read, and reads the carry this one wrote
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  }   // stripes
  return __any(tiny);
}

// Four wavefronts per SIMD (128 registers) is what the five-operation sweep wants up to K = 13 (its loop takes 8 K + 16); the
// prologue's batched loads would take a few more if the compiler were not told to stay within that.
// (fp64, five operations, K = 8 -- the largest rescue class: 171 registers left two wavefronts per SIMD, three fit from 168 down)
template <typename T, int K, int XF> constexpr int phmm_min_waves() { return (sizeof(T) == 4 && XF == 5 && K <= 13) ? 4 : (sizeof(T) == 8 && XF == 5 && K == 8) ? 3 : 1; }
template <typename T, int K, int LPP, bool STRICT, bool RESCUE, int XF = 0, bool STRIPED = false, int W = 1>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(phmm_min_waves<T, K, XF>()))) void phmm_kernel(PhmmArgs<T> a, uint32_t work_base, uint32_t n_work) {
  if (!RESCUE && a.zero_words && blockIdx.x == 0)          // see PhmmArgs::zero_words
    for (int i = threadIdx.x; i < a.n_zero; i += 64 * W) a.zero_words[i] = 0u;
  if constexpr (W > 1 && RESCUE) {
    // Rescue jobs in pairs (phmm_rescue_plan with PhmmPlanArgs::pairs): items 2 i and 2 i + 1 hold the same reads against two runs of
    // haplotypes and share one dist table, which is most of a job's LDS -- an fp64 table of K = 8 is 20 KB, and one per wavefront
    // left seven wavefronts on a CU.  Both wavefronts of a workgroup walk the pairs with the grid's stride.
    const uint32_t n_dev = __builtin_amdgcn_readfirstlane(*a.job_count);
    const uint32_t n = (n_dev < n_work ? n_dev : n_work) & ~1u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x >> 6));
    for (uint32_t i = blockIdx.x * 2u; i < n; i += gridDim.x * 2u) {
      const uint32_t job = i + wave;
      const bool tiny = phmm_job<T, K, LPP, STRICT, RESCUE, XF, STRIPED, W>(a, work_base, job, !a.is_redo);
      __syncthreads();                                       // the next pair rebuilds the table both wavefronts still read
      if (!STRICT && tiny && a.redo_count && (threadIdx.x & 63) == 0) a.redo_list[atomicAdd(a.redo_count, 1u)] = job;
    }
    return;
  }
  if constexpr (W > 1) {
    // (the launch's first wavefront times its own job on the shader clock and on the constant-rate wall clock: the clock the
    // card holds UNDER THIS KERNEL, which a bench line quotes next to the fraction of the issue roof)
    const bool probe = !RESCUE && a.clock_out && blockIdx.x == 0 && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0;   // (uniform: the ticks stay scalar)
    unsigned c0 = 0, w0 = 0;                   // (low words: a job lasts well under the two seconds they wrap in)
    if (probe) { c0 = (unsigned)__builtin_amdgcn_s_memtime(); w0 = (unsigned)wall_clock64(); }
    phmm_job<T, K, LPP, STRICT, RESCUE, XF, STRIPED, W>(a, work_base, blockIdx.x * W + (threadIdx.x >> 6));
    if (probe && threadIdx.x == 0) { a.clock_out[0] = (unsigned)__builtin_amdgcn_s_memtime() - c0; a.clock_out[1] = (unsigned)wall_clock64() - w0; }
    return;
  }
  if (RESCUE && a.job_count) {
    // the number of jobs is only known on the device (phmm_rescue_plan); the grid is capped on the host and every wavefront
    // walks the job array with the grid's stride, so a class with nothing to do costs a few hundred empty wavefronts instead
    // of one per potential job
    // (never beyond the class's slots: the planner counts the jobs it had to drop too, phmm_rescue_plan)
    const uint32_t n_dev = __builtin_amdgcn_readfirstlane(*a.job_count);
    const uint32_t n = n_dev < n_work ? n_dev : n_work;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
      const uint32_t job = a.job_map ? a.job_map[i] : i;
      const bool tiny = phmm_job<T, K, LPP, STRICT, RESCUE, XF, STRIPED>(a, work_base, job, !a.is_redo);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the next job rebuilds the LDS tables this one still read
      __builtin_amdgcn_wave_barrier();
      // The contracted column is within 1e-8 of the reference's order as long as the result stays clear of the denormal range;
      // below PHMM_F64_TINY what gets flushed (x86 FTZ, matched on the device) depends on the last bits of every intermediate and
      // only the reference's own operation order reproduces compute_fp_avxd: such a job is listed and redone that way by the
      // launch that follows (a separate kernel: inlined here the strict column set this kernel's register count).
      if (!STRICT && tiny && a.redo_count && threadIdx.x == 0) a.redo_list[atomicAdd(a.redo_count, 1u)] = job;
    }
  } else {
    const bool probe = !RESCUE && a.clock_out && blockIdx.x == 0;
    unsigned c0 = 0, w0 = 0;
    if (probe) { c0 = (unsigned)__builtin_amdgcn_s_memtime(); w0 = (unsigned)wall_clock64(); }
    phmm_job<T, K, LPP, STRICT, RESCUE, XF, STRIPED>(a, work_base, blockIdx.x);
    if (probe && threadIdx.x == 0) { a.clock_out[0] = (unsigned)__builtin_amdgcn_s_memtime() - c0; a.clock_out[1] = (unsigned)wall_clock64() - w0; }
  }
}

// Several (lanes, K) classes in ONE launch (the prepared five-operation sweep only): the jobs of all classes with K in [KLO, KHI] and 8
// or 16 lanes per read, each wavefront taking the sweep of its own shape (all reads of a wavefront share it: PhmmRowRecs::shape).
// Why: every launch ends in a tail during which the chip runs empty, hardware queues take launches one at a time, and a batch of a
// few hundred regions is nine launches of a few hundred to a few thousand jobs each (a 128-region configs[3] shard: the fp32 phase
// spanned 1.33 ms for 0.86 ms of work).  Registers are those of KHI -- at most 123 up to K = 13, so the merged launch still runs
// four wavefronts per SIMD.  The windows follow the occupancy classes: K <= 5 fits eight wavefronts per SIMD, K = 6..13 four to seven,
// which the launches pin to four (phmm_host.cpp, pinned_wpc).
template <int LPP, int K, int KHI, int W>
__device__ __forceinline__ void phmm_multi_dispatch(const PhmmArgs<float>& a, uint32_t work_base, uint32_t job, int k) {
  if (k == K) phmm_job<float, K, LPP, false, false, 5, false, W>(a, work_base, job);
  else if constexpr (K < KHI) phmm_multi_dispatch<LPP, K + 1, KHI, W>(a, work_base, job, k);
}
template <int KLO, int KHI, int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(4))) void phmm_kernel_multi(PhmmArgs<float> a, uint32_t work_base, uint32_t n_work) {
  static_assert(KLO >= 2 && KHI <= 13, "K classes within the 128-register budget");
  if (a.zero_words && blockIdx.x == 0)
    for (int i = threadIdx.x; i < a.n_zero; i += 64 * W) a.zero_words[i] = 0u;
  const uint32_t job = W > 1 ? blockIdx.x * W + (threadIdx.x >> 6) : blockIdx.x;
  const bool probe = a.clock_out && blockIdx.x == 0 && (W == 1 || __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0);
  unsigned c0 = 0, w0 = 0;
  if (probe) { c0 = (unsigned)__builtin_amdgcn_s_memtime(); w0 = (unsigned)wall_clock64(); }
  // (the first read of a work item always exists; with W = 2 both items of a pair list the same reads)
  const uint32_t rid0 = __builtin_amdgcn_readfirstlane(a.work[work_base + job].read[0]);
  const uint32_t shape = __builtin_amdgcn_readfirstlane(a.rec.shape[rid0]);
  const int K = (int)(shape & 255u);
  if ((shape >> 8) == 16u) phmm_multi_dispatch<16, KLO, KHI, W>(a, work_base, job, K);
  else phmm_multi_dispatch<8, KLO, KHI, W>(a, work_base, job, K);
  if (probe && threadIdx.x == 0) { a.clock_out[0] = (unsigned)__builtin_amdgcn_s_memtime() - c0; a.clock_out[1] = (unsigned)wall_clock64() - w0; }
}
// lds_bytes = the largest request of the merged classes (host: phmm_lds_bytes per class)
template <int W>
hipError_t launch_multi(int k_lo, int k_hi, size_t lds_bytes, const PhmmArgs<float>& a, uint32_t work_base, uint32_t n_work, hipStream_t st) {
  if (n_work == 0) return hipSuccess;
  if (n_work % W != 0) return hipErrorInvalidValue;
  size_t lds = lds_bytes;
  if (lds < (size_t)a.lds_min) lds = (size_t)a.lds_min;
  dim3 grid(n_work / W), block(64 * W);
#define ACCG_MULTI(LO, HI)                                                                                     \
  if (k_lo >= LO && k_hi <= HI) {                                                                              \
    hipLaunchKernelGGL((phmm_kernel_multi<LO, HI, W>), grid, block, lds, st, a, work_base, n_work);             \
    return hipGetLastError();                                                                                  \
  }
  ACCG_MULTI(6, 13) ACCG_MULTI(2, 5)
#undef ACCG_MULTI
  return hipErrorInvalidValue;
}

template <typename T, bool STRICT, bool RESCUE, int XF = 0, int W = 1>
hipError_t launch(int K, int lpp, const PhmmArgs<T>& a, uint32_t work_base, uint32_t n_work, hipStream_t st, bool striped = false,
                  uint32_t grid_cap = PHMM_RESCUE_GRID) {
  if (n_work == 0) return hipSuccess;
  if (W > 1 && (striped || n_work % W != 0)) return hipErrorInvalidValue;
  dim3 grid(RESCUE && a.job_count ? (n_work / W < grid_cap ? n_work / W : grid_cap) : n_work / W), block(64 * W);
  if (striped) {      // reads of 1024 bases and more: 64 lanes x 16 rows per stripe, the generic column
    if (K != 16 || lpp != 64 || XF != 0) return hipErrorInvalidValue;
    size_t lds = phmm_lds_bytes(16, (int)sizeof(T), a.nchar, a.stream_cap, a.haps_cap, 64, false, true);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL((phmm_kernel<T, 16, 64, STRICT, RESCUE, 0, true>), grid, block, lds, st, a, work_base, n_work);
    return hipGetLastError();
  }
#define ACCG_CASE(KK, LL)                                                                                     \
  case KK: {                                                                                                  \
    size_t lds = phmm_lds_bytes(KK, (int)sizeof(T), a.nchar, a.stream_cap, a.haps_cap, LL, phmm_is_compact((int)sizeof(T), STRICT, KK), false, W); \
    if (lds < (size_t)a.lds_min) lds = (size_t)a.lds_min;                                                     \
    hipLaunchKernelGGL((phmm_kernel<T, KK, LL, STRICT, RESCUE, XF, false, W>), grid, block, lds, st, a, work_base, n_work); \
  } break;
#ifdef ACCG_PHMM_DEV_SUBSET      // development builds: only the configs[1] kernels (seconds instead of minutes to compile)
  if (lpp == 8 && K == 13) {
    if constexpr (!RESCUE && sizeof(T) == 4) { switch (K) { ACCG_CASE(13, 8) default: return hipErrorInvalidValue; } return hipGetLastError(); }
  }
  if (lpp == 16 && (K == 2 || K == 7)) { switch (K) { ACCG_CASE(2, 16) ACCG_CASE(7, 16) default: return hipErrorInvalidValue; } return hipGetLastError(); }
  return hipErrorInvalidValue;
#else
  if (lpp == 8) {      // not for the rescue pass, whose jobs are planned on the device in 16/32/64-lane classes
    if constexpr (!RESCUE) {
      switch (K) {
        ACCG_CASE(1, 8) ACCG_CASE(2, 8) ACCG_CASE(3, 8) ACCG_CASE(4, 8) ACCG_CASE(5, 8) ACCG_CASE(6, 8) ACCG_CASE(7, 8) ACCG_CASE(8, 8)
        ACCG_CASE(9, 8) ACCG_CASE(10, 8) ACCG_CASE(11, 8) ACCG_CASE(12, 8) ACCG_CASE(13, 8) ACCG_CASE(14, 8) ACCG_CASE(15, 8) ACCG_CASE(16, 8)
        default: return hipErrorInvalidValue;
      }
    } else return hipErrorInvalidValue;
  } else if (lpp == 16) {
    switch (K) {
      ACCG_CASE(1, 16) ACCG_CASE(2, 16) ACCG_CASE(3, 16) ACCG_CASE(4, 16) ACCG_CASE(5, 16) ACCG_CASE(6, 16) ACCG_CASE(7, 16) ACCG_CASE(8, 16)
      ACCG_CASE(9, 16) ACCG_CASE(10, 16) ACCG_CASE(11, 16) ACCG_CASE(12, 16) ACCG_CASE(13, 16) ACCG_CASE(14, 16) ACCG_CASE(15, 16) ACCG_CASE(16, 16)
      default: return hipErrorInvalidValue;
    }
  } else if (lpp == 32) {      // the sweep's classes for reads of 256 to 1022 bases (phmm_pick) / the rescue's (phmm_rescue_shape)
    if constexpr (RESCUE) {
      switch (K) { ACCG_CASE(5, 32) ACCG_CASE(6, 32) ACCG_CASE(7, 32) ACCG_CASE(8, 32) default: return hipErrorInvalidValue; }
    } else {
      switch (K) { ACCG_CASE(9, 32) ACCG_CASE(10, 32) ACCG_CASE(12, 32) ACCG_CASE(14, 32) ACCG_CASE(16, 32) default: return hipErrorInvalidValue; }
    }
  } else {
    if constexpr (RESCUE) {
      switch (K) { ACCG_CASE(5, 64) ACCG_CASE(6, 64) ACCG_CASE(7, 64) ACCG_CASE(8, 64) ACCG_CASE(16, 64) default: return hipErrorInvalidValue; }
    } else {
      switch (K) { ACCG_CASE(9, 64) ACCG_CASE(10, 64) ACCG_CASE(12, 64) ACCG_CASE(14, 64) ACCG_CASE(16, 64) default: return hipErrorInvalidValue; }
    }
  }
  return hipGetLastError();
#endif
#undef ACCG_CASE
}

// ---- merged rescue launches (phmm_dev.h: PHMM_RESCUE_MERGED) ----------------------------------------------------------------------
template <int WIN, int I, int N, int W>
__device__ __forceinline__ bool phmm_rescue_dispatch(const PhmmArgs<double>& a, const PhmmRescueSet& rs, int ci, uint32_t r, uint32_t wave, uint32_t* abs_item) {
  constexpr int CLS = phmm_rescue_win_class(WIN, I);
  if (ci == I) {
    const uint32_t job = r * W + wave;
    *abs_item = rs.off[CLS] + job;
    return phmm_job<double, phmm_rescue_k(CLS), phmm_rescue_lpp(CLS), false, true, 5, false, W>(a, rs.off[CLS], job, true);
  }
  if constexpr (I + 1 < N) return phmm_rescue_dispatch<WIN, I + 1, N, W>(a, rs, ci, r, wave, abs_item);
  return false;
}
template <int WIN, int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(WIN == 0 ? 4 : 3))) void phmm_rescue_multi(PhmmArgs<double> a, PhmmRescueSet rs) {
  constexpr int N = WIN == 0 ? PHMM_RESCUE_WIN0_N : PHMM_RESCUE_WIN1_N;
  // units (items, or pairs of items) per class of the window, in the window's order; never beyond a class's slots.  (Recomputed from the
  // counts for every unit -- a handful of scalar loads -- rather than held across the sweeps, where every register counts.)
  auto units_of = [&](int i) {
    const int c = phmm_rescue_win_class(WIN, i);
    const uint32_t cap = rs.off[c + 1] - rs.off[c], n = __builtin_amdgcn_readfirstlane(rs.counts[c]);
    return (n < cap ? n : cap) / (uint32_t)W;
  };
  uint32_t total = 0;
#pragma unroll
  for (int i = 0; i < N; i++) total += units_of(i);
  const uint32_t wave = W > 1 ? __builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x >> 6)) : 0u;
  for (uint32_t u = blockIdx.x; u < total; u += gridDim.x) {
    uint32_t r = u;
    int ci = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
      const uint32_t n_i = units_of(i);
      if (ci == i && r >= n_i) { r -= n_i; ci = i + 1; }
    }
    uint32_t abs_item = 0;
    const bool tiny = phmm_rescue_dispatch<WIN, 0, N, W>(a, rs, ci, r, wave, &abs_item);
    if constexpr (W > 1) __syncthreads();                     // the next pair rebuilds the table both wavefronts still read
    else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    if (tiny && a.redo_count && (threadIdx.x & 63) == 0) a.redo_list[atomicAdd(a.redo_count, 1u)] = abs_item;
  }
}
template <int C>
__device__ __forceinline__ void phmm_redo_dispatch(const PhmmArgs<double>& a, const PhmmRescueSet& rs, uint32_t item) {
  if (item >= rs.off[C] && item < rs.off[C + 1]) {
    phmm_job<double, phmm_rescue_k(C), phmm_rescue_lpp(C), true, true, 0, false, 1>(a, 0u, item, false);
    return;
  }
  if constexpr (C + 1 < PHMM_RESCUE_MERGED) phmm_redo_dispatch<C + 1>(a, rs, item);
}
template <int UNUSED = 0>      // (a template so that only the translation unit that launches it compiles it)
__global__ __launch_bounds__(64) void phmm_redo_multi(PhmmArgs<double> a, PhmmRescueSet rs) {
  const uint32_t n_dev = __builtin_amdgcn_readfirstlane(*a.redo_count), cap = rs.off[PHMM_RESCUE_MERGED];
  const uint32_t n = n_dev < cap ? n_dev : cap;
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
    phmm_redo_dispatch<0>(a, rs, __builtin_amdgcn_readfirstlane(a.redo_list[i]));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}
template <int UNUSED = 0>
hipError_t launch_rescue_multi(int window, int wg, size_t lds, const PhmmArgs<double>& a, const PhmmRescueSet& rs, uint32_t grid, hipStream_t st) {
  if (grid == 0) return hipSuccess;
  if ((window != 0 && window != 1) || (wg != 1 && wg != 2) || lds > 160 * 1024) return hipErrorInvalidValue;
  if (window == 0) {
    if (wg == 2) hipLaunchKernelGGL((phmm_rescue_multi<0, 2>), dim3(grid), dim3(128), lds, st, a, rs);
    else hipLaunchKernelGGL((phmm_rescue_multi<0, 1>), dim3(grid), dim3(64), lds, st, a, rs);
  } else {
    if (wg == 2) hipLaunchKernelGGL((phmm_rescue_multi<1, 2>), dim3(grid), dim3(128), lds, st, a, rs);
    else hipLaunchKernelGGL((phmm_rescue_multi<1, 1>), dim3(grid), dim3(64), lds, st, a, rs);
  }
  return hipGetLastError();
}
template <int UNUSED = 0>
hipError_t launch_redo_multi(size_t lds, const PhmmArgs<double>& a, const PhmmRescueSet& rs, uint32_t grid, hipStream_t st) {
  if (grid == 0) return hipSuccess;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL((phmm_redo_multi<0>), dim3(grid), dim3(64), lds, st, a, rs);
  return hipGetLastError();
}

// Per-row records of the five-operation sweep (PhmmRowRecs, phmm_dev.h): one block of 64 threads per read, once per pass.  The
// arithmetic -- which float is multiplied with and divided by which, in which order -- is the prologue's own of the kernels that do
// not use the records (phmm_job, the X5 block), so both give the same bits (oracle model: orc_phmm_forward_f32_fma5).
__global__ __launch_bounds__(128) void phmm_prepare_rows(PhmmArgs<float> a, uint32_t n_reads, uint32_t* state, uint32_t state_words) {
  if (state) for (uint32_t i = blockIdx.x * 128u + threadIdx.x; i < state_words; i += gridDim.x * 128u) state[i] = 0u;
  const uint32_t rid = blockIdx.x;
  if (rid >= n_reads) return;
  const uint32_t shape = a.rec.shape[rid];
  if (shape == 0) return;                                   // this read's wavefront runs another form of the sweep
  const int K = (int)(shape & 255u), LPP = (int)(shape >> 8), SROWS = K * LPP;
  const SeqRef rr = a.rd[rid];
  const int R = (int)rr.len, pad = SROWS - R;               // rows are right-aligned: the first `pad` flat rows are clones of row 0
  const uint8_t* rb = a.rblob + rr.off;
  const uint32_t row0 = a.rec.row0[rid];
  // One thread per flat row, everything it needs loaded by itself: the row's five bytes, the next row's three qualities and the
  // previous row's insertion quality in one round trip, the table entries they select in a second (no exchange between threads).
  for (int f = threadIdx.x; f < SROWS; f += 128) {
    const int l = f / K, k = f - l * K, r = f - pad;
    const uint32_t at = row0 + (uint32_t)(k * LPP + l);
    auto tri = [](int x, int y) { const int lo = x < y ? x : y, hi = x < y ? y : x; return ((hi * (hi + 1)) >> 1) + lo; };
    if (r < 0) {                                            // a clone of row 0; the last one hands Y = INIT/H to the first row
      float bclone = 0.f;
      if (r == -1 && R > 0) {
        const float m1 = a.tab.m2m[tri(rb[2 * R] & 127, rb[3 * R] & 127)], g1 = a.tab.omph[rb[4 * R] & 127];
        bclone = m1 != 0.f ? (1.0f * g1) / m1 : 0.f;
      }
      a.rec.coef[at] = make_float4(0.f, bclone, 1.f, 0.f);
      a.rec.dist[at] = make_float4(0.f, 0.f, 0.f, 0.f);
      a.rec.misc[at] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    const int rn = r + 1 < R ? r + 1 : r, rp = r > 0 ? r - 1 : r;
    const int base = char_index(rb[r]), qq = rb[R + r] & 127, qi = rb[2 * R + r] & 127, qd = rb[3 * R + r] & 127, qc = rb[4 * R + r] & 127;
    const int ni = rb[2 * R + rn] & 127, nd = rb[3 * R + rn] & 127, nc = rb[4 * R + rn] & 127, pi = rb[2 * R + rp] & 127;
    const float pMM = a.tab.m2m[tri(qi, qd)], pMX = a.tab.ph[qi], pXX = a.tab.ph[qc], pMY = a.tab.ph[qd];
    const float nMM = a.tab.m2m[tri(ni, nd)], nGM = a.tab.omph[nc], qMX = a.tab.ph[pi];
    const float dM = a.tab.omph[qq] * pMM, dX = a.tab.phd3[qq] * pMM;      // baseline_impl.cpp:79-83, times the row's pMM
    float ca = 0.f, cb = 0.f, cx = 0.f;
    if (r + 1 < R) {
      ca = nMM != 0.f ? (pMX * nGM) / nMM : 0.f;
      cb = nMM != 0.f ? (pMY * nGM) / nMM : 0.f;
    }
    if (r > 0) cx = pMX != 0.f ? (pXX * qMX) / pMX : 0.f;
    a.rec.coef[at] = make_float4(ca, cb, pXX, cx);
    a.rec.dist[at] = make_float4(base == CH_N || base == CH_A ? dM : dX, base == CH_N || base == CH_C ? dM : dX,
                                 base == CH_N || base == CH_G ? dM : dX, base == CH_N || base == CH_T ? dM : dX);
    a.rec.misc[at] = make_float4(dM, pMX, 0.f, 0.f);
  }
}

// One workgroup per region: list the reads with an fp32 result below MIN_ACCEPTED (host_type.h:21), in the
// region's length order, cut the list into wavefront-sized groups and emit (group x haplotype run) jobs into the
// job array of the group's class.  The fp64 kernel then drops the haplotypes none of the group's reads needs.
__global__ __launch_bounds__(256) void phmm_rescue_plan(PhmmPlanArgs p) {
  __shared__ uint32_t s_base, s_cnt;
  __shared__ uint32_t s_wave[4];
  const PhmmRegionDev R = p.regions[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // nothing flagged in this region (the usual case for reads that belong to their haplotypes): out after one sweep over the flags
  __shared__ uint32_t s_any;
  if (tid == 0) { s_base = 0; s_any = 0; }
  __syncthreads();
  {
    uint32_t any = 0;
    for (uint32_t i = tid; i < R.n_reads; i += 256) any |= p.read_flag[R.read0 + i];
    if (any) s_any = 1u;
  }
  __syncthreads();
  if (!s_any) return;
  if (tid == 0 && p.host_flag) __hip_atomic_store(p.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  uint32_t* out = p.flagged + R.read0;
  for (uint32_t i0 = 0; i0 < R.n_reads; i0 += 256) {          // ordered compaction, 256 reads at a time
    const uint32_t i = i0 + tid;
    bool f = false;
    uint32_t rid = 0;
    if (i < R.n_reads) {
      rid = p.sorted_reads[R.read0 + i];
      f = p.read_flag[rid] != 0;
      if (f) p.read_flag[rid] = 0u;            // read and cleared: the next pass's sweep finds its flags at zero
    }
    const unsigned long long m = __ballot(f);
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = s_base;
    for (int w = 0; w < wave; w++) before += s_wave[w];
    if (f) out[before + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = rid;
    __syncthreads();
    if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  const uint32_t nf = s_base;
  if (nf == 0) return;
  // groups: the first (longest) read of a group decides its class; `per` reads per wavefront
  for (uint32_t i = 0; i < nf;) {
    int cls, lpp, K;
    phmm_rescue_class(p.rd[out[i]].len, &cls, &lpp, &K);
    const uint32_t per = 64u / (uint32_t)lpp;
    // (pairs: an even number of items per group, the last one empty when the region has an odd number of haplotype runs)
    const uint32_t n_items = p.pairs ? (R.n_chunks + 1u) & ~1u : R.n_chunks;
    if (tid == 0) s_cnt = atomicAdd(&p.counts[cls], n_items);
    __syncthreads();
    const uint32_t base = s_cnt;
    for (uint32_t c = tid; c < n_items; c += 256) {
      PhmmWork w;
      for (uint32_t g = 0; g < PHMM_GROUPS; g++) w.read[g] = (g < per && i + g < nf) ? out[i + g] : PHMM_NO_READ;
      w.hap_off = c < R.n_chunks ? p.chunks[R.chunk0 + c].ids0 : 0u; w.n_haps = c < R.n_chunks ? p.chunks[R.chunk0 + c].n : 0u; w.pad_[0] = w.pad_[1] = 0;
      if (base + c < p.class_off[cls + 1] - p.class_off[cls]) p.jobs[(size_t)p.class_off[cls] + base + c] = w;
    }
    __syncthreads();
    i += per;
  }
}

}  // namespace
}  // namespace accg

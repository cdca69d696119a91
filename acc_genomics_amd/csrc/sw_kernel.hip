// HTC (GATK SWPairwiseAlignment) Smith-Waterman fill + end-cell selection for gfx950, hand-written HIP.
//
// What it computes: the matrix of calculateMatrixOneBatch (htc-sw/host/FalconSW_AVX.cpp:1693-1823)
//     diag = H[i-1][j-1] + (ref[i]==alt[j] ? w_match : w_mismatch)
//     V[i][j]  = max(V[i-1][j]  + w_extend, H[i-1][j] + w_open)      (best_gap_v[j], :1772-1779)
//     Hh[i][j] = max(Hh[i][j-1] + w_extend, H[i][j-1] + w_open)      (best_gap_h[i], :1784-1793)
//     H[i][j]  = max(diag, Hh, V)                                     (:1797-1810; the cutoff -1e8 never binds)
// with H[0][*] = H[*][0] = 0, or open+(k-1)*extend for the INDEL / LEADING_INDEL strategies (:1732-1746),
// followed by the end-cell rule of calculateCigarOneBatch (:2314-2339).  Outputs per pair: the score
// sw[p1][p2] and the cell (p1, p2) -- the quantities BASELINE.json requires bit-exact.
//
// Mapping: a pair lives in one DPP row (16 lanes).  One of its two sequences (the "lane sequence",
// the shorter one) is spread over the lanes, K consecutive positions per lane in registers,
// right-aligned so that its last position is always (lane 15, k = K-1); the other (the "sweep
// sequence") is streamed from LDS one position per step, skewed one position per lane, so the only
// cross-lane traffic is a DPP row_shr:1 of two values per step (H and the lane-direction gap of the
// lane's last position).  Positions in front of the sequence are border clones that reproduce
// H[.][0] out of the same recurrence (score 0, or -inf with a -inf hand-off from the left under the prefill strategies), so there is no border special
// case in the loop.  In 16-bit mode two pairs share a row in the lo/hi halves of every register
// (v_pk_add_i16 clamp / v_pk_max_i16), so a wavefront works on 8 pairs.  Lanes are switched off
// (EXEC) before their first and after their last sweep position, which freezes exactly the values
// the end-cell rule needs: the registers end up holding H[.][last sweep position], and the last
// lane's last position is logged to LDS every step.  No MFMA: integer max-plus recurrence.
#include <stdlib.h>
#include <algorithm>
#include "sw_dev.h"

namespace accg {
namespace {

typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short u2 __attribute__((ext_vector_type(2)));

constexpr int NEG16 = -32768;
constexpr int NEG32 = -1073741824;   // FalconSW_AVX.cpp:1700 lowInitValue

// ---- value abstraction: packed 2 x int16 (saturating) or int32 ------------------------------------
template <bool P16> struct Val;
template <> struct Val<true> {
  typedef s2 T;
  static __device__ __forceinline__ T splat(int v) { T r = {(short)v, (short)v}; return r; }
  static __device__ __forceinline__ T make(int lo, int hi) { T r = {(short)lo, (short)hi}; return r; }
  static __device__ __forceinline__ T adds(T a, T b) { return __builtin_elementwise_add_sat(a, b); }
  static __device__ __forceinline__ T mx(T a, T b) { return __builtin_elementwise_max(a, b); }
  static __device__ __forceinline__ T score(T a, T b, T wm, T wd) {   // a == b ? wm : wm + wd, per half
    // three packed ops; written as asm because hipcc otherwise scalarises the equality into
    // 2 x (v_cmp_ne_u16 + v_cndmask) + v_perm + add per row (measured: 18 instead of 12 VALU ops per cell pair)
    int r;
    asm("v_xor_b32 %0, %1, %2\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]\n\tv_pk_mad_i16 %0, %0, %3, %4"
        : "=&v"(r)
        : "v"(__builtin_bit_cast(int, a)), "v"(__builtin_bit_cast(int, b)), "v"(__builtin_bit_cast(int, wd)),
          "v"(__builtin_bit_cast(int, wm)));
    return __builtin_bit_cast(T, r);
  }
  static __device__ __forceinline__ int bits(T a) { return __builtin_bit_cast(int, a); }
  static __device__ __forceinline__ T from_bits(int b) { return __builtin_bit_cast(T, b); }
  static __device__ __forceinline__ int get(T a, int half) { return half ? (int)a.y : (int)a.x; }
  // plane with (a < b) of both halves shifted in: the sign bits of the saturating difference go to bits 15 and 31, what was
  // there moves down by one (a 32-bit shift is enough: at most 16 bits are ever collected, so nothing crosses into the other
  // half).  v_pk_sub_i16 clamp + v_lshrrev_b32 + v_and_or_b32: one half-rate packed operation and two full-rate ones per
  // plane and cell pair; the earlier v_pk_lshrrev_b16 + shift + or form had two packed ones and 2.1 others.
  static __device__ __forceinline__ unsigned push_lt(unsigned plane, T a, T b) {
    const unsigned d = __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(a, b));
    return (plane >> 1) | (d & 0x80008000u);
  }
  // where the bit of local row k ends up after all K rows of a step went in
  static __device__ __forceinline__ int plane_bit(int K, int k, int half) { return 16 - K + k + 16 * half; }
  enum { NEG = NEG16 };
};
template <> struct Val<false> {
  typedef int T;
  static __device__ __forceinline__ T splat(int v) { return v; }
  static __device__ __forceinline__ T make(int lo, int) { return lo; }
  static __device__ __forceinline__ T adds(T a, T b) { return a + b; }
  static __device__ __forceinline__ T mx(T a, T b) { return a > b ? a : b; }
  static __device__ __forceinline__ T score(T a, T b, T wm, T wd) { return a == b ? wm : wm + wd; }
  static __device__ __forceinline__ int bits(T a) { return a; }
  static __device__ __forceinline__ T from_bits(int b) { return b; }
  static __device__ __forceinline__ int get(T a, int) { return a; }
  static __device__ __forceinline__ unsigned push_lt(unsigned plane, T a, T b) { return (plane << 1) | ((unsigned)(a - b) >> 31); }   // |values| < 2^30 + 2^20
  static __device__ __forceinline__ int plane_bit(int K, int k, int) { return K - 1 - k; }
  enum { NEG = NEG32 };
};

// lane l <- lane l-1 inside a group of LPP lanes; the first lane of every group receives OLD.
// 16 lanes = one DPP row (row_shr:1); 32/64 lanes: wave_shr:1 plus one select for lane 32.
template <int LPP>
__device__ __forceinline__ int group_shr1_var(int old, int v, bool leader) {   // same with a run-time value for the first lane
  if (LPP == 16) return __builtin_amdgcn_update_dpp(old, v, 0x111, 0xF, 0xF, false);
  const int r = __builtin_amdgcn_update_dpp(old, v, 0x138, 0xF, 0xF, false);
  return (LPP == 32 && leader) ? old : r;
}
template <int OLD, int LPP>
__device__ __forceinline__ int group_shr1_old(int v, bool leader) {
  if (LPP == 16) return __builtin_amdgcn_update_dpp(OLD, v, 0x111, 0xF, 0xF, false);
  const int r = __builtin_amdgcn_update_dpp(OLD, v, 0x138, 0xF, 0xF, false);
  return (LPP == 32 && leader) ? OLD : r;
}

__device__ __forceinline__ int iabs(int x) { return x < 0 ? -x : x; }

// Candidate ordering of the end-cell rule (FalconSW_AVX.cpp:2320-2337) as one key: higher score wins;
// then the smaller distance to the main diagonal; then the earlier candidate (the last-column
// winner comes first, then bottom-row cells in increasing j).
__device__ __forceinline__ unsigned long long cand_key(int score, int dist, int order) {
  return ((unsigned long long)(unsigned)(score + 0x40000000) << 24) | ((unsigned long long)(4095 - dist) << 12) |
         (unsigned long long)(4095 - order);
}
template <int LPP>
__device__ __forceinline__ unsigned long long group_max_u64(unsigned long long v) {   // max over the LPP lanes of a group
#pragma unroll
  for (int m = 1; m < LPP; m <<= 1) {
    unsigned lo = __shfl_xor((unsigned)v, m, LPP), hi = __shfl_xor((unsigned)(v >> 32), m, LPP);
    unsigned long long o = ((unsigned long long)hi << 32) | lo;
    v = o > v ? o : v;
  }
  return v;
}

// Backtrace record (BT = true).  Every (step t, lane) stores one uint4 = four bit planes; plane bit
// Val<P16>::plane_bit(K, k, half) belongs to the cell (lane position of local row k, sweep index t - lane):
//   x: lane-direction gap OPENED here  (open > extension, strictly: FalconSW_AVX.cpp:1774 / :1786)
//   y: sweep-direction gap OPENED here
//   z: cell is NOT diagonal            (diag >= down && diag >= right fails, :1798)
//   w: DOWN wins over RIGHT            (right >= down fails, :1803)
// With the planes, btrack's +-k (:1805-1809) is the run length of not-opened cells walked by sw_trace.
// MASK (BT, packed int16, 16 lanes per pair): the record as lane masks through scalar stores (SwArgs::bt_masks).  What it saves: the
// arithmetic extraction is a half-rate packed subtract and two full-rate operations per plane and row (as much issue time as the fill
// itself); a v_cmp_lt_i16_sdwa per plane, row and half costs 0.87 of a packed operation and the scalar stores ride the
// scalar-memory pipe for free (tools/ubench_sstore.hip, profiles/r04_ubench_sstore.txt).
template <int K, int LPP, bool P16, bool LANE_IS_ALT, bool BT, bool MASK = false>
__global__ __launch_bounds__(64) void sw_kernel(SwArgs a, uint32_t work_base, uint32_t bt_first, int sweep_cap) {
  static_assert(!MASK || (BT && P16 && LPP == 16), "lane-mask record: packed int16, 16 lanes per pair");
  typedef Val<P16> VT;
  typedef typename VT::T T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // per group: sweep characters (two pairs packed lo|hi<<16) and the log of the last lane position
  // group stride = 16 mod 32 words: the two groups of a 32-lane half then read disjoint banks (ds_read_b32 has 32 banks)
  const int gs = sw_group_stride(sweep_cap);
  uint32_t* sweep_ch = reinterpret_cast<uint32_t*>(smem);                 // [4][gs] (NG <= 4 used)
  int32_t* edge_log = reinterpret_cast<int32_t*>(smem) + 4 * gs;          // [4][gs] (bits of T)

  constexpr int NG = 64 / LPP;                 // pair groups per wavefront
  const int lane = threadIdx.x, g = lane / LPP, l = lane % LPP;
  const SwWork* wp = a.work + (work_base + blockIdx.x);
  const uint32_t pr[2] = {g < NG ? wp->pair[2 * g] : SW_NO_PAIR, (P16 && g < NG) ? wp->pair[2 * g + 1] : SW_NO_PAIR};
  const bool have[2] = {pr[0] != SW_NO_PAIR, pr[1] != SW_NO_PAIR};

  int nl[2] = {0, 0}, ns = 0, prefill[2] = {0, 0}, strat[2] = {0, 0};
  const uint8_t* lseq[2] = {nullptr, nullptr};
  const uint8_t* sseq[2] = {nullptr, nullptr};
#pragma unroll
  for (int h = 0; h < 2; h++)
    if (have[h]) {
      const int rl = a.ref_len[pr[h]], al = a.alt_len[pr[h]];
      const uint8_t* rp = a.refs + (size_t)pr[h] * a.ref_stride;
      const uint8_t* ap = a.alts + (size_t)pr[h] * a.alt_stride;
      nl[h] = LANE_IS_ALT ? al : rl;
      ns = LANE_IS_ALT ? rl : al;              // the two halves of a group have the same sweep length (host)
      lseq[h] = LANE_IS_ALT ? ap : rp;
      sseq[h] = LANE_IS_ALT ? rp : ap;
      strat[h] = a.strategy[pr[h]];
      prefill[h] = (strat[h] == 1 || strat[h] == 2);   // INDEL, LEADING_INDEL: FalconSW_AVX.cpp:1732
    }

  // ---- sweep characters into LDS -----------------------------------------------------------------
  uint32_t* my_ch = sweep_ch + g * gs;
  int32_t* my_log = edge_log + g * gs;
  for (int i = l; i < ns; i += LPP) {
    uint32_t c0 = have[0] ? sseq[0][i] : 0u, c1 = have[1] ? sseq[1][i] : 0u;
    my_ch[i + 1] = c0 | (c1 << 16);
  }

  // ---- per-position constants ---------------------------------------------------------------------
  // position p (1-based along the lane sequence) of half h sits at flat index p - 1 + pad_h; flat < pad_h
  // is a border clone: score 0 (no prefill) or -inf (prefill), lane-direction open = -inf.
  T ch[K], wm[K], wd[K], Hp[K], Ho[K], G[K];   // Ho = H + w_open of the previous sweep index (shared by both gap recurrences)
  const int W_M = a.w_match, W_X = a.w_mismatch, W_O = a.w_open, W_E = a.w_extend;
  const int NEG = VT::NEG;
#pragma unroll
  for (int k = 0; k < K; k++) {
    int c_[2], wm_[2], wd_[2], h_[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int pad = LPP * K - nl[h];
      const int p = l * K + k - pad + 1;       // 1-based position, <= 0: border clone
      if (have[h] && p >= 1) {
        c_[h] = lseq[h][p - 1]; wm_[h] = W_M; wd_[h] = W_X - W_M;
        h_[h] = prefill[h] ? W_O + (p - 1) * W_E : 0;                 // H at sweep index 0
      } else {
        c_[h] = 0x100;                                                 // never equals a base byte
        wm_[h] = prefill[h] ? NEG / 2 : 0; wd_[h] = 0; h_[h] = 0;
      }
    }
    ch[k] = VT::make(c_[0], c_[1]); wm[k] = VT::make(wm_[0], wm_[1]); wd[k] = VT::make(wd_[0], wd_[1]);
    Hp[k] = VT::make(h_[0], h_[1]); G[k] = VT::splat(NEG);
  }
  const T ext = VT::splat(W_E), opn_s = VT::splat(W_O);
#pragma unroll
  for (int k = 0; k < K; k++) Ho[k] = VT::adds(Hp[k], opn_s);
  // What the first lane of a group receives from "the left": H = 0 (its k = 0 is always a border clone whose own H is
  // produced by score 0 / the sweep-direction gap), except under a prefill strategy, where -inf keeps the clone's
  // lane-direction gap at -inf so that the clone reproduces H[i][0] = open + (i-1)*extend exactly.
  const T h_left = VT::make(prefill[0] ? NEG : 0, prefill[1] ? NEG : 0);
  T h_in = h_left, d_in = h_left, f_in = VT::splat(NEG);
  T h_last = Hp[K - 1], f_last = VT::splat(NEG);
  // what the border clone in front of lane 0 hands over: H[i][0] along the sweep
  // (lane 0, k = 0 is always a clone, so only its *inputs* matter: 0 / NEG below)
  int t_end = ns + LPP - 1;
#pragma unroll
  for (int m = LPP; m < 64; m <<= 1) { int o = __shfl_xor(t_end, m); t_end = o > t_end ? o : t_end; }
  t_end = __builtin_amdgcn_readfirstlane(t_end);
  __syncthreads();

  // ---- sweep ---------------------------------------------------------------------------------------
  unsigned dbg_acc = 0;
  // (MASK: the job's block of the record; a step's K x 64 bytes at t * K * 64 -- kept in SGPRs, the stores are scalar)
  const uint64_t rec_job = MASK ? (uint64_t)(a.bt + (uint64_t)(blockIdx.x + work_base - bt_first) * a.bt_item_stride) : 0ull;
  for (int t = 1; t <= t_end; t++) {
    // hand-off from the lane on the left (all lanes, also the switched-off ones: their registers are frozen)
    d_in = h_in;
    h_in = VT::from_bits(group_shr1_var<LPP>(VT::bits(h_left), VT::bits(h_last), l == 0));
    f_in = VT::from_bits(P16 ? group_shr1_old<(int)0x80008000, LPP>(VT::bits(f_last), l == 0)
                             : group_shr1_old<NEG32, LPP>(VT::bits(f_last), l == 0));
    const int i = t - l;                       // this lane's sweep index
    if (i >= 1 && i <= ns) {
      const T c = VT::from_bits((int)my_ch[i]);
      T hup = h_in, hdiag = d_in, f = f_in;
      T fo = VT::adds(h_in, opn_s);                                      // H of the position above + w_open
      unsigned pf = 0, pg = 0, pn = 0, pd = 0;
      const uint64_t rec_step = MASK ? rec_job + (uint64_t)t * (uint64_t)(K * 64) : 0ull;
#pragma unroll
      for (int k = 0; k < K; k++) {
        const T hold = Hp[k];
        const T fe = VT::adds(f, ext);
        f = VT::mx(fe, fo);                                              // gap along the lanes
        const T ge = VT::adds(G[k], ext), go = Ho[k];
        G[k] = VT::mx(ge, go);                                           // gap along the sweep
        const T dg = VT::adds(hdiag, VT::score(ch[k], c, wm[k], wd[k]));
        const T m = VT::mx(G[k], f);
        const T hn = VT::mx(dg, m);
        if constexpr (MASK) {
          // eight compares into fixed SGPR pairs, four 16-byte scalar stores: [half][x, y, z, w].  The wait in front: the stores of the
          // row before have read their SGPRs (measured free: the scalar-memory pipe is idle otherwise); the s_nop: VALU-written
          // SGPRs read by a scalar-memory instruction (inline assembly gets no hazard padding from the compiler).
          const int wa = VT::bits(LANE_IS_ALT ? f : G[k]), wb = VT::bits(LANE_IS_ALT ? G[k] : f);      // right < down
          asm volatile(
              "s_waitcnt lgkmcnt(0)\n\t"
              "v_cmp_lt_i16_sdwa s[40:41], %0, %1 src0_sel:WORD_0 src1_sel:WORD_0\n\t"
              "v_cmp_lt_i16_sdwa s[42:43], %2, %3 src0_sel:WORD_0 src1_sel:WORD_0\n\t"
              "v_cmp_lt_i16_sdwa s[44:45], %4, %5 src0_sel:WORD_0 src1_sel:WORD_0\n\t"
              "v_cmp_lt_i16_sdwa s[46:47], %6, %7 src0_sel:WORD_0 src1_sel:WORD_0\n\t"
              "v_cmp_lt_i16_sdwa s[48:49], %0, %1 src0_sel:WORD_1 src1_sel:WORD_1\n\t"
              "v_cmp_lt_i16_sdwa s[50:51], %2, %3 src0_sel:WORD_1 src1_sel:WORD_1\n\t"
              "v_cmp_lt_i16_sdwa s[52:53], %4, %5 src0_sel:WORD_1 src1_sel:WORD_1\n\t"
              "v_cmp_lt_i16_sdwa s[54:55], %6, %7 src0_sel:WORD_1 src1_sel:WORD_1\n\t"
              "s_nop 4\n\t"
              "s_store_dwordx4 s[40:43], %8, %9\n\t"
              "s_store_dwordx4 s[44:47], %8, %10\n\t"
              "s_store_dwordx4 s[48:51], %8, %11\n\t"
              "s_store_dwordx4 s[52:55], %8, %12\n\t"
              :
              : "v"(VT::bits(fe)), "v"(VT::bits(fo)), "v"(VT::bits(ge)), "v"(VT::bits(go)), "v"(VT::bits(dg)), "v"(VT::bits(m)), "v"(wa), "v"(wb),
                "s"(rec_step), "n"(k * 64), "n"(k * 64 + 16), "n"(k * 64 + 32), "n"(k * 64 + 48)
              : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "memory");
        } else if (BT) {
          pf = VT::push_lt(pf, fe, fo);
          pg = VT::push_lt(pg, ge, go);
          pn = VT::push_lt(pn, dg, m);
          pd = LANE_IS_ALT ? VT::push_lt(pd, f, G[k]) : VT::push_lt(pd, G[k], f);   // right < down
        }
        hdiag = hold; hup = hn; Hp[k] = hn;
        fo = VT::adds(hn, opn_s); Ho[k] = fo;
      }
      h_last = hup; f_last = f;
      if (BT && !MASK) {
        if (a.bt) a.bt[(uint64_t)(blockIdx.x + work_base - bt_first) * a.bt_item_stride + ((uint64_t)g * (sweep_cap + LPP) + t) * LPP + l] = make_uint4(pf, pg, pn, pd);
        else dbg_acc ^= pf ^ (pg * 3u) ^ (pn * 5u) ^ (pd * 7u);       // measurement aid (ACCG_SW_BT_DEBUG=2, tools/exp_sw_bt.py): planes formed, nothing stored
      }
      if (l == LPP - 1) my_log[i] = VT::bits(hup);
    }
  }
  if constexpr (MASK) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");    // the scalar cache's dirty lines, before the backtrace kernel reads them
  __syncthreads();
  if (BT && !a.bt && dbg_acc == 0x12345u) a.score[0] = (int)dbg_acc;

  // ---- end cell (calculateCigarOneBatch, FalconSW_AVX.cpp:2314-2339) ---------------------------------
  // "edge" candidates: H[lane-seq end][sweep index s], s = 1..ns (from the log);
  // "final" candidates: H[lane position p][sweep end], p = 1..nl (in registers).
  // LANE_IS_ALT: edge = last column (i = s, j = altLen), final = bottom row (i = refLen, j = p); else the converse.
#pragma unroll
  for (int h = 0; h < (P16 ? 2 : 1); h++) {
    if (!have[h]) continue;   // uniform per group of 16 lanes; shuffles below stay inside the group
    const int refLen = LANE_IS_ALT ? ns : nl[h], altLen = LANE_IS_ALT ? nl[h] : ns;
    const int pad = LPP * K - nl[h];
    // last column: ties -> larger i  (:2320-2326)
    unsigned long long ck = 0;
    if (LANE_IS_ALT) {
      for (int s = l + 1; s <= ns; s += LPP) {
        int v = VT::get(VT::from_bits(my_log[s]), h);
        unsigned long long key = ((unsigned long long)(unsigned)(v + 0x40000000) << 16) | (unsigned)s;
        ck = key > ck ? key : ck;
      }
    } else {
#pragma unroll
      for (int k = 0; k < K; k++) {
        const int p = l * K + k - pad + 1;
        if (p >= 1) {
          int v = VT::get(Hp[k], h);
          unsigned long long key = ((unsigned long long)(unsigned)(v + 0x40000000) << 16) | (unsigned)p;
          ck = key > ck ? key : ck;
        }
      }
    }
    ck = group_max_u64<LPP>(ck);
    const int col_best = (int)(unsigned)(ck >> 16) - 0x40000000, col_i = (int)(ck & 0xFFFF);
    int p1, p2, best;
    if (strat[h] == 1) {            // INDEL: the corner
      p1 = refLen; p2 = altLen;
      best = __shfl(VT::get(Hp[K - 1], h), g * LPP + LPP - 1);
    } else if (strat[h] == 2) {     // LEADING_INDEL: last column only
      p1 = col_i; p2 = altLen; best = col_best;
    } else {
      unsigned long long bk = (l == 0) ? cand_key(col_best, iabs(col_i - altLen), 0) : 0ull;
      if (LANE_IS_ALT) {
#pragma unroll
        for (int k = 0; k < K; k++) {
          const int j = l * K + k - pad + 1;
          if (j >= 1) { unsigned long long key = cand_key(VT::get(Hp[k], h), iabs(refLen - j), j); bk = key > bk ? key : bk; }
        }
      } else {
        for (int j = l + 1; j <= ns; j += LPP) {
          unsigned long long key = cand_key(VT::get(VT::from_bits(my_log[j]), h), iabs(refLen - j), j);
          bk = key > bk ? key : bk;
        }
      }
      bk = group_max_u64<LPP>(bk);
      best = (int)(unsigned)(bk >> 24) - 0x40000000;
      const int order = 4095 - (int)(bk & 0xFFF);
      if (order == 0) { p1 = col_i; p2 = altLen; } else { p1 = refLen; p2 = order; }
    }
    if (l == 0) { a.score[pr[h]] = best; a.p1[pr[h]] = p1; a.p2[pr[h]] = p2; }
  }
}

// ---- backtrace: one thread per pair ------------------------------------------------------------------
// Restates calculateCigarOneBatch (FalconSW_AVX.cpp:2341-2417) on top of the bit planes: the walk, the
// strategy-specific tail, alignment_offset and the final reversal.
template <bool LANE_IS_ALT>
__device__ __forceinline__ void sw_trace_pairs(const SwArgs& a, uint32_t work_base, uint32_t n_work, uint32_t bt_first, int K, int LPP, int p16, int sweep_cap, int masks,
                                               const uint32_t tid) {
  const uint32_t wi = tid >> 3, slot = tid & 7;
  const uint32_t pair = wi < n_work ? a.work[work_base + wi].pair[slot] : SW_NO_PAIR;
  const int g = slot >> 1, half = slot & 1;
  const bool live = !(pair == SW_NO_PAIR || (!p16 && half) || g >= 64 / LPP);
  int n = 0;
  int32_t* el = nullptr;
  if (live) {
  const int refLen = a.ref_len[pair], altLen = a.alt_len[pair], strat = a.strategy[pair];
  const int nl = LANE_IS_ALT ? altLen : refLen;
  const int pad = LPP * K - nl;
  const uint4* bt = a.bt + (uint64_t)(work_base + wi - bt_first) * a.bt_item_stride + (masks ? 0ull : (uint64_t)g * (sweep_cap + LPP) * LPP);
  const int lane_bit0 = g * LPP;              // (mask layout: this pair's lanes in the wavefront's masks)
  // planes: 0 lane-direction gap opened, 1 sweep-direction gap opened; "right" (insertion) runs along the alternate
  const int P_HOPEN = LANE_IS_ALT ? 0 : 1, P_VOPEN = LANE_IS_ALT ? 1 : 0;
  int p1 = a.p1[pair], p2 = a.p2[pair];
  el = a.cig_el + (size_t)pair * a.max_el * 2;
  const int cap = a.max_el;
  auto push = [&](int len, int st) {           // addCigarElement (sw_host.cpp:17-26): non-positive lengths are dropped
    if (len <= 0) return;
    if (n < cap) { el[2 * n] = len; el[2 * n + 1] = st; }
    n++;
  };
  int seg = 0;
  if (strat != 1 && strat != 2 && p1 == refLen && p2 != altLen) seg = altLen - p2;   // set by the bottom-row scan (:2335)
  // (when the bottom-row scan picked j == altLen the segment is 0 as well)
  if (seg > 0 && strat == 0) { push(seg, 4); seg = 0; }                               // soft clip, :2342-2345
  int state = 0;
  // The walk is a chain of dependent loads (the next cell is known only from this cell's bits).  Nearly every step is
  // diagonal, so the record entries of the next eight diagonal cells -- and, inside a gap, of the next eight cells along the
  // gap -- are fetched together and then examined in order: one memory latency per eight cells instead of one per cell.
  constexpr int LOOK = 8;
  // the four decision bits of cell (i, j) as planes {x, y, z, w} with the cell's bit at `shift`
  // (gap_bits: the caller looks at x / y only -- inside a gap run -- else at z / w only: one 16-byte load per cell in the mask layout too)
  auto entry = [&](int i, int j, int& shift, bool gap_bits) -> uint4 {
    const int sidx = LANE_IS_ALT ? i : j, pos = LANE_IS_ALT ? j : i;
    const int flat = pos - 1 + pad, l = flat / K, k = flat - l * K;
    if (masks) {      // [half][x, y, z, w] lane masks of (step, row): this cell's lane bit of each, as planes with the bit at 0
      const uint4 m = bt[((uint64_t)(sidx + l) * K + k) * 4 + half * 2 + (gap_bits ? 0 : 1)];
      const int bit = lane_bit0 + l;
      const unsigned b0 = (unsigned)((((unsigned long long)m.y << 32 | m.x) >> bit) & 1ull), b1 = (unsigned)((((unsigned long long)m.w << 32 | m.z) >> bit) & 1ull);
      shift = 0;
      return gap_bits ? make_uint4(b0, b1, 0u, 0u) : make_uint4(0u, 0u, b0, b1);
    }
    shift = p16 ? Val<true>::plane_bit(K, k, half) : Val<false>::plane_bit(K, k, half);
    return bt[(uint64_t)(sidx + l) * LPP + l];
  };
  do {
    // ---- run of diagonal cells starting at (p1, p2)
    uint4 e[LOOK]; int sh[LOOK];
    const int nv = min(LOOK, min(p1, p2));
#pragma unroll
    for (int d = 0; d < LOOK; d++) {
      sh[d] = 0; e[d] = make_uint4(0, 0, 0, 0);
      if (d < nv) e[d] = entry(p1 - d, p2 - d, sh[d], false);
    }
    int d = 0;
#pragma unroll
    for (int q = 0; q < LOOK; q++)
      if (d == q && q < nv && !((e[q].z >> sh[q]) & 1u)) d = q + 1;        // cell q is diagonal: pass it
    if (d > 0) {
      if (state == 0) seg += d; else { push(seg, state); seg = d; state = 0; }
      p1 -= d; p2 -= d;
    }
    if (d == nv) continue;                    // window used up (or the border reached) without meeting a gap
    // ---- (p1, p2) is not diagonal; its entry is e[d]
    uint4 ce = e[0]; int cs = sh[0];
#pragma unroll
    for (int q = 1; q < LOOK; q++) if (d == q) { ce = e[q]; cs = sh[q]; }
    const bool down = (ce.w >> cs) & 1u;      // down: deletion of kd rows; else right: insertion of ki columns
    const int ns_ = down ? 2 : 1;
    int step = 1;
    // run length = 1 + number of leading cells (from this one, walking back) that did not open the gap; the cell at
    // index 1 always opens (its extension source is -inf), the bound is a guard
    int r = down ? p1 : p2;
    bool open_found = false;
    while (!open_found && r > 1) {
      uint4 ge[LOOK]; int gs[LOOK];
      const int gn = min(LOOK, r - 1);
#pragma unroll
      for (int q = 0; q < LOOK; q++) {
        gs[q] = 0; ge[q] = make_uint4(0, 0, 0, 0);
        if (q < gn) ge[q] = down ? entry(r - q, p2, gs[q], true) : entry(p1, r - q, gs[q], true);
      }
      int t = 0;
#pragma unroll
      for (int q = 0; q < LOOK; q++) {
        const unsigned w = (down ? (P_VOPEN == 0 ? ge[q].x : ge[q].y) : (P_HOPEN == 0 ? ge[q].x : ge[q].y));
        if (t == q && q < gn) { if ((w >> gs[q]) & 1u) open_found = true; else t = q + 1; }
      }
      step += t; r -= t;
      if (t < gn) open_found = true;
    }
    if (ns_ == 1) p2 -= step; else p1 -= step;
    if (ns_ == state) seg += step;
    else { push(seg, state); seg = step; state = ns_; }
  } while (p1 > 0 && p2 > 0);
  int off;
  if (strat == 0) { push(seg, state); if (p2 > 0) push(p2, 4); off = p1; }            // :2379-2385
  else if (strat == 3) { push(seg + p2, state); off = p1 - p2; }                      // :2386-2389
  else { push(seg, state); if (p1 > 0) push(p1, 2); else if (p2 > 0) push(p2, 1); off = 0; }   // :2390-2400
  a.cig_off[pair] = off;
  if (n <= 0) { a.cig_n[pair] = -1; n = 0; }
  else if (n > cap) { a.cig_n[pair] = -n; n = 0; }
  else a.cig_n[pair] = n;
  }
  // Space in the packed result: one atomic per wavefront, lanes take consecutive ranges; the copy reverses the list
  // (the walk produced it from the end cell backwards, calculateCigarOneBatch reverses it at :2408-2417).
  const int lane = threadIdx.x & 63;
  int incl = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
  const int total = __shfl(incl, 63);
  if (total == 0) return;
  unsigned long long base = 0;
  if (lane == 63) base = atomicAdd(a.cig_total, (unsigned long long)total);
  base = __shfl(base, 63);
  if (!live) return;
  const unsigned long long start = base + (unsigned long long)(incl - n);
  a.cig_start[pair] = start;
  int2* dst = reinterpret_cast<int2*>(a.cig_packed) + start;
  const int2* src = reinterpret_cast<const int2*>(el);
  for (int x = 0; x < n; x++) dst[x] = src[n - 1 - x];
}
// The walk is latency bound (dependent loads) and shares the chip with the NEXT slice's fill, which wants its four wavefronts per SIMD:
// a grid of every pair at once puts two to three backtrace wavefronts on every SIMD and pushes a fill wavefront off each of them, so the
// grid is capped (ACCG_SW_TRACE_WAVES wavefronts, sw_host.cpp) and every wavefront walks its share of the slice's pairs, 64 at a time.
template <bool LANE_IS_ALT>
__global__ __launch_bounds__(64) void sw_trace_kernel(SwArgs a, uint32_t work_base, uint32_t n_work, uint32_t bt_first, int K, int LPP, int p16, int sweep_cap, int masks) {
  const uint32_t n_threads = n_work * 8u;
  for (uint32_t base = blockIdx.x * 64u; base < n_threads; base += gridDim.x * 64u)
    sw_trace_pairs<LANE_IS_ALT>(a, work_base, n_work, bt_first, K, LPP, p16, sweep_cap, masks, base + threadIdx.x);
}

template <int LPP, bool P16, bool LIA, bool BT>
hipError_t launch(int K, const SwArgs& a, uint32_t wb, uint32_t n, uint32_t bt_first, int cap, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const size_t lds = sw_lds_bytes(cap);
#define ACCG_CASE(KK) case KK: hipLaunchKernelGGL((sw_kernel<KK, LPP, P16, LIA, BT>), dim3(n), dim3(64), lds, st, a, wb, bt_first, cap); break;
  if constexpr (LPP == 16 && P16 && BT) {
    if (a.bt_masks && a.bt) {
#define ACCG_MCASE(KK) case KK: hipLaunchKernelGGL((sw_kernel<KK, 16, true, LIA, true, true>), dim3(n), dim3(64), lds, st, a, wb, bt_first, cap); break;
      switch (K) {
        ACCG_MCASE(1) ACCG_MCASE(2) ACCG_MCASE(3) ACCG_MCASE(4) ACCG_MCASE(5) ACCG_MCASE(6) ACCG_MCASE(7) ACCG_MCASE(8)
        ACCG_MCASE(9) ACCG_MCASE(10) ACCG_MCASE(11) ACCG_MCASE(12) ACCG_MCASE(13) ACCG_MCASE(14) ACCG_MCASE(15) ACCG_MCASE(16)
        default: return hipErrorInvalidValue;
      }
#undef ACCG_MCASE
      return hipGetLastError();
    }
  }
  if (LPP == 16) {
    switch (K) {
      ACCG_CASE(1) ACCG_CASE(2) ACCG_CASE(3) ACCG_CASE(4) ACCG_CASE(5) ACCG_CASE(6) ACCG_CASE(7) ACCG_CASE(8)
      ACCG_CASE(9) ACCG_CASE(10) ACCG_CASE(11) ACCG_CASE(12) ACCG_CASE(13) ACCG_CASE(14) ACCG_CASE(15) ACCG_CASE(16)
      default: return hipErrorInvalidValue;
    }
  } else if (LPP == 32) {
    switch (K) { ACCG_CASE(10) ACCG_CASE(12) ACCG_CASE(14) ACCG_CASE(16) default: return hipErrorInvalidValue; }
  } else {
    switch (K) { ACCG_CASE(10) ACCG_CASE(12) ACCG_CASE(14) ACCG_CASE(16) ACCG_CASE(20) ACCG_CASE(24) default: return hipErrorInvalidValue; }
  }
#undef ACCG_CASE
  return hipGetLastError();
}

template <int LPP>
hipError_t launch_sel(int K, bool pack16, bool lane_is_alt, bool with_bt, const SwArgs& a, uint32_t wb, uint32_t n, uint32_t bt_first,
                      int cap, hipStream_t s) {
  const int sel = (pack16 ? 4 : 0) | (lane_is_alt ? 2 : 0) | (with_bt ? 1 : 0);
  switch (sel) {
    case 0: return launch<LPP, false, false, false>(K, a, wb, n, bt_first, cap, s);
    case 1: return launch<LPP, false, false, true>(K, a, wb, n, bt_first, cap, s);
    case 2: return launch<LPP, false, true, false>(K, a, wb, n, bt_first, cap, s);
    case 3: return launch<LPP, false, true, true>(K, a, wb, n, bt_first, cap, s);
    case 4: return launch<LPP, true, false, false>(K, a, wb, n, bt_first, cap, s);
    case 5: return launch<LPP, true, false, true>(K, a, wb, n, bt_first, cap, s);
    case 6: return launch<LPP, true, true, false>(K, a, wb, n, bt_first, cap, s);
    default: return launch<LPP, true, true, true>(K, a, wb, n, bt_first, cap, s);
  }
}

}  // namespace

size_t sw_lds_bytes(int sweep_cap) { return (size_t)8 * sw_group_stride(sweep_cap) * 4; }

hipError_t sw_launch(int K, int lpp, bool pack16, bool lane_is_alt, bool with_bt, const SwArgs& a, uint32_t wb, uint32_t n,
                     uint32_t bt_first, int cap, hipStream_t s) {
  if (lpp == 16) return launch_sel<16>(K, pack16, lane_is_alt, with_bt, a, wb, n, bt_first, cap, s);
  if (lpp == 32) return launch_sel<32>(K, pack16, lane_is_alt, with_bt, a, wb, n, bt_first, cap, s);
  return launch_sel<64>(K, pack16, lane_is_alt, with_bt, a, wb, n, bt_first, cap, s);
}

hipError_t sw_trace_launch(int K, int lpp, bool pack16, bool lane_is_alt, const SwArgs& a, uint32_t wb, uint32_t n, uint32_t bt_first,
                           int cap, hipStream_t s) {
  if (n == 0) return hipSuccess;
  // (default 1024 = one per SIMD of the 256 CUs: configs[2] with CIGARs 19.0 -> 18.1 ms; 512: 20.2, the walk becomes the longer half;
  // 768 ... 4096: 18.0-18.5 and 19.0; ACCG_SW_TRACE_WAVES=-1: a wavefront per 64 pairs as before)
  static const int wave_knob = [] { const char* e = getenv("ACCG_SW_TRACE_WAVES"); return e ? atoi(e) : 0; }();
  const uint32_t wave_cap = wave_knob < 0 ? 0u : wave_knob > 0 ? (uint32_t)wave_knob : 1024u;
  const uint32_t threads = n * 8, block = 64, all = (threads + block - 1) / block, grid = wave_cap ? std::min(all, wave_cap) : all;
  const int masks = a.bt_masks && pack16 && lpp == 16 ? 1 : 0;
  if (lane_is_alt) hipLaunchKernelGGL((sw_trace_kernel<true>), dim3(grid), dim3(block), 0, s, a, wb, n, bt_first, K, lpp, (int)pack16, cap, masks);
  else hipLaunchKernelGGL((sw_trace_kernel<false>), dim3(grid), dim3(block), 0, s, a, wb, n, bt_first, K, lpp, (int)pack16, cap, masks);
  return hipGetLastError();
}

// K actually instantiated for a lane sequence of nl positions spread over lpp lanes (0 = not supported)
int sw_pick_k(int nl, int lpp) {
  const int need = (nl + 1 + lpp - 1) / lpp;
  if (lpp == 16) return need <= 16 ? need : 0;
  static const int k32[] = {10, 12, 14, 16}, k64[] = {10, 12, 14, 16, 20, 24};
  if (lpp == 32) { for (int k : k32) if (k >= need) return k; return 0; }
  for (int k : k64) if (k >= need) return k;
  return 0;
}

}  // namespace accg

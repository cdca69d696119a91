// BWA-MEM seed extension (banded, adaptive trimming) on gfx950.  Behaviour restated from bwa-sw/sdaccel/smithwaterman.cpp:
// sw_extend :75-273 (one side, up to two band tries) and seed_proc :586-670 (left side, then right side seeded with the left
// score).  The FPGA walks one row cell by cell; here a row is one step of a 16-lane group:
//
//   * a seed owns 16 lanes, a wavefront four seeds; lane l keeps K consecutive entries of the eh[] row buffer (H of the row
//     above shifted by one column, E) in registers, right-aligned so that entry `qlen` is the last entry of lane 15;
//   * the insertion chain f[j+1] = max(f[j]-1, M[j]-7, 0) is a max-plus scan: each lane runs it locally from 0, the carry-in
//     comes from an exclusive prefix max over the group (4 DPP row_shr steps), and the true f is max(local, carry - k);
//   * the row's arg-max (last column on ties), the first/last non-zero entries that drive the adaptive band, and the value at
//     the query end are group reductions (DPP butterflies), so the trimming state (beg, end) stays replicated per lane;
//   * the substitution score is a 4-bit field of a per-column word selected by the row's target base.
//
// All arithmetic is int32; the FPGA's ap_int widths never wrap inside the limits the host enforces.
#include "bwasw_dev.h"

namespace accg {
namespace {

typedef short s2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ int dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, 0xF, false); }
// lanes without a source read 0 (bound_ctrl): lets the compiler fold the move into the consuming v_max_i32 (..._dpp)
template <int CTRL>
__device__ __forceinline__ int dpp0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imax3(int a, int b, int c) { return imax(imax(a, b), c); }

// all-reduce max over the 16 lanes of a DPP row (every lane has a source in these permutations)
__device__ __forceinline__ int group_allmax(int v) {
  v = imax(v, dpp0<0xB1>(v));        // quad_perm [1,0,3,2]
  v = imax(v, dpp0<0x4E>(v));        // quad_perm [2,3,0,1]
  v = imax(v, dpp0<0x141>(v));       // row_half_mirror
  v = imax(v, dpp0<0x140>(v));       // row_mirror
  return v;
}
__device__ __forceinline__ int pkmax(int a, int b) {
  return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b)));
}
__device__ __forceinline__ int group_allmax_pk(int v) {
  v = pkmax(v, dpp0<0xB1>(v));
  v = pkmax(v, dpp0<0x4E>(v));
  v = pkmax(v, dpp0<0x141>(v));
  v = pkmax(v, dpp0<0x140>(v));
  return v;
}
// exclusive prefix max over the row of non-negative values; lane 0 gets 0
__device__ __forceinline__ int group_exscan_max0(int v) {
  v = imax(v, dpp0<0x111>(v));
  v = imax(v, dpp0<0x112>(v));
  v = imax(v, dpp0<0x114>(v));
  v = imax(v, dpp0<0x118>(v));
  return dpp0<0x111>(v);
}

// SIDE 0 = left extension (writes its partial result into the seed's output record), SIDE 1 = right extension (reads it back
// and finishes the record).  Two passes let each side run with the K and the row count of its own query/target.
template <int K, int SIDE>
__global__ __launch_bounds__(64) void bwasw_kernel(BwaswArgs a) {
  const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
  const uint32_t sid = a.work[blockIdx.x].seed[g];
  const bool have = sid != BWASW_EMPTY;
  const BwaswSeed* sp = a.seeds + (have ? sid : 0);
  const uint32_t q_off = have ? sp->q_off[SIDE] : 0, t_off = have ? sp->t_off[SIDE] : 0;
  const int qlen = have ? sp->qlen[SIDE] : 0, tlen = have ? sp->tlen[SIDE] : 0;
  const int seed_len = have ? sp->seed_len : 0, seed_qbeg = have ? sp->seed_qbeg : 0;
  int16_t* rec = a.out + (size_t)(have ? sid : 0) * 8;

  int regScore = seed_len;
  int qBeg = 0, qEnd = qlen, rBeg = 0, rEnd = 0, trueScore = seed_len, score = 0, aw0 = 100, aw1s = 100;
  if (SIDE == 1 && have) {                                   // the left pass left {qBeg, -, rBeg, -, regScore, trueScore, aw[0], -}
    const uint4 r = *(const uint4*)rec;
    qBeg = (int16_t)(r.x & 0xFFFF); rBeg = (int16_t)(r.y & 0xFFFF);
    regScore = (int16_t)(r.z & 0xFFFF); trueScore = (int16_t)(r.z >> 16); aw0 = (int16_t)(r.w & 0xFFFF);
  }

  {
    constexpr int side = SIDE;
    const int sc0 = regScore, h0 = side == 0 ? seed_len : sc0;
    const int j0 = l * K;                                    // column of this lane's entry 0; entries past qlen are padding
    const int lq = qlen / K, kq = qlen - lq * K;             // where entry `qlen` lives
    const int end_src = ((lane & 48) | lq) << 2;             // ds_bpermute address of that lane
    const uint8_t* qp = a.blob + q_off;
    const uint8_t* tp = a.blob + t_off;
    uint32_t W[K];                                           // signed nibble t of W = score(t, q): 1, -4 (0xC), -1 (0xF)
    int eh_h[K], eh_e[K];
    // The device code never clears eh[] between band tries, so the second try (w = 200) can read, right of column 200, what
    // the first try left behind -- including entries this kernel zeroes when the band limit steps over them.  Only queries
    // longer than 201 can see that; for those (K >= 13) the stepped-over values are kept aside and put back.
    constexpr bool STASH = K >= 13;
    int st_h[STASH ? K : 1], st_e[STASH ? K : 1];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int j = j0 + k;
      const int c = j < qlen ? qp[j] : 4;
      W[k] = c > 3 ? 0xFFFFFu : ((0xFCCCCu & ~(0xFu << (4 * c))) | (1u << (4 * c)));
      eh_h[k] = 0; eh_e[k] = 0;
      if constexpr (STASH) { st_h[k] = 0; st_e[k] = 0; }
    }
    int mx = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    bool stop = !have;
    int aw_used = 100;

    for (int bt = 0; bt < 2; bt++) {                         // band tries (:136-272)
      const bool need = !stop;
      if (__ballot(need) == 0) break;
      const int prev = regScore;
      const int aw_tmp = bt ? 200 : 100;                     // (w_in << k) as uint8_t
      const int aw1 = aw_tmp < qlen ? aw_tmp : qlen;         // max_ins = max_del = qlen (:600-603)
      int beg = 0, end = qlen;
      const int eme0 = imax(h0 - 7, 0);
      int h1_init = h0 - 6;
      if (need) {                                            // row 0 reads h0's decay, not the buffer (:175-191)
        const int end0 = aw1 + 1 < qlen ? aw1 + 1 : qlen;
#pragma unroll
        for (int k = 0; k < K; k++) {
          const int j = j0 + k;
          if (j < end0) { eh_h[k] = j == 0 ? h0 : imax(eme0 - (j - 1), 0); eh_e[k] = 0; }
          else if constexpr (STASH) {
            if (bt == 1 && (st_h[k] | st_e[k]) != 0) { eh_h[k] = st_h[k]; eh_e[k] = st_e[k]; }
          }
        }
      }
      bool active = need && tlen > 0;
      int i = 0;
      uint32_t tw = 0, tw_next = active ? *(const uint32_t*)tp : 0;
      // The row body runs for every group of the wavefront as long as one of them is active; a finished group computes
      // along (lanes under EXEC masking cost the same issue slots) with all its writes switched off.
      while (__ballot(active) != 0) {
        if ((i & 3) == 0) { tw = tw_next; tw_next = (active && i + 4 < tlen) ? *(const uint32_t*)(tp + i + 4) : 0; }
        const int sh = ((tw >> ((i & 3) * 8)) & 0xFF) * 4;
        if (__ballot(active && beg < i - aw1) != 0) {        // the band limit may step over live entries (:151): clear them,
          if (active && beg < i - aw1) {                     // the row code relies on "everything left of beg is zero"
            const int kb0 = beg - j0, kb1 = i - aw1 - j0;
#pragma unroll
            for (int k = 0; k < K; k++)
              if (k >= kb0 && k < kb1) {
                if constexpr (STASH) { st_h[k] = eh_h[k]; st_e[k] = eh_e[k]; }
                eh_h[k] = 0; eh_e[k] = 0;
              }
            beg = i - aw1;
          }
        }
        if (active) {
          if (end > i + aw1 + 1) end = i + aw1 + 1;
          if (end > qlen) end = qlen;
        }
        int h1row = 0;                                       // H left of the band's first column; 0 once beg has left column 0
        if (active && beg == 0) { h1_init -= 1; h1row = imax(h1_init, 0); }
        const int ke = active ? end - j0 : -1;               // this lane's entries [.., ke) are inside the band; -1: write nothing

        // pass 1: diagonal term, E, local insertion chain.  No band mask is needed here: entries left of beg are zero, so
        // they produce H = E = F = 0, and whatever is computed right of end is never stored.
        int hp[K], en[K];
        int f = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
          const int M0 = eh_h[k], e = eh_e[k];
          const int s = __builtin_amdgcn_sbfe((int)W[k], sh, 4);
          const int Mn = M0 + (s < M0 ? s : M0);             // M ? M + s : 0 up to a negative value where 0 is meant: both lose
          const int tm = Mn - 7;                             // against E, F >= 0 and give tm < 0
          hp[k] = imax3(Mn, e, f);
          en[k] = imax3(e - 1, tm, 0);
          f = imax3(f - 1, tm, 0);
        }
        // f entering this lane's first entry: max over the lanes to the left of (their f_out - K per lane in between)
        const int carry = imax(group_exscan_max0(f + K * (l + 1)) - K * l, 0);

        // pass 2: finish H, shift it into the buffer, row statistics (positions as k, converted to columns afterwards)
        int hk[K];
#pragma unroll
        for (int k = 0; k < K; k++) hk[k] = imax(hp[k], carry - k);
        // H(i, j-1) for this lane's entry 0; column 0 takes the row's left border (0 when the band has moved on: then entry 0
        // is left of beg and stays zero)
        const int hin = dpp<0x111>(h1row, hk[K - 1]);
        int key = -1, firstk = -1, lastk = -1;
        bool nzl[K];
        bool le = -1 < ke;                                   // k - 1 < ke  <=>  k <= ke
#pragma unroll
        for (int k = 0; k < K; k++) {
          const bool lt = k < ke;                            // entries [.., end)
          const int nh = k == 0 ? hin : hk[k - 1];
          const int ne = lt ? en[k] : 0;                     // entry `end` gets E = 0
          if (le) { eh_h[k] = nh; eh_e[k] = ne; }
          key = imax(key, ((lt ? hk[k] : -1) << 4) | k);     // last column wins ties (:216)
          const bool nz = le && ((nh | ne) != 0);
          if (nz) lastk = k;
          nzl[k] = nz && lt;
          le = lt;
        }
#pragma unroll
        for (int k = K - 1; k >= 0; k--)
          if (nzl[k]) firstk = k;
        const int first_l = firstk < 0 ? 511 : j0 + firstk;
        int last = lastk < 0 ? -1 : j0 + lastk;
        key = key < 0 ? -1 : (((key >> 4) << 8) | (j0 + (key & 15)));
        key = group_allmax(key);
        const int pk = group_allmax_pk(((last + 1) << 16) | (512 - first_l));
        last = (pk >> 16) - 1;
        const int first = 512 - (pk & 0xFFFF);
        const int m = key < 0 ? 0 : key >> 8, mj = key < 0 ? -1 : (key & 0xFF);
        int hq = eh_h[0];                                    // entry `qlen` of its lane
#pragma unroll
        for (int k = 1; k < K; k++) hq = k == kq ? eh_h[k] : hq;
        const int h1 = __builtin_amdgcn_ds_bpermute(end_src, hq);
        if (active) {
          if (end == qlen && gscore <= h1) { max_ie = i; gscore = h1; }   // the row reached the query end (:238-243)
          if (m == 0) active = false;
          else {
            if (m > mx) {
              mx = m; max_i = i; max_j = mj;
              const int d = mj > i ? mj - i : i - mj;
              if (max_off < d) max_off = d;
            }
            beg = first == 511 ? end : first;                // beg + leading zero entries (:222-228, :263)
            end = last + 2 < qlen ? last + 2 : qlen;         // end - trailing zero entries + 2 (:229-236, :264)
            i++;
            if (i >= tlen) active = false;
          }
        }
      }
      if (need) {
        regScore = mx;
        stop = (mx == prev) || (max_off < (aw_tmp >> 1) + (aw_tmp >> 2));
        aw_used = aw_tmp;
      }
    }
    if (side == 0) aw0 = aw_used; else aw1s = aw_used;
    const int qle = max_j + 1, tle = max_i + 1, gtle = max_ie + 1;
    score = regScore;
    if (gscore <= 0 || gscore <= regScore - 5) {
      if (side == 0) { qBeg = seed_qbeg - qle; rBeg = -tle; trueScore = regScore; }
      else { qEnd = qle; rEnd = tle; trueScore += regScore - sc0; }
    } else {
      if (side == 0) { qBeg = 0; rBeg = -gtle; trueScore = gscore; }
      else { qEnd = qlen; rEnd = gtle; trueScore += gscore - sc0; }
    }
  }
  if (have && l == 0) {
    const int w = SIDE == 0 ? aw0 : (aw0 > aw1s ? aw0 : aw1s);
    uint4 o;
    o.x = (uint32_t)(qBeg & 0xFFFF) | ((uint32_t)(qEnd & 0xFFFF) << 16);
    o.y = (uint32_t)(rBeg & 0xFFFF) | ((uint32_t)(rEnd & 0xFFFF) << 16);
    o.z = (uint32_t)(score & 0xFFFF) | ((uint32_t)(trueScore & 0xFFFF) << 16);
    o.w = (uint32_t)(w & 0xFFFF);
    *(uint4*)rec = o;
  }
}

}  // namespace

hipError_t bwasw_launch(int K, int side, const BwaswArgs& a, uint32_t n_work, hipStream_t s) {
  if (n_work == 0) return hipSuccess;
#define ACCG_BWASW_CASE(k) case k: \
    if (side == 0) hipLaunchKernelGGL((bwasw_kernel<k, 0>), dim3(n_work), dim3(64), 0, s, a); \
    else hipLaunchKernelGGL((bwasw_kernel<k, 1>), dim3(n_work), dim3(64), 0, s, a); \
    break;
  switch (K) {
    ACCG_BWASW_CASE(1) ACCG_BWASW_CASE(2) ACCG_BWASW_CASE(3) ACCG_BWASW_CASE(4) ACCG_BWASW_CASE(5) ACCG_BWASW_CASE(6)
    ACCG_BWASW_CASE(7) ACCG_BWASW_CASE(8) ACCG_BWASW_CASE(9) ACCG_BWASW_CASE(10) ACCG_BWASW_CASE(11) ACCG_BWASW_CASE(12)
    ACCG_BWASW_CASE(13) ACCG_BWASW_CASE(14) ACCG_BWASW_CASE(15) ACCG_BWASW_CASE(16)
    default: return hipErrorInvalidValue;
  }
#undef ACCG_BWASW_CASE
  return hipGetLastError();
}

}  // namespace accg

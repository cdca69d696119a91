// BWA-MEM SMEM seeding (Falcon's three-pass variant) for gfx950: one thread per read.
//
// What it computes: mem_collect_intv_new of the reference (smem/host/baseline.cpp:387-422) = bwt_smem1a_new
// (:180-304) over the read, re-seeding inside long low-occurrence SMEMs, and the LAST-like bwt_seed_strategy1 pass
// (:306-327), on top of bwt_extend / bwt_occ4 (:17-100).  PARITY: bit-exact with the CPU restatement kept with the tests (smem_oracle.c), which restates
// those functions; the reference file itself cannot be built here (libbwa is not in the tree), see DESIGN.md.
//
// Why one thread per read: every bwt_extend depends on the previous one and costs two random 64-byte block reads, so
// the path is bound by memory latency, not by arithmetic; the only parallelism that hides it is many independent
// reads in flight (2048 per CU).  The 64 MB index of configs[4] sits in the 256 MB Infinity Cache.  The per-read
// interval lists (curr / back of bwt_smem1a_new, up to 255 entries each) live in a thread-interleaved HBM scratch so
// that the lanes of a wave touch neighbouring 32-byte elements.  Occ is computed with popcounts over the 2-bit
// words instead of the reference's byte table; the counts are identical by definition.
#include <stdlib.h>
#include "smem_dev.h"

namespace accg {
namespace {

constexpr int MIN_SEED_LEN = 19;   // smem/common/common.h:37
#ifndef SMEM_MIN_WAVES
#define SMEM_MIN_WAVES 1
#endif

struct Ctx {
  const uint32_t* bwt;
  uint64_t primary, L2[5];
  bool compact;
};

// number of symbols equal to c among the first `upto + 1` symbols (0-based, MSB first) of the eight words of a block
__device__ __forceinline__ void count_block(const uint4 w0, const uint4 w1, int upto, uint64_t cnt[4]) {
  const uint32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
  const int wi = upto >> 4, r = upto & 15;
  uint32_t c1 = 0, c2 = 0, c3 = 0, total = (uint32_t)upto + 1;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    uint32_t keep = j < wi ? 0xFFFFFFFFu : j == wi ? (0xFFFFFFFFu << ((15 - r) << 1)) : 0u;
    const uint32_t v = w[j];
    const uint32_t lo = v & 0x55555555u, hi = (v >> 1) & 0x55555555u;     // low / high bit of every symbol
    keep &= 0x55555555u;
    c1 += __popc(lo & ~hi & keep);
    c2 += __popc(hi & ~lo & keep);
    c3 += __popc(hi & lo & keep);
  }
  cnt[1] += c1; cnt[2] += c2; cnt[3] += c3; cnt[0] += total - c1 - c2 - c3;
}

// the same over the four words of a compact block (64 symbols)
__device__ __forceinline__ void count_half(const uint4 w4, int upto, uint64_t cnt[4]) {
  const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
  const int wi = upto >> 4, r = upto & 15;
  uint32_t c1 = 0, c2 = 0, c3 = 0, total = (uint32_t)upto + 1;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t keep = j < wi ? 0xFFFFFFFFu : j == wi ? (0xFFFFFFFFu << ((15 - r) << 1)) : 0u;
    const uint32_t v = w[j];
    const uint32_t lo = v & 0x55555555u, hi = (v >> 1) & 0x55555555u;
    keep &= 0x55555555u;
    c1 += __popc(lo & ~hi & keep);
    c2 += __popc(hi & ~lo & keep);
    c3 += __popc(hi & lo & keep);
  }
  cnt[1] += c1; cnt[2] += c2; cnt[3] += c3; cnt[0] += total - c1 - c2 - c3;
}

// bwt_occ4 (baseline.cpp:17-38)
__device__ __forceinline__ void occ4(const Ctx& f, uint64_t k, uint64_t cnt[4]) {
  if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
  k -= (k >= f.primary);
  if (f.compact) {       // wave-uniform: one 32-byte sector per lookup, two loads, four words to count
    const uint4* blk = reinterpret_cast<const uint4*>(f.bwt + ((k >> 6) << 3));
    const uint4 h = blk[0], w = blk[1];
    cnt[0] = h.x; cnt[1] = h.y; cnt[2] = h.z; cnt[3] = h.w;
    count_half(w, (int)(k & 63), cnt);
    return;
  }
  const uint4* blk = reinterpret_cast<const uint4*>(f.bwt + ((k >> 7) << 4));
  const uint4 h0 = blk[0], h1 = blk[1], w0 = blk[2], w1 = blk[3];
  cnt[0] = ((uint64_t)h0.y << 32) | h0.x; cnt[1] = ((uint64_t)h0.w << 32) | h0.z;
  cnt[2] = ((uint64_t)h1.y << 32) | h1.x; cnt[3] = ((uint64_t)h1.w << 32) | h1.z;
  count_block(w0, w1, (int)(k & 127), cnt);
}

// bwt_2occ4 (baseline.cpp:40-85): Occ at k and at l; when both fall into one block (the usual case once the interval is
// narrower than a block) the block is fetched once.  Lanes that do need a second block fetch it under EXEC.
__device__ __forceinline__ void occ4_2(const Ctx& f, uint64_t k, uint64_t l, uint64_t tk[4], uint64_t tl[4]) {
  if (!f.compact || k == (uint64_t)-1 || l == (uint64_t)-1) { occ4(f, k, tk); occ4(f, l, tl); return; }
  k -= (k >= f.primary); l -= (l >= f.primary);
  const uint4* bk = reinterpret_cast<const uint4*>(f.bwt + ((k >> 6) << 3));
  uint4 h = bk[0], w = bk[1];
  tk[0] = h.x; tk[1] = h.y; tk[2] = h.z; tk[3] = h.w;
  count_half(w, (int)(k & 63), tk);
  if ((k >> 6) != (l >> 6)) {
    const uint4* bl = reinterpret_cast<const uint4*>(f.bwt + ((l >> 6) << 3));
    h = bl[0]; w = bl[1];
  }
  tl[0] = h.x; tl[1] = h.y; tl[2] = h.z; tl[3] = h.w;
  count_half(w, (int)(l & 63), tl);
}

// bwt_extend (baseline.cpp:87-100); x[is_back ? 0 : 1] is the strand that is looked up
__device__ __forceinline__ void extend(const Ctx& f, const SmemIntv& ik, SmemIntv ok[4], bool is_back) {
  uint64_t tk[4], tl[4];
  const uint64_t look = is_back ? ik.x0 : ik.x1, other = is_back ? ik.x1 : ik.x0;
  occ4_2(f, look - 1, look - 1 + ik.x2, tk, tl);
  uint64_t lk[4], sz[4];
#pragma unroll
  for (int c = 0; c < 4; c++) { lk[c] = f.L2[c] + 1 + tk[c]; sz[c] = tl[c] - tk[c]; }
  uint64_t o[4];
  o[3] = other + ((look <= f.primary && look + ik.x2 - 1 >= f.primary) ? 1 : 0);
  o[2] = o[3] + sz[3]; o[1] = o[2] + sz[2]; o[0] = o[1] + sz[1];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    ok[c].x0 = is_back ? lk[c] : o[c];
    ok[c].x1 = is_back ? o[c] : lk[c];
    ok[c].x2 = sz[c];
    ok[c].info = 0;
  }
}

__device__ __forceinline__ SmemIntv pick(const SmemIntv ok[4], int c) {
  SmemIntv r = ok[0];
  if (c == 1) r = ok[1]; else if (c == 2) r = ok[2]; else if (c == 3) r = ok[3];
  return r;
}

__device__ __forceinline__ SmemIntv set_intv1(const Ctx& f, int c) {      // baseline.h:6
  SmemIntv ik;
  ik.x0 = f.L2[c] + 1; ik.x2 = f.L2[c + 1] - f.L2[c]; ik.x1 = f.L2[3 - c] + 1; ik.info = 0;
  return ik;
}

struct Lists {            // thread-interleaved scratch
  SmemIntv* base; uint32_t stride;
  __device__ __forceinline__ SmemIntv& curr(int e) const { return base[(size_t)e * stride]; }
  __device__ __forceinline__ SmemIntv& back(int e) const { return base[(size_t)(256 + e) * stride]; }
};

struct Out {
  SmemIntv* a; uint32_t cap; int n;
  __device__ __forceinline__ void push(const SmemIntv& v) { if ((uint32_t)n < cap) a[n] = v; n++; }
};

// bwt_smem1a_new (baseline.cpp:180-304), max_intv = 0
__device__ int smem1a_new(const Ctx& f, int len, const uint8_t* q, int x, int min_intv, Out& mem, const Lists& L) {
  SmemIntv ik, ok[4], temp;
  if (q[x] > 3) return x + 1;
  if (min_intv < 1) min_intv = 1;
  temp.x0 = temp.x1 = temp.x2 = temp.info = 0;
  ik = set_intv1(f, q[x]);
  ik.info = (uint64_t)(x + 1);
  int n_curr = 0, n_back = 0, i;
  for (i = x + 1; i < len; i++) {
    if (q[i] < 4) {
      const int c = 3 - q[i];
      extend(f, ik, ok, false);
      const SmemIntv nx = pick(ok, c);
      if (nx.x2 != ik.x2) { L.curr(n_curr++) = ik; if (nx.x2 < (uint64_t)min_intv) break; }
      ik = nx; ik.info = (uint64_t)(i + 1);
    } else { L.curr(n_curr++) = ik; break; }
  }
  if (i == len) L.curr(n_curr++) = ik;
  const int ret = (int)L.curr(n_curr - 1).info;
  int start = x, stop = x, max_len = 0;
  i = 0;
  while (i < n_curr) {
    const SmemIntv ci = L.curr(i);
    ik = ci;
    ik.info |= (uint64_t)x << 32;
    if (n_back == 0 || stop - start >= 3) {
      n_back = 0;
      L.back(n_back++) = ik;
      for (int k = x - 1; k >= 0; k--) {
        if (q[k] >= 4) break;
        extend(f, ik, ok, true);
        const SmemIntv nx = pick(ok, q[k]);
        if (nx.x2 < (uint64_t)min_intv) break;
        ik = nx;
        ik.info = ci.info | (uint64_t)k << 32;
        L.back(n_back++) = ik;
      }
      start = (int)ci.info;
      stop = (i == n_curr - 1) ? len : (int)L.curr(i + 1).info;
      if (i != 0 && (ik.info >> 32) > (temp.info >> 32) && (int)temp.info - (int)(temp.info >> 32) >= MIN_SEED_LEN) mem.push(temp);
      temp = ik;
    } else {
      stop = (int)ci.info;
      for (int k = n_back - 1; k >= 0; k--) {
        ik = L.back(k);
        bool reached = false;
        for (int m = start + 1; m <= stop; m++) {
          extend(f, ik, ok, false);
          const SmemIntv nx = pick(ok, 3 - q[m - 1]);
          if (nx.x2 < (uint64_t)min_intv) break;
          ik = nx;
          if (m == stop) { ik.info = ci.info | (uint64_t)(x - k) << 32; reached = true; }
        }
        if (reached) {
          if ((uint64_t)(x - k) > (temp.info >> 32) && (int)temp.info - (int)(temp.info >> 32) >= MIN_SEED_LEN) mem.push(temp);
          temp = ik;
          break;
        }
      }
    }
    i++;
    if (i < n_curr) max_len = (int)(temp.info >> 32) + (int)L.curr(i).info;
    while (max_len < MIN_SEED_LEN && i < n_curr) {
      i++;
      if (i < n_curr) stop = (int)L.curr(i).info;
      max_len = (int)(temp.info >> 32) + stop;
    }
    if (i >= n_curr && (int)temp.info - (int)(temp.info >> 32) >= MIN_SEED_LEN) mem.push(temp);
  }
  return ret;
}

// bwt_seed_strategy1 (baseline.cpp:306-327)
__device__ int seed_strategy1(const Ctx& f, int len, const uint8_t* q, int x, int min_len, int max_intv, SmemIntv& mem) {
  SmemIntv ik, ok[4];
  mem.x0 = mem.x1 = mem.x2 = mem.info = 0;
  if (q[x] > 3) return x + 1;
  ik = set_intv1(f, q[x]);
  for (int i = x + 1; i < len; i++) {
    if (q[i] >= 4) return i + 1;
    extend(f, ik, ok, false);
    const SmemIntv nx = pick(ok, 3 - q[i]);
    if (nx.x2 < (uint64_t)max_intv && i - x >= min_len) { mem = nx; mem.info = (uint64_t)x << 32 | (uint64_t)(i + 1); return i + 1; }
    ik = nx;
  }
  return len;
}

__global__ __launch_bounds__(64, SMEM_MIN_WAVES) void smem_kernel(SmemArgs a, uint32_t read_base, uint32_t n_reads) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n_reads) return;
  const uint32_t rd = read_base + tid;
  Ctx f; f.bwt = a.bwt; f.primary = a.primary; f.compact = a.compact != 0;
#pragma unroll
  for (int c = 0; c < 5; c++) f.L2[c] = a.L2[c];
  const uint8_t* q = a.seq + (size_t)rd * a.seq_stride;
  const int len = a.seq_len[rd];
  Lists L; L.base = a.scratch + tid; L.stride = a.n_threads;
  Out mem; mem.a = a.out + (size_t)rd * a.max_out; mem.cap = a.max_out; mem.n = 0;
  // mem_collect_intv_new (baseline.cpp:387-422)
  for (int x = 0; x < len;) x = q[x] < 4 ? smem1a_new(f, len, q, x, 1, mem, L) : x + 1;
  const int old_n = mem.n < (int)mem.cap ? mem.n : (int)mem.cap;   // entries beyond the slot are counted, not kept
  for (int k = 0; k < old_n; k++) {
    const SmemIntv p = mem.a[k];
    const int start = (int)(p.info >> 32), end = (int)(int32_t)p.info;
    if (end - start < 28 || p.x2 > 10) continue;
    smem1a_new(f, len, q, (start + end) >> 1, (int)p.x2 + 1, mem, L);
  }
  for (int x = 0; x < len;) {
    if (q[x] < 4) { SmemIntv m; x = seed_strategy1(f, len, q, x, MIN_SEED_LEN, 20, m); if (m.x2 > 0) mem.push(m); }
    else x++;
  }
  a.mem_num[rd] = mem.n;
}

// ---- lock-step variant -------------------------------------------------------------------------------------
// The direct transcription above lets the 64 reads of a wavefront drift into different loops, so that the expensive
// part - bwt_extend: two 64-byte block reads + ~150 VALU ops - runs with ~22 % of the lanes on average
// (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU, profiles/).  Here every read is an explicit state machine: each round all
// lanes advance their own control flow up to their next bwt_extend request, then the whole wavefront executes ONE
// bwt_extend together.  Same functions, same order of results.
enum : int {
  P1_NEXT, A_INIT, A_FWD, A_FWD_RES, A_BACK_INIT, A_ITER, A_BK_LOOP, A_BK_RES, A_BK_DONE, A_FE_K, A_FE_M, A_FE_RES, A_POST,
  A_RETURN, P2_NEXT, P3_NEXT, P3_LOOP, P3_RES, DONE
};

__global__ __launch_bounds__(64) void smem_kernel_fsm(SmemArgs a, uint32_t read_base, uint32_t n_reads) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = tid < n_reads;
  const uint32_t rd = read_base + (live ? tid : 0);
  Ctx f; f.bwt = a.bwt; f.primary = a.primary; f.compact = a.compact != 0;
#pragma unroll
  for (int c = 0; c < 5; c++) f.L2[c] = a.L2[c];
  const uint8_t* q = a.seq + (size_t)rd * a.seq_stride;
  const int len = live ? a.seq_len[rd] : 0;
  Lists L; L.base = a.scratch + (live ? tid : 0); L.stride = a.n_threads;
  Out mem; mem.a = a.out + (size_t)rd * a.max_out; mem.cap = a.max_out; mem.n = 0;

  int st = live ? P1_NEXT : DONE, pass = 1;
  int x = 0, i = 0, i2 = 0, kk = 0, m = 0, k2 = 0, old_n = 0, min_intv = 1, ret = 0;
  int n_curr = 0, n_back = 0, start = 0, stop = 0, max_len = 0;
  SmemIntv ik, temp, ci, ok[4];
  ik.x0 = ik.x1 = ik.x2 = ik.info = 0; temp = ik; ci = ik;
  bool req = false, req_back = false;

  for (;;) {
    // ---- every lane runs its own control flow up to its next bwt_extend (or to the end) ----
    while (!req && st != DONE) {
      switch (st) {
        case P1_NEXT:                                         // mem_collect_intv_new, first pass (:394-400)
          while (x < len && q[x] >= 4) x++;
          if (x >= len) { old_n = mem.n < (int)mem.cap ? mem.n : (int)mem.cap; k2 = 0; pass = 2; st = P2_NEXT; }
          else { min_intv = 1; st = A_INIT; }
          break;
        case P2_NEXT: {                                       // second pass (:403-408)
          bool go = false;
          while (k2 < old_n && !go) {
            const SmemIntv p = mem.a[k2++];
            const int s0 = (int)(p.info >> 32), e0 = (int)(int32_t)p.info;
            if (e0 - s0 < 28 || p.x2 > 10) continue;
            x = (s0 + e0) >> 1; min_intv = (int)p.x2 + 1; go = true;
          }
          if (go) st = (q[x] > 3) ? P2_NEXT : A_INIT;
          else { x = 0; pass = 3; st = P3_NEXT; }
        } break;
        case A_INIT:                                          // bwt_smem1a_new prologue (:193-197)
          if (min_intv < 1) min_intv = 1;
          ik = set_intv1(f, q[x]); ik.info = (uint64_t)(x + 1);
          n_curr = 0; n_back = 0; i = x + 1; temp.x0 = temp.x1 = temp.x2 = temp.info = 0;
          st = A_FWD;
          break;
        case A_FWD:                                           // forward extension (:199-216)
          if (i >= len) { L.curr(n_curr++) = ik; st = A_BACK_INIT; }
          else if (q[i] < 4) { req = true; req_back = false; st = A_FWD_RES; }
          else { L.curr(n_curr++) = ik; st = A_BACK_INIT; }
          break;
        case A_FWD_RES: {
          const SmemIntv nx = pick(ok, 3 - q[i]);
          bool stop_now = false;
          if (nx.x2 != ik.x2) { L.curr(n_curr++) = ik; stop_now = nx.x2 < (uint64_t)min_intv; }
          if (stop_now) st = A_BACK_INIT;
          else { ik = nx; ik.info = (uint64_t)(i + 1); i++; st = A_FWD; }
        } break;
        case A_BACK_INIT:
          ret = (int)L.curr(n_curr - 1).info;
          start = x; stop = x; max_len = 0; i2 = 0;
          st = A_ITER;
          break;
        case A_ITER:                                          // :220-299
          if (i2 >= n_curr) { st = A_RETURN; break; }
          ci = L.curr(i2);
          ik = ci; ik.info |= (uint64_t)x << 32;
          if (n_back == 0 || stop - start >= 3) { n_back = 0; L.back(n_back++) = ik; kk = x - 1; st = A_BK_LOOP; }
          else { stop = (int)ci.info; kk = n_back - 1; st = A_FE_K; }
          break;
        case A_BK_LOOP:                                       // "backenlarge" (:224-241)
          if (kk < 0 || q[kk] >= 4) st = A_BK_DONE;
          else { req = true; req_back = true; st = A_BK_RES; }
          break;
        case A_BK_RES: {
          const SmemIntv nx = pick(ok, q[kk]);
          if (nx.x2 < (uint64_t)min_intv) st = A_BK_DONE;
          else { ik = nx; ik.info = ci.info | (uint64_t)kk << 32; L.back(n_back++) = ik; kk--; st = A_BK_LOOP; }
        } break;
        case A_BK_DONE:
          start = (int)ci.info;
          stop = (i2 == n_curr - 1) ? len : (int)L.curr(i2 + 1).info;
          if (i2 != 0 && (ik.info >> 32) > (temp.info >> 32) && (int)temp.info - (int)(temp.info >> 32) >= MIN_SEED_LEN) mem.push(temp);
          temp = ik;
          st = A_POST;
          break;
        case A_FE_K:                                          // "forwardenlarge" (:255-281)
          if (kk < 0) { st = A_POST; break; }
          ik = L.back(kk); m = start + 1;
          st = A_FE_M;
          break;
        case A_FE_M:
          if (m > stop) { kk--; st = A_FE_K; }               // empty inner loop: nothing reached
          else { req = true; req_back = false; st = A_FE_RES; }
          break;
        case A_FE_RES: {
          const SmemIntv nx = pick(ok, 3 - q[m - 1]);
          if (nx.x2 < (uint64_t)min_intv) { kk--; st = A_FE_K; break; }
          ik = nx;
          if (m == stop) {
            ik.info = ci.info | (uint64_t)(x - kk) << 32;
            if ((uint64_t)(x - kk) > (temp.info >> 32) && (int)temp.info - (int)(temp.info >> 32) >= MIN_SEED_LEN) mem.push(temp);
            temp = ik;
            st = A_POST;
          } else { m++; st = A_FE_M; }
        } break;
        case A_POST:                                          // :283-298
          i2++;
          if (i2 < n_curr) max_len = (int)(temp.info >> 32) + (int)L.curr(i2).info;
          while (max_len < MIN_SEED_LEN && i2 < n_curr) {
            i2++;
            if (i2 < n_curr) stop = (int)L.curr(i2).info;
            max_len = (int)(temp.info >> 32) + stop;
          }
          if (i2 >= n_curr && (int)temp.info - (int)(temp.info >> 32) >= MIN_SEED_LEN) mem.push(temp);
          st = A_ITER;
          break;
        case A_RETURN:
          if (pass == 1) { x = ret; st = P1_NEXT; } else st = P2_NEXT;
          break;
        case P3_NEXT:                                         // third pass (:411-419) + bwt_seed_strategy1 (:306-327)
          while (x < len && q[x] >= 4) x++;
          if (x >= len) st = DONE;
          else { ik = set_intv1(f, q[x]); i = x + 1; st = P3_LOOP; }
          break;
        case P3_LOOP:
          if (i >= len) { x = len; st = P3_NEXT; }
          else if (q[i] >= 4) { x = i + 1; st = P3_NEXT; }
          else { req = true; req_back = false; st = P3_RES; }
          break;
        case P3_RES: {
          const SmemIntv nx = pick(ok, 3 - q[i]);
          if (nx.x2 < 20 && i - x >= MIN_SEED_LEN) {
            SmemIntv mm = nx; mm.info = (uint64_t)x << 32 | (uint64_t)(i + 1);
            if (mm.x2 > 0) mem.push(mm);
            x = i + 1; st = P3_NEXT;
          } else { ik = nx; i++; st = P3_LOOP; }
        } break;
        default: st = DONE; break;
      }
    }
    if (!__any(req)) break;                                   // every lane is DONE
    // ---- one bwt_extend for the whole wavefront ----
    if (req) { extend(f, ik, ok, req_back); req = false; }
  }
  if (live) a.mem_num[rd] = mem.n;
}

}  // namespace

hipError_t smem_launch(const SmemArgs& a, uint32_t read_base, uint32_t n_reads, hipStream_t s) {
  if (n_reads == 0) return hipSuccess;
  // Measured on configs[4]: the lock-step variant is 35 % SLOWER (43.5 vs 32.1 ms per 2^20 reads) - serialising the cheap
  // per-lane control flow costs more than the divergent bwt_extend it removes, and the loads of a wave bunch up.
  // It stays selectable for A/B runs.
  static const bool fsm = getenv("ACCG_SMEM_FSM") != nullptr;
  if (fsm) hipLaunchKernelGGL(smem_kernel_fsm, dim3((n_reads + 63) / 64), dim3(64), 0, s, a, read_base, n_reads);
  else hipLaunchKernelGGL(smem_kernel, dim3((n_reads + 63) / 64), dim3(64), 0, s, a, read_base, n_reads);
  return hipGetLastError();
}

}  // namespace accg

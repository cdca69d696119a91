// BWA-MEM SMEM seeding (Falcon's three-pass variant) for gfx950: one thread per read.
//
// What it computes: mem_collect_intv_new of the reference (smem/host/baseline.cpp:387-422) = bwt_smem1a_new
// (:180-304) over the read, re-seeding inside long low-occurrence SMEMs, and the LAST-like bwt_seed_strategy1 pass
// (:306-327), on top of bwt_extend / bwt_occ4 (:17-100).  PARITY: bit-exact with the CPU restatement kept with the tests (smem_oracle.c), which restates
// those functions; the reference file itself cannot be built here (libbwa is not in the tree), see DESIGN.md.
//
// Why one thread per read: every bwt_extend depends on the previous one and costs two random block reads, so
// the path is bound by memory latency, not by arithmetic; the only parallelism that hides it is many independent
// reads in flight.  The 64 MB index of configs[4] sits in the 256 MB Infinity Cache.  The per-read
// interval lists (curr / back of bwt_smem1a_new, up to 255 entries each) live in a thread-interleaved HBM scratch so
// that the lanes of a wave touch neighbouring elements.  Occ is computed with popcounts over the 2-bit
// words instead of the reference's byte table; the counts are identical by definition.
//
// Two instantiations: IT = uint64_t over BWA's own blocks, and IT = uint32_t over the half-block layout the host builds
// when the index has fewer than 2^32 symbols (smem_host.cpp) -- then every interval bound fits 32 bits, the interval record
// is 16 bytes instead of 32 and the kernel needs about half the registers (more reads in flight per CU).
#include <stdlib.h>
#include <algorithm>
#include "smem_dev.h"

// -DSMEM_COUNT (a second object built from this file, Makefile): the same kernels with three counters -- half-block (32-byte
// sector) fetches of the index actually performed, prefix-table entries fetched, bwt_extend calls -- so that the roofline of the
// launch is priced on the lookups the kernel really makes, not on the reference's count of requested blocks.  The exported launchers
// get a _count suffix; the product build has none of this.
#ifdef SMEM_COUNT
#define ACCG_SMEM_SUFFIX(n) n##_count
#else
#define ACCG_SMEM_SUFFIX(n) n
#endif

namespace accg {
#ifdef SMEM_COUNT
__device__ unsigned long long g_smem_counts[4];     // sectors fetched, table entries fetched, extend calls, (unused)
hipError_t smem_counts_read(uint64_t out[4], bool reset, hipStream_t s) {
  unsigned long long h[4] = {0, 0, 0, 0};
  hipError_t e = hipStreamSynchronize(s);
  if (e == hipSuccess) e = hipMemcpyFromSymbol(h, HIP_SYMBOL(g_smem_counts), sizeof h);
  for (int i = 0; i < 4; i++) out[i] = h[i];
  if (e == hipSuccess && reset) { const unsigned long long z[4] = {0, 0, 0, 0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_smem_counts), z, sizeof z); }
  return e;
}
#endif
namespace {
#ifdef SMEM_COUNT
// one atomic per wavefront and call site: the lanes that are here together add up first
__device__ __forceinline__ void smem_count(int which, unsigned n) {      // n is 1 or 2
  const unsigned long long m = __ballot(1), m2 = __ballot(n == 2u);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicAdd(&g_smem_counts[which], (unsigned long long)(__popcll(m) + __popcll(m2)));
}
#define ACCG_SMEM_COUNT(which, n) smem_count(which, n)
#else
#define ACCG_SMEM_COUNT(which, n) ((void)0)
#endif

#ifndef SMEM_KEEP_LAST_BACK
#define SMEM_KEEP_LAST_BACK 0
#endif
#ifndef SMEM_SHARE_KL
#define SMEM_SHARE_KL 1      // 1: k and l in one half-block share the fetch (lanes that need a second one fetch it under EXEC); 0: always two fetches
#endif
constexpr int MIN_SEED_LEN = 19;   // smem/common/common.h:37

// bwtintv_t with `info` kept as its two halves: lo = end of the match (query position), hi = its start (or, inside
// bwt_smem1a_new's forward enlargement, the match length)
template <typename IT> struct Intv;
template <> struct Intv<uint64_t> {
  uint64_t x0, x1, x2; uint32_t lo_, hi_;                    // same bytes as SmemIntv {x0, x1, x2, info}
  __device__ __forceinline__ uint32_t lo() const { return lo_; }
  __device__ __forceinline__ uint32_t hi() const { return hi_; }
  __device__ __forceinline__ void set(uint32_t lo, uint32_t hi) { lo_ = lo; hi_ = hi; }
};
template <> struct alignas(16) Intv<uint32_t> {
  uint32_t x0, x1, x2, info;                                 // info = lo | hi << 16 (reads are at most 255 bases)
  __device__ __forceinline__ uint32_t lo() const { return info & 0xFFFFu; }
  __device__ __forceinline__ uint32_t hi() const { return info >> 16; }
  __device__ __forceinline__ void set(uint32_t lo, uint32_t hi) { info = lo | (hi << 16); }
};

template <typename IT>
struct Ctx {
  const uint32_t* bwt;
  IT primary, L2[5];
  const uint4* ktab;        // prefix table (smem_dev.h), null: none
};

// number of symbols equal to c among the first `upto + 1` symbols (0-based, MSB first) of NW words of 16 symbols
template <int NW, typename IT>
__device__ __forceinline__ void count_words(const uint32_t (&w)[NW], int upto, IT cnt[4]) {
  const int wi = upto >> 4, r = upto & 15;
  uint32_t c1 = 0, c2 = 0, c3 = 0, total = (uint32_t)upto + 1;
#pragma unroll
  for (int j = 0; j < NW; j++) {
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

// bwt_occ4 (baseline.cpp:17-38) over BWA's block: 64 B per 128 symbols = 4 x u64 cumulative counts + 8 words
__device__ __forceinline__ void occ4(const Ctx<uint64_t>& f, uint64_t k, uint64_t cnt[4]) {
  if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
  k -= (k >= f.primary);
  const uint4* blk = reinterpret_cast<const uint4*>(f.bwt + ((k >> 7) << 4));
  const uint4 h0 = blk[0], h1 = blk[1], w0 = blk[2], w1 = blk[3];
  cnt[0] = ((uint64_t)h0.y << 32) | h0.x; cnt[1] = ((uint64_t)h0.w << 32) | h0.z;
  cnt[2] = ((uint64_t)h1.y << 32) | h1.x; cnt[3] = ((uint64_t)h1.w << 32) | h1.z;
  const uint32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
  count_words<8>(w, (int)(k & 127), cnt);
}
// bwt_2occ4 (baseline.cpp:40-85)
__device__ __forceinline__ void occ4_2(const Ctx<uint64_t>& f, uint64_t k, uint64_t l, uint64_t tk[4], uint64_t tl[4]) {
  occ4(f, k, tk); occ4(f, l, tl);
}
// ... over the half-block layout: 32 B per 64 symbols = 4 x u32 counts + the 64 symbols as two bit planes (low bits, high bits;
// symbol p of the half-block at bit p).  An extension by base c needs, at k and at l, two numbers only: how many symbols up to there
// are >= c and how many are > c (Occ of c is their difference; the sizes of the bases above c, which stack up the other strand's
// bound, sum to the difference of the second between l and k).  So the header holds the counts cumulated from the top --
// H[j] = symbols >= j in front of the half-block, H[0] = its first position -- and a lane fetches the pair (H[c], H[c+1]) by address
// (H[4] = 0 is not stored: selected).  The host adds T[j] = sum over b >= j of (L2[b] + 1) to H[j] (smem_host.cpp), so that the
// difference of the pair at k is L2[c] + 1 + Occ(c, k) -- the new bound of the strand looked up -- as it stands, while T cancels in
// everything that is a difference between l and k (all arithmetic mod 2^32).
// Both sets are one three-input boolean of the planes and two all-or-nothing words made from c's bits:
//   >= c :  c = 0 all, 1 lo|hi, 2 hi, 3 lo&hi   =  c1 ? hi & (lo | ~c0) : hi | (lo | ~c0)
//   >  c :  c = 0 lo|hi, 1 hi, 2 lo&hi, 3 none  =  c1 ? hi & (lo & ~c0) : hi | (lo & ~c0)
// -- two masked 64-bit popcounts per lookup and no selection among four counts afterwards (round 4: ~105 -> ~65 VALU instructions
// per extension; this kernel's backward halves are bound by instruction issue).
// When k and l fall into one half-block (the usual case once the interval is narrower than a block) it is fetched once; lanes
// that do need a second one fetch it under EXEC.
struct __attribute__((packed, aligned(4))) HdrPair { uint32_t x, y; };      // (H[c], H[c+1]): 4-byte aligned
struct BaseSel { uint32_t c0m, c1m; bool c3; uint32_t hoff; };     // per extension: all-ones words from c's two bits, c == 3, 4 * c
__device__ __forceinline__ BaseSel base_sel(int c) {
  BaseSel b; b.c0m = (uint32_t)-(c & 1); b.c1m = (uint32_t)-((c >> 1) & 1); b.c3 = c == 3; b.hoff = (uint32_t)c << 2;
  return b;
}
__device__ __forceinline__ uint32_t bcnt_add(uint32_t x, uint32_t acc) {      // popcount(x) + acc in one instruction (the compiler prefers v_bcnt + v_add3)
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}
// a = symbols >= c, b = symbols > c among the first r + 1 of the half-block, on top of the header pair hd.  Per 32 bits of the planes
// five v_bitop3/v_and (truth tables over a = 0xF0, b = 0xCC, c = 0xAA) and two v_bcnt that add as they count.
__device__ __forceinline__ void count_ge(const HdrPair hd, const uint4 pl, uint32_t r, const BaseSel& bs, uint32_t& a, uint32_t& b) {
  const uint64_t m = ~0ull >> (63u - r);
  const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
  const uint32_t h0 = pl.z & m0, h1 = pl.w & m1;
  const uint32_t ge0 = __builtin_amdgcn_bitop3_b32(pl.x, bs.c0m, m0, 0xA2), ge1 = __builtin_amdgcn_bitop3_b32(pl.y, bs.c0m, m1, 0xA2);   // (lo | ~c0) & m
  const uint32_t gt0 = __builtin_amdgcn_bitop3_b32(pl.x, bs.c0m, m0, 0x20), gt1 = __builtin_amdgcn_bitop3_b32(pl.y, bs.c0m, m1, 0x20);   // (lo & ~c0) & m
  const uint32_t a0 = __builtin_amdgcn_bitop3_b32(h0, ge0, bs.c1m, 0xD4), a1 = __builtin_amdgcn_bitop3_b32(h1, ge1, bs.c1m, 0xD4);        // c1 ? h & p : h | p
  const uint32_t b0 = __builtin_amdgcn_bitop3_b32(h0, gt0, bs.c1m, 0xD4), b1 = __builtin_amdgcn_bitop3_b32(h1, gt1, bs.c1m, 0xD4);
  a = bcnt_add(a1, bcnt_add(a0, hd.x));
  b = bcnt_add(b1, bcnt_add(b0, bs.c3 ? 0u : hd.y));
}
// bwt_occ4's k == -1 case (:19) cannot occur here and is not tested for: every interval bound this kernel holds is >= 1
// (set_intv1 starts at L2[c] + 1, bwt_extend yields L2[b] + 1 + count and other-strand bound + non-negative terms), so k = bound - 1 >= 0.

// bwt_extend (baseline.cpp:87-100), returning only the interval of base c; x[is_back ? 0 : 1] is the strand that is looked up
__device__ __forceinline__ Intv<uint64_t> extend(const Ctx<uint64_t>& f, const Intv<uint64_t>& ik, bool is_back, int c) {
  typedef uint64_t IT;
  IT tk[4], tl[4];
  const IT look = is_back ? ik.x0 : ik.x1, other = is_back ? ik.x1 : ik.x0;
  occ4_2(f, (IT)(look - 1), (IT)(look - 1 + ik.x2), tk, tl);
  IT sz[4];
#pragma unroll
  for (int b = 0; b < 4; b++) sz[b] = tl[b] - tk[b];
  // the other strand's bounds stack up from base 3 down (o[3] = other + sentinel fix-up, o[b] = o[b+1] + sz[b+1])
  IT o = other + ((look <= f.primary && (IT)(look + ik.x2 - 1) >= f.primary) ? 1 : 0);
  IT lk = f.L2[3] + 1 + tk[3], s = sz[3];
#pragma unroll
  for (int b = 2; b >= 0; b--) {
    if (c <= b) { o += sz[b + 1]; lk = f.L2[b] + 1 + tk[b]; s = sz[b]; }
  }
  Intv<IT> r;
  r.x0 = is_back ? lk : o;
  r.x1 = is_back ? o : lk;
  r.x2 = s;
  r.set(0, 0);
  return r;
}
// ... over the half-blocks (bwt_2occ4 :40-85 and bwt_extend in one: see count_ge)
__device__ __forceinline__ Intv<uint32_t> extend(const Ctx<uint32_t>& f, const Intv<uint32_t>& ik, bool is_back, int c) {
  const uint32_t look = is_back ? ik.x0 : ik.x1, other = is_back ? ik.x1 : ik.x0;
  const uint32_t k0 = look - 1, l0 = look - 1 + ik.x2;
  const uint32_t pk = k0 >= f.primary, pl_ = l0 >= f.primary;       // the sentinel is not stored: positions behind it move up by one
  const uint32_t k = k0 - pk, l = l0 - pl_;
  const BaseSel bs = base_sel(c);
  const char* base = reinterpret_cast<const char*>(f.bwt);
  const uint32_t ok = (k >> 6) << 5, ol = (l >> 6) << 5;
  HdrPair hd = *reinterpret_cast<const HdrPair*>(base + (ok + bs.hoff));
  uint4 pl = *reinterpret_cast<const uint4*>(base + ok + 16);
  uint32_t ak, bk, al, bl;
  count_ge(hd, pl, k & 63u, bs, ak, bk);
  ACCG_SMEM_COUNT(0, ok != ol ? 2u : 1u);
  ACCG_SMEM_COUNT(2, 1u);
#if SMEM_SHARE_KL
  if (ok != ol)
#endif
  {
    hd = *reinterpret_cast<const HdrPair*>(base + (ol + bs.hoff));
    pl = *reinterpret_cast<const uint4*>(base + ol + 16);
  }
  count_ge(hd, pl, l & 63u, bs, al, bl);
  const uint32_t lk = ak - bk;                                      // L2[c] + 1 + Occ(c, k): the header carries L2[c] + 1
  const uint32_t o = other + (pl_ - pk) + (bl - bk);                // the sizes of the bases above c; pl_ - pk = 1 exactly when the sentinel lies inside (k0, l0]: look <= primary <= look + x2 - 1
  Intv<uint32_t> r;
  r.x0 = is_back ? lk : o;
  r.x1 = is_back ? o : lk;
  r.x2 = (al - bl) - lk;
  r.info = 0;
  return r;
}

template <typename IT>
__device__ __forceinline__ Intv<IT> set_intv1(const Ctx<IT>& f, int c) {      // baseline.h:6
  Intv<IT> ik;
  // (selected, not indexed: an array indexed by a lane's value lives in scratch memory, one more round trip in front of every call)
  const IT l0 = f.L2[0], l1 = f.L2[1], l2 = f.L2[2], l3 = f.L2[3], l4 = f.L2[4];
  const bool c0 = c == 0, c1 = c == 1, c2 = c == 2;
  const IT lo = c0 ? l0 : c1 ? l1 : c2 ? l2 : l3;
  const IT up = c0 ? l1 : c1 ? l2 : c2 ? l3 : l4;
  const IT rc = c0 ? l3 : c1 ? l2 : c2 ? l1 : l0;
  ik.x0 = lo + 1; ik.x2 = up - lo; ik.x1 = rc + 1; ik.set(0, 0);
  return ik;
}

template <typename IT>
struct Lists {            // thread-interleaved scratch (lists_of): SMEM_CURR_CAP curr entries, then 256 back entries
  Intv<IT>* base; uint32_t stride;
  int curr0;              // first curr entry of the bwt_smem1a_new call in progress (the split path keeps every call's list)
  __device__ __forceinline__ Intv<IT>& curr(int e) const { return base[(size_t)(curr0 + e) * stride]; }
  __device__ __forceinline__ Intv<IT>& back(int e) const { return base[(size_t)(SMEM_CURR_CAP + e) * stride]; }
};

// Entry e of thread t of the launch at [e * n_threads + t]: a wave instruction touches one run of 1 KB.  (Round 4 measured two other
// layouts on configs[4]: a block of SMEM_SCRATCH_ENTRIES x 64 per wavefront -- every entry a wavefront touches in one or two 2 MB
// pages -- 10.39 against 10.05 ms, and one run of SMEM_SCRATCH_ENTRIES records per lane -- consecutive entries share a 64-byte
// fetch -- 10.73 ms.)
template <typename IT>
__device__ __forceinline__ Lists<IT> lists_of(const SmemArgs& a, uint32_t tid) {
  Lists<IT> L;
  L.base = reinterpret_cast<Intv<IT>*>(a.scratch) + tid; L.stride = a.n_threads; L.curr0 = 0;
  return L;
}

struct Out {
  SmemIntv* a; uint32_t cap; int n;
  template <typename IT>
  __device__ __forceinline__ void push(const Intv<IT>& v) {
    if ((uint32_t)n < cap) { SmemIntv r; r.x0 = v.x0; r.x1 = v.x1; r.x2 = v.x2; r.info = ((uint64_t)v.hi() << 32) | v.lo(); a[n] = r; }
    n++;
  }
};

// The lane's read, 4 bits per base, in LDS (a row of a.read_words words per lane -- the batch's longest read rounded up to an odd
// number of words, so that the lanes of a wavefront spread over the banks; dynamic LDS: 150-base reads take 4.9 KB per wavefront).  Read straight from global memory every q[i] is a byte load from the lane's own 128-byte line: 64 lines per wave
// instruction, no reuse that the L1 could hold on to (24 wavefronts x 64 reads x 128 B per CU) -- on configs[4] that was 20 GB
// of line traffic per launch for 150 MB of bases, on top of the index lookups.
struct ReadLds {
  const uint32_t* w;
  __device__ __forceinline__ int operator[](int i) const { return (int)((w[i >> 3] >> ((i & 7) << 2)) & 0xFu); }
};
__device__ __forceinline__ ReadLds stage_read(uint32_t* s_read, uint32_t row_words, const uint8_t* src, int len, uint32_t row_bytes) {
  uint32_t* my = s_read + threadIdx.x * row_words;
  // eight bases per word of the row.  From a 4-byte-aligned row they arrive as two word loads (while the row has eight bytes left);
  // byte by byte each load waits for the one before (the guard on len makes them conditional): 150 round trips per read
  const bool words = (reinterpret_cast<uintptr_t>(src) & 3u) == 0;
  for (int w8 = 0; w8 * 8 < len; w8++) {
    uint32_t packed = 0;
    if (words && (uint32_t)(w8 * 8 + 8) <= row_bytes) {
      const uint32_t v0 = *reinterpret_cast<const uint32_t*>(src + w8 * 8), v1 = *reinterpret_cast<const uint32_t*>(src + w8 * 8 + 4);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        uint32_t c = ((e < 4 ? v0 : v1) >> ((e & 3) << 3)) & 0xFFu;
        c = w8 * 8 + e < len ? c : 4u;
        packed |= (c > 4u ? 4u : c) << (e << 2);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int pos = w8 * 8 + e;
        const uint32_t c = pos < len ? src[pos] : 4u;
        packed |= (c > 4u ? 4u : c) << (e << 2);
      }
    }
    my[w8] = packed;
  }
  return ReadLds{my};
}

// bwt_smem1a_new (baseline.cpp:180-304), max_intv = 0, in its two halves.
// Forward half (:193-218): extends q[x..] to the right, L.curr(0 .. n_curr) receives the interval in front of every change of
// size; returns n_curr, *ret = where the next call of the first pass starts.
// Are the SMEM_KTAB_L bases from x clean, and which table entry do they spell?  (q[x] < 4 is the caller's.)
template <typename Q>
__device__ __forceinline__ bool ktab_code(const Q& q, int x, int len, uint32_t& code) {
  if (x + SMEM_KTAB_L > len) return false;
  uint32_t c = (uint32_t)q[x];
  bool ok = true;
#pragma unroll
  for (int j = 1; j < SMEM_KTAB_L; j++) { const int b = q[x + j]; ok = ok && b < 4; c = (c << 2) | (uint32_t)(b & 3); }
  code = c;
  return ok;
}
template <typename IT> __device__ __forceinline__ Intv<IT> ktab_entry(const uint4* tab, int L, uint32_t code);
template <> __device__ __forceinline__ Intv<uint32_t> ktab_entry<uint32_t>(const uint4* tab, int L, uint32_t code) {
  const uint4 e = tab[smem_ktab_off(L) + (code >> (2 * (SMEM_KTAB_L - L)))];
  ACCG_SMEM_COUNT(1, 1u);
  Intv<uint32_t> r; r.x0 = e.x; r.x1 = e.y; r.x2 = e.z; r.info = 0;
  return r;
}
template <> __device__ __forceinline__ Intv<uint64_t> ktab_entry<uint64_t>(const uint4*, int, uint32_t) { Intv<uint64_t> r; r.x0 = r.x1 = r.x2 = 0; r.set(0, 0); return r; }

template <typename IT, typename Q>
__device__ __forceinline__ int smem1a_fwd(const Ctx<IT>& f, int len, const Q& q, int x, int min_intv, const Lists<IT>& L, int* ret) {
  typedef Intv<IT> I;
  if (min_intv < 1) min_intv = 1;
  I ik = set_intv1(f, q[x]);
  ik.set((uint32_t)(x + 1), 0);
  int n_curr = 0, i = x + 1;
  bool done = false;
  uint32_t code;
  if (sizeof(IT) == 4 && f.ktab && ktab_code(q, x, len, code)) {
    // the intervals of q[x .. x+2), ... q[x .. x+SMEM_KTAB_L) from the prefix table, all loads in flight together; then the loop
    // body below for each of them (the same pushes: an entry holds exactly what the extension would have returned)
    I e[SMEM_KTAB_L - 1];
#pragma unroll
    for (int j = 0; j < SMEM_KTAB_L - 1; j++) e[j] = ktab_entry<IT>(f.ktab, j + 2, code);
#pragma unroll
    for (int j = 0; j < SMEM_KTAB_L - 1; j++) {
      if (!done) {
        const I nx = e[j];
        if (nx.x2 != ik.x2) { L.curr(n_curr++) = ik; done = nx.x2 < (IT)min_intv; }
        if (!done) { ik = nx; ik.set((uint32_t)(i + 1), 0); i++; }
      }
    }
  }
  if (!done) {
    for (; i < len; i++) {
      if (q[i] < 4) {
        const I nx = extend(f, ik, false, 3 - q[i]);
        if (nx.x2 != ik.x2) { L.curr(n_curr++) = ik; if (nx.x2 < (IT)min_intv) break; }
        ik = nx; ik.set((uint32_t)(i + 1), 0);
      } else { L.curr(n_curr++) = ik; break; }
    }
    if (i == len) L.curr(n_curr++) = ik;
  }
  *ret = (int)ik.lo();             // = L.curr(n_curr - 1).lo(): every way out of the loops ends with a push of ik as it stands
  return n_curr;
}
// Backward half (:219-299): over the n_curr entries of L.curr.
//
// What the reference does per list entry [x, end): find the longest backward extension [x - k, end) that still has min_intv
// occurrences, report the previous winner (`temp`) if the new one starts later, and make the new one `temp`.  It has two ways to
// find it: "backenlarge" (:224-253) extends the entry backwards base by base from x - 1 and rebuilds the back list (the
// extensions of that entry by 0, 1, 2 ... bases), run for the first entry and whenever the end has moved 3 bases beyond the last
// such run; "forwardenlarge" (:255-281) tries the back-list entries from the longest down, enlarging each forwards to the new
// end, until one survives.  Both compute the same thing -- occurrence counts only fall when a string grows, so the k they find
// is the largest k whose string [x - k, end) survives, and the interval of a string does not depend on how it was grown -- and
// they differ only in what they write to `start` / `stop` afterwards (which steers the entries to come).  So this function
// keeps the reference's decisions and bookkeeping and is free in HOW it finds k for an entry:
//   * the search never tries an entry longer than the last winner (it failed for a shorter end already) and starts from the last
//     winner's interval, already enlarged to the previous end (`temp`, `m_done`);
//   * any entry is found by that search when its end is only a few bases beyond the last one and the last winner survives, and by
//     the backward chain (with the prefix table for its first steps) when that is the shorter way -- the first entry, typically
//     the last one, whose end is far out, and whenever the winner is lost and the back list is more than 3 bases behind;
//   * otherwise entries shorter than the winner are enlarged from the back list of the last backward chain (`b_start`).
// Measured on configs[4] (fused kernel): 12.4 -> 11.1 ms with the first point, 10.2 ms with all three.
template <typename IT, typename Q>
__device__ void smem1a_back(const Ctx<IT>& f, int len, const Q& q, int x, int min_intv, Out& mem, const Lists<IT>& L, int n_curr) {
  typedef Intv<IT> I;
  I ik, temp;
  if (min_intv < 1) min_intv = 1;
  temp.x0 = temp.x1 = temp.x2 = 0; temp.set(0, 0);
  int n_back = 0, i;
  int start = x, stop = x, max_len = 0;
  int k_try = -1, m_done = 0;  // the last winner's k, and the end its interval (`temp`) has been enlarged to
  int b_start = x;             // end of the strings in the back list (set by the last backward chain)
  i = 0;
  // the entry behind the one in hand is fetched whole where the bookkeeping first wants its end, and is the entry in hand of the
  // next round (one load per entry instead of three)
  I c_next = L.curr(0);          // always entry i at the top of the loop
  while (i < n_curr) {
    const I ci = c_next;
    const int end = (int)ci.lo();
    const bool is_back = n_back == 0 || stop - start >= 3;        // the reference's choice: decides the bookkeeping below
    const int ell = end - x;                                        // bases matched so far (a curr entry starts at x)
    // which way is shorter: enlarging the last winner by end - m_done bases, or a chain of at most k_try + 1 backward steps of
    // which the prefix table answers those that stay within SMEM_KTAB_L bases
    bool chain = n_back == 0;
    // (a list entry below the bound itself -- possible only for a first base rarer than the bound -- has no surviving extension at
    // all: "forwardenlarge" finds nothing and leaves everything as it is, "backenlarge" keeps the entry as it stands)
    const bool hopeless = !is_back && ci.x2 < (IT)min_intv;
    if (!chain && !hopeless) {
      const int by_table = (sizeof(IT) == 4 && f.ktab && ell < SMEM_KTAB_L) ? SMEM_KTAB_L - ell : 0;
      const int chain_cost = k_try + 1 > by_table ? k_try + 1 - by_table : 1;
      chain = end - m_done > chain_cost;
    }
    int k_found = -1;
    if (!chain && !hopeless) {
      // the last winner first, from where it stands; if it does not survive, the shorter ones from the back list -- unless that
      // list is stale (its strings end more than 3 bases back, where the reference would have rebuilt it): then the chain
      int k = k_try;
      bool resume = true;
      for (; k >= 0; k--) {
        int m = b_start + 1;
        if (resume) { ik = temp; m = m_done + 1; }
        else ik = L.back(k);
        bool reached = m > end;                              // (cannot happen: ends only grow)
        for (; m <= end; m++) {
          const I nx = extend(f, ik, false, 3 - q[m - 1]);
          if (nx.x2 < (IT)min_intv) break;
          ik = nx;
          if (m == end) reached = true;
        }
        if (reached) { ik.set(ci.lo(), ci.hi() | (uint32_t)(x - k)); k_found = k; break; }
        if (resume && end - b_start > 3) break;
        resume = false;
      }
      if (k_found < 0) chain = true;
    }
    if (chain) {
      ik = ci;
      ik.set(ci.lo(), ci.hi() | (uint32_t)x);
      n_back = 0;
      // nothing reads the back list behind the call's last entry (the next call starts an empty one): its chain -- the longest, the
      // match is unique by then and runs back to the previous mismatch -- only counts its steps
      const bool keep = SMEM_KEEP_LAST_BACK || i != n_curr - 1;
      if (keep) L.back(n_back) = ik;
      n_back++;
      int k = x - 1;
      bool stopped = false;
      // The interval of q[k .. end) is the prefix table's entry for that string whichever side it grew from, so while the match
      // is shorter than SMEM_KTAB_L bases its backward extensions are independent table loads instead of a chain of lookups
      // (an entry that is empty only has to say so: the loop leaves before it would keep one).
      if (sizeof(IT) == 4 && f.ktab && ell < SMEM_KTAB_L && k >= 0) {
        uint32_t code0 = 0;
        for (int j = 0; j < ell; j++) code0 = (code0 << 2) | (uint32_t)(q[x + j] & 3);
        // first the sizes alone (one register per step), all loads in flight: how many steps succeed
        uint32_t sz[SMEM_KTAB_L - 1];
        uint32_t code = code0;
        int bad = SMEM_KTAB_L;                              // first step whose base is ambiguous or in front of the read
#pragma unroll
        for (int j = 1; j < SMEM_KTAB_L; j++) {
          sz[j - 1] = 0;
          if (ell + j <= SMEM_KTAB_L && j < bad) {
            const int b = x - j >= 0 ? q[x - j] : 4;
            if (b >= 4) bad = j;
            else { code |= (uint32_t)b << (2 * (ell + j - 1)); sz[j - 1] = f.ktab[smem_ktab_off(ell + j) + code].z; }
          }
        }
        int n_ok = 0;
#pragma unroll
        for (int j = 1; j < SMEM_KTAB_L; j++)
          if (ell + j <= SMEM_KTAB_L && n_ok == j - 1 && j < bad && sz[j - 1] >= (uint32_t)min_intv) n_ok = j;
        stopped = n_ok < SMEM_KTAB_L - ell;                 // the loop ended inside the table's range
        // then the entries of the steps that did
        code = code0;
        for (int j = 1; j <= n_ok; j++) {
          code |= (uint32_t)(q[x - j] & 3) << (2 * (ell + j - 1));
          const uint4 t = f.ktab[smem_ktab_off(ell + j) + code];
          ik.x0 = (IT)t.x; ik.x1 = (IT)t.y; ik.x2 = (IT)t.z;
          ik.set(ci.lo(), ci.hi() | (uint32_t)(x - j));
          if (keep) L.back(n_back) = ik;
          n_back++;
        }
        k = x - n_ok - 1;
      }
      if (!stopped)
      for (; k >= 0; k--) {
        if (q[k] >= 4) break;
        const I nx = extend(f, ik, true, q[k]);
        if (nx.x2 < (IT)min_intv) break;
        ik = nx;
        ik.set(ci.lo(), ci.hi() | (uint32_t)k);
        if (keep) L.back(n_back) = ik;
        n_back++;
      }
      b_start = end;
      k_found = n_back - 1;
    }
    // the reference's bookkeeping (:242-253 behind "backenlarge", :255 and :270-279 around "forwardenlarge")
    if (i + 1 < n_curr) c_next = L.curr(i + 1);
    if (is_back) { start = end; stop = (i == n_curr - 1) ? len : (int)c_next.lo(); }
    else stop = end;
    if (k_found >= 0) {
      if (i != 0 && ik.hi() > temp.hi() && (int)temp.lo() - (int)temp.hi() >= MIN_SEED_LEN) mem.push(temp);
      temp = ik;
      k_try = k_found; m_done = end;
    }
    i++;
    if (i < n_curr) max_len = (int)temp.hi() + (int)c_next.lo();
    while (max_len < MIN_SEED_LEN && i < n_curr) {
      i++;
      if (i < n_curr) { c_next = L.curr(i); stop = (int)c_next.lo(); }
      max_len = (int)temp.hi() + stop;
    }
    if (i >= n_curr && (int)temp.lo() - (int)temp.hi() >= MIN_SEED_LEN) mem.push(temp);
  }
}
template <typename IT, typename Q>
__device__ int smem1a_new(const Ctx<IT>& f, int len, const Q& q, int x, int min_intv, Out& mem, const Lists<IT>& L) {
  if (q[x] > 3) return x + 1;
  int ret;
  const int n_curr = smem1a_fwd(f, len, q, x, min_intv, L, &ret);
  smem1a_back(f, len, q, x, min_intv, mem, L, n_curr);
  return ret;
}

// bwt_seed_strategy1 (baseline.cpp:306-327)
template <typename IT, typename Q>
__device__ int seed_strategy1(const Ctx<IT>& f, int len, const Q& q, int x, int min_len, int max_intv, Intv<IT>& mem) {
  Intv<IT> ik;
  mem.x0 = mem.x1 = mem.x2 = 0; mem.set(0, 0);
  if (q[x] > 3) return x + 1;
  ik = set_intv1(f, q[x]);
  int i = x + 1;
  uint32_t code;      // the first SMEM_KTAB_L - 1 extensions cannot report anything (min_len): their result comes from the prefix table
  if (sizeof(IT) == 4 && f.ktab && SMEM_KTAB_L - 1 < min_len && ktab_code(q, x, len, code)) { ik = ktab_entry<IT>(f.ktab, SMEM_KTAB_L, code); i = x + SMEM_KTAB_L; }
  for (; i < len; i++) {
    if (q[i] >= 4) return i + 1;
    const Intv<IT> nx = extend(f, ik, false, 3 - q[i]);
    if (nx.x2 < (IT)max_intv && i - x >= min_len) { mem = nx; mem.set((uint32_t)(i + 1), (uint32_t)x); return i + 1; }
    ik = nx;
  }
  return len;
}

#ifndef SMEM_WAVES_PER_EU
#define SMEM_WAVES_PER_EU 5     // what the allocator reaches on its own (91 VGPRs); 6 (80 VGPRs, 16 bytes spilled) and 8 (64, 80 bytes) measure the same or worse
#endif
// One read through mem_collect_intv_new (baseline.cpp:387-422): the lane's lists L, its LDS row for the read.
template <typename IT>
__device__ __forceinline__ void smem_one_read(const SmemArgs& a, const Ctx<IT>& f, uint32_t rd, const Lists<IT>& L, uint32_t* s_read) {
  const int len = a.seq_len[rd];
  const ReadLds q = stage_read(s_read, a.read_words, a.seq + (size_t)rd * a.seq_stride, len, a.seq_stride);
  Out mem; mem.a = a.out + (size_t)rd * a.max_out; mem.cap = a.max_out; mem.n = 0;
  for (int x = 0; x < len;) x = q[x] < 4 ? smem1a_new(f, len, q, x, 1, mem, L) : x + 1;
  const int old_n = mem.n < (int)mem.cap ? mem.n : (int)mem.cap;   // entries beyond the slot are counted, not kept
  for (int k = 0; k < old_n; k++) {
    const SmemIntv p = mem.a[k];
    const int start = (int)(p.info >> 32), end = (int)(int32_t)p.info;
    if (end - start < 28 || p.x2 > 10) continue;
    smem1a_new(f, len, q, (start + end) >> 1, (int)p.x2 + 1, mem, L);
  }
  if (!a.skip_pass3)
  for (int x = 0; x < len;) {
    if (q[x] < 4) { Intv<IT> m; x = seed_strategy1(f, len, q, x, MIN_SEED_LEN, 20, m); if (m.x2 > 0) mem.push(m); }
    else x++;
  }
  a.mem_num[rd] = mem.n;
}
template <typename IT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SMEM_WAVES_PER_EU))) void smem_kernel(SmemArgs a, uint32_t read_base, uint32_t n_reads) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n_reads) return;
  Ctx<IT> f; f.bwt = a.bwt; f.primary = (IT)a.primary; f.ktab = sizeof(IT) == 4 ? a.ktab : nullptr;
#pragma unroll
  for (int c = 0; c < 5; c++) f.L2[c] = (IT)a.L2[c];
  extern __shared__ uint32_t s_read[];
  smem_one_read<IT>(a, f, read_base + tid, lists_of<IT>(a, tid), s_read);
}


// ---- the same work in three kernels (ACCG_SMEM_SPLIT=1) -----------------------------------------------------------------------
// In the fused kernel a wavefront executes the union of its 64 reads' paths through three nested passes; counted on configs[4]
// 29 % of the Occ lookups are the forward extensions of the first pass and 28 % the LAST-like third pass -- both plain "extend to
// the right until it fails" loops whose total trip count per read is nearly the same for every read (the read length), and both
// independent of everything else: the first pass restarts where the forward extension failed (bwt_smem1a_new returns that
// position before it looks backwards), the third pass only appends behind the other two.  Run on their own, flattened to one
// extension per iteration, they keep all 64 lanes busy:
//   smem_fwd_kernel    forward halves of the whole first pass; the curr lists of all calls go to the scratch back to back, with a
//                      segment table {x, first entry, entries} per read
//   smem_back_kernel   backward halves over those lists, then the re-seeding pass (as in the fused kernel)
//   smem_pass3_kernel  the third pass, appending behind what the second kernel stored
// Same functions, same order of every push: the output is the fused kernel's bit for bit.
// MEASURED (configs[4], round 2): the forward and third-pass kernels do run with 55 and 63 of 64 lanes and sit on the random-sector
// ceiling of the memory system (3.3 + 2.5 ms at 12 wavefronts per CU -- more wavefronts make them SLOWER, 4.8 + 3.9 ms at 28 --
// against 3.2 ms for the same number of dependent sector pairs in tools/ubench_random.hip), but the backward kernel alone takes
// 9.1 ms: 14.9 ms in total against 13.9 ms for the one-kernel form, in which the issue-bound backward code of some wavefronts
// overlaps the memory-bound forward code of others (taking the third pass out of the fused kernel saves it only 1 ms).  Kept
// selectable and under test; the default is the fused kernel.
template <typename IT>
__global__ __launch_bounds__(64) void smem_fwd_kernel(SmemArgs a, uint32_t read_base, uint32_t n_reads) {
  typedef Intv<IT> I;
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n_reads) return;
  const uint32_t rd = read_base + tid;
  Ctx<IT> f; f.bwt = a.bwt; f.primary = (IT)a.primary; f.ktab = sizeof(IT) == 4 ? a.ktab : nullptr;
#pragma unroll
  for (int c = 0; c < 5; c++) f.L2[c] = (IT)a.L2[c];
  const int len = a.seq_len[rd];
  extern __shared__ uint32_t s_read[];
  const ReadLds q = stage_read(s_read, a.read_words, a.seq + (size_t)rd * a.seq_stride, len, a.seq_stride);
  Lists<IT> L = lists_of<IT>(a, tid);
  uint32_t* seg = a.seg + tid;
  // one loop, one bwt_extend per iteration: a lane is either inside a forward extension (i < len, q[i] < 4) or between two
  int n_tot = 0, n_seg = 0, x = 0, i = 0, seg0 = 0;
  bool inside = false;
  I ik; ik.x0 = ik.x1 = ik.x2 = 0; ik.set(0, 0);
  while (x < len) {
    if (!inside) {                       // mem_collect_intv_new's loop head (:394-400) + bwt_smem1a_new's prologue (:193-197)
      if (q[x] > 3) { x++; continue; }
      ik = set_intv1(f, q[x]); ik.set((uint32_t)(x + 1), 0);
      i = x + 1; seg0 = n_tot; inside = true;
    }
    bool end = false;
    if (i < len && q[i] < 4) {
      const I nx = extend(f, ik, false, 3 - q[i]);
      if (nx.x2 != ik.x2) { L.curr(n_tot++) = ik; end = nx.x2 < (IT)1; }
      if (!end) { ik = nx; ik.set((uint32_t)(i + 1), 0); i++; }
    } else { L.curr(n_tot++) = ik; end = true; }          // ambiguous base, or the end of the read (:211-216)
    if (end) {
      seg[(size_t)n_seg * a.n_threads] = (uint32_t)x | ((uint32_t)seg0 << 8) | ((uint32_t)(n_tot - seg0) << 18);
      n_seg++;
      x = (int)L.curr(n_tot - 1).lo();                    // ret (:217)
      inside = false;
    }
  }
  a.nseg[tid] = (uint32_t)n_seg;
}

// Backward halves of the first pass over the lists the forward kernel left, then the re-seeding pass (as in the fused kernel).
// Two other forms of this kernel were built and measured on configs[4], both bit-exact (DESIGN.md 4b has the numbers): one in
// which the wavefront votes on which inner loop runs next (short "forwardenlarge" loops first, long "backenlarge" loops together),
// and one flat loop with three extension states in which every round executes one bwt_extend for all lanes (25 of 64 lanes
// instead of 10).  Both were slower than these nested loops (10.8 and 12.5 ms against 9.1 ms): with more lanes per round every
// round waits for the slowest of more random loads -- the index sectors and, between two inner loops, the lane's list entries
// in the HBM scratch -- and the launch is bound by that latency, not by the instruction count.
template <typename IT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SMEM_WAVES_PER_EU))) void smem_back_kernel(SmemArgs a, uint32_t read_base, uint32_t n_reads) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n_reads) return;
  const uint32_t rd = read_base + tid;
  Ctx<IT> f; f.bwt = a.bwt; f.primary = (IT)a.primary; f.ktab = sizeof(IT) == 4 ? a.ktab : nullptr;
#pragma unroll
  for (int c = 0; c < 5; c++) f.L2[c] = (IT)a.L2[c];
  const int len = a.seq_len[rd];
  extern __shared__ uint32_t s_read[];
  const ReadLds q = stage_read(s_read, a.read_words, a.seq + (size_t)rd * a.seq_stride, len, a.seq_stride);
  Lists<IT> L = lists_of<IT>(a, tid);
  Out mem; mem.a = a.out + (size_t)rd * a.max_out; mem.cap = a.max_out; mem.n = 0;
  const int n_seg = (int)a.nseg[tid];
  for (int sgi = 0; sgi < n_seg; sgi++) {                 // first pass: the backward half of every call, on the list the forward kernel left
    const uint32_t sg = a.seg[(size_t)sgi * a.n_threads + tid];
    L.curr0 = (int)((sg >> 8) & 0x3FFu);
    smem1a_back(f, len, q, (int)(sg & 0xFFu), 1, mem, L, (int)(sg >> 18));
  }
  L.curr0 = 0;
  const int old_n = mem.n < (int)mem.cap ? mem.n : (int)mem.cap;   // entries beyond the slot are counted, not kept
  for (int k = 0; k < old_n; k++) {
    const SmemIntv p = mem.a[k];
    const int start = (int)(p.info >> 32), end = (int)(int32_t)p.info;
    if (end - start < 28 || p.x2 > 10) continue;
    smem1a_new(f, len, q, (start + end) >> 1, (int)p.x2 + 1, mem, L);
  }
  a.mem_num[rd] = mem.n;
}

template <typename IT>
__global__ __launch_bounds__(64) void smem_pass3_kernel(SmemArgs a, uint32_t read_base, uint32_t n_reads) {
  typedef Intv<IT> I;
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n_reads) return;
  const uint32_t rd = read_base + tid;
  Ctx<IT> f; f.bwt = a.bwt; f.primary = (IT)a.primary; f.ktab = sizeof(IT) == 4 ? a.ktab : nullptr;
#pragma unroll
  for (int c = 0; c < 5; c++) f.L2[c] = (IT)a.L2[c];
  const int len = a.seq_len[rd];
  extern __shared__ uint32_t s_read[];
  const ReadLds q = stage_read(s_read, a.read_words, a.seq + (size_t)rd * a.seq_stride, len, a.seq_stride);
  Out mem; mem.a = a.out + (size_t)rd * a.max_out; mem.cap = a.max_out; mem.n = a.mem_num[rd];
  // bwt_seed_strategy1 (:306-327) from every restart point (:411-419), one bwt_extend per iteration
  int x = 0, i = 0;
  bool inside = false;
  I ik; ik.x0 = ik.x1 = ik.x2 = 0; ik.set(0, 0);
  while (x < len) {
    if (!inside) {
      if (q[x] > 3) { x++; continue; }
      ik = set_intv1(f, q[x]); i = x + 1; inside = true;
      uint32_t code;  // (see seed_strategy1)
      if (sizeof(IT) == 4 && f.ktab && ktab_code(q, x, len, code)) { ik = ktab_entry<IT>(f.ktab, SMEM_KTAB_L, code); i = x + SMEM_KTAB_L; }
    }
    if (i >= len) break;                                   // `return len`: nothing reported, the scan is over
    if (q[i] >= 4) { x = i + 1; inside = false; continue; }
    const I nx = extend(f, ik, false, 3 - q[i]);
    if (nx.x2 < 20 && i - x >= MIN_SEED_LEN) {
      I m = nx; m.set((uint32_t)(i + 1), (uint32_t)x);
      if (m.x2 > 0) mem.push(m);
      x = i + 1; inside = false;
    } else { ik = nx; i++; }
  }
  a.mem_num[rd] = mem.n;
}

// ---- engine variant (32-bit intervals only) ----------------------------------------------------------------------------
// In the kernel above the 64 reads of a wavefront share one program counter: a read that leaves a loop early idles until the
// slowest one is through, and bwt_extend -- the only expensive step -- runs with 14 of 64 lanes on average
// (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU).  The table itself can serve 2.6x the lookups that kernel asks for
// (tools/ubench_random.hip: 180 G random sectors/s out of a 64 MB table with full wavefronts).  Here every lane is a small
// state machine over the same functions: each round all lanes that want a bwt_extend get it together (full wavefront, 64
// lookups in flight per instruction), then each lane consumes its result and, in the common case (it stays inside the loop it
// is in), posts its next request at once; only lanes that leave a loop go through the general transition switch.  Lanes that
// finish their read take the next one from a queue, so a wavefront never waits for its slowest read.
// MEASURED (configs[4]): bit-exact, VALU lanes 21 of 64 instead of 14, the same fabric traffic -- but 27.7 ms against 17.5 ms:
// the transition code reads the curr / back lists from the HBM scratch with dependent loads, and that latency now stalls
// the whole wavefront every few rounds instead of one subset of lanes.  Kept selectable (ACCG_SMEM_ENGINE=1) and under test
// as the starting point for a version with the lists' hot entries in LDS; the default is the plain kernel above.
enum : int {
  E_FETCH, P1_NEXT, A_INIT, A_FWD_RES, A_BACK_INIT, A_ITER, A_BK_RES, A_BK_DONE, A_FE_K, A_FE_RES, A_POST, A_RETURN, P2_NEXT,
  P3_NEXT, P3_RES, E_DONE
};

#ifndef SMEM_SLOW_BATCH
#define SMEM_SLOW_BATCH 12
#endif
__global__ __launch_bounds__(64) void smem_engine(SmemArgs a, uint32_t read_base, uint32_t n_reads) {
  typedef uint32_t IT;
  typedef Intv<IT> I;
  constexpr int SLOW_BATCH = SMEM_SLOW_BATCH;
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  Ctx<IT> f; f.bwt = a.bwt; f.primary = (IT)a.primary; f.ktab = sizeof(IT) == 4 ? a.ktab : nullptr;
#pragma unroll
  for (int c = 0; c < 5; c++) f.L2[c] = (IT)a.L2[c];
  Lists<IT> L = lists_of<IT>(a, tid);
  // the lane's current read, 4 bits per base, in LDS (row of 128 B per lane, rows 132 B apart to spread the banks): the
  // state machine looks at a base of the read in almost every transition
  __shared__ uint32_t s_read[64 * 33];
  uint32_t* my_read = s_read + threadIdx.x * 33;
  struct ReadView {
    const uint32_t* w;
    __device__ __forceinline__ int operator[](int i) const { return (int)((w[i >> 3] >> ((i & 7) << 2)) & 0xFu); }
  } q{my_read};
  int len = 0;
  uint32_t rd = 0;
  Out mem; mem.a = a.out; mem.cap = a.max_out; mem.n = 0;

  int st = E_FETCH, pass = 1;
  int x = 0, i = 0, i2 = 0, kk = 0, m = 0, k2 = 0, old_n = 0, min_intv = 1, ret = 0;
  int n_curr = 0, n_back = 0, start = 0, stop = 0, max_len = 0;
  I ik, temp, ci, nx;
  ik.x0 = ik.x1 = ik.x2 = 0; ik.set(0, 0); temp = ik; ci = ik; nx = ik;
  bool req = false, req_back = false;
  int req_c = 0;

  // forward step of bwt_smem1a_new's first loop (:199-216) at position i: request, or close the list and go backwards
  auto fwd_step = [&]() {
    if (i < len && q[i] < 4) { req = true; req_back = false; req_c = 3 - q[i]; st = A_FWD_RES; }
    else { L.curr(n_curr++) = ik; st = A_BACK_INIT; }
  };
  auto bk_step = [&]() {                                   // "backenlarge" loop head (:224-241)
    if (kk < 0 || q[kk] >= 4) st = A_BK_DONE;
    else { req = true; req_back = true; req_c = q[kk]; st = A_BK_RES; }
  };
  auto p3_step = [&]() {                                   // bwt_seed_strategy1 loop head (:312-325)
    if (i >= len) { x = len; st = P3_NEXT; }
    else if (q[i] >= 4) { x = i + 1; st = P3_NEXT; }
    else { req = true; req_back = false; req_c = 3 - q[i]; st = P3_RES; }
  };

  for (;;) {
    // ---- transitions of the lanes that are between two loops ------------------------------------------------------------
    // Run them in batches: a lane between two loops waits (without a request) until SLOW_BATCH lanes are in that position or
    // nobody has a request left -- the transition code is executed by the whole wavefront whoever needs it.
    const unsigned long long slow = __ballot(!req && st != E_DONE);
    if (__popcll(slow) >= SLOW_BATCH || !__any(req))
    while (__any(!req && st != E_DONE)) {
      if (!req && st != E_DONE) {
        switch (st) {
          case E_FETCH: {                                       // next read of the launch
            const uint32_t r = atomicAdd(a.queue, 1u);
            if (r >= n_reads) { st = E_DONE; break; }
            rd = read_base + r;
            len = a.seq_len[rd];
            {
              const uint8_t* src = a.seq + (size_t)rd * a.seq_stride;
              for (int w8 = 0; w8 * 8 < len; w8++) {
                uint32_t packed = 0;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                  const int pos = w8 * 8 + e;
                  const uint32_t c = pos < len ? src[pos] : 4u;
                  packed |= (c > 4u ? 4u : c) << (e << 2);
                }
                my_read[w8] = packed;
              }
            }
            mem.a = a.out + (size_t)rd * a.max_out; mem.n = 0;
            x = 0; pass = 1; st = P1_NEXT;
          } break;
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
            ik = set_intv1(f, q[x]); ik.set((uint32_t)(x + 1), 0);
            n_curr = 0; n_back = 0; i = x + 1; temp.x0 = temp.x1 = temp.x2 = 0; temp.set(0, 0);
            fwd_step();
            break;
          case A_BACK_INIT:
            ret = (int)L.curr(n_curr - 1).lo();
            start = x; stop = x; max_len = 0; i2 = 0;
            st = A_ITER;
            break;
          case A_ITER:                                          // :220-299
            if (i2 >= n_curr) { st = A_RETURN; break; }
            ci = L.curr(i2);
            ik = ci; ik.set(ci.lo(), ci.hi() | (uint32_t)x);
            if (n_back == 0 || stop - start >= 3) { n_back = 0; L.back(n_back++) = ik; kk = x - 1; bk_step(); }
            else { stop = (int)ci.lo(); kk = n_back - 1; st = A_FE_K; }
            break;
          case A_BK_DONE:
            start = (int)ci.lo();
            stop = (i2 == n_curr - 1) ? len : (int)L.curr(i2 + 1).lo();
            if (i2 != 0 && ik.hi() > temp.hi() && (int)temp.lo() - (int)temp.hi() >= MIN_SEED_LEN) mem.push(temp);
            temp = ik;
            st = A_POST;
            break;
          case A_FE_K:                                          // "forwardenlarge" (:255-281)
            if (kk < 0) { st = A_POST; break; }
            ik = L.back(kk); m = start + 1;
            if (m > stop) { kk--; break; }                      // empty inner loop: nothing reached, next k
            req = true; req_back = false; req_c = 3 - q[m - 1]; st = A_FE_RES;
            break;
          case A_POST:                                          // :283-298
            i2++;
            if (i2 < n_curr) max_len = (int)temp.hi() + (int)L.curr(i2).lo();
            while (max_len < MIN_SEED_LEN && i2 < n_curr) {
              i2++;
              if (i2 < n_curr) stop = (int)L.curr(i2).lo();
              max_len = (int)temp.hi() + stop;
            }
            if (i2 >= n_curr && (int)temp.lo() - (int)temp.hi() >= MIN_SEED_LEN) mem.push(temp);
            st = A_ITER;
            break;
          case A_RETURN:
            if (pass == 1) { x = ret; st = P1_NEXT; } else st = P2_NEXT;
            break;
          case P3_NEXT:                                         // third pass (:411-419) + bwt_seed_strategy1 (:306-327)
            while (x < len && q[x] >= 4) x++;
            if (x >= len) { a.mem_num[rd] = mem.n; st = E_FETCH; }
            else { ik = set_intv1(f, q[x]); i = x + 1; p3_step(); }
            break;
          default: st = E_DONE; break;
        }
      }
    }
    if (!__any(req)) break;                                     // every lane is done and the queue is empty
    // ---- one bwt_extend for every lane that asked ------------------------------------------------------------------------
    if (req) {
      nx = extend(f, ik, req_back, req_c);
      req = false;
      // ---- consume the result; staying inside the same loop posts the next request right away ---------------------------
      if (st == A_FWD_RES) {
        bool stop_now = false;
        if (nx.x2 != ik.x2) { L.curr(n_curr++) = ik; stop_now = nx.x2 < (IT)min_intv; }
        if (stop_now) st = A_BACK_INIT;
        else { ik = nx; ik.set((uint32_t)(i + 1), 0); i++; fwd_step(); }
      } else if (st == A_BK_RES) {
        if (nx.x2 < (IT)min_intv) st = A_BK_DONE;
        else { ik = nx; ik.set(ci.lo(), ci.hi() | (uint32_t)kk); L.back(n_back++) = ik; kk--; bk_step(); }
      } else if (st == A_FE_RES) {
        if (nx.x2 < (IT)min_intv) { kk--; st = A_FE_K; }
        else {
          ik = nx;
          if (m == stop) {
            ik.set(ci.lo(), ci.hi() | (uint32_t)(x - kk));
            if ((uint32_t)(x - kk) > temp.hi() && (int)temp.lo() - (int)temp.hi() >= MIN_SEED_LEN) mem.push(temp);
            temp = ik;
            st = A_POST;
          } else { m++; req = true; req_back = false; req_c = 3 - q[m - 1]; }
        }
      } else {   // P3_RES
        if (nx.x2 < 20 && i - x >= MIN_SEED_LEN) {
          I mm = nx; mm.set((uint32_t)(i + 1), (uint32_t)x);
          if (mm.x2 > 0) mem.push(mm);
          x = i + 1; st = P3_NEXT;
        } else { ik = nx; i++; p3_step(); }
      }
    }
  }
}

// One level of the prefix table from the level above it, with the kernel's own bwt_extend (so that an entry is bit for bit what the
// stepwise extension returns, empty intervals included).
__global__ __launch_bounds__(256) void smem_ktab_kernel(SmemArgs a, uint4* tab, int L) {
  const uint32_t code = blockIdx.x * blockDim.x + threadIdx.x;
  if (code >= (1u << (2 * L))) return;
  Ctx<uint32_t> f; f.bwt = a.bwt; f.primary = (uint32_t)a.primary; f.ktab = nullptr;
#pragma unroll
  for (int c = 0; c < 5; c++) f.L2[c] = (uint32_t)a.L2[c];
  Intv<uint32_t> r;
  if (L == 1) r = set_intv1(f, (int)code);
  else {
    const uint4 e = tab[smem_ktab_off(L - 1) + (code >> 2)];
    Intv<uint32_t> p; p.x0 = e.x; p.x1 = e.y; p.x2 = e.z; p.info = 0;
    r = extend(f, p, false, 3 - (int)(code & 3u));
  }
  tab[smem_ktab_off(L) + code] = make_uint4(r.x0, r.x1, r.x2, 0u);
}

}  // namespace

hipError_t ACCG_SMEM_SUFFIX(smem_build_ktab)(const SmemArgs& a, uint4* ktab, hipStream_t s) {
  if (!a.compact) return hipErrorInvalidValue;
  for (int L = 1; L <= SMEM_KTAB_L; L++) {
    const uint32_t n = 1u << (2 * L);
    hipLaunchKernelGGL(smem_ktab_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, ktab, L);
  }
  return hipGetLastError();
}

hipError_t ACCG_SMEM_SUFFIX(smem_launch_engine)(const SmemArgs& a, uint32_t read_base, uint32_t n_reads, uint32_t n_waves, hipStream_t s) {
  if (n_reads == 0) return hipSuccess;
  hipLaunchKernelGGL(smem_engine, dim3(n_waves), dim3(64), 0, s, a, read_base, n_reads);
  return hipGetLastError();
}

#ifndef SMEM_COUNT
namespace {
__global__ __launch_bounds__(256) void smem_merge3_kernel(SmemIntv* out, int32_t* num, const int32_t* num12, uint32_t max_out, const SmemIntv* out3, const int32_t* num3,
                                                          uint32_t max3, uint32_t read_base, uint32_t n_reads) {
  // four lanes per read (a read has one or two third-pass seeds, at most twelve): lane j moves entries j, j + 4, ...
  const uint32_t t = blockIdx.x * 256u + threadIdx.x, j0 = t & 3u;
  if ((t >> 2) >= n_reads) return;
  const uint32_t rd = read_base + (t >> 2);
  const int n = num12[rd], n3 = num3[rd];
  const int kept3 = n3 < (int)max3 ? n3 : (int)max3;
  const uint4* src = reinterpret_cast<const uint4*>(out3 + (size_t)rd * max3);
  uint4* dst = reinterpret_cast<uint4*>(out + (size_t)rd * max_out + n);
  for (int k = (int)j0; k < kept3; k += 4)
    if ((uint32_t)(n + k) < max_out) { const uint4 lo = src[2 * k], hi = src[2 * k + 1]; dst[2 * k] = lo; dst[2 * k + 1] = hi; }
  if (j0 == 0) num[rd] = n + n3;
}
}  // namespace
hipError_t smem_launch_pass3(const SmemArgs& a, uint32_t read_base, uint32_t n_reads, hipStream_t s) {
  if (n_reads == 0) return hipSuccess;
  const dim3 grid((n_reads + 63) / 64), block(64);
  const size_t row = (size_t)64 * a.read_words * sizeof(uint32_t);
  static const int p3_wpc = [] { const char* e = getenv("ACCG_SMEM_PASS3_WPC"); return e && atoi(e) > 0 ? atoi(e) : 12; }();
  const size_t lds = std::max<size_t>(row, std::min<size_t>(64 * 1024, (160 * 1024 / p3_wpc) & ~(size_t)255));      // twelve wavefronts per CU: where the flat passes sit on the sector ceiling
  if (a.compact) hipLaunchKernelGGL(smem_pass3_kernel<uint32_t>, grid, block, lds, s, a, read_base, n_reads);
  else hipLaunchKernelGGL(smem_pass3_kernel<uint64_t>, grid, block, lds, s, a, read_base, n_reads);
  return hipGetLastError();
}
hipError_t smem_launch_merge3(SmemIntv* out, int32_t* num, const int32_t* num12, uint32_t max_out, const SmemIntv* out3, const int32_t* num3, uint32_t max3,
                              uint32_t read_base, uint32_t n_reads, hipStream_t s) {
  if (n_reads == 0) return hipSuccess;
  hipLaunchKernelGGL(smem_merge3_kernel, dim3((uint32_t)(((uint64_t)n_reads * 4 + 255) / 256)), dim3(256), 0, s, out, num, num12, max_out, out3, num3, max3, read_base, n_reads);
  return hipGetLastError();
}
#endif

hipError_t ACCG_SMEM_SUFFIX(smem_launch)(const SmemArgs& a, uint32_t read_base, uint32_t n_reads, hipStream_t s) {
  if (n_reads == 0) return hipSuccess;
  const dim3 grid((n_reads + 63) / 64), block(64);
  const size_t row = (size_t)64 * a.read_words * sizeof(uint32_t);
  auto lds_for = [&](uint32_t waves_per_cu) {      // an LDS request that admits only this many wavefronts per CU (0: no limit)
    // (never more than the 64 KB a launch may ask for without a function attribute: a limit below 3 wavefronts per CU is read as 3)
    return waves_per_cu ? std::max<size_t>(row, std::min<size_t>(64 * 1024, (160 * 1024 / waves_per_cu) & ~(size_t)255)) : row;
  };
  if (a.seg) {         // three kernels (ACCG_SMEM_SPLIT=1)
    const size_t lds_f = lds_for(a.waves_per_cu ? a.waves_per_cu : 12), lds_b = lds_for(a.waves_per_cu);
    if (a.compact) {
      hipLaunchKernelGGL(smem_fwd_kernel<uint32_t>, grid, block, lds_f, s, a, read_base, n_reads);
      hipLaunchKernelGGL(smem_back_kernel<uint32_t>, grid, block, lds_b, s, a, read_base, n_reads);
      hipLaunchKernelGGL(smem_pass3_kernel<uint32_t>, grid, block, lds_f, s, a, read_base, n_reads);
    } else {
      hipLaunchKernelGGL(smem_fwd_kernel<uint64_t>, grid, block, lds_f, s, a, read_base, n_reads);
      hipLaunchKernelGGL(smem_back_kernel<uint64_t>, grid, block, lds_b, s, a, read_base, n_reads);
      hipLaunchKernelGGL(smem_pass3_kernel<uint64_t>, grid, block, lds_f, s, a, read_base, n_reads);
    }
    return hipGetLastError();
  }
  const size_t lds = lds_for(a.waves_per_cu);
  if (a.compact) hipLaunchKernelGGL(smem_kernel<uint32_t>, grid, block, lds, s, a, read_base, n_reads);
  else hipLaunchKernelGGL(smem_kernel<uint64_t>, grid, block, lds, s, a, read_base, n_reads);
  return hipGetLastError();
}

}  // namespace accg

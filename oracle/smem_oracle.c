/* TEST INFRASTRUCTURE ONLY -- see oracle.h.  BWA-MEM SMEM seeding (Falcon's three-pass variant), CPU restatement.
 *
 * PARITY UNPINNED: the reference file this follows, smem/host/baseline.cpp, cannot be compiled here (it
 * includes <bwa/bwa.h> from libbwa, which is not in the tree: smem/Makefile:12,38), and its own test
 * (smem/main.cpp:217-373) needs a real BWA index plus read files that are not in the tree either.  What pins this
 * file instead: Occ against a brute-force count, bidirectional intervals against brute-force occurrence counts of
 * the matched substring on both strands, and a brute-force check that every reported interval is a maximal exact match
 * (tests/test_smem_oracle.py).  Each function cites the reference lines it restates.
 *
 * Index layout (smem/host/baseline.cpp:8,26-37; smem/common/common.h:30-35): the BWT of genome + revcomp(genome)
 * without the sentinel, in blocks of 128 symbols = 4 x uint64 running counts (before the block) + 8 x uint32 of
 * 16 two-bit symbols, first symbol in the top bits.  para = {primary, L2[0..4]}. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define MIN_SEED_LEN 19 /* smem/common/common.h:37 */

typedef struct { const uint32_t* bwt; uint64_t primary, L2[5]; } fmidx;
typedef struct { uint64_t x[3], info; } intv;          /* bwtintv_t */
typedef struct { intv* a; int n, cap; } ivec;

static void push(ivec* v, const intv* e) {
  if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 64; v->a = (intv*)realloc(v->a, sizeof(intv) * (size_t)v->cap); }
  v->a[v->n++] = *e;
}

/* Occ of the four symbols in B[0..k] (bwt_occ4, baseline.cpp:17-38); k == -1 -> zeros. */
static _Thread_local uint64_t t_lookups;   /* 64-byte index blocks requested (the reference's DRAM_trans_size accounting, coarser) */
static void occ4(const fmidx* f, uint64_t k, uint64_t cnt[4]) {
  if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
  k -= (k >= f->primary);                                   /* the sentinel is not stored */
  t_lookups++;
  const uint32_t* blk = f->bwt + ((k >> 7) << 4);
  memcpy(cnt, blk, 32);
  const uint32_t* w = blk + 8;
  const int upto = (int)(k & 127);                          /* symbols 0..upto of the block are counted */
  for (int s = 0; s <= upto; s++) cnt[(w[s >> 4] >> ((~s & 15) << 1)) & 3]++;
}

/* bwt_extend (baseline.cpp:87-100) */
static void extend(const fmidx* f, const intv* ik, intv ok[4], int is_back) {
  uint64_t tk[4], tl[4];
  const int o = !is_back;
  occ4(f, ik->x[o] - 1, tk);
  occ4(f, ik->x[o] - 1 + ik->x[2], tl);
  for (int c = 0; c < 4; c++) { ok[c].x[o] = f->L2[c] + 1 + tk[c]; ok[c].x[2] = tl[c] - tk[c]; }
  ok[3].x[is_back] = ik->x[is_back] + (ik->x[o] <= f->primary && ik->x[o] + ik->x[2] - 1 >= f->primary);
  ok[2].x[is_back] = ok[3].x[is_back] + ok[3].x[2];
  ok[1].x[is_back] = ok[2].x[is_back] + ok[2].x[2];
  ok[0].x[is_back] = ok[1].x[is_back] + ok[1].x[2];
}

static void set_intv1(const fmidx* f, int c, intv* ik) {     /* baseline.h:6 */
  ik->x[0] = f->L2[c] + 1; ik->x[2] = f->L2[c + 1] - f->L2[c]; ik->x[1] = f->L2[3 - c] + 1; ik->info = 0;
}

/* bwt_smem1a_new (baseline.cpp:180-304) with max_intv = 0; appends to mem */
static int smem1a_new(const fmidx* f, int len, const uint8_t* q, int x, int min_intv, ivec* mem, ivec* curr, ivec* back) {
  intv ik, ok[4], temp;
  if (q[x] > 3) return x + 1;
  if (min_intv < 1) min_intv = 1;
  memset(&temp, 0, sizeof temp);
  set_intv1(f, q[x], &ik);
  ik.info = (uint64_t)(x + 1);
  curr->n = 0; back->n = 0;
  int i;
  for (i = x + 1; i < len; i++) {                                                   /* forward, :199-216 */
    if (q[i] < 4) {
      const int c = 3 - q[i];
      extend(f, &ik, ok, 0);
      if (ok[c].x[2] != ik.x[2]) { push(curr, &ik); if (ok[c].x[2] < (uint64_t)min_intv) break; }
      ik = ok[c]; ik.info = (uint64_t)(i + 1);
    } else { push(curr, &ik); break; }
  }
  if (i == len) push(curr, &ik);
  const int ret = (int)curr->a[curr->n - 1].info;
  int start = x, stop = x, max_len = 0;
  i = 0;
  while (i < curr->n) {                                                             /* :220-299 */
    ik = curr->a[i];
    ik.info |= (uint64_t)x << 32;
    if (back->n == 0 || stop - start >= 3) {                                        /* "backenlarge" */
      back->n = 0;
      push(back, &ik);
      for (int k = x - 1; k >= -1; k--) {
        const int c = k < 0 ? -1 : q[k] < 4 ? q[k] : -1;
        if (c < 0) break;
        extend(f, &ik, ok, 1);
        if (ok[c].x[2] < (uint64_t)min_intv) break;
        ik = ok[c];
        ik.info = curr->a[i].info | (uint64_t)k << 32;
        push(back, &ik);
      }
      start = (int)curr->a[i].info;
      stop = (i == curr->n - 1) ? len : (int)curr->a[i + 1].info;
      if (i != 0 && (ik.info >> 32) > (temp.info >> 32) && (int)temp.info - (int)(temp.info >> 32) >= MIN_SEED_LEN) push(mem, &temp);
      temp = ik;
    } else {                                                                        /* "forwardenlarge" */
      stop = (int)curr->a[i].info;
      for (int k = back->n - 1; k >= 0; k--) {
        ik = back->a[k];
        int reached = 0;
        for (int m = start + 1; m <= stop; m++) {
          const int c = 3 - q[m - 1];
          extend(f, &ik, ok, 0);
          if (ok[c].x[2] < (uint64_t)min_intv) break;
          ik = ok[c];
          if (m == stop) { ik.info = curr->a[i].info | (uint64_t)(x - k) << 32; reached = 1; }
        }
        if (reached) {
          if ((uint64_t)(x - k) > (temp.info >> 32) && (int)temp.info - (int)(temp.info >> 32) >= MIN_SEED_LEN) push(mem, &temp);
          temp = ik;
          break;
        }
      }
    }
    i++;
    if (i < curr->n) max_len = (int)(temp.info >> 32) + (int)curr->a[i].info;
    while (max_len < MIN_SEED_LEN && i < curr->n) {
      i++;
      if (i < curr->n) stop = (int)curr->a[i].info;
      max_len = (int)(temp.info >> 32) + stop;
    }
    if (i >= curr->n && (int)temp.info - (int)(temp.info >> 32) >= MIN_SEED_LEN) push(mem, &temp);
  }
  return ret;
}

/* bwt_seed_strategy1 (baseline.cpp:306-327) */
static int seed_strategy1(const fmidx* f, int len, const uint8_t* q, int x, int min_len, int max_intv, intv* mem) {
  intv ik, ok[4];
  memset(mem, 0, sizeof *mem);
  if (q[x] > 3) return x + 1;
  set_intv1(f, q[x], &ik);
  for (int i = x + 1; i < len; i++) {
    if (q[i] >= 4) return i + 1;
    const int c = 3 - q[i];
    extend(f, &ik, ok, 0);
    if (ok[c].x[2] < (uint64_t)max_intv && i - x >= min_len) { *mem = ok[c]; mem->info = (uint64_t)x << 32 | (uint64_t)(i + 1); return i + 1; }
    ik = ok[c];
  }
  return len;
}

/* mem_collect_intv_new (baseline.cpp:387-422); *n_pass1 (nullable) receives the number of first-pass entries */
static void collect_counted(const fmidx* f, int len, const uint8_t* seq, ivec* mem, ivec* curr, ivec* back, int* n_pass1, int* n_pass2) {
  mem->n = 0;
  for (int x = 0; x < len;) x = seq[x] < 4 ? smem1a_new(f, len, seq, x, 1, mem, curr, back) : x + 1;
  const int old_n = mem->n;
  if (n_pass1) *n_pass1 = old_n;
  for (int k = 0; k < old_n; k++) {
    const int start = (int)(mem->a[k].info >> 32), end = (int)(int32_t)mem->a[k].info;
    const uint64_t occ = mem->a[k].x[2];
    if (end - start < 28 || occ > 10) continue;                                     /* split_len, split_width */
    smem1a_new(f, len, seq, (start + end) >> 1, (int)occ + 1, mem, curr, back);
  }
  if (n_pass2) *n_pass2 = mem->n;
  for (int x = 0; x < len;) {
    if (seq[x] < 4) { intv m; x = seed_strategy1(f, len, seq, x, MIN_SEED_LEN, 20, &m); if (m.x[2] > 0) push(mem, &m); }
    else x++;
  }
}

static void collect(const fmidx* f, int len, const uint8_t* seq, ivec* mem, ivec* curr, ivec* back) {
  collect_counted(f, len, seq, mem, curr, back, NULL, NULL);
}

/* One read, with the boundaries between the three passes (for the from-the-definition tests): out gets all entries, bounds =
 * {entries after pass 1, after pass 2, after pass 3}.  Returns the total. */
int orc_smem_read_passes(const uint32_t* bwt, const uint64_t* para, const uint8_t* seq, int len, int max_out, uint64_t* out, int bounds[3]) {
  fmidx f; f.bwt = bwt; f.primary = para[0];
  for (int c = 0; c < 5; c++) f.L2[c] = para[1 + c];
  ivec mem = {0, 0, 0}, curr = {0, 0, 0}, back = {0, 0, 0};
  collect_counted(&f, len, seq, &mem, &curr, &back, &bounds[0], &bounds[1]);
  bounds[2] = mem.n;
  const int keep = mem.n < max_out ? mem.n : max_out;
  memcpy(out, mem.a, sizeof(intv) * (size_t)keep);
  const int n = mem.n;
  free(mem.a); free(curr.a); free(back.a);
  return n;
}

/* smem_baseline (baseline.cpp:425-463): seq is batch x seq_stride codes (0-3, >= 4 ambiguous), out is batch x max_out
 * intervals of 4 uint64 {x0, x1, x2, info}; mem_num[i] is the uncapped count. */
static uint64_t g_lookups;
uint64_t orc_smem_last_lookups(void) { return g_lookups; }

void orc_smem_batch(const uint32_t* bwt, const uint64_t* para, const uint8_t* seq, int seq_stride, const uint8_t* seq_len,
                    int batch, int max_out, uint64_t* out, int* mem_num, int n_threads) {
  uint64_t total = 0;
  fmidx f; f.bwt = bwt; f.primary = para[0];
  for (int c = 0; c < 5; c++) f.L2[c] = para[1 + c];
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads) reduction(+ : total)
  {
    ivec mem = {0, 0, 0}, curr = {0, 0, 0}, back = {0, 0, 0};
    t_lookups = 0;
#pragma omp for schedule(dynamic, 64)
    for (int i = 0; i < batch; i++) {
      collect(&f, seq_len[i], seq + (size_t)i * seq_stride, &mem, &curr, &back);
      mem_num[i] = mem.n;
      const int keep = mem.n < max_out ? mem.n : max_out;
      memcpy(out + (size_t)i * max_out * 4, mem.a, sizeof(intv) * (size_t)keep);
    }
    free(mem.a); free(curr.a); free(back.a);
    total += t_lookups;
  }
  g_lookups = total;
}

/* Occ and one extension step exposed for the brute-force tests */
void orc_smem_occ4(const uint32_t* bwt, const uint64_t* para, uint64_t k, uint64_t cnt[4]) {
  fmidx f; f.bwt = bwt; f.primary = para[0];
  for (int c = 0; c < 5; c++) f.L2[c] = para[1 + c];
  occ4(&f, k, cnt);
}

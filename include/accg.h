/* acc_genomics_amd -- C ABI of the MI355X (gfx950) hot path: PairHMM forward + HTC Smith-Waterman.
 *
 * Every entry point is plain C (pointers + sizes, int status, nothing thrown across the boundary,
 * caller owns every buffer it passes).  Each one names the reference interface it stands in for;
 * INTEGRATION.md shows the C++ wrappers a maintainer of the reference would add on top.
 *
 * There is no CPU fallback behind this ABI: if no gfx950 device is present, accg_init() fails with
 * ACCG_ERR_NO_DEVICE and every other call with ACCG_ERR_NOT_INITIALISED.
 *
 * A context owns its HIP streams, a cache of device blocks and a pinned staging buffer: use it from one host thread at a
 * time (one context per thread for concurrent callers) and destroy its batches before accg_shutdown().
 */
#ifndef ACCG_H
#define ACCG_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
  ACCG_OK = 0,
  ACCG_ERR_NO_DEVICE = -1,       /* no HIP device / not gfx950 */
  ACCG_ERR_NOT_INITIALISED = -2,
  ACCG_ERR_BAD_ARG = -3,
  ACCG_ERR_BAD_WIRE = -4,        /* serialized reads/haps blob inconsistent with its size */
  ACCG_ERR_BAD_BASE = -5,        /* base other than A,C,G,T,N (SURVEY.md appendix B caveat 1) */
  ACCG_ERR_EMPTY_SEQ = -6,       /* zero-length read or haplotype (reference divides by haplen) */
  ACCG_ERR_TOO_LONG = -7,        /* read > ACCG_PHMM_MAX_READ or hap > ACCG_PHMM_MAX_HAP; SW length > ACCG_SW_MAX_LEN */
  ACCG_ERR_HIP = -8,             /* a HIP runtime call failed; see accg_last_hip_error() */
  ACCG_ERR_NOMEM = -9,
  ACCG_ERR_RCCL = -10,           /* an RCCL call failed; see accg_last_hip_error() */
  ACCG_ERR_NO_RCCL = -11         /* a communicator over more than one rank was asked for and librccl cannot be loaded */
};

#define ACCG_PHMM_MAX_READ 16383 /* up to 1023 bases a read's rows are held in registers (64 lanes x 16 rows, one row reserved); longer reads
                                    are swept in stripes of 1024 rows, as the reference's CPU code does with its own stripe height
                                    (pairhmm/xlnx/host/avx-pairhmm-template.h:224,265-297) */
#define ACCG_PHMM_MAX_HAP 4000   /* one haplotype must fit the per-wave LDS stream */
#define ACCG_SW_MAX_LEN 1535     /* htc-sw/host/common.h:13 MAX_SEQ_LENGTH - 1 */

/* arithmetic mode of the fp32 PairHMM kernel */
#define ACCG_PHMM_FAST 0    /* FMA-contracted recurrence (default) */
#define ACCG_PHMM_STRICT 1  /* operation order of compute_full_prob_baseline<float>, no contraction: bit-exact with it */

typedef struct accg_ctx accg_ctx;
typedef struct accg_phmm_batch accg_phmm_batch;
typedef struct accg_sw_batch accg_sw_batch;

typedef struct {
  uint64_t cells;        /* sum of read_len * hap_len over all pairs (the reference's GCUPS numerator,
                            pairhmm/xlnx/pairhmm_test.cpp:280-289) */
  uint64_t pairs;
  uint64_t kernel_ns;    /* device time of the last run, fp32 pass + fp64 rescue pass (hipEvents) */
  uint64_t rescued;      /* pairs recomputed in fp64 (raw < 1e-28f, FalconPairHMM.cpp:84) */
} accg_counters;

/* ---- context -------------------------------------------------------------------------------
 * Side effect on the calling thread: the host loops of this library are OpenMP loops, and accg_init (and every ring worker) asks libomp
 * to let idle team threads sleep after 1 ms instead of spinning for 200 (kmp_set_blocktime(1): spinners eat a container's CPU quota).
 * That setting is per thread and also governs the caller's own OpenMP regions on that thread.  KMP_BLOCKTIME in the environment wins;
 * ACCG_KEEP_OMP_BLOCKTIME=1 leaves the application's setting alone; with an OpenMP runtime other than libomp nothing is changed. */
int accg_init(int device, accg_ctx** out);
void accg_shutdown(accg_ctx* ctx);
const char* accg_strerror(int status);
const char* accg_last_hip_error(void);
/* the stream every launch of this context goes to (a hipStream_t) */
void* accg_stream(accg_ctx* ctx);
/* waits until everything queued on that stream has finished (the fence around a timed region) */
int accg_ctx_synchronize(accg_ctx* ctx);
int accg_device_name(accg_ctx* ctx, char* buf, size_t n);
/* gives the context's cached device blocks and pinned staging back to the driver (the FPGA host's _release_smithWaterman,
 * htc-sw/host/smithWatermanHost.cpp:306-320); the context stays usable */
int accg_ctx_trim(accg_ctx* ctx);

/* ---- PairHMM --------------------------------------------------------------------------------
 * One "region" = all reads x all haplotypes, given in the reference's wire format
 * (pairhmm/interface/PairHMMHostInterface.cpp:175-206: int32 num; per read int32 len + _b,_q,_i,_d,_c;
 * per hap int32 len + bases).  Output index = read * n_haps + hap, as compute_fpga returns it
 * (pairhmm/host/PairHMMFpga.cpp:125-162) and FalconPairHMM::computePairhmmAVX fills it
 * (pairhmm/xlnx/host/FalconPairHMM.cpp:69-95). */

/* Replaces compute_fpga() (pairhmm/host/PairHMMFpga.h:16-20) and the prepare()+compute() pair of the
 * Blaze task (pairhmm/task/xlnx/PairHMMTask.cpp:27-143): raw fp32 likelihood x 2^120 per pair.
 * out_log10 (nullable) additionally receives the final log10 likelihood with the fp64 rescue applied,
 * i.e. the output of FalconPairHMM::computePairhmm / PairHMMWorker::getOutput
 * (pairhmm/client/PairHMMWorker.cpp:157-197). */
int accg_phmm_region(accg_ctx* ctx, const void* reads_ser, size_t reads_bytes, const void* haps_ser,
                     size_t haps_bytes, int mode, float* out_raw, double* out_log10, accg_counters* counters);

/* Regions in flight.  accg_phmm_region is a latency chain (the caller waits); a caller that owns the loop over active regions keeps up
 * to `slots` of them in flight instead: accg_phmm_ring_submit does the host half of a region (parse, job sizing, staging) and queues
 * its upload, kernels and downloads on a stream of the slot's own without waiting; accg_phmm_ring_wait returns the results of a
 * submitted region -- while region i computes, region i + 1 is parsed and region i - 1 read back.  Tickets count up from 0; at most
 * `slots` may be outstanding (submit fails with ACCG_ERR_BAD_ARG when the slot of ticket - slots has not been waited for).  Same
 * results, bit for bit, as accg_phmm_region.  The blobs may be reused as soon as submit returns.  One thread at a time per ring. */
typedef struct accg_phmm_ring accg_phmm_ring;
int accg_phmm_ring_create(accg_ctx* ctx, int slots, accg_phmm_ring** out);
/* The same ring with a worker thread per slot: submit only checks the blobs' headers, hands them to the slot's worker and returns, so
 * the host halves of up to `slots` tickets run side by side (each on its share of the host threads).  THE BLOBS (and nothing else: the
 * pointer and size arrays are copied) MUST STAY UNTOUCHED UNTIL THE TICKET HAS BEEN WAITED FOR; errors of the host half come back from
 * accg_phmm_ring_wait.  Still one caller thread at a time per ring. */
int accg_phmm_ring_create_threaded(accg_ctx* ctx, int slots, accg_phmm_ring** out);
/* THREADING CONTRACT of both rings: submit, wait and destroy of one ring are called by ONE thread at a time (the ticket counter and the
 * slot bookkeeping are the caller's side of the ring; the worker threads of a threaded ring are the library's own).  Callers on several
 * threads take a ring each -- or accg_phmm_mux_region below, which is made for that.  A ticket's counters carry the device time of its
 * pass in kernel_ns, like accg_phmm_region's. */
int accg_phmm_ring_submit(accg_phmm_ring* ring, const void* reads_ser, size_t reads_bytes, const void* haps_ser, size_t haps_bytes,
                          int mode, uint64_t* ticket);
/* several regions under one ticket (one device batch: its results come back concatenated in region order, like accg_phmm_batch_results) */
int accg_phmm_ring_submit_many(accg_phmm_ring* ring, int n_regions, const void* const* reads_ser, const size_t* reads_bytes,
                               const void* const* haps_ser, const size_t* haps_bytes, int mode, uint64_t* ticket);
int accg_phmm_ring_wait(accg_phmm_ring* ring, uint64_t ticket, float* out_raw, double* out_log10, accg_counters* counters);
void accg_phmm_ring_destroy(accg_phmm_ring* ring);

/* Regions from CONCURRENT blocking callers -- the reference's own calling pattern: compute_fpga / FalconPairHMM::computePairhmm hand over
 * one region per blocking call (pairhmm/host/PairHMMFpga.cpp:125-162, pairhmm/xlnx/host/FalconPairHMM.cpp:1184-1193) and an accelerator
 * manager runs one PairHMM task per request, several at a time (pairhmm/task/xlnx/PairHMMTask.cpp:27-143).  accg_phmm_mux_region has the
 * arguments, results and errors of accg_phmm_region (bit for bit) and may be called from any number of threads at once: a caller
 * parses its own region and either leads one device batch of everything queued at that moment, on one of the mux's `lanes` contexts
 * (2 or 3: the host half of one batch behind the device half of another), or is taken along by a leader.  A lone caller runs at once on
 * its own thread.  max_regions caps a batch.  The blobs must stay untouched until the call returns (as for any blocking call). */
typedef struct accg_phmm_mux accg_phmm_mux;
int accg_phmm_mux_create(int device, int lanes, int max_regions, accg_phmm_mux** out);
int accg_phmm_mux_region(accg_phmm_mux* mux, const void* reads_ser, size_t reads_bytes, const void* haps_ser, size_t haps_bytes, int mode,
                         float* out_raw, double* out_log10, accg_counters* counters);
/* device batches run and regions served so far (regions / batches = how many callers a leader took along on average) */
void accg_phmm_mux_stats(accg_phmm_mux* mux, uint64_t* batches, uint64_t* regions);
void accg_phmm_mux_destroy(accg_phmm_mux* mux);      /* no call may be in progress */

/* The same region entirely in fp64 (compute_fp_avxd, avx_impl.h:6; use_double, FalconPairHMM.cpp:82): raw x 2^1020. */
int accg_phmm_region_f64(accg_ctx* ctx, const void* reads_ser, size_t reads_bytes, const void* haps_ser, size_t haps_bytes,
                         double* out_raw64);
int accg_phmm_batch_run_f64(accg_phmm_batch* b);
int accg_phmm_batch_results_f64(accg_phmm_batch* b, double* out_raw64);
/* fp32 pass only / whole run timing with hipEvents on the launch stream; what: 0 = fp32 + rescue, 1 = fp32 pass only */
int accg_phmm_batch_time2(accg_phmm_batch* b, int mode, int what, int warmup, int iters, float* ms_per_run);
/* `iters` whole passes back to back with HIP events around each pass's fp32 sweep launches, on the stream they are launched on:
 * kernel_ms = mean of the dominant kernel measured INSIDE the step, step_ms = mean whole pass (same run, same clock state) */
int accg_phmm_batch_time_in_step(accg_phmm_batch* b, int mode, int iters, float* kernel_ms, float* step_ms);
/* The same in three parts, for a caller that brackets the passes with a clock of its own and wants nothing but the passes inside the
 * bracket: _reserve makes the events (before the bracket), _run queues `iters` passes with their events and returns without waiting,
 * _times reads them (after the caller's own accg_ctx_synchronize). */
int accg_phmm_batch_steps_reserve(accg_phmm_batch* b, int iters);
int accg_phmm_batch_steps_run(accg_phmm_batch* b, int mode, int iters);
int accg_phmm_batch_steps_times(accg_phmm_batch* b, float* kernel_ms, float* step_ms);
/* The per-row coefficient records of the fast sweep are written ONCE, at batch creation (a pure function of the reads, like the haplotype
 * streams): a pass over a device-resident batch does not rewrite them.  This times the kernel that writes them, mean of `iters` launches. */
int accg_phmm_batch_time_prepare(accg_phmm_batch* b, int iters, float* ms_per_run);
/* the shader clock the device holds under load right now, in GHz: a ~0.3 ms full-chip fp32 kernel whose first wavefront reads the
 * shader-clock counter and the constant-rate wall clock at both ends */
int accg_ctx_clock_ghz(accg_ctx* ctx, float* ghz);
/* ... and the shader clock held while the first wavefront of the batch's last sweep launch ran (its own shader-clock ticks over its
 * own 100 MHz wall-clock ticks): the clock under the PairHMM kernel itself */
int accg_phmm_batch_clock_ghz(accg_phmm_batch* b, float* ghz);
uint64_t accg_phmm_batch_jobs(const accg_phmm_batch* b);
/* Context<float>/<double> tables as uploaded (ph2pr[128], matchToMatchProb[8256], INITIAL_CONSTANT, its log10) */
void accg_phmm_tables_f32(float* ph2pr, float* m2m, float* init, float* log10_init);
void accg_phmm_tables_f64(double* ph2pr, double* m2m, double* init, double* log10_init);

/* Device-resident multi-region batch: upload once, run many times (pipelines, bench.py).
 * Regions are independent; outputs are concatenated in region order, each region row-major. */
int accg_phmm_batch_create(accg_ctx* ctx, int n_regions, const void* const* reads_ser, const size_t* reads_bytes,
                           const void* const* haps_ser, const size_t* haps_bytes, accg_phmm_batch** out);
uint64_t accg_phmm_batch_pairs(const accg_phmm_batch* b);
uint64_t accg_phmm_batch_cells(const accg_phmm_batch* b);
/* bytes the kernels must move for this batch: wire blobs in + 4 B per pair out */
uint64_t accg_phmm_batch_algorithmic_bytes(const accg_phmm_batch* b);
/* fp32 pass over every pair, then the fp64 rescue pass; asynchronous on accg_stream(). */
int accg_phmm_batch_run(accg_phmm_batch* b, int mode);
/* `iters` back-to-back runs bracketed by hipEvents on the launch stream; returns mean ms per run. */
int accg_phmm_batch_time(accg_phmm_batch* b, int mode, int warmup, int iters, float* ms_per_run);
/* waits for the stream, copies results back; either pointer may be NULL */
int accg_phmm_batch_results(accg_phmm_batch* b, float* out_raw, double* out_log10, accg_counters* counters);
void accg_phmm_batch_destroy(accg_phmm_batch* b);

/* ---- HTC Smith-Waterman ------------------------------------------------------------------------
 * GATK SWPairwiseAlignment as the reference implements it on the CPU
 * (htc-sw/host/FalconSW_AVX.cpp: fill :1693-1823, end cell :2314-2339; weights htc-sw/host/common.h:19-22).
 * A batch is n independent (reference window, alternate/read) pairs given as two strided byte matrices;
 * row k of `refs` is the window of pair k (ref_lens[k] bytes used), likewise `alts`.  One ref x B alts,
 * the shape of SWPairwiseAlignmentMultiBatch (:304) / FalconSWFPGA_run (htc-sw/host/FalconSW_FPGA.cpp:28),
 * is the special case ref_stride = 0.
 * Results per pair: score = sw[p1][p2] and the end cell (p1, p2) that calculateCigarOneBatch starts its
 * backtrace from -- bit-exact with the CPU path; accg_sw_batch_run_cigar below adds the backtrace (CIGAR, alignment_offset).
 * strategies: per pair, 0 SOFTCLIP, 1 INDEL, 2 LEADING_INDEL, 3 IGNORE (common.h:15-18); NULL = all SOFTCLIP.
 * Limits: 1 <= length <= ACCG_SW_MAX_LEN (the reference's MAX_SEQ_LENGTH - 1) for both sequences.  The shorter one is
 * spread over 16 lanes (<= 255), 32 lanes (<= 511) or a whole wavefront (<= 1535). */
int accg_sw_batch_create(accg_ctx* ctx, int n_pairs, const uint8_t* refs, size_t ref_stride, const int32_t* ref_lens,
                         const uint8_t* alts, size_t alt_stride, const int32_t* alt_lens, const uint8_t* strategies,
                         int w_match, int w_mismatch, int w_open, int w_extend, accg_sw_batch** out);
uint64_t accg_sw_batch_cells(const accg_sw_batch* b);               /* sum ref_len * alt_len (sw_host.cpp:314) */
uint64_t accg_sw_batch_algorithmic_bytes(const accg_sw_batch* b);   /* ref_len + alt_len + 16 per pair */
int accg_sw_batch_run(accg_sw_batch* b);                            /* asynchronous on accg_stream() */
int accg_sw_batch_time(accg_sw_batch* b, int warmup, int iters, float* ms_per_run);
int accg_sw_batch_results(accg_sw_batch* b, int32_t* score, int32_t* p1, int32_t* p2);
/* Fill + end cell + backtrace: the whole of SWPairwiseAlignmentOneBatch (FalconSW_AVX.cpp:315-411).  Per pair at
 * most max_el CIGAR elements {length, state} in final (forward) order, states as htc-sw/host/common.h:23-26
 * (M 0, I 1, D 2, S 4), and alignment_offset (:2379-2401).  accg_sw_batch_results is valid afterwards too. */
int accg_sw_batch_run_cigar(accg_sw_batch* b, int max_el);
/* n_el[k] > 0: element count; -1: calculateCigarOneBatch's "no element" failure (:2404-2407);
 * < -1: -(elements needed) > max_el, rerun with a larger max_el.  elements: int32[n][max_el][2]. */
int accg_sw_batch_cigars(accg_sw_batch* b, int32_t* n_el, int32_t* alignment_offsets, int32_t* elements);
/* The same results as the device holds them: all CIGARs back to back, pair k's n_el[k] elements at elements[2 * starts[k]].
 * *total = elements in all; elements (capacity in elements) may be NULL to query it.  Any pointer may be NULL. */
int accg_sw_batch_cigars_packed(accg_sw_batch* b, int32_t* n_el, int32_t* alignment_offsets, uint64_t* starts, int32_t* elements,
                                uint64_t capacity, uint64_t* total);
/* Zero-copy form of the same: pointers into the context's pinned host staging block, filled by two device-to-host copies and
 * valid until the next call on this context that returns results.  Any pointer may be NULL. */
int accg_sw_batch_cigars_packed_view(accg_sw_batch* b, const int32_t** n_el, const int32_t** alignment_offsets, const uint64_t** starts,
                                     const int32_t** elements, uint64_t* total);
void accg_sw_batch_destroy(accg_sw_batch* b);

/* ---- SMEM seeding (BWA-MEM, configs[4]) -----------------------------------------------------------------
 * mem_collect_intv_new of the reference (smem/host/baseline.cpp:387-422) for batches of reads against an FM-index in
 * BWA's block layout.  accg_smem_index_create plays the role of ocl_init (smem/host/ocl.h:29): the index is uploaded
 * once; bwt_para = {primary, L2[0..4], ...} as smem/main.cpp:221 builds it.  A batch is smem_ocl's input
 * (smem/host/ocl.h:30-32): seq = n x seq_stride base codes (0-3, >= 4 ambiguous), seq_len uint8 per read; outputs
 * mem_output = n x max_out intervals {x[0], x[1], x[2], info} (bwtintv_t, 32 B) in the order the CPU code produces them
 * and mem_num = the uncapped count (a count > max_out means "redo on the CPU", smem/main.cpp:159-164).
 * PARITY of this path is unpinned: the reference file needs libbwa and cannot be built here. */
typedef struct accg_smem_index accg_smem_index;
typedef struct accg_smem_batch accg_smem_batch;
/* bwt_words = the caller's true uint32 count (a BWA index is generally not a whole number of 16-word blocks: the tail is
 * padded inside; nothing beyond bwt[bwt_words - 1] is read).  What the device holds is the library's business: below 2^32
 * symbols the blocks are re-laid out (32-byte half-blocks of 32-bit counts + two bit planes) and a 22 MB table of the
 * intervals of all strings of up to ten bases is built next to them; the intervals a batch returns are the same numbers. */
int accg_smem_index_create(accg_ctx* ctx, const uint32_t* bwt, uint64_t bwt_words, const uint64_t* bwt_para, accg_smem_index** out);
/* Index construction for a synthetic genome (the reference loads an existing one with libbwa's bwa_idx_load,
 * smem/main.cpp:434; SURVEY.md 8f row 1 asks for a constructor of our own): text = genome ++ reverse complement, suffix
 * array by prefix doubling on the device, BWT and block layout as smem/host/baseline.cpp:26-37 reads it.  genome_codes:
 * n_genome bytes over {0,1,2,3}; bwt_out: accg_smem_index_words(n_genome) uint32; bwt_para[7] = {primary, L2[0..4], blocks}. */
uint64_t accg_smem_index_words(uint64_t n_genome);
int accg_smem_index_build(accg_ctx* ctx, const uint8_t* genome_codes, uint64_t n_genome, uint32_t* bwt_out, uint64_t bwt_words_cap,
                          uint64_t* bwt_para);
void accg_smem_index_destroy(accg_smem_index* idx);
int accg_smem_batch_create(accg_smem_index* idx, const uint8_t* seq, uint32_t seq_stride, const uint8_t* seq_len, uint32_t n_reads,
                           uint32_t max_out, accg_smem_batch** out);
uint64_t accg_smem_batch_bases(const accg_smem_batch* b);
int accg_smem_batch_run(accg_smem_batch* b);
int accg_smem_batch_time(accg_smem_batch* b, int warmup, int iters, float* ms_per_run);
int accg_smem_batch_results(accg_smem_batch* b, void* mem_output, int32_t* mem_num);
void accg_smem_batch_destroy(accg_smem_batch* b);
/* Measurement aid: with ACCG_SMEM_COUNT=1 in the environment the batches run a counting build of the same kernels; this returns and
 * resets {32-byte index sectors fetched, prefix-table entries fetched, bwt_extend calls, 0} (zeros otherwise).  The reference accounts
 * the blocks it REQUESTS (smem/host/baseline.cpp:28-75); this is what the device fetched. */
int accg_smem_debug_counts(accg_ctx* ctx, uint64_t out[4]);

/* ---- BWA-MEM seed extension (bwa-sw) ---------------------------------------------------------------------
 * seed_proc + sw_extend of the reference's FPGA kernel (bwa-sw/sdaccel/smithwaterman.cpp:511-672, :75-273): for each seed
 * a banded left extension, then a right extension seeded with the left score (match 1, mismatch -4, N -1, gap 6+1,
 * pen_clip 5, w 100, up to two band tries).  Inputs are the two streams the FPGA host feeds (:564-584): per seed
 * params = {leftQlen, leftRlen, rightQlen, rightRlen, seed_len, seed_qbeg, seed_index} (uint16 x 7) and the codes (0-3,
 * >= 4 = N) [left query][right query][left target][right target] at seqs + seq_off[i].  Results: fields = int16[n][7]
 * {qBeg, qEnd, rBeg, rEnd, score, trueScore, width} and/or words = int32[n][5], the match stream of :666-670.
 * Limits are the device code's own (uint8_t qlen, uint11_t tlen, seq_mem[2048]).  PARITY of this path is unpinned (see
 * DESIGN.md): the reference is HLS device code and its host needs libbwa; neither builds here. */
#define ACCG_BWASW_MAX_QLEN 254
#define ACCG_BWASW_MAX_TLEN 2047
typedef struct accg_bwasw_batch accg_bwasw_batch;
int accg_bwasw_batch_create(accg_ctx* ctx, uint32_t n_seeds, const uint8_t* seqs, const uint32_t* seq_off, const uint16_t* params,
                            accg_bwasw_batch** out);
uint64_t accg_bwasw_batch_cells(const accg_bwasw_batch* b);
int accg_bwasw_batch_run(accg_bwasw_batch* b);
int accg_bwasw_batch_time(accg_bwasw_batch* b, int warmup, int iters, float* ms_per_run);
int accg_bwasw_batch_results(accg_bwasw_batch* b, int16_t* fields, int32_t* words);
void accg_bwasw_batch_destroy(accg_bwasw_batch* b);
/* The FPGA kernel's own buffers, sw_top(input, output, pac_input, size) (smithwaterman.cpp:1046-1054): `input` is the host's
 * int stream of reads / chains / seeds (data_parse :311-458, written by the bwa-flow host, re-read in main_cl.cpp:73-90),
 * `pac` the 2-bit packed reference (16 bases per int).  results = five ints per seed (:666-670) in input order (the FPGA
 * emits them in completion order, seed_index identifies them).  results == NULL only counts the tasks into *n_tasks. */
int accg_bwasw_records(accg_ctx* ctx, const int32_t* input, int64_t size, const uint32_t* pac, uint64_t pac_words,
                       int32_t* results, int64_t results_cap, int64_t* n_tasks);

/* ---- counters (multi-GPU) ---------------------------------------------------------------------
 * One process per GPU; a batch is cut into contiguous shards of regions (PairHMM) or pairs (SW) in proportion to cell
 * counts -- the rule the reference applies across its compute dies (pairhmm/xlnx/host/FalconPairHMM.cpp:169-249,
 * cell-proportional split :187-197) -- and every rank uploads, computes and reads back its own shard.  There is no
 * data-path collective.  What the reference adds up over its dies at the end (kernel time and cells for the GCUPS it
 * prints, FalconPairHMM.cpp:1214-1220) is here one RCCL all-reduce over xGMI of uint64[4] {cells, pairs, kernel_ns,
 * rescued} (sum) plus the wall time (max).  librccl is opened at run time (dlopen); a world of one needs none.
 *
 * Bring-up: rank 0 calls accg_comm_unique_id and hands the ACCG_COMM_ID_BYTES to the other ranks out of band (a file,
 * a socket, the launcher's environment); every rank then calls accg_comm_init on its own context (collective). */
#define ACCG_COMM_ID_BYTES 128
typedef struct accg_comm accg_comm;
void accg_counters_pack(const accg_counters* c, uint64_t out[4]);
/* ACCG_OK when librccl can be loaded (ACCG_RCCL_LIB names it, else the system's librccl.so), ACCG_ERR_NO_RCCL otherwise; needs no
 * device.  Every rank calls it and the ranks compare notes BEFORE any of them enters accg_comm_init, which is collective. */
int accg_comm_available(void);
int accg_comm_unique_id(void* id_bytes);                       /* ncclGetUniqueId */
int accg_comm_init(accg_ctx* ctx, int rank, int world, const void* id_bytes, accg_comm** out);   /* id may be NULL for world == 1 */
int accg_comm_rank(const accg_comm* c);
int accg_comm_world(const accg_comm* c);
int accg_comm_uses_rccl(const accg_comm* c);                   /* 0 for a world of one (no collective is issued) */
/* sum of the counters and max of wall_s over all ranks, same result on every rank; collective, waits for the context's
 * stream.  total / wall_max may be NULL. */
int accg_counters_allreduce(accg_comm* c, const accg_counters* mine, double wall_s, accg_counters* total, double* wall_max);
/* this rank's stream has drained and every rank has arrived */
int accg_comm_barrier(accg_comm* c);
void accg_comm_destroy(accg_comm* c);

#ifdef __cplusplus
}
#endif
#endif

// Host-side mirror of the reference's own entry points for the hot path, implemented on the C ABI
// (include/accg.h).  Names, argument meaning, ownership and error behaviour follow the reference so
// that its callers and its test drivers compile against this header unchanged:
//   read_t / hap_t, serialize / deserialize, free_reads / free_haps   pairhmm/interface/PairHMMHostInterface.h:27-83
//   compute_fpga, curr/peak_kernel_gcups, cleanup                      pairhmm/host/PairHMMFpga.h:13-22
//   Read / Hap / pairhmmInput / pairhmmOutput, FalconPairHMM           pairhmm/xlnx/host/host_type.h:100-119, FalconPairHMM.h:17-27
//   struct Cigar / CigarElement, weights, strategies                   htc-sw/host/common.h:13-57
//   FalconSWFPGA_init / _run / _release                                htc-sw/host/FalconSW_FPGA.cpp:16,28,92
//   _smithWatermanRun (byte contract of the FPGA kernel)               htc-sw/host/smithWatermanHost.h:14, FalconSW_FPGA.cpp:53-88
//   _init_opencl / _init_kernel_buffer / _release_smithWaterman        htc-sw/host/smithWatermanHost.h:12,13,15
//   SWPairwiseAlignmentMultiBatch                                      htc-sw/host/FalconSW_AVX.cpp:304
//   ocl_init / smem_ocl                                                smem/host/ocl.h:29-32
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <vector>

// ---- PairHMM wire types ---------------------------------------------------------------------------
typedef struct { int len; char* _b; char* _q; char* _i; char* _d; char* _c; } read_t;
typedef struct { int len; char* _b; } hap_t;
void free_reads(read_t* r, int n);
void free_haps(hap_t* h, int n);
uint64_t serialize(void* buf, const read_t* reads, int num);
uint64_t serialize(void* buf, const hap_t* haps, int num);
int deserialize(const void* buf, read_t*& reads);
int deserialize(const void* buf, hap_t*& haps);
std::string serialize(const read_t* reads, int num);
std::string serialize(const hap_t* haps, int num);
int deserialize(const std::string& data, read_t*& reads);
int deserialize(const std::string& data, hap_t*& haps);

// host_tb text format (pairhmm/host/main.cpp:67-159): `input<i>` = header "<label> numRead <label> numHap", per read a
// length line followed by five (label line, integer line) pairs for _b (ASCII codes), _q, _i, _d, _c, an empty line,
// per hap a length line, a label line and the literal bases; `output<i>` = per pair "<decimal> <int64 bit pattern>".
// Arrays are malloc'ed as the reference does (free_reads / free_haps; get_input's hap bases come from strdup).
void get_input(int& num_read, int& num_hap, read_t*& reads, hap_t*& haps, const char* filename);
int get_output(double* likelihood, int size, const char* filename);

#define MAX_READ_LEN 192      /* pairhmm/xlnx/common/common.h:3-6: limits of the FPGA bundle, kept for callers */
#define MAX_HAP_LEN 1024
#define MAX_RSDATA_NUM 2048
#define MAX_HAPDATA_NUM 128

// ---- compute_fpga ---------------------------------------------------------------------------------
extern double peak_kernel_gcups;
extern double curr_kernel_gcups;
// Returns a library-owned float[>= num_read*num_hap], row-major [read][hap], raw likelihood x 2^120;
// NULL = this batch cannot run on the device (caller falls back, host/main.cpp:330-333); throws
// std::runtime_error on a device failure.  Not re-entrant (as the reference: globals).
float* compute_fpga(const char* bit_path, std::string read_data, std::string hap_data, uint64_t num_cell);
void cleanup();

// ---- FalconPairHMM --------------------------------------------------------------------------------
typedef struct { std::string bases, _q, _i, _d, _c; } Read;
typedef struct { std::string bases; } Hap;
typedef struct { std::vector<Read> reads; std::vector<Hap> haps; } pairhmmInput;
typedef struct { std::vector<double> likelihoodData; } pairhmmOutput;
struct accg_ctx;
struct accg_phmm_mux;
// the process-wide mux the PairHMM entry points of this layer share (created at first use; cleanup() destroys it)
accg_phmm_mux* accg_compat_mux();
// The reference's computePairhmm falls back to its own CPU code (computePairhmmAVX, FalconPairHMM.cpp:69-95) for a region the
// accelerator cannot take (:1184-1193).  This library holds no CPU arithmetic: a caller that wants that branch installs its CPU
// function here; it is called with the same input and must fill output->likelihoodData.  Without one such a region comes back with
// usedFPGA = false and an empty output.
typedef void (*FalconPairHMM_cpu_fn)(pairhmmInput* input, pairhmmOutput* output);
void FalconPairHMM_set_cpu_fallback(FalconPairHMM_cpu_fn fn);
class FalconPairHMM {
 public:
  FalconPairHMM();
  explicit FalconPairHMM(char* bitstream);   // the argument selects nothing here; device 0 (or ACCG_DEVICE)
  ~FalconPairHMM();
  // final log10 likelihoods, index read*numHap + hap; usedFPGA = the device computed them
  void computePairhmm(pairhmmInput* input, pairhmmOutput* output, bool& usedFPGA);
  double get_kernel_time();                  // accumulated device ns
 private:
  accg_phmm_mux* mux_;
  double kernel_ns_;
};

// ---- per-pair entry points ------------------------------------------------------------------------
// testcase + compute_fp_avxs / compute_fp_avxd (pairhmm/xlnx/host/host_type.h:69-73, avx_impl.h:5-6): raw likelihood
// x 2^120 / x 2^1020 of ONE pair.  A device call per pair is only meant for callers that are written that way; batches
// belong in compute_fpga / FalconPairHMM.
typedef struct { int rslen, haplen; const char *q, *i, *d, *c; const char *hap, *rs; } testcase;
extern float (*compute_fp_avxs)(testcase*);
extern double (*compute_fp_avxd)(testcase*);

// ---- SMEM seeding (smem/host/ocl.h:29-32) ----------------------------------------------------------------
typedef uint64_t bwtint_t;
typedef struct { bwtint_t x[3], info; } bwtintv_t;     // libbwa's bwt.h type the reference uses for intervals
#define BANK_NUM 4            /* smem/Makefile:20 */
#define MAX_INTV_ALLOC 256    /* smem/common/common.h:39 */
#define SEQ_LENGTH 256        /* smem/common/common.h:41 */
// Uploads the index once (the reference replicates it per DDR bank here).  btsm (bitstream path) and mem are unused.
void ocl_init(char* btsm, const uint32_t* bwt, const uint64_t* bwt_para, uint64_t bwt_size, bwtintv_t* mem, int batch_size);
// seq: batch x SEQ_LENGTH base codes, seq_len: uint8 per read; mem_output: batch x MAX_INTV_ALLOC intervals, mem_num: counts.
// kernel_time[BANK_NUM] receives the device ns (same value in every slot).  Returns 0.
int smem_ocl(char* btsm, const uint32_t* bwt, const uint64_t* bwt_para, uint8_t* seq, uint8_t* seq_len, int batch_size,
             bwtintv_t* mem_output, int* mem_num, double kernel_time[BANK_NUM]);
void ocl_release();

// ---- HTC Smith-Waterman ---------------------------------------------------------------------------
#define MAX_SEQ_LENGTH 1536
#define MAX_BATCH_SIZE 260
#define OVERHANG_STRATEGY_SOFTCLIP 0
#define OVERHANG_STRATEGY_INDEL 1
#define OVERHANG_STRATEGY_LEADING_INDEL 2
#define OVERHANG_STRATEGY_IGNORE 3
#define W_MATCH 200
#define W_MISMATCH -150
#define W_OPEN -260
#define W_EXTEND -11
#define STATE_MATCH 0
#define STATE_INSERTION 1
#define STATE_DELETION 2
#define STATE_CLIP 4
struct CigarElement { int length; int state; };
struct Cigar { struct CigarElement cigarElements[MAX_SEQ_LENGTH]; int CigarElementNum; };

// context lifecycle under the FPGA host's names (htc-sw/host/smithWatermanHost.h:12-15), called by FalconSWFPGA_init / _release
// in the reference's order (FalconSW_FPGA.cpp:16-27,92-94)
int _init_opencl(const char* bitstream);      // 1: device brought up now; 0: already up, or no device
int _init_kernel_buffer();                    // loads the kernels and sizes the context's device / pinned buffers; 0
int _release_smithWaterman();                 // gives the cached device blocks back; 0
bool FalconSWFPGA_init(char* bitstream);
// Returns device time in ns.  Results in place.  Where the reference computes on the CPU instead (isFPGA false, or a batch its
// device cannot take: FalconSW_FPGA.cpp:43-51 calls SWPairwiseAlignmentMultiBatch, its AVX code), this library has no CPU path of
// its own: the caller installs one with FalconSWFPGA_set_cpu_fallback (the reference's own function, which it still links), and
// FalconSWFPGA_run then calls it and returns its wall time in ns; without one it returns -1 with nothing written.
typedef int (*FalconSW_cpu_fn)(char* ref, int refLength, char alts[][MAX_SEQ_LENGTH], int batchSize, int* altLengths, struct Cigar* cigarResults,
                               int* alignmentOffsets, int overhang_strategy, int option);
void FalconSWFPGA_set_cpu_fallback(FalconSW_cpu_fn fn);
double FalconSWFPGA_run(char* ref, int refLength, char alts[][MAX_SEQ_LENGTH], int* altLengths, int batchSize,
                        int overhang_strategy, int w_match, int w_mismatch, int w_open, int w_extend,
                        struct Cigar* cigarResults, int* alignmentOffsets, bool isFPGA);
void FalconSWFPGA_release();
// inputs: 2*B bytes of little-endian int16 alt lengths, 512 B ref, B x 512 B alts; outputs: shorts
// [0..1] total length (32 bit), [2..2+B) element counts, then per alt (len,state)* in reverse order + alignment_offset.
void _smithWatermanRun(char* inputs, int refLength, int batchSize, int overhang_strategy, int w_match, int w_mismatch,
                       int w_open, int w_extend, short* outputs);
// runSWOnePairBT_fp_avx2 (htc-sw/intel_avx/avx2_impl.h:6): one pair, returns alignment_offset, CIGAR in *cigarRet
extern int32_t (*runSWOnePairBT_fp_avx2)(int32_t match, int32_t mismatch, int32_t open, int32_t extend, uint8_t* seq1, uint8_t* seq2,
                                         int32_t len1, int32_t len2, int8_t overhangStrategy, struct Cigar* cigarRet);
int SWPairwiseAlignmentMultiBatch(char* ref, int refLength, char alts[][MAX_SEQ_LENGTH], int batchSize, int* altLengths,
                                  struct Cigar* cigarResults, int* alignmentOffsets, int overhang_strategy, int option);

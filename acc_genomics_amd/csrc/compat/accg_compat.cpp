// See accg_compat.h.  Everything here is host glue over include/accg.h; no arithmetic of the hot path
// happens on the CPU.
#include "accg_compat.h"
#include <stdio.h>
#include <string.h>
#include <fstream>
#include <sstream>
#include <chrono>
#include <mutex>
#include <stdexcept>
#include "../../../include/accg.h"

// ---- wire format (pairhmm/interface/PairHMMHostInterface.cpp:175-338) -----------------------------
namespace {
struct Writer {
  char* p; uint64_t n = 0;
  void put(const void* s, size_t b) { if (p) memcpy(p + n, s, b); n += b; }
  void i32(int v) { put(&v, 4); }
};
uint64_t write_reads(char* buf, const read_t* r, int num) {
  Writer w{buf};
  w.i32(num);
  for (int i = 0; i < num; i++) {
    w.i32(r[i].len);
    const char* f[5] = {r[i]._b, r[i]._q, r[i]._i, r[i]._d, r[i]._c};
    for (const char* s : f) w.put(s, (size_t)r[i].len);
  }
  return w.n;
}
uint64_t write_haps(char* buf, const hap_t* h, int num) {
  Writer w{buf};
  w.i32(num);
  for (int i = 0; i < num; i++) { w.i32(h[i].len); w.put(h[i]._b, (size_t)h[i].len); }
  return w.n;
}
char* take(const char*& p, int len) {   // malloc(len+1), NUL-terminated, as getStr (:34-41); NULL for len == 0
  if (len <= 0) return nullptr;
  char* d = (char*)malloc((size_t)len + 1);
  memcpy(d, p, (size_t)len); d[len] = '\0'; p += len;
  return d;
}
int get_i32(const char*& p) { int v; memcpy(&v, p, 4); p += 4; return v; }
}  // namespace

void free_reads(read_t* r, int n) {
  for (int i = 0; i < n; i++) { free(r[i]._b); free(r[i]._q); free(r[i]._i); free(r[i]._d); free(r[i]._c); }
  free(r);
}
void free_haps(hap_t* h, int n) { for (int i = 0; i < n; i++) free(h[i]._b); free(h); }
uint64_t serialize(void* buf, const read_t* reads, int num) { return write_reads((char*)buf, reads, num); }
uint64_t serialize(void* buf, const hap_t* haps, int num) { return write_haps((char*)buf, haps, num); }
std::string serialize(const read_t* reads, int num) { std::string s(write_reads(nullptr, reads, num), '\0'); write_reads(&s[0], reads, num); return s; }
std::string serialize(const hap_t* haps, int num) { std::string s(write_haps(nullptr, haps, num), '\0'); write_haps(&s[0], haps, num); return s; }
int deserialize(const void* buf, read_t*& reads) {
  const char* p = (const char*)buf;
  int num = get_i32(p);
  reads = (read_t*)malloc(sizeof(read_t) * (size_t)(num > 0 ? num : 1));
  for (int i = 0; i < num; i++) {
    int len = get_i32(p);
    reads[i].len = len;
    reads[i]._b = take(p, len); reads[i]._q = take(p, len); reads[i]._i = take(p, len); reads[i]._d = take(p, len); reads[i]._c = take(p, len);
  }
  return num;
}
int deserialize(const void* buf, hap_t*& haps) {
  const char* p = (const char*)buf;
  int num = get_i32(p);
  haps = (hap_t*)malloc(sizeof(hap_t) * (size_t)(num > 0 ? num : 1));
  for (int i = 0; i < num; i++) { int len = get_i32(p); haps[i].len = len; haps[i]._b = take(p, len); }
  return num;
}
int deserialize(const std::string& data, read_t*& reads) { return deserialize((const void*)data.data(), reads); }
int deserialize(const std::string& data, hap_t*& haps) { return deserialize((const void*)data.data(), haps); }

// ---- host_tb text files (pairhmm/host/main.cpp:67-159) -------------------------------------------------------
namespace {
std::vector<std::string> tokens_of_next_line(std::ifstream& in) {
  std::string line;
  std::getline(in, line);
  std::istringstream ss(line);
  std::vector<std::string> t;
  for (std::string w; ss >> w;) t.push_back(w);
  return t;
}
}  // namespace
void get_input(int& num_read, int& num_hap, read_t*& reads, hap_t*& haps, const char* filename) {
  std::ifstream in(filename);
  if (!in.good()) throw std::runtime_error(std::string("get_input: cannot open ") + filename);
  auto head = tokens_of_next_line(in);
  if (head.size() != 4) throw std::runtime_error("get_input: bad header");
  num_read = std::stoi(head[1]); num_hap = std::stoi(head[3]);
  reads = (read_t*)malloc(sizeof(read_t) * (size_t)(num_read > 0 ? num_read : 1));
  haps = (hap_t*)malloc(sizeof(hap_t) * (size_t)(num_hap > 0 ? num_hap : 1));
  std::string skip;
  for (int i = 0; i < num_read; i++) {
    auto t = tokens_of_next_line(in);
    if (t.size() != 1) throw std::runtime_error("get_input: bad read length line");
    const int len = std::stoi(t[0]);
    read_t& r = reads[i];
    r.len = len;
    char** field[5] = {&r._b, &r._q, &r._i, &r._d, &r._c};
    for (int f = 0; f < 5; f++) {
      *field[f] = (char*)malloc((size_t)len + 1);
      std::getline(in, skip);                       // label line
      auto v = tokens_of_next_line(in);
      if ((int)v.size() != len) throw std::runtime_error("get_input: field length mismatch");
      for (int k = 0; k < len; k++) (*field[f])[k] = (char)std::stoi(v[k]);
      (*field[f])[len] = '\0';
    }
  }
  std::getline(in, skip);                           // empty line
  for (int j = 0; j < num_hap; j++) {
    auto t = tokens_of_next_line(in);
    if (t.size() != 1) throw std::runtime_error("get_input: bad hap length line");
    haps[j].len = std::stoi(t[0]);
    std::getline(in, skip);                         // label line
    std::string bases;
    std::getline(in, bases);
    if ((int)bases.size() != haps[j].len) throw std::runtime_error("get_input: hap length mismatch");
    haps[j]._b = strdup(bases.c_str());
  }
}
int get_output(double* likelihood, int size, const char* filename) {
  std::ifstream in(filename);
  if (!in.good()) { printf("bad file name %s\n", filename); return 1; }
  for (int i = 0; i < size; i++) {
    double shown; long long bits;
    in >> shown >> bits;
    memcpy(&likelihood[i], &bits, 8);               // the bit pattern is authoritative (main.cpp:150-156)
  }
  return 0;
}

// ---- one lazily created context per process (the reference keeps a global OpenCLEnv, PairHMMFpga.cpp:8) ----
namespace {
accg_ctx* g_ctx = nullptr;
float* g_ret = nullptr;
size_t g_ret_n = 0;
accg_ctx* ctx() {
  if (!g_ctx) {
    const char* d = getenv("ACCG_DEVICE");
    int st = accg_init(d ? atoi(d) : 0, &g_ctx);
    if (st != ACCG_OK) throw std::runtime_error(std::string("accg_init: ") + accg_strerror(st));
  }
  return g_ctx;
}
// The HIP runtime maps streams onto four hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and it reads that variable when it
// first comes up.  This library is loaded by a process whose accelerator work is the mux's lanes -- six small kernel chains side by
// side -- so, unless the variable is set already (or ACCG_KEEP_HW_QUEUES=1), it asks for eight at load time: + 6 % at sixteen caller
// threads (task plugin 1.89 -> 2.01 TCUPS).
__attribute__((constructor)) void accg_compat_hw_queues() {
  const char* keep = getenv("ACCG_KEEP_HW_QUEUES");
  if (!(keep && keep[0] == '1')) setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);
}
// ... and one process-wide mux for the PairHMM entry points (include/accg.h: accg_phmm_mux): compute_fpga, every FalconPairHMM object
// and every task instance put their region through it, so callers on several threads -- an accelerator manager runs several task
// instances at a time -- share device batches instead of queueing small kernels behind each other.  ACCG_MUX_LANES (default 6) and
// ACCG_MUX_MAX_REGIONS (default 64) size it.
accg_phmm_mux* g_mux = nullptr;
std::mutex g_mux_mu;
}  // namespace
accg_phmm_mux* accg_compat_mux() {
  std::lock_guard<std::mutex> g(g_mux_mu);
  if (!g_mux) {
    const char* d = getenv("ACCG_DEVICE"); const char* l = getenv("ACCG_MUX_LANES"); const char* r = getenv("ACCG_MUX_MAX_REGIONS");
    int st = accg_phmm_mux_create(d ? atoi(d) : 0, l && atoi(l) > 0 ? atoi(l) : 6, r && atoi(r) > 0 ? atoi(r) : 64, &g_mux);
    if (st != ACCG_OK) throw std::runtime_error(std::string("accg_phmm_mux_create: ") + accg_strerror(st));
  }
  return g_mux;
}
namespace {
bool is_capability_error(int st) { return st == ACCG_ERR_TOO_LONG || st == ACCG_ERR_BAD_BASE || st == ACCG_ERR_EMPTY_SEQ; }
}  // namespace

double peak_kernel_gcups = 0;
double curr_kernel_gcups = 0;

float* compute_fpga(const char*, std::string read_data, std::string hap_data, uint64_t num_cell) {
  int nr = 0, nh = 0;
  if (read_data.size() >= 4) memcpy(&nr, read_data.data(), 4);
  if (hap_data.size() >= 4) memcpy(&nh, hap_data.data(), 4);
  size_t need = (size_t)MAX_RSDATA_NUM * MAX_HAPDATA_NUM;   // PairHMMFpga.cpp:153-157
  if ((size_t)nr * nh > need) need = (size_t)nr * nh;
  if (need > g_ret_n) { free(g_ret); g_ret = (float*)aligned_alloc(4096, sizeof(float) * need); g_ret_n = need; }
  accg_counters c;
  int st = accg_phmm_mux_region(accg_compat_mux(), read_data.data(), read_data.size(), hap_data.data(), hap_data.size(), ACCG_PHMM_FAST,
                                g_ret, nullptr, &c);
  if (is_capability_error(st)) return NULL;
  if (st != ACCG_OK) throw std::runtime_error(std::string("compute_fpga: ") + accg_strerror(st) + " " + accg_last_hip_error());
  if (c.kernel_ns) {
    curr_kernel_gcups = (double)num_cell / (double)c.kernel_ns;                          // PairHMMFpga.cpp:90-96
    if (curr_kernel_gcups > peak_kernel_gcups) peak_kernel_gcups = curr_kernel_gcups;
  }
  return g_ret;
}
void cleanup() {
  ocl_release();
  { std::lock_guard<std::mutex> g(g_mux_mu); if (g_mux) accg_phmm_mux_destroy(g_mux); g_mux = nullptr; }
  if (g_ctx) accg_shutdown(g_ctx);
  g_ctx = nullptr; free(g_ret); g_ret = nullptr; g_ret_n = 0;
}

// ---- FalconPairHMM ----------------------------------------------------------------------------------
static FalconPairHMM_cpu_fn g_phmm_cpu_fallback = nullptr;
void FalconPairHMM_set_cpu_fallback(FalconPairHMM_cpu_fn fn) { g_phmm_cpu_fallback = fn; }
FalconPairHMM::FalconPairHMM() : mux_(accg_compat_mux()), kernel_ns_(0) {}     // (throws when there is no device, like the reference's constructor)
FalconPairHMM::FalconPairHMM(char*) : FalconPairHMM() {}
FalconPairHMM::~FalconPairHMM() {}
double FalconPairHMM::get_kernel_time() { return kernel_ns_; }
void FalconPairHMM::computePairhmm(pairhmmInput* in, pairhmmOutput* out, bool& usedFPGA) {
  std::vector<read_t> r(in->reads.size());
  std::vector<hap_t> h(in->haps.size());
  for (size_t i = 0; i < r.size(); i++) {
    Read& x = in->reads[i];
    r[i] = {(int)x.bases.size(), &x.bases[0], &x._q[0], &x._i[0], &x._d[0], &x._c[0]};
  }
  for (size_t j = 0; j < h.size(); j++) h[j] = {(int)in->haps[j].bases.size(), &in->haps[j].bases[0]};
  std::string rs = serialize(r.data(), (int)r.size()), hs = serialize(h.data(), (int)h.size());
  out->likelihoodData.assign(r.size() * h.size(), 0.0);
  accg_counters c;
  int st = accg_phmm_mux_region(mux_, rs.data(), rs.size(), hs.data(), hs.size(), ACCG_PHMM_FAST, nullptr,
                                out->likelihoodData.data(), &c);
  usedFPGA = (st == ACCG_OK);
  if (st == ACCG_OK) { kernel_ns_ += (double)c.kernel_ns; return; }
  out->likelihoodData.clear();
  if (!is_capability_error(st)) throw std::runtime_error(std::string("computePairhmm: ") + accg_strerror(st));
  // FalconPairHMM.cpp:1184-1193: a region the accelerator cannot take goes to the host's own CPU code (computePairhmmAVX, :69-95) --
  // the caller's, installed with FalconPairHMM_set_cpu_fallback; this library holds no CPU arithmetic
  if (g_phmm_cpu_fallback) g_phmm_cpu_fallback(in, out);
}

// ---- per-pair PairHMM ------------------------------------------------------------------------------------
namespace {
std::string one_read(const testcase* tc) {
  read_t r = {tc->rslen, (char*)tc->rs, (char*)tc->q, (char*)tc->i, (char*)tc->d, (char*)tc->c};
  return serialize(&r, 1);
}
std::string one_hap(const testcase* tc) { hap_t h = {tc->haplen, (char*)tc->hap}; return serialize(&h, 1); }
float pair_f32(testcase* tc) {
  std::string rs = one_read(tc), hs = one_hap(tc);
  float raw = 0;
  int st = accg_phmm_mux_region(accg_compat_mux(), rs.data(), rs.size(), hs.data(), hs.size(), ACCG_PHMM_FAST, &raw, nullptr, nullptr);
  if (st != ACCG_OK) throw std::runtime_error(std::string("compute_fp_avxs: ") + accg_strerror(st));
  return raw;
}
}  // namespace
extern "C" int accg_phmm_region_f64(accg_ctx*, const void*, size_t, const void*, size_t, double*);
namespace {
double pair_f64(testcase* tc) {
  std::string rs = one_read(tc), hs = one_hap(tc);
  double raw = 0;
  int st = accg_phmm_region_f64(ctx(), rs.data(), rs.size(), hs.data(), hs.size(), &raw);
  if (st != ACCG_OK) throw std::runtime_error(std::string("compute_fp_avxd: ") + accg_strerror(st));
  return raw;
}
}  // namespace
float (*compute_fp_avxs)(testcase*) = &pair_f32;
double (*compute_fp_avxd)(testcase*) = &pair_f64;

// ---- SMEM -----------------------------------------------------------------------------------------------
namespace { accg_smem_index* g_smem = nullptr; uint64_t g_smem_words = 0; }
void ocl_init(char*, const uint32_t* bwt, const uint64_t* bwt_para, uint64_t bwt_size, bwtintv_t*, int) {
  if (g_smem) { accg_smem_index_destroy(g_smem); g_smem = nullptr; }
  // bwt_size counts uint32 words of the block layout (smem/main.cpp:221 passes ceil(bwt_size/16) blocks in bwt_para[6]); it
  // need not be a multiple of 16 and exactly that many words are read (smem/host/ocl.cpp:214-224)
  const uint64_t words = bwt_size;
  int st = accg_smem_index_create(ctx(), bwt, words, bwt_para, &g_smem);
  if (st != ACCG_OK) throw std::runtime_error(std::string("ocl_init: ") + accg_strerror(st));
  g_smem_words = words;
}
int smem_ocl(char*, const uint32_t*, const uint64_t*, uint8_t* seq, uint8_t* seq_len, int batch_size, bwtintv_t* mem_output,
             int* mem_num, double kernel_time[BANK_NUM]) {
  if (!g_smem) throw std::runtime_error("smem_ocl: ocl_init was not called");
  accg_smem_batch* b = nullptr;
  int st = accg_smem_batch_create(g_smem, seq, SEQ_LENGTH, seq_len, (uint32_t)batch_size, MAX_INTV_ALLOC, &b);
  if (st != ACCG_OK) throw std::runtime_error(std::string("smem_ocl: ") + accg_strerror(st));
  float ms = 0;
  st = accg_smem_batch_time(b, 0, 1, &ms);
  if (st == ACCG_OK) st = accg_smem_batch_results(b, mem_output, mem_num);
  accg_smem_batch_destroy(b);
  if (st != ACCG_OK) throw std::runtime_error(std::string("smem_ocl: ") + accg_strerror(st));
  for (int i = 0; i < BANK_NUM; i++) kernel_time[i] = (double)ms * 1e6;
  return 0;
}
void ocl_release() { if (g_smem) accg_smem_index_destroy(g_smem); g_smem = nullptr; }

// ---- HTC Smith-Waterman -------------------------------------------------------------------------------
namespace {
// one ref x B alts -> CIGARs; returns device ns, or -1 when the device cannot take the batch
double sw_batch(const char* ref, int refLength, const char* alts, size_t alt_stride, const int* altLengths, int B, int strategy,
                int wm, int wx, int wo, int we, struct Cigar* cig, int* offs) {
  if (B <= 0) return 0;
  std::vector<int32_t> rl((size_t)B, refLength);
  std::vector<uint8_t> st((size_t)B, (uint8_t)strategy);
  accg_sw_batch* b = nullptr;
  int rc = accg_sw_batch_create(ctx(), B, (const uint8_t*)ref, 0, rl.data(), (const uint8_t*)alts, alt_stride, altLengths, st.data(),
                                wm, wx, wo, we, &b);
  if (is_capability_error(rc)) return -1;
  if (rc != ACCG_OK) throw std::runtime_error(std::string("accg_sw_batch_create: ") + accg_strerror(rc));
  int max_el = 64;
  std::vector<int32_t> n_el((size_t)B), el;
  float ms = 0;
  for (;;) {
    rc = accg_sw_batch_run_cigar(b, max_el);
    el.resize((size_t)B * max_el * 2);
    if (rc == ACCG_OK) rc = accg_sw_batch_cigars(b, n_el.data(), offs, el.data());
    if (rc != ACCG_OK) { accg_sw_batch_destroy(b); throw std::runtime_error(std::string("sw run: ") + accg_strerror(rc)); }
    int need = 0;
    for (int k = 0; k < B; k++) if (n_el[k] < -1 && -n_el[k] > need) need = -n_el[k];
    if (!need) break;
    max_el = need;                                   // a CIGAR longer than the slot: rerun with room (<= MAX_SEQ_LENGTH)
  }
  accg_sw_batch_time(b, 0, 1, &ms);                  // score-only pass timing as the kernel-time figure
  for (int k = 0; k < B; k++) {
    cig[k].CigarElementNum = n_el[k] > 0 ? n_el[k] : 0;
    for (int e = 0; e < cig[k].CigarElementNum; e++) {
      cig[k].cigarElements[e].length = el[((size_t)k * max_el + e) * 2];
      cig[k].cigarElements[e].state = el[((size_t)k * max_el + e) * 2 + 1];
    }
  }
  accg_sw_batch_destroy(b);
  return (double)ms * 1e6;
}
}  // namespace

namespace {
int32_t one_pair_sw(int32_t match, int32_t mismatch, int32_t open, int32_t extend, uint8_t* seq1, uint8_t* seq2, int32_t len1,
                    int32_t len2, int8_t strategy, struct Cigar* cigarRet) {
  int off = 0, al = len2;
  if (sw_batch((const char*)seq1, len1, (const char*)seq2, 0, &al, 1, strategy, match, mismatch, open, extend, cigarRet, &off) < 0)
    throw std::runtime_error("runSWOnePairBT: pair outside the device limits");
  return off;
}
}  // namespace
int32_t (*runSWOnePairBT_fp_avx2)(int32_t, int32_t, int32_t, int32_t, uint8_t*, uint8_t*, int32_t, int32_t, int8_t, struct Cigar*) = &one_pair_sw;

// Context lifecycle under the names of the FPGA host (htc-sw/host/smithWatermanHost.h:12-15).
namespace { int g_sw_init_count = 0; }
// smithWatermanHost.cpp:162-174: brings the device up once; 1 = did it now, 0 = already up (or no device, where the
// reference exits the process -- here the caller's fallback to the CPU code stays possible)
int _init_opencl(const char* /*bitstream*/) {
  if (g_sw_init_count) return 0;
  try { ctx(); } catch (const std::exception&) { return 0; }
  g_sw_init_count++;
  return 1;
}
// smithWatermanHost.cpp:204-209 creates the kernel object and the two device buffers of one FPGA batch (134152 B in, 2 x 266764 B
// out); here one dummy pair goes through the whole path so that the code objects are loaded and the context's device block cache
// and pinned staging are in place before the first timed FalconSWFPGA_run
int _init_kernel_buffer() {
  static struct Cigar c;
  const char ref[] = "ACGTACGTAC", alt[] = "ACGTTCGTAC";
  int al = 10, off = 0;
  sw_batch(ref, 10, alt, 0, &al, 1, OVERHANG_STRATEGY_SOFTCLIP, W_MATCH, W_MISMATCH, W_OPEN, W_EXTEND, &c, &off);
  return 0;
}
// smithWatermanHost.cpp:306-320 releases kernel, event and buffers: the cached device blocks and staging go back to the driver
int _release_smithWaterman() {
  if (g_ctx) accg_ctx_trim(g_ctx);
  return 0;
}
bool FalconSWFPGA_init(char* bitstream) {      // FalconSW_FPGA.cpp:16-27
  static bool init = false;
  if (init) return true;
  if (!_init_opencl(bitstream) && !g_ctx) return false;
  _init_kernel_buffer();
  init = true;
  return true;
}
void FalconSWFPGA_release() { _release_smithWaterman(); }      // FalconSW_FPGA.cpp:92-94
static FalconSW_cpu_fn g_sw_cpu_fallback = nullptr;
void FalconSWFPGA_set_cpu_fallback(FalconSW_cpu_fn fn) { g_sw_cpu_fallback = fn; }
double FalconSWFPGA_run(char* ref, int refLength, char alts[][MAX_SEQ_LENGTH], int* altLengths, int batchSize, int strategy,
                        int wm, int wx, int wo, int we, struct Cigar* cig, int* offs, bool isFPGA) {
  if (batchSize <= 0) return -1;
  double ns = -1;
  if (isFPGA) ns = sw_batch(ref, refLength, &alts[0][0], MAX_SEQ_LENGTH, altLengths, batchSize, strategy, wm, wx, wo, we, cig, offs);
  if (ns >= 0 || !g_sw_cpu_fallback) return ns;
  // FalconSW_FPGA.cpp:43-51: not for the accelerator (or refused by it) -> the caller's CPU code, option 0, timed around the call
  const auto t0 = std::chrono::steady_clock::now();
  if (g_sw_cpu_fallback(ref, refLength, alts, batchSize, altLengths, cig, offs, strategy, 0) != 0) return -1;
  return std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
}
int SWPairwiseAlignmentMultiBatch(char* ref, int refLength, char alts[][MAX_SEQ_LENGTH], int batchSize, int* altLengths,
                                  struct Cigar* cig, int* offs, int strategy, int /*option: CPU variant selector*/) {
  double ns = sw_batch(ref, refLength, &alts[0][0], MAX_SEQ_LENGTH, altLengths, batchSize, strategy, W_MATCH, W_MISMATCH, W_OPEN,
                       W_EXTEND, cig, offs);
  if (ns < 0) return -1;
  for (int k = 0; k < batchSize; k++) if (cig[k].CigarElementNum <= 0) return -1;   // FalconSW_AVX.cpp:307-310
  return 0;
}
void _smithWatermanRun(char* inputs, int refLength, int B, int strategy, int wm, int wx, int wo, int we, short* outputs) {
  const int S = 512;                                 // MAX_FPGA_SEQ_LENGTH, FalconSW_FPGA.cpp:14
  std::vector<int> al((size_t)B), offs((size_t)B);
  for (int i = 0; i < B; i++) al[i] = (int)(unsigned char)inputs[2 * i] | ((int)(signed char)inputs[2 * i + 1] << 8);   // :54-57
  const char* ref = inputs + 2 * B;
  const char* alts = ref + S;
  std::vector<struct Cigar> cig((size_t)B);
  if (sw_batch(ref, refLength, alts, S, al.data(), B, strategy, wm, wx, wo, we, cig.data(), offs.data()) < 0)
    throw std::runtime_error("_smithWatermanRun: batch outside the device limits");
  int p = B + 2;
  for (int i = 0; i < B; i++) {
    outputs[2 + i] = (short)cig[i].CigarElementNum;
    for (int e = cig[i].CigarElementNum - 1; e >= 0; e--) {       // reverse order, reader at :83-86
      outputs[p++] = (short)cig[i].cigarElements[e].length;
      outputs[p++] = (short)cig[i].cigarElements[e].state;
    }
    outputs[p++] = (short)offs[i];
  }
  outputs[0] = (short)(p & 0xFFFF); outputs[1] = (short)((unsigned)p >> 16);
}

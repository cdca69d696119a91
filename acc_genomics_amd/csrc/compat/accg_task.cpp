#include "accg_task.h"
#include <stdlib.h>
#include <string.h>
#include <stdexcept>
#include "../../../include/accg.h"

PairHMM::PairHMM() : task_host::Task(3), ctx_(nullptr), batch_(nullptr), num_cell_(0) {}
PairHMM::~PairHMM() {
  if (batch_) accg_phmm_batch_destroy(batch_);
  if (ctx_) accg_shutdown(ctx_);
}
void PairHMM::prepare() {
  if (!ctx_) {
    std::string dev;
    int st = accg_init(get_conf("device", dev) ? atoi(dev.c_str()) : 0, &ctx_);    // the reference reads bankID / num_pe here
    if (st != ACCG_OK) throw std::runtime_error(std::string("PairHMM::prepare: ") + accg_strerror(st));
  }
  if (getInputBytes(0) >= 8) memcpy(&num_cell_, getInput(0), 8);                    // PairHMMTask.cpp:35
  const void* rs[1] = {getInput(1)}; const void* hs[1] = {getInput(2)};
  size_t rb[1] = {getInputBytes(1)}, hb[1] = {getInputBytes(2)};
  if (batch_) { accg_phmm_batch_destroy(batch_); batch_ = nullptr; }
  int st = accg_phmm_batch_create(ctx_, 1, rs, rb, hs, hb, &batch_);
  if (st != ACCG_OK) throw std::runtime_error(std::string("PairHMM::prepare: ") + accg_strerror(st));   // blaze::invalidParam in the reference
}
void PairHMM::compute() {
  if (!batch_) throw std::runtime_error("PairHMM::compute before prepare");
  int st = accg_phmm_batch_run(batch_, ACCG_PHMM_FAST);
  std::vector<float>& out = setOutput(0, (size_t)accg_phmm_batch_pairs(batch_));
  if (st == ACCG_OK) st = accg_phmm_batch_results(batch_, out.data(), nullptr, nullptr);
  if (st != ACCG_OK) throw std::runtime_error(std::string("PairHMM::compute: ") + accg_strerror(st));
}
extern "C" task_host::Task* create() { return new PairHMM(); }
extern "C" void destroy(task_host::Task* p) { delete p; }

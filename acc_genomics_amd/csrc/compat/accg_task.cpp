#include "accg_task.h"
#include <stdlib.h>
#include <string.h>
#include <stdexcept>
#include "../../../include/accg.h"
#include "accg_compat.h"

// An accelerator manager makes one task instance per request and runs several at a time.  What the reference caches in the task
// environment across instances (the OpenCL context, the device buffers: PairHMMTask.cpp:70-78) is here the process-wide mux of the
// compat layer: an instance owns nothing on the device, create() .. destroy() is a few pointer assignments around one region call, and
// concurrent instances share device batches (accg_phmm_mux, include/accg.h).
PairHMM::PairHMM() : task_host::Task(3), num_cell_(0), prepared_(false) {}
PairHMM::~PairHMM() {}
void PairHMM::prepare() {
  // the reference deserializes and packs here; the wire blobs ARE the device's input format, so what is left is the checks that can
  // fail before any device work (blaze::invalidParam in the reference)
  if (getInputBytes(0) >= 8) memcpy(&num_cell_, getInput(0), 8);                    // PairHMMTask.cpp:35
  if (!getInput(1) || !getInput(2) || getInputBytes(1) < 4 || getInputBytes(2) < 4) throw std::runtime_error("PairHMM::prepare: missing input block");
  prepared_ = true;
}
void PairHMM::compute() {
  if (!prepared_) throw std::runtime_error("PairHMM::compute before prepare");
  int32_t nr = 0, nh = 0;
  memcpy(&nr, getInput(1), 4); memcpy(&nh, getInput(2), 4);
  if (nr < 0 || nh < 0) throw std::runtime_error("PairHMM::compute: malformed input block");
  std::vector<float>& out = setOutput(0, (size_t)nr * (size_t)nh);
  const int st = accg_phmm_mux_region(accg_compat_mux(), getInput(1), getInputBytes(1), getInput(2), getInputBytes(2), ACCG_PHMM_FAST, out.data(), nullptr, nullptr);
  if (st != ACCG_OK) throw std::runtime_error(std::string("PairHMM::compute: ") + accg_strerror(st));
}
extern "C" task_host::Task* create() { return new PairHMM(); }
extern "C" void destroy(task_host::Task* p) { delete p; }

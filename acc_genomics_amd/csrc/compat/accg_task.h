// The PairHMM accelerator-task plugin shape of the reference (pairhmm/task/xlnx/PairHMMTask.h:58-96, .cpp:27-143):
// a Task subclass with prepare()/compute(), exported through extern "C" create()/destroy() for dlopen by an
// accelerator manager.  Blaze itself is not in the reference tree (pairhmm/cmake/FindBlaze.cmake:1-7 downloads it), so
// `task_host::Task` below is this repo's own minimal stand-in for the part of blaze::Task the plugin uses:
//   three input blocks   [0] uint64 num_cell, [1] serialized reads, [2] serialized haps   (PairHMMTask.cpp:35-38)
//   one output block     [0] float[num_read * num_hap], raw likelihood x 2^120           (:70-78)
//   string configuration get_conf(key, value)                                             (PairHMMTask.h:71-75)
// Porting to the real Blaze is a change of base class and of the three accessor names.
#pragma once
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

namespace task_host {
class Task {
 public:
  explicit Task(int n_inputs) : in_(n_inputs), in_bytes_(n_inputs, 0) {}
  virtual ~Task() {}
  virtual void prepare() = 0;
  virtual void compute() = 0;
  // manager side
  void setInput(int i, const void* p, size_t bytes) { in_.at(i) = p; in_bytes_.at(i) = bytes; }
  void set_conf(const std::string& k, const std::string& v) { conf_[k] = v; }
  const std::vector<float>& getOutputBlock(int i) const { return out_.at(i); }
  // plugin side
 protected:
  const void* getInput(int i) const { return in_.at(i); }
  size_t getInputBytes(int i) const { return in_bytes_.at(i); }
  bool get_conf(const std::string& k, std::string& v) const { auto it = conf_.find(k); if (it == conf_.end()) return false; v = it->second; return true; }
  std::vector<float>& setOutput(int i, size_t n) { if ((int)out_.size() <= i) out_.resize(i + 1); out_[i].assign(n, 0.f); return out_[i]; }
 private:
  std::vector<const void*> in_;
  std::vector<size_t> in_bytes_;
  std::vector<std::vector<float>> out_;
  std::map<std::string, std::string> conf_;
};
}  // namespace task_host

class PairHMM : public task_host::Task {
 public:
  PairHMM();
  virtual ~PairHMM();
  virtual uint64_t estimateClientTime() { return 0; }
  virtual uint64_t estimateTaskTime() { return 0; }
  virtual void prepare();    // checks the input blocks (replaces deserialize + pack_fpga_input: the wire blobs are the device's input format)
  virtual void compute();    // the region through the process-wide mux, output block 0 filled (replaces clEnqueueMigrateMemObjects + clEnqueueTask)
 private:
  uint64_t num_cell_;
  bool prepared_;
};

extern "C" task_host::Task* create();
extern "C" void destroy(task_host::Task* p);

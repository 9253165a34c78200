// torch_nodes.h — what the translation units of _irx_nodes.so share (torch_nodes.cpp: one node per operator; heads_nodes.cpp: one
// node per HEAD, round 6): the gradient-sink handle of optim.FlatAdam and the error check around a C-ABI call.
#pragma once
#include <torch/extension.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace irxn {

typedef const char* (*last_error_fn)();
extern last_error_fn g_last_error;          // irx_last_error of the bound library (thread-local message: read on the failing thread)

inline void check(int rc, const char* what) {
  if (rc == 0) return;
  const char* msg = g_last_error ? g_last_error() : "";
  throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + (msg ? msg : ""));
}

// Gradient-sink handle (optim.FlatAdam.native_sink): slot addresses followed by [record address, generation, address of the
// generation counter]. deliver() is true when the node may write the optimizer's slots NOW: the expected number of slots came
// along, the optimizer that handed them out is still the current one (generation unchanged since the forward) and the producer
// has not delivered since the last zero_grad() (record[0] == 0). The tensors the addresses point into travel with the node, so
// a retired optimizer's buffers stay valid until the graph is gone.
struct Sink {
  std::vector<int64_t> slots;
  int64_t* rec = nullptr;
  int64_t gen = 0;
  const int64_t* gen_now = nullptr;
  Sink() = default;
  explicit Sink(const std::vector<int64_t>& v) {
    if (v.size() > 3) {
      slots.assign(v.begin(), v.end() - 3);
      rec = (int64_t*)v[v.size() - 3];
      gen = v[v.size() - 2];
      gen_now = (const int64_t*)v[v.size() - 1];
    }
  }
  bool deliver(size_t expected) const {
    return slots.size() == expected && rec != nullptr && gen_now != nullptr && *gen_now == gen && rec[0] == 0;
  }
  void delivered(void* stream) const { rec[1] = (int64_t)stream; rec[0] = 1; }
};

inline const float* fp(const at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline float* fpm(const at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

}  // namespace irxn

// Compiles a reference op file (csrc/spmm.cpp, csrc/convert.cpp) UNMODIFIED, where it
// lies under /root/reference, but makes its static `torch::RegisterOperators().op(
// "torch_sparse::...")` land in the `ts_ref::` namespace so that the compiled reference
// can be loaded next to the product (which owns `torch_sparse::`) in one process.
// TEST INFRASTRUCTURE ONLY (see oracle/build_ref.py).
#pragma once
#include <torch/script.h>
#include <torch/torch.h>

#include <string>
#include <utility>

namespace torch {
struct TsRefRegisterOperators {
  c10::RegisterOperators real;
  template <class F>
  TsRefRegisterOperators &&op(const std::string &name, F &&f) && {
    std::string n = name;
    const std::string from = "torch_sparse::";
    if (n.compare(0, from.size(), from) == 0) n = "ts_ref::" + n.substr(from.size());
    (void)std::move(real).op(n, std::forward<F>(f));  // registers in place, returns *this
    return std::move(*this);
  }
};
}  // namespace torch
#define RegisterOperators TsRefRegisterOperators

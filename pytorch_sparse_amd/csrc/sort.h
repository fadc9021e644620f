#pragma once
#include "common.h"
namespace tsamd {
size_t sort_pairs_workspace_bytes(int64_t n);
// Stable sort of (key, payload) by the low `key_bits` bits of key.  vals_in == nullptr sorts
// the identity permutation (argsort).  Inputs are not modified; in/out must not alias.
// todo != nullptr: device word; when it holds 0 at run time the passes do nothing (the outputs are then
// left untouched -- the caller fills them) -- a sort that is decided on the device, without a host sync.
int sort_pairs(const int64_t *keys_in, const int64_t *vals_in, int64_t *keys_out,
               int64_t *vals_out, int64_t n, int key_bits, void *workspace, hipStream_t stream,
               const int64_t *todo = nullptr);
}  // namespace tsamd

#pragma once
#include "common.h"
namespace tsamd {
size_t sort_pairs_workspace_bytes(int64_t n);
// Stable sort of (key, payload) by the low `key_bits` bits of key.  vals_in == nullptr sorts
// the identity permutation (argsort).  Inputs are not modified; in/out must not alias.
int sort_pairs(const int64_t *keys_in, const int64_t *vals_in, int64_t *keys_out,
               int64_t *vals_out, int64_t n, int key_bits, void *workspace, hipStream_t stream);
}  // namespace tsamd

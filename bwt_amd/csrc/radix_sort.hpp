// Stable LSD radix sort of (uint64 key, uint32 value) pairs in HBM, 8 bits per
// pass.  Used by the prefix-doubling path of the suffix-array build.
#pragma once

#include "common.hpp"

namespace bwtgpu {

size_t radix_sort_workspace_bytes(uint64_t count);

// Sorts by key bits [0, key_bits).  keys/vals hold the input and, on return,
// the sorted output; keys_alt/vals_alt are same-sized scratch.
int radix_sort_pairs(uint64_t *keys, uint32_t *vals, uint64_t *keys_alt, uint32_t *vals_alt,
                     uint64_t count, int key_bits, void *d_ws, size_t ws_bytes, hipStream_t s);

}  // namespace bwtgpu

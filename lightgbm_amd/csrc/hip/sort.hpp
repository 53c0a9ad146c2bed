// Device sort of 64-bit keys (ascending), used by the bin finder (per-feature value samples) and by
// the AUC metric (scores).  Several independent arrays of the same padded length can be sorted at once.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace lga {

// Length the key arrays must be padded to (a power of two >= 2048); fill the tail with ~0ull.
int64_t sort_padded_size(int64_t n);
// keys: [batches][npad], sorted in place.
void bitonic_sort_u64(uint64_t* keys, int64_t npad, int batches, hipStream_t s);

}  // namespace lga

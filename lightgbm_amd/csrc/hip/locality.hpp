// Locality order of a booster's training rows (locality.hip).
#pragma once
#include <vector>
#include "device.hpp"

namespace lga {

// The two index lists of a locality order: internal position -> the caller's (root) row and back.
struct LocalityOrder {
  DeviceBuffer<int32_t> ext_of_int;
  DeviceBuffer<int32_t> int_of_ext;
};

// Fills `dst` with copies of `src`'s matrices, labels and weights in a locality order of the rows (single feature tile
// only) and `ord` with the index lists.  `src` must hold its labels already.  False: nothing was built (several
// tiles, no usable feature).
bool build_locality_mirror(const DeviceMirror& src, const std::vector<FeatureMeta>& meta, DeviceMirror* dst, LocalityOrder* ord,
                           hipStream_t s);
// dst[j] = src[ext_of_int[j]]
void gather_rows_f64(const double* src, const int32_t* ext_of_int, int64_t n, double* dst, hipStream_t s);

}  // namespace lga

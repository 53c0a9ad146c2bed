// Per-feature quantile binning (what `lgb.Dataset(matrix, label=)` at
// r/gridsearchCV.R:52-53 triggers inside the engine): values -> at most
// max_bin integer bins, with real-valued upper bounds kept so that models can
// store real thresholds.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lga {

enum class MissingType : int32_t { None = 0, Zero = 1, NaN = 2 };

struct BinMapper {
  std::vector<double> upper;   // upper[i]: inclusive upper bound of value bin i; last is +inf
  int32_t num_bin = 1;         // value bins + (1 if a NaN bin exists)
  MissingType missing = MissingType::None;
  int32_t missing_bin = -1;    // == num_bin-1 when missing == NaN
  double min_val = 0.0, max_val = 0.0;
  bool trivial = true;         // a single bin: the feature cannot split

  // Build from a sample of one feature.  `values` may contain NaN; it is
  // modified (sorted).  total_sample_cnt is the number of sampled rows.
  void find_bins(std::vector<double>& values, int max_bin, int min_data_in_bin, bool use_missing);

  // value -> bin.  NaN goes to the NaN bin when one exists, otherwise it is
  // treated as 0.0 (the rule the predictor applies too).
  inline int value_to_bin(double v) const {
    if (v != v) {
      if (missing == MissingType::NaN) return missing_bin;
      v = 0.0;
    }
    int nv = (int)upper.size();
    int lo = 0, hi = nv - 1;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (v <= upper[mid]) hi = mid; else lo = mid + 1;
    }
    return lo;
  }

  // Real threshold written to the model for "bin <= t".
  double bin_to_threshold(int t) const { return upper[t]; }
  std::string info_string() const;   // "[min:max]" or "none"
};

// Greedy equal-frequency cut of sorted distinct values into <= max_bin bins.
// Returns the inclusive upper bounds (last = +inf).
std::vector<double> greedy_find_bounds(const std::vector<double>& distinct, const std::vector<int>& counts,
                                       int max_bin, int64_t total_cnt, int min_data_in_bin);

}  // namespace lga

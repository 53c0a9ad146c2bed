// Per-feature quantile binning (what `lgb.Dataset(matrix, label=)` at
// r/gridsearchCV.R:52-53 triggers inside the engine): values -> at most
// max_bin integer bins, with real-valued upper bounds kept so that models can
// store real thresholds.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace lga {

enum class MissingType : int32_t { None = 0, Zero = 1, NaN = 2 };

struct BinMapper {
  std::vector<double> upper;   // upper[i]: inclusive upper bound of value bin i; last is +inf
  int32_t num_bin = 1;         // value bins + (1 if a NaN bin exists)
  MissingType missing = MissingType::None;
  int32_t missing_bin = -1;    // == num_bin-1 when a missing bin exists (NaN, or zero and NaN)
  double min_val = 0.0, max_val = 0.0;
  bool trivial = true;         // a single bin: the feature cannot split
  // Categorical feature: values are non-negative integer codes.  `upper` then holds the kept category
  // values (bin i = the i-th most frequent category), and one more bin (missing_bin == upper.size()) takes
  // everything else: NaN, negative codes, categories too rare or unseen when the bins were found.
  bool categorical = false;
  std::vector<std::pair<int64_t, int32_t>> cat_lookup;   // (category, bin) sorted by category

  // Build from a sample of one feature.  `values` may contain NaN; it is
  // modified (sorted).  total_sample_cnt is the number of sampled rows.
  // zero_as_missing: zeros (|v| <= 1e-35) share the missing bin with NaN (MissingType::Zero).
  // forced: upper bounds the feature must have (forcedbins_filename; ascending).  They are taken first - as many as
  // max_bin leaves room for - and the bins that remain are dealt to the intervals between them in proportion to
  // the sampled rows each holds, cut the usual greedy way inside an interval.
  void find_bins(std::vector<double>& values, int max_bin, int min_data_in_bin, bool use_missing,
                 bool zero_as_missing = false, const std::vector<double>* forced = nullptr);
  // Categorical form: the (max_bin - 1) most frequent categories that occur at least min_data_in_bin times
  // (ties: smaller category first) get a bin each.
  void find_bins_categorical(const std::vector<double>& values, int max_bin, int min_data_in_bin);
  void build_cat_lookup();

  // value -> bin.  NaN goes to the NaN bin when one exists, otherwise it is
  // treated as 0.0 (the rule the predictor applies too).
  inline int value_to_bin(double v) const {
    if (categorical) {
      if (!(v >= 0.0)) return missing_bin;            // NaN and negative codes
      const int64_t c = (int64_t)v;
      size_t lo = 0, hi = cat_lookup.size();
      while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if (cat_lookup[mid].first < c) lo = mid + 1; else hi = mid;
      }
      return (lo < cat_lookup.size() && cat_lookup[lo].first == c) ? cat_lookup[lo].second : missing_bin;
    }
    if (missing == MissingType::Zero && (v != v || (v > -1e-35 && v < 1e-35))) return missing_bin;
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
  std::string info_string() const;   // "[min:max]", "none", or the categories "c0:c1:..." of a categorical feature
};

// Greedy equal-frequency cut of sorted distinct values into <= max_bin bins.
// Returns the inclusive upper bounds (last = +inf).
std::vector<double> greedy_find_bounds(const std::vector<double>& distinct, const std::vector<int>& counts,
                                       int max_bin, int64_t total_cnt, int min_data_in_bin);

}  // namespace lga

// Leaf outputs of the objectives whose optimum is a percentile of the residuals rather than a Newton step:
// regression_l1 (median), quantile (alpha level) and mape (median weighted by 1 / max(1, |label|)).  After a
// tree is grown on the sign gradients, every leaf's value is replaced by the (weighted) percentile of
// label - score over the leaf's in-bag rows.  Residuals are rounded to float before they are ranked, on the
// host and on the device alike, so both produce the same values.
#pragma once
#include "backend.hpp"
#include "common.hpp"
#include <algorithm>
#include <vector>

namespace lga {

// asc(i): i-th smallest residual.  The rank rule counts (1 - alpha) * n values down from the largest and
// interpolates between the two neighbours; below one value from the top it returns the largest, beyond the
// last the smallest.
template <typename Asc>
LGA_HD double percentile_unweighted(Asc asc, int64_t n, double alpha) {
  if (n <= 1) return n == 1 ? asc(0) : 0.0;
  const double float_pos = (1.0 - alpha) * (double)n;
  const int64_t pos = (int64_t)float_pos;
  if (pos < 1) return asc(n - 1);
  if (pos >= n) return asc(0);
  const double bias = float_pos - (double)pos;
  const double v1 = asc(n - pos), v2 = asc(n - 1 - pos);
  return v1 - (v1 - v2) * bias;
}

struct WeightedResidual {
  float value;
  float weight;
};

// items sorted ascending by value.  Walks the cumulative weights to alpha * total; interpolates towards the
// previous value only when the step of the cumulative weight at that position is at least one.
inline double percentile_weighted(const WeightedResidual* items, int64_t n, double alpha) {
  if (n <= 1) return n == 1 ? (double)items[0].value : 0.0;
  std::vector<double> cdf((size_t)n);
  double run = 0.0;
  for (int64_t i = 0; i < n; ++i) { run += (double)items[i].weight; cdf[(size_t)i] = run; }
  const double threshold = run * alpha;
  int64_t pos = std::upper_bound(cdf.begin(), cdf.end(), threshold) - cdf.begin();
  pos = std::min(pos, n - 1);
  if (pos == 0 || pos == n - 1) return (double)items[pos].value;
  const double v1 = (double)items[pos - 1].value, v2 = (double)items[pos].value;
  const double step = cdf[(size_t)pos + 1] - cdf[(size_t)pos];
  if (step >= 1.0) return (threshold - cdf[(size_t)pos]) / step * (v2 - v1) + v1;
  return v2;
}

// Distributed learners: a leaf's value is the mean of the local percentiles of the ranks that hold rows of it
// (values[l] is ignored where has[l] == 0).  Leaves no rank holds keep `keep[l]`.
inline void merge_leaf_percentiles(Comm* comm, std::vector<double>& values, const std::vector<int>& has,
                                   const std::vector<double>& keep) {
  const size_t L = values.size();
  if (comm && comm->size() > 1) {
    std::vector<double> buf(2 * L);
    for (size_t l = 0; l < L; ++l) { buf[l] = has[l] ? values[l] : 0.0; buf[L + l] = has[l] ? 1.0 : 0.0; }
    comm->allreduce_sum_f64(buf.data(), buf.size(), false, nullptr);
    for (size_t l = 0; l < L; ++l) values[l] = buf[L + l] > 0.0 ? buf[l] / buf[L + l] : keep[l];
  } else {
    for (size_t l = 0; l < L; ++l) if (!has[l]) values[l] = keep[l];
  }
}

}  // namespace lga

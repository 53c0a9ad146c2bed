#include "bin_mapper.hpp"
#include "binning_core.hpp"
#include <algorithm>
#include <cmath>
#include <limits>
#include <map>
#include <sstream>

namespace lga {

std::vector<double> greedy_find_bounds(const std::vector<double>& distinct, const std::vector<int>& counts,
                                       int max_bin, int64_t total_cnt, int min_data_in_bin) {
  std::vector<double> out((size_t)std::max(max_bin, 1) + 1);
  int nb = greedy_bounds_core(distinct.data(), counts.data(), (int)distinct.size(), max_bin, total_cnt,
                              min_data_in_bin, out.data());
  out.resize(nb);
  return out;
}

namespace {
// Bounds with the forced ones among them (upstream LightGBM: bin.cpp, FindBinWithPredefinedBin - the same idea, not
// the same cuts: the forced bounds first, the rest of the budget spread over the intervals by their row counts).
std::vector<double> bounds_with_forced(const std::vector<double>& distinct, const std::vector<int>& counts, int max_bin,
                                       int64_t total_cnt, int min_data_in_bin, const std::vector<double>& forced_in) {
  const double inf = std::numeric_limits<double>::infinity();
  std::vector<double> forced;
  for (double b : forced_in)
    if (b == b && b < inf && b > -inf && (forced.empty() || b > forced.back())) forced.push_back(b);
  if ((int)forced.size() > max_bin - 1) forced.resize((size_t)std::max(max_bin - 1, 0));
  std::vector<double> out(forced);
  int remaining = max_bin - (int)forced.size() - 1;
  size_t lo = 0;
  for (size_t k = 0; k <= forced.size() && remaining > 0 && total_cnt > 0; ++k) {
    // distinct values in (forced[k-1], forced[k]]
    size_t hi = lo;
    int64_t cnt = 0;
    while (hi < distinct.size() && (k == forced.size() || distinct[hi] <= forced[k])) cnt += counts[hi++];
    const int extra = (int)((double)remaining * (double)cnt / (double)total_cnt);
    if (extra >= 1 && hi - lo >= 2) {
      std::vector<double> d(distinct.begin() + (std::ptrdiff_t)lo, distinct.begin() + (std::ptrdiff_t)hi);
      std::vector<int> c(counts.begin() + (std::ptrdiff_t)lo, counts.begin() + (std::ptrdiff_t)hi);
      std::vector<double> inner = greedy_find_bounds(d, c, extra + 1, cnt, min_data_in_bin);
      for (size_t j = 0; j + 1 < inner.size(); ++j)
        if (k == forced.size() || inner[j] < forced[k]) out.push_back(inner[j]);
    }
    lo = hi;
  }
  std::sort(out.begin(), out.end());
  out.erase(std::unique(out.begin(), out.end()), out.end());
  out.push_back(inf);
  return out;
}
}  // namespace

void BinMapper::find_bins(std::vector<double>& values, int max_bin, int min_data_in_bin, bool use_missing,
                          bool zero_as_missing, const std::vector<double>* forced) {
  // strip NaN (and zeros when they count as missing)
  const bool zeros_missing = use_missing && zero_as_missing;
  size_t n_all = values.size();
  size_t w = 0;
  for (size_t i = 0; i < n_all; ++i)
    if (values[i] == values[i] && !(zeros_missing && values[i] > -1e-35 && values[i] < 1e-35)) values[w++] = values[i];
  size_t n_nan = n_all - w;
  values.resize(w);
  std::sort(values.begin(), values.end());
  std::vector<double> distinct;
  std::vector<int> counts;
  for (size_t i = 0; i < values.size(); ++i) {
    if (i == 0 || values[i] != values[i - 1]) { distinct.push_back(values[i]); counts.push_back(1); }
    else counts.back()++;
  }
  missing = (use_missing && n_nan > 0) ? (zeros_missing ? MissingType::Zero : MissingType::NaN) : MissingType::None;
  int value_bins = max_bin - (missing != MissingType::None ? 1 : 0);
  if (value_bins < 1) value_bins = 1;
  upper = (forced && !forced->empty())
              ? bounds_with_forced(distinct, counts, value_bins, (int64_t)values.size(), min_data_in_bin, *forced)
              : greedy_find_bounds(distinct, counts, value_bins, (int64_t)values.size(), min_data_in_bin);
  int nv = (int)upper.size();
  if (missing != MissingType::None) { missing_bin = nv; num_bin = nv + 1; }
  else { missing_bin = -1; num_bin = nv; }
  if (!distinct.empty()) { min_val = distinct.front(); max_val = distinct.back(); }
  else { min_val = max_val = 0.0; }
  trivial = num_bin <= 1;
}

void BinMapper::find_bins_categorical(const std::vector<double>& values, int max_bin, int min_data_in_bin) {
  categorical = true;
  std::map<int64_t, int64_t> counts;
  for (double v : values)
    if (v >= 0.0) ++counts[(int64_t)v];
  std::vector<std::pair<int64_t, int64_t>> order(counts.begin(), counts.end());   // (category, count)
  std::stable_sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
  const size_t cap = (size_t)std::max(1, std::min(max_bin, 256) - 1);
  upper.clear();
  for (const auto& e : order) {
    if (upper.size() >= cap || e.second < std::max(1, min_data_in_bin)) break;
    upper.push_back((double)e.first);
  }
  const int k = (int)upper.size();
  missing = MissingType::NaN;
  missing_bin = k;
  num_bin = k + 1;
  min_val = max_val = k ? upper[0] : 0.0;
  for (double c : upper) { min_val = std::min(min_val, c); max_val = std::max(max_val, c); }
  trivial = k < 1;
  if (trivial) upper.assign(1, 0.0);   // keep the "one bound at least" shape of every mapper
  build_cat_lookup();
}

void BinMapper::build_cat_lookup() {
  cat_lookup.clear();
  if (!categorical || trivial) return;
  for (size_t i = 0; i < upper.size(); ++i) cat_lookup.emplace_back((int64_t)upper[i], (int32_t)i);
  std::sort(cat_lookup.begin(), cat_lookup.end());
}

std::string BinMapper::info_string() const {
  if (trivial) return "none";
  if (categorical) {
    std::ostringstream os;
    for (size_t i = 0; i < upper.size(); ++i) os << (i ? ":" : "") << (long long)upper[i];
    return os.str();
  }
  std::ostringstream os;
  os.precision(17);
  os << "[" << min_val << ":" << max_val << "]";
  return os.str();
}

}  // namespace lga

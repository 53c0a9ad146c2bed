// Bin-boundary search shared by the host BinMapper and the device binning
// kernel (one feature per wave on the GPU runs exactly this routine).
#pragma once
#include "common.hpp"

namespace lga {

LGA_HD double bound_between(double a, double b) {
  // a < b.  A value strictly inside [a, b) so that "x <= bound" keeps a left and b right.
  double m = a + (b - a) * 0.5;
  if (!(m < b)) m = a;
  return m;
}

// distinct[0..n) ascending, counts[i] > 0.  Writes at most max_bin inclusive
// upper bounds to `out` (the last one is +inf) and returns how many.
LGA_HD int greedy_bounds_core(const double* distinct, const int* counts, int n, int max_bin,
                              int64_t total_cnt, int min_data_in_bin, double* out) {
  const double inf = std::numeric_limits<double>::infinity();
  int nb = 0;
  if (n <= 0) { out[nb++] = inf; return nb; }
  if (max_bin < 1) max_bin = 1;
  if (n <= max_bin) {
    // one bin per distinct value, merging neighbours until a bin holds min_data_in_bin rows
    int64_t cur = 0;
    for (int i = 0; i + 1 < n; ++i) {
      cur += counts[i];
      if (cur >= min_data_in_bin) {
        out[nb++] = bound_between(distinct[i], distinct[i + 1]);
        cur = 0;
      }
    }
    out[nb++] = inf;
    return nb;
  }
  // equal-frequency: values at least as frequent as the mean bin size get a bin of their own
  double mean = (double)total_cnt / (double)max_bin;
  if (mean < (double)min_data_in_bin) mean = (double)min_data_in_bin;
  int big = 0;
  int64_t big_cnt = 0;
  for (int i = 0; i < n; ++i) if ((double)counts[i] >= mean) { ++big; big_cnt += counts[i]; }
  int rest_bins = max_bin - big;
  int64_t rest_cnt = total_cnt - big_cnt;
  double mean_small = rest_bins > 0 ? (double)rest_cnt / (double)rest_bins : (double)rest_cnt + 1.0;
  if (mean_small < (double)min_data_in_bin) mean_small = (double)min_data_in_bin;
  int64_t cur = 0;
  for (int i = 0; i + 1 < n; ++i) {
    bool big_i = (double)counts[i] >= mean;
    bool big_next = (double)counts[i + 1] >= mean;
    if (!big_i) rest_cnt -= counts[i];
    cur += counts[i];
    if (big_i || big_next || (double)cur >= mean_small) {
      if (big_i || cur >= min_data_in_bin) {
        out[nb++] = bound_between(distinct[i], distinct[i + 1]);
        cur = 0;
        if (nb >= max_bin - 1) break;
        if (!big_i) {
          --rest_bins;
          mean_small = rest_bins > 0 ? (double)rest_cnt / (double)rest_bins : (double)rest_cnt + 1.0;
          if (mean_small < (double)min_data_in_bin) mean_small = (double)min_data_in_bin;
        }
      }
    }
  }
  out[nb++] = inf;
  return nb;
}

// The same cut as greedy_bounds_core, computed by jumping from cut to cut instead of visiting every
// distinct value: with prefix sums a cut position is one binary search away.  Used by the device bin
// finder, where one lane walking 200 k values would take tens of milliseconds per feature.
//   prefix[i]     = counts[0] + .. + counts[i-1]                         (n + 1 entries)
//   small_pref[i] = the same over values that are NOT "big" (count >= mean bin size)
//   next_big[i]   = smallest j >= i whose value is big, n if none         (n + 1 entries)
LGA_HD int greedy_bounds_jump(const double* distinct, const int* counts, const int64_t* prefix, const int64_t* small_pref,
                              const int* next_big, int n, int max_bin, int64_t total_cnt, int min_data_in_bin,
                              double* out) {
  const double inf = std::numeric_limits<double>::infinity();
  if (max_bin < 1) max_bin = 1;
  if (n <= max_bin) return greedy_bounds_core(distinct, counts, n, max_bin, total_cnt, min_data_in_bin, out);
  int nb = 0;
  double mean = (double)total_cnt / (double)max_bin;
  if (mean < (double)min_data_in_bin) mean = (double)min_data_in_bin;
  // number of big values: they are exactly the indices j with next_big[j] == j
  int big = 0;
  for (int j = next_big[0]; j < n; j = next_big[j + 1]) ++big;
  const int64_t total_small = small_pref[n];
  int rest_bins = max_bin - big;
  int64_t rest_cnt = total_small;
  double mean_small = rest_bins > 0 ? (double)rest_cnt / (double)rest_bins : (double)rest_cnt + 1.0;
  if (mean_small < (double)min_data_in_bin) mean_small = (double)min_data_in_bin;
  int start = 0;   // the running bin holds values start .. i
  int pos = 0;
  while (pos <= n - 2) {
    // first i >= pos where value i or value i+1 is big
    const int nbig = next_big[pos];
    const int i1 = nbig == pos ? pos : nbig - 1;
    // first i >= pos where the running bin reaches mean_small
    int lo = pos, hi = n - 1;   // hi = n-1 means "never within 0..n-2"
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((double)(prefix[mid + 1] - prefix[start]) >= mean_small) hi = mid; else lo = mid + 1;
    }
    const int i = i1 < lo ? i1 : lo;
    if (i > n - 2) break;
    const int64_t cur = prefix[i + 1] - prefix[start];
    const bool big_i = (double)counts[i] >= mean;
    if (big_i || cur >= min_data_in_bin) {
      out[nb++] = bound_between(distinct[i], distinct[i + 1]);
      start = i + 1;
      if (nb >= max_bin - 1) break;
      if (!big_i) {
        --rest_bins;
        rest_cnt = total_small - small_pref[i + 1];
        mean_small = rest_bins > 0 ? (double)rest_cnt / (double)rest_bins : (double)rest_cnt + 1.0;
        if (mean_small < (double)min_data_in_bin) mean_small = (double)min_data_in_bin;
      }
    }
    pos = i + 1;
  }
  out[nb++] = inf;
  return nb;
}

// Branch-free-ish lower bound: first i with v <= upper[i]; upper[nv-1] = +inf.
LGA_HD int search_bin(const double* upper, int nv, double v) {
  int lo = 0, hi = nv - 1;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (v <= upper[mid]) hi = mid; else lo = mid + 1;
  }
  return lo;
}

}  // namespace lga

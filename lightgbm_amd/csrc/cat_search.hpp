// Split search of a categorical feature over SETS of categories (Fisher 1958; the LightGBM treatment):
// the categories with enough rows are sorted by sum_gradient / (sum_hessian + cat_smooth), and the prefixes
// of that order - from either end, at most max_cat_threshold categories long - are the candidate left
// children, valued with an extra cat_l2 of L2 regularisation.  A candidate is only taken every
// min_data_per_group rows, so that a split never hinges on a tiny category.
//
// Runs on the host for both engines, from the node's finished histogram row of the feature: the sort is a
// few hundred elements per (node, feature), and one implementation keeps the engines' trees identical.
#pragma once
#include "common.hpp"
#include <algorithm>
#include <cstdint>
#include <limits>
#include <vector>

namespace lga {

struct CatParams {
  double cat_smooth = 10.0;
  double cat_l2 = 10.0;
  int max_cat_threshold = 32;
  int min_data_per_group = 100;
};

struct CatSetSplit {
  double gain = -std::numeric_limits<double>::infinity();
  int feature = -1;
  uint64_t bins[kCatSetWords] = {0, 0, 0, 0};
  int num_cat = 0;
  int64_t lg = 0, lh = 0;
};

// g, h: the feature's histogram row; bins 0 .. value_bins-1 are categories.  Improves *best in place (a
// candidate must beat it strictly).
inline void categorical_set_search(const int64_t* g, const int64_t* h, int value_bins, int feature, const LeafScanCtx& ctx,
                                   const SplitParams& sp, const CatParams& cp, CatSetSplit* best) {
  SplitParams sp_cat = sp;
  sp_cat.lambda_l2 += cp.cat_l2;
  std::vector<int> order;
  std::vector<double> ctr((size_t)value_bins, 0.0);
  for (int b = 0; b < value_bins; ++b) {
    if ((double)est_count(h[b], ctx) < cp.cat_smooth) continue;
    ctr[(size_t)b] = ((double)g[b] * ctx.gunit) / ((double)h[b] * ctx.hunit + cp.cat_smooth);
    order.push_back(b);
  }
  const int used = (int)order.size();
  if (used < 2) return;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ctr[(size_t)a] < ctr[(size_t)b]; });
  const int max_num_cat = std::min(cp.max_cat_threshold, (used + 1) / 2);
  for (int dir = 1; dir >= -1; dir -= 2) {
    int pos = dir == 1 ? 0 : used - 1;
    int64_t lg = 0, lh = 0, group_cnt = 0;
    for (int i = 0; i < used && i < max_num_cat; ++i, pos += dir) {
      const int b = order[(size_t)pos];
      lg += g[b];
      lh += h[b];
      group_cnt += est_count(h[b], ctx);
      const int64_t lc = est_count(lh, ctx), rc = ctx.total_cnt - lc;
      if (lc < sp.min_data_in_leaf || (double)lh * ctx.hunit < sp.min_sum_hessian_in_leaf) continue;
      if (rc < sp.min_data_in_leaf || rc < cp.min_data_per_group) break;
      if ((double)(ctx.total_h - lh) * ctx.hunit < sp.min_sum_hessian_in_leaf) break;
      if (group_cnt < cp.min_data_per_group) continue;
      group_cnt = 0;
      const double gain = adjust_feature_gain(split_gain(lg, lh, ctx, sp_cat, 0), feature, ctx.total_cnt, sp);
      if (!(gain > best->gain)) continue;
      best->gain = gain;
      best->feature = feature;
      best->lg = lg;
      best->lh = lh;
      best->num_cat = i + 1;
      for (int w = 0; w < kCatSetWords; ++w) best->bins[w] = 0;
      int p2 = dir == 1 ? 0 : used - 1;
      for (int j = 0; j <= i; ++j, p2 += dir) {
        const int bb = order[(size_t)p2];
        best->bins[bb >> 6] |= 1ull << (bb & 63);
      }
    }
  }
}

}  // namespace lga

// Decision tree in structure-of-arrays form with the LightGBM node numbering:
// the s-th split creates internal node s; its left child keeps the split
// leaf's index, its right child becomes leaf s+1; a child reference < 0
// encodes leaf ~child.  Serialises to / parses from the "Tree=i" block of the
// LightGBM text model format.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lga {

// decision_type bits: bit0 categorical, bit1 default_left, bits 2-3 missing type (0 none, 1 zero, 2 NaN)
constexpr int8_t kCategoricalMask = 1;
constexpr int8_t kDefaultLeftMask = 2;

struct Tree {
  int num_leaves = 1;
  // per internal node (num_leaves-1)
  std::vector<int32_t> split_feature;        // real feature index
  std::vector<int32_t> split_feature_inner;  // index among used features (not serialised)
  std::vector<double> split_gain;
  std::vector<double> threshold;             // real-valued: x <= threshold goes left
  std::vector<int32_t> threshold_bin;        // bin-valued (training-time traversal; not serialised)
  std::vector<int8_t> decision_type;
  std::vector<int32_t> left_child, right_child;
  std::vector<double> internal_value, internal_weight;
  std::vector<int64_t> internal_count;
  // per leaf (num_leaves)
  std::vector<double> leaf_value, leaf_weight;
  std::vector<int64_t> leaf_count;
  std::vector<int32_t> leaf_depth;           // not serialised
  std::vector<int32_t> leaf_parent;          // not serialised
  double shrinkage = 1.0;
  // Categorical splits (decision_type bit 0): the node's `threshold` is an index k into cat_boundaries;
  // words cat_threshold[cat_boundaries[k] .. cat_boundaries[k+1]) form a bitset over category values - a row
  // goes left when its (non-negative, integer) value has its bit set.  Splits grown here send one category left.
  std::vector<int32_t> cat_boundaries;   // starts with 0 once a categorical split exists
  std::vector<uint32_t> cat_threshold;
  int num_cat() const { return cat_boundaries.empty() ? 0 : (int)cat_boundaries.size() - 1; }
  // The same sets over the feature's BINS, 4 words per categorical split k (training-time traversal; not
  // serialised).  A split on a set of several categories has threshold_bin -1.
  std::vector<uint64_t> cat_bins;
  bool bin_in_set(int node, int b) const {
    const size_t k = (size_t)threshold[(size_t)node];
    return (cat_bins[4 * k + (size_t)(b >> 6)] >> (b & 63)) & 1ull;
  }

  Tree() { reset(1); }
  void reset(int max_leaves);

  // Split `leaf` (LightGBM numbering); returns the new right leaf's index.
  int split(int leaf, int feature_real, int feature_inner, int threshold_bin, double threshold, bool default_left,
            int missing_type, double gain, double left_value, double right_value, int64_t left_cnt,
            int64_t right_cnt, double left_weight, double right_weight);

  // The categorical form: rows whose bin == category_bin (value == category) go left, everything else right.
  int split_categorical(int leaf, int feature_real, int feature_inner, int category_bin, int64_t category, double gain,
                        double left_value, double right_value, int64_t left_cnt, int64_t right_cnt, double left_weight,
                        double right_weight);

  // Rows whose bin is in `bins` (256 bits; their values are `categories`) go left.
  int split_categorical_set(int leaf, int feature_real, int feature_inner, const uint64_t* bins,
                            const std::vector<int64_t>& categories, double gain, double left_value, double right_value,
                            int64_t left_cnt, int64_t right_cnt, double left_weight, double right_weight);

  void apply_shrinkage(double rate);
  void add_bias(double v);
  void set_leaf_value(int leaf, double v) { leaf_value[(size_t)leaf] = v; }

  // Decision rule on raw values (shared with the device predictor).
  static inline bool go_left(double fval, double thr, int8_t dtype) {
    int missing_type = (dtype >> 2) & 3;
    if (fval != fval && missing_type != 2) fval = 0.0;
    if ((missing_type == 1 && (fval > -1e-35 && fval < 1e-35)) || (missing_type == 2 && fval != fval))
      return (dtype & kDefaultLeftMask) != 0;
    return fval <= thr;
  }
  // Decision of internal node `node` on a raw feature value (numerical or categorical).
  inline bool decide(int node, double fval) const {
    const int8_t dt = decision_type[(size_t)node];
    if (dt & kCategoricalMask) {
      if (!(fval >= 0.0)) return false;                       // NaN / negative codes: right
      const int64_t c = (int64_t)fval;
      const int k = (int)threshold[(size_t)node];
      const int64_t words = cat_boundaries[(size_t)k + 1] - cat_boundaries[(size_t)k];
      if ((c >> 5) >= words) return false;
      return (cat_threshold[(size_t)cat_boundaries[(size_t)k] + (size_t)(c >> 5)] >> (c & 31)) & 1u;
    }
    return go_left(fval, threshold[(size_t)node], dt);
  }
  // Categories of a categorical node (ascending).
  std::vector<int64_t> node_categories(int node) const;
  int predict_leaf(const double* row) const;
  double predict(const double* row) const { return leaf_value[(size_t)predict_leaf(row)]; }
  // Adds this tree's per-feature contributions (path-dependent SHAP values weighted by the training row
  // counts of the nodes) for one row to phi[0 .. n_features); phi[n_features] receives the tree's expected
  // value.  The entries of one row sum to predict(row).
  void add_contributions(const double* row, int n_features, double* phi) const;
  double expected_value() const;
  // Traversal over binned features: bins(inner_feature) -> bin of this row.
  template <typename BinGetter>
  int predict_leaf_binned(BinGetter bins, const int32_t* missing_bin_of_inner) const {
    if (num_leaves <= 1) return 0;
    int node = 0;
    while (node >= 0) {
      int f = split_feature_inner[(size_t)node];
      int b = bins(f);
      bool left;
      if (decision_type[(size_t)node] & kCategoricalMask) left = bin_in_set(node, b);
      else if (b == missing_bin_of_inner[f]) left = (decision_type[(size_t)node] & kDefaultLeftMask) != 0;
      else left = b <= threshold_bin[(size_t)node];
      node = left ? left_child[(size_t)node] : right_child[(size_t)node];
    }
    return ~node;
  }

  std::string to_string(int index) const;            // "Tree=<index>\n..." block incl. trailing blank line
  // Parses one block (text between "Tree=i" and the next blank line).
  static Tree from_string(const std::string& block);
  std::string to_json(int index) const;
  int max_depth() const;
};

}  // namespace lga

// Parameter registry: one table (name, aliases, type, default, range) drives
// alias resolution, validation, defaults and the "parameters:" block of the
// text model.  Mirrors the flat named-list parameter style the reference
// scripts use (r/gridsearchCV.R:55, 94-100, 108: `params` list + extra
// named args + a data.frame row all end up in one namespace).
#pragma once
#include <string>
#include <vector>
#include <map>
#include <cstdint>

namespace lga {

struct Config {
  // core
  std::string objective = "regression";
  std::string boosting = "gbdt";
  std::string tree_learner = "serial";
  int top_k = 20;            // voting-parallel learner: features every rank votes for per leaf (2 x top_k histograms are summed)
  std::string device_type = "cpu";
  int num_iterations = 100;
  double learning_rate = 0.1;
  int num_leaves = 31;
  int num_threads = 0;
  int seed = 0;
  bool deterministic = true;
  // learning control
  int max_depth = -1;
  int min_data_in_leaf = 20;
  double min_sum_hessian_in_leaf = 1e-3;
  double bagging_fraction = 1.0;
  double pos_bagging_fraction = 1.0;   // balanced bagging (binary): share of the positive / negative rows kept
  double neg_bagging_fraction = 1.0;
  int bagging_freq = 0;
  int bagging_seed = 3;
  std::vector<int> categorical_feature;    // column indices holding integer category codes
  // categorical features with more than max_cat_to_onehot categories are split on category SETS (cat_search.hpp)
  int max_cat_to_onehot = 4;
  double cat_smooth = 10.0;
  double cat_l2 = 10.0;
  int max_cat_threshold = 32;
  int min_data_per_group = 100;
  std::vector<int> monotone_constraints;   // per feature of the raw matrix: -1 / 0 / +1
  // dart
  double drop_rate = 0.1;
  int max_drop = 50;
  double skip_drop = 0.5;
  bool uniform_drop = false;
  bool xgboost_dart_mode = false;
  int drop_seed = 4;
  std::string data_sample_strategy = "bagging";   // bagging | goss
  double top_rate = 0.2;     // goss: share of rows kept for their large gradients
  double other_rate = 0.1;   // goss: share of rows sampled from the rest
  double feature_fraction = 1.0;
  double feature_fraction_bynode = 1.0;
  bool extra_trees = false;    // extremely randomised trees: one random threshold per feature and node
  int extra_seed = 6;
  int feature_fraction_seed = 2;
  int early_stopping_round = 0;
  bool first_metric_only = false;
  double max_delta_step = 0.0;
  double lambda_l1 = 0.0;
  double lambda_l2 = 0.0;
  double min_gain_to_split = 0.0;
  std::vector<double> feature_contri;                 // per column: factor on the gains of splits on it
  double cegb_tradeoff = 1.0;                         // cost-effective gradient boosting: weight of the penalties
  double cegb_penalty_split = 0.0;                    // ... per row of a leaf that is split
  std::vector<double> cegb_penalty_feature_coupled;   // ... per column, paid once, when the model first uses it
  std::vector<double> cegb_penalty_feature_lazy;      // accepted for its name; per-row lazy costs are not implemented
  // Gradients enter the histograms as fixed-point integers (exact, order-independent sums).  true: 16 bits below
  // the iteration's largest |gradient| - what the device's packed LDS atomics hold; false (CPU engine only): 36
  // bits, finer than the float32 gradients themselves for everything within 2^-12 of the largest one.
  bool use_quantized_grad = true;
  // After a tree is grown its leaf values are recomputed from 36-bit sums of the float gradients of the leaf's
  // rows, so the quantisation only ever decides WHERE a tree splits, never what it outputs.
  bool quant_train_renew_leaf = true;
  double path_smooth = 0.0;                // > 0: pull a child's output towards its parent's (less with more rows)
  std::string interaction_constraints;     // "[0,1,2],[2,3]": features that may appear together on a branch
  int verbosity = 1;
  // dataset
  int max_bin = 255;
  std::vector<int> max_bin_by_feature;   // per column of the raw matrix; empty: max_bin everywhere
  int min_data_in_bin = 3;
  int bin_construct_sample_cnt = 200000;
  int data_random_seed = 1;
  bool use_missing = true;
  bool zero_as_missing = false;
  bool feature_pre_filter = true;
  // objective
  bool boost_from_average = true;
  double sigmoid = 1.0;
  int num_class = 1;
  bool is_unbalance = false;
  double scale_pos_weight = 1.0;
  double alpha = 0.9;          // huber / quantile
  double fair_c = 1.0;
  double poisson_max_delta_step = 0.7;
  double tweedie_variance_power = 1.5;
  // ranking
  int lambdarank_truncation_level = 30;   // pairs need one document among the best-scored this many
  bool lambdarank_norm = true;
  std::vector<double> label_gain;         // gain per relevance grade; empty: 2^grade - 1
  int objective_seed = 5;                 // rank_xendcg's per-document noise
  std::vector<int> eval_at;               // cut-offs of ndcg / map; empty: 1..5
  int multi_error_top_k = 1;              // multi_error: a row counts as right while fewer than this many classes score at least as high as the true one
  std::vector<double> auc_mu_weights;     // auc_mu: num_class x num_class misclassification weights, row-major, zero diagonal; empty: all ones
  // metric
  std::vector<std::string> metric;   // empty: derived from the objective
  int metric_freq = 1;
  bool is_provide_training_metric = false;
  // io
  int snapshot_freq = -1;
  std::string output_model = "LightGBM_model.txt";
  std::string input_model = "";
  std::string forcedsplits_filename = "";   // JSON file: splits every tree starts with ({"feature", "threshold", "left", "right"} nested)
  std::string label_column = "0";   // text data files: index of the label column
  bool header = false;              // text data files: the first line holds column names
  // distributed
  int num_machines = 1;
  // device
  int gpu_device_id = 0;
  int max_batch_leaves = 0;   // 0: as many leaves per growth round as the budget allows
  int whole_tree_max_rows = 65536;   // device: bags up to this many rows are grown by the one-workgroup whole-tree kernel (0: never)
  bool profile = false;
  bool locality_order = true;   // device engine, large single-tile data: private copies of the bins in a locality order of the rows
  int iteration_offset = 0;       // number the iterations from here: the random streams (bagging, feature
                                  // sampling) of iteration k are the same wherever that iteration is trained
  bool device_bin_find = true;   // device-resident raw matrices: find bin boundaries on the GPU

  // accepted for scripts written against upstream LightGBM; they select between implementations this engine has
  // one of (histograms are always built row-wise over device-resident rows, gradients quantised to 16 bits with
  // round-to-nearest, dense storage), so they change nothing here
  bool force_col_wise = false, force_row_wise = false;
  double histogram_pool_size = -1.0;
  int gpu_platform_id = -1;
  bool gpu_use_dp = false;
  int num_gpu = 1;
  bool two_round = false;
  bool precise_float_parser = false;
  bool enable_bundle = true;
  int num_grad_quant_bins = 4;
  bool stochastic_rounding = false;       // 16-bit histograms: unbiased floor(s + u) rounding instead of round-to-nearest
  // prediction (read by Booster.predict from the booster's parameters; the trainers ignore them)
  std::string forcedbins_filename;   // JSON: [{"feature": 0, "bin_upper_bound": [0.3, 0.4]}, ...] - bounds those features must have
  bool reg_sqrt = false;       // objective=regression: fit sign(y) sqrt|y|, square the predictions back
  bool pred_early_stop = false;
  int pred_early_stop_freq = 10;
  double pred_early_stop_margin = 10.0;
  std::string monotone_constraints_method = "basic";   // the one method implemented; the stricter upstream methods fall back to it
  // behaviour
  double early_stopping_min_delta = 0.0;      // an improvement counts from this much on (train / cv early stopping)
  int saved_feature_importance_type = 0;      // feature_importances block of saved models: 0 split counts, 1 total gains

  // Names of parameters that were explicitly set (after alias resolution).
  std::vector<std::string> explicit_keys;
  // Unknown keys seen while parsing (reported as warnings, not errors).
  std::vector<std::string> unknown_keys;

  // Parse "k=v k2=v2" (whitespace or newline separated); later keys override.
  // Throws std::runtime_error on a malformed or out-of-range value.
  void set(const std::string& params);
  void set(const std::map<std::string, std::string>& kv);
  void check();   // cross-parameter validation (rf needs sub-sampling, ...)

  // "[name: value]" lines, in table order, for the model file.
  std::string to_model_string() const;
  // Canonical objective name ("regression", "binary") and its metric default.
  static std::string canonical_objective(const std::string& name);
  static std::string canonical_metric(const std::string& name);
  std::vector<std::string> resolved_metrics() const;

  // Alias handling, usable from the bindings.
  static std::string resolve_alias(const std::string& name);  // "" if unknown
  static std::string alias_table_json();
};

}  // namespace lga

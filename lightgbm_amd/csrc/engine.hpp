// Execution engine of one booster: owns scores, gradients, the current bag
// and the tree backend for one training dataset (+ validation datasets).
// The boosting driver (gbdt.cpp) is written against this interface; the host
// (OpenMP) and HIP engines implement it with the same numerics.
#pragma once
#include "backend.hpp"
#include "config.hpp"
#include "dataset.hpp"
#include "grower.hpp"
#include "objective.hpp"
#include "ranking.hpp"
#include "tree.hpp"
#include <memory>
#include <string>
#include <vector>

namespace lga {

struct EvalResult {
  std::string name;
  double value;
  bool higher_better;
};

// A chunk of boosting iterations run by the engine itself, without the driver in between (device engine, small
// data): what the driver decides per iteration, and what comes back.
struct ChunkIterPlan {
  std::vector<FeatureMeta> meta;   // the iteration's feature_fraction mask applied
  int64_t tree_index = 0;
  bool redraw = false;             // draw a new bag before the tree, from the random stream `bag_iter`
  uint64_t bag_iter = 0;
};
struct ChunkPlan {
  std::vector<ChunkIterPlan> iters;
  ObjectiveSpec obj;
  int bagging = 0;                 // 0: every row, 1: bagging_fraction, 3: by label (bag_fraction for positives, neg_fraction)
  double bag_fraction = 1.0, neg_fraction = 1.0;
  uint64_t bag_seed = 0;
  double learning_rate = 0.1;
  bool renew = true;               // leaf values from the 36-bit sums
  bool eval_valid = false;         // evaluate validation set 0 after every iteration
  std::vector<int32_t> metrics;    // pointwise metrics (at most 4)
};
struct ChunkTree {
  Tree tree;                       // leaf values as the search left them (not renewed, not shrunk)
  std::vector<int64_t> leaf_sums;  // [leaf][2]: 36-bit sums of g and h over the leaf's in-bag rows (ChunkPlan::renew)
  int kg = 0, kh = 0;              // their exponents
  std::vector<double> metric_sums; // [metric][2]: sum of weighted losses, sum of weights
};

class Engine {
 public:
  virtual ~Engine() {}
  virtual const char* name() const = 0;
  // Called at the top of any worker thread (OpenMP) that is about to drive this engine: a device engine makes its
  // GPU the thread's current device - a new thread starts on device 0, where its allocations and launches would
  // otherwise land, whatever device the engine was built on.
  virtual void bind_thread() {}
  // which: -1 training data, >=0 validation set index
  virtual int add_valid(std::shared_ptr<Dataset> ds) = 0;
  virtual int num_valid() const = 0;
  virtual int64_t num_rows(int which) const = 0;

  // Sum of weight*label and sum of weight over the (global) training rows; sqrt_labels: of the labels that
  // reg_sqrt fits (objective.hpp: sqrt_label).
  virtual void label_stats(double* sum_wy, double* sum_w, double* sum_w_pos, bool sqrt_labels) = 0;
  // scores = dataset init_score (or 0) + v, for training and validation data.
  virtual void reset_scores(double v) = 0;
  virtual void scale_scores(int which, double factor) = 0;
  virtual void add_constant(int which, double v) = 0;

  // Multiclass: the engine keeps num_class (Config::num_class) score and gradient vectors per dataset.
  // select_class(k) makes class k's vectors the ones reset_scores / scale_scores / add_constant /
  // set_gradients / train_tree / add_* operate on; compute_gradients fills all classes at once, eval and
  // get_scores ([class][row]) see all of them.
  virtual void select_class(int k) = 0;
  virtual void compute_gradients(const ObjectiveSpec& obj) = 0;
  // The objective whose gradients every iteration starts with (null: none such - ranking, custom gradients, rf).  An
  // engine may then take the next iteration's gradients in the pass that adds a finished tree to the training
  // scores; compute_gradients finds them done unless the scores have changed since.  invalidate_... after a change
  // of the objective's parameters.
  virtual void set_iteration_objective(const ObjectiveSpec* obj) { (void)obj; }
  virtual void invalidate_precomputed_gradients() {}
  // Weighted number of training rows per class label (summed over ranks).
  virtual void class_weights(int num_class, std::vector<double>* out) = 0;
  // Ranking gradients (ranking.hpp) taken by the engine itself from its current scores; bounds are the
  // training dataset's query boundaries.  false: the engine leaves them to the caller (who reads the scores,
  // computes on the host and calls set_gradients) - the host engine, or a query too large for the device kernels.
  virtual bool ranking_gradients(const RankingSpec& spec, const std::vector<int64_t>& bounds,
                                 const std::vector<double>& inv_max_dcg, uint64_t iter) {
    (void)spec; (void)bounds; (void)inv_max_dcg; (void)iter;
    return false;
  }
  // ndcg@k and map@k of every query of a dataset (which: -1 training, >= 0 validation) at the engine's current
  // scores: values[q * 2 * at.size() + k] = ndcg@at[k], values[... + at.size() + k] = map@at[k].  false: the
  // engine leaves the metrics to the caller (ranking.cpp on the host).
  virtual bool ranking_metric_values(int which, const RankingSpec& spec, const std::vector<int64_t>& bounds,
                                     const std::vector<int>& at, std::vector<double>* values) {
    (void)which; (void)spec; (void)bounds; (void)at; (void)values;
    return false;
  }
  // Gradients supplied by the caller, ordered like the training dataset's rows.
  virtual void set_gradients(const float* g, const float* h, int64_t n) = 0;
  // Draw a new bag (every row when fraction >= 1).  neg_fraction >= 0: balanced bagging - rows with a
  // positive label stay with probability `fraction`, the others with `neg_fraction`.
  virtual void resample(double fraction, uint64_t seed, uint64_t iter, double neg_fraction = -1.0) = 0;

  // Gradient-based one-side sampling on the gradients just computed: a row's weight is the sum over classes
  // of |g * h|; the top_k heaviest rows stay (ties with the k-th included), every other row stays with
  // probability other_k / (n - top_k) and has its gradients and hessians multiplied by (n - top_k) / other_k.
  // The kept rows become the bag.  n, top_k = max(1, top_rate * n), other_k = max(1, other_rate * n) are local.
  virtual void goss_sample(double top_rate, double other_rate, uint64_t seed, uint64_t iter) = 0;

  // Grow a tree on the current gradients/bag.  Leaf values are unshrunk.
  // Index of the iteration the next train_tree() belongs to (seeds per-node feature sampling).
  virtual void set_tree_index(int64_t i) = 0;
  // Split-search parameters changed between iterations (TreeGrower::update_split_config).
  virtual void update_split_config(const Config& c) = 0;
  virtual Tree train_tree(const std::vector<FeatureMeta>& meta) = 0;
  // train_tree in two steps, so that the trees of several boosters - the folds of a cross-validation - can be
  // started before any of them is waited for.  train_tree_begin returns true when the engine has PREPARED a
  // device build that still has to be launched: launch_prepared_trees starts the prepared builds of a group of
  // engines with as few launches as possible.  train_tree_finish returns the tree (and grows it itself when
  // nothing was prepared).
  // How many boosters are advanced together with this one (Booster::update_group); 1 = alone.  A device engine
  // grows the trees of a small booster with ONE workgroup only in company: alone, batched growth rounds over the
  // whole device are faster.
  virtual void set_group_size(int n) { (void)n; }
  virtual bool train_tree_begin(const std::vector<FeatureMeta>& meta) { pending_meta_ = meta; return false; }
  virtual Tree train_tree_finish() { return train_tree(pending_meta_); }
  // Called on one engine of the group with all engines whose train_tree_begin returned true (itself included).
  virtual void launch_prepared_trees(Engine* const* group, int n) { (void)group; (void)n; }
  // A chunk of iterations on the engine (ChunkPlan): gradients, bag, tree, leaf values, score updates of the
  // training data and of validation set 0, metric sums - per iteration, with no driver in between.
  // train_chunk_begin prepares it (false: this engine / configuration cannot, nothing has happened);
  // launch_prepared_chunks starts the prepared chunks of a group of engines; train_chunk_finish waits and
  // reports the iterations' trees.  The engine's scores then stand after the chunk's last iteration.
  virtual bool train_chunk_begin(const ChunkPlan& plan) { (void)plan; return false; }
  virtual void launch_prepared_chunks(Engine* const* group, int n) { (void)group; (void)n; }
  virtual void train_chunk_finish(std::vector<ChunkTree>* out) { (void)out; }
  // Add `tree` (final leaf values) to the training scores, using the row
  // ranges of the tree just grown for in-bag rows and traversal for the rest.
  virtual void add_trained_tree_to_train_scores(const Tree& tree) = 0;
  // Objectives with percentile leaf outputs (percentile.hpp): for every leaf of the tree just grown, the
  // alpha-percentile of label - score over the leaf's in-bag rows, weighted by the row weights (times
  // 1 / max(1, |label|) with mape_scale).  values arrives holding the tree's current leaf values, which
  // leaves without rows keep.
  virtual void leaf_residual_percentiles(const Tree& tree, double alpha, bool mape_scale,
                                         std::vector<double>* values) = 0;
  // Exact gradient statistics of the leaves of the tree just grown: per leaf the sums over its in-bag LOCAL rows
  // of quantize_wide(g, kg) and quantize_wide(h, kh) (common.hpp), with kg / kh = the tree's quantisation
  // exponents + kWideExtraBits.  sums: [leaf][2] = (sum g, sum h).
  virtual void leaf_gradient_sums(const Tree& tree, std::vector<int64_t>* sums, int* kg, int* kh) = 0;
  // true: the trees' leaf values already come from those very sums (full-precision histograms: a node's histogram
  // totals ARE the 36-bit sums over its rows), so renewing them would compute the same numbers again
  virtual bool leaf_values_from_exact_sums() const { return false; }
  // Add (sign=+1) or remove (sign=-1) a tree's output by traversal of the binned rows.
  virtual void add_tree_by_traversal(int which, const Tree& tree, double sign) = 0;

  virtual std::vector<EvalResult> eval(int which, const std::vector<int32_t>& metrics, const ObjectiveSpec& obj) = 0;
  virtual void get_scores(int which, double* out) = 0;   // dataset-local row order
  virtual void get_gradients(float* g, float* h) = 0;    // dataset-local row order
  // In-place sum of host values over the ranks of a distributed run (no-op on a single rank).
  virtual void sum_over_ranks(double* v, size_t n) = 0;
  virtual void sum_over_ranks_i64(int64_t* v, size_t n) = 0;
  virtual std::string profile_json() const { return "{}"; }
  virtual GrowStats grow_stats() const = 0;

 protected:
  std::vector<FeatureMeta> pending_meta_;
};

}  // namespace lga

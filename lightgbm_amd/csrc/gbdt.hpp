// Boosting drivers: gradient boosting (`lgb.train`, r/gridsearchCV.R:57-61,
// 134-138) and the random-forest / bagging mode that
// bagging_boosting.ipynb:10-13 contrasts it with.  One Booster = one model +
// (when training) one execution engine.
#pragma once
#include "config.hpp"
#include "dataset.hpp"
#include "engine.hpp"
#include "ranking.hpp"
#include "tree.hpp"
#include <functional>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace lga {

using EngineFactory = std::function<std::unique_ptr<Engine>(std::shared_ptr<Dataset>, const Config&)>;

enum PredictType : int { kPredictNormal = 0, kPredictRaw = 1, kPredictLeaf = 2, kPredictContrib = 3 };

class Booster {
 public:
  // Training booster.
  Booster(std::shared_ptr<Dataset> train, const std::string& params, const EngineFactory& factory);
  // Prediction-only booster from a LightGBM text model.
  explicit Booster(const std::string& model_text);

  void add_valid(std::shared_ptr<Dataset> ds, const std::string& name);
  // One boosting iteration; returns true when no further split was possible
  // (the iteration's tree is then discarded, except for the very first one).
  bool update_one_iter();
  // One iteration in two steps (single-output boosters; others run the whole iteration in update_finish):
  // update_begin takes gradients and the bag and prepares the tree - true: a device build is waiting for
  // launch_prepared_trees - update_finish waits for the tree and applies it.  update_group runs one iteration of
  // several boosters with their prepared builds started together; finished[i] = boosters[i] could not split.
  bool update_begin();
  bool update_finish();
  static void update_group(Booster* const* boosters, int n, int* finished);
  // n_iters iterations of a group of boosters in lockstep, each followed - eval_valid - by the evaluation of every
  // booster's first validation set: values[(iteration * n + booster) * stride + metric].  When every booster and
  // its engine can, the iterations run as ONE chunk on the engines (Engine::train_chunk_begin) without the driver
  // in between; otherwise iteration by iteration (update_group + eval).  The loop ends after an iteration in which
  // no booster found a split; finished[iteration * n + b] = booster b found none in that iteration.  Returns the
  // iterations run.
  static int update_group_chunk(Booster* const* boosters, int n, int n_iters, bool eval_valid, double* values, int stride,
                                int* finished);
  bool update_one_iter_custom(const float* grad, const float* hess, int64_t n);
  void rollback_one_iter();
  void reset_parameter(const std::string& params);
  // Keeps every tree's structure and refits its leaf values to other data: a leaf becomes
  // decay * old + (1 - decay) * new, new being the Newton step of the leaf's rows at the scores of the
  // iterations before it (times the tree's shrinkage).  lambda_l1 / lambda_l2 / max_delta_step come from
  // `params` over this booster's configuration.  weight may be null.
  void refit(const void* data, int dtype, int64_t nrow, int ncol, bool row_major, const float* label, const float* weight,
             double decay, const std::string& params);
  // Put `first`'s trees in front of this booster's (continued training from init_model).
  void prepend_trees(const Booster& first) {
    if (first.num_class_ != num_class_) throw std::runtime_error("init_model has a different number of classes");
    trees_.insert(trees_.begin(), first.trees_.begin(), first.trees_.end());
  }

  // A multiclass iteration holds num_class trees: tree t belongs to iteration t / K and class t % K.
  int num_class() const { return num_class_; }
  int current_iteration() const { return (int)trees_.size() / num_class_; }
  int num_trees() const { return (int)trees_.size(); }
  int num_features() const { return max_feature_idx_ + 1; }
  int num_valid() const { return engine_ ? engine_->num_valid() : 0; }
  const std::vector<std::string>& valid_names() const { return valid_names_; }
  std::vector<EvalResult> eval(int which);
  // Overwrites one leaf value of the model (scores already accumulated during training stay as they are).
  void set_leaf_value(int tree, int leaf, double v) {
    if (tree < 0 || tree >= (int)trees_.size() || leaf < 0 || leaf >= trees_[(size_t)tree].num_leaves)
      throw std::runtime_error("set_leaf_value: tree or leaf out of range");
    trees_[(size_t)tree].set_leaf_value(leaf, v);
  }
  // Puts iterations [start, end) (end <= 0: to the last one) into a seeded random order.
  void shuffle_iterations(int start, int end, uint64_t seed);
  // (name, higher is better) of the values eval() returns, in its order.
  std::vector<std::pair<std::string, bool>> eval_names() const;
  // Current (transformed unless raw) predictions for training (-1) / validation data.
  void inner_predict(int which, bool raw, double* out);
  int64_t inner_num_rows(int which) const { return engine_->num_rows(which); }
  void get_gradients(float* g, float* h) { engine_->get_gradients(g, h); }

  // Dense prediction on the host.  out per row: num_class values (normal / raw), used_iterations * num_class
  // leaf indices as doubles (leaf), or num_class * (ncol + 1) contributions.
  void predict(const void* data, int dtype, int64_t nrow, int ncol, bool row_major, int start_iter, int num_iter,
               int type, double* out) const;
  int used_iterations(int start_iter, int num_iter) const;

  std::string save_model_to_string(int start_iter, int num_iter, int importance_type) const;
  std::string dump_json(int start_iter, int num_iter) const;
  std::vector<double> feature_importance(int num_iter, int importance_type) const;  // 0 split, 1 gain
  const std::vector<std::string>& feature_names() const { return feature_names_; }
  const std::vector<Tree>& trees() const { return trees_; }
  const Config& config() const { return cfg_; }
  const std::string& objective_string() const { return objective_str_; }
  bool average_output() const { return average_output_; }
  ObjectiveSpec objective_spec() const { return obj_; }
  std::string last_profile() const { return engine_ ? engine_->profile_json() : "{}"; }
  GrowStats last_grow_stats() const { return engine_ ? engine_->grow_stats() : GrowStats(); }
  Engine* engine() { return engine_.get(); }
  double init_score_value() const { return init_scores_[0]; }

 private:
  void setup_objective();
  void ensure_scores();
  double boost_from_average();
  std::vector<FeatureMeta> sample_features(int cls);
  bool finish_tree(Tree&& tree, int cls);
  bool finish_iteration(bool all_stumps);
  void transform_row(double* v) const;
  void load_model_from_string(const std::string& text);

  Config cfg_;
  std::shared_ptr<Dataset> train_;
  std::unique_ptr<Engine> engine_;
  std::vector<std::shared_ptr<Dataset>> valid_sets_;
  std::vector<std::string> valid_names_;
  std::vector<int32_t> metrics_;
  ObjectiveSpec obj_{};
  std::string objective_str_;
  bool is_rf_ = false;
  bool average_output_ = false;
  bool scores_initialised_ = false;
  // chunks of iterations on the engine
  bool chunk_eligible(bool eval_valid) const;
  bool chunk_begin(int n_iters, bool eval_valid);
  void chunk_fetch(std::vector<char>* stump);      // waits for the chunk; stump[j] = iteration j found no split
  void chunk_commit(int n_keep, bool eval_valid, double* values, int stride, int n_group, int index);
  std::vector<ChunkTree> chunk_trees_;
  void renew_from_sums(Tree& tree, const std::vector<int64_t>& sums, int kg, int kh) const;
  std::vector<FeatureMeta> sample_features_at(int cls, int iter);
  bool two_step_ = false;          // between update_begin and update_finish of a two-step iteration
  std::vector<double> init_scores_ = {0.0};   // per class
  int num_class_ = 1;
  int iter_ = 0;                 // iterations attempted (drives bagging / feature sampling streams)
  bool bag_drawn_ = false;
  void draw_bag();
  void sample_rows();
  // dart: trees dropped for the running iteration, and the weights bookkeeping (gbdt.cpp)
  void dart_drop();
  void dart_normalize();
  Tree without_bias(int tree_index) const;
  bool is_dart_ = false;
  std::vector<int> dropped_;            // iteration indices
  std::vector<double> tree_weight_;     // per iteration: the factor its trees currently carry
  double sum_tree_weight_ = 0.0;
  double dart_shrinkage_ = 1.0;         // learning rate of the iteration in progress
  void renew_leaf_outputs(Tree& tree);
  void renew_leaf_values_exact(Tree& tree);
  // ranking (ranking.hpp): per-query gradients; eval_host_metrics adds the metrics the engines do not compute
  // themselves (ndcg / map per query, average_precision, auc_mu) from the engine's scores
  struct RankMetric { bool ndcg; int at; };
  bool is_ranking() const { return obj_.kind == kObjLambdarank || obj_.kind == kObjXendcg; }
  void setup_ranking();
  void compute_gradients();
  void eval_host_metrics(int which, std::vector<EvalResult>* out);
  bool eval_auc_mu_ = false;              // multiclass AUC (Kleiman & Page): mean pairwise class separability
  bool eval_average_precision_ = false;   // evaluated here from the engine's scores, like the ranking metrics
  RankingSpec rank_;
  std::vector<RankMetric> rank_metrics_;
  std::set<int> rank_labels_checked_;
  std::vector<double> inv_max_dcg_;
  std::vector<float> rank_label_, rank_weight_;   // training rows, local order
  std::vector<Tree> trees_;
  // model meta
  int max_feature_idx_ = -1;
  int label_index_ = 0;
  std::vector<std::string> feature_names_;
  std::vector<std::string> feature_infos_;
  std::string loaded_parameters_;
};

}  // namespace lga

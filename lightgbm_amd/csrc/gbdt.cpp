#include "gbdt.hpp"
#include "metric.hpp"
#include <algorithm>
#include <cmath>
#include <numeric>
#include <sstream>
#include <stdexcept>

namespace lga {

Booster::Booster(std::shared_ptr<Dataset> train, const std::string& params, const EngineFactory& factory)
    : train_(train) {
  cfg_.set(params);
  if (train_->root_label().empty()) throw std::runtime_error("training data has no label");
  is_rf_ = cfg_.boosting == "rf";
  is_dart_ = cfg_.boosting == "dart";
  average_output_ = is_rf_;
  iter_ = cfg_.iteration_offset;
  setup_objective();
  num_class_ = obj_.num_class;
  init_scores_.assign((size_t)num_class_, 0.0);
  rank_.label_gain = cfg_.label_gain.empty() ? default_label_gain() : cfg_.label_gain;
  for (const auto& m : cfg_.resolved_metrics()) {
    if (m == "ndcg" || m == "map") {
      std::vector<int> at = cfg_.eval_at.empty() ? std::vector<int>{1, 2, 3, 4, 5} : cfg_.eval_at;
      for (int k : at) rank_metrics_.push_back(RankMetric{m == "ndcg", k});
    } else if (m == "average_precision") {
      eval_average_precision_ = true;
    } else if (m == "auc_mu") {
      eval_auc_mu_ = true;
    } else {
      metrics_.push_back(metric_kind_from_name(m));
    }
  }
  engine_ = factory(train_, cfg_);
  if (is_ranking()) setup_ranking();
  // every iteration starts with the gradients of obj_ at the current scores: the engine may take them early
  if (!is_rf_ && !is_dart_ && !is_ranking() && obj_.kind != kObjCustom && num_class_ == 1) engine_->set_iteration_objective(&obj_);
  max_feature_idx_ = train_->num_total_features() - 1;
  feature_names_ = train_->names();
  feature_infos_.clear();
  for (const auto& m : train_->mappers()) feature_infos_.push_back(m.info_string());
}

Booster::Booster(const std::string& model_text) { load_model_from_string(model_text); }

void Booster::setup_objective() {
  std::string o = Config::canonical_objective(cfg_.objective);
  obj_ = ObjectiveSpec{};
  obj_.num_class = 1;
  obj_.sigmoid = cfg_.sigmoid;
  obj_.pos_weight = 1.0;
  obj_.neg_weight = 1.0;
  obj_.alpha = cfg_.alpha;
  obj_.fair_c = cfg_.fair_c;
  obj_.poisson_max_delta = cfg_.poisson_max_delta_step;
  obj_.tweedie_rho = cfg_.tweedie_variance_power;
  obj_.multi_error_top_k = cfg_.multi_error_top_k;
  if (o == "regression") {
    obj_.kind = cfg_.reg_sqrt ? kObjRegressionSqrt : kObjRegression;
    objective_str_ = cfg_.reg_sqrt ? "regression sqrt" : "regression";
  }
  else if (o == "regression_l1") { obj_.kind = kObjRegressionL1; objective_str_ = "regression_l1"; }
  else if (o == "huber") { obj_.kind = kObjHuber; objective_str_ = "huber"; }
  else if (o == "fair") { obj_.kind = kObjFair; objective_str_ = "fair"; }
  else if (o == "poisson") { obj_.kind = kObjPoisson; objective_str_ = "poisson"; }
  else if (o == "quantile") { obj_.kind = kObjQuantile; objective_str_ = "quantile"; }
  else if (o == "mape") { obj_.kind = kObjMape; objective_str_ = "mape"; }
  else if (o == "gamma") { obj_.kind = kObjGamma; objective_str_ = "gamma"; }
  else if (o == "tweedie") { obj_.kind = kObjTweedie; objective_str_ = "tweedie"; }
  else if (o == "binary") {
    obj_.kind = kObjBinary;
    std::ostringstream os;
    os << "binary sigmoid:" << cfg_.sigmoid;
    objective_str_ = os.str();
  } else if (o == "cross_entropy") { obj_.kind = kObjCrossEntropy; objective_str_ = "cross_entropy"; }
  else if (o == "lambdarank") { obj_.kind = kObjLambdarank; objective_str_ = "lambdarank"; }
  else if (o == "rank_xendcg") { obj_.kind = kObjXendcg; objective_str_ = "rank_xendcg"; }
  else if (o == "multiclass" || o == "multiclassova") {
    obj_.kind = o == "multiclass" ? kObjMulticlass : kObjMulticlassOva;
    obj_.num_class = cfg_.num_class;
    std::ostringstream os;
    os << o << " num_class:" << cfg_.num_class;
    if (o == "multiclassova") os << " sigmoid:" << cfg_.sigmoid;
    objective_str_ = os.str();
  } else {
    obj_.kind = kObjCustom;
    obj_.num_class = std::max(1, cfg_.num_class);   // caller-supplied gradients may be multiclass
    objective_str_ = "custom";
  }
}

void Booster::add_valid(std::shared_ptr<Dataset> ds, const std::string& name) {
  if (!engine_) throw std::runtime_error("add_valid on a prediction-only booster");
  if (scores_initialised_) throw std::runtime_error("validation data must be added before the first boosting iteration");
  engine_->add_valid(ds);
  valid_sets_.push_back(ds);
  valid_names_.push_back(name);
}

double Booster::boost_from_average() {
  if (!cfg_.boost_from_average) return 0.0;
  if (!train_->root_init_score().empty()) return 0.0;
  double swy, sw, swp;
  engine_->label_stats(&swy, &sw, &swp, obj_.kind == kObjRegressionSqrt);
  if (!(sw > 0.0)) return 0.0;
  switch (obj_.kind) {
    case kObjRegression:
    case kObjRegressionSqrt:
    case kObjHuber:
    case kObjFair:
    case kObjRegressionL1:   // mean instead of the median: a constant start either way
    case kObjQuantile:
    case kObjMape:
      return swy / sw;
    case kObjGamma:
    case kObjTweedie:
    case kObjPoisson: {
      double m = swy / sw;
      return std::log(m > 1e-10 ? m : 1e-10);
    }
    case kObjBinary: {
      double pos = swp * obj_.pos_weight, neg = (sw - swp) * obj_.neg_weight;
      double p = pos / (pos + neg);
      p = std::min(std::max(p, 1e-15), 1.0 - 1e-15);
      return std::log(p / (1.0 - p)) / obj_.sigmoid;
    }
    case kObjCrossEntropy: {
      double p = std::min(std::max(swy / sw, 1e-15), 1.0 - 1e-15);
      return std::log(p / (1.0 - p));
    }
    default: return 0.0;
  }
}

std::vector<FeatureMeta> Booster::sample_features(int cls) { return sample_features_at(cls, iter_); }

std::vector<FeatureMeta> Booster::sample_features_at(int cls, int iter) {
  std::vector<FeatureMeta> meta = train_->feature_meta();
  int F = (int)meta.size();
  if (!cfg_.monotone_constraints.empty()) {
    if ((int)cfg_.monotone_constraints.size() != train_->num_total_features())
      throw std::runtime_error("monotone_constraints lists " + std::to_string(cfg_.monotone_constraints.size()) +
                               " values, the data has " + std::to_string(train_->num_total_features()) + " features");
    for (int f = 0; f < F; ++f) meta[(size_t)f].monotone = cfg_.monotone_constraints[(size_t)train_->real_feature(f)];
  }
  // categorical features with many categories are searched over category sets; not under feature-parallel
  // learning (a rank only holds its own features' histograms) and not with extra_trees' single random threshold
  if (cfg_.tree_learner != "feature" && !cfg_.extra_trees)
    for (auto& m : meta)
      if (m.categorical && (m.missing_bin >= 0 ? m.num_bin - 1 : m.num_bin) > cfg_.max_cat_to_onehot) m.categorical = kCategoricalSorted;
  double ff = cfg_.feature_fraction;
  if (ff >= 1.0 || F <= 1) return meta;
  int k = std::max(1, (int)(F * ff + 0.5));
  std::vector<int> perm((size_t)F);
  std::iota(perm.begin(), perm.end(), 0);
  for (int j = 0; j < k; ++j) {
    uint64_t r = rng_u64((uint64_t)cfg_.feature_fraction_seed, (uint64_t)iter * (uint64_t)num_class_ + (uint64_t)cls,
                         (uint64_t)j);
    int pick = j + (int)(r % (uint64_t)(F - j));
    std::swap(perm[(size_t)j], perm[(size_t)pick]);
  }
  for (auto& m : meta) m.enabled = 0;
  for (int j = 0; j < k; ++j) meta[(size_t)perm[(size_t)j]].enabled = 1;
  return meta;
}

void Booster::ensure_scores() {
  if (scores_initialised_) return;
  scores_initialised_ = true;
  const int K = num_class_;
  init_scores_.assign((size_t)K, 0.0);
  if (obj_.kind == kObjBinary) {
    double swy, sw, swp;
    engine_->label_stats(&swy, &sw, &swp, false);
    double cp = swp, cn = sw - swp;
    if (cfg_.is_unbalance && cp > 0 && cn > 0) {
      if (cp > cn) { obj_.pos_weight = 1.0; obj_.neg_weight = cp / cn; }
      else { obj_.pos_weight = cn / cp; obj_.neg_weight = 1.0; }
    }
    obj_.pos_weight *= cfg_.scale_pos_weight;
  }
  if (is_multiclass(obj_.kind)) {
    // class priors: log p_k for softmax, logit(p_k) / sigmoid for one-vs-all; also validates the labels
    std::vector<double> cw;
    engine_->class_weights(K, &cw);
    double total = 0.0;
    for (double c : cw) total += c;
    if (cfg_.boost_from_average && train_->root_init_score().empty() && total > 0.0) {
      for (int k = 0; k < K; ++k) {
        double p = std::min(std::max(cw[(size_t)k] / total, 1e-15), 1.0 - 1e-15);
        init_scores_[(size_t)k] = obj_.kind == kObjMulticlass ? std::log(p) : std::log(p / (1.0 - p)) / obj_.sigmoid;
      }
    }
  } else if (obj_.kind != kObjCustom) {
    init_scores_[0] = boost_from_average();
  }
  for (int k = 0; k < K; ++k) {
    engine_->select_class(k);
    engine_->reset_scores(init_scores_[(size_t)k]);
  }
  engine_->select_class(0);
  if (is_rf_) compute_gradients();
}

// ---------------------------------------------------------------------------
// ranking
// ---------------------------------------------------------------------------
void Booster::setup_ranking() {
  if (train_->query_boundaries().empty())
    throw std::runtime_error("objective=" + objective_str_ + " needs query information (Dataset group=)");
  rank_.xendcg = obj_.kind == kObjXendcg;
  rank_.sigmoid = cfg_.sigmoid;
  rank_.norm = cfg_.lambdarank_norm;
  rank_.truncation_level = cfg_.lambdarank_truncation_level;
  rank_.seed = (uint64_t)cfg_.objective_seed;
  rank_label_ = train_->local_label();
  rank_weight_ = train_->local_weight();
  check_rank_labels(rank_label_.data(), (int64_t)rank_label_.size(), rank_.label_gain);
  if (!rank_.xendcg)
    inv_max_dcg_ = inverse_max_dcg(rank_label_.data(), train_->query_boundaries(), rank_.label_gain, rank_.truncation_level);
}

void Booster::compute_gradients() {
  if (!is_ranking()) { engine_->compute_gradients(obj_); return; }
  if (engine_->ranking_gradients(rank_, train_->query_boundaries(), inv_max_dcg_, (uint64_t)iter_)) return;
  const int64_t n = engine_->num_rows(-1);
  std::vector<double> score((size_t)n);
  std::vector<float> g((size_t)n), h((size_t)n);
  engine_->get_scores(-1, score.data());
  ranking_gradients(rank_, score.data(), rank_label_.data(), rank_weight_.empty() ? nullptr : rank_weight_.data(),
                    train_->query_boundaries(), inv_max_dcg_, (uint64_t)iter_, g.data(), h.data());
  engine_->set_gradients(g.data(), h.data(), n);
}

void Booster::eval_host_metrics(int which, std::vector<EvalResult>* out) {
  if (eval_average_precision_) {
    const Dataset* ds = which < 0 ? train_.get() : valid_sets_[(size_t)which].get();
    const int64_t n = engine_->num_rows(which);
    std::vector<double> score((size_t)n * (size_t)num_class_);
    engine_->get_scores(which, score.data());
    const std::vector<float> label = ds->local_label(), weight = ds->local_weight();
    std::vector<std::pair<double, std::pair<float, float>>> items((size_t)n);
    for (int64_t i = 0; i < n; ++i) items[(size_t)i] = {score[(size_t)i], {label[(size_t)i], weight.empty() ? 1.0f : weight[(size_t)i]}};
    double pos = 0.0;
    const double ap = average_precision_host(items, &pos);
    // distributed: the mean of the ranks' values, weighted by their positives
    double b[2] = {ap * pos, pos};
    engine_->sum_over_ranks(b, 2);
    out->push_back(EvalResult{"average_precision", b[1] > 0.0 ? b[0] / b[1] : 1.0, true});
  }
  if (eval_auc_mu_) {
    if (num_class_ < 2) throw std::runtime_error("auc_mu needs a multiclass objective");
    const Dataset* ds = which < 0 ? train_.get() : valid_sets_[(size_t)which].get();
    const int64_t n = engine_->num_rows(which);
    const int K = num_class_;
    std::vector<double> score((size_t)n * (size_t)K);      // [class][row]
    engine_->get_scores(which, score.data());
    const std::vector<float> label = ds->local_label(), weight = ds->local_weight();
    // for every pair of classes: the AUC of telling them apart along (A_i - A_j) . scores, A the matrix of
    // misclassification weights (auc_mu_weights; all ones off the diagonal: the difference of the two raw scores)
    const std::vector<double>& A = cfg_.auc_mu_weights;
    if (!A.empty()) {
      if ((int)A.size() != K * K) throw std::runtime_error("auc_mu_weights needs num_class x num_class values");
      for (int i = 0; i < K; ++i)
        if (A[(size_t)i * (size_t)K + (size_t)i] != 0.0) throw std::runtime_error("auc_mu_weights: the diagonal must be zero");
    }
    double sum = 0.0;
    std::vector<double> v((size_t)K);
    for (int i = 0; i < K; ++i)
      for (int j = i + 1; j < K; ++j) {
        for (int k = 0; k < K; ++k)
          v[(size_t)k] = A.empty() ? (k == j ? 1.0 : (k == i ? -1.0 : 0.0))
                                   : A[(size_t)i * (size_t)K + (size_t)k] - A[(size_t)j * (size_t)K + (size_t)k];
        std::vector<std::pair<double, std::pair<float, float>>> items;
        for (int64_t r = 0; r < n; ++r) {
          const int c = (int)label[(size_t)r];
          if (c != i && c != j) continue;
          double t = 0.0;
          for (int k = 0; k < K; ++k)
            if (v[(size_t)k] != 0.0) t += v[(size_t)k] * score[(size_t)k * (size_t)n + (size_t)r];
          items.push_back({t, {c == j ? 1.0f : 0.0f, weight.empty() ? 1.0f : weight[(size_t)r]}});
        }
        sum += auc_host(items);
      }
    double b[2] = {sum * 2.0 / ((double)K * (double)(K - 1)), 1.0};   // distributed: the mean of the ranks' values
    engine_->sum_over_ranks(b, 2);
    out->push_back(EvalResult{"auc_mu", b[0] / b[1], true});
  }
  if (rank_metrics_.empty()) return;
  const Dataset* ds = which < 0 ? train_.get() : valid_sets_[(size_t)which].get();
  if (ds->query_boundaries().empty())
    throw std::runtime_error("the ndcg / map metrics need query information (Dataset group=) on every evaluated dataset");
  const std::vector<float> weight = ds->local_weight();
  if (!rank_labels_checked_.count(which)) {     // once per dataset: grades must be integers inside the gain table
    const std::vector<float> all_labels = ds->local_label();
    check_rank_labels(all_labels.data(), (int64_t)all_labels.size(), rank_.label_gain);
    rank_labels_checked_.insert(which);
  }
  // every cut-off either metric asks for; the engine may hand back both metrics per query in one go
  std::vector<int> all_at;
  for (const auto& m : rank_metrics_)
    if (std::find(all_at.begin(), all_at.end(), m.at) == all_at.end()) all_at.push_back(m.at);
  std::vector<double> per_query;
  const bool from_engine = engine_->ranking_metric_values(which, rank_, ds->query_boundaries(), all_at, &per_query);
  std::vector<double> score;
  std::vector<float> label;
  if (!from_engine) {
    const int64_t n = engine_->num_rows(which);
    score.resize((size_t)n * (size_t)num_class_);
    engine_->get_scores(which, score.data());
    label = ds->local_label();
  }
  for (int pass = 0; pass < 2; ++pass) {
    const bool ndcg = pass == 0;
    std::vector<int> at;
    for (const auto& m : rank_metrics_) if (m.ndcg == ndcg) at.push_back(m.at);
    if (at.empty()) continue;
    std::vector<double> sums;
    if (from_engine) {
      // pick this metric's cut-offs out of the engine's [ndcg x all_at | map x all_at] rows
      std::vector<double> all_sums;
      weighted_query_sums(per_query.data(), 2 * all_at.size(), ndcg ? 0 : all_at.size(), all_at.size(),
                          weight.empty() ? nullptr : weight.data(), ds->query_boundaries(), &all_sums);
      for (int k : at) sums.push_back(all_sums[(size_t)(std::find(all_at.begin(), all_at.end(), k) - all_at.begin())]);
      sums.push_back(all_sums.back());
    } else if (ndcg) ndcg_sums(score.data(), label.data(), weight.empty() ? nullptr : weight.data(), ds->query_boundaries(),
                               rank_.label_gain, at, &sums);
    else map_sums(score.data(), label.data(), weight.empty() ? nullptr : weight.data(), ds->query_boundaries(), at, &sums);
    engine_->sum_over_ranks(sums.data(), sums.size());
    for (size_t k = 0; k < at.size(); ++k)
      out->push_back(EvalResult{std::string(ndcg ? "ndcg@" : "map@") + std::to_string(at[k]),
                                sums.back() > 0.0 ? sums[k] / sums.back() : 0.0, true});
  }
}

bool Booster::update_one_iter() {
  if (!engine_) throw std::runtime_error("update on a prediction-only booster");
  if (obj_.kind == kObjCustom) throw std::runtime_error("objective=custom needs gradients (fobj)");
  engine_->set_group_size(1);
  ensure_scores();
  if (is_dart_) dart_drop();
  if (!is_rf_) compute_gradients();
  sample_rows();
  bool all_stumps = true;
  for (int k = 0; k < num_class_; ++k) {
    engine_->select_class(k);
    engine_->set_tree_index((int64_t)iter_ * num_class_ + k);
    Tree tree = engine_->train_tree(sample_features(k));
    renew_leaf_values_exact(tree);
    renew_leaf_outputs(tree);
    all_stumps = finish_tree(std::move(tree), k) && all_stumps;
  }
  engine_->select_class(0);
  if (is_dart_) dart_normalize();
  return finish_iteration(all_stumps);
}

bool Booster::update_begin() {
  two_step_ = false;
  if (!engine_) throw std::runtime_error("update on a prediction-only booster");
  if (num_class_ != 1 || is_dart_ || obj_.kind == kObjCustom) return false;   // update_finish runs the iteration
  ensure_scores();
  if (!is_rf_) compute_gradients();
  sample_rows();
  engine_->select_class(0);
  engine_->set_tree_index((int64_t)iter_);
  two_step_ = true;
  return engine_->train_tree_begin(sample_features(0));
}

bool Booster::update_finish() {
  if (!two_step_) {
    // (update_one_iter announces a group of one: keep the size the group set)
    return update_one_iter();
  }
  two_step_ = false;
  Tree tree = engine_->train_tree_finish();
  renew_leaf_values_exact(tree);
  renew_leaf_outputs(tree);
  const bool all_stumps = finish_tree(std::move(tree), 0);
  engine_->select_class(0);
  return finish_iteration(all_stumps);
}

void Booster::update_group(Booster* const* boosters, int n, int* finished) {
  std::vector<Engine*> prepared;
  for (int i = 0; i < n; ++i) {
    if (boosters[i]->engine_) boosters[i]->engine_->set_group_size(n);
    if (boosters[i]->update_begin()) prepared.push_back(boosters[i]->engine_.get());
  }
  if (!prepared.empty()) prepared[0]->launch_prepared_trees(prepared.data(), (int)prepared.size());
  for (int i = 0; i < n; ++i) finished[i] = boosters[i]->update_finish() ? 1 : 0;
}

// ---------------------------------------------------------------------------
// chunks of iterations on the engine (Engine::train_chunk_begin)
// ---------------------------------------------------------------------------
namespace {
bool pointwise_objective(int32_t kind) {
  switch (kind) {
    case kObjRegression: case kObjRegressionSqrt: case kObjHuber: case kObjFair: case kObjPoisson: case kObjBinary:
    case kObjCrossEntropy: case kObjGamma: case kObjTweedie: return true;
    default: return false;   // percentile leaf outputs, ranking, multiclass, custom
  }
}
bool pointwise_metric(int32_t m) { return m != kMetAuc && m != kMetMultiLogloss && m != kMetMultiError; }
}  // namespace

bool Booster::chunk_eligible(bool eval_valid) const {
  if (!engine_ || num_class_ != 1 || is_dart_ || is_rf_ || !pointwise_objective(obj_.kind)) return false;
  if (cfg_.data_sample_strategy == "goss") return false;
  if (cfg_.path_smooth > 0.0 || !cfg_.interaction_constraints.empty()) return false;
  for (int m : cfg_.monotone_constraints) if (m != 0) return false;
  if (!cfg_.feature_contri.empty() || !cfg_.cegb_penalty_feature_coupled.empty() || cfg_.cegb_penalty_split > 0.0) return false;
  if (eval_valid) {
    if (engine_->num_valid() != 1 || metrics_.empty() || metrics_.size() > 4) return false;
    if (!rank_metrics_.empty() || eval_average_precision_ || eval_auc_mu_) return false;
    for (int32_t m : metrics_) if (!pointwise_metric(m)) return false;
  } else if (engine_->num_valid() > 1) {
    return false;
  }
  return true;
}

bool Booster::chunk_begin(int n_iters, bool eval_valid) {
  ensure_scores();
  ChunkPlan plan;
  plan.obj = obj_;
  plan.learning_rate = cfg_.learning_rate;
  plan.renew = cfg_.quant_train_renew_leaf;
  plan.eval_valid = eval_valid;
  if (eval_valid) plan.metrics = metrics_;
  const bool balanced = cfg_.pos_bagging_fraction < 1.0 || cfg_.neg_bagging_fraction < 1.0;
  if (balanced && obj_.kind != kObjBinary)
    throw std::runtime_error("pos_bagging_fraction / neg_bagging_fraction need objective=binary");
  const bool bagging = cfg_.bagging_freq > 0 && (cfg_.bagging_fraction < 1.0 || balanced);
  plan.bagging = !bagging ? 0 : (balanced ? 3 : 1);
  plan.bag_fraction = balanced ? cfg_.pos_bagging_fraction : cfg_.bagging_fraction;
  plan.neg_fraction = cfg_.neg_bagging_fraction;
  plan.bag_seed = (uint64_t)cfg_.bagging_seed;
  bool drawn = bag_drawn_;
  for (int j = 0; j < n_iters; ++j) {
    const int it = iter_ + j;
    ChunkIterPlan ip;
    ip.meta = sample_features_at(0, it);
    ip.tree_index = (int64_t)it;
    if (bagging) {
      // the bag of iteration k is drawn at the last multiple of bagging_freq <= k (draw_bag)
      const int period_start = it - it % cfg_.bagging_freq;
      ip.redraw = !drawn || period_start == it;
      ip.bag_iter = (uint64_t)period_start;
      drawn = true;
    }
    plan.iters.push_back(std::move(ip));
  }
  if (!engine_->train_chunk_begin(plan)) return false;
  if (bagging) bag_drawn_ = true;
  return true;
}

void Booster::chunk_fetch(std::vector<char>* stump) {
  engine_->train_chunk_finish(&chunk_trees_);
  stump->assign(chunk_trees_.size(), 0);
  for (size_t j = 0; j < chunk_trees_.size(); ++j) (*stump)[j] = chunk_trees_[j].tree.num_leaves <= 1 ? 1 : 0;
}

// Takes over the first n_keep iterations of the fetched chunk: renews and shrinks the trees' leaf values the way the
// engine did on its side, stores them, and turns the metric sums into values.  Iterations beyond n_keep - the
// group stopped before them - are taken out of the engine's scores again.
void Booster::chunk_commit(int n_keep, bool eval_valid, double* values, int stride, int n_group, int index) {
  const int n_iters = (int)chunk_trees_.size();
  const int nv = engine_->num_valid();
  for (int j = 0; j < n_iters; ++j) {
    ChunkTree& ct = chunk_trees_[(size_t)j];
    Tree tree = std::move(ct.tree);
    const bool is_stump = tree.num_leaves <= 1;
    if (!is_stump) {
      if (cfg_.quant_train_renew_leaf) renew_from_sums(tree, ct.leaf_sums, ct.kg, ct.kh);
      tree.apply_shrinkage(cfg_.learning_rate);
    }
    if (j >= n_keep) {
      if (!is_stump) for (int w = -1; w < nv; ++w) engine_->add_tree_by_traversal(w, tree, -1.0);
      continue;
    }
    const bool first_iteration = (int)trees_.size() < num_class_;
    const double init = init_scores_[0];
    if (is_stump) {
      tree.leaf_value[0] = 0.0;
      if (first_iteration) tree.add_bias(init);
    } else if (first_iteration && init != 0.0) {
      tree.add_bias(init);
    }
    trees_.push_back(std::move(tree));
    // an iteration without a split is dropped again unless it is the very first one (finish_iteration)
    ++iter_;
    if (is_stump && (int)trees_.size() > num_class_) trees_.pop_back();
    if (eval_valid && values) {
      for (size_t m = 0; m < metrics_.size(); ++m) {
        const double s0 = ct.metric_sums[2 * m], s1 = ct.metric_sums[2 * m + 1];
        double v = s1 > 0.0 ? s0 / s1 : 0.0;
        if (metrics_[m] == kMetRmse) v = std::sqrt(v);
        values[((size_t)j * (size_t)n_group + (size_t)index) * (size_t)stride + m] = v;
      }
    }
  }
  chunk_trees_.clear();
}

int Booster::update_group_chunk(Booster* const* boosters, int n, int n_iters, bool eval_valid, double* values, int stride,
                                int* finished) {
  for (size_t i = 0; i < (size_t)n * (size_t)std::max(n_iters, 0); ++i) finished[i] = 0;
  if (n_iters < 1 || n < 1) return 0;
  bool chunked = true;
  for (int i = 0; i < n; ++i) {
    if (boosters[i]->engine_) boosters[i]->engine_->set_group_size(n);
    chunked = chunked && boosters[i]->chunk_eligible(eval_valid);
  }
  if (chunked) {
    // preparing, waiting for and taking over the chunks is host work per booster: spread it over the cores
    std::string failure;
    auto guarded = [&](auto&& body) {
#pragma omp parallel for schedule(dynamic, 1) if (n >= 8)
      for (int i = 0; i < n; ++i) {
        try {
          if (boosters[i]->engine_) boosters[i]->engine_->bind_thread();
          body(i);
        } catch (const std::exception& e) {
#pragma omp critical(lga_chunk_failure)
          if (failure.empty()) failure = e.what();
        }
      }
      if (!failure.empty()) throw std::runtime_error(failure);
    };
    std::vector<char> prepared((size_t)n, 0);
    guarded([&](int i) { prepared[(size_t)i] = boosters[i]->chunk_begin(n_iters, eval_valid) ? 1 : 0; });
    int n_prepared = 0;
    for (int i = 0; i < n; ++i) n_prepared += prepared[(size_t)i];
    if (n_prepared == n) {
      std::vector<Engine*> engines((size_t)n);
      for (int i = 0; i < n; ++i) engines[(size_t)i] = boosters[i]->engine_.get();
      engines[0]->launch_prepared_chunks(engines.data(), n);
      std::vector<std::vector<char>> stump((size_t)n);
      guarded([&](int i) { boosters[i]->chunk_fetch(&stump[(size_t)i]); });
      // the plain loop stops after an iteration in which no booster found a split
      int done = n_iters;
      for (int j = 0; j < n_iters; ++j) {
        bool all = true;
        for (int i = 0; i < n; ++i) all = all && stump[(size_t)i][(size_t)j];
        if (all) { done = j + 1; break; }
      }
      guarded([&](int i) {
        for (int j = 0; j < done; ++j) finished[(size_t)j * (size_t)n + (size_t)i] = stump[(size_t)i][(size_t)j];
        boosters[i]->chunk_commit(done, eval_valid, values, stride, n, i);
      });
      return done;
    }
    if (n_prepared > 0)
      throw std::runtime_error("a group of boosters must run on engines of one kind: some prepared a chunk of iterations, others refused");
  }
  int done = 0;
  for (int it = 0; it < n_iters; ++it) {
    int* fin = finished + (size_t)it * (size_t)n;
    update_group(boosters, n, fin);
    bool all = true;
    for (int i = 0; i < n; ++i) {
      all = all && fin[i] != 0;
      if (eval_valid && values) {
        auto r = boosters[i]->eval(0);
        if ((int)r.size() > stride) throw std::runtime_error("buffer too small");
        for (size_t m = 0; m < r.size(); ++m) values[((size_t)it * (size_t)n + (size_t)i) * (size_t)stride + m] = r[m].value;
      }
    }
    done = it + 1;
    if (all) break;
  }
  return done;
}

// ---------------------------------------------------------------------------
// DART (Rashmi & Gilad-Bachrach 2015): every iteration a random subset of the existing iterations is
// muted while the new trees are fitted, then the new trees and the muted ones are rescaled so that the
// ensemble's output stays on the same scale: with k muted iterations the new trees get the factor
// 1 / (k + 1) on top of the learning rate and the muted ones k / (k + 1) (xgboost_dart_mode: lr / (k + lr)
// and k / (k + lr)).  The initial score stays out of it: it is folded into the first iteration's trees
// only when the model is written, and taken out again (without_bias) whenever such a tree is replayed.
// ---------------------------------------------------------------------------
Tree Booster::without_bias(int tree_index) const {
  Tree t = trees_[(size_t)tree_index];
  const double init = init_scores_[(size_t)(tree_index % num_class_)];
  if (tree_index < num_class_ && init != 0.0) t.add_bias(-init);
  return t;
}

void Booster::dart_drop() {
  dropped_.clear();
  const int n_iter = current_iteration();
  tree_weight_.resize((size_t)n_iter, cfg_.learning_rate);
  const uint64_t seed = (uint64_t)cfg_.drop_seed;
  const bool skip = rng_uniform(seed, (uint64_t)iter_, 0xffffffffull) < (float)cfg_.skip_drop;
  if (!skip && n_iter > 0) {
    double rate = cfg_.drop_rate;
    if (!cfg_.uniform_drop) {
      // a heavier iteration is proportionally more likely to be muted
      const double inv_avg = (double)n_iter / sum_tree_weight_;
      if (cfg_.max_drop > 0) rate = std::min(rate, (double)cfg_.max_drop * inv_avg / sum_tree_weight_);
      for (int i = 0; i < n_iter; ++i)
        if (rng_uniform(seed, (uint64_t)iter_, (uint64_t)i) < (float)(rate * tree_weight_[(size_t)i] * inv_avg))
          dropped_.push_back(i);
    } else {
      if (cfg_.max_drop > 0) rate = std::min(rate, (double)cfg_.max_drop / (double)n_iter);
      for (int i = 0; i < n_iter; ++i)
        if (rng_uniform(seed, (uint64_t)iter_, (uint64_t)i) < (float)rate) dropped_.push_back(i);
    }
    if (cfg_.max_drop > 0 && (int)dropped_.size() > cfg_.max_drop) dropped_.resize((size_t)cfg_.max_drop);
  }
  // mute them in the training scores (validation scores keep them until the rescaling)
  for (int it : dropped_)
    for (int k = 0; k < num_class_; ++k) {
      engine_->select_class(k);
      engine_->add_tree_by_traversal(-1, without_bias(it * num_class_ + k), -1.0);
    }
  engine_->select_class(0);
  const double k = (double)dropped_.size();
  dart_shrinkage_ = cfg_.xgboost_dart_mode ? (dropped_.empty() ? cfg_.learning_rate : cfg_.learning_rate / (cfg_.learning_rate + k))
                                           : cfg_.learning_rate / (1.0 + k);
}

void Booster::dart_normalize() {
  const double k = (double)dropped_.size();
  const double lr = cfg_.learning_rate;
  // factor of a muted iteration afterwards
  const double keep = cfg_.xgboost_dart_mode ? k / (k + lr) : k / (k + 1.0);
  const int nv = engine_->num_valid();
  for (int it : dropped_) {
    for (int c = 0; c < num_class_; ++c) {
      const int ti = it * num_class_ + c;
      Tree plain = without_bias(ti);
      engine_->select_class(c);
      engine_->add_tree_by_traversal(-1, plain, keep);                 // was muted: comes back scaled
      for (int v = 0; v < nv; ++v) engine_->add_tree_by_traversal(v, plain, keep - 1.0);   // was there in full
      const double init = init_scores_[(size_t)c];
      Tree& stored = trees_[(size_t)ti];
      if (ti < num_class_ && init != 0.0) { stored.add_bias(-init); stored.apply_shrinkage(keep); stored.add_bias(init); }
      else stored.apply_shrinkage(keep);
    }
    sum_tree_weight_ -= tree_weight_[(size_t)it] * (1.0 - keep);
    tree_weight_[(size_t)it] *= keep;
  }
  engine_->select_class(0);
  tree_weight_.push_back(dart_shrinkage_);
  sum_tree_weight_ += dart_shrinkage_;
}

// The trees are grown on 16-bit fixed-point gradients; the values their leaves output are then taken from
// 36-bit sums of the float gradients of each leaf's rows (exact integer sums: the same on every engine and for
// every sharding).  Left alone: trees whose leaf outputs are shaped by the search itself (monotone
// constraints, path smoothing) and leaves below a split on a category set, which carry that split's extra
// regularisation.
void Booster::renew_leaf_values_exact(Tree& tree) {
  if (!cfg_.quant_train_renew_leaf || tree.num_leaves <= 1) return;
  if (cfg_.path_smooth > 0.0) return;
  for (int m : cfg_.monotone_constraints) if (m != 0) return;
  if (cfg_.forcedsplits_filename.empty() && engine_->leaf_values_from_exact_sums()) return;
  std::vector<int64_t> sums;
  int kg = 0, kh = 0;
  engine_->leaf_gradient_sums(tree, &sums, &kg, &kh);
  engine_->sum_over_ranks_i64(sums.data(), sums.size());
  renew_from_sums(tree, sums, kg, kh);
}

void Booster::renew_from_sums(Tree& tree, const std::vector<int64_t>& sums, int kg, int kh) const {
  const double gunit = unit_of(kg), hunit = unit_of(kh);
  std::vector<char> below_set((size_t)tree.num_leaves, 0);
  for (int i = 0; i < tree.num_leaves - 1; ++i) {
    const bool set_split = (tree.decision_type[(size_t)i] & kCategoricalMask) && tree.threshold_bin[(size_t)i] < 0;
    if (!set_split) continue;
    for (int c : {tree.left_child[(size_t)i], tree.right_child[(size_t)i]})
      if (c < 0) below_set[(size_t)~c] = 1;
  }
  for (int l = 0; l < tree.num_leaves; ++l) {
    const double G = (double)sums[(size_t)l * 2] * gunit, H = (double)sums[(size_t)l * 2 + 1] * hunit;
    if (!(H > 0.0)) continue;   // no in-bag row here (possible on a rank's shard only in theory): keep the value
    const double l2 = cfg_.lambda_l2 + (below_set[(size_t)l] ? cfg_.cat_l2 : 0.0);
    tree.set_leaf_value(l, leaf_output(G, H, cfg_.lambda_l1, l2, cfg_.max_delta_step));
    tree.leaf_weight[(size_t)l] = H;
  }
}

// regression_l1, quantile and mape take a sign step per row, whose Newton leaf value is bounded by one; the
// loss-minimising constant of a leaf is a percentile of its residuals, which replaces it (percentile.hpp).
void Booster::renew_leaf_outputs(Tree& tree) {
  if (tree.num_leaves <= 1) return;
  double alpha;
  bool mape = false;
  switch (obj_.kind) {
    case kObjRegressionL1: alpha = 0.5; break;
    case kObjQuantile: alpha = obj_.alpha; break;
    case kObjMape: alpha = 0.5; mape = true; break;
    default: return;
  }
  std::vector<double> values(tree.leaf_value.begin(), tree.leaf_value.begin() + tree.num_leaves);
  engine_->leaf_residual_percentiles(tree, alpha, mape, &values);
  for (int l = 0; l < tree.num_leaves; ++l) tree.set_leaf_value(l, values[(size_t)l]);
}

// Row sampling of the iteration: gradient-based one-side sampling (after a warm-up of 1 / learning_rate
// iterations on all rows, like the reference implementation) or bagging.
void Booster::sample_rows() {
  if (cfg_.data_sample_strategy == "goss") {
    const int warmup = (int)(1.0 / cfg_.learning_rate);
    if (iter_ - cfg_.iteration_offset < warmup) return;
    engine_->goss_sample(cfg_.top_rate, cfg_.other_rate, (uint64_t)cfg_.bagging_seed, (uint64_t)iter_);
    return;
  }
  const bool balanced = cfg_.pos_bagging_fraction < 1.0 || cfg_.neg_bagging_fraction < 1.0;
  if (balanced && obj_.kind != kObjBinary)
    throw std::runtime_error("pos_bagging_fraction / neg_bagging_fraction need objective=binary");
  if (cfg_.bagging_freq > 0 && (cfg_.bagging_fraction < 1.0 || balanced)) draw_bag();
}

// The bag of iteration k is drawn at the last multiple of bagging_freq <= k, also when training starts
// in the middle of such a period (iteration_offset).
void Booster::draw_bag() {
  const int period_start = iter_ - iter_ % cfg_.bagging_freq;
  if (bag_drawn_ && period_start != iter_) return;
  if (cfg_.pos_bagging_fraction < 1.0 || cfg_.neg_bagging_fraction < 1.0)
    engine_->resample(cfg_.pos_bagging_fraction, (uint64_t)cfg_.bagging_seed, (uint64_t)period_start, cfg_.neg_bagging_fraction);
  else
    engine_->resample(cfg_.bagging_fraction, (uint64_t)cfg_.bagging_seed, (uint64_t)period_start);
  bag_drawn_ = true;
}

// grad / hess: [class][row] in the training dataset's row order (num_class * num_rows values).
bool Booster::update_one_iter_custom(const float* grad, const float* hess, int64_t n) {
  if (!engine_) throw std::runtime_error("update on a prediction-only booster");
  if (is_rf_) throw std::runtime_error("custom gradients are not supported with boosting=rf");
  if (is_dart_) throw std::runtime_error("custom gradients are not supported with boosting=dart (they would be taken before the trees are muted)");
  const int64_t rows = engine_->num_rows(-1);
  if (n != rows * num_class_) throw std::runtime_error("custom gradients: expected num_class * num_rows values");
  if (!scores_initialised_) {
    // gradients come from the caller: no boost-from-average, scores start at the data's init_score (or 0)
    init_scores_.assign((size_t)num_class_, 0.0);
    for (int k = 0; k < num_class_; ++k) { engine_->select_class(k); engine_->reset_scores(0.0); }
    scores_initialised_ = true;
  }
  for (int k = 0; k < num_class_; ++k) {
    engine_->select_class(k);
    engine_->set_gradients(grad + (int64_t)k * rows, hess + (int64_t)k * rows, rows);
  }
  sample_rows();
  bool all_stumps = true;
  for (int k = 0; k < num_class_; ++k) {
    engine_->select_class(k);
    engine_->set_tree_index((int64_t)iter_ * num_class_ + k);
    Tree tree = engine_->train_tree(sample_features(k));
    renew_leaf_values_exact(tree);
    all_stumps = finish_tree(std::move(tree), k) && all_stumps;
  }
  engine_->select_class(0);
  return finish_iteration(all_stumps);
}

// Adds class `cls`'s tree of the running iteration to the scores and to the model; returns true for a tree
// without a split (kept as a zero tree so that every iteration holds num_class trees).
bool Booster::finish_tree(Tree&& tree, int cls) {
  const bool first_iteration = (int)trees_.size() < num_class_;
  const double init = init_scores_[(size_t)cls];
  int nv = engine_->num_valid();
  if (tree.num_leaves <= 1) {
    tree.leaf_value[0] = 0.0;
    if (is_rf_) {
      // a forest member that predicts the initial score everywhere
      const double t = (double)((int)trees_.size() / num_class_);
      tree.add_bias(init);
      for (int w = -1; w < nv; ++w) {
        engine_->scale_scores(w, t / (t + 1.0));
        engine_->add_constant(w, init / (t + 1.0));
      }
    } else if (first_iteration) {
      tree.add_bias(init);
    }
    trees_.push_back(std::move(tree));
    return true;
  }
  if (!is_rf_) {
    tree.apply_shrinkage(is_dart_ ? dart_shrinkage_ : cfg_.learning_rate);
    engine_->add_trained_tree_to_train_scores(tree);
    for (int v = 0; v < nv; ++v) engine_->add_tree_by_traversal(v, tree, 1.0);
    if (first_iteration && init != 0.0) tree.add_bias(init);
  } else {
    // running average of (init + tree) over the forest
    double t = (double)((int)trees_.size() / num_class_);
    tree.add_bias(init);
    Tree scaled = tree;
    for (auto& v : scaled.leaf_value) v /= (t + 1.0);
    engine_->scale_scores(-1, t / (t + 1.0));
    engine_->add_trained_tree_to_train_scores(scaled);
    for (int v = 0; v < nv; ++v) {
      engine_->scale_scores(v, t / (t + 1.0));
      engine_->add_tree_by_traversal(v, scaled, 1.0);
    }
  }
  trees_.push_back(std::move(tree));
  return false;
}

// Ends an iteration.  When no class found a split there is nothing learnable any more: the iteration is
// dropped (except the very first one, whose zero trees carry the initial scores) and true is returned.
bool Booster::finish_iteration(bool all_stumps) {
  ++iter_;
  if (!all_stumps) return false;
  if (!is_rf_ && (int)trees_.size() > num_class_) {
    trees_.resize(trees_.size() - (size_t)num_class_);
    if (is_dart_ && !tree_weight_.empty()) { sum_tree_weight_ -= tree_weight_.back(); tree_weight_.pop_back(); }
  }
  return true;
}

void Booster::rollback_one_iter() {
  if (!engine_ || (int)trees_.size() < num_class_) return;
  if (is_dart_) throw std::runtime_error("boosting=dart rescales earlier trees every iteration; an iteration cannot be rolled back");
  const int K = num_class_;
  const int nv = engine_->num_valid();
  const bool only_iteration = (int)trees_.size() == K;
  for (int k = 0; k < K; ++k) {
    Tree last = trees_[trees_.size() - (size_t)K + (size_t)k];
    engine_->select_class(k);
    if (!is_rf_) {
      if (only_iteration && init_scores_[(size_t)k] != 0.0) last.add_bias(-init_scores_[(size_t)k]);
      for (int w = -1; w < nv; ++w) engine_->add_tree_by_traversal(w, last, -1.0);
    } else {
      double t = (double)((int)trees_.size() / K);
      if (only_iteration) {
        engine_->reset_scores(init_scores_[(size_t)k]);
      } else {
        for (int w = -1; w < nv; ++w) {
          engine_->add_tree_by_traversal(w, last, -1.0 / t);
          engine_->scale_scores(w, t / (t - 1.0));
        }
      }
    }
  }
  engine_->select_class(0);
  trees_.resize(trees_.size() - (size_t)K);
  --iter_;
}

void Booster::reset_parameter(const std::string& params) {
  Config c = cfg_;
  c.set(params);
  // What may change between iterations: everything that only steers the NEXT trees - shrinkage, row / feature
  // sampling, the split search's thresholds and regularisation, DART's drop rates.  Every accepted parameter
  // is applied; anything else that differs is refused rather than silently dropped.
  // a depth limit caps num_leaves when a configuration is read; a booster keeps the leaf budget it was built with
  // (the grower enforces the new depth limit itself)
  if (c.max_depth != cfg_.max_depth && params.find("leav") == std::string::npos && params.find("leaf_nodes") == std::string::npos)
    c.num_leaves = cfg_.num_leaves;
  Config applied = cfg_;
  applied.learning_rate = c.learning_rate;
  applied.bagging_fraction = c.bagging_fraction; applied.bagging_freq = c.bagging_freq;
  applied.pos_bagging_fraction = c.pos_bagging_fraction; applied.neg_bagging_fraction = c.neg_bagging_fraction;
  applied.feature_fraction = c.feature_fraction; applied.feature_fraction_bynode = c.feature_fraction_bynode;
  applied.lambda_l1 = c.lambda_l1; applied.lambda_l2 = c.lambda_l2;
  applied.min_gain_to_split = c.min_gain_to_split; applied.min_sum_hessian_in_leaf = c.min_sum_hessian_in_leaf;
  applied.min_data_in_leaf = c.min_data_in_leaf; applied.max_depth = c.max_depth; applied.max_delta_step = c.max_delta_step;
  applied.path_smooth = c.path_smooth; applied.extra_trees = c.extra_trees; applied.cat_l2 = c.cat_l2;
  applied.drop_rate = c.drop_rate; applied.max_drop = c.max_drop; applied.skip_drop = c.skip_drop;
  applied.top_rate = c.top_rate; applied.other_rate = c.other_rate;
  applied.quant_train_renew_leaf = c.quant_train_renew_leaf;
  applied.verbosity = c.verbosity; applied.metric_freq = c.metric_freq; applied.early_stopping_round = c.early_stopping_round;
  applied.first_metric_only = c.first_metric_only; applied.num_iterations = c.num_iterations;
  applied.snapshot_freq = c.snapshot_freq; applied.output_model = c.output_model; applied.num_threads = c.num_threads;
  if (applied.to_model_string() != c.to_model_string())
    throw std::runtime_error("reset_parameter: only the learning rate, row / feature sampling fractions, the split search's "
                             "regularisation and limits (lambda_l1, lambda_l2, min_data_in_leaf, min_sum_hessian_in_leaf, "
                             "min_gain_to_split, max_depth, max_delta_step, path_smooth, extra_trees) and the DART / GOSS "
                             "rates can be reset during training; got: " + params);
  if ((applied.path_smooth > 0.0) != (cfg_.path_smooth > 0.0) && engine_)
    throw std::runtime_error("reset_parameter: path_smooth cannot be switched on or off during training");
  cfg_ = applied;
  if (engine_) engine_->update_split_config(cfg_);
}

void Booster::refit(const void* data, int dtype, int64_t nrow, int ncol, bool row_major, const float* label,
                    const float* weight, double decay, const std::string& params) {
  if (ncol <= max_feature_idx_) throw std::runtime_error("refit: the data has fewer features than the model");
  if (is_ranking() || obj_.kind == kObjCustom)
    throw std::runtime_error("refit needs an objective with per-row gradients (not ranking, not custom)");
  if (average_output_) throw std::runtime_error("refit is not defined for averaged (boosting=rf) models");
  if (!(decay >= 0.0 && decay <= 1.0)) throw std::runtime_error("refit: decay_rate must be in [0, 1]");
  Config c = cfg_;
  c.set(params);
  const int K = num_class_, n_it = current_iteration();
  std::vector<double> scores((size_t)K * (size_t)nrow, 0.0);
  std::vector<float> g((size_t)K * (size_t)nrow), h((size_t)K * (size_t)nrow);
  std::vector<int32_t> leaf((size_t)nrow);
  for (int it = 0; it < n_it; ++it) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nrow; ++i) {
      const float w = weight ? weight[i] : 1.0f;
      if (K == 1) grad_hess(obj_, scores[(size_t)i], label[i], w, &g[(size_t)i], &h[(size_t)i]);
      else multiclass_grad_hess(obj_, scores.data(), nrow, i, label[i], w, g.data(), h.data());
    }
    for (int k = 0; k < K; ++k) {
      Tree& tr = trees_[(size_t)it * (size_t)K + (size_t)k];
#pragma omp parallel
      {
        std::vector<double> row((size_t)ncol);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < nrow; ++i) {
          for (int col = 0; col < ncol; ++col) {
            const int64_t off = row_major ? i * (int64_t)ncol + col : (int64_t)col * nrow + i;
            row[(size_t)col] = dtype == 0 ? (double)((const float*)data)[off] : ((const double*)data)[off];
          }
          leaf[(size_t)i] = tr.predict_leaf(row.data());
        }
      }
      std::vector<double> sg((size_t)tr.num_leaves, 0.0), sh((size_t)tr.num_leaves, 0.0);
      std::vector<int64_t> cnt((size_t)tr.num_leaves, 0);
      for (int64_t i = 0; i < nrow; ++i) {     // in row order: the sums do not depend on the thread count
        const size_t l = (size_t)leaf[(size_t)i];
        sg[l] += (double)g[(size_t)k * (size_t)nrow + (size_t)i];
        sh[l] += (double)h[(size_t)k * (size_t)nrow + (size_t)i];
        ++cnt[l];
      }
      for (int l = 0; l < tr.num_leaves; ++l) {
        if (cnt[(size_t)l] == 0) continue;       // no row reaches the leaf: it keeps its value
        const double fresh = leaf_output(sg[(size_t)l], sh[(size_t)l], c.lambda_l1, c.lambda_l2, c.max_delta_step) * tr.shrinkage;
        tr.set_leaf_value(l, decay * tr.leaf_value[(size_t)l] + (1.0 - decay) * fresh);
      }
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < nrow; ++i) scores[(size_t)k * (size_t)nrow + (size_t)i] += tr.leaf_value[(size_t)leaf[(size_t)i]];
    }
  }
}

void Booster::shuffle_iterations(int start, int end, uint64_t seed) {
  const int K = num_class_, n_it = current_iteration();
  start = std::max(0, start);
  end = end <= 0 ? n_it : std::min(end, n_it);
  if (end - start < 2) return;
  // the first iteration's trees carry the initial score: it moves with them, which leaves every sum unchanged
  std::vector<int> order((size_t)n_it);
  std::iota(order.begin(), order.end(), 0);
  for (int i = end - 1; i > start; --i) {
    const int j = start + (int)(rng_u64(seed, 0x5aff1eull, (uint64_t)i) % (uint64_t)(i - start + 1));
    std::swap(order[(size_t)i], order[(size_t)j]);
  }
  std::vector<Tree> shuffled;
  shuffled.reserve(trees_.size());
  for (int it = 0; it < n_it; ++it)
    for (int k = 0; k < K; ++k) shuffled.push_back(std::move(trees_[(size_t)order[(size_t)it] * (size_t)K + (size_t)k]));
  trees_ = std::move(shuffled);
  if (!tree_weight_.empty() && (int)tree_weight_.size() == n_it) {
    std::vector<double> w((size_t)n_it);
    for (int it = 0; it < n_it; ++it) w[(size_t)it] = tree_weight_[(size_t)order[(size_t)it]];
    tree_weight_ = std::move(w);
  }
}

std::vector<EvalResult> Booster::eval(int which) {
  if (!engine_) throw std::runtime_error("eval on a prediction-only booster");
  ensure_scores();
  std::vector<EvalResult> out = engine_->eval(which, metrics_, obj_);
  eval_host_metrics(which, &out);
  return out;
}

std::vector<std::pair<std::string, bool>> Booster::eval_names() const {
  std::vector<std::pair<std::string, bool>> out;
  for (int32_t m : metrics_) out.emplace_back(metric_name(m), metric_higher_better(m));
  if (eval_average_precision_) out.emplace_back("average_precision", true);
  if (eval_auc_mu_) out.emplace_back("auc_mu", true);
  for (int pass = 0; pass < 2; ++pass)
    for (const auto& m : rank_metrics_)
      if (m.ndcg == (pass == 0)) out.emplace_back(std::string(m.ndcg ? "ndcg@" : "map@") + std::to_string(m.at), true);
  return out;
}

// Link function over one row's num_class raw scores (in place).
void Booster::transform_row(double* v) const {
  const int K = num_class_;
  if (obj_.kind == kObjMulticlass) {
    double mx = v[0];
    for (int k = 1; k < K; ++k) mx = std::max(mx, v[k]);
    double sum = 0.0;
    for (int k = 0; k < K; ++k) { v[k] = std::exp(v[k] - mx); sum += v[k]; }
    for (int k = 0; k < K; ++k) v[k] /= sum;
  } else if (obj_.kind == kObjMulticlassOva) {
    for (int k = 0; k < K; ++k) v[k] = 1.0 / (1.0 + std::exp(-obj_.sigmoid * v[k]));
  } else {
    for (int k = 0; k < K; ++k) v[k] = transform_score(obj_.kind, obj_.sigmoid, v[k]);
  }
}

// out: [class][row]
void Booster::inner_predict(int which, bool raw, double* out) {
  ensure_scores();
  engine_->get_scores(which, out);
  if (raw) return;
  const int64_t n = engine_->num_rows(which);
  const int K = num_class_;
  if (K == 1) {
    for (int64_t i = 0; i < n; ++i) out[i] = transform_score(obj_.kind, obj_.sigmoid, out[i]);
    return;
  }
  std::vector<double> row((size_t)K);
  for (int64_t i = 0; i < n; ++i) {
    for (int k = 0; k < K; ++k) row[(size_t)k] = out[(int64_t)k * n + i];
    transform_row(row.data());
    for (int k = 0; k < K; ++k) out[(int64_t)k * n + i] = row[(size_t)k];
  }
}

int Booster::used_iterations(int start_iter, int num_iter) const {
  int total = current_iteration();
  start_iter = std::max(0, std::min(start_iter, total));
  int n = num_iter > 0 ? std::min(num_iter, total - start_iter) : total - start_iter;
  return n;
}

void Booster::predict(const void* data, int dtype, int64_t nrow, int ncol, bool row_major, int start_iter,
                      int num_iter, int type, double* out) const {
  if (ncol <= max_feature_idx_)
    throw std::runtime_error("predict: the data has " + std::to_string(ncol) + " features, the model needs " +
                             std::to_string(max_feature_idx_ + 1));
  const int K = num_class_;
  int total = current_iteration();
  start_iter = std::max(0, std::min(start_iter, total));
  int n_it = used_iterations(start_iter, num_iter);
  auto tree_at = [&](int it, int k) -> const Tree& { return trees_[(size_t)(start_iter + it) * (size_t)K + (size_t)k]; };
#pragma omp parallel
  {
    std::vector<double> row((size_t)ncol);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < nrow; ++i) {
      for (int c = 0; c < ncol; ++c) {
        int64_t off = row_major ? i * (int64_t)ncol + c : (int64_t)c * nrow + i;
        row[(size_t)c] = dtype == 0 ? (double)((const float*)data)[off] : ((const double*)data)[off];
      }
      if (type == kPredictLeaf) {
        for (int t = 0; t < n_it; ++t)
          for (int k = 0; k < K; ++k)
            out[(i * (int64_t)n_it + t) * K + k] = (double)tree_at(t, k).predict_leaf(row.data());
      } else if (type == kPredictContrib) {
        // per class [ncol contributions | expected value], in raw-score units
        for (int k = 0; k < K; ++k) {
          double* phi = out + (i * (int64_t)K + k) * (int64_t)(ncol + 1);
          for (int c = 0; c <= ncol; ++c) phi[c] = 0.0;
          for (int t = 0; t < n_it; ++t) tree_at(t, k).add_contributions(row.data(), ncol, phi);
          if (average_output_ && n_it > 0)
            for (int c = 0; c <= ncol; ++c) phi[c] /= (double)n_it;
        }
      } else {
        double* o = out + i * (int64_t)K;
        for (int k = 0; k < K; ++k) {
          double s = 0.0;
          for (int t = 0; t < n_it; ++t) s += tree_at(t, k).predict(row.data());
          if (average_output_ && n_it > 0) s /= (double)n_it;
          o[k] = s;
        }
        if (type != kPredictRaw) transform_row(o);
      }
    }
  }
}

std::vector<double> Booster::feature_importance(int num_iter, int importance_type) const {
  std::vector<double> imp((size_t)(max_feature_idx_ + 1), 0.0);
  int n = (num_iter > 0 ? std::min(num_iter, current_iteration()) : current_iteration()) * num_class_;
  for (int t = 0; t < n; ++t) {
    const Tree& tr = trees_[(size_t)t];
    for (int i = 0; i + 1 < tr.num_leaves; ++i) {
      if (tr.split_gain[(size_t)i] <= 0.0 && importance_type == 1) continue;
      imp[(size_t)tr.split_feature[(size_t)i]] += importance_type == 0 ? 1.0 : tr.split_gain[(size_t)i];
    }
  }
  return imp;
}

}  // namespace lga

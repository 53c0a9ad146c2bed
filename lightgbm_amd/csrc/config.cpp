#include "config.hpp"
#include <algorithm>
#include <cstdlib>
#include <sstream>
#include <stdexcept>
#include <limits>
#include <cctype>

namespace lga {
namespace {

enum class PType { Int, Double, Bool, String, StringList, IntList, DoubleList };

struct ParamDef {
  const char* name;
  const char* aliases;  // comma separated
  PType type;
  size_t offset;        // offset of the member inside Config
  double lo, hi;        // numeric range (inclusive), NaN = unbounded
};

const double kNone = std::numeric_limits<double>::quiet_NaN();
#define P(member, aliases, type, lo, hi) \
  { #member, aliases, PType::type, offsetof(Config, member), lo, hi }

// Table order = order of the "parameters:" block in the model file.
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Winvalid-offsetof"
const ParamDef kParams[] = {
    P(boosting, "boosting_type,boost", String, kNone, kNone),
    P(objective, "objective_type,app,application,loss", String, kNone, kNone),
    P(metric, "metrics,metric_types", StringList, kNone, kNone),
    P(tree_learner, "tree,tree_type,tree_learner_type", String, kNone, kNone),
    P(device_type, "device", String, kNone, kNone),
    P(top_k, "topk", Int, 1, kNone),
    P(num_iterations,
      "num_iteration,n_iter,num_tree,num_trees,num_round,num_rounds,nrounds,num_boost_round,n_estimators,max_iter",
      Int, 0, kNone),
    P(learning_rate, "shrinkage_rate,eta", Double, 1e-300, kNone),
    P(num_leaves, "num_leaf,max_leaves,max_leaf,max_leaf_nodes", Int, 2, 131072),
    P(num_threads, "num_thread,nthread,nthreads,n_jobs", Int, kNone, kNone),
    P(seed, "random_seed,random_state", Int, kNone, kNone),
    P(deterministic, "", Bool, kNone, kNone),
    P(max_depth, "", Int, kNone, kNone),
    P(min_data_in_leaf, "min_data_per_leaf,min_data,min_child_samples,min_samples_leaf", Int, 0, kNone),
    P(min_sum_hessian_in_leaf, "min_sum_hessian_per_leaf,min_sum_hessian,min_hessian,min_child_weight", Double, 0, kNone),
    P(bagging_fraction, "sub_row,subsample,bagging", Double, 1e-300, 1.0),
    P(pos_bagging_fraction, "pos_sub_row,pos_subsample,pos_bagging", Double, 1e-300, 1.0),
    P(neg_bagging_fraction, "neg_sub_row,neg_subsample,neg_bagging", Double, 1e-300, 1.0),
    P(bagging_freq, "subsample_freq", Int, 0, kNone),
    P(bagging_seed, "bagging_fraction_seed", Int, kNone, kNone),
    P(categorical_feature, "cat_feature,categorical_column,cat_column,categorical_features", IntList, 0, kNone),
    P(max_cat_to_onehot, "", Int, 1, kNone),
    P(cat_smooth, "", Double, 0, kNone),
    P(cat_l2, "", Double, 0, kNone),
    P(max_cat_threshold, "", Int, 1, kNone),
    P(min_data_per_group, "", Int, 1, kNone),
    P(monotone_constraints, "mc,monotone_constraint,monotonic_cst", IntList, -1, 1),
    P(drop_rate, "rate_drop", Double, 0, 1.0),
    P(max_drop, "", Int, kNone, kNone),
    P(skip_drop, "", Double, 0, 1.0),
    P(uniform_drop, "", Bool, kNone, kNone),
    P(xgboost_dart_mode, "", Bool, kNone, kNone),
    P(drop_seed, "", Int, kNone, kNone),
    P(data_sample_strategy, "", String, kNone, kNone),
    P(top_rate, "", Double, 0, 1.0),
    P(other_rate, "", Double, 0, 1.0),
    P(feature_fraction, "sub_feature,colsample_bytree", Double, 1e-300, 1.0),
    P(feature_fraction_bynode, "sub_feature_bynode,colsample_bynode", Double, 1e-300, 1.0),
    P(extra_trees, "extra_tree", Bool, kNone, kNone),
    P(extra_seed, "", Int, kNone, kNone),
    P(feature_fraction_seed, "", Int, kNone, kNone),
    P(early_stopping_round, "early_stopping_rounds,early_stopping,n_iter_no_change", Int, kNone, kNone),
    P(first_metric_only, "", Bool, kNone, kNone),
    P(max_delta_step, "max_tree_output,max_leaf_output", Double, kNone, kNone),
    P(lambda_l1, "reg_alpha,l1_regularization", Double, 0, kNone),
    P(lambda_l2, "reg_lambda,lambda,l2_regularization", Double, 0, kNone),
    P(min_gain_to_split, "min_split_gain", Double, 0, kNone),
    P(feature_contri, "feature_contrib,fc,fp,feature_penalty", DoubleList, 0, kNone),
    P(cegb_tradeoff, "", Double, 0, kNone),
    P(cegb_penalty_split, "", Double, 0, kNone),
    P(cegb_penalty_feature_coupled, "", DoubleList, 0, kNone),
    P(cegb_penalty_feature_lazy, "", DoubleList, 0, kNone),
    P(use_quantized_grad, "quantized_grad", Bool, kNone, kNone),
    P(quant_train_renew_leaf, "renew_leaf_values", Bool, kNone, kNone),
    P(path_smooth, "", Double, 0, kNone),
    P(interaction_constraints, "", String, kNone, kNone),
    P(verbosity, "verbose", Int, kNone, kNone),
    P(max_bin, "max_bins", Int, 2, 255),
    P(max_bin_by_feature, "", IntList, 2, 255),
    P(min_data_in_bin, "", Int, 1, kNone),
    P(bin_construct_sample_cnt, "subsample_for_bin", Int, 1, kNone),
    P(data_random_seed, "data_seed", Int, kNone, kNone),
    P(use_missing, "", Bool, kNone, kNone),
    P(zero_as_missing, "", Bool, kNone, kNone),
    P(feature_pre_filter, "", Bool, kNone, kNone),
    P(boost_from_average, "", Bool, kNone, kNone),
    P(sigmoid, "", Double, 1e-300, kNone),
    P(num_class, "num_classes", Int, 1, kNone),
    P(is_unbalance, "unbalance,unbalanced_sets", Bool, kNone, kNone),
    P(scale_pos_weight, "", Double, 1e-300, kNone),
    P(alpha, "", Double, 1e-300, kNone),
    P(fair_c, "", Double, 1e-300, kNone),
    P(poisson_max_delta_step, "", Double, 1e-300, kNone),
    P(tweedie_variance_power, "", Double, 1.0, 1.999999999),
    P(lambdarank_truncation_level, "", Int, 1, kNone),
    P(lambdarank_norm, "", Bool, kNone, kNone),
    P(label_gain, "", DoubleList, 0, kNone),
    P(objective_seed, "", Int, kNone, kNone),
    P(eval_at, "ndcg_eval_at,ndcg_at,map_eval_at,map_at", IntList, 1, kNone),
    P(metric_freq, "output_freq", Int, 1, kNone),
    P(is_provide_training_metric, "training_metric,is_training_metric,train_metric", Bool, kNone, kNone),
    P(snapshot_freq, "save_period", Int, kNone, kNone),
    P(output_model, "model_output,model_out", String, kNone, kNone),
    P(input_model, "model_input,model_in", String, kNone, kNone),
    P(forcedsplits_filename, "fs,forced_splits_filename,forced_splits_file,forced_splits", String, kNone, kNone),
    P(label_column, "label", String, kNone, kNone),
    P(header, "has_header", Bool, kNone, kNone),
    P(num_machines, "num_machine", Int, 1, kNone),
    P(gpu_device_id, "", Int, kNone, kNone),
    P(max_batch_leaves, "", Int, 0, kNone),
    P(whole_tree_max_rows, "gpu_whole_tree_max_rows", Int, 0, 65536),
    P(profile, "", Bool, kNone, kNone),
    P(locality_order, "", Bool, kNone, kNone),
    P(device_bin_find, "", Bool, kNone, kNone),
    P(iteration_offset, "", Int, 0, kNone),
    P(early_stopping_min_delta, "", Double, 0, kNone),
    P(multi_error_top_k, "", Int, 1, kNone),
    P(auc_mu_weights, "", DoubleList, 0, kNone),
    P(saved_feature_importance_type, "", Int, 0, 1),
    P(monotone_constraints_method, "monotone_constraining_method,mc_method", String, kNone, kNone),
    P(force_col_wise, "", Bool, kNone, kNone),
    P(force_row_wise, "", Bool, kNone, kNone),
    P(histogram_pool_size, "hist_pool_size", Double, kNone, kNone),
    P(gpu_platform_id, "", Int, kNone, kNone),
    P(gpu_use_dp, "", Bool, kNone, kNone),
    P(num_gpu, "", Int, 1, kNone),
    P(two_round, "two_round_loading,use_two_round_loading", Bool, kNone, kNone),
    P(precise_float_parser, "", Bool, kNone, kNone),
    P(enable_bundle, "is_enable_bundle,bundle", Bool, kNone, kNone),
    P(num_grad_quant_bins, "", Int, 2, kNone),
    P(stochastic_rounding, "", Bool, kNone, kNone),
    P(forcedbins_filename, "", String, kNone, kNone),
    P(reg_sqrt, "", Bool, kNone, kNone),
    P(pred_early_stop, "", Bool, kNone, kNone),
    P(pred_early_stop_freq, "", Int, 1, kNone),
    P(pred_early_stop_margin, "", Double, 0, kNone),
};
#pragma GCC diagnostic pop
#undef P

std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> out;
  std::string cur;
  for (char c : s) {
    if (c == sep) { if (!cur.empty()) out.push_back(cur); cur.clear(); }
    else cur.push_back(c);
  }
  if (!cur.empty()) out.push_back(cur);
  return out;
}

std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && std::isspace((unsigned char)s[a])) ++a;
  while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}

std::string lower(std::string s) {
  for (auto& c : s) c = (char)std::tolower((unsigned char)c);
  return s;
}

const ParamDef* find_param(const std::string& name) {
  for (const auto& p : kParams) {
    if (name == p.name) return &p;
    for (const auto& a : split(p.aliases, ',')) if (name == a) return &p;
  }
  return nullptr;
}

bool parse_bool(const std::string& v, const std::string& key) {
  std::string s = lower(v);
  if (s == "true" || s == "1" || s == "+" || s == "yes" || s == "t") return true;
  if (s == "false" || s == "0" || s == "-" || s == "no" || s == "f") return false;
  throw std::runtime_error("parameter " + key + " expects a boolean, got \"" + v + "\"");
}

template <typename T>
T& field(Config* c, size_t off) { return *reinterpret_cast<T*>(reinterpret_cast<char*>(c) + off); }
template <typename T>
const T& cfield(const Config* c, size_t off) { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(c) + off); }

std::string fmt_double(double v) {
  std::ostringstream os;
  os.precision(17);
  os << v;
  return os.str();
}

}  // namespace

std::string Config::resolve_alias(const std::string& name) {
  const ParamDef* p = find_param(name);
  return p ? p->name : "";
}

std::string Config::alias_table_json() {
  std::ostringstream os;
  os << "{";
  bool first = true;
  for (const auto& p : kParams) {
    if (!first) os << ",";
    first = false;
    os << "\"" << p.name << "\":[";
    bool f2 = true;
    for (const auto& a : split(p.aliases, ',')) {
      if (!f2) os << ",";
      f2 = false;
      os << "\"" << a << "\"";
    }
    os << "]";
  }
  os << "}";
  return os.str();
}

void Config::set(const std::map<std::string, std::string>& kv) {
  for (const auto& it : kv) {
    const std::string key = trim(it.first);
    const std::string val = trim(it.second);
    const ParamDef* p = find_param(key);
    if (!p) {
      if (std::find(unknown_keys.begin(), unknown_keys.end(), key) == unknown_keys.end()) unknown_keys.push_back(key);
      continue;
    }
    char* end = nullptr;
    switch (p->type) {
      case PType::Int: {
        // accept "31", "31.0" (a data.frame row turns every column into a double)
        double d = std::strtod(val.c_str(), &end);
        if (end == val.c_str() || *end != '\0' || d != (double)(long long)d)
          throw std::runtime_error("parameter " + key + " expects an integer, got \"" + val + "\"");
        if ((p->lo == p->lo && d < p->lo) || (p->hi == p->hi && d > p->hi))
          throw std::runtime_error("parameter " + key + "=" + val + " is out of range");
        field<int>(this, p->offset) = (int)d;
        break;
      }
      case PType::Double: {
        double d = std::strtod(val.c_str(), &end);
        if (end == val.c_str() || *end != '\0')
          throw std::runtime_error("parameter " + key + " expects a number, got \"" + val + "\"");
        if ((p->lo == p->lo && d < p->lo) || (p->hi == p->hi && d > p->hi))
          throw std::runtime_error("parameter " + key + "=" + val + " is out of range");
        field<double>(this, p->offset) = d;
        break;
      }
      case PType::Bool:
        field<bool>(this, p->offset) = parse_bool(val, key);
        break;
      case PType::String:
        field<std::string>(this, p->offset) = val;
        break;
      case PType::StringList: {
        auto& lst = field<std::vector<std::string>>(this, p->offset);
        lst.clear();
        for (auto& m : split(val, ',')) {
          std::string t = trim(m);
          if (!t.empty()) lst.push_back(t);
        }
        break;
      }
      case PType::IntList: {
        auto& lst = field<std::vector<int>>(this, p->offset);
        lst.clear();
        for (auto& m : split(val, ',')) {
          std::string t = trim(m);
          if (t.empty()) continue;
          double d = std::strtod(t.c_str(), &end);
          if (end == t.c_str() || *end != '\0' || d != (double)(long long)d)
            throw std::runtime_error("parameter " + key + " expects integers, got \"" + t + "\"");
          if ((p->lo == p->lo && d < p->lo) || (p->hi == p->hi && d > p->hi))
            throw std::runtime_error("parameter " + key + ": value " + t + " is out of range");
          lst.push_back((int)d);
        }
        break;
      }
      case PType::DoubleList: {
        auto& lst = field<std::vector<double>>(this, p->offset);
        lst.clear();
        for (auto& m : split(val, ',')) {
          std::string t = trim(m);
          if (t.empty()) continue;
          double d = std::strtod(t.c_str(), &end);
          if (end == t.c_str() || *end != '\0')
            throw std::runtime_error("parameter " + key + " expects numbers, got \"" + t + "\"");
          if ((p->lo == p->lo && d < p->lo) || (p->hi == p->hi && d > p->hi))
            throw std::runtime_error("parameter " + key + ": value " + t + " is out of range");
          lst.push_back(d);
        }
        break;
      }
    }
    if (std::find(explicit_keys.begin(), explicit_keys.end(), p->name) == explicit_keys.end())
      explicit_keys.push_back(p->name);
  }
  check();
}

void Config::set(const std::string& params) {
  // tokens are separated by whitespace; a token is key=value
  std::map<std::string, std::string> kv;
  std::vector<std::pair<std::string, std::string>> ordered;
  std::istringstream is(params);
  std::string tok;
  while (is >> tok) {
    size_t eq = tok.find('=');
    if (eq == std::string::npos || eq == 0)
      throw std::runtime_error("malformed parameter token \"" + tok + "\" (expected key=value)");
    ordered.emplace_back(tok.substr(0, eq), tok.substr(eq + 1));
  }
  // canonical name wins over alias when both are given; otherwise last wins
  std::map<std::string, std::string> by_canon;
  std::map<std::string, bool> canon_direct;
  for (auto& p : ordered) {
    std::string canon = resolve_alias(p.first);
    if (canon.empty()) { kv[p.first] = p.second; continue; }
    bool direct = (canon == p.first);
    if (by_canon.count(canon) && canon_direct[canon] && !direct) continue;
    by_canon[canon] = p.second;
    canon_direct[canon] = direct;
  }
  for (auto& p : by_canon) kv[p.first] = p.second;
  set(kv);
}

std::string Config::canonical_objective(const std::string& name0) {
  std::string name = lower(name0);
  static const std::map<std::string, std::string> m = {
      {"regression", "regression"}, {"regression_l2", "regression"}, {"l2", "regression"},
      {"mean_squared_error", "regression"}, {"mse", "regression"}, {"l2_root", "regression"},
      {"root_mean_squared_error", "regression"}, {"rmse", "regression"}, {"reg:linear", "regression"},
      {"reg:squarederror", "regression"},
      {"regression_l1", "regression_l1"}, {"l1", "regression_l1"}, {"mean_absolute_error", "regression_l1"},
      {"mae", "regression_l1"},
      {"huber", "huber"}, {"fair", "fair"}, {"poisson", "poisson"},
      {"quantile", "quantile"}, {"mape", "mape"}, {"mean_absolute_percentage_error", "mape"},
      {"gamma", "gamma"}, {"tweedie", "tweedie"},
      {"multiclass", "multiclass"}, {"softmax", "multiclass"}, {"multi:softprob", "multiclass"},
      {"multiclassova", "multiclassova"}, {"multiclass_ova", "multiclassova"}, {"ova", "multiclassova"},
      {"ovr", "multiclassova"},
      {"binary", "binary"}, {"binary_logloss", "binary"}, {"binary:logistic", "binary"},
      {"cross_entropy", "cross_entropy"}, {"xentropy", "cross_entropy"},
      {"lambdarank", "lambdarank"}, {"rank_xendcg", "rank_xendcg"}, {"xendcg", "rank_xendcg"},
      {"xe_ndcg", "rank_xendcg"}, {"xe_ndcg_mart", "rank_xendcg"}, {"xendcg_mart", "rank_xendcg"},
      {"custom", "custom"}, {"none", "custom"}, {"null", "custom"}, {"na", "custom"},
  };
  auto it = m.find(name);
  if (it == m.end()) throw std::runtime_error("unknown objective \"" + name0 + "\"");
  return it->second;
}

std::string Config::canonical_metric(const std::string& name0) {
  std::string name = lower(name0);
  static const std::map<std::string, std::string> m = {
      {"l2", "l2"}, {"mean_squared_error", "l2"}, {"mse", "l2"}, {"regression", "l2"}, {"regression_l2", "l2"},
      {"rmse", "rmse"}, {"root_mean_squared_error", "rmse"}, {"l2_root", "rmse"},
      {"l1", "l1"}, {"mean_absolute_error", "l1"}, {"mae", "l1"}, {"regression_l1", "l1"},
      {"huber", "huber"}, {"fair", "fair"}, {"poisson", "poisson"},
      {"quantile", "quantile"}, {"mape", "mape"}, {"mean_absolute_percentage_error", "mape"},
      {"gamma", "gamma"}, {"gamma_deviance", "gamma_deviance"}, {"tweedie", "tweedie"},
      {"multi_logloss", "multi_logloss"}, {"multiclass", "multi_logloss"}, {"softmax", "multi_logloss"},
      {"multiclassova", "multi_logloss"}, {"multiclass_ova", "multi_logloss"}, {"ova", "multi_logloss"},
      {"ovr", "multi_logloss"}, {"multi_error", "multi_error"},
      {"binary_logloss", "binary_logloss"}, {"binary", "binary_logloss"}, {"logloss", "binary_logloss"},
      {"binary_error", "binary_error"}, {"error", "binary_error"},
      {"auc", "auc"},
      {"cross_entropy", "cross_entropy"}, {"xentropy", "cross_entropy"},
      {"ndcg", "ndcg"}, {"lambdarank", "ndcg"}, {"rank_xendcg", "ndcg"}, {"xendcg", "ndcg"}, {"xe_ndcg", "ndcg"},
      {"xe_ndcg_mart", "ndcg"}, {"xendcg_mart", "ndcg"},
      {"map", "map"}, {"mean_average_precision", "map"},
      {"average_precision", "average_precision"}, {"auc_mu", "auc_mu"},
      {"none", "none"}, {"null", "none"}, {"custom", "none"}, {"na", "none"}, {"", "none"},
  };
  auto it = m.find(name);
  if (it == m.end()) throw std::runtime_error("unknown metric \"" + name0 + "\"");
  return it->second;
}

std::vector<std::string> Config::resolved_metrics() const {
  std::vector<std::string> out;
  if (metric.empty()) {
    std::string o = canonical_objective(objective);
    if (o == "regression") out.push_back("l2");
    else if (o == "regression_l1") out.push_back("l1");
    else if (o == "huber") out.push_back("huber");
    else if (o == "fair") out.push_back("fair");
    else if (o == "poisson") out.push_back("poisson");
    else if (o == "quantile" || o == "mape" || o == "gamma" || o == "tweedie") out.push_back(o);
    else if (o == "multiclass" || o == "multiclassova") out.push_back("multi_logloss");
    else if (o == "binary") out.push_back("binary_logloss");
    else if (o == "cross_entropy") out.push_back("cross_entropy");
    else if (o == "lambdarank" || o == "rank_xendcg") out.push_back("ndcg");
    return out;
  }
  for (const auto& m : metric) {
    std::string c = canonical_metric(m);
    if (c == "none") continue;
    if (std::find(out.begin(), out.end(), c) == out.end()) out.push_back(c);
  }
  return out;
}

void Config::check() {
  // an explicit master seed settles every random stream that was not given a seed of its own
  if (std::find(explicit_keys.begin(), explicit_keys.end(), "seed") != explicit_keys.end()) {
    struct Derived { const char* name; int* value; };
    const Derived streams[] = {{"data_random_seed", &data_random_seed}, {"bagging_seed", &bagging_seed},
                               {"drop_seed", &drop_seed}, {"feature_fraction_seed", &feature_fraction_seed},
                               {"objective_seed", &objective_seed}, {"extra_seed", &extra_seed}};
    uint64_t x = (uint64_t)(uint32_t)seed * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull;
    for (const Derived& d : streams) {
      x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
      if (std::find(explicit_keys.begin(), explicit_keys.end(), d.name) == explicit_keys.end())
        *d.value = (int)(x & 0x7fffffffull);
    }
  }
  std::string b = lower(boosting);
  if (b == "gbrt" || b == "gbdt") boosting = "gbdt";
  else if (b == "rf" || b == "random_forest") boosting = "rf";
  else if (b == "goss") { boosting = "gbdt"; data_sample_strategy = "goss"; }   // the older spelling
  else if (b == "dart") boosting = "dart";
  else throw std::runtime_error("unknown boosting type \"" + boosting + "\" (supported: gbdt, rf, dart, goss)");
  {
    std::string d = lower(data_sample_strategy);
    if (d != "bagging" && d != "goss")
      throw std::runtime_error("unknown data_sample_strategy \"" + data_sample_strategy + "\" (supported: bagging, goss)");
    data_sample_strategy = d;
    if (d == "goss") {
      if (boosting == "rf") throw std::runtime_error("data_sample_strategy=goss cannot be combined with boosting=rf");
      if (!(top_rate > 0.0) || !(other_rate > 0.0) || top_rate + other_rate > 1.0)
        throw std::runtime_error("goss needs top_rate > 0, other_rate > 0 and top_rate + other_rate <= 1");
    }
  }
  {
    const std::string o = canonical_objective(objective);
    const bool multi = o == "multiclass" || o == "multiclassova";
    if (multi && num_class < 2) throw std::runtime_error("objective=" + o + " needs num_class >= 2");
    if (!multi && o != "custom" && num_class != 1)
      throw std::runtime_error("num_class must be 1 for objective=" + o);
  }
  for (double v : cegb_penalty_feature_lazy)
    if (v != 0.0) throw std::runtime_error("cegb_penalty_feature_lazy is not supported (split and coupled feature penalties are)");
  for (const auto& m : metric) (void)canonical_metric(m);
  std::string t = lower(tree_learner);
  if (t == "serial") tree_learner = "serial";
  else if (t == "feature" || t == "feature_parallel") tree_learner = "feature";
  else if (t == "data" || t == "data_parallel") tree_learner = "data";
  else if (t == "voting" || t == "voting_parallel") tree_learner = "voting";
  else throw std::runtime_error("unknown tree_learner \"" + tree_learner + "\"");
  {
    const std::string m = lower(monotone_constraints_method);
    if (m != "basic" && m != "intermediate" && m != "advanced")
      throw std::runtime_error("unknown monotone_constraints_method \"" + monotone_constraints_method + "\"");
    monotone_constraints_method = m;   // intermediate / advanced: served by the basic method (same guarantee, more conservative splits)
  }
  std::string d = lower(device_type);
  if (d == "cpu") device_type = "cpu";
  else if (d == "gpu" || d == "hip" || d == "cuda" || d == "rocm") device_type = "gpu";
  else throw std::runtime_error("unknown device_type \"" + device_type + "\"");
  // gpu_use_dp is upstream's switch for full-precision histograms on the device: here that is the 36-bit fixed-point
  // form, unless the caller settled the histogram precision explicitly
  if (gpu_use_dp && std::find(explicit_keys.begin(), explicit_keys.end(), "use_quantized_grad") == explicit_keys.end())
    use_quantized_grad = false;
  if (boosting == "rf") {
    bool row_sub = bagging_freq > 0 && (bagging_fraction < 1.0 || pos_bagging_fraction < 1.0 || neg_bagging_fraction < 1.0);
    bool col_sub = feature_fraction < 1.0 || feature_fraction_bynode < 1.0;
    if (!row_sub && !col_sub)
      throw std::runtime_error(
          "boosting=rf needs row sub-sampling (bagging_freq > 0 and bagging_fraction < 1) "
          "or column sub-sampling (feature_fraction < 1)");
  }
  if (max_depth > 0) {
    // a tree of depth d has at most 2^d leaves
    if (max_depth < 30) num_leaves = std::min(num_leaves, 1 << max_depth);
  }
}

std::string Config::to_model_string() const {
  std::ostringstream os;
  for (const auto& p : kParams) {
    os << "[" << p.name << ": ";
    switch (p.type) {
      case PType::Int: os << cfield<int>(this, p.offset); break;
      case PType::Double: os << fmt_double(cfield<double>(this, p.offset)); break;
      case PType::Bool: os << (cfield<bool>(this, p.offset) ? 1 : 0); break;
      case PType::String: os << cfield<std::string>(this, p.offset); break;
      case PType::StringList: {
        const auto& l = cfield<std::vector<std::string>>(this, p.offset);
        for (size_t i = 0; i < l.size(); ++i) os << (i ? "," : "") << l[i];
        break;
      }
      case PType::IntList: {
        const auto& l = cfield<std::vector<int>>(this, p.offset);
        for (size_t i = 0; i < l.size(); ++i) os << (i ? "," : "") << l[i];
        break;
      }
      case PType::DoubleList: {
        const auto& l = cfield<std::vector<double>>(this, p.offset);
        for (size_t i = 0; i < l.size(); ++i) os << (i ? "," : "") << fmt_double(l[i]);
        break;
      }
    }
    os << "]\n";
  }
  return os.str();
}

}  // namespace lga

#include "tree.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <sstream>
#include <stdexcept>

namespace lga {

void Tree::reset(int max_leaves) {
  num_leaves = 1;
  int m = std::max(max_leaves, 1);
  auto rs = [&](auto& v, size_t n) { v.clear(); v.reserve(n); };
  rs(split_feature, m); rs(split_feature_inner, m); rs(split_gain, m); rs(threshold, m); rs(threshold_bin, m);
  rs(decision_type, m); rs(left_child, m); rs(right_child, m); rs(internal_value, m); rs(internal_weight, m);
  rs(internal_count, m);
  leaf_value.assign(1, 0.0); leaf_weight.assign(1, 0.0); leaf_count.assign(1, 0);
  leaf_depth.assign(1, 0); leaf_parent.assign(1, -1);
  cat_boundaries.clear(); cat_threshold.clear(); cat_bins.clear();
  shrinkage = 1.0;
}

int Tree::split_categorical(int leaf, int feature_real, int feature_inner, int category_bin, int64_t category, double gain,
                            double left_value, double right_value, int64_t left_cnt, int64_t right_cnt,
                            double left_weight, double right_weight) {
  uint64_t bins[4] = {0, 0, 0, 0};
  bins[category_bin >> 6] |= 1ull << (category_bin & 63);
  const int right = split_categorical_set(leaf, feature_real, feature_inner, bins, std::vector<int64_t>(1, category), gain,
                                          left_value, right_value, left_cnt, right_cnt, left_weight, right_weight);
  threshold_bin[(size_t)(num_leaves - 2)] = category_bin;   // the node just created
  return right;
}

int Tree::split_categorical_set(int leaf, int feature_real, int feature_inner, const uint64_t* bins,
                                const std::vector<int64_t>& categories, double gain, double left_value, double right_value,
                                int64_t left_cnt, int64_t right_cnt, double left_weight, double right_weight) {
  if (categories.empty()) throw std::runtime_error("categorical split on an empty set");
  int64_t largest = 0;
  for (int64_t c : categories) {
    if (c < 0) throw std::runtime_error("categorical split on a negative category");
    largest = std::max(largest, c);
  }
  if (cat_boundaries.empty()) cat_boundaries.push_back(0);
  const int k = (int)cat_boundaries.size() - 1;
  const size_t words = (size_t)(largest >> 5) + 1;
  const size_t base = cat_threshold.size();
  cat_threshold.resize(base + words, 0u);
  for (int64_t c : categories) cat_threshold[base + (size_t)(c >> 5)] |= 1u << (c & 31);
  cat_boundaries.push_back((int32_t)cat_threshold.size());
  cat_bins.insert(cat_bins.end(), bins, bins + 4);
  // missing type NaN: unknown values fall to the right, like every value outside the set
  int right = split(leaf, feature_real, feature_inner, -1, (double)k, false, 2, gain, left_value, right_value,
                    left_cnt, right_cnt, left_weight, right_weight);
  decision_type[(size_t)(num_leaves - 2)] |= kCategoricalMask;
  return right;
}

std::vector<int64_t> Tree::node_categories(int node) const {
  std::vector<int64_t> out;
  if (!(decision_type[(size_t)node] & kCategoricalMask)) return out;
  const int k = (int)threshold[(size_t)node];
  for (int32_t w = cat_boundaries[(size_t)k]; w < cat_boundaries[(size_t)k + 1]; ++w)
    for (int b = 0; b < 32; ++b)
      if ((cat_threshold[(size_t)w] >> b) & 1u) out.push_back((int64_t)(w - cat_boundaries[(size_t)k]) * 32 + b);
  return out;
}

int Tree::split(int leaf, int feature_real, int feature_inner, int thr_bin, double thr, bool default_left,
                int missing_type, double gain, double left_value, double right_value, int64_t left_cnt,
                int64_t right_cnt, double left_weight, double right_weight) {
  int node = num_leaves - 1;       // new internal node
  int parent = leaf_parent[(size_t)leaf];
  if (parent >= 0) {
    if (left_child[(size_t)parent] == ~leaf) left_child[(size_t)parent] = node;
    else right_child[(size_t)parent] = node;
  }
  split_feature.push_back(feature_real);
  split_feature_inner.push_back(feature_inner);
  split_gain.push_back(gain);
  threshold.push_back(thr);
  threshold_bin.push_back(thr_bin);
  int8_t dt = 0;
  if (default_left) dt |= kDefaultLeftMask;
  dt |= (int8_t)((missing_type & 3) << 2);
  decision_type.push_back(dt);
  left_child.push_back(~leaf);
  right_child.push_back(~num_leaves);
  internal_value.push_back(leaf_value[(size_t)leaf]);
  internal_weight.push_back(leaf_weight[(size_t)leaf]);
  internal_count.push_back(left_cnt + right_cnt);
  int depth = leaf_depth[(size_t)leaf] + 1;
  // left child reuses the leaf slot
  leaf_value[(size_t)leaf] = left_value;
  leaf_weight[(size_t)leaf] = left_weight;
  leaf_count[(size_t)leaf] = left_cnt;
  leaf_depth[(size_t)leaf] = depth;
  leaf_parent[(size_t)leaf] = node;
  leaf_value.push_back(right_value);
  leaf_weight.push_back(right_weight);
  leaf_count.push_back(right_cnt);
  leaf_depth.push_back(depth);
  leaf_parent.push_back(node);
  return num_leaves++;
}

void Tree::apply_shrinkage(double rate) {
  for (auto& v : leaf_value) v *= rate;
  for (auto& v : internal_value) v *= rate;
  shrinkage *= rate;
}

void Tree::add_bias(double b) {
  for (auto& v : leaf_value) v += b;
  for (auto& v : internal_value) v += b;
}

int Tree::predict_leaf(const double* row) const {
  if (num_leaves <= 1) return 0;
  int node = 0;
  while (node >= 0) {
    double v = row[split_feature[(size_t)node]];
    node = decide(node, v) ? left_child[(size_t)node]
                                                                             : right_child[(size_t)node];
  }
  return ~node;
}

// ---------------------------------------------------------------------------
// Path-dependent SHAP values (Lundberg, Erion & Lee 2018, "Consistent individualized feature attribution
// for tree ensembles", algorithm 2).  A path keeps, for every distinct feature met on the way down, the
// fraction of "feature unknown" paths (by training row counts) and of "feature known" paths (1 on the
// branch the row takes, 0 on the other) that flow through it, plus the permutation weights per subset size.
// ---------------------------------------------------------------------------
namespace {
struct PathStep {
  int feature;
  double zero, one, weight;
};

void path_extend(std::vector<PathStep>& p, int d, double zero, double one, int feature) {
  p[(size_t)d] = PathStep{feature, zero, one, d == 0 ? 1.0 : 0.0};
  for (int i = d - 1; i >= 0; --i) {
    p[(size_t)i + 1].weight += one * p[(size_t)i].weight * (double)(i + 1) / (double)(d + 1);
    p[(size_t)i].weight = zero * p[(size_t)i].weight * (double)(d - i) / (double)(d + 1);
  }
}

void path_unwind(std::vector<PathStep>& p, int d, int k) {
  const double one = p[(size_t)k].one, zero = p[(size_t)k].zero;
  double next = p[(size_t)d].weight;
  for (int i = d - 1; i >= 0; --i) {
    if (one != 0.0) {
      const double keep = p[(size_t)i].weight;
      p[(size_t)i].weight = next * (double)(d + 1) / ((double)(i + 1) * one);
      next = keep - p[(size_t)i].weight * zero * (double)(d - i) / (double)(d + 1);
    } else {
      p[(size_t)i].weight = p[(size_t)i].weight * (double)(d + 1) / (zero * (double)(d - i));
    }
  }
  for (int i = k; i < d; ++i) {
    p[(size_t)i].feature = p[(size_t)i + 1].feature;
    p[(size_t)i].zero = p[(size_t)i + 1].zero;
    p[(size_t)i].one = p[(size_t)i + 1].one;
  }
}

double path_unwound_sum(const std::vector<PathStep>& p, int d, int k) {
  const double one = p[(size_t)k].one, zero = p[(size_t)k].zero;
  double next = p[(size_t)d].weight, total = 0.0;
  for (int i = d - 1; i >= 0; --i) {
    if (one != 0.0) {
      const double w = next * (double)(d + 1) / ((double)(i + 1) * one);
      total += w;
      next = p[(size_t)i].weight - w * zero * (double)(d - i) / (double)(d + 1);
    } else {
      total += p[(size_t)i].weight * (double)(d + 1) / (zero * (double)(d - i));
    }
  }
  return total;
}

struct ShapWalk {
  const Tree& t;
  const double* row;
  double* phi;
  double count_of(int child) const {
    return (double)(child < 0 ? t.leaf_count[(size_t)~child] : t.internal_count[(size_t)child]);
  }
  // `parent` holds d steps (indices 0..d-1); this node adds step d.
  void visit(int node, const std::vector<PathStep>& parent, int d, double zero, double one, int feature) const {
    std::vector<PathStep> p(parent.begin(), parent.begin() + d);
    p.resize((size_t)d + 2);
    path_extend(p, d, zero, one, feature);
    if (node < 0) {
      const double v = t.leaf_value[(size_t)~node];
      for (int i = 1; i <= d; ++i)
        phi[p[(size_t)i].feature] += path_unwound_sum(p, d, i) * (p[(size_t)i].one - p[(size_t)i].zero) * v;
      return;
    }
    const int f = t.split_feature[(size_t)node];
    const bool left = t.decide(node, row[f]);
    const int hot = left ? t.left_child[(size_t)node] : t.right_child[(size_t)node];
    const int cold = left ? t.right_child[(size_t)node] : t.left_child[(size_t)node];
    const double w = (double)t.internal_count[(size_t)node];
    const double hot_zero = w > 0 ? count_of(hot) / w : 0.5, cold_zero = w > 0 ? count_of(cold) / w : 0.5;
    double in_zero = 1.0, in_one = 1.0;
    int k = 1;
    for (; k <= d; ++k) if (p[(size_t)k].feature == f) break;
    if (k <= d) {   // the feature was split on above: fold that step into this one
      in_zero = p[(size_t)k].zero;
      in_one = p[(size_t)k].one;
      path_unwind(p, d, k);
      d -= 1;
    }
    visit(hot, p, d + 1, hot_zero * in_zero, in_one, f);
    visit(cold, p, d + 1, cold_zero * in_zero, 0.0, f);
  }
};
}  // namespace

double Tree::expected_value() const {
  if (num_leaves <= 1) return leaf_value[0];
  double s = 0.0, n = 0.0;
  for (int l = 0; l < num_leaves; ++l) {
    s += (double)leaf_count[(size_t)l] * leaf_value[(size_t)l];
    n += (double)leaf_count[(size_t)l];
  }
  return n > 0 ? s / n : 0.0;
}

void Tree::add_contributions(const double* row, int n_features, double* phi) const {
  phi[n_features] += expected_value();
  if (num_leaves <= 1) return;
  ShapWalk walk{*this, row, phi};
  std::vector<PathStep> empty;
  walk.visit(0, empty, 0, 1.0, 1.0, -1);
}

int Tree::max_depth() const {
  int d = 0;
  for (int v : leaf_depth) d = std::max(d, v);
  return d;
}

namespace {
std::string g17(double v) {
  char buf[64];
  std::snprintf(buf, sizeof(buf), "%.17g", v);
  return buf;
}
template <typename T, typename F>
void join(std::ostringstream& os, const char* key, const std::vector<T>& v, size_t n, F fmt) {
  os << key << "=";
  for (size_t i = 0; i < n; ++i) { if (i) os << " "; os << fmt(v[i]); }
  os << "\n";
}
std::vector<std::string> tokens(const std::string& s) {
  std::vector<std::string> out;
  std::istringstream is(s);
  std::string t;
  while (is >> t) out.push_back(t);
  return out;
}
}  // namespace

std::string Tree::to_string(int index) const {
  std::ostringstream os;
  os << "Tree=" << index << "\n";
  os << "num_leaves=" << num_leaves << "\n";
  os << "num_cat=" << num_cat() << "\n";
  size_t ni = (size_t)num_leaves - 1, nl = (size_t)num_leaves;
  auto id = [](auto v) { return std::to_string((long long)v); };
  auto dd = [](double v) { return g17(v); };
  if (num_leaves > 1) {
    join(os, "split_feature", split_feature, ni, id);
    join(os, "split_gain", split_gain, ni, [](double v) { char b[64]; std::snprintf(b, 64, "%g", v); return std::string(b); });
    join(os, "threshold", threshold, ni, dd);
    join(os, "decision_type", decision_type, ni, id);
    join(os, "left_child", left_child, ni, id);
    join(os, "right_child", right_child, ni, id);
    join(os, "leaf_value", leaf_value, nl, dd);
    join(os, "leaf_weight", leaf_weight, nl, dd);
    join(os, "leaf_count", leaf_count, nl, id);
    join(os, "internal_value", internal_value, ni, dd);
    join(os, "internal_weight", internal_weight, ni, dd);
    join(os, "internal_count", internal_count, ni, id);
    if (num_cat() > 0) {
      join(os, "cat_boundaries", cat_boundaries, cat_boundaries.size(), id);
      join(os, "cat_threshold", cat_threshold, cat_threshold.size(), id);
    }
  } else {
    join(os, "leaf_value", leaf_value, 1, dd);
  }
  os << "is_linear=0\n";
  os << "shrinkage=" << g17(shrinkage) << "\n";
  os << "\n\n";
  return os.str();
}

Tree Tree::from_string(const std::string& block) {
  std::map<std::string, std::string> kv;
  std::istringstream is(block);
  std::string line;
  while (std::getline(is, line)) {
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
    size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    kv[line.substr(0, eq)] = line.substr(eq + 1);
  }
  auto need = [&](const char* k) -> const std::string& {
    auto it = kv.find(k);
    if (it == kv.end()) throw std::runtime_error(std::string("model parse: tree block lacks \"") + k + "\"");
    return it->second;
  };
  Tree t;
  t.num_leaves = std::atoi(need("num_leaves").c_str());
  if (t.num_leaves < 1) throw std::runtime_error("model parse: num_leaves < 1");
  const int n_cat = kv.count("num_cat") ? std::atoi(kv["num_cat"].c_str()) : 0;
  if (n_cat < 0) throw std::runtime_error("model parse: num_cat < 0");
  size_t ni = (size_t)t.num_leaves - 1, nl = (size_t)t.num_leaves;
  auto ints = [&](const std::string& s, size_t n, auto& out, const char* k) {
    auto tk = tokens(s);
    if (tk.size() != n) throw std::runtime_error(std::string("model parse: wrong length of ") + k);
    out.resize(n);
    for (size_t i = 0; i < n; ++i) out[i] = (typename std::decay<decltype(out[0])>::type)std::strtoll(tk[i].c_str(), nullptr, 10);
  };
  auto dbls = [&](const std::string& s, size_t n, std::vector<double>& out, const char* k) {
    auto tk = tokens(s);
    if (tk.size() != n) throw std::runtime_error(std::string("model parse: wrong length of ") + k);
    out.resize(n);
    for (size_t i = 0; i < n; ++i) out[i] = std::strtod(tk[i].c_str(), nullptr);
  };
  dbls(need("leaf_value"), nl, t.leaf_value, "leaf_value");
  if (t.num_leaves > 1) {
    ints(need("split_feature"), ni, t.split_feature, "split_feature");
    dbls(need("threshold"), ni, t.threshold, "threshold");
    ints(need("decision_type"), ni, t.decision_type, "decision_type");
    ints(need("left_child"), ni, t.left_child, "left_child");
    ints(need("right_child"), ni, t.right_child, "right_child");
    if (kv.count("split_gain")) dbls(kv["split_gain"], ni, t.split_gain, "split_gain"); else t.split_gain.assign(ni, 0.0);
    if (kv.count("leaf_weight")) dbls(kv["leaf_weight"], nl, t.leaf_weight, "leaf_weight"); else t.leaf_weight.assign(nl, 0.0);
    if (kv.count("leaf_count")) ints(kv["leaf_count"], nl, t.leaf_count, "leaf_count"); else t.leaf_count.assign(nl, 0);
    if (kv.count("internal_value")) dbls(kv["internal_value"], ni, t.internal_value, "internal_value"); else t.internal_value.assign(ni, 0.0);
    if (kv.count("internal_weight")) dbls(kv["internal_weight"], ni, t.internal_weight, "internal_weight"); else t.internal_weight.assign(ni, 0.0);
    if (kv.count("internal_count")) ints(kv["internal_count"], ni, t.internal_count, "internal_count"); else t.internal_count.assign(ni, 0);
    if (n_cat > 0) {
      ints(need("cat_boundaries"), (size_t)n_cat + 1, t.cat_boundaries, "cat_boundaries");
      if (t.cat_boundaries[0] != 0) throw std::runtime_error("model parse: cat_boundaries must start at 0");
      for (int k = 0; k < n_cat; ++k)
        if (t.cat_boundaries[(size_t)k + 1] < t.cat_boundaries[(size_t)k]) throw std::runtime_error("model parse: cat_boundaries decrease");
      std::vector<int64_t> words;
      ints(need("cat_threshold"), (size_t)t.cat_boundaries.back(), words, "cat_threshold");
      t.cat_threshold.assign(words.begin(), words.end());
    }
    for (size_t i = 0; i < ni; ++i) {
      if (t.decision_type[i] & kCategoricalMask) {
        const double k = t.threshold[i];
        if (!(k >= 0.0) || k != (double)(int)k || (int)k >= n_cat)
          throw std::runtime_error("model parse: categorical node points outside cat_boundaries");
      }
      auto ok = [&](int c) { return (c >= 0 && c < (int)ni) || (c < 0 && ~c < (int)nl); };
      if (!ok(t.left_child[i]) || !ok(t.right_child[i])) throw std::runtime_error("model parse: child index out of range");
    }
  } else {
    t.leaf_weight.assign(1, 0.0);
    t.leaf_count.assign(1, 0);
  }
  if (kv.count("is_linear") && std::atoi(kv["is_linear"].c_str()) != 0)
    throw std::runtime_error("model parse: linear trees are not supported");
  t.shrinkage = kv.count("shrinkage") ? std::strtod(kv["shrinkage"].c_str(), nullptr) : 1.0;
  // derived, not serialised
  t.split_feature_inner.assign(ni, -1);
  t.threshold_bin.assign(ni, -1);
  t.leaf_depth.assign(nl, 0);
  t.leaf_parent.assign(nl, -1);
  if (t.num_leaves > 1) {
    std::vector<int> depth(ni, 0);
    for (size_t i = 0; i < ni; ++i) {
      for (int c : {t.left_child[i], t.right_child[i]}) {
        if (c >= 0) depth[(size_t)c] = depth[i] + 1;
        else { t.leaf_depth[(size_t)~c] = depth[i] + 1; t.leaf_parent[(size_t)~c] = (int)i; }
      }
    }
  }
  return t;
}

std::string Tree::to_json(int index) const {
  std::ostringstream os;
  os << "{\"tree_index\":" << index << ",\"num_leaves\":" << num_leaves << ",\"num_cat\":" << num_cat() << ",\"shrinkage\":" << g17(shrinkage)
     << ",\"tree_structure\":";
  // recursive lambda over nodes
  struct Rec {
    const Tree& t;
    std::ostringstream& os;
    void leaf(int l) {
      os << "{\"leaf_index\":" << l << ",\"leaf_value\":" << g17(t.leaf_value[(size_t)l])
         << ",\"leaf_weight\":" << g17(t.leaf_weight[(size_t)l]) << ",\"leaf_count\":" << t.leaf_count[(size_t)l] << "}";
    }
    void node(int n) {
      if (n < 0) { leaf(~n); return; }
      size_t i = (size_t)n;
      int mt = (t.decision_type[i] >> 2) & 3;
      const bool cat = (t.decision_type[i] & kCategoricalMask) != 0;
      os << "{\"split_index\":" << n << ",\"split_feature\":" << t.split_feature[i] << ",\"split_gain\":" << g17(t.split_gain[i])
         << ",\"threshold\":";
      if (cat) {
        // the categories that go left, "a||b||c"
        os << "\"";
        bool first = true;
        for (int64_t c : t.node_categories(n)) { os << (first ? "" : "||") << c; first = false; }
        os << "\"";
      } else {
        os << g17(t.threshold[i]);
      }
      os << ",\"decision_type\":\"" << (cat ? "==" : "<=") << "\",\"default_left\":"
         << ((t.decision_type[i] & kDefaultLeftMask) ? "true" : "false") << ",\"missing_type\":\""
         << (mt == 0 ? "None" : mt == 1 ? "Zero" : "NaN") << "\",\"internal_value\":" << g17(t.internal_value[i])
         << ",\"internal_weight\":" << g17(t.internal_weight[i]) << ",\"internal_count\":" << t.internal_count[i]
         << ",\"left_child\":";
      node(t.left_child[i]);
      os << ",\"right_child\":";
      node(t.right_child[i]);
      os << "}";
    }
  } rec{*this, os};
  if (num_leaves <= 1) rec.leaf(0); else rec.node(0);
  os << "}";
  return os.str();
}

}  // namespace lga

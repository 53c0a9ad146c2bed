// LightGBM text model format: writer and strict-but-tolerant parser.
// The reference throws its refit models away (r/gridsearchCV.R:127 "keep test
// predictions, no model"); keeping the upstream text format means boosters
// saved here remain loadable by LightGBM tooling and vice versa.
#include "gbdt.hpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <sstream>
#include <stdexcept>

namespace lga {

namespace {
std::vector<std::string> split_ws(const std::string& s) {
  std::vector<std::string> out;
  std::istringstream is(s);
  std::string t;
  while (is >> t) out.push_back(t);
  return out;
}
}  // namespace

std::string Booster::save_model_to_string(int start_iter, int num_iter, int importance_type) const {
  const int K = num_class_;
  int total = current_iteration();
  start_iter = std::max(0, std::min(start_iter, total));
  const int first_tree = start_iter * K;
  const int n_it = used_iterations(start_iter, num_iter) * K;   // trees written
  std::ostringstream os;
  os << "tree\n";
  os << "version=v4\n";
  os << "num_class=" << K << "\n";
  os << "num_tree_per_iteration=" << K << "\n";
  os << "label_index=" << label_index_ << "\n";
  os << "max_feature_idx=" << max_feature_idx_ << "\n";
  os << "objective=" << objective_str_ << "\n";
  if (average_output_) os << "average_output\n";
  os << "feature_names=";
  for (size_t i = 0; i < feature_names_.size(); ++i) os << (i ? " " : "") << feature_names_[i];
  os << "\n";
  os << "feature_infos=";
  for (size_t i = 0; i < feature_infos_.size(); ++i) os << (i ? " " : "") << feature_infos_[i];
  os << "\n";
  std::vector<std::string> blocks;
  blocks.reserve((size_t)n_it);
  for (int t = 0; t < n_it; ++t) blocks.push_back(trees_[(size_t)(first_tree + t)].to_string(t));
  os << "tree_sizes=";
  for (size_t i = 0; i < blocks.size(); ++i) os << (i ? " " : "") << blocks[i].size();
  os << "\n\n";
  for (const auto& b : blocks) os << b;
  os << "end of trees\n\n";
  // importances: descending, zero entries omitted
  std::vector<double> imp((size_t)(max_feature_idx_ + 1), 0.0);
  for (int t = 0; t < n_it; ++t) {
    const Tree& tr = trees_[(size_t)(first_tree + t)];
    for (int i = 0; i + 1 < tr.num_leaves; ++i)
      imp[(size_t)tr.split_feature[(size_t)i]] += importance_type == 0 ? 1.0 : tr.split_gain[(size_t)i];
  }
  std::vector<int> order;
  for (int f = 0; f <= max_feature_idx_; ++f) if (imp[(size_t)f] > 0.0) order.push_back(f);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return imp[(size_t)a] > imp[(size_t)b]; });
  os << "feature_importances:\n";
  for (int f : order) {
    os << feature_names_[(size_t)f] << "=";
    if (importance_type == 0) os << (long long)imp[(size_t)f]; else os << imp[(size_t)f];
    os << "\n";
  }
  os << "\nparameters:\n";
  if (engine_ || loaded_parameters_.empty()) os << cfg_.to_model_string();
  else os << loaded_parameters_;
  os << "end of parameters\n\n";
  os << "pandas_categorical:null\n";
  return os.str();
}

void Booster::load_model_from_string(const std::string& text) {
  std::istringstream is(text);
  std::string line;
  std::map<std::string, std::string> header;
  bool saw_tree_tag = false;
  average_output_ = false;
  std::vector<std::string> lines;
  while (std::getline(is, line)) {
    while (!line.empty() && (line.back() == '\r')) line.pop_back();
    lines.push_back(line);
  }
  size_t i = 0;
  // header: up to the first "Tree=" line or "end of trees"
  for (; i < lines.size(); ++i) {
    const std::string& l = lines[i];
    if (l.rfind("Tree=", 0) == 0 || l == "end of trees") break;
    if (l.empty()) continue;
    if (l == "tree") { saw_tree_tag = true; continue; }
    if (l == "average_output") { average_output_ = true; continue; }
    size_t eq = l.find('=');
    if (eq == std::string::npos) continue;   // tolerate unknown bare flags
    header[l.substr(0, eq)] = l.substr(eq + 1);
  }
  if (!saw_tree_tag && header.empty()) throw std::runtime_error("model parse: not a LightGBM text model");
  auto get = [&](const char* k, const char* dflt) -> std::string {
    auto it = header.find(k);
    if (it == header.end()) {
      if (!dflt) throw std::runtime_error(std::string("model parse: header lacks \"") + k + "\"");
      return dflt;
    }
    return it->second;
  };
  num_class_ = std::atoi(get("num_tree_per_iteration", "1").c_str());
  if (num_class_ < 1) throw std::runtime_error("model parse: num_tree_per_iteration < 1");
  if (std::atoi(get("num_class", "1").c_str()) != num_class_)
    throw std::runtime_error("model parse: num_class and num_tree_per_iteration differ");
  init_scores_.assign((size_t)num_class_, 0.0);
  label_index_ = std::atoi(get("label_index", "0").c_str());
  max_feature_idx_ = std::atoi(get("max_feature_idx", nullptr).c_str());
  objective_str_ = get("objective", "regression");
  feature_names_ = split_ws(get("feature_names", ""));
  feature_infos_ = split_ws(get("feature_infos", ""));
  if ((int)feature_names_.size() != max_feature_idx_ + 1) {
    feature_names_.resize((size_t)(max_feature_idx_ + 1));
    for (int f = 0; f <= max_feature_idx_; ++f)
      if (feature_names_[(size_t)f].empty()) feature_names_[(size_t)f] = "Column_" + std::to_string(f);
  }
  // objective string -> spec
  {
    auto tok = split_ws(objective_str_);
    std::string name = tok.empty() ? "regression" : tok[0];
    cfg_.objective = Config::canonical_objective(name) == "custom" ? "regression" : name;
    for (size_t k = 1; k < tok.size(); ++k) {
      if (tok[k].rfind("sigmoid:", 0) == 0) cfg_.sigmoid = std::strtod(tok[k].c_str() + 8, nullptr);
      if (tok[k].rfind("num_class:", 0) == 0) cfg_.num_class = std::atoi(tok[k].c_str() + 10);
      if (tok[k] == "sqrt") cfg_.reg_sqrt = true;
    }
    const std::string canon = Config::canonical_objective(name);
    if (canon == "multiclass" || canon == "multiclassova") {
      if (cfg_.num_class != num_class_) throw std::runtime_error("model parse: objective num_class differs from the header");
    } else {
      cfg_.num_class = num_class_;
    }
    std::string keep = objective_str_;
    setup_objective();
    objective_str_ = keep;
  }
  // trees
  trees_.clear();
  while (i < lines.size()) {
    const std::string& l = lines[i];
    if (l == "end of trees") { ++i; break; }
    if (l.rfind("Tree=", 0) == 0) {
      std::string block;
      size_t j = i + 1;
      for (; j < lines.size(); ++j) {
        if (lines[j].rfind("Tree=", 0) == 0 || lines[j] == "end of trees") break;
        block += lines[j];
        block += "\n";
      }
      trees_.push_back(Tree::from_string(block));
      const Tree& t = trees_.back();
      for (int s = 0; s + 1 < t.num_leaves; ++s)
        if (t.split_feature[(size_t)s] < 0 || t.split_feature[(size_t)s] > max_feature_idx_)
          throw std::runtime_error("model parse: split_feature out of range");
      i = j;
    } else {
      ++i;
    }
  }
  if (header.count("tree_sizes")) {
    size_t declared = split_ws(header["tree_sizes"]).size();
    if (declared != trees_.size())
      throw std::runtime_error("model parse: tree_sizes lists " + std::to_string(declared) + " trees, found " +
                               std::to_string(trees_.size()));
  }
  if (trees_.size() % (size_t)num_class_ != 0)
    throw std::runtime_error("model parse: the number of trees is not a multiple of num_tree_per_iteration");
  // parameters block (kept verbatim for re-saving)
  loaded_parameters_.clear();
  bool in_params = false;
  for (; i < lines.size(); ++i) {
    if (lines[i] == "parameters:") { in_params = true; continue; }
    if (lines[i] == "end of parameters") break;
    if (in_params) {
      loaded_parameters_ += lines[i] + "\n";
      // "[name: value]" -> config, best effort
      const std::string& l = lines[i];
      if (l.size() > 2 && l.front() == '[' && l.back() == ']') {
        size_t c = l.find(": ");
        if (c != std::string::npos) {
          std::string k = l.substr(1, c - 1), v = l.substr(c + 2, l.size() - c - 3);
          if (!v.empty() && k != "objective" && k != "metric") {
            try { cfg_.set(k + "=" + v); } catch (const std::exception&) {}
          }
        }
      }
    }
  }
  is_rf_ = average_output_;
}

std::string Booster::dump_json(int start_iter, int num_iter) const {
  const int K = num_class_;
  int total = current_iteration();
  start_iter = std::max(0, std::min(start_iter, total));
  const int n_it = used_iterations(start_iter, num_iter) * K;   // trees written
  start_iter *= K;                                              // first tree
  std::ostringstream os;
  os << "{\"name\":\"tree\",\"version\":\"v4\",\"num_class\":" << K << ",\"num_tree_per_iteration\":" << K
     << ",\"label_index\":" << label_index_
     << ",\"max_feature_idx\":" << max_feature_idx_ << ",\"objective\":\"" << objective_str_ << "\",\"average_output\":"
     << (average_output_ ? "true" : "false") << ",\"feature_names\":[";
  for (size_t i = 0; i < feature_names_.size(); ++i) os << (i ? "," : "") << "\"" << feature_names_[i] << "\"";
  os << "],\"tree_info\":[";
  for (int t = 0; t < n_it; ++t) os << (t ? "," : "") << trees_[(size_t)(start_iter + t)].to_json(t);
  os << "]}";
  return os.str();
}

}  // namespace lga

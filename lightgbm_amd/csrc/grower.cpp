#include "grower.hpp"
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <queue>
#include <functional>
#include <stdexcept>
#include <thread>

namespace lga {

TreeGrower::TreeGrower(const Config& cfg, const Dataset* ds) : cfg_(cfg), ds_(ds) {
  // What a wasted split costs grows with the width of the data - a histogram per feature is built, stored and searched -
  // so beyond one 32-feature tile small leaves are speculated on as strictly as large ones (measured at 255 leaves:
  // 2.27 M x 137 3.51 -> 3.18 ms per tree, 1 M x 968 8.65 -> 7.81, with the same number of rounds).
  if (ds_->num_features() > kTileFeatures) spec_lo_ = spec_hi_;
  if (const char* e = std::getenv("LGA_SPEC_FLOOR")) spec_lo_ = spec_hi_ = std::atof(e);
  if (const char* e = std::getenv("LGA_SPEC_LO")) spec_lo_ = std::atof(e);
  if (const char* e = std::getenv("LGA_SPEC_HI")) spec_hi_ = std::atof(e);
  if (const char* e = std::getenv("LGA_SPEC_RATIO")) spec_ratio_ = std::atoi(e) != 0;
  if (const char* e = std::getenv("LGA_CERTAIN_SPLITS")) certain_splits_ = std::atoi(e) != 0;
  if (const char* e = std::getenv("LGA_FILL_ROWS")) fill_rows_ = std::max(0.0, std::atof(e));
  if (const char* e = std::getenv("LGA_FILL_LEAF_ROWS")) fill_leaf_rows_ = std::max(0.0, std::atof(e));
  if (const char* e = std::getenv("LGA_OWN_ROUND_ROWS")) own_round_rows_ = std::max(1.0, std::atof(e));
  if (const char* e = std::getenv("LGA_SPEC_ROWS_LO")) spec_rows_lo_ = std::max(1.0, std::atof(e));
  if (const char* e = std::getenv("LGA_SPEC_ROWS_HI")) spec_rows_hi_ = std::max(spec_rows_lo_ * 2.0, std::atof(e));
  {
    const int F = ds_->num_features(), total = ds_->num_total_features();
    auto sized = [&](const std::vector<double>& v, const char* name) {
      if (!v.empty() && (int)v.size() != total)
        throw std::runtime_error(std::string(name) + " lists " + std::to_string(v.size()) + " values, the data has " +
                                 std::to_string(total) + " features");
      return !v.empty();
    };
    const bool contri = sized(cfg_.feature_contri, "feature_contri");
    const bool coupled = sized(cfg_.cegb_penalty_feature_coupled, "cegb_penalty_feature_coupled");
    if (contri || coupled || cfg_.cegb_penalty_split > 0.0) {
      gain_scale_.assign((size_t)std::max(F, 1), 1.0);
      gain_offset_.assign((size_t)std::max(F, 1), 0.0);
      coupled_penalty_.assign((size_t)std::max(F, 1), 0.0);
      for (int f = 0; f < F; ++f) {
        if (contri) gain_scale_[(size_t)f] = cfg_.feature_contri[(size_t)ds_->real_feature(f)];
        if (coupled) coupled_penalty_[(size_t)f] = cfg_.cegb_penalty_feature_coupled[(size_t)ds_->real_feature(f)];
      }
      refresh_gain_adjustment();
    }
  }
  if (!cfg_.forcedsplits_filename.empty()) load_forced_splits(cfg_.forcedsplits_filename);
  // interaction_constraints = "[0,1,2],[2,3]": groups of raw-matrix columns; columns without a usable
  // feature (constant ones) drop out
  const std::string& ic = cfg_.interaction_constraints;
  if (!ic.empty()) {
    const int F = ds_->num_features(), total = ds_->num_total_features();
    mask_words_ = (F + 63) / 64;
    std::vector<uint64_t> cur;
    bool open = false;
    size_t i = 0;
    auto bad = [&]() { throw std::runtime_error("interaction_constraints: expected a list like [0,1,2],[2,3], got \"" + ic + "\""); };
    while (i < ic.size()) {
      const char ch = ic[i];
      if (ch == '[') {
        if (open) bad();
        open = true;
        cur.assign((size_t)std::max(mask_words_, 1), 0ull);
        ++i;
      } else if (ch == ']') {
        if (!open) bad();
        open = false;
        groups_.push_back(cur);
        ++i;
      } else if (ch == ',' || ch == ' ') {
        ++i;
      } else if (ch >= '0' && ch <= '9') {
        if (!open) bad();
        long v = 0;
        while (i < ic.size() && ic[i] >= '0' && ic[i] <= '9') { v = v * 10 + (ic[i] - '0'); ++i; if (v > 1000000000) bad(); }
        if (v >= total) throw std::runtime_error("interaction_constraints: feature " + std::to_string(v) + " does not exist");
        const int inner = ds_->inner_feature((int)v);
        if (inner >= 0) cur[(size_t)(inner >> 6)] |= 1ull << (inner & 63);
      } else {
        bad();
      }
    }
    if (open || groups_.empty()) bad();
  }
}

namespace {
// The forced-splits file is one JSON object per node: {"feature": 3, "threshold": 0.5, "left": {...}, "right": {...}}
// ("left" / "right" optional).  A reader for exactly that shape (numbers, nested objects; anything else is an error).
struct ForcedJson {
  const std::string& t;
  size_t i = 0;
  explicit ForcedJson(const std::string& text) : t(text) {}
  [[noreturn]] void bad(const char* what) const {
    throw std::runtime_error(std::string("forcedsplits_filename: ") + what + " near offset " + std::to_string(i));
  }
  void ws() { while (i < t.size() && (t[i] == ' ' || t[i] == '\n' || t[i] == '\t' || t[i] == '\r')) ++i; }
  bool eat(char c) { ws(); if (i < t.size() && t[i] == c) { ++i; return true; } return false; }
  std::string key() {
    ws();
    if (i >= t.size() || t[i] != '"') bad("expected a quoted key");
    const size_t e = t.find('"', i + 1);
    if (e == std::string::npos) bad("unterminated string");
    std::string k = t.substr(i + 1, e - i - 1);
    i = e + 1;
    return k;
  }
  double number() {
    ws();
    char* end = nullptr;
    const double v = std::strtod(t.c_str() + i, &end);
    if (end == t.c_str() + i) bad("expected a number");
    i = (size_t)(end - t.c_str());
    return v;
  }
};
}  // namespace

void TreeGrower::load_forced_splits(const std::string& path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("forcedsplits_filename: cannot read " + path);
  std::stringstream buf;
  buf << in.rdbuf();
  const std::string text = buf.str();
  ForcedJson js(text);
  // recursive descent over nested objects; returns the node's index in forced_ (-1 for an empty object)
  std::function<int(int)> node = [&](int depth) -> int {
    if (depth > 64) js.bad("nesting too deep");
    if (!js.eat('{')) js.bad("expected {");
    ForcedSplit f;
    bool has_feature = false, has_threshold = false;
    const int me = (int)forced_.size();
    forced_.push_back(f);
    if (!js.eat('}')) {
      do {
        const std::string k = js.key();
        if (!js.eat(':')) js.bad("expected :");
        if (k == "feature") { f.feature = (int)js.number(); has_feature = true; }
        else if (k == "threshold") { f.threshold = js.number(); has_threshold = true; }
        else if (k == "left") f.left = node(depth + 1);
        else if (k == "right") f.right = node(depth + 1);
        else js.bad("unknown key (feature, threshold, left, right)");
      } while (js.eat(','));
      if (!js.eat('}')) js.bad("expected }");
    }
    if (!has_feature || !has_threshold) {
      if (has_feature != has_threshold) js.bad("a node needs both feature and threshold");
      f.feature = -1;   // an empty object: nothing forced below here
    }
    if (f.feature >= ds_->num_total_features()) throw std::runtime_error("forcedsplits_filename: feature " + std::to_string(f.feature) + " does not exist");
    forced_[(size_t)me] = f;
    return f.feature < 0 ? -1 : me;
  };
  const int root = node(0);
  js.ws();
  if (js.i != text.size()) js.bad("trailing text");
  if (root < 0) forced_.clear();
}

void TreeGrower::refresh_gain_adjustment() {
  for (size_t f = 0; f < gain_offset_.size(); ++f) gain_offset_[f] = cfg_.cegb_tradeoff * coupled_penalty_[f];
}

SplitParams TreeGrower::split_params() const {
  SplitParams p;
  p.feature_scale = gain_scale_.empty() ? nullptr : gain_scale_.data();
  p.feature_offset = gain_scale_.empty() ? nullptr : gain_offset_.data();
  p.split_penalty = cfg_.cegb_tradeoff * cfg_.cegb_penalty_split;
  p.lambda_l1 = cfg_.lambda_l1;
  p.lambda_l2 = cfg_.lambda_l2;
  p.min_gain_to_split = cfg_.min_gain_to_split;
  p.min_sum_hessian_in_leaf = cfg_.min_sum_hessian_in_leaf;
  p.max_delta_step = cfg_.max_delta_step;
  p.min_data_in_leaf = cfg_.min_data_in_leaf;
  p.max_depth = cfg_.max_depth;
  p.bynode_fraction = cfg_.feature_fraction_bynode;
  p.node_bounds = nullptr;
  p.node_masks = nullptr;
  p.mask_words = 0;
  p.path_smooth = cfg_.path_smooth;
  p.extra_trees = cfg_.extra_trees ? 1 : 0;
  p.extra_seed = mix64((uint64_t)cfg_.extra_seed * 0x9e3779b97f4a7c15ull + (uint64_t)tree_index_ + 0x7e57ull);
  p.bynode_seed = mix64((uint64_t)cfg_.feature_fraction_seed * 0x9e3779b97f4a7c15ull + (uint64_t)tree_index_ + 0x51ed27ull);
  return p;
}

namespace {
struct Entry {
  double gain;
  int seq;    // push order of the replay: ties go to the earlier one
  int node;
};
struct EntryLess {
  bool operator()(const Entry& a, const Entry& b) const {
    if (a.gain != b.gain) return a.gain < b.gain;
    return a.seq > b.seq;
  }
};
inline bool splittable(const SplitInfo& s) { return s.feature >= 0 && std::isfinite(s.gain); }
}  // namespace

namespace {
// LGA_GROW_TRACE=1: host-side time of grow() outside the backend calls, printed every 100 trees.
struct GrowTrace {
  bool on = std::getenv("LGA_GROW_TRACE") != nullptr;
  double total_us = 0, backend_us = 0, tail_us = 0;
  double rows_split = 0, rows_used = 0;   // rows of all partitioned leaves / of the leaves whose split was used
  double rounds = 0, tasks = 0, est_ratio = 0;   // est_ratio: the tree's real cutoff gain over the estimate it started with
  int trees = 0;
  using clock = std::chrono::steady_clock;
  static double us(clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
  void report() {
    if (++trees % 100 == 0) {
      std::fprintf(stderr, "[grow trace] per tree: total %.1f us, in backend %.1f us, host %.1f us (of which after the last round %.1f us); "
                   "rows partitioned %.0f, of used splits %.0f; rounds %.1f, splits computed %.0f, cutoff / estimate %.3f\n",
                   total_us / 100, backend_us / 100, (total_us - backend_us) / 100, tail_us / 100, rows_split / 100, rows_used / 100,
                   rounds / 100, tasks / 100, est_ratio / 100);
      total_us = backend_us = tail_us = rows_split = rows_used = rounds = tasks = est_ratio = 0;
    }
  }
};
GrowTrace g_trace;
}  // namespace

Tree TreeGrower::grow(TreeBackend& be, const std::vector<FeatureMeta>& meta, std::vector<LeafPiece>* pieces) {
  if (grow_begin(be, meta)) be.whole_tree_launch();
  return grow_finish(be, meta, pieces);
}

// A backend that can grow a whole tree by itself (TreeBackend::whole_tree_begin) is asked to whenever the tree is
// a plain one: no output ranges (monotone constraints, path smoothing) and no interaction constraints, which the
// replay below hands down node by node.
bool TreeGrower::plain_trees(const std::vector<FeatureMeta>& meta) const {
  bool any_monotone = false;
  for (const FeatureMeta& m : meta) any_monotone = any_monotone || (m.enabled && m.monotone != 0);
  return groups_.empty() && forced_.empty() && !any_monotone && !(cfg_.path_smooth > 0.0) && cfg_.max_batch_leaves <= 0;
}

bool TreeGrower::grow_begin(TreeBackend& be, const std::vector<FeatureMeta>& meta) {
  sp_cur_ = split_params();
  be.begin_tree(meta, sp_cur_);
  whole_ = plain_trees(meta) && be.whole_tree_begin(cfg_.num_leaves - 1);
  return whole_;
}

Tree TreeGrower::assemble_whole(TreeBackend& be, const std::vector<FeatureMeta>& meta, const TreeBackend::WholeTree& wt,
                                QuantScale qs, std::vector<LeafPiece>* pieces) {
  sp_cur_ = split_params();
  whole_ = true;
  given_ = &wt;
  given_qs_ = qs;
  Tree t = grow_finish(be, meta, pieces);
  given_ = nullptr;
  return t;
}

Tree TreeGrower::grow_finish(TreeBackend& be, const std::vector<FeatureMeta>& meta, std::vector<LeafPiece>* pieces) {
  const auto t_enter = GrowTrace::clock::now();
  auto t_last_backend = t_enter;
  double backend_us = 0;
  stats_ = GrowStats();
  const SplitParams sp = sp_cur_;
  const int capacity_splits = (be.node_capacity() - 1) / 2;
  const int budget_total = cfg_.num_leaves - 1;
  if (capacity_splits < budget_total) throw std::runtime_error("backend node capacity below num_leaves");

  std::vector<Node> nodes;
  std::vector<int> chosen;   // node ids in strict best-first order
  int materialized = 0;
  double bar_used = 0.0;
  bool node_info = false;    // nodes carry an output range and their own output
  if (whole_) {
    // the backend replayed best-first growth itself: the k-th split's children are nodes 2k+1 and 2k+2
    TreeBackend::WholeTree own;
    if (!given_) be.whole_tree_finish(&own);
    const TreeBackend::WholeTree& wt = given_ ? *given_ : own;
    if (g_trace.on) { t_last_backend = GrowTrace::clock::now(); backend_us += GrowTrace::us(t_enter, t_last_backend); }
    const size_t n_splits = wt.split_node.size();
    nodes.resize(2 * n_splits + 1);
    for (size_t i = 0; i < nodes.size(); ++i) nodes[i].best = wt.best[i];
    for (size_t k = 0; k < n_splits; ++k) {
      const int p = wt.split_node[k];
      Node& n = nodes[(size_t)p];
      n.left = (int)(2 * k + 1);
      n.right = (int)(2 * k + 2);
      n.ready = true;
      for (int c : {n.left, n.right}) { nodes[(size_t)c].parent = p; nodes[(size_t)c].depth = n.depth + 1; }
      chosen.push_back(p);
    }
    materialized = (int)n_splits;
    stats_.rounds = 1;
    prev_root_gain_ = splittable(nodes[0].best) ? nodes[0].best.gain : 0.0;
  } else {
  nodes.reserve((size_t)be.node_capacity());
  nodes.emplace_back();
  // Interaction constraints: a node may be split on the features already on its branch and on those of every
  // group that holds all of them; the root on any feature of any group.
  const bool constrained = !groups_.empty();
  const size_t W = (size_t)mask_words_;
  std::vector<uint64_t> path_bits;   // [node][W]: features on the branch down to the node
  std::vector<uint64_t> allowed(W);
  auto allowed_for = [&](const uint64_t* path) {
    for (size_t w = 0; w < W; ++w) allowed[w] = path[w];
    for (const auto& g : groups_) {
      bool holds_all = true;
      for (size_t w = 0; w < W; ++w) holds_all = holds_all && (path[w] & ~g[w]) == 0;
      if (holds_all)
        for (size_t w = 0; w < W; ++w) allowed[w] |= g[w];
    }
  };
  if (constrained) {
    path_bits.assign(W, 0ull);
    allowed_for(path_bits.data());
    be.set_node_mask(0, allowed.data());
  }
  be.root_round(&nodes[0].best);
  if (g_trace.on) { t_last_backend = GrowTrace::clock::now(); backend_us += GrowTrace::us(t_enter, t_last_backend); }
  stats_.rounds = 1;
  const QuantScale qs = be.quant_scale();
  const double gunit = unit_of(qs.kg), hunit = unit_of(qs.kh);
  bool any_monotone = false;
  for (const FeatureMeta& m : meta) any_monotone = any_monotone || (m.enabled && m.monotone != 0);
  const bool smooth = sp.path_smooth > 0.0;
  node_info = any_monotone || smooth;
  {
    const SplitInfo& r = nodes[0].best;
    nodes[0].out = leaf_output((double)r.total_g * gunit, (double)r.total_h * hunit, sp.lambda_l1, sp.lambda_l2, sp.max_delta_step);
  }
  // outputs of the two children of `n`'s best split as the scan valued them: smoothed towards n's output
  // (with the scan's row estimates), inside n's range
  auto child_outputs = [&](const Node& n, double* lv, double* rv) {
    const SplitInfo& s = n.best;
    const double lg = (double)s.left_g * gunit, lh = (double)s.left_h * hunit;
    const double rg = (double)(s.total_g - s.left_g) * gunit, rh = (double)(s.total_h - s.left_h) * hunit;
    const double l2 = sp.lambda_l2 + (s.default_left == kCategoricalSetTask ? cfg_.cat_l2 : 0.0);   // as the search valued it
    double l = leaf_output(lg, lh, sp.lambda_l1, l2, sp.max_delta_step);
    double r = leaf_output(rg, rh, sp.lambda_l1, l2, sp.max_delta_step);
    if (smooth) {
      l = smoothed_output(l, s.left_cnt, n.out, sp.path_smooth);
      r = smoothed_output(r, s.total_cnt - s.left_cnt, n.out, sp.path_smooth);
    }
    *lv = clamp_output(l, n.lo, n.hi);
    *rv = clamp_output(r, n.lo, n.hi);
  };

  auto depth_blocked = [&](int depth) { return cfg_.max_depth > 0 && depth >= cfg_.max_depth; };
  if (depth_blocked(0)) nodes[0].best.feature = -1;
  // Forced splits (forcedsplits_filename): a node that has one takes it instead of the split the search found and
  // goes first in the replay (kForcedGain); where the left sums of a searched split would come from the histogram
  // scan, a forced split's are its left child's totals, known once the children have been searched.  A forced
  // split that cannot be made (feature unused or categorical, threshold beyond the last value bin, depth limit)
  // ends the forcing on its branch.
  constexpr double kForcedGain = 1e300;
  if (!forced_.empty() && node_info)
    throw std::runtime_error("forcedsplits_filename cannot be combined with monotone constraints or path smoothing");
  const double searched_root_gain = splittable(nodes[0].best) ? nodes[0].best.gain : 0.0;
  auto apply_forced = [&](int id) {
    Node& n = nodes[(size_t)id];
    if (n.forced_index < 0 || depth_blocked(n.depth) || n.best.total_cnt < 2) { n.forced_index = -1; return; }
    const ForcedSplit& F = forced_[(size_t)n.forced_index];
    const int inner = F.feature >= 0 ? ds_->inner_feature(F.feature) : -1;
    if (inner < 0 || meta[(size_t)inner].categorical) { n.forced_index = -1; return; }
    const FeatureMeta& m = meta[(size_t)inner];
    const int value_bins = m.missing_bin >= 0 ? m.num_bin - 1 : m.num_bin;
    const int thr = ds_->inner_mapper(inner).value_to_bin(F.threshold);
    if (thr < 0 || thr >= value_bins - 1) { n.forced_index = -1; return; }   // every value would go left
    SplitInfo& b = n.best;
    b.feature = inner;
    b.threshold = thr;
    b.default_left = 0;
    b.gain = kForcedGain;
    b.left_cnt = b.total_cnt / 2;   // placeholders until the children are known
    b.left_g = b.total_g / 2;
    b.left_h = b.total_h / 2;
    n.forced = true;
  };
  if (!forced_.empty()) { nodes[0].forced_index = 0; apply_forced(0); }

  std::priority_queue<Entry, std::vector<Entry>, EntryLess> frontier;
  int seq = 0;
  auto push = [&](int node) {
    const SplitInfo& s = nodes[(size_t)node].best;
    if (splittable(s)) frontier.push(Entry{s.gain, seq, node});
    ++seq;
  };
  push(0);

  int budget = budget_total;
  const int max_flight = cfg_.max_batch_leaves > 0 ? 1 : std::max(1, be.max_in_flight());
  struct Flight {
    int ticket;
    std::vector<SplitTask> tasks;
  };
  std::vector<Flight> flights;           // oldest first
  std::vector<SplitInfo> infos;
  // Which leaves are worth splitting before the replay asks for them?  A leaf that enters the frontier after k
  // splits have been chosen gets its turn iff its gain exceeds the smallest frontier maximum of the remaining
  // choices.  The previous tree's sequence of chosen gains gives that bar as a function of k (its suffix
  // minima, prev_bar_); residuals - and with them all gains - shrink from tree to tree, fast in the first
  // iterations, so the bar is carried over in proportion to the root's best gain, the one gain of this tree
  // that is known from the start.  What a wrong guess costs depends on the leaf: partitioning a leaf of a
  // million rows for nothing costs about as much device time as a whole extra round, a leaf of a few thousand
  // rows next to nothing, so the bar is scaled by a factor that rises with the row count, spec_lo_ .. spec_hi_.
  double bar_scale = 1.0;
  const double root_gain = searched_root_gain;
  if (spec_ratio_ && prev_root_gain_ > 0.0 && root_gain > 0.0) {
    bar_scale = root_gain / prev_root_gain_;
    bar_scale = bar_scale < 0.25 ? 0.25 : (bar_scale > 1.5 ? 1.5 : bar_scale);
  }
  prev_root_gain_ = root_gain;
  const double log_lo = std::log(spec_rows_lo_ * row_shards_), log_span = std::log(spec_rows_hi_ * row_shards_) - log_lo;
  // n_chosen: splits chosen (really or virtually) before the leaf is considered
  auto floor_gain_for = [&](int64_t rows, int n_chosen) {
    if (prev_bar_.empty()) return 0.0;
    const double bar = prev_bar_[(size_t)std::min<int>(std::max(n_chosen, 0), (int)prev_bar_.size() - 1)] * bar_scale;
    double t = rows <= 1 ? 0.0 : (std::log((double)rows) - log_lo) / log_span;
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    return (spec_lo_ + (spec_hi_ - spec_lo_) * t) * bar;
  };

  const bool dump_rounds = g_trace.on && std::getenv("LGA_GROW_DUMP") && tree_index_ == std::atoll(std::getenv("LGA_GROW_DUMP"));
  auto submit = [&](std::vector<SplitTask>&& tasks) {
    Flight f;
    f.tasks = std::move(tasks);
    if (dump_rounds) {
      int64_t most = 0, sum = 0, hist = 0;
      for (const SplitTask& t : f.tasks) { const int64_t r = nodes[(size_t)t.parent_node].best.total_cnt; most = std::max(most, r); sum += r; hist += t.hist_rows; }
      std::fprintf(stderr, "[grow dump] round %d: %zu leaves, %lld rows (largest %lld), %lld rows to histogram, %zu rounds already in flight, %zu splits chosen, at %.1f us\n",
                   stats_.rounds, f.tasks.size(), (long long)sum, (long long)most, (long long)hist, flights.size(), chosen.size(),
                   GrowTrace::us(t_enter, GrowTrace::clock::now()));
    }
    const auto t_call = GrowTrace::clock::now();
    f.ticket = be.submit_round(f.tasks);
    if (g_trace.on) { t_last_backend = GrowTrace::clock::now(); backend_us += GrowTrace::us(t_call, t_last_backend); }
    ++stats_.rounds;
    flights.push_back(std::move(f));
  };
  // the next round to replay: the first one found finished, the oldest when polling has gone on for long (its
  // blocking wait notices a failed launch)
  auto wait_oldest = [&]() {
    size_t pick = 0;
    if (flights.size() > 1) {
      bool found = false;
      for (unsigned spins = 0; !found && spins < 4000000u; ++spins) {
        for (size_t i = 0; i < flights.size() && !found; ++i)
          if (be.round_done(flights[i].ticket)) { pick = i; found = true; }
        if (!found && spins > 20000u) std::this_thread::yield();
      }
    }
    Flight f = std::move(flights[pick]);
    flights.erase(flights.begin() + (std::ptrdiff_t)pick);
    infos.assign(f.tasks.size() * 2, SplitInfo{});
    const auto t_call = GrowTrace::clock::now();
    be.wait_round(f.ticket, infos.data());
    if (g_trace.on) { t_last_backend = GrowTrace::clock::now(); backend_us += GrowTrace::us(t_call, t_last_backend); }
    for (size_t i = 0; i < f.tasks.size(); ++i) {
      for (int side = 0; side < 2; ++side) {
        int id = side == 0 ? f.tasks[i].left_node : f.tasks[i].right_node;
        nodes[(size_t)id].best = infos[i * 2 + (size_t)side];
        if (depth_blocked(nodes[(size_t)id].depth)) nodes[(size_t)id].best.feature = -1;
      }
      Node& parent = nodes[(size_t)f.tasks[i].parent_node];
      if (parent.forced) {
        const SplitInfo& l = nodes[(size_t)f.tasks[i].left_node].best;
        parent.best.left_g = l.total_g; parent.best.left_h = l.total_h; parent.best.left_cnt = l.total_cnt;
        const ForcedSplit& F = forced_[(size_t)parent.forced_index];
        nodes[(size_t)f.tasks[i].left_node].forced_index = F.left;
        nodes[(size_t)f.tasks[i].right_node].forced_index = F.right;
        apply_forced(f.tasks[i].left_node);
        apply_forced(f.tasks[i].right_node);
      }
      nodes[(size_t)f.tasks[i].parent_node].ready = true;
    }
  };

  while (true) {
    // 1. the real replay: strict best-first over leaves whose children are known
    bool stalled_on_flight = false;
    while (budget > 0 && !frontier.empty()) {
      const Entry top = frontier.top();
      Node& n = nodes[(size_t)top.node];
      if (n.left < 0) break;                              // not partitioned yet
      if (!n.ready) { stalled_on_flight = true; break; }  // its round is still in flight
      frontier.pop();
      chosen.push_back(top.node);
      --budget;
      push(n.left);
      push(n.right);
    }
    if (budget == 0 || frontier.empty()) break;

    // 2. the virtual replay: what would the replay reach with everything known so far?  Partitioned
    //    leaves hand over their children; every leaf reached that is not partitioned yet becomes a task.
    std::vector<SplitTask> tasks;
    if ((int)flights.size() < max_flight) {
      // Capacity rule: after any submission there must be one free split per remaining unit of budget
      // (every remaining pick may need a fresh partition).  A round that contains the stalled leaf gets
      // that leaf picked as soon as it is back, so it may use the last free split; a purely speculative
      // round may not.
      const int pool_room = capacity_splits - materialized - (budget - 1);
      const bool top_needs_task = nodes[(size_t)frontier.top().node].left < 0;
      int max_new = std::min(budget, top_needs_task ? pool_room : pool_room - 1);
      if (max_new < 1 && flights.empty()) throw std::runtime_error("tree grower ran out of node capacity");
      if (cfg_.max_batch_leaves > 0) max_new = std::min(max_new, cfg_.max_batch_leaves);
      auto virt = frontier;
      int vseq = seq;
      int vbudget = budget;
      // Splits that are CERTAIN to get their turn need no gain estimate: the subtree of a leaf cannot take more
      // splits than its rows and its hessian sum leave room for (min_data_in_leaf, min_sum_hessian_in_leaf), so a
      // leaf is split for sure if the budget outlasts everything that could be chosen ahead of it - the known
      // candidates with larger gains (the virtual replay pops them first) and all that the leaves among them whose
      // children are not known yet could still grow (`absorb`).  Typical case: a leaf of millions of rows with a
      // modest gain behind a few small leaves with large ones.
      int64_t absorb = 0;
      auto max_subtree_splits = [&](const Node& n) -> int64_t {
        const SplitInfo& b = n.best;
        int64_t most = std::numeric_limits<int64_t>::max() / 4;
        if (cfg_.min_data_in_leaf > 1) most = std::min<int64_t>(most, b.total_cnt / cfg_.min_data_in_leaf);
        if (cfg_.min_sum_hessian_in_leaf > 0.0) {
          const double leaves = (double)b.total_h * hunit / cfg_.min_sum_hessian_in_leaf;
          if (leaves < (double)most) most = (int64_t)leaves;
        }
        return std::max<int64_t>(most - 1, 0);
      };
      std::vector<int> held_back;   // leaves the gain bar kept out of this round, best first
      int64_t round_rows = 0;
      auto add_task = [&](int node_id) {
        Node& c = nodes[(size_t)node_id];
        round_rows += c.best.total_cnt;
        SplitTask t{};
        t.parent_node = node_id;
        t.left_node = (int)nodes.size();
        t.right_node = (int)nodes.size() + 1;
        t.feature = c.best.feature;
        t.threshold = c.best.threshold;
        t.default_left = c.best.default_left == kCategoricalSetTask ? kCategoricalSetTask
                         : meta[(size_t)c.best.feature].categorical   ? kCategoricalTask
                                                                      : c.best.default_left;
        t.missing_bin = meta[(size_t)c.best.feature].missing_bin;
        int64_t lc = c.best.left_cnt, rc = c.best.total_cnt - c.best.left_cnt;
        t.hist_child = lc <= rc ? 0 : 1;
        t.hist_rows = (int32_t)std::min<int64_t>(std::min(lc, rc), INT32_MAX);
        t.parent_slot = c.slot;
        t.new_slot = 1 + materialized + (int)tasks.size();
        t.depth = c.depth + 1;
        t.parent_key = (int32_t)c.key;
        Node l, r;
        l.key = child_node_key(c.key, 0);
        r.key = child_node_key(c.key, 1);
        l.parent = r.parent = node_id;
        l.depth = r.depth = c.depth + 1;
        l.slot = t.hist_child == 0 ? t.new_slot : t.parent_slot;
        r.slot = t.hist_child == 1 ? t.new_slot : t.parent_slot;
        if (node_info) {
          // children inherit the range; a split on a constrained feature also separates them at the
          // midpoint of their outputs, which keeps every later descendant on its side
          l.lo = r.lo = c.lo;
          l.hi = r.hi = c.hi;
          child_outputs(c, &l.out, &r.out);
          const int mono = meta[(size_t)c.best.feature].monotone;
          if (mono != 0) {
            const double mid = 0.5 * (l.out + r.out);
            if (mono > 0) { l.hi = std::min(l.hi, mid); r.lo = std::max(r.lo, mid); }
            else { l.lo = std::max(l.lo, mid); r.hi = std::min(r.hi, mid); }
          }
          be.set_node_bounds(t.left_node, l.lo, l.hi, l.out);
          be.set_node_bounds(t.right_node, r.lo, r.hi, r.out);
        }
        if (constrained) {
          // both children: the parent's branch plus the split feature
          path_bits.resize((nodes.size() + 2) * W, 0ull);
          for (int side = 0; side < 2; ++side) {
            uint64_t* cp = &path_bits[(nodes.size() + (size_t)side) * W];
            for (size_t w = 0; w < W; ++w) cp[w] = path_bits[(size_t)node_id * W + w];
            cp[(size_t)(c.best.feature >> 6)] |= 1ull << (c.best.feature & 63);
          }
          allowed_for(&path_bits[nodes.size() * W]);
          be.set_node_mask(t.left_node, allowed.data());
          be.set_node_mask(t.right_node, allowed.data());
        }
        nodes[(size_t)node_id].left = t.left_node;   // (c may dangle once nodes grows)
        nodes[(size_t)node_id].right = t.right_node;
        nodes.push_back(l);
        nodes.push_back(r);
        tasks.push_back(t);
      };
      while (!virt.empty() && vbudget > 0 && (int)tasks.size() < max_new) {
        Entry e = virt.top();
        virt.pop();
        --vbudget;
        Node& c = nodes[(size_t)e.node];
        if (c.left >= 0) {
          if (!c.ready) { absorb += max_subtree_splits(c); continue; }   // in flight: children unknown
          for (int child : {c.left, c.right}) {
            const SplitInfo& s = nodes[(size_t)child].best;
            if (splittable(s)) virt.push(Entry{s.gain, vseq, child});
            ++vseq;
          }
          continue;
        }
        const bool is_stall_leaf = e.node == frontier.top().node;
        const bool certain = certain_splits_ && (int64_t)vbudget >= absorb;
        absorb += max_subtree_splits(c);
        if (!is_stall_leaf && !certain && e.gain < floor_gain_for(c.best.total_cnt, budget_total - vbudget - 1)) {
          held_back.push_back(e.node);
          continue;
        }
        add_task(e.node);
      }
      // a nearly empty round is filled with the best leaves the bar held back (see grower.hpp: fill_rows_)
      if (!tasks.empty() && fill_rows_ > 0.0) {
        const double room = fill_rows_ * row_shards_, leaf_most = fill_leaf_rows_ * row_shards_;
        for (int id : held_back) {
          if ((int)tasks.size() >= max_new) break;
          const int64_t r = nodes[(size_t)id].best.total_cnt;
          if ((double)r > leaf_most || (double)(round_rows + r) > room) continue;
          add_task(id);
        }
      }
    }
    if (!tasks.empty()) {
      // Rounds in flight run side by side on the device, so a batch is dealt out over the free places: every large
      // leaf (a round of its own costs it nothing, and its long row-level work then does not hold back anybody
      // else's split search and hand-over) while places last, the remaining leaves together - halved when two
      // places are still free and nothing else is in flight, so that one half's hand-over and replay overlap with
      // the other half's device work.  Slots are handed out again afterwards: a round's new slots are consecutive.
      const int free_places = max_flight - (int)flights.size();
      std::vector<size_t> cut;   // start index of every round after the first
      if (free_places >= 2 && tasks.size() >= 2) {
        auto rows_of = [&](const SplitTask& t) { return nodes[(size_t)t.parent_node].best.total_cnt; };
        if (max_flight > 2) {
          std::stable_sort(tasks.begin(), tasks.end(), [&](const SplitTask& a, const SplitTask& b) { return rows_of(a) > rows_of(b); });
          size_t own = 0;
          while (own < tasks.size() && (int)own < free_places - 1 && (double)rows_of(tasks[own]) >= own_round_rows_ * row_shards_) ++own;
          if (own == tasks.size()) --own;   // the last one is "the rest"
          for (size_t i = 1; i <= own; ++i) cut.push_back(i);
          const size_t rest = tasks.size() - own;
          if (free_places - (int)own >= 2 && rest >= 2 && (own == 0 ? flights.empty() : rest >= 8)) cut.push_back(own + (rest + 1) / 2);
        } else if (flights.empty()) {
          cut.push_back((tasks.size() + 1) / 2);
        }
      }
      for (size_t i = 0; i < tasks.size(); ++i) {
        SplitTask& t = tasks[i];
        t.new_slot = 1 + materialized + (int)i;
        nodes[(size_t)(t.hist_child == 0 ? t.left_node : t.right_node)].slot = t.new_slot;
      }
      materialized += (int)tasks.size();
      size_t begin = 0;
      for (size_t c = 0; c <= cut.size(); ++c) {
        const size_t end = c < cut.size() ? cut[c] : tasks.size();
        submit(std::vector<SplitTask>(tasks.begin() + (std::ptrdiff_t)begin, tasks.begin() + (std::ptrdiff_t)end));
        begin = end;
      }
      continue;
    }
    // 3. nothing new to start: a round must finish first
    if (flights.empty()) {
      if (stalled_on_flight) throw std::runtime_error("tree grower: waiting for a round that was never submitted");
      throw std::runtime_error("tree grower stalled without work");
    }
    wait_oldest();
  }
  // rounds still in flight only carried speculation: their results are not needed, the device work is
  while (!flights.empty()) wait_oldest();
  be.tree_rounds_done();
  stats_.materialized_splits = materialized;
  stats_.used_splits = (int)chosen.size();
  // the bar for the next tree: suffix minima of this tree's chosen gains (empty when the tree ran out of
  // splittable leaves before its budget: then nothing is held back)
  bar_used = prev_bar_.empty() ? 0.0 : prev_bar_.back() * bar_scale;
  }   // growth by rounds
  const QuantScale qs = given_ ? given_qs_ : be.quant_scale();
  const double gunit = unit_of(qs.kg), hunit = unit_of(qs.kh);
  prev_bar_.clear();
  if ((int)chosen.size() == budget_total && !chosen.empty()) {
    prev_bar_.resize(chosen.size());
    double m = std::numeric_limits<double>::infinity();
    for (size_t i = chosen.size(); i-- > 0;) {
      m = std::min(m, nodes[(size_t)chosen[i]].best.gain);
      prev_bar_[i] = m;
    }
  }
  // Assemble the tree in LightGBM numbering by replaying the final order.
  Tree tree;
  tree.reset(cfg_.num_leaves);
  std::vector<int> leaf_of_node(nodes.size(), -1);
  leaf_of_node[0] = 0;
  std::vector<int32_t> leaf_nodes(1, 0);
  {
    const SplitInfo& r = nodes[0].best;
    double G = (double)r.total_g * gunit, H = (double)r.total_h * hunit;
    tree.leaf_value[0] = leaf_output(G, H, sp.lambda_l1, sp.lambda_l2, sp.max_delta_step);
    tree.leaf_weight[0] = H;
    tree.leaf_count[0] = r.total_cnt;
  }
  for (int id : chosen) {
    const Node& n = nodes[(size_t)id];
    const SplitInfo& s = n.best;
    int leaf = leaf_of_node[(size_t)id];
    const BinMapper& bm = ds_->inner_mapper(s.feature);
    double lg = (double)s.left_g * gunit, lh = (double)s.left_h * hunit;
    double rg = (double)(s.total_g - s.left_g) * gunit, rh = (double)(s.total_h - s.left_h) * hunit;
    const bool set_split = s.default_left == kCategoricalSetTask;
    const double l2 = sp.lambda_l2 + (set_split ? cfg_.cat_l2 : 0.0);
    double lv = leaf_output(lg, lh, sp.lambda_l1, l2, sp.max_delta_step);
    double rv = leaf_output(rg, rh, sp.lambda_l1, l2, sp.max_delta_step);
    double split_gain_out = s.gain;
    if (n.forced) {
      // a forced split was never valued by the search: its gain is what its sums give (0 when it lowers the objective
      // or leaves a child without rows, whose output then is 0 as well)
      const double G = (double)s.total_g * gunit, H = (double)s.total_h * hunit;
      split_gain_out = leaf_gain(lg, lh, sp.lambda_l1, sp.lambda_l2) + leaf_gain(rg, rh, sp.lambda_l1, sp.lambda_l2) -
                       leaf_gain(G, H, sp.lambda_l1, sp.lambda_l2);
      if (!(split_gain_out > 0.0) || !std::isfinite(split_gain_out)) split_gain_out = 0.0;
      if (!std::isfinite(lv) || nodes[(size_t)n.left].best.total_cnt == 0) lv = 0.0;
      if (!std::isfinite(rv) || nodes[(size_t)n.right].best.total_cnt == 0) rv = 0.0;
    }
    if (node_info) {
      // the outputs the split was valued with (smoothed / clamped), kept inside the children's own ranges
      lv = clamp_output(nodes[(size_t)n.left].out, nodes[(size_t)n.left].lo, nodes[(size_t)n.left].hi);
      rv = clamp_output(nodes[(size_t)n.right].out, nodes[(size_t)n.right].lo, nodes[(size_t)n.right].hi);
    }
    // exact child counts come from the children's own scans
    int64_t lc = nodes[(size_t)n.left].best.total_cnt, rc = nodes[(size_t)n.right].best.total_cnt;
    // "every value left, only missing right" has an infinite bin bound: keep the model text finite
    double real_thr = bm.bin_to_threshold(s.threshold);
    if (std::isinf(real_thr)) real_thr = 1e300;
    int right_leaf;
    if (set_split) {
      uint64_t bins[kCatSetWords];
      be.node_cat_set(id, bins);
      std::vector<int64_t> categories;
      for (int b = 0; b < (int)bm.upper.size() && b < kMaxBins; ++b)
        if ((bins[b >> 6] >> (b & 63)) & 1ull) categories.push_back((int64_t)bm.upper[(size_t)b]);
      right_leaf = tree.split_categorical_set(leaf, ds_->real_feature(s.feature), s.feature, bins, categories, split_gain_out, lv, rv,
                                              lc, rc, lh, rh);
    } else if (bm.categorical)
      right_leaf = tree.split_categorical(leaf, ds_->real_feature(s.feature), s.feature, s.threshold,
                                          (int64_t)bm.upper[(size_t)s.threshold], split_gain_out, lv, rv, lc, rc, lh, rh);
    else
      right_leaf = tree.split(leaf, ds_->real_feature(s.feature), s.feature, s.threshold,
                              real_thr, s.default_left != 0, (int)bm.missing, split_gain_out, lv, rv,
                              lc, rc, lh, rh);
    leaf_of_node[(size_t)n.left] = leaf;
    leaf_of_node[(size_t)n.right] = right_leaf;
    leaf_nodes[(size_t)leaf] = n.left;
    leaf_nodes.push_back(n.right);
  }
  if (!coupled_penalty_.empty()) {
    // a feature's coupled cost is paid by the first tree that uses it
    for (int id : chosen) coupled_penalty_[(size_t)nodes[(size_t)id].best.feature] = 0.0;
    refresh_gain_adjustment();
  }
  // Rows of a final leaf: a leaf that was partitioned speculatively (its split never got its turn)
  // has its rows in the ranges of its deepest partitioned descendants.
  pieces->clear();
  std::vector<int> stack;
  for (size_t l = 0; l < leaf_nodes.size(); ++l) {
    stack.assign(1, leaf_nodes[l]);
    while (!stack.empty()) {
      int id = stack.back();
      stack.pop_back();
      const Node& n = nodes[(size_t)id];
      if (n.left < 0) pieces->push_back(LeafPiece{id, (int32_t)l});
      else { stack.push_back(n.right); stack.push_back(n.left); }
    }
  }
  if (g_trace.on && std::getenv("LGA_GROW_DUMP") && tree_index_ == std::atoll(std::getenv("LGA_GROW_DUMP"))) {
    std::vector<char> is_chosen(nodes.size(), 0);
    for (int id : chosen) is_chosen[(size_t)id] = 1;
    const double prev_cutoff_ = prev_bar_.empty() ? 0.0 : prev_bar_.back();
    std::fprintf(stderr, "[grow dump] tree %lld last chosen gain %.6g estimate %.6g\n", (long long)tree_index_, prev_cutoff_, bar_used);
    for (size_t i = 0; i < nodes.size(); ++i) {
      const Node& n = nodes[i];
      if (n.left < 0) continue;
      std::fprintf(stderr, "[grow dump] node %zu parent %d depth %d rows %lld gain %.6g rel %.3f chosen %d parent_chosen %d\n", i, n.parent,
                   n.depth, (long long)n.best.total_cnt, n.best.gain, prev_cutoff_ > 0 ? n.best.gain / prev_cutoff_ : 0.0,
                   (int)is_chosen[i], n.parent >= 0 ? (int)is_chosen[(size_t)n.parent] : 1);
    }
  }
  if (g_trace.on) {
    for (const Node& n : nodes) if (n.left >= 0) g_trace.rows_split += (double)n.best.total_cnt;
    for (int id : chosen) g_trace.rows_used += (double)nodes[(size_t)id].best.total_cnt;
    g_trace.rounds += stats_.rounds;
    g_trace.tasks += materialized;
    g_trace.est_ratio += bar_used > 0.0 && !prev_bar_.empty() ? prev_bar_.back() / bar_used : 1.0;
    const auto t_exit = GrowTrace::clock::now();
    g_trace.total_us += GrowTrace::us(t_enter, t_exit);
    g_trace.backend_us += backend_us;
    g_trace.tail_us += GrowTrace::us(t_last_backend, t_exit);
    g_trace.report();
  }
  return tree;
}

}  // namespace lga

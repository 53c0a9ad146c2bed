#include "grower.hpp"
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <stdexcept>

namespace lga {

TreeGrower::TreeGrower(const Config& cfg, const Dataset* ds) : cfg_(cfg), ds_(ds) {}

SplitParams TreeGrower::split_params() const {
  SplitParams p;
  p.lambda_l1 = cfg_.lambda_l1;
  p.lambda_l2 = cfg_.lambda_l2;
  p.min_gain_to_split = cfg_.min_gain_to_split;
  p.min_sum_hessian_in_leaf = cfg_.min_sum_hessian_in_leaf;
  p.max_delta_step = cfg_.max_delta_step;
  p.min_data_in_leaf = cfg_.min_data_in_leaf;
  p.max_depth = cfg_.max_depth;
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
  int trees = 0;
  using clock = std::chrono::steady_clock;
  static double us(clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
  void report() {
    if (++trees % 100 == 0) {
      std::fprintf(stderr, "[grow trace] per tree: total %.1f us, in backend %.1f us, host %.1f us (of which after the last round %.1f us)\n",
                   total_us / 100, backend_us / 100, (total_us - backend_us) / 100, tail_us / 100);
      total_us = backend_us = tail_us = 0;
    }
  }
};
GrowTrace g_trace;
}  // namespace

Tree TreeGrower::grow(TreeBackend& be, const std::vector<FeatureMeta>& meta, std::vector<LeafPiece>* pieces) {
  const auto t_enter = GrowTrace::clock::now();
  auto t_last_backend = t_enter;
  double backend_us = 0;
  stats_ = GrowStats();
  const SplitParams sp = split_params();
  be.begin_tree(meta, sp);
  const int capacity_splits = (be.node_capacity() - 1) / 2;
  const int budget_total = cfg_.num_leaves - 1;
  if (capacity_splits < budget_total) throw std::runtime_error("backend node capacity below num_leaves");

  std::vector<Node> nodes;
  nodes.reserve((size_t)be.node_capacity());
  nodes.emplace_back();
  be.root_round(&nodes[0].best);
  if (g_trace.on) { t_last_backend = GrowTrace::clock::now(); backend_us += GrowTrace::us(t_enter, t_last_backend); }
  stats_.rounds = 1;
  const QuantScale qs = be.quant_scale();
  const double gunit = unit_of(qs.kg), hunit = unit_of(qs.kh);

  auto depth_blocked = [&](int depth) { return cfg_.max_depth > 0 && depth >= cfg_.max_depth; };
  if (depth_blocked(0)) nodes[0].best.feature = -1;

  std::priority_queue<Entry, std::vector<Entry>, EntryLess> frontier;
  int seq = 0;
  auto push = [&](int node) {
    const SplitInfo& s = nodes[(size_t)node].best;
    if (splittable(s)) frontier.push(Entry{s.gain, seq, node});
    ++seq;
  };
  push(0);

  std::vector<int> chosen;   // node ids in strict best-first order
  int budget = budget_total;
  int materialized = 0;
  std::vector<SplitTask> tasks;
  std::vector<SplitInfo> infos;

  while (budget > 0 && !frontier.empty()) {
    Entry top = frontier.top();
    Node& n = nodes[(size_t)top.node];
    if (n.left >= 0) {
      // already partitioned and scanned: the split is final
      frontier.pop();
      chosen.push_back(top.node);
      --budget;
      push(n.left);
      push(n.right);
      continue;
    }
    // Stall: continue the replay VIRTUALLY over everything that is known - leaves that are already
    // partitioned hand over their children - and split every leaf it reaches that is not
    // partitioned yet, all in one backend round.
    int pool_room = capacity_splits - materialized - (budget - 1);
    if (pool_room < 1) throw std::runtime_error("tree grower ran out of node capacity");
    int max_new = std::min(budget, pool_room);
    if (cfg_.max_batch_leaves > 0) max_new = std::min(max_new, cfg_.max_batch_leaves);
    auto virt = frontier;
    int vseq = seq;
    int vbudget = budget;
    tasks.clear();
    while (!virt.empty() && vbudget > 0 && (int)tasks.size() < max_new) {
      Entry e = virt.top();
      virt.pop();
      --vbudget;
      Node& c = nodes[(size_t)e.node];
      if (c.left >= 0) {
        for (int child : {c.left, c.right}) {
          const SplitInfo& s = nodes[(size_t)child].best;
          if (splittable(s)) virt.push(Entry{s.gain, vseq, child});
          ++vseq;
        }
        continue;
      }
      SplitTask t{};
      t.parent_node = e.node;
      t.left_node = (int)nodes.size();
      t.right_node = (int)nodes.size() + 1;
      t.feature = c.best.feature;
      t.threshold = c.best.threshold;
      t.default_left = c.best.default_left;
      t.missing_bin = meta[(size_t)c.best.feature].missing_bin;
      int64_t lc = c.best.left_cnt, rc = c.best.total_cnt - c.best.left_cnt;
      t.hist_child = lc <= rc ? 0 : 1;
      t.parent_slot = c.slot;
      t.new_slot = 1 + materialized + (int)tasks.size();
      t.depth = c.depth + 1;
      Node l, r;
      l.parent = r.parent = e.node;
      l.depth = r.depth = c.depth + 1;
      l.slot = t.hist_child == 0 ? t.new_slot : t.parent_slot;
      r.slot = t.hist_child == 1 ? t.new_slot : t.parent_slot;
      c.left = t.left_node;
      c.right = t.right_node;
      nodes.push_back(l);
      nodes.push_back(r);
      tasks.push_back(t);
    }
    infos.assign(tasks.size() * 2, SplitInfo{});
    const auto t_call = GrowTrace::clock::now();
    be.split_round(tasks, infos.data());
    if (g_trace.on) { t_last_backend = GrowTrace::clock::now(); backend_us += GrowTrace::us(t_call, t_last_backend); }
    ++stats_.rounds;
    materialized += (int)tasks.size();
    for (size_t i = 0; i < tasks.size(); ++i) {
      for (int side = 0; side < 2; ++side) {
        int id = side == 0 ? tasks[i].left_node : tasks[i].right_node;
        nodes[(size_t)id].best = infos[i * 2 + (size_t)side];
        if (depth_blocked(nodes[(size_t)id].depth)) nodes[(size_t)id].best.feature = -1;
      }
    }
  }
  stats_.materialized_splits = materialized;
  stats_.used_splits = (int)chosen.size();

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
    double lv = leaf_output(lg, lh, sp.lambda_l1, sp.lambda_l2, sp.max_delta_step);
    double rv = leaf_output(rg, rh, sp.lambda_l1, sp.lambda_l2, sp.max_delta_step);
    // exact child counts come from the children's own scans
    int64_t lc = nodes[(size_t)n.left].best.total_cnt, rc = nodes[(size_t)n.right].best.total_cnt;
    // "every value left, only missing right" has an infinite bin bound: keep the model text finite
    double real_thr = bm.bin_to_threshold(s.threshold);
    if (std::isinf(real_thr)) real_thr = 1e300;
    int right_leaf = tree.split(leaf, ds_->real_feature(s.feature), s.feature, s.threshold,
                                real_thr, s.default_left != 0, (int)bm.missing, s.gain, lv, rv,
                                lc, rc, lh, rh);
    leaf_of_node[(size_t)n.left] = leaf;
    leaf_of_node[(size_t)n.right] = right_leaf;
    leaf_nodes[(size_t)leaf] = n.left;
    leaf_nodes.push_back(n.right);
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
  if (g_trace.on) {
    const auto t_exit = GrowTrace::clock::now();
    g_trace.total_us += GrowTrace::us(t_enter, t_exit);
    g_trace.backend_us += backend_us;
    g_trace.tail_us += GrowTrace::us(t_last_backend, t_exit);
    g_trace.report();
  }
  return tree;
}

}  // namespace lga

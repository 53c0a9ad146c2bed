#include "host_engine.hpp"
#include "metric.hpp"
#include "cat_search.hpp"
#include "percentile.hpp"
#include <algorithm>
#include <functional>
#include <cmath>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace lga {

void quantize_rows_host(const float* g, const float* h, const int32_t* rows, int64_t n, QuantScale qs,
                        int64_t* packed, const RoundingSpec& rounding) {
#pragma omp parallel for schedule(static) if (n > 65536)
  for (int64_t i = 0; i < n; ++i) {
    int64_t r = rows ? rows[i] : i;
    packed[r] = pack_gh(quantize_with(g[r], qs.kg, rounding, 0, (uint64_t)r), quantize_with(h[r], qs.kh, rounding, 1, (uint64_t)r));
  }
}

namespace {

struct PhaseTimer {
  double part = 0, hist = 0, scan = 0, reduce = 0;
  bool on = std::getenv("LGA_HOST_PROFILE") != nullptr;
  ~PhaseTimer() {
    if (on) std::fprintf(stderr, "[lga host] partition %.3fs hist %.3fs reduce %.3fs scan %.3fs\n", part, hist, reduce, scan);
  }
};
inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------
// Host backend
// ---------------------------------------------------------------------------
class HostBackend : public TreeBackend {
 public:
  HostBackend(const Dataset* ds, const Config& cfg, Comm* comm, ParallelMode mode)
      : root_(ds->root()), comm_(comm), mode_(mode) {
    F_ = root_->num_features();
    TF_ = root_->num_tiles() * kTileFeatures;
    N_ = root_->root_rows();
    plane_ = (size_t)TF_ * kMaxBins;
    slot_words_ = 2 * plane_ + kSlotHeader;
    int L = std::max(cfg.num_leaves, 2);
    capacity_splits_ = 2 * (L - 1) + 1;
    pool_.assign((size_t)(capacity_splits_ + 1) * slot_words_, 0);
    ranges_.resize((size_t)(2 * capacity_splits_ + 1));
    cat_sets_.assign(ranges_.size() * kCatSetWords, 0ull);
    cat_.cat_smooth = cfg.cat_smooth;
    cat_.cat_l2 = cfg.cat_l2;
    cat_.max_cat_threshold = cfg.max_cat_threshold;
    cat_.min_data_per_group = cfg.min_data_per_group;
    top_k_ = cfg.top_k;
    if (comm_ && comm_->size() > 1 && mode_ == ParallelMode::Serial) mode_ = ParallelMode::Data;
    if (!comm_ || comm_->size() <= 1) mode_ = ParallelMode::Serial;
  }

  void set_bag(const int32_t* rows, int64_t n) { bag_ = rows; bag_n_ = n; }
  void set_gradients(const int64_t* packed, QuantScale qs) { packed_ = packed; wide_g_ = wide_h_ = nullptr; qs_ = qs; }
  void set_wide_gradients(const int64_t* g, const int64_t* h, QuantScale qs) { wide_g_ = g; wide_h_ = h; packed_ = nullptr; qs_ = qs; }

  int node_capacity() const override { return 2 * capacity_splits_ + 1; }
  QuantScale quant_scale() const override { return qs_; }
  int64_t node_local_count(int node) const override { return ranges_[(size_t)node].count; }
  void set_node_bounds(int node, double lo, double hi, double out) override {
    if (bounds_.empty()) {
      bounds_.assign(ranges_.size() * kNodeInfoStride, std::numeric_limits<double>::quiet_NaN());
      for (size_t i = 0; i < ranges_.size(); ++i) { bounds_[kNodeInfoStride * i] = -kInf; bounds_[kNodeInfoStride * i + 1] = kInf; }
    }
    double* b = &bounds_[(size_t)kNodeInfoStride * (size_t)node];
    b[0] = lo; b[1] = hi; b[2] = out;
  }
  void node_cat_set(int node, uint64_t* words) const override {
    for (int w = 0; w < kCatSetWords; ++w) words[w] = cat_sets_[(size_t)node * kCatSetWords + (size_t)w];
  }
  void set_node_mask(int node, const uint64_t* words) override {
    const size_t W = (size_t)((F_ + 63) / 64);
    if (masks_.empty()) masks_.assign(ranges_.size() * W, ~0ull);
    for (size_t w = 0; w < W; ++w) masks_[(size_t)node * W + w] = words[w];
    sp_.node_masks = masks_.data();
    sp_.mask_words = (int32_t)W;
  }
  // Rounds run at submission here; LGA_HOST_IN_FLIGHT=2 lets the tests drive the grower's
  // several-rounds-in-flight logic through this backend.
  int max_in_flight() const override {
    const char* e = std::getenv("LGA_HOST_IN_FLIGHT");
    return e ? std::max(1, std::atoi(e)) : 1;
  }
  const NodeRange& range(int node) const { return ranges_[(size_t)node]; }
  ParallelMode mode() const { return mode_; }
  const int32_t* arena(int buf) const { return arena_[buf].data(); }

  void begin_tree(const std::vector<FeatureMeta>& meta, const SplitParams& params) override {
    meta_ = meta;
    sp_ = params;
    bounds_.clear();   // filled again by the grower when a feature of this tree is constrained
    masks_.clear();
    for (int b = 0; b < 2; ++b)
      if ((int64_t)arena_[b].size() < bag_n_) arena_[b].resize((size_t)bag_n_);
    std::memcpy(arena_[0].data(), bag_, (size_t)bag_n_ * sizeof(int32_t));
    ranges_[0] = NodeRange{0, (int32_t)bag_n_, 0, 0};
  }

  void root_round(SplitInfo* out) override {
    build_hist(0, 0);
    reduce_slots(0, 1);
    ScanTask st{0, 0, -1, -1, 0, kRootNodeKey};
    scan_nodes(&st, 1, out);
  }

  void split_round(std::vector<SplitTask>& tasks, SplitInfo* infos) override {
    size_t B = tasks.size();
    if (B == 0) return;
    double t0 = now_s();
    // 1. partition
    int64_t batch_rows = 0;
    for (size_t i = 0; i < B; ++i) batch_rows += ranges_[(size_t)tasks[i].parent_node].count;
    (void)batch_rows;   // only read by the OpenMP clause below
#pragma omp parallel for schedule(dynamic) if (B > 1 && batch_rows > 200000)
    for (size_t i = 0; i < B; ++i) partition(tasks[i]);
    double t1 = now_s();
    // 2. histograms of the chosen children (fresh, contiguous slots)
    for (size_t i = 0; i < B; ++i) {
      const SplitTask& t = tasks[i];
      build_hist(t.hist_child == 0 ? t.left_node : t.right_node, t.new_slot);
    }
    double t2 = now_s();
    reduce_slots(tasks[0].new_slot, (int)B);
    double t3 = now_s();
    timer_.part += t1 - t0; timer_.hist += t2 - t1; timer_.reduce += t3 - t2;
    // 3. scan both children of every task (the sibling by subtraction, in place in the parent's slot)
    std::vector<ScanTask> st(B * 2);
    for (size_t i = 0; i < B; ++i) {
      const SplitTask& t = tasks[i];
      for (int side = 0; side < 2; ++side) {
        ScanTask& s = st[i * 2 + (size_t)side];
        s.node = side == 0 ? t.left_node : t.right_node;
        s.depth = t.depth;
        s.key = child_node_key((uint32_t)t.parent_key, side);
        if (side == t.hist_child) { s.slot = t.new_slot; s.parent_slot = -1; s.sibling_slot = -1; }
        else { s.slot = t.parent_slot; s.parent_slot = t.parent_slot; s.sibling_slot = t.new_slot; }
      }
    }
    scan_nodes(st.data(), (int)st.size(), infos);
    timer_.scan += now_s() - t3;
  }

 private:
  PhaseTimer timer_;
  int64_t* slot(int s) { return pool_.data() + (size_t)s * slot_words_; }

  bool owns_feature(int f) const {
    if (mode_ != ParallelMode::Feature) return true;
    return ((f / kTileFeatures) % comm_->size()) == comm_->rank();
  }

  void partition(const SplitTask& t) {
    const NodeRange p = ranges_[(size_t)t.parent_node];
    const int32_t* src = arena_[p.buf].data() + p.begin;
    int32_t* dst = arena_[1 - p.buf].data() + p.begin;
    const uint8_t* col = root_->bins_fm() + (size_t)t.feature * (size_t)N_;
    const uint64_t* set = t.default_left == kCategoricalSetTask ? &cat_sets_[(size_t)t.parent_node * kCatSetWords] : nullptr;
    auto goes_left = [&](int b) {
      if (set) return ((set[b >> 6] >> (b & 63)) & 1ull) != 0;
      if (t.default_left == kCategoricalTask) return b == t.threshold;
      return (b == t.missing_bin) ? (t.default_left != 0) : (b <= t.threshold);
    };
    int32_t nl = 0;
    for (int32_t i = 0; i < p.count; ++i) nl += goes_left(col[src[i]]) ? 1 : 0;
    int32_t li = 0, ri = nl;
    for (int32_t i = 0; i < p.count; ++i) {
      int32_t row = src[i];
      if (goes_left(col[row])) dst[li++] = row; else dst[ri++] = row;
    }
    ranges_[(size_t)t.left_node] = NodeRange{p.begin, nl, 1 - p.buf, 0};
    ranges_[(size_t)t.right_node] = NodeRange{p.begin + nl, p.count - nl, 1 - p.buf, 0};
  }

  void build_hist(int node, int s) {
    const NodeRange r = ranges_[(size_t)node];
    int64_t* base = slot(s);
    std::memset(base, 0, slot_words_ * sizeof(int64_t));
    const int32_t* rows = arena_[r.buf].data() + r.begin;
    const int64_t n = r.count;
    int64_t* gp = base;
    int64_t* hp = base + plane_;
    const uint8_t* fm = root_->bins_fm();
#pragma omp parallel for schedule(dynamic) if (n * (int64_t)F_ > 400000)
    for (int f = 0; f < F_; ++f) {
      if (!owns_feature(f)) continue;
      const uint8_t* col = fm + (size_t)f * (size_t)N_;
      int64_t* g = gp + (size_t)f * kMaxBins;
      int64_t* h = hp + (size_t)f * kMaxBins;
      if (wide_g_) {
        for (int64_t i = 0; i < n; ++i) {
          const int32_t row = rows[i];
          const int b = col[row];
          g[b] += wide_g_[row];
          h[b] += wide_h_[row];
        }
      } else {
        for (int64_t i = 0; i < n; ++i) {
          int32_t row = rows[i];
          int64_t p = packed_[row];
          int b = col[row];
          g[b] += packed_g(p);
          h[b] += packed_h(p);
        }
      }
    }
    int64_t tg = 0, th = 0;
    if (wide_g_) {
      for (int64_t i = 0; i < n; ++i) { tg += wide_g_[rows[i]]; th += wide_h_[rows[i]]; }
    } else {
      for (int64_t i = 0; i < n; ++i) {
        int64_t p = packed_[rows[i]];
        tg += packed_g(p);
        th += packed_h(p);
      }
    }
    int64_t* hd = base + 2 * plane_;
    hd[0] = n; hd[1] = tg; hd[2] = th;
  }

  void reduce_slots(int first, int count) {
    if (mode_ != ParallelMode::Data) return;   // (voting keeps local histograms: vote_and_search sums what it needs)
    comm_->allreduce_sum_i64(slot(first), (size_t)count * slot_words_, false, nullptr);
  }

  struct Cand {
    double gain;
    uint64_t key;
    int64_t lg, lh;
  };

  void scan_feature(const int64_t* g, const int64_t* h, int f, const LeafScanCtx& ctx, uint32_t key, Cand* best) const {
    scan_feature_with(sp_, g, h, f, ctx, key, best);
  }
  void scan_feature_with(const SplitParams& sp_, const int64_t* g, const int64_t* h, int f, const LeafScanCtx& ctx, uint32_t key,
                         Cand* best) const {
    const FeatureMeta& m = meta_[(size_t)f];
    int nb = m.num_bin, mb = m.missing_bin;
    int vb = mb >= 0 ? nb - 1 : nb;   // value bins
    int64_t mg = mb >= 0 ? g[mb] : 0, mh = mb >= 0 ? h[mb] : 0;
    int64_t lg = 0, lh = 0;
    int last = mb >= 0 ? vb - 1 : vb - 2;
    const int only = extra_trees_threshold(sp_, key, f, last + 1);
    if (m.categorical) {
      // one category against the rest: the left child is a single value bin
      for (int t = 0; t < vb; ++t) {
        if (only >= 0 && t != only) continue;
        double ga = adjust_feature_gain(split_gain(g[t], h[t], ctx, sp_, 0), f, ctx.total_cnt, sp_);
        uint64_t ka = cand_key(f, t, 0);
        if (cand_better(ga, ka, best->gain, best->key)) *best = Cand{ga, ka, g[t], h[t]};
      }
      return;
    }
    for (int t = 0; t <= last; ++t) {
      lg += g[t];
      lh += h[t];
      if (only >= 0 && t != only) continue;
      double ga = adjust_feature_gain(split_gain(lg, lh, ctx, sp_, m.monotone), f, ctx.total_cnt, sp_);
      uint64_t ka = cand_key(f, t, 0);
      if (cand_better(ga, ka, best->gain, best->key)) *best = Cand{ga, ka, lg, lh};
      if (mb >= 0 && t <= vb - 2) {
        double gb = adjust_feature_gain(split_gain(lg + mg, lh + mh, ctx, sp_, m.monotone), f, ctx.total_cnt, sp_);
        uint64_t kb = cand_key(f, t, 1);
        if (cand_better(gb, kb, best->gain, best->key)) *best = Cand{gb, kb, lg + mg, lh + mh};
      }
    }
  }

  void scan_nodes(const ScanTask* st, int n, SplitInfo* out) {
    const double gunit = unit_of(qs_.kg), hunit = unit_of(qs_.kh);
    const double ninf = -std::numeric_limits<double>::infinity();
    const uint64_t worst_key = ~0ull;
    std::vector<Cand> local((size_t)n);
    std::vector<CatSetSplit> sets((size_t)n);
#pragma omp parallel for schedule(dynamic) if (n * F_ >= 512)
    for (int i = 0; i < n; ++i) {
      const ScanTask& s = st[i];
      int64_t* self = slot(s.slot);
      if (s.parent_slot >= 0) {
        // self (== parent's storage) = parent - sibling, element-wise, header included
        const int64_t* sib = slot(s.sibling_slot);
        for (size_t w = 0; w < slot_words_; ++w) self[w] -= sib[w];
      }
      const int64_t* hd = self + 2 * plane_;
      int64_t tc = hd[0], tg = hd[1], th = hd[2];
      if (mode_ == ParallelMode::Voting) {   // local sums: the search follows the vote (vote_and_search)
        local[(size_t)i] = Cand{ninf, worst_key, 0, 0};
        sets[(size_t)i].gain = ninf;
        continue;
      }
      const bool ranged = !bounds_.empty();
      const double* nb = ranged ? &bounds_[(size_t)kNodeInfoStride * (size_t)s.node] : nullptr;
      LeafScanCtx ctx = make_scan_ctx(tg, th, tc, gunit, hunit, sp_, ranged ? nb[0] : -kInf, ranged ? nb[1] : kInf,
                                      ranged ? nb[2] : std::numeric_limits<double>::quiet_NaN());
      Cand best{ninf, worst_key, 0, 0};
      for (int f = 0; f < F_; ++f) {
        if (!meta_[(size_t)f].enabled || !owns_feature(f) || !node_feature_on(sp_, s.key, f, F_) ||
            !node_allows(sp_, s.node, f) || meta_[(size_t)f].categorical == kCategoricalSorted)
          continue;
        scan_feature(self + (size_t)f * kMaxBins, self + plane_ + (size_t)f * kMaxBins, f, ctx, s.key, &best);
      }
      local[(size_t)i] = best;
      // features searched over category sets: a set must beat the best split found above
      CatSetSplit& cs = sets[(size_t)i];
      cs.gain = best.gain;
      for (int f = 0; f < F_; ++f) {
        const FeatureMeta& m = meta_[(size_t)f];
        if (m.categorical != kCategoricalSorted || !m.enabled || !node_feature_on(sp_, s.key, f, F_) ||
            !node_allows(sp_, s.node, f))
          continue;
        categorical_set_search(self + (size_t)f * kMaxBins, self + plane_ + (size_t)f * kMaxBins,
                               m.missing_bin >= 0 ? m.num_bin - 1 : m.num_bin, f, ctx, sp_, cat_, &cs);
      }
      SplitInfo& o = out[i];
      o.total_g = tg; o.total_h = th; o.total_cnt = tc;
    }
    if (mode_ == ParallelMode::Voting) vote_and_search(st, n, gunit, hunit, local.data(), sets.data(), out);
    if (mode_ == ParallelMode::Feature) {
      // exchange per-rank winners; every rank applies the same total order
      int P = comm_->size();
      std::vector<Cand> all((size_t)n * P);
      comm_->allgather_bytes(local.data(), all.data(), (size_t)n * sizeof(Cand), false, nullptr);
      for (int i = 0; i < n; ++i) {
        Cand best = all[(size_t)i];
        for (int r = 1; r < P; ++r) {
          const Cand& c = all[(size_t)r * n + i];
          if (cand_better(c.gain, c.key, best.gain, best.key)) best = c;
        }
        local[(size_t)i] = best;
      }
    }
    for (int i = 0; i < n; ++i) {
      const Cand& b = local[(size_t)i];
      SplitInfo& o = out[i];
      o.pad0 = 0;
      if (sets[(size_t)i].feature >= 0) {
        const CatSetSplit& cs = sets[(size_t)i];
        o.gain = cs.gain;
        o.feature = cs.feature;
        o.threshold = cs.num_cat;
        o.default_left = kCategoricalSetTask;
        o.left_g = cs.lg; o.left_h = cs.lh;
        o.left_cnt = est_count(cs.lh, make_scan_ctx(o.total_g, o.total_h, o.total_cnt, gunit, hunit, sp_));
        for (int w = 0; w < kCatSetWords; ++w) cat_sets_[(size_t)st[i].node * kCatSetWords + (size_t)w] = cs.bins[w];
      } else if (b.gain == ninf) {
        o.gain = ninf; o.feature = -1; o.threshold = 0; o.default_left = 0; o.left_g = o.left_h = 0; o.left_cnt = 0;
      } else {
        o.gain = b.gain;
        o.feature = (int32_t)(b.key >> 16);
        o.threshold = (int32_t)((b.key >> 1) & 0x7fff);
        o.default_left = (int32_t)(b.key & 1);
        o.left_g = b.lg; o.left_h = b.lh;
        o.left_cnt = est_count(b.lh, make_scan_ctx(o.total_g, o.total_h, o.total_cnt, gunit, hunit, sp_));
      }
    }
  }

  // Voting-parallel search (PV-Tree; upstream LightGBM's tree_learner=voting).  The slots hold LOCAL histograms
  // (siblings by local subtraction).  Per searched node: (1) every rank searches its local histogram - leaf-size
  // limits scaled down by the number of ranks - and votes for its top_k features by local gain; (2) the votes are
  // all-gathered and the 2 x top_k features with the most votes (ties: larger sum of local gains, then lower
  // index) are chosen, the same on every rank; (3) only those features' histogram rows, and the node's totals,
  // are summed over the ranks; (4) the split is searched in the summed rows with the global totals.  With
  // 2 x top_k >= the number of features the result is the data-parallel learner's.
  void vote_and_search(const ScanTask* st, int n, double gunit, double hunit, Cand* best_out, CatSetSplit* sets, SplitInfo* out) {
    const int P = comm_->size();
    const int k = std::max(1, std::min(top_k_, F_));
    const int n_sel = std::min(F_, 2 * k);
    const double ninf = -std::numeric_limits<double>::infinity();
    struct Vote { int32_t feature; float gain; };
    std::vector<Vote> mine((size_t)n * (size_t)k, Vote{-1, 0.0f});
    SplitParams local_sp = sp_;
    local_sp.min_data_in_leaf = std::max(1, (int)(sp_.min_data_in_leaf / P));
    local_sp.min_sum_hessian_in_leaf = sp_.min_sum_hessian_in_leaf / P;
    const bool ranged = !bounds_.empty();
    auto usable = [&](const ScanTask& s, int f) {
      return meta_[(size_t)f].enabled && node_feature_on(sp_, s.key, f, F_) && node_allows(sp_, s.node, f);
    };
#pragma omp parallel for schedule(dynamic) if (n * F_ >= 512)
    for (int i = 0; i < n; ++i) {
      const ScanTask& s = st[i];
      const int64_t* self = slot(s.slot);
      const int64_t* hd = self + 2 * plane_;
      const double* nb = ranged ? &bounds_[(size_t)kNodeInfoStride * (size_t)s.node] : nullptr;
      const LeafScanCtx ctx = make_scan_ctx(hd[1], hd[2], hd[0], gunit, hunit, local_sp, ranged ? nb[0] : -kInf, ranged ? nb[1] : kInf,
                                            ranged ? nb[2] : std::numeric_limits<double>::quiet_NaN());
      std::vector<std::pair<double, int>> gains;
      for (int f = 0; f < F_; ++f) {
        if (!usable(s, f)) continue;
        Cand c{ninf, ~0ull, 0, 0};
        scan_feature_with(local_sp, self + (size_t)f * kMaxBins, self + plane_ + (size_t)f * kMaxBins, f, ctx, s.key, &c);
        if (c.gain > ninf) gains.emplace_back(c.gain, f);
      }
      std::sort(gains.begin(), gains.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) {
        return a.first != b.first ? a.first > b.first : a.second < b.second;
      });
      for (int j = 0; j < k && j < (int)gains.size(); ++j)
        mine[(size_t)i * (size_t)k + (size_t)j] = Vote{gains[(size_t)j].second, (float)gains[(size_t)j].first};
    }
    std::vector<Vote> all((size_t)n * (size_t)k * (size_t)P);
    comm_->allgather_bytes(mine.data(), all.data(), mine.size() * sizeof(Vote), false, nullptr);
    // the chosen features of every node, then their local rows + the node's local totals in one buffer
    const size_t per_node = 4 + (size_t)n_sel * 2 * kMaxBins;
    std::vector<int32_t> chosen((size_t)n * (size_t)n_sel, -1);
    std::vector<int64_t> buf((size_t)n * per_node, 0);
    for (int i = 0; i < n; ++i) {
      std::vector<int> count((size_t)F_, 0);
      std::vector<double> weight((size_t)F_, 0.0);
      for (int r = 0; r < P; ++r)
        for (int j = 0; j < k; ++j) {
          const Vote& v = all[((size_t)r * (size_t)n + (size_t)i) * (size_t)k + (size_t)j];
          if (v.feature >= 0 && v.feature < F_) { ++count[(size_t)v.feature]; weight[(size_t)v.feature] += v.gain; }
        }
      std::vector<int> order;
      for (int f = 0; f < F_; ++f)
        if (count[(size_t)f] > 0) order.push_back(f);
      std::sort(order.begin(), order.end(), [&](int a, int b) {
        if (count[(size_t)a] != count[(size_t)b]) return count[(size_t)a] > count[(size_t)b];
        if (weight[(size_t)a] != weight[(size_t)b]) return weight[(size_t)a] > weight[(size_t)b];
        return a < b;
      });
      const int64_t* self = slot(st[i].slot);
      int64_t* dst = buf.data() + (size_t)i * per_node;
      const int64_t* hd = self + 2 * plane_;
      dst[0] = hd[0]; dst[1] = hd[1]; dst[2] = hd[2];
      for (int j = 0; j < n_sel && j < (int)order.size(); ++j) {
        const int f = order[(size_t)j];
        chosen[(size_t)i * (size_t)n_sel + (size_t)j] = f;
        std::memcpy(dst + 4 + (size_t)j * 2 * kMaxBins, self + (size_t)f * kMaxBins, kMaxBins * sizeof(int64_t));
        std::memcpy(dst + 4 + (size_t)j * 2 * kMaxBins + kMaxBins, self + plane_ + (size_t)f * kMaxBins, kMaxBins * sizeof(int64_t));
      }
    }
    comm_->allreduce_sum_i64(buf.data(), buf.size(), false, nullptr);
#pragma omp parallel for schedule(dynamic) if (n * n_sel >= 512)
    for (int i = 0; i < n; ++i) {
      const ScanTask& s = st[i];
      const int64_t* src = buf.data() + (size_t)i * per_node;
      const int64_t tc = src[0], tg = src[1], th = src[2];
      const double* nb = ranged ? &bounds_[(size_t)kNodeInfoStride * (size_t)s.node] : nullptr;
      const LeafScanCtx ctx = make_scan_ctx(tg, th, tc, gunit, hunit, sp_, ranged ? nb[0] : -kInf, ranged ? nb[1] : kInf,
                                            ranged ? nb[2] : std::numeric_limits<double>::quiet_NaN());
      Cand best{ninf, ~0ull, 0, 0};
      CatSetSplit& cs = sets[(size_t)i];
      auto rows_of = [&](int j) { return src + 4 + (size_t)j * 2 * kMaxBins; };
      for (int j = 0; j < n_sel; ++j) {
        const int f = chosen[(size_t)i * (size_t)n_sel + (size_t)j];
        if (f >= 0 && meta_[(size_t)f].categorical != kCategoricalSorted) scan_feature(rows_of(j), rows_of(j) + kMaxBins, f, ctx, s.key, &best);
      }
      cs.gain = best.gain;   // a category set must beat the best plain split
      for (int j = 0; j < n_sel; ++j) {
        const int f = chosen[(size_t)i * (size_t)n_sel + (size_t)j];
        if (f < 0 || meta_[(size_t)f].categorical != kCategoricalSorted) continue;
        const FeatureMeta& m = meta_[(size_t)f];
        categorical_set_search(rows_of(j), rows_of(j) + kMaxBins, m.missing_bin >= 0 ? m.num_bin - 1 : m.num_bin, f, ctx, sp_, cat_, &cs);
      }
      best_out[(size_t)i] = best;
      SplitInfo& o = out[i];
      o.total_g = tg; o.total_h = th; o.total_cnt = tc;
    }
  }

  static constexpr double kInf = std::numeric_limits<double>::infinity();
  int top_k_ = 20;
  std::vector<double> bounds_;   // [node][lo, hi, out]; empty unless monotone constraints / path smoothing are active
  std::vector<uint64_t> masks_;  // [node][words]; empty unless interaction constraints are active
  std::vector<uint64_t> cat_sets_;   // [node][kCatSetWords]: bins going left in the node's best split, when it is a set split
  CatParams cat_;
  const Dataset* root_;
  Comm* comm_;
  ParallelMode mode_;
  int F_, TF_;
  int64_t N_;
  size_t plane_, slot_words_;
  int capacity_splits_;
  std::vector<int64_t> pool_;
  std::vector<int32_t> arena_[2];
  std::vector<NodeRange> ranges_;
  std::vector<FeatureMeta> meta_;
  SplitParams sp_{};
  const int32_t* bag_ = nullptr;
  int64_t bag_n_ = 0;
  const int64_t* packed_ = nullptr;
  const int64_t* wide_g_ = nullptr;   // full-precision mode: 36-bit fixed-point gradients instead of packed_
  const int64_t* wide_h_ = nullptr;
  QuantScale qs_{0, 0};
};

// ---------------------------------------------------------------------------
// Host engine
// ---------------------------------------------------------------------------
std::vector<int32_t> base_rows_of(const Dataset& ds) {
  if (ds.used_indices()) return *ds.used_indices();
  std::vector<int32_t> r((size_t)ds.root_rows());
  std::iota(r.begin(), r.end(), 0);
  return r;
}

class HostEngine : public Engine {
 public:
  HostEngine(std::shared_ptr<Dataset> train, const Config& cfg, Comm* comm, ParallelMode mode)
      : train_(train), cfg_(cfg), comm_(comm), backend_(train.get(), cfg, comm, mode), grower_(cfg, train.get()) {
    if (!train_->has_host_bins()) throw std::runtime_error("host engine: the dataset has no host bin matrices");
    if (train_->root_label().empty()) throw std::runtime_error("training data has no label");
    N_ = train_->root_rows();
    base_ = base_rows_of(*train_);
    K_ = std::max(1, cfg.num_class);
    scores_.assign((size_t)N_ * (size_t)K_, 0.0);
    grad_.assign((size_t)N_ * (size_t)K_, 0.f);
    hess_.assign((size_t)N_ * (size_t)K_, 0.f);
    packed_.assign((size_t)N_, 0);
    bag_ = base_;
    int F = train_->num_features();
    missing_bin_.resize((size_t)F);
    for (int f = 0; f < F; ++f) missing_bin_[(size_t)f] = train_->inner_mapper(f).missing_bin;
#ifdef _OPENMP
    if (cfg.num_threads > 0) omp_set_num_threads(cfg.num_threads);
#endif
  }

  const char* name() const override { return "host"; }

  int add_valid(std::shared_ptr<Dataset> ds) override {
    if (ds->num_total_features() != train_->num_total_features())
      throw std::runtime_error("validation data has a different number of features");
    if (!ds->has_host_bins()) throw std::runtime_error("host engine: validation data has no host bin matrices");
    if (ds->root_label().empty()) throw std::runtime_error("validation data has no label");
    Valid v;
    v.ds = ds;
    v.rows = base_rows_of(*ds);
    v.stride = ds->root_rows();
    v.scores.assign((size_t)ds->root_rows() * (size_t)K_, 0.0);
    valid_.push_back(std::move(v));
    return (int)valid_.size() - 1;
  }
  int num_valid() const override { return (int)valid_.size(); }
  int64_t num_rows(int which) const override {
    return which < 0 ? (int64_t)base_.size() : (int64_t)valid_[(size_t)which].rows.size();
  }

  void label_stats(double* sum_wy, double* sum_w, double* sum_w_pos, bool sqrt_labels) override {
    const auto& y = train_->root_label();
    const auto& w = train_->root_weight();
    double a = 0, b = 0, c = 0;
    for (int32_t r : base_) {
      double ww = w.empty() ? 1.0 : w[(size_t)r];
      a += ww * (sqrt_labels ? sqrt_label(y[(size_t)r]) : y[(size_t)r]);
      b += ww;
      if (y[(size_t)r] > 0.0f) c += ww;
    }
    if (dist()) {
      double buf[3] = {a, b, c};
      comm_->allreduce_sum_f64(buf, 3, false, nullptr);
      a = buf[0]; b = buf[1]; c = buf[2];
    }
    *sum_wy = a; *sum_w = b; *sum_w_pos = c;
  }

  void reset_scores(double v) override {
    // initial scores: one vector for every class, or one per class ([class][root row])
    auto class_offset = [&](const std::vector<double>& is, int64_t rows) {
      return (int64_t)is.size() >= (int64_t)(cls_ + 1) * rows && (int64_t)is.size() > rows ? (size_t)cls_ * (size_t)rows : (size_t)0;
    };
    const auto& is = train_->root_init_score();
    const size_t off = class_offset(is, N_);
    double* s = sc();
    for (int32_t r : base_) s[r] = v + (is.empty() ? 0.0 : is[off + (size_t)r]);
    for (auto& vs : valid_) {
      const auto& vis = vs.ds->root_init_score();
      const size_t voff = class_offset(vis, vs.stride);
      double* t = vsc(vs);
      for (int32_t r : vs.rows) t[r] = v + (vis.empty() ? 0.0 : vis[voff + (size_t)r]);
    }
  }
  void scale_scores(int which, double f) override {
    if (which < 0) { double* s = sc(); for (int32_t r : base_) s[r] *= f; }
    else { auto& v = valid_[(size_t)which]; double* t = vsc(v); for (int32_t r : v.rows) t[r] *= f; }
  }
  void add_constant(int which, double c) override {
    if (which < 0) { double* s = sc(); for (int32_t r : base_) s[r] += c; }
    else { auto& v = valid_[(size_t)which]; double* t = vsc(v); for (int32_t r : v.rows) t[r] += c; }
  }

  void select_class(int k) override {
    if (k < 0 || k >= K_) throw std::runtime_error("select_class: class out of range");
    cls_ = k;
  }
  void class_weights(int num_class, std::vector<double>* out) override {
    const auto& y = train_->root_label();
    const auto& w = train_->root_weight();
    out->assign((size_t)num_class, 0.0);
    for (int32_t r : base_) {
      const int c = (int)y[(size_t)r];
      if (c < 0 || c >= num_class || (float)c != y[(size_t)r])
        throw std::runtime_error("multiclass labels must be integers in [0, num_class)");
      (*out)[(size_t)c] += w.empty() ? 1.0 : (double)w[(size_t)r];
    }
    if (dist()) comm_->allreduce_sum_f64(out->data(), out->size(), false, nullptr);
  }

  void compute_gradients(const ObjectiveSpec& obj) override {
    const auto& y = train_->root_label();
    const auto& w = train_->root_weight();
    int64_t n = (int64_t)base_.size();
    if (is_multiclass(obj.kind)) {
#pragma omp parallel for schedule(static) if (n > 16384)
      for (int64_t i = 0; i < n; ++i) {
        int32_t r = base_[(size_t)i];
        multiclass_grad_hess(obj, scores_.data(), N_, r, y[(size_t)r], w.empty() ? 1.0f : w[(size_t)r], grad_.data(),
                             hess_.data());
      }
      return;
    }
#pragma omp parallel for schedule(static) if (n > 16384)
    for (int64_t i = 0; i < n; ++i) {
      int32_t r = base_[(size_t)i];
      grad_hess(obj, scores_[(size_t)r], y[(size_t)r], w.empty() ? 1.0f : w[(size_t)r], &grad_[(size_t)r],
                &hess_[(size_t)r]);
    }
  }
  void set_gradients(const float* g, const float* h, int64_t n) override {
    if (n != (int64_t)base_.size()) throw std::runtime_error("custom gradients: wrong length");
    float* gc = gr();
    float* hc = he();
    for (int64_t i = 0; i < n; ++i) {
      int32_t r = base_[(size_t)i];
      gc[r] = g[i];
      hc[r] = h[i];
    }
  }
  void get_gradients(float* g, float* h) override {
    const size_t n = base_.size();
    for (int k = 0; k < K_; ++k)
      for (size_t i = 0; i < n; ++i) {
        g[(size_t)k * n + i] = grad_[(size_t)k * (size_t)N_ + (size_t)base_[i]];
        h[(size_t)k * n + i] = hess_[(size_t)k * (size_t)N_ + (size_t)base_[i]];
      }
  }

  void resample(double fraction, uint64_t seed, uint64_t iter, double neg_fraction) override {
    if (fraction >= 1.0 && (neg_fraction < 0.0 || neg_fraction >= 1.0)) { bag_ = base_; oob_.clear(); return; }
    bag_.clear(); oob_.clear();
    uint64_t salt = dist() ? ((uint64_t)comm_->rank() << 40) : 0;
    const float fr = (float)fraction, nfr = (float)(neg_fraction < 0.0 ? fraction : neg_fraction);
    const auto& y = train_->root_label();
    for (int32_t r : base_) {
      const float keep = y[(size_t)r] > 0.0f ? fr : nfr;
      if (rng_uniform(seed, iter, salt + (uint64_t)r) < keep) bag_.push_back(r); else oob_.push_back(r);
    }
  }

  void goss_sample(double top_rate, double other_rate, uint64_t seed, uint64_t iter) override {
    const int64_t n = (int64_t)base_.size();
    bag_.clear(); oob_.clear();
    if (n == 0) return;
    const GossPlan plan = goss_plan(n, top_rate, other_rate);
    std::vector<float> w((size_t)n);
    for (int64_t i = 0; i < n; ++i) w[(size_t)i] = goss_weight(grad_.data(), hess_.data(), base_[(size_t)i], K_, N_);
    std::vector<float> sorted = w;
    std::nth_element(sorted.begin(), sorted.begin() + (plan.top_k - 1), sorted.end(), std::greater<float>());
    const float threshold = sorted[(size_t)(plan.top_k - 1)];   // the top_k-th largest weight
    const uint64_t salt = dist() ? ((uint64_t)comm_->rank() << 40) : 0;
    for (int64_t i = 0; i < n; ++i) {
      const int32_t r = base_[(size_t)i];
      if (w[(size_t)i] >= threshold) { bag_.push_back(r); continue; }
      if (rng_uniform(seed, iter, salt + (uint64_t)r) < plan.keep_prob) {
        bag_.push_back(r);
        for (int k = 0; k < K_; ++k) {
          grad_[(size_t)k * (size_t)N_ + (size_t)r] *= plan.multiplier;
          hess_[(size_t)k * (size_t)N_ + (size_t)r] *= plan.multiplier;
        }
      } else {
        oob_.push_back(r);
      }
    }
  }

  void set_tree_index(int64_t i) override { tree_index_ = i; grower_.set_tree_index(i); }
  void update_split_config(const Config& c) override { grower_.update_split_config(c); }
  Tree train_tree(const std::vector<FeatureMeta>& meta) override {
    // quantisation scale from the largest |g| and h of the training rows
    float mx[2] = {0.f, 0.f};
    const float* gc = gr();
    const float* hc = he();
    for (int32_t r : base_) {
      mx[0] = std::max(mx[0], std::fabs(gc[r]));
      mx[1] = std::max(mx[1], hc[r]);
    }
    if (dist()) comm_->allreduce_max_f32(mx, 2, false, nullptr);
    QuantScale qs{quant_exponent(mx[0]), quant_exponent(mx[1])};
    tree_qs_ = qs;
    backend_.set_bag(bag_.data(), (int64_t)bag_.size());
    if (cfg_.use_quantized_grad) {
      quantize_rows_host(gc, hc, bag_.data(), (int64_t)bag_.size(), qs, packed_.data(),
                         rounding_spec(cfg_, tree_index_, dist() ? comm_->rank() : 0));
      backend_.set_gradients(packed_.data(), qs);
    } else {
      // full precision: the histograms take 36-bit fixed-point gradients (separate int64 arrays)
      const QuantScale wide{qs.kg + kWideExtraBits, qs.kh + kWideExtraBits};
      wide_g_.resize(packed_.size());
      wide_h_.resize(packed_.size());
      const int64_t n = (int64_t)bag_.size();
#pragma omp parallel for schedule(static) if (n > 65536)
      for (int64_t i = 0; i < n; ++i) {
        const int32_t r = bag_[(size_t)i];
        wide_g_[(size_t)r] = quantize_wide(gc[r], wide.kg);
        wide_h_[(size_t)r] = quantize_wide(hc[r], wide.kh);
      }
      backend_.set_wide_gradients(wide_g_.data(), wide_h_.data(), wide);
    }
    return grower_.grow(backend_, meta, &leaf_nodes_);
  }

  void add_trained_tree_to_train_scores(const Tree& tree) override {
    for (const LeafPiece& pc : leaf_nodes_) {
      const NodeRange& r = backend_.range(pc.node);
      const int32_t* rows = backend_.arena(r.buf) + r.begin;
      double v = tree.leaf_value[(size_t)pc.leaf];
      double* s = sc();
      for (int32_t i = 0; i < r.count; ++i) s[rows[i]] += v;
    }
    if (!oob_.empty()) traverse_add(train_->root(), oob_, sc(), tree, 1.0);
  }

  void leaf_residual_percentiles(const Tree& tree, double alpha, bool mape_scale,
                                 std::vector<double>* values) override {
    const int L = tree.num_leaves;
    const std::vector<double> keep = *values;
    const auto& label = train_->root_label();
    const auto& w = train_->root_weight();
    const bool weighted = mape_scale || !w.empty();
    std::vector<std::vector<WeightedResidual>> items((size_t)L);
    for (const LeafPiece& pc : leaf_nodes_) {
      const NodeRange& r = backend_.range(pc.node);
      const int32_t* rows = backend_.arena(r.buf) + r.begin;
      auto& v = items[(size_t)pc.leaf];
      for (int32_t i = 0; i < r.count; ++i) {
        const int32_t row = rows[i];
        float wt = w.empty() ? 1.0f : w[(size_t)row];
        if (mape_scale) wt = (float)((double)wt / std::max(1.0, std::fabs((double)label[(size_t)row])));
        v.push_back(WeightedResidual{(float)((double)label[(size_t)row] - sc()[row]), wt});
      }
    }
    std::vector<int> has((size_t)L, 0);
#pragma omp parallel for schedule(dynamic)
    for (int l = 0; l < L; ++l) {
      auto& v = items[(size_t)l];
      if (v.empty()) continue;
      has[(size_t)l] = 1;
      // equal residuals: order by weight as well, so the result does not depend on the row order
      std::sort(v.begin(), v.end(), [](const WeightedResidual& a, const WeightedResidual& b) {
        return a.value != b.value ? a.value < b.value : a.weight < b.weight;
      });
      if (weighted) (*values)[(size_t)l] = percentile_weighted(v.data(), (int64_t)v.size(), alpha);
      else (*values)[(size_t)l] = percentile_unweighted([&](int64_t i) { return (double)v[(size_t)i].value; },
                                                        (int64_t)v.size(), alpha);
    }
    merge_leaf_percentiles(comm_, *values, has, keep);
  }

  void add_tree_by_traversal(int which, const Tree& tree, double sign) override {
    if (which < 0) traverse_add(train_->root(), base_, sc(), tree, sign);
    else { auto& v = valid_[(size_t)which]; traverse_add(v.ds->root(), v.rows, vsc(v), tree, sign); }
  }

  std::vector<EvalResult> eval(int which, const std::vector<int32_t>& metrics, const ObjectiveSpec& obj) override {
    if (which < 0) {
      const auto& w = train_->root_weight();
      return eval_metrics_host(scores_.data(), train_->root_label().data(), w.empty() ? nullptr : w.data(),
                               base_.data(), (int64_t)base_.size(), metrics, obj, comm_, N_);
    }
    auto& v = valid_[(size_t)which];
    const auto& w = v.ds->root_weight();
    return eval_metrics_host(v.scores.data(), v.ds->root_label().data(), w.empty() ? nullptr : w.data(),
                             v.rows.data(), (int64_t)v.rows.size(), metrics, obj, comm_, v.stride);
  }

  void get_scores(int which, double* out) override {
    for (int k = 0; k < K_; ++k) {
      if (which < 0) {
        const size_t n = base_.size();
        for (size_t i = 0; i < n; ++i) out[(size_t)k * n + i] = scores_[(size_t)k * (size_t)N_ + (size_t)base_[i]];
      } else {
        auto& v = valid_[(size_t)which];
        const size_t n = v.rows.size();
        for (size_t i = 0; i < n; ++i) out[(size_t)k * n + i] = v.scores[(size_t)k * (size_t)v.stride + (size_t)v.rows[i]];
      }
    }
  }
  GrowStats grow_stats() const override { return grower_.last_stats(); }
  void sum_over_ranks_i64(int64_t* v, size_t n) override {   // feature-parallel ranks all hold every row
    if (dist() && (backend_.mode() == ParallelMode::Data || backend_.mode() == ParallelMode::Voting)) comm_->allreduce_sum_i64(v, n, false, nullptr);
  }
  void leaf_gradient_sums(const Tree& tree, std::vector<int64_t>* sums, int* kg, int* kh) override {
    const int L = tree.num_leaves;
    sums->assign((size_t)L * 2, 0);
    *kg = tree_qs_.kg + kWideExtraBits;
    *kh = tree_qs_.kh + kWideExtraBits;
    const float* gc = gr();
    const float* hc = he();
    for (const LeafPiece& pc : leaf_nodes_) {
      const NodeRange& r = backend_.range(pc.node);
      const int32_t* rows = backend_.arena(r.buf) + r.begin;
      int64_t sg = 0, sh = 0;
#pragma omp parallel for reduction(+ : sg, sh) schedule(static) if (r.count > 65536)
      for (int32_t i = 0; i < r.count; ++i) {
        sg += quantize_wide(gc[rows[i]], *kg);
        sh += quantize_wide(hc[rows[i]], *kh);
      }
      (*sums)[(size_t)pc.leaf * 2] += sg;
      (*sums)[(size_t)pc.leaf * 2 + 1] += sh;
    }
  }
  void sum_over_ranks(double* v, size_t n) override {
    if (dist()) comm_->allreduce_sum_f64(v, n, false, nullptr);
  }

 private:
  struct Valid {
    std::shared_ptr<Dataset> ds;
    std::vector<int32_t> rows;
    std::vector<double> scores;   // [class][root row]
    int64_t stride = 0;
  };
  // the selected class's vectors
  double* sc() { return scores_.data() + (size_t)cls_ * (size_t)N_; }
  float* gr() { return grad_.data() + (size_t)cls_ * (size_t)N_; }
  float* he() { return hess_.data() + (size_t)cls_ * (size_t)N_; }
  double* vsc(Valid& v) { return v.scores.data() + (size_t)cls_ * (size_t)v.stride; }
  bool dist() const { return comm_ && comm_->size() > 1; }

  void traverse_add(const Dataset* root, const std::vector<int32_t>& rows, double* scores, const Tree& tree,
                    double sign) {
    if (tree.num_leaves <= 1) {
      double v = tree.leaf_value[0] * sign;
      for (int32_t r : rows) scores[r] += v;
      return;
    }
    if (tree.split_feature_inner.empty() || tree.split_feature_inner[0] < 0)
      throw std::runtime_error("tree has no bin thresholds (loaded from a model file); cannot traverse binned data");
    const uint8_t* fm = root->bins_fm();
    int64_t NR = root->root_rows();
    int64_t n = (int64_t)rows.size();
#pragma omp parallel for schedule(static) if (n > 16384)
    for (int64_t i = 0; i < n; ++i) {
      int32_t r = rows[(size_t)i];
      int leaf = tree.predict_leaf_binned([&](int f) { return (int)fm[(size_t)f * (size_t)NR + (size_t)r]; },
                                          missing_bin_.data());
      scores[r] += sign * tree.leaf_value[(size_t)leaf];
    }
  }

  std::shared_ptr<Dataset> train_;
  Config cfg_;
  Comm* comm_;
  HostBackend backend_;
  TreeGrower grower_;
  int64_t N_ = 0;
  int K_ = 1, cls_ = 0;
  std::vector<int32_t> base_, bag_, oob_;
  std::vector<double> scores_;
  std::vector<float> grad_, hess_;
  std::vector<int64_t> packed_, wide_g_, wide_h_;
  int64_t tree_index_ = 0;
  QuantScale tree_qs_{0, 0};   // quantisation exponents of the tree being / just grown (16-bit form)
  std::vector<LeafPiece> leaf_nodes_;
  std::vector<int32_t> missing_bin_;
  std::vector<Valid> valid_;
};

}  // namespace

std::unique_ptr<Engine> make_host_engine(std::shared_ptr<Dataset> train, const Config& cfg, Comm* comm,
                                         ParallelMode mode) {
  return std::unique_ptr<Engine>(new HostEngine(train, cfg, comm, mode));
}

}  // namespace lga

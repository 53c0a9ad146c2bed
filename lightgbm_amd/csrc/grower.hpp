// Leaf-wise (best-first) tree growth under a num_leaves budget - the split
// strategy "LightGBM R.ipynb":17 refers to and r/gridsearchCV.R:95-96 tunes
// (num_leaves, min_data_in_leaf) - re-organised for a throughput device:
//
// Strict best-first growth is a chain of num_leaves-1 dependent steps.  Here
// the chain is replayed on the host over split gains that are already known,
// and it only stalls when the next leaf to split has not been partitioned /
// histogrammed yet.  At a stall the replay is continued virtually over every
// gain that is known - leaves that are already partitioned contribute their
// children - and every leaf this virtual replay reaches within the remaining
// budget is split in ONE batched backend round.  Speculation only decides
// WHAT is computed early: the real replay keeps exactly the splits strict
// best-first would have made, so the tree is identical, while the number of
// device round trips drops from num_leaves-1 to roughly the tree depth.  A
// leaf whose speculative split never gets its turn stays a leaf; its rows are
// then found in the ranges of its partitioned descendants (LeafPiece).
#pragma once
#include <limits>
#include "backend.hpp"
#include "config.hpp"
#include "dataset.hpp"
#include "tree.hpp"
#include <vector>

namespace lga {

// Rows [range of `node`] belong to final leaf `leaf`.
struct LeafPiece {
  int32_t node;
  int32_t leaf;
};

struct GrowStats {
  int rounds = 0;
  int materialized_splits = 0;
  int used_splits = 0;
};

class TreeGrower {
 public:
  TreeGrower(const Config& cfg, const Dataset* ds);
  // Grows one tree (leaf values NOT yet shrunk).  `pieces` lists the backend
  // nodes whose row ranges make up each leaf.
  Tree grow(TreeBackend& be, const std::vector<FeatureMeta>& meta, std::vector<LeafPiece>* pieces);
  // The same in two steps, for callers that start the trees of several boosters before waiting for any of them:
  // grow_begin returns true when the backend has prepared a whole-tree build (TreeBackend::whole_tree_begin) that
  // is waiting to be launched - by whole_tree_launch, or by a batched launch of many backends' trees at once.
  bool grow_begin(TreeBackend& be, const std::vector<FeatureMeta>& meta);
  Tree grow_finish(TreeBackend& be, const std::vector<FeatureMeta>& meta, std::vector<LeafPiece>* pieces);
  // Assembles a tree the backend grew outside begin / finish (one of a chunk of boosting iterations): `wt` as
  // whole_tree_finish reports it, `qs` the iteration's quantisation scales.  set_tree_index first.
  Tree assemble_whole(TreeBackend& be, const std::vector<FeatureMeta>& meta, const TreeBackend::WholeTree& wt, QuantScale qs,
                      std::vector<LeafPiece>* pieces);
  // True when trees over `meta` are plain ones a backend may grow whole (see grow_begin) and nothing adjusts the
  // gains from tree to tree (feature_contri, cost-effective gradient boosting).
  bool plain_trees(const std::vector<FeatureMeta>& meta) const;
  const GrowStats& last_stats() const { return stats_; }
  // Index of the tree about to be grown: seeds its per-node feature sampling.
  void set_tree_index(int64_t i) { tree_index_ = i; }
  // Parameters that may change between iterations (Booster::reset_parameter): the split search's thresholds
  // and regularisation, depth limit, per-node feature sampling.
  void update_split_config(const Config& c) {
    cfg_.lambda_l1 = c.lambda_l1; cfg_.lambda_l2 = c.lambda_l2;
    cfg_.min_gain_to_split = c.min_gain_to_split; cfg_.min_sum_hessian_in_leaf = c.min_sum_hessian_in_leaf;
    cfg_.min_data_in_leaf = c.min_data_in_leaf; cfg_.max_depth = c.max_depth; cfg_.max_delta_step = c.max_delta_step;
    cfg_.feature_fraction_bynode = c.feature_fraction_bynode; cfg_.path_smooth = c.path_smooth;
    cfg_.extra_trees = c.extra_trees; cfg_.cat_l2 = c.cat_l2;
  }
  // Data-parallel over `ranks` shards: a leaf's rows (and what splitting it for nothing costs a rank) are spread
  // over the ranks, so the row counts that make speculation cautious scale with the number of ranks.
  void set_row_shards(int ranks) { row_shards_ = ranks > 1 ? ranks : 1; }
  SplitParams split_params() const;   // pointers inside stay valid until the next grow()

 private:
  struct Node {
    int parent = -1;
    int depth = 0;
    int slot = 0;
    int left = -1, right = -1;   // children node ids once the leaf's split has been submitted
    bool ready = false;          // ... and once that round's results are in
    bool forced = false;         // its split comes from forcedsplits_filename, not from the search
    int forced_index = -1;       // entry of forced_ that applies to this node (-1: none)
    uint32_t key = kRootNodeKey; // position key (per-node feature sampling)
    double lo = -std::numeric_limits<double>::infinity();   // output range (monotone constraints)
    double hi = std::numeric_limits<double>::infinity();
    double out = std::numeric_limits<double>::quiet_NaN();   // the node's own output (monotone / path smoothing)
    SplitInfo best;
  };
  // per-feature gain adjustment (feature_contri, CEGB): SplitParams points into these; offsets follow the
  // features the model has used so far
  std::vector<double> gain_scale_, gain_offset_;
  std::vector<double> coupled_penalty_;   // per inner feature, 0 once the feature has been used
  void refresh_gain_adjustment();
  // forcedsplits_filename: the splits every tree starts with, as a flat tree (index 0 = the root's; empty: none)
  struct ForcedSplit {
    int feature = -1;        // column of the raw matrix
    double threshold = 0.0;  // rows with value <= threshold go left
    int left = -1, right = -1;
  };
  std::vector<ForcedSplit> forced_;
  void load_forced_splits(const std::string& path);
  // interaction constraints as bit sets over inner features (empty: none)
  std::vector<std::vector<uint64_t>> groups_;
  int mask_words_ = 0;
  Config cfg_;
  const Dataset* ds_;
  GrowStats stats_;
  SplitParams sp_cur_{};          // of the tree between grow_begin and grow_finish
  bool whole_ = false;            // ... which the backend grows in one go
  const TreeBackend::WholeTree* given_ = nullptr;   // assemble_whole: the result to assemble instead of asking the backend
  QuantScale given_qs_{0, 0};
  int64_t tree_index_ = 0;
  // the previous tree's chosen gains as suffix minima: prev_bar_[k] = smallest gain chosen from the k-th choice on
  std::vector<double> prev_bar_;
  // speculate on a leaf only when its gain reaches a multiple of that bar which rises with its row count
  double spec_lo_ = 0.2, spec_hi_ = 0.95;
  double spec_rows_lo_ = 131072.0, spec_rows_hi_ = 2097152.0;
  int row_shards_ = 1;
  // A round costs its latency whatever it holds: one that the rules above leave nearly empty is filled with the best
  // held-back leaves of at most fill_leaf_rows_ rows each, up to fill_rows_ rows in the round (LGA_FILL_ROWS,
  // LGA_FILL_LEAF_ROWS; 0 rows: off).  Trees whose gains fall short of the previous tree's otherwise run one leaf per round.
  double fill_rows_ = 524288.0, fill_leaf_rows_ = 131072.0;
  bool certain_splits_ = true;   // split leaves whose turn is certain ahead of time (LGA_CERTAIN_SPLITS=0: gain estimates only)
  double own_round_rows_ = 262144.0;   // leaves of at least this many rows (per rank) get a round of their own while places are free (LGA_OWN_ROUND_ROWS)
  bool spec_ratio_ = true;        // scale the previous cutoff by the ratio of the roots' best gains
  double prev_root_gain_ = 0.0;
};

}  // namespace lga

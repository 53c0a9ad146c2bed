// Shared plain-data types and arithmetic used by BOTH the host (g++) and the
// device (hipcc, gfx950) builds.  Everything that decides the shape of a tree
// (gradient quantisation, split gain, candidate ordering, count estimates)
// lives here so the CPU oracle and the HIP kernels execute the same IEEE
// operations on the same integers and therefore grow bit-identical trees.
//
// Build both sides with -ffp-contract=off: a fused multiply-add on one side
// only would break that parity.
#pragma once
#include <cstdint>
#include <cmath>
#include <limits>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LGA_HD __host__ __device__ __forceinline__
#else
#define LGA_HD inline
#endif

namespace lga {

// ---- geometry constants -------------------------------------------------
// Features are processed in tiles of 32: one row of one tile is a 32-byte
// record (two 16-byte loads per lane) and the per-workgroup LDS histogram of a
// tile is 32 x 256 x 8 B = 64 KiB, so two workgroups fit the 160 KiB LDS of a CU.
constexpr int kTileFeatures = 32;
constexpr int kMaxBins = 256;          // bins are stored as uint8
// Gradients/hessians are quantised to |q| <= 2^kQuantBits.  A workgroup
// accumulates packed (g:int32 | h:uint32) pairs in LDS, so it may add at most
// kMaxRowsPerPass rows before flushing: 2^16 * 2^14 = 2^30 < 2^31.
constexpr int kQuantBits = 16;
constexpr int kMaxRowsPerPass = 16384;
// The most a growth-round unit or histogram chunk may hold: 2^16 * 30720 < 2^31 still (|q| <= 2^16 exactly), and
// the unsigned hessian half has twice that room.  Full-precision cells are 64 bits wide: no limit from overflow.
constexpr int kMaxUnitRows = 30720;
constexpr int kMaxUnitRowsWide = 1 << 18;
constexpr int kSlotHeader = 8;         // int64 words appended to every histogram slot

// ---- plain data ----------------------------------------------------------
struct SplitParams {
  double lambda_l1;
  double lambda_l2;
  double min_gain_to_split;
  double min_sum_hessian_in_leaf;
  double max_delta_step;
  int32_t min_data_in_leaf;
  int32_t max_depth;  // <=0: unlimited
  double bynode_fraction;   // feature_fraction_bynode (>= 1: every enabled feature at every node)
  uint64_t bynode_seed;     // differs per tree
  int32_t extra_trees;      // 1: only one random threshold per (node, feature) is a candidate
  int32_t mask_words;       // 64-bit words per node of node_masks
  uint64_t extra_seed;      // differs per tree
  // Monotone constraints / path smoothing (device backend): kNodeInfoStride doubles per node - the output
  // range [lo, hi] and the node's own output - in the backend's memory space; nullptr when neither is active.
  const double* node_bounds;
  // Interaction constraints (device backend): mask_words words per node, bit f = inner feature f may split
  // the node; nullptr when there are none.
  const uint64_t* node_masks;
  double path_smooth;       // > 0: child outputs are pulled towards the parent's, less so the more rows they hold
  // Per-feature gain adjustment (feature_contri, cost-effective gradient boosting), in the backend's memory
  // space, indexed by inner feature; nullptr when unused.  A candidate's gain becomes
  // gain * feature_scale[f] - feature_offset[f] - split_penalty * rows_in_leaf.
  const double* feature_scale;
  const double* feature_offset;
  double split_penalty;
};
constexpr int kNodeInfoStride = 3;

// Power-of-two scales of the current boosting iteration: real = q * 2^-k.
struct QuantScale {
  int32_t kg;
  int32_t kh;
};

// Best split of one leaf (written by the scan, read by the host grower).
struct SplitInfo {
  double gain;          // improvement over the unsplit leaf; -inf when none
  int32_t feature;      // inner (used-feature) index, -1 when none
  int32_t threshold;    // rows with bin <= threshold go left
  int32_t default_left; // side taken by the missing-value bin
  int32_t pad0;
  int64_t left_g, left_h;     // quantised sums of the left child
  int64_t total_g, total_h;   // quantised sums of the leaf
  int64_t left_cnt;           // estimated rows on the left (hessian-proportional)
  int64_t total_cnt;          // exact rows in the leaf (global in data-parallel mode)
};
static_assert(sizeof(SplitInfo) == 72, "SplitInfo layout");

// SplitTask::default_left value of a categorical split: rows whose bin EQUALS `threshold` go left.
constexpr int32_t kCategoricalTask = 2;
// ... of a split on a SET of categories: rows whose bin is in the parent node's set (TreeBackend::node_cat_set,
// kCatSetWords words) go left.  SplitInfo::default_left carries the same value for such a split; its
// `threshold` is then the number of categories in the set.
constexpr int32_t kCategoricalSetTask = 3;
constexpr int kCatSetWords = 4;   // 256 bins
// FeatureMeta::categorical of a feature whose splits are searched over category sets (sorted by gradient
// statistics, on the host: cat_search.hpp) instead of one category against the rest by the scan.
constexpr int16_t kCategoricalSorted = 2;

// One leaf to split in a batched growth round.
struct SplitTask {
  int32_t parent_node;   // node-table index of the leaf being split
  int32_t left_node;     // node-table indices that receive the children
  int32_t right_node;
  int32_t feature;       // inner feature
  int32_t threshold;     // bin threshold
  int32_t default_left;
  int32_t missing_bin;   // bin holding missing values for that feature, -1 if none
  int32_t hist_child;    // 0: build the left child's histogram, 1: the right one
  int32_t parent_slot;   // histogram slot of the parent (inherited by the other child)
  int32_t new_slot;      // fresh slot for hist_child
  int32_t first_pchunk;  // first partition chunk of this task (prefix over tasks)
  int32_t num_pchunk;
  int32_t first_hchunk;  // first histogram chunk
  int32_t num_hchunk;
  int32_t depth;         // depth of the children
  int32_t parent_key;    // position key of the parent (uint32 bits; see node_feature_on)
  int32_t hist_rows;     // rows of hist_child over all ranks: an upper bound of the rows any rank histograms
};

// Row range of a node inside the two ping-pong row-index arenas.
struct NodeRange {
  int32_t begin;
  int32_t count;   // local rows
  int32_t buf;     // which arena holds the rows
  int32_t pad;
};

// One child whose best split is searched in a round.
struct ScanTask {
  int32_t node;        // node-table index (for the local row count)
  int32_t slot;        // slot that holds / receives this node's histogram
  int32_t parent_slot; // -1: histogram already in `slot`; else slot = parent - sibling
  int32_t sibling_slot;
  int32_t depth;
  uint32_t key;        // position key of the node (node_feature_on)
};

struct FeatureMeta {
  int32_t num_bin;      // bins in use (<=256)
  int32_t missing_bin;  // index of the NaN bin or -1
  int32_t enabled;      // 0 when masked out by feature_fraction for this tree
  int16_t monotone;     // +1: the output may not decrease with the feature, -1: may not increase, 0: free
  int16_t categorical;  // 1: bins are categories; a split sends ONE category left (one against the rest); 2: kCategoricalSorted
};
static_assert(sizeof(FeatureMeta) == 16, "FeatureMeta layout");

// ---- arithmetic ----------------------------------------------------------
LGA_HD double threshold_l1(double g, double l1) {
  if (l1 <= 0.0) return g;
  double a = fabs(g) - l1;
  if (a <= 0.0) return 0.0;
  return g > 0.0 ? a : -a;
}

// Un-normalised leaf objective reduction: T(G)^2 / (H + l2).
LGA_HD double leaf_gain(double g, double h, double l1, double l2) {
  double t = threshold_l1(g, l1);
  double d = h + l2;
  if (!(d > 0.0)) return 0.0;
  return (t * t) / d;
}

LGA_HD double leaf_output(double g, double h, double l1, double l2, double max_delta_step) {
  double d = h + l2;
  if (!(d > 0.0)) return 0.0;
  double out = -threshold_l1(g, l1) / d;
  if (max_delta_step > 0.0 && fabs(out) > max_delta_step) out = out > 0.0 ? max_delta_step : -max_delta_step;
  return out;
}

// 2^-k as an exact double.
LGA_HD double unit_of(int k) { return ldexp(1.0, -k); }

// Quantisation exponent: the largest k with vmax * 2^k < 2^kQuantBits.
LGA_HD int quant_exponent(float vmax) {
  if (!(vmax > 0.0f) || !(vmax < 3.0e38f)) return 0;
  int e;
  (void)frexpf(vmax, &e);  // vmax = m * 2^e, m in [0.5, 1)
  return kQuantBits - e;
}

LGA_HD int32_t quantize(float v, int k) {
  float s = ldexpf(v, k);
  const float lim = (float)(1 << kQuantBits);
  if (s > lim) s = lim;
  if (s < -lim) s = -lim;
  return (int32_t)rintf(s);
}

// Unbiased rounding of the 16-bit form (stochastic_rounding=true): floor(s + u) with u uniform in [0, 1), drawn from
// the counter-based generator below at (seed, stream, index) - the same draw on host and device, whatever the thread
// layout, so both engines still grow the same trees.
struct RoundingSpec {
  uint64_t seed;
  uint64_t stream;     // differs per tree
  uint64_t salt;       // added to the row id (distributed learners: ranks draw different numbers for their local rows)
  int32_t stochastic;  // 0: round to nearest
  int32_t pad;
  const int32_t* ext_ids;   // device engine with its rows in a private order: position -> the caller's row id (else null)
};
LGA_HD float rng_uniform(uint64_t seed, uint64_t stream, uint64_t index);
// which: 0 gradient, 1 hessian
LGA_HD int32_t quantize_with(float v, int k, const RoundingSpec& r, int which, uint64_t row) {
  if (!r.stochastic) return quantize(v, k);
  float s = ldexpf(v, k);
  const float lim = (float)(1 << kQuantBits);
  if (s > lim) s = lim;
  if (s < -lim) s = -lim;
  const double u = (double)rng_uniform(r.seed, r.stream * 2u + (uint64_t)which, r.salt + (r.ext_ids ? (uint64_t)(uint32_t)r.ext_ids[row] : row));
  return (int32_t)floor((double)s + u);
}

// Wide fixed point (leaf-value renewal, the CPU engine's full-precision histograms): kWideExtraBits more bits
// than the packed form, i.e. |q| <= 2^36; int64 sums of up to 2^26 rows cannot overflow.
constexpr int kWideExtraBits = 20;
LGA_HD int64_t quantize_wide(float v, int k) {
  return (int64_t)rint(ldexp((double)v, k));
}

// (g,h) packed for a single 64-bit add: g in the high, h in the low word.
// Sums stay separable while the low word does not carry (h >= 0, bounded rows).
LGA_HD int64_t pack_gh(int32_t g, int32_t h) {
  return (int64_t)(((uint64_t)(int64_t)g << 32) + (uint64_t)(uint32_t)h);
}
LGA_HD int32_t packed_g(int64_t p) { return (int32_t)(p >> 32); }
LGA_HD uint32_t packed_h(int64_t p) { return (uint32_t)((uint64_t)p & 0xffffffffull); }

// Ordering of split candidates: larger gain first; ties -> smaller key
// (feature, threshold, default_left).  A strict total order, so a parallel
// reduction and a sequential loop pick the same winner.
LGA_HD uint64_t cand_key(int feature, int threshold, int default_left) {
  return ((uint64_t)(uint32_t)feature << 16) | ((uint64_t)(uint32_t)threshold << 1) | (uint64_t)(default_left & 1);
}
LGA_HD bool cand_better(double gain_a, uint64_t key_a, double gain_b, uint64_t key_b) {
  if (gain_a > gain_b) return true;
  if (gain_a < gain_b) return false;
  return key_a < key_b;
}

// Per-leaf constants of a split search, computed once per (leaf) and shared by
// every candidate of that leaf.
struct LeafScanCtx {
  int64_t total_g, total_h, total_cnt;
  double gunit, hunit;
  double cnt_factor;    // rows per quantised hessian unit
  double parent_gain;
  double lo, hi;        // monotone constraints: the range this leaf's (and its descendants') outputs must stay in
  double node_out;      // path smoothing: this leaf's own output, which its children's outputs are pulled towards
  int32_t bounded;      // lo / hi are not both infinite
  int32_t smooth;       // path smoothing is on
};

LGA_HD double clamp_output(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
// Objective reduction of a leaf that is given the output `out` (equals leaf_gain at the unconstrained optimum).
LGA_HD double gain_given_output(double g, double h, double l1, double l2, double out) {
  const double t = threshold_l1(g, l1);
  return -(2.0 * t * out + (h + l2) * out * out);
}

// Path smoothing: a child with n rows gets raw * w + parent * (1 - w), w = (n / s) / (n / s + 1).
LGA_HD double smoothed_output(double raw, int64_t n, double parent_out, double path_smooth) {
  const double r = (double)n / path_smooth;
  return raw * r / (r + 1.0) + parent_out / (r + 1.0);
}

// node_out: the leaf's own output (NaN: take the unconstrained optimum - the root); only read with smoothing.
LGA_HD LeafScanCtx make_scan_ctx(int64_t tg, int64_t th, int64_t tc, double gunit, double hunit, const SplitParams& p,
                                 double lo = -std::numeric_limits<double>::infinity(),
                                 double hi = std::numeric_limits<double>::infinity(),
                                 double node_out = std::numeric_limits<double>::quiet_NaN()) {
  LeafScanCtx c;
  c.total_g = tg; c.total_h = th; c.total_cnt = tc;
  c.gunit = gunit; c.hunit = hunit;
  c.cnt_factor = th > 0 ? (double)tc / (double)th : 0.0;
  c.lo = lo; c.hi = hi;
  c.bounded = (lo > -std::numeric_limits<double>::infinity() || hi < std::numeric_limits<double>::infinity()) ? 1 : 0;
  c.smooth = p.path_smooth > 0.0 ? 1 : 0;
  const double G = (double)tg * gunit, H = (double)th * hunit;
  c.node_out = 0.0;
  if (c.smooth) {
    c.node_out = node_out == node_out ? node_out : leaf_output(G, H, p.lambda_l1, p.lambda_l2, p.max_delta_step);
    c.parent_gain = gain_given_output(G, H, p.lambda_l1, p.lambda_l2, c.node_out);
  } else if (c.bounded) {
    const double out = clamp_output(leaf_output(G, H, p.lambda_l1, p.lambda_l2, p.max_delta_step), lo, hi);
    c.parent_gain = gain_given_output(G, H, p.lambda_l1, p.lambda_l2, out);
  } else {
    c.parent_gain = leaf_gain(G, H, p.lambda_l1, p.lambda_l2);
  }
  return c;
}

// Estimated row count of a part of a leaf from its hessian share.
LGA_HD int64_t est_count(int64_t part_h, const LeafScanCtx& c) {
  return (int64_t)((double)part_h * c.cnt_factor + 0.5);
}

// Gain of one candidate split or -inf when a constraint fails.  monotone: the split feature's constraint.
// In a leaf with an output range, or on a constrained feature, the children's outputs are clamped to the
// range, must be ordered as the constraint asks, and the gain is the one those outputs really achieve.
LGA_HD double split_gain(int64_t lg, int64_t lh, const LeafScanCtx& c, const SplitParams& p, int monotone = 0) {
  const double ninf = -std::numeric_limits<double>::infinity();
  int64_t rg = c.total_g - lg, rh = c.total_h - lh;
  int64_t lc = est_count(lh, c);
  int64_t rc = c.total_cnt - lc;
  if (lc < p.min_data_in_leaf || rc < p.min_data_in_leaf) return ninf;
  double lhr = (double)lh * c.hunit, rhr = (double)rh * c.hunit;
  if (lhr < p.min_sum_hessian_in_leaf || rhr < p.min_sum_hessian_in_leaf) return ninf;
  double lgr = (double)lg * c.gunit, rgr = (double)rg * c.gunit;
  double gain;
  if (c.bounded || c.smooth || monotone != 0) {
    double lo_ = leaf_output(lgr, lhr, p.lambda_l1, p.lambda_l2, p.max_delta_step);
    double ro_ = leaf_output(rgr, rhr, p.lambda_l1, p.lambda_l2, p.max_delta_step);
    if (c.smooth) {
      lo_ = smoothed_output(lo_, lc, c.node_out, p.path_smooth);
      ro_ = smoothed_output(ro_, rc, c.node_out, p.path_smooth);
    }
    lo_ = clamp_output(lo_, c.lo, c.hi);
    ro_ = clamp_output(ro_, c.lo, c.hi);
    if ((monotone > 0 && lo_ > ro_) || (monotone < 0 && lo_ < ro_)) return ninf;
    gain = gain_given_output(lgr, lhr, p.lambda_l1, p.lambda_l2, lo_) + gain_given_output(rgr, rhr, p.lambda_l1, p.lambda_l2, ro_) -
           c.parent_gain;
  } else {
    gain = leaf_gain(lgr, lhr, p.lambda_l1, p.lambda_l2) + leaf_gain(rgr, rhr, p.lambda_l1, p.lambda_l2) - c.parent_gain;
  }
  if (!(gain > p.min_gain_to_split)) return ninf;
  return gain;
}

// The gain a candidate on feature f competes with (see SplitParams::feature_scale); -inf stays -inf, and a
// candidate whose adjusted gain does not clear min_gain_to_split is no candidate.  Monotone in `gain` for a
// fixed feature and leaf, so it may be applied to a feature's best candidate instead of to every one.
LGA_HD double adjust_feature_gain(double gain, int f, int64_t leaf_rows, const SplitParams& sp) {
  if (!sp.feature_scale || !(gain > -std::numeric_limits<double>::infinity())) return gain;
  const double scale = sp.feature_scale[f];
  if (!(scale > 0.0)) return -std::numeric_limits<double>::infinity();
  const double g = gain * scale - sp.feature_offset[f] - sp.split_penalty * (double)leaf_rows;
  return g > sp.min_gain_to_split ? g : -std::numeric_limits<double>::infinity();
}

// Counter-based RNG (splitmix64 finaliser over a (seed, stream, index) triple):
// the same draw on host and device, independent of thread layout.
LGA_HD uint64_t mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
LGA_HD uint64_t rng_u64(uint64_t seed, uint64_t stream, uint64_t index) {
  return mix64(mix64(seed * 0x2545f4914f6cdd1dull + stream) ^ (index * 0x9e3779b97f4a7c15ull));
}
// Uniform in [0,1) with 24 random bits (exactly representable in float).
LGA_HD float rng_uniform(uint64_t seed, uint64_t stream, uint64_t index) {
  return (float)(rng_u64(seed, stream, index) >> 40) * (1.0f / 16777216.0f);
}

// Gradient-based one-side sampling: the counts and factors both engines derive from the local row count.
struct GossPlan {
  int64_t top_k, other_k;
  float keep_prob;    // chance of a row outside the top to stay
  float multiplier;   // applied to the gradients / hessians of such a row
};
inline GossPlan goss_plan(int64_t n, double top_rate, double other_rate) {
  GossPlan p;
  p.top_k = (int64_t)((double)n * top_rate);
  p.other_k = (int64_t)((double)n * other_rate);
  if (p.top_k < 1) p.top_k = 1;
  if (p.other_k < 1) p.other_k = 1;
  if (p.top_k > n) p.top_k = n;
  const int64_t rest = n - p.top_k;
  p.keep_prob = rest > 0 ? (float)((double)p.other_k / (double)rest) : 0.0f;
  p.multiplier = rest > 0 ? (float)((double)rest / (double)p.other_k) : 1.0f;
  return p;
}
// A row's sampling weight: sum over classes of |g * h| in float arithmetic, class 0 first.
LGA_HD float goss_weight(const float* g, const float* h, int64_t row, int num_class, int64_t stride) {
  float w = 0.0f;
  for (int k = 0; k < num_class; ++k) w += fabsf(g[(int64_t)k * stride + row] * h[(int64_t)k * stride + row]);
  return w;
}

// Per-node feature sampling (feature_fraction_bynode).  A node is named by its POSITION in the tree - a key
// derived from its parent's key and its side - so the subset does not depend on the order in which leaves
// happen to be partitioned.  Every feature is kept with probability `bynode_fraction`, one (hashed from the
// key) always.
constexpr uint32_t kRootNodeKey = 1u;
LGA_HD uint32_t child_node_key(uint32_t parent_key, int side) {
  return (uint32_t)mix64(((uint64_t)parent_key << 1) | (uint64_t)(side & 1)) | 1u;
}
// Interaction constraints: may inner feature f split backend node `node`?
LGA_HD bool node_allows(const SplitParams& sp, int node, int f) {
  if (!sp.node_masks) return true;
  return (sp.node_masks[(size_t)node * (size_t)sp.mask_words + (size_t)(f >> 6)] >> (f & 63)) & 1ull;
}
LGA_HD bool node_feature_on(const SplitParams& sp, uint32_t key, int f, int num_features) {
  if (!(sp.bynode_fraction < 1.0)) return true;
  if ((int)(mix64(sp.bynode_seed ^ ((uint64_t)key << 24)) % (uint64_t)(num_features > 0 ? num_features : 1)) == f) return true;
  return rng_uniform(sp.bynode_seed, (uint64_t)key, (uint64_t)f) < (float)sp.bynode_fraction;
}

// extra_trees: the one threshold bin of feature f that is a candidate at the node with position key `key`
// (thresholds 0 .. n_thresholds-1 exist); -1 = every threshold is a candidate.
LGA_HD int extra_trees_threshold(const SplitParams& sp, uint32_t key, int f, int n_thresholds) {
  if (!sp.extra_trees || n_thresholds <= 1) return -1;
  return (int)(rng_u64(sp.extra_seed, (uint64_t)key, (uint64_t)f) % (uint64_t)n_thresholds);
}

LGA_HD double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }

}  // namespace lga

// Launch wrappers of the gfx950 training kernels (definitions in kernels.hip).
#pragma once
#include <string>
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../common.hpp"
#include "../objective.hpp"

namespace lga {
namespace k {

constexpr int kPartChunk = 2048;    // rows per partition workgroup (256 threads x 8)
constexpr int kPartThreads = 256;
constexpr int kPartCursorStride = 16;   // 64-bit words between the reservation cursors of two tasks
constexpr int kHistThreads = 512;   // 8 waves; 64 KiB LDS histogram -> 2 workgroups per CU
constexpr int kHistThreadsWide = 1024;   // full-precision histograms: 128 KiB of LDS -> 1 workgroup of 16 waves per CU

// Small rounds carry their task descriptors inside the kernel arguments.
constexpr int kInlineTasks = 48;   // 48 x 64 B: leaves room for the round's other arguments in the 4 KiB kernel-argument block
struct InlineTasks {
  SplitTask t[kInlineTasks];
  int32_t count;
  int32_t pad[3];
};
// Host-mapped result block: [completion flag (64 B) | SplitInfo x n_scan | left counts x n_tasks | QuantScale].
constexpr int kHostDataOffset = 64;
constexpr int kFusedScanMaxFeatures = 512;

// Histogram pool addressing.  A slot holds, per feature, 256 gradient words followed by 256 hessian words.
// The slots a round creates form one contiguous region that is laid out rank-chunk-major: for each of the
// `ranks` feature owners the chunk [slot of the round][feature of that owner][512] followed by a copy of the
// slot header (rows, sum g, sum h), so that ONE in-place reduce-scatter over the region leaves every rank with
// the reduced histograms of the features it scans plus the reduced totals.  ranks == 1: a slot is one
// contiguous block.  slot_table[slot] = {word offset of the slot's rank-0 chunk, word stride between the
// rank chunks}; the kernel that creates a slot writes its entry.
struct PoolLayout {
  int64_t* pool;
  int64_t* slot_table;
  int64_t slot_words;    // ranks * chunk_words
  int64_t chunk_words;   // fpr * 512 + kSlotHeader
  int32_t fpr;           // features per rank chunk
  int32_t ranks;
};
// The slots created by one round: [first_slot, first_slot + n_slots).
struct RoundSlots {
  int32_t first_slot;
  int32_t n_slots;
};

// Best candidate of one (leaf, feature) pair.
struct Cand {
  double gain;
  uint64_t key;
  int64_t lg, lh;
};

struct TreeNodeDev {       // internal node of a tree for binned traversal (16 bytes, one LDS read)
  int32_t feature;         // inner feature
  int32_t meta;            // threshold bin | default_left << 8 | categorical (bin == threshold) << 9 |
                           // category set (node_sets) << 10 | (missing_bin + 1) << 16
  int32_t left, right;     // >= 0 internal node, < 0: ~leaf
};

// Debug builds (-DLGA_DEBUG): kernels check scattered indices before use and record the first violation
// instead of faulting.  take_check_failure waits for the stream and returns a description of it ("" when
// there was none, and always in release builds); device_checks_enabled tells which build this is.
bool device_checks_enabled();
std::string take_check_failure(hipStream_t s);

struct LeafSpan {          // rows [begin, end) of arena buffer `buf` belong to final leaf `leaf`
  int32_t begin, end, buf, leaf;
  double value;
};

void gradients(const ObjectiveSpec& obj, const double* scores, const float* label, const float* weight,
               const int32_t* rows, int64_t n, float* g, float* h, uint32_t* maxbits, hipStream_t s);
// Multiclass: class k's scores / gradients of row r live at [k * stride + r]; fills every class.
void multiclass_gradients(const ObjectiveSpec& obj, const double* scores, int64_t stride, const float* label,
                          const float* weight, const int32_t* rows, int64_t n, float* g, float* h, hipStream_t s);
void multi_metric_sum(int32_t metric, const ObjectiveSpec& obj, const double* scores, int64_t stride, const float* label,
                      const float* weight, const int32_t* rows, int64_t n, double* out2, hipStream_t s);
// Gradient-based one-side sampling.  goss_weights: w[i] = goss_weight of row rows[i].  kth_largest: the k-th
// largest of n non-negative floats (exact, by three radix passes over the bit patterns) -> *out, all on the
// stream; scratch holds 2048 + 8 uint32.  goss_amplify: rows of the bag whose weight is below *threshold get
// their gradients and hessians (all classes) multiplied.
void goss_weights(const float* g, const float* h, int num_class, int64_t stride, const int32_t* rows, int64_t n,
                  float* w, hipStream_t s);
void kth_largest(const float* v, int64_t n, int64_t k, uint32_t* scratch, float* out, hipStream_t s);
void goss_amplify(float* g, float* h, int num_class, int64_t stride, const int32_t* rows, int64_t n,
                  const float* threshold, float multiplier, hipStream_t s);
void absmax(const float* g, const float* h, const int32_t* rows, int64_t n, uint32_t* maxbits, hipStream_t s);
// wide (full-precision histograms): packed[row] = the float bit patterns (gradient high, hessian low) and *qs = the
// exponents of their 36-bit fixed-point form (quantize_wide), which the histogram kernels apply on the way.
void quantize(const float* g, const float* h, const int32_t* rows, int64_t n, const uint32_t* maxbits, QuantScale* qs,
              int64_t* packed, bool wide, const RoundingSpec& rounding, hipStream_t s);
// Row lines (single feature tile): line r/3 = [3 x 32 bin bytes | 3 x 8-byte packed gradients | 8 spare];
// the 32 bin bytes of row r are rotated by r (byte j = feature (j + r) mod 32).
inline int64_t row_line_bytes(int64_t N) { return ((N + 2) / 3) * 128; }
void build_row_lines(const uint8_t* bins_rt_tile0, int64_t N, uint8_t* lines, hipStream_t s);
// Writes already packed gradients into the lines (what the root pass does itself when it quantises on the way).
void store_line_gradients(uint8_t* lines, const int64_t* packed, int64_t N, hipStream_t s);

// Partition (or bagging when task.feature < 0): a0/a1 are the two row-index buffers a node's `buf` selects.
// Row sampling predicate of a partition task with feature < 0.  feature == -1 (bagging): a row goes left
// (= stays in the bag) with probability `fraction`.  feature == -2 (gradient-based one-side sampling): rows
// whose weight goss_weight(g, h) reaches *threshold always stay, the others with probability `fraction`.
struct BagParams {
  uint64_t seed, iter, salt;
  float fraction;
  const float* g = nullptr;
  const float* h = nullptr;
  const float* threshold = nullptr;
  int num_class = 1;
  int64_t stride = 0;
  const float* label = nullptr;         // feature -3 (balanced bagging): rows with label > 0 stay with probability
  float neg_fraction = 0.0f;            // `fraction`, the others with `neg_fraction`
  const uint64_t* cat_sets = nullptr;   // growth rounds: kCatSetWords words per node, read for kCategoricalSetTask
  const int32_t* ext_ids = nullptr;     // rows in a private order: position -> the caller's row id, which keys the random draws
};
// Ordered gradients (data with several feature tiles): the packed gradients of a node's rows in the order of its row
// range - arena[b] runs parallel to row arena b - moved along with the row ids by the partition, so that the histogram
// workgroups of EVERY tile read them as a stream instead of gathering 8 bytes (a 64-byte sector) per row and tile.
// by_row: the gradients by row id, the source for the children of node 0, whose rows are the bag itself.
struct GradArenas {
  const int64_t* by_row = nullptr;
  int64_t* arena[2] = {nullptr, nullptr};
};
// tasks == nullptr: descriptors come from `inl`.  any_multi_chunk tells whether pass 2 is needed.
// zero_ptr/zero_words: int64 range the partition workgroups clear on the way (the round's new histogram slots).
void partition(const SplitTask* tasks, const int32_t* chunk_task, const InlineTasks& inl, int total_chunks,
               bool any_multi_chunk, NodeRange* nodes, const int32_t* a0_src, const int32_t* a1_src, int32_t* a0_dst,
               int32_t* a1_dst, const uint8_t* bins_fm, int64_t N, uint64_t* mask, int32_t* chunk_left,
               int32_t* task_left, BagParams bag, int64_t* zero_ptr, int64_t zero_words, hipStream_t s,
               const GradArenas& grads = GradArenas{});

// packed == nullptr: bins_rt points at row lines (see build_row_lines) that carry the gradients themselves.
// quant (row lines, root pass): quantise the float gradients inside the pass and store them in the lines.
struct FusedQuant {
  const float* g = nullptr;
  const float* h = nullptr;
  const uint32_t* maxbits = nullptr;
  QuantScale* qs_out = nullptr;
  RoundingSpec rounding{};   // 16-bit form only
};
// Feature-tile records + separate gradient array (data with more than one feature tile, feature-parallel).
// wide_qs != nullptr: full-precision histograms - `packed` holds float bit patterns, *wide_qs their 36-bit exponents.
void histogram(const SplitTask* tasks, const int32_t* chunk_task, const InlineTasks& inl, int total_chunks, int rows_per_chunk,
               const NodeRange* nodes, const int32_t* a0, const int32_t* a1, const uint8_t* bins_rt, int64_t N,
               const int64_t* packed, const PoolLayout& pool, RoundSlots round, int F, int tile_first,
               int tile_step, int tile_count, const QuantScale* wide_qs, hipStream_t s,
               const GradArenas& grads = GradArenas{});   // grads.arena: the histogram children's gradients in row-range order

// ---- fused growth round over row lines (single feature tile) ----
// One launch decides every row's side for all leaves of the round and builds the smaller children's
// histograms from the row lines.  A leaf of up to kSoloMaxRows rows is handled by ONE workgroup from start to
// finish: flags, row movement, histogram written straight to its slot (no atomics on global memory, no zeroing,
// no second launch).  A larger leaf is cut into units of `unit_rows` rows; its units add their LDS histograms
// to the (zeroed) slot and grow_scatter moves the rows.  The split search follows as scan_blocks.
#ifndef LGA_GROW_THREADS
#define LGA_GROW_THREADS 512
#endif
constexpr int kGrowThreads = LGA_GROW_THREADS;
constexpr int kGrowThreadsWide = 1024;   // full-precision histograms (GrowArgs::wide)
constexpr int kSubRows = 2048;        // rows per block of 32 mask words (4 rows per thread)
constexpr int kSoloMaxRows = 4096;    // upper bound of the solo limit: eight rows per lane (four with 1024 threads), one trip (the measured best limit is 2048; 8192 here cost sixteen row registers and spills)
constexpr int kFlagBlocks = 1;        // blocks of kPartChunk rows a working workgroup of grow_flags_kernel takes (measured: 4 is slower - a big leaf's launch is one wave of workgroups either way, and fewer, longer workgroups only serialise it)
constexpr int kGrowListRows = 3072;   // LDS list of a unit's histogram rows (worked off in windows of this many)
struct GrowArgs {
  const SplitTask* tasks;   // staged descriptors of a round above kInlineTasks leaves (n_staged of them), else null: inline
  int32_t n_staged;
  NodeRange* nodes;
  const int32_t* a0;        // row arenas a parent's `buf` selects (a0 is the bag itself for node 0)
  const int32_t* a1;
  int32_t* a0_dst;          // children go to the OTHER arena
  int32_t* a1_dst;
  uint8_t* lines;
  const uint8_t* bins_fm;   // [feature][row]: the split feature's bin of a row, one byte, coalesced while a leaf is dense
  int64_t N;
  PoolLayout pool;
  RoundSlots round;
  int32_t F;
  int32_t unit_rows;        // rows per unit of a multi-unit leaf (multiple of kSubRows; <= kMaxUnitRows, or kMaxUnitRowsWide with full-precision cells)
  uint64_t* mask;           // [mask block][32]: go-left bits of multi-unit leaves, read by grow_scatter
  int32_t* chunk_left;      // [mask block]
  int32_t* task_left;       // [task of the round]: rows that went left
  const uint64_t* cat_sets;
  FusedQuant quant;         // root pass: quantise the float gradients on the way (g != nullptr)
  int32_t wide;             // 1: full-precision histograms - the lines' gradient words are float bit patterns, added as 36-bit fixed point
  const QuantScale* qs;     // wide, non-root rounds: the exponents the root pass published
  unsigned long long* phase_ticks;   // LGA_GROW_TIMING: [8] sums of the workgroups' phase durations (100 MHz ticks), else null
  // single-pass partition (grow_partition): per task [rights reserved : 32 | lefts reserved : 32], zero between rounds,
  // kPartCursorStride words apart (atomics on one 128-byte line queue up behind each other: measured, 32 leaves' cursors
  // on 2 lines cost as much as one leaf's)
  unsigned long long* part_cursor;
  int32_t part_block_rows;           // 2048, 4096 or 8192 rows per workgroup
};
// Sides of the rows of the multi-unit leaves (total_chunks blocks of kPartChunk rows): masks + left counts for grow_scatter.
void grow_flags(const InlineTasks& inl, const GrowArgs& a, int total_chunks, hipStream_t s);
// Sides AND row movement of the multi-unit leaves in ONE launch (instead of grow_flags + grow_scatter): a block of
// kPartChunk rows keeps its rows in registers, reserves room for its lefts and rights with one atomic add on the
// leaf's cursor and moves its rows right away - lefts to the front of the leaf's range, rights to its END backwards
// (the total of the lefts is only known to the block that reserves last, which writes the children's ranges).  The
// blocks of a leaf land in the order of their reservations: nothing depends on the order of the rows inside a leaf
// (the histograms are integer sums).
void grow_partition(const InlineTasks& inl, const GrowArgs& a, int total_chunks, hipStream_t s);
// root: the single task's "leaf" is node 0 and nothing is partitioned.  total_units = sum of the tasks' units.
void grow_round(const InlineTasks& inl, const GrowArgs& a, int total_units, bool root, hipStream_t s);
// Moves the rows of the multi-unit leaves of a grow_round launch (total_chunks mask blocks).
void grow_scatter(const SplitTask* staged, int n_staged, const InlineTasks& inl, int total_chunks, NodeRange* nodes,
                  const int32_t* a0_src, const int32_t* a1_src, int32_t* a0_dst, int32_t* a1_dst, const uint64_t* mask,
                  const int32_t* chunk_left, int32_t* task_left, hipStream_t s);

// ---- split search as its own launch (distributed learners, data with several feature tiles) ----
// Which features a rank searches: block b covers features
//   [f_begin + (group_first + b * group_step) * group_size, + group_size)  cut at f_end.
struct ScanRange {
  int32_t f_begin, f_end;
  int32_t group_size, group_first, group_step;
  int32_t n_blocks;
};
struct ScanArgs {
  const SplitTask* tasks;   // staged descriptors, or null: inl
  int32_t task0;            // index of the launch's first task among the round's tasks
  int32_t n_scan;           // scan tasks of the whole round: 2 x tasks, or 1 (root)
  int32_t n_tasks;          // tasks of the whole round
  int32_t root;
  int32_t pairs;            // 1: one workgroup searches both children of a split (scan_pairs_kernel; wide data)
  const FeatureMeta* meta;
  int32_t F;
  PoolLayout pool;
  RoundSlots round;
  ScanRange range;
  int32_t my_rank;          // whose header copy holds this rank's (reduced) totals
  int64_t* node_totals;
  int64_t* scan_totals;     // [n_scan][4]: rows, sum g, sum h of every searched child (read by merge_winners)
  const QuantScale* qs;
  SplitParams sp;
  Cand* block_best;         // [n_scan][n_blocks]
  unsigned* tickets;        // [n_scan], zero between launches
  Cand* best_out;           // distributed: [n_scan] winners of this rank (then all-gather + merge_winners); else null
  SplitInfo* results;       // not distributed: [n_scan] in device memory; the last search copies them to host_out
  const int32_t* task_left;
  char* host_out;           // host-mapped result block
  unsigned* done_counter;
  unsigned seq;
  // 1: every search hands its own result to the host - SplitInfo written straight into the host-mapped block, one
  // system-scope fence, then `seq` into SplitInfo::pad0, which the host polls per result - instead of a second ticket,
  // a last workgroup that copies everything, and one flag (three dependent memory round trips shorter).  A task's
  // left count travels with its left child's record.  Not for the root, nor where winners are merged over ranks.
  // 2: small rounds go one step further - every BLOCK of features hands its own winner to the host (BlockResult in
  // `host_blocks`, [child][block]), which merges a child's blocks itself: no block ticket, no re-read of the blocks'
  // winners either (two more round trips).
  int32_t direct;
  char* host_blocks;
};
// One feature block's winner for one searched child, written by the block itself (ScanArgs::direct == 2).
struct BlockResult {
  double gain;
  uint64_t key;
  int64_t lg, lh;
  int64_t tc, tg, th;   // the child's totals
  int32_t left;         // block 0 of a left child: the task's local left count
  int32_t tag;          // the round's sequence number, stored last
};
static_assert(sizeof(BlockResult) == 64, "one cache line per block result");
constexpr int kDirectBlocksMax = 1024;   // children x blocks up to which a round's blocks report by themselves
// launch_scan: scan tasks of this launch (inl.count x 2, the staged tasks x 2, or 1 for the root)
void scan_blocks(const InlineTasks& inl, const ScanArgs& a, int launch_scan, hipStream_t s);

// Distributed learners: merge the ranks' winners (gathered [rank][scan task]) into SplitInfo, written to the
// host-mapped result block together with the left counts and the completion flag.
void merge_winners(const Cand* gathered, int ranks, int n_scan, int n_left, const int64_t* scan_totals, const QuantScale* qs,
                   const SplitParams& sp, const int32_t* task_left, char* host_out, unsigned seq, hipStream_t s);

// ---- voting-parallel search (PV-Tree; tree_learner=voting): rows sharded as for data-parallel, histograms LOCAL ----
// Per round: (1) vote_gains - every rank searches its local histograms (siblings by local subtraction; `a.sp` carries
// the leaf-size limits divided by the number of ranks) and records every feature's best local gain; (2) vote_topk -
// its top_k features per searched child, by gain then lower index; the votes are all-gathered; (3) vote_select - per
// child the 2 x top_k features with the most votes (ties: larger sum of the voted gains, added in rank order, then
// lower index), the same on every rank, and their local histogram rows plus the child's local totals gathered into
// one buffer, which is all-reduced; (4) vote_final - the split searched in the summed rows with the global totals
// (`a.sp`: the real limits); merge_winners (one "rank") hands the results to the host.  Same arithmetic and the
// same orders as host_engine.cpp: vote_and_search, so both engines grow the same trees.
struct Vote {
  int32_t feature;   // -1: no (further) feature with a valid split
  float gain;
};
constexpr int kVoteMaxFeatures = 4096;   // per-feature vote tallies live in LDS
constexpr int kVoteMaxVotes = 1024;      // ranks x top_k
constexpr int kVoteMaxSelected = 128;    // 2 x top_k
constexpr int64_t vote_words_per_child(int n_sel) { return 4 + (int64_t)n_sel * 2 * kMaxBins; }
// gains: [n_scan][F] doubles (-inf: no valid split); a.range must cover every feature in groups of 8
void vote_gains(const InlineTasks& inl, const ScanArgs& a, int n_scan, double* gains, hipStream_t s);
void vote_topk(const double* gains, int n_scan, int F, int top_k, Vote* votes, hipStream_t s);
// all: [rank][n_scan][top_k] votes; chosen: [n_scan][n_sel]; buf: [n_scan][vote_words_per_child(n_sel)]
void vote_select(const InlineTasks& inl, const ScanArgs& a, const Vote* all, int ranks, int n_scan, int top_k, int n_sel,
                 int32_t* chosen, int64_t* buf, hipStream_t s);
// best: [n_scan] winners; a.scan_totals receives the GLOBAL totals of every child (node_totals keep the local ones)
void vote_final(const InlineTasks& inl, const ScanArgs& a, int n_scan, const int32_t* chosen, const int64_t* buf, int n_sel,
                Cand* best, hipStream_t s);

// ---- whole tree in one launch (small data) ----
// ONE workgroup grows a complete tree, strictly best-first, without the host: it gathers the bag's rows into
// record arenas (row id, packed gradients, bin bytes - partitioned physically, so every later pass reads
// contiguous memory), and per split partitions the leaf's records, builds the smaller child's histogram in LDS,
// derives the sibling's and searches both children.  A launch with several descriptors grows one tree per
// workgroup: the fold boosters of a cross-validation, the rows of a parameter grid.  The results of a tree -
// the split sequence, every node's best split and row range, the 36-bit gradient sums of the final leaves -
// arrive in host-mapped memory behind a completion flag.
constexpr int kTreeThreads = 512;                          // 8 waves: 256 vector registers each for the fp64 gain code
constexpr int kTreeMaxRows = 65536;                        // go-left mask words kept in LDS
constexpr int kTreeMaxSplits = 255;                        // num_leaves <= 256
constexpr int kTreeMaxNodes = 2 * kTreeMaxSplits + 1;
// Host-mapped result block of one tree: this header, then - at the offsets of TreeLayout - the split sequence
// (node split by the k-th choice; its children are nodes 2k+1 and 2k+2), every node's best split and row range,
// and per node the sums of quantize_wide(g | h, k + kWideExtraBits) over its rows (final leaves only).
struct TreeHeader {
  unsigned flag;            // = seq once everything is written (a boosting chunk raises only its last tree's)
  int32_t n_splits;
  int32_t kg, kh;           // the iteration's quantisation exponents
  int32_t bag_n;            // boosting chunk: rows of the bag the tree was grown on
  int32_t pad[3];
  double metric[4][2];      // boosting chunk: per metric (sum of weighted losses, sum of weights) over the validation rows
  int64_t phase_ticks[8];   // TreeArgs::timing: 100 MHz ticks spent in root pass, selection, sides + histogram, prefix, search,
                            // row movement, per-split bookkeeping, hand-over
  int64_t shader_cycles;    // TreeArgs::timing: shader clock cycles of the whole build (with phase_ticks: the clock rate it ran at)
};
struct TreeLayout {
  int32_t off_split, off_best, off_range, off_wide, bytes;
};
inline TreeLayout tree_layout(int max_splits) {
  const int nodes = 2 * max_splits + 1;
  TreeLayout l;
  l.off_split = (int32_t)sizeof(TreeHeader);
  l.off_best = l.off_split + ((max_splits * 4 + 7) & ~7);
  l.off_range = l.off_best + nodes * (int32_t)sizeof(SplitInfo);
  l.off_wide = l.off_range + nodes * (int32_t)sizeof(NodeRange);
  l.bytes = (l.off_wide + nodes * 16 + 63) & ~63;
  return l;
}
struct TreeArgs {
  const int32_t* bag;       // root row ids of the tree's rows
  int32_t n;                // 1 .. kTreeMaxRows
  int32_t F;                // features (one tile)
  int32_t budget;           // splits at most (num_leaves - 1 <= kTreeMaxSplits)
  int32_t renew;            // 1: fill the per-node wide sums
  int32_t timing;           // 1: fill TreeHeader::phase_ticks
  int32_t pad0;
  const uint8_t* bins_rt;   // [root row][kTileFeatures] bin bytes
  const float* g;           // root-indexed float gradients / hessians and their absolute maxima (bit patterns)
  const float* h;
  const uint32_t* maxbits;
  QuantScale* qs_out;
  int32_t* idx[2];          // record arenas, ping-pong per node: row ids (the arenas leaf_add / leaf_sums read),
  unsigned long long* gh[2];   // packed quantised gradients,
  uint32_t* bins[2];        // bin bytes, BW 32-bit words per row (BW = 2, 4, 8 for F <= 8, 16, 32)
  const FeatureMeta* meta;
  SplitParams sp;
  int64_t* pool;            // histogram slots (one rank): slot s = pool + s * slot_words, feature f at + f * 512
  int64_t slot_words;
  NodeRange* nodes;         // device node table (kTreeMaxNodes entries from index 0)
  SplitInfo* best;          // device scratch, kTreeMaxNodes entries
  char* out;                // host-mapped result block (TreeHeader + arrays at `lay`)
  TreeLayout lay;
  unsigned seq;
};

// ---- boosting chunk: several whole ITERATIONS per launch ----
// The workgroup that grows a booster's trees also does everything between two trees: gradients of the objective
// at the current scores (with their maxima), the bag of the iteration, the tree, its leaf values (renewed from the
// 36-bit sums, times the learning rate), the score update of the training rows (in-bag rows through their leaf's
// row range, the others by walking the tree) and of ONE validation set, and that set's metric sums.  The host sees
// a chunk of iterations at once: per iteration a tree result block whose header also carries the metric sums.
// Arithmetic and summation order are those of the per-iteration kernels (gradients, leaf_add, traverse_add,
// metric_sum), so a chunk leaves the scores, trees and metric values the step-by-step path would.
struct BoostIter {          // what differs between the iterations of a chunk
  uint32_t enabled;         // feature_fraction mask of the iteration's tree
  uint32_t redraw;          // 1: draw a new bag before the tree (stream `bag_iter`)
  uint64_t bag_iter;
  SplitParams sp;           // the tree's split parameters (its per-tree seeds differ); read through scalar loads
};
struct BoostArgs {
  TreeArgs t;               // t.bag / t.n: the bag as it stands (a redraw rewrites bag_rows and the count); t.out: first block
  int32_t n_iters;
  int32_t n_base;           // training rows
  const int32_t* base_rows; // their root row ids
  int32_t* bag_rows;        // [bag | out-of-bag], n_base entries
  double* scores;           // root-indexed training scores
  const float* label;
  const float* weight;      // or null
  float* g;                 // = t.g / t.h, writable
  float* h;
  uint32_t* maxbits;
  ObjectiveSpec obj;
  int32_t bagging;          // 0: every row, 1: bagging_fraction, 3: by label (pos / neg fractions)
  float bag_fraction, neg_fraction;
  int32_t pad0;
  uint64_t bag_seed;
  double learning_rate;
  // one validation set (n_valid == 0: none)
  int32_t n_valid;
  int32_t n_metrics;
  int64_t valid_N;          // root rows of the validation data
  const int32_t* valid_rows;      // root row ids, or null = identity
  const uint8_t* valid_bins_fm;   // [feature][root row]
  double* valid_scores;
  const float* valid_label;
  const float* valid_weight;
  int32_t metrics[4];
  const uint8_t* bins_fm;   // training data, [feature][root row]: out-of-bag rows walk the tree
  int64_t N;                // root rows of the training data
  const BoostIter* iters;   // n_iters entries
};
void boost_chunks(const BoostArgs* descs, int n_boosters, int max_features, hipStream_t s);

constexpr int kTreeBatch = 16;    // descriptors per launch (kernel-argument block)
// descs: n_trees descriptors in DEVICE-visible memory (pinned host memory works: each workgroup reads its own once).
void grow_trees(const TreeArgs* descs, int n_trees, int max_features, hipStream_t s);

void leaf_add(const LeafSpan* leaves, int n_leaves, const int32_t* a0, const int32_t* a1, int64_t n_rows,
              double* scores, hipStream_t s);
// Leaf-value renewal: acc[2 * leaf] / acc[2 * leaf + 1] = sums over the leaf's rows (spans over the bag arena, sorted
// by begin) of quantize_wide(g, qs->kg + extra_bits) / quantize_wide(h, qs->kh + extra_bits); extra_bits is
// kWideExtraBits over 16-bit histograms and 0 when *qs already is the 36-bit scale (full-precision histograms).
void leaf_sums(const LeafSpan* spans, int n_spans, int n_leaves, const int32_t* a0, const int32_t* a1, int64_t n_rows,
               const float* g, const float* h, const QuantScale* qs, int extra_bits, int64_t* acc, hipStream_t s);
// Percentile leaf outputs (../percentile.hpp).  residual_keys: per in-bag row one sortable 64-bit key
// (leaf id << 32 | order-preserving bits of float(label - score)), padding = ~0; leaf_percentiles: per leaf
// the alpha-percentile of its (sorted) key segment and the segment length.  residual_items: the unsorted
// (leaf, residual, weight) triples for the weighted case, which the host ranks.
void residual_keys(const LeafSpan* spans, int n_spans, const int32_t* a0, const int32_t* a1, int64_t n_rows,
                   const double* scores, const float* label, uint64_t* keys, int64_t npad, hipStream_t s);
void leaf_percentiles(const uint64_t* keys, int64_t n_rows, int n_leaves, double alpha, double* values,
                      int32_t* counts, hipStream_t s);
void residual_items(const LeafSpan* spans, int n_spans, const int32_t* a0, const int32_t* a1, int64_t n_rows,
                    const double* scores, const float* label, const float* weight, int mape_scale,
                    int32_t* leaf_out, float* value_out, float* weight_out, hipStream_t s);
void traverse_add(const TreeNodeDev* nodes, int n_internal, const double* leaf_value, int num_leaves,
                  const uint8_t* bins_fm, int64_t N, const int32_t* rows, int64_t n, double sign, double* scores,
                  const uint64_t* node_sets, hipStream_t s);
// Large data, every row in the bag: one walk in storage order gives every row's leaf (leaf_of_row, indexed by root row)
// and the leaves' 36-bit gradient sums (acc as for leaf_sums); leaf_value_add then updates the scores by lookup.
// walk_leaves returns false when the tree does not fit the kernel's LDS (nothing launched).
// records != nullptr (data of ONE feature tile): the row-major 32-byte bin records; the walk then reads a row's bins once.
bool walk_leaves(const TreeNodeDev* nodes, int n_internal, int num_leaves, const uint8_t* bins_fm, const uint8_t* records, int64_t N,
                 const int32_t* rows, int64_t n, const float* g, const float* h, const QuantScale* qs, int extra_bits,
                 uint16_t* leaf_of_row, int64_t* acc, const uint64_t* node_sets, hipStream_t s);
// The walk, the score update and the next iteration's gradients in ONE pass over the rows (leaf values known before
// the walk: full-precision histograms, where nothing is renewed): a row's record is read once, walked to its leaf in
// LDS, and its score, gradient and hessian are written right there - no leaf ids, no leaf sums, no second and third
// pass.  Same arithmetic per row as leaf_value_add_gradients.  false: the tree does not fit the kernel's LDS.
bool walk_update_gradients(const ObjectiveSpec& obj, const TreeNodeDev* nodes, int n_internal, const double* leaf_value, int num_leaves,
                           const uint8_t* records, int64_t N, const int32_t* rows, int64_t n, double* scores, const float* label,
                           const float* weight, float* g, float* h, uint32_t* maxbits, const uint64_t* node_sets, hipStream_t s);
// leaf_value_add and the gradients of the next iteration (gradients()) in one pass.
void leaf_value_add_gradients(const ObjectiveSpec& obj, const uint16_t* leaf_of_row, const double* leaf_value, int num_leaves,
                              const int32_t* rows, int64_t n, double* scores, const float* label, const float* weight, float* g,
                              float* h, uint32_t* maxbits, hipStream_t s);
void leaf_value_add(const uint16_t* leaf_of_row, const double* leaf_value, int num_leaves, const int32_t* rows, int64_t n,
                    double* scores, hipStream_t s);
// node_sets: kCatSetWords words per internal node, read for nodes whose meta has bit 10 set (a categorical
// split on a set of bins); may be null when the tree has none.
// items: n_items x (slot, feature); out: n_items x 2 x kMaxBins (gradient row, hessian row).
void gather_feature_hist(const PoolLayout& pool, const int32_t* items, int n_items, int64_t* out, hipStream_t s);
void scale_add(double* scores, const int32_t* rows, int64_t n, double scale, double add, hipStream_t s);
void init_scores(double* scores, const double* init, const int32_t* rows, int64_t n, double v, hipStream_t s);
// Reduction outputs need kReduceDoubles doubles: results first, then per-workgroup partials.
constexpr int kReduceDoubles = 8 + 3 * 1024;
void metric_sum(int32_t metric, const ObjectiveSpec& obj, const double* scores, const float* label,
                const float* weight, const int32_t* rows, int64_t n, double* out2, hipStream_t s);
void label_sums(const float* label, const float* weight, const int32_t* rows, int64_t n, bool sqrt_labels, double* out3, hipStream_t s);
void gather_f64(const double* src, const int32_t* rows, int64_t n, double* dst, hipStream_t s);
void gather_f32(const float* src, const int32_t* rows, int64_t n, float* dst, hipStream_t s);
void scatter_f32(const float* src, const int32_t* rows, int64_t n, float* dst, hipStream_t s);
void iota(int32_t* dst, int64_t n, hipStream_t s);

// ---- ranking gradients (ranking.hip): one workgroup per query ----
constexpr int kMaxDeviceQuery = 4096;   // documents of the largest query the kernels hold in LDS
struct RankArgs {
  const double* score;        // root-indexed scores
  const float* label;         // root-indexed relevance grades
  const float* weight;        // root-indexed, or null
  const int32_t* to_root;     // dataset row -> root row, or null when they coincide
  const int64_t* bounds;      // nq + 1 dataset-row offsets of the queries
  const double* inv_max_dcg;  // per query (lambdarank)
  const double* label_gain;   // per grade
  const double* discount;     // per 0-based rank, max_query entries
  float* g;
  float* h;                   // root-indexed outputs
  int nq, max_query;
  double sigmoid;
  int norm, trunc;
  uint64_t seed, iter;        // rank_xendcg noise stream
};
constexpr int kMaxRankCutoffs = 8;
struct RankCutoffs {
  int n;
  int k[kMaxRankCutoffs];
};
// ndcg@k / map@k per query (score, label, to_root, bounds, label_gain, discount of `a` are read);
// out: nq x [ndcg x at.n | map x at.n].  Labels are grades below 64 with a label_gain entry each.
void rank_metrics(const RankArgs& a, const RankCutoffs& at, double* out, hipStream_t s);
void lambdarank_gradients(const RankArgs& a, hipStream_t s);
void xendcg_gradients(const RankArgs& a, hipStream_t s);

}  // namespace k
}  // namespace lga

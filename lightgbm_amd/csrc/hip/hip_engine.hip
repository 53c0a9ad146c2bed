// HIP backend + engine: everything row-sized stays in HBM for the whole
// training run; per growth round the host sends one descriptor block down and
// receives the children's best splits back (one synchronisation per round).
#include "hip_engine.hpp"
#include "kernels.hpp"
#include "../metric.hpp"
#include "../cat_search.hpp"
#include "../percentile.hpp"
#include "sort.hpp"
#include "locality.hpp"
#include <algorithm>
#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <sstream>
#include <thread>

namespace lga {

// ---------------------------------------------------------------------------
// allocator / stream hooks
// ---------------------------------------------------------------------------
namespace {
DeviceAllocFn g_alloc = nullptr;
DeviceFreeFn g_free = nullptr;
// The stream is a per-thread setting (like torch's current stream): fold boosters of one CV run are
// driven from their own host threads on their own streams.  Threads that never set one use the
// stream the first caller installed.
hipStream_t g_default_stream = nullptr;
thread_local hipStream_t t_stream = nullptr;
thread_local bool t_stream_set = false;
}  // namespace

void set_device_allocator(DeviceAllocFn a, DeviceFreeFn f) { g_alloc = a; g_free = f; }
void* device_alloc(size_t nbytes) {
  if (nbytes == 0) nbytes = 16;
  if (g_alloc) {
    void* p = g_alloc((int64_t)nbytes);
    if (!p) throw std::runtime_error("device allocation of " + std::to_string(nbytes) + " bytes failed");
    return p;
  }
  void* p = nullptr;
  LGA_HIP_CHECK(hipMalloc(&p, nbytes));
  return p;
}
void device_free(void* p) {
  if (!p) return;
  if (g_free) g_free(p); else (void)hipFree(p);
}
namespace {
struct HostBlockCache {
  std::mutex mu;
  std::map<size_t, std::vector<void*>> free_blocks[2];   // [mapped][capacity]
  size_t small_bytes = 0;                                // bytes of cached blocks of at most 1 MiB
  // blocks stay allocated until the process ends: the runtime may already be shutting down when statics die
};
HostBlockCache& host_block_cache() {
  static HostBlockCache* c = new HostBlockCache();
  return *c;
}
}  // namespace

void* host_block_acquire(size_t bytes, bool mapped, size_t* capacity) {
  size_t cap = 256;
  while (cap < bytes) cap <<= 1;
  *capacity = cap;
  HostBlockCache& c = host_block_cache();
  {
    std::lock_guard<std::mutex> lock(c.mu);
    auto& v = c.free_blocks[mapped ? 1 : 0][cap];
    if (!v.empty()) {
      void* p = v.back();
      v.pop_back();
      if (cap <= ((size_t)1 << 20)) c.small_bytes -= cap;
      return p;
    }
  }
  void* p = nullptr;
  LGA_HIP_CHECK(hipHostMalloc(&p, cap, mapped ? (hipHostMallocMapped | hipHostMallocCoherent) : hipHostMallocDefault));
  return p;
}
void host_block_release(void* p, size_t capacity, bool mapped) {
  if (!p) return;
  HostBlockCache& c = host_block_cache();
  {
    std::lock_guard<std::mutex> lock(c.mu);
    auto& v = c.free_blocks[mapped ? 1 : 0][capacity];
    // Small blocks are kept up to a byte budget, large ones four per size class; what exceeds that is freed:
    // page-locked memory must not grow to the peak of everything that ever lived at once.  The budget holds the
    // blocks of the few hundred boosters a parameter grid keeps alive (a hipHostFree costs ~1.4 ms, and a grid
    // releases its boosters by the hundred).
    static const size_t budget = [] {
      const char* e = std::getenv("LGA_HOST_CACHE_MB");
      return (size_t)std::max(0, e ? std::atoi(e) : 512) << 20;
    }();
    const bool small = capacity <= ((size_t)1 << 20);
    if (small ? c.small_bytes + capacity <= budget : v.size() < 4) {
      v.push_back(p);
      if (small) c.small_bytes += capacity;
      return;
    }
  }
  (void)hipHostFree(p);
}

void set_current_stream(hipStream_t s) {
  static bool have_default = false;
  if (!have_default) { g_default_stream = s; have_default = true; }
  t_stream = s;
  t_stream_set = true;
}
hipStream_t current_stream() { return t_stream_set ? t_stream : g_default_stream; }

namespace {
struct RoctxApi {
  bool tried = false, on = false;
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
RoctxApi g_roctx;
}  // namespace

void trace_enable(bool on) {
  if (on && !g_roctx.tried) {
    g_roctx.tried = true;
    for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      g_roctx.push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      g_roctx.pop = (int (*)())dlsym(h, "roctxRangePop");
      if (g_roctx.push && g_roctx.pop) break;
    }
  }
  g_roctx.on = on && g_roctx.push && g_roctx.pop;
}
void trace_push(const char* name) { if (g_roctx.on) (void)g_roctx.push(name); }
void trace_pop() { if (g_roctx.on) (void)g_roctx.pop(); }

namespace {

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------
// profiling: HIP events around the phases of a round (profile=true only)
// ---------------------------------------------------------------------------
struct PhaseProfile {
  bool on = false;
  enum { kUpload, kPartition, kHist, kComm, kScan, kFetch, kCount };
  double ms[kCount] = {0, 0, 0, 0, 0, 0};
  double host_ms = 0;
  int64_t rounds = 0, trees = 0, tasks = 0;
  // host wall clock (ms): enqueueing a round, waiting for its flag, and the time between rounds (replay)
  double host_launch_ms = 0, host_wait_ms = 0, host_between_ms = 0;
  std::chrono::steady_clock::time_point last_end{};
  int64_t part_rows = 0, hist_rows = 0;   // rows partitioned / rows accumulated into histograms (without the roots)
  hipEvent_t ev[kCount + 1];
  bool created = false;
  void init() {
    if (!on || created) return;
    for (auto& e : ev) LGA_HIP_CHECK(hipEventCreate(&e));
    created = true;
  }
  void mark(int i, hipStream_t s) { if (on) LGA_HIP_CHECK(hipEventRecord(ev[i], s)); }
  void collect(int upto) {
    if (!on) return;
    for (int i = 0; i < upto; ++i) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) == hipSuccess) ms[i] += t;
    }
  }
  ~PhaseProfile() { if (created) for (auto& e : ev) (void)hipEventDestroy(e); }
  std::string json() const {
    std::ostringstream os;
    os << "{\"rounds\":" << rounds << ",\"trees\":" << trees << ",\"split_tasks\":" << tasks << ",\"partition_rows\":" << part_rows
       << ",\"hist_rows\":" << hist_rows << ",\"upload_ms\":" << ms[kUpload]
       << ",\"partition_ms\":" << ms[kPartition] << ",\"hist_ms\":" << ms[kHist] << ",\"comm_ms\":" << ms[kComm]
       << ",\"scan_ms\":" << ms[kScan] << ",\"fetch_ms\":" << ms[kFetch] << ",\"host_launch_ms\":" << host_launch_ms
       << ",\"host_wait_ms\":" << host_wait_ms << ",\"host_between_rounds_ms\":" << host_between_ms << "}";
    return os.str();
  }
};

// ---------------------------------------------------------------------------
// backend
// ---------------------------------------------------------------------------
class HipBackend : public TreeBackend {
 public:
  HipBackend(const Dataset* ds, DeviceMirror* mirror, const Config& cfg, Comm* comm, ParallelMode mode,
             int64_t max_rows)
      : root_(ds->root()), mir_(mirror), comm_(comm), mode_(mode) {
    F_ = root_->num_features();
    T_ = root_->num_tiles();
    N_ = root_->root_rows();
    // LGA_DP_KEEP_SINGLE_RANK=1 keeps the distributed learner for a communicator of ONE rank (the collectives, the
    // streams and events around them, the distributed search hand-over all run; the sums change nothing): the way to
    // time that code path on a machine with a single GPU
    const char* keep_single = std::getenv("LGA_DP_KEEP_SINGLE_RANK");
    const bool keep_dist = comm_ && comm_->size() == 1 && keep_single && keep_single[0] == '1' && mode_ != ParallelMode::Serial;
    if (!comm_ || (comm_->size() <= 1 && !keep_dist)) mode_ = ParallelMode::Serial;
    else if (mode_ == ParallelMode::Serial) mode_ = ParallelMode::Data;
    top_k_ = std::max(1, std::min(cfg.top_k, std::max(F_, 1)));
    if (mode_ == ParallelMode::Voting &&
        (F_ > k::kVoteMaxFeatures || comm_->size() * top_k_ > k::kVoteMaxVotes || std::min(F_, 2 * top_k_) > k::kVoteMaxSelected)) {
      std::fprintf(stderr, "[lightgbm_amd] tree_learner=voting: %d features / %d ranks x top_k %d exceed the vote kernels' tables; "
                           "training with the exact data-parallel learner\n", F_, comm_->size(), top_k_);
      mode_ = ParallelMode::Data;
    }
    // Histogram pool layout (kernels.hpp: PoolLayout): the data-parallel learner splits every slot into one
    // chunk per rank, so that one in-place reduce-scatter per round hands each rank the features it searches.
    // Narrow data (a slot of at most 512 KiB: up to four feature tiles) is summed WHOLE instead: one all-reduce per
    // round and every rank searches every feature - the exchange is latency-bound there, and this way a round costs
    // one collective instead of two (reduce-scatter + winner all-gather) and no merge launch.  Wide data keeps the
    // reduce-scatter, whose traffic per link is 1/P of an all-reduce's.  LGA_DP_EXCHANGE=allreduce / reduce_scatter
    // overrides the size rule (for A/B runs on a multi-GPU node).
    dp_allreduce_ = mode_ == ParallelMode::Data && (size_t)T_ * kTileFeatures * 2 * kMaxBins * sizeof(int64_t) <= (size_t)512 << 10;
    if (const char* e = std::getenv("LGA_DP_EXCHANGE")) {
      const std::string v = e;
      if (v == "allreduce") dp_allreduce_ = mode_ == ParallelMode::Data;
      else if (v == "reduce_scatter") dp_allreduce_ = false;
      else throw std::runtime_error("LGA_DP_EXCHANGE is allreduce or reduce_scatter");
    }
    lay_.ranks = (mode_ == ParallelMode::Data && !dp_allreduce_) ? comm_->size() : 1;
    lay_.fpr = lay_.ranks > 1 ? std::max(1, (F_ + lay_.ranks - 1) / lay_.ranks) : std::max(1, T_ * kTileFeatures);
    lay_.chunk_words = (int64_t)lay_.fpr * 2 * kMaxBins + kSlotHeader;
    lay_.slot_words = lay_.chunk_words * lay_.ranks;
    slot_words_ = lay_.slot_words;
    int L = std::max(cfg.num_leaves, 2);
    max_batch_ = L - 1;
    // Histogram slots for 4 x (L - 1) splits: a tree that wastes many speculative splits must not run out of them, or
    // its last rounds are allowed one leaf each (with 3 x the 500-tree benchmark had trees of up to 99 rounds, with 4 x
    // the worst has 50 and nothing hits the limit: 2.46 -> 2.42 ms per tree; LGA_POOL_FACTOR)
    int pool_factor = 4;
    if (const char* e = std::getenv("LGA_POOL_FACTOR")) pool_factor = std::max(2, std::min(8, std::atoi(e)));
    capacity_splits_ = pool_factor * (L - 1) + 1;
    // keep the histogram pool below ~24 GiB
    while (capacity_splits_ > (L - 1) + 1 && (double)(capacity_splits_ + 1) * slot_words_ * 8.0 > 24.0 * (1u << 30))
      capacity_splits_ -= 1;
    stream_ = current_stream();
    pool_.resize((size_t)(capacity_splits_ + 1) * slot_words_);
    slots_dirty_ = capacity_splits_ + 1;   // the first tree clears the whole pool
    slot_table_.resize((size_t)2 * (size_t)(capacity_splits_ + 1));
    lay_.pool = pool_.get();
    lay_.slot_table = slot_table_.get();
    int n_nodes = 2 * capacity_splits_ + 2;
    nodes_.resize((size_t)n_nodes + 2);   // + the bagging partition's scratch entries
    node_totals_.resize((size_t)n_nodes * 4);
    ranges_host_.resize((size_t)n_nodes);
    arena_[0].resize((size_t)std::max<int64_t>(max_rows, 1));
    arena_[1].resize((size_t)std::max<int64_t>(max_rows, 1));
    max_pchunks_ = ceil_div(std::max<int64_t>(max_rows, 1), k::kPartChunk) + max_batch_ + 1;
    max_hchunks_ = ceil_div(std::max<int64_t>(max_rows, 1), 1024) + max_batch_ + 1;
    // one staging block per round context: [tasks | pchunk table | hchunk table]
    off_tasks_ = 64;   // the first 64 bytes stage one NodeRange
    off_pch_ = off_tasks_ + sizeof(SplitTask) * (size_t)(max_batch_ + 1);
    off_hch_ = off_pch_ + sizeof(int32_t) * (size_t)max_pchunks_;
    stage_bytes_ = off_hch_ + sizeof(int32_t) * (size_t)max_hchunks_;
    stage_bytes_ = (stage_bytes_ + 255) & ~(size_t)255;
    cat_.cat_smooth = cfg.cat_smooth;
    cat_.cat_l2 = cfg.cat_l2;
    cat_.max_cat_threshold = cfg.max_cat_threshold;
    cat_.min_data_per_group = cfg.min_data_per_group;
    prof.on = cfg.profile;
    if (cfg.profile || std::getenv("LGA_ROCTX")) trace_enable(true);
    // Which features this rank searches (kernels.hpp: ScanRange).
    scan_range_ = k::ScanRange{0, F_, 64, 0, 1, std::max(1, (F_ + 63) / 64)};
    if (mode_ == ParallelMode::Data && !dp_allreduce_) {
      const int r = comm_->rank();
      const int f0 = std::min(F_, r * lay_.fpr), f1 = std::min(F_, (r + 1) * lay_.fpr);
      scan_range_ = k::ScanRange{f0, f1, 64, 0, 1, std::max(1, (f1 - f0 + 63) / 64)};
    } else if (mode_ == ParallelMode::Feature) {
      // feature tiles are dealt round-robin: rank r owns tiles r, r + P, ...
      const int P = comm_->size(), r = comm_->rank();
      const int owned = r < T_ ? (T_ - r + P - 1) / P : 0;
      scan_range_ = k::ScanRange{0, F_, kTileFeatures, r, P, std::max(1, owned)};
      if (owned == 0) scan_range_.f_end = 0;   // nothing to search: one empty block keeps the hand-over alive
    }
    // Rounds in flight: ONE by default.  A second round lets one round's latency-bound tail (split search, result
    // hand-over, host replay) overlap with the other's row-level work, but a level's leaves are then dealt over two
    // rounds, and since leaves whose turn is certain are split ahead of time the rounds are few and full: measured
    // on the 10.5 M x 28 benchmark, 500 trees take 1.57 s with one round in flight and 1.60 s with two (three
    // interleaved repeats each), the same within noise from 0.6 M to 2.6 M rows.  The distributed learners would
    // pay for a second round twice over: each round brings a collective of its own on the shared collective stream
    // plus two event hand-overs between streams.  LGA_ROUNDS_IN_FLIGHT=2..4 keeps the machinery reachable.
    n_ctx_ = 1;
    if (const char* e = std::getenv("LGA_ROUNDS_IN_FLIGHT")) n_ctx_ = prof.on ? 1 : std::max(1, std::min(kMaxRoundsInFlight, std::atoi(e)));
    if (const char* e = std::getenv("LGA_SOLO_ROWS"))
      solo_rows_ = std::max(k::kSubRows, std::min(k::kSoloMaxRows, std::atoi(e)));
    single_pass_partition_ = !(std::getenv("LGA_TWO_PASS_PARTITION") && std::atoi(std::getenv("LGA_TWO_PASS_PARTITION")) != 0);
    if (const char* e = std::getenv("LGA_PART_ROWS")) part_rows_forced_ = std::atoi(e);
    if (part_rows_forced_ != 0 && part_rows_forced_ != 2048 && part_rows_forced_ != 4096 && part_rows_forced_ != 8192)
      throw std::invalid_argument("LGA_PART_ROWS: 2048, 4096 or 8192");
    if (const char* e = std::getenv("LGA_PART_ROWS4K_ABOVE")) part_rows4k_above_ = std::atoll(e);
    if (const char* e = std::getenv("LGA_PART_ROWS8K_ABOVE")) part_rows8k_above_ = std::atoll(e);
    scan_pairs_ = F_ > 4 * kTileFeatures;
    if (const char* e = std::getenv("LGA_SCAN_PAIRS")) scan_pairs_ = std::atoi(e) != 0;
    tree_kernel_rows_ = std::min<int64_t>(k::kTreeMaxRows, cfg.whole_tree_max_rows);
    if (const char* e = std::getenv("LGA_TREE_KERNEL_ROWS"))
      tree_kernel_rows_ = std::max<int64_t>(0, std::min<int64_t>(k::kTreeMaxRows, std::atoll(e)));
    // full-precision histograms (use_quantized_grad=false): the growth rounds carry them; the one-workgroup builds
    // keep packed 16-bit records and stay out of it
    wide_ = !cfg.use_quantized_grad;
    // Units of a round's large leaves: about one workgroup slot's worth each (16-bit: two 512-thread workgroups per CU,
    // 512 slots; full precision: one 1024-thread workgroup per CU, 256 slots) - fewer, longer units beat more, shorter
    // ones (every unit clears and flushes a whole LDS histogram); measured over 256..2048 units on the 10.5 M x 28 shape.
    target_units_ = wide_ ? 256 : 512;
    {
      int dev = 0, cus = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
        scan_wg_target_ = F_ > kTileFeatures ? 2 * cus : cus;
      if (const char* e = std::getenv("LGA_SCAN_WGS")) scan_wg_target_ = std::max(1, std::atoi(e));
    }
    if (const char* e = std::getenv("LGA_TARGET_UNITS")) target_units_ = std::max(1, std::atoi(e));
    max_unit_rows_ = wide_ ? 65536 : kMaxUnitRows;
    if (const char* e = std::getenv("LGA_MAX_UNIT_ROWS")) max_unit_rows_ = std::max(k::kSubRows, std::min(max_unit_rows_, std::atoi(e)));
    // (the one-workgroup builds round to nearest: stochastic rounding takes the growth rounds as well)
    tree_kernel_on_ = tree_kernel_rows_ > 0 && !wide_ && !cfg.stochastic_rounding;
    // an explicit whole_tree_max_rows asks for the one-workgroup builds also for a booster that trains alone
    tree_kernel_forced_ = tree_kernel_on_ && std::find(cfg.explicit_keys.begin(), cfg.explicit_keys.end(), "whole_tree_max_rows") != cfg.explicit_keys.end();
    tree_renew_ = cfg.quant_train_renew_leaf ? 1 : 0;
    const size_t n_scan_max = (size_t)(2 * max_batch_ + 2);
    for (int c = 0; c < n_ctx_; ++c) {
      RoundCtx& x = ctx_[c];
      if (c == 0) x.stream = stream_;
      else { LGA_HIP_CHECK(hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking)); x.owns_stream = true; }
      if (n_ctx_ > 1 && mode_ != ParallelMode::Serial) {
        LGA_HIP_CHECK(hipEventCreateWithFlags(&x.ev_ready, hipEventDisableTiming));
        LGA_HIP_CHECK(hipEventCreateWithFlags(&x.ev_reduced, hipEventDisableTiming));
      }
      x.mask.resize((size_t)max_pchunks_ * (k::kPartChunk / 64));
      x.part_cursor.resize(((size_t)max_batch_ + 1) * k::kPartCursorStride);
      x.part_cursor.zero(stream_);
      x.chunk_left.resize((size_t)max_pchunks_);
      x.task_left.resize((size_t)max_batch_ + 1);
      x.stage_host.resize(stage_bytes_);
      x.stage_dev.resize(stage_bytes_);
      // host-mapped result block: [flag | SplitInfo x 2B | task_left x B | QuantScale]
      x.blocks_off = (k::kHostDataOffset + sizeof(SplitInfo) * n_scan_max + sizeof(int32_t) * (size_t)(max_batch_ + 1) + 64 + 63) & ~(size_t)63;
      x.host_block.resize(x.blocks_off + sizeof(k::BlockResult) * (size_t)k::kDirectBlocksMax);
      std::memset(x.host_block.get(), 0, x.host_block.size());
      // counters the kernels leave at zero: [round completion | per-scan-task block tickets]
      x.counters.resize(4 + n_scan_max);
      x.counters.zero(stream_);
      x.block_best.resize(n_scan_max * (size_t)std::max(scan_range_.n_blocks, (F_ + 7) / 8));
      x.scan_totals.resize(n_scan_max * 4);
      x.results.resize(n_scan_max);
      if (mode_ != ParallelMode::Serial) {
        x.best_local.resize(n_scan_max);
        x.best_all.resize(n_scan_max * (size_t)comm_->size());
      }
    }
    // Distributed learners with two rounds in flight: every collective goes through ONE stream of its own,
    // in submission order (the same on every rank), tied to the rounds' streams by events - a communicator
    // never sees calls from two streams.
    if (n_ctx_ > 1 && mode_ != ParallelMode::Serial)
      LGA_HIP_CHECK(hipStreamCreateWithFlags(&comm_stream_, hipStreamNonBlocking));
    LGA_HIP_CHECK(hipStreamSynchronize(stream_));
    meta_dev_.resize((size_t)std::max(F_, 1));
    qs_dev_.resize(1);
    prof.init();
  }
  ~HipBackend() override {
    report_grow_timing();
    for (int c = 0; c < n_ctx_; ++c)
      if (ctx_[c].owns_stream) { (void)hipStreamSynchronize(ctx_[c].stream); (void)hipStreamDestroy(ctx_[c].stream); }
    for (int c = 0; c < n_ctx_; ++c) {
      if (ctx_[c].ev_ready) (void)hipEventDestroy(ctx_[c].ev_ready);
      if (ctx_[c].ev_reduced) (void)hipEventDestroy(ctx_[c].ev_reduced);
    }
    if (comm_stream_) { (void)hipStreamSynchronize(comm_stream_); (void)hipStreamDestroy(comm_stream_); }
  }

  void set_bag(const int32_t* rows_dev, int64_t n) { bag_ = rows_dev; bag_n_ = n; }
  // Row lines (kernels.hpp: build_row_lines) carry the gradients themselves and take the fused growth rounds.
  void set_lines(uint8_t* lines) { lines_ = lines; packed_ = nullptr; hist_src_ = nullptr; }
  // Feature-tile records + a separate array of packed gradients.
  void set_packed(const int64_t* packed_dev, const uint8_t* rows_src) {
    packed_ = packed_dev; hist_src_ = rows_src; lines_ = nullptr;
    // several feature tiles: the partition moves the packed gradients along with the row ids (k::GradArenas)
    if (ordered_grads_ && T_ >= 2 && gh_arena_[0].size() == 0)
      for (int b = 0; b < 2; ++b) gh_arena_[b].resize(arena_[b].size());
  }
  // Row lines only: the root pass quantises these float gradients itself (no separate quantisation kernel).
  void set_root_quant(const float* g, const float* h, const uint32_t* maxbits, const RoundingSpec& rounding) {
    root_quant_ = k::FusedQuant{g, h, maxbits, qs_dev_.get(), rounding};
  }
  QuantScale* qs_device() { return qs_dev_.get(); }

  int node_capacity() const override { return 2 * capacity_splits_ + 1; }
  QuantScale quant_scale() const override { return qs_host_; }
  int64_t node_local_count(int node) const override { return ranges_host_[(size_t)node].count; }
  const NodeRange& range(int node) const { return ranges_host_[(size_t)node]; }
  const int32_t* arena(int b) const { return arena_[b].get(); }
  ParallelMode mode() const { return mode_; }
  PhaseProfile prof;

  void set_node_bounds(int node, double lo, double hi, double out) override {
    double* b = reinterpret_cast<double*>(bounds_host_.get()) + (size_t)kNodeInfoStride * (size_t)node;
    b[0] = lo; b[1] = hi; b[2] = out;
  }
  void node_cat_set(int node, uint64_t* words) const override {
    for (int w = 0; w < kCatSetWords; ++w) words[w] = cat_sets_host_.get()[(size_t)node * kCatSetWords + (size_t)w];
  }
  // Interaction constraints: masks live in a device array like the ranges (SplitParams::node_masks).
  void set_node_mask(int node, const uint64_t* words) override {
    const size_t W = (size_t)((F_ + 63) / 64);
    if (!masked_) {
      masked_ = true;
      const size_t n_nodes = (size_t)(2 * capacity_splits_ + 2);
      masks_host_.resize(n_nodes * W);
      masks_dev_.resize(n_nodes * W);
      sp_.node_masks = masks_dev_.get();
      sp_.mask_words = (int32_t)W;
    }
    for (size_t w = 0; w < W; ++w) masks_host_.get()[(size_t)node * W + w] = words[w];
    if (node == 0)
      LGA_HIP_CHECK(hipMemcpyAsync(masks_dev_.get(), masks_host_.get(), W * sizeof(uint64_t), hipMemcpyHostToDevice, stream_));
  }

  void begin_tree(const std::vector<FeatureMeta>& meta, const SplitParams& params) override {
    sp_ = params;
    // Monotone constraints: node output ranges live in a device array the scan reads (SplitParams::
    // node_bounds); the grower announces them per node, every round uploads the entries of its children.
    // per-feature gain adjustment: the grower's host tables go to the device; the host keeps its pointers
    // for the category-set pass
    gain_scale_host_ = params.feature_scale;
    gain_offset_host_ = params.feature_offset;
    if (params.feature_scale) {
      gain_dev_.resize((size_t)2 * (size_t)std::max(F_, 1));
      gain_pinned_.resize((size_t)2 * (size_t)std::max(F_, 1));
      std::memcpy(gain_pinned_.get(), params.feature_scale, sizeof(double) * (size_t)F_);
      std::memcpy(gain_pinned_.get() + F_, params.feature_offset, sizeof(double) * (size_t)F_);
      LGA_HIP_CHECK(hipMemcpyAsync(gain_dev_.get(), gain_pinned_.get(), sizeof(double) * 2 * (size_t)F_, hipMemcpyHostToDevice, stream_));
      sp_.feature_scale = gain_dev_.get();
      sp_.feature_offset = gain_dev_.get() + F_;
    }
    masked_ = false;
    sorted_cat_.clear();
    for (int f = 0; f < (int)meta.size(); ++f)
      if (meta[(size_t)f].enabled && meta[(size_t)f].categorical == kCategoricalSorted) sorted_cat_.push_back(f);
    if (mode_ == ParallelMode::Voting && !sorted_cat_.empty())
      throw std::runtime_error("tree_learner=voting on the device does not search category sets (categorical features with more than "
                               "max_cat_to_onehot categories): use tree_learner=data, or the CPU engine");
    if (!sorted_cat_.empty() && cat_sets_dev_.size() == 0) {
      const size_t n_nodes = (size_t)(2 * capacity_splits_ + 2);
      cat_sets_host_.resize(n_nodes * kCatSetWords);
      cat_sets_dev_.resize(n_nodes * kCatSetWords);
    }
    ranged_ = params.path_smooth > 0.0;
    for (const FeatureMeta& m : meta) ranged_ = ranged_ || (m.enabled && m.monotone != 0);
    if (ranged_) {
      const size_t n_nodes = (size_t)(2 * capacity_splits_ + 2);
      bounds_host_.resize(n_nodes * kNodeInfoStride);
      bounds_dev_.resize(n_nodes * kNodeInfoStride);
      set_node_bounds(0, -std::numeric_limits<double>::infinity(), std::numeric_limits<double>::infinity(),
                      std::numeric_limits<double>::quiet_NaN());
      LGA_HIP_CHECK(hipMemcpyAsync(bounds_dev_.get(), bounds_host_.get(), kNodeInfoStride * sizeof(double),
                                   hipMemcpyHostToDevice, stream_));
      sp_.node_bounds = bounds_dev_.get();
    }
    if (meta.size() != meta_host_.size() ||
        std::memcmp(meta.data(), meta_host_.data(), sizeof(FeatureMeta) * meta.size()) != 0) {
      meta_host_ = meta;
      LGA_HIP_CHECK(hipMemcpyAsync(meta_dev_.get(), meta_host_.data(), sizeof(FeatureMeta) * meta_host_.size(),
                                   hipMemcpyHostToDevice, stream_));
      // meta_host_ stays alive and unchanged until the copy has run (next sync is in root_round)
    }
    ranges_host_[0] = NodeRange{0, (int32_t)bag_n_, 0, 0};
    tree_wide_valid_ = false;
    ++prof.trees;
  }

  // The slots a tree has used are cleared as soon as its last round is in (while the host assembles the tree the
  // device has nothing else to do); the next root round then only finds slot 0 to clear.
  void tree_rounds_done() override {
    if (slots_dirty_ <= 1) return;
    LGA_HIP_CHECK(hipMemsetAsync(pool_.get() + slot_words_, 0, sizeof(int64_t) * (size_t)slot_words_ * (size_t)(slots_dirty_ - 1), stream_));
    slots_dirty_ = 1;
  }

  void root_round(SplitInfo* out) override {
    std::vector<SplitTask> tasks(1);
    SplitTask& t = tasks[0];
    t = SplitTask{};
    t.parent_node = -1; t.left_node = 0; t.right_node = 0; t.feature = 0; t.hist_child = 0;
    t.parent_slot = 0; t.new_slot = 0;
    // the root's row range goes to the device through the staging block's first bytes
    RoundCtx& x = ctx_[0];
    std::memcpy(x.stage_host.get(), &ranges_host_[0], sizeof(NodeRange));
    LGA_HIP_CHECK(hipMemcpyAsync(nodes_.get(), x.stage_host.get(), sizeof(NodeRange), hipMemcpyHostToDevice, stream_));
    // Large leaves ADD their histograms to their slots, so slots start a tree at zero: one clear per tree of
    // what the previous tree used, instead of one per round.
    LGA_HIP_CHECK(hipMemsetAsync(pool_.get(), 0, sizeof(int64_t) * (size_t)slot_words_ * (size_t)slots_dirty_, stream_));
    slots_dirty_ = 1;
    launch_round(x, tasks, true);
    finish_round(x, out);
  }

  void split_round(std::vector<SplitTask>& tasks, SplitInfo* infos) override {
    if (tasks.empty()) return;
    launch_round(ctx_[0], tasks, false);
    finish_round(ctx_[0], infos);
  }

  int max_in_flight() const override { return n_ctx_; }
  int submit_round(std::vector<SplitTask>& tasks) override {
    int c = 0;
    while (c < n_ctx_ && ctx_[c].busy) ++c;
    if (c == n_ctx_) throw std::runtime_error("more rounds submitted than the backend keeps in flight");
    launch_round(ctx_[c], tasks, false);
    return c;
  }
  void wait_round(int ticket, SplitInfo* infos) override { finish_round(ctx_[ticket], infos); }
  bool round_done(int ticket) override {
    if (distributed() || prof.on) return true;   // every rank replays the rounds in submission order
    const RoundCtx& x = ctx_[ticket];
    if (x.direct && !x.root) return direct_done(x, 2 * (int)x.tasks.size());
    return *reinterpret_cast<const volatile unsigned*>(x.host_block.get()) == x.seq;
  }

  // ---- whole tree in one launch (kernels.hpp: TreeArgs): small bags on one device, plain numeric / one-vs-rest
  // categorical features of one tile ----
  // What does not depend on the tree: the learner, the data layout, the row and leaf limits.
  void set_in_company(bool yes) { in_company_ = yes; }
  bool tree_kernel_possible(int64_t rows, int max_splits) const {
    if (!tree_kernel_on_ || !(in_company_ || tree_kernel_forced_) || mode_ != ParallelMode::Serial || T_ != 1 || F_ < 1 || prof.on) return false;
    return rows >= 1 && rows <= tree_kernel_rows_ && max_splits >= 1 && max_splits <= k::kTreeMaxSplits;
  }
  // Descriptor of a build over the current bag; `out_dev` is the device address of its host-mapped result block.
  void fill_tree_args(k::TreeArgs& a, int max_splits, char* out_dev) {
    const int bw = F_ <= 8 ? 2 : (F_ <= 16 ? 4 : 8);
    const size_t cap = arena_[0].size();
    if (rec_gh_[0].size() < cap) {
      for (int b = 0; b < 2; ++b) { rec_gh_[b].resize(cap); rec_bins_[b].resize(cap * (size_t)bw); }
      tree_best_.resize(k::kTreeMaxNodes);
    }
    a = k::TreeArgs{};
    a.bag = bag_; a.n = (int32_t)bag_n_; a.F = F_; a.budget = max_splits; a.renew = tree_renew_;
    a.bins_rt = mir_->bins_rt.get();
    a.g = root_quant_.g; a.h = root_quant_.h; a.maxbits = root_quant_.maxbits; a.qs_out = qs_dev_.get();
    for (int b = 0; b < 2; ++b) { a.idx[b] = arena_[b].get(); a.gh[b] = rec_gh_[b].get(); a.bins[b] = rec_bins_[b].get(); }
    a.meta = meta_dev_.get();
    a.sp = sp_;
    a.pool = pool_.get(); a.slot_words = slot_words_;
    a.nodes = nodes_.get();
    a.best = tree_best_.get();
    a.out = out_dev;
    a.lay = k::tree_layout(max_splits);
    if (++tree_seq_ == 0) ++tree_seq_;
    a.seq = tree_seq_;
    a.timing = tree_timing_ ? 1 : 0;
    slots_dirty_ = std::max(slots_dirty_, max_splits + 1);
  }
  bool whole_tree_begin(int max_splits) override {
    tree_prepared_ = false;
    tree_wide_valid_ = false;
    if (!tree_kernel_possible(bag_n_, max_splits) || !root_quant_.g) return false;
    if (ranged_ || masked_ || !sorted_cat_.empty()) return false;
    tree_lay_ = k::tree_layout(max_splits);
    if (tree_out_.size() < (size_t)tree_lay_.bytes) {
      tree_out_.resize((size_t)k::tree_layout(k::kTreeMaxSplits).bytes);
      tree_desc_.resize(sizeof(k::TreeArgs));
      std::memset(tree_out_.get(), 0, tree_out_.size());
    }
    fill_tree_args(*reinterpret_cast<k::TreeArgs*>(tree_desc_.get()), max_splits, tree_out_.device_ptr());
    tree_prepared_ = true;
    tree_launched_ = false;
    return true;
  }
  // The prepared descriptor, for a launch that grows the trees of several backends at once; the caller launches
  // on this backend's stream and then calls whole_tree_launched().
  const k::TreeArgs* whole_tree_descriptor() const {
    return tree_prepared_ ? reinterpret_cast<const k::TreeArgs*>(tree_desc_.get()) : nullptr;
  }
  void whole_tree_launched() { tree_launched_ = true; }
  void whole_tree_launch() override {
    if (!tree_prepared_ || tree_launched_) return;
    TraceRange range("lga.whole_tree");
    k::grow_trees(reinterpret_cast<const k::TreeArgs*>(tree_desc_.device_ptr()), 1, F_, stream_);
    tree_launched_ = true;
  }
  // Reads one result block (host address) into `out`; returns the header.
  static const k::TreeHeader& read_tree_block(const char* block, const k::TreeLayout& lay, WholeTree* out) {
    const k::TreeHeader& hd = *reinterpret_cast<const k::TreeHeader*>(block);
    const int n_splits = hd.n_splits;
    if (n_splits < 0 || n_splits > k::kTreeMaxSplits) throw std::runtime_error("whole-tree build returned a bad split count");
    const int32_t* sn = reinterpret_cast<const int32_t*>(block + lay.off_split);
    const SplitInfo* best = reinterpret_cast<const SplitInfo*>(block + lay.off_best);
    out->split_node.assign(sn, sn + n_splits);
    out->best.assign(best, best + (2 * n_splits + 1));
    return hd;
  }
  void whole_tree_finish(WholeTree* out) override {
    if (!tree_prepared_ || !tree_launched_) throw std::runtime_error("whole-tree build was never launched");
    {
      TraceRange wait_range("lga.wait_tree");
      wait_for(reinterpret_cast<volatile unsigned*>(tree_out_.get()), tree_seq_, stream_);
    }
    tree_prepared_ = false;
    const k::TreeHeader& r = read_tree_block(tree_out_.get(), tree_lay_, out);
    const int n_splits = r.n_splits, n_nodes = 2 * n_splits + 1;
    qs_host_ = QuantScale{r.kg, r.kh};
    const NodeRange* ranges = reinterpret_cast<const NodeRange*>(tree_out_.get() + tree_lay_.off_range);
    for (int i = 0; i < n_nodes; ++i) ranges_host_[(size_t)i] = ranges[i];
    tree_wide_valid_ = tree_renew_ != 0;
    ++prof.rounds;
    ++whole_trees_;
    if (tree_timing_) note_tree_timing(r, 1);
  }
  // LGA_TREE_TIMING=1: the kernel's own phase profile, averaged over 100 trees
  void note_tree_timing(const k::TreeHeader& r, int trees) {
    static const char* names[8] = {"root pass", "selection", "sides + histogram", "prefix", "search", "row movement", "bookkeeping", "hand-over"};
    for (int i = 0; i < 8; ++i) tree_ticks_[i] += (double)r.phase_ticks[i];
    tree_ticks_[8] += r.n_splits;
    tree_cycles_ += (double)r.shader_cycles;
    tree_timed_ += trees;
    if (tree_timed_ % 100 == 0) {
      std::fprintf(stderr, "[tree kernel] per tree (%.0f splits):", tree_ticks_[8] / 100);
      double total = 0;
      for (int i = 0; i < 8; ++i) { std::fprintf(stderr, " %s %.1f us,", names[i], tree_ticks_[i] / 100 * 0.01); total += tree_ticks_[i]; }
      std::fprintf(stderr, " total %.1f us", total / 100 * 0.01);
      if (tree_cycles_ > 0) std::fprintf(stderr, " at %.0f MHz shader clock", tree_cycles_ / (total * 0.01));
      std::fprintf(stderr, "\n");
      for (double& v : tree_ticks_) v = 0;
      tree_cycles_ = 0;
    }
  }
  // 36-bit gradient sums of a final leaf's rows as the last whole-tree build left them (false: not available).
  bool whole_tree_leaf_sums(int node, int64_t* g, int64_t* h) const {
    if (!tree_wide_valid_) return false;
    const int64_t* wide = reinterpret_cast<const int64_t*>(tree_out_.get() + tree_lay_.off_wide);
    *g = wide[2 * node]; *h = wide[2 * node + 1];
    return true;
  }
  unsigned tree_seq() const { return tree_seq_; }
  int64_t whole_trees_grown() const { return whole_trees_; }
  bool tree_timing() const { return tree_timing_; }
  bool renews_leaves() const { return tree_renew_ != 0; }
  bool wide() const { return wide_; }
  // bits between the histograms' scale (quant_scale) and the 36-bit scale of the leaf-value sums
  int renew_extra_bits() const { return wide_ ? 0 : kWideExtraBits; }
  int num_features() const { return F_; }
  void set_quant_scale(QuantScale q) { qs_host_ = q; }
  // For descriptors filled outside begin_tree (boosting chunks): the tree's parameters and feature table.
  void set_tree_params(const std::vector<FeatureMeta>& meta, const SplitParams& params) {
    sp_ = params;
    if (meta.size() != meta_host_.size() || std::memcmp(meta.data(), meta_host_.data(), sizeof(FeatureMeta) * meta.size()) != 0) {
      meta_host_ = meta;
      LGA_HIP_CHECK(hipMemcpyAsync(meta_dev_.get(), meta_host_.data(), sizeof(FeatureMeta) * meta_host_.size(), hipMemcpyHostToDevice, stream_));
    }
    ranged_ = false; masked_ = false; sorted_cat_.clear();
    tree_wide_valid_ = false;
  }
  static void wait_chunk(volatile unsigned* flag, unsigned seq, hipStream_t stream) { wait_for(flag, seq, stream); }

  // Row partition used for bagging: base rows -> [bag | out-of-bag] in `out`; returns the bag size.
  int64_t bag_partition(const int32_t* base_rows, int64_t n, int32_t* out, double fraction, uint64_t seed,
                        uint64_t iter, uint64_t salt, const int32_t* ext_ids = nullptr) {
    k::BagParams bp{};
    bp.seed = seed; bp.iter = iter; bp.salt = salt; bp.fraction = (float)fraction;
    bp.ext_ids = ext_ids;
    return sample_partition(base_rows, n, out, -1, bp);
  }

  // feature: -1 bagging, -2 gradient-based one-side sampling, -3 bagging by label (kernels.hpp: BagParams)
  int64_t sample_partition(const int32_t* base_rows, int64_t n, int32_t* out, int feature, const k::BagParams& bp) {
    if (n == 0) return 0;
    SplitTask t{};
    // Scratch node entries beyond the grower's range.  The children must not alias the parent: the first
    // chunk's workgroup writes the child ranges while the other chunks still read the parent's.
    const int node_id = 2 * capacity_splits_ + 1;
    t.parent_node = node_id; t.left_node = node_id + 1; t.right_node = node_id + 2;
    t.feature = feature; t.first_pchunk = 0; t.num_pchunk = ceil_div(n, k::kPartChunk);
    if (t.num_pchunk > max_pchunks_) throw std::runtime_error("bagging chunk table overflow");
    RoundCtx& x = ctx_[0];
    char* st = x.stage_host.get();
    std::memcpy(st + off_tasks_, &t, sizeof(t));
    int32_t* pc = reinterpret_cast<int32_t*>(st + off_pch_);
    for (int c = 0; c < t.num_pchunk; ++c) pc[c] = 0;
    NodeRange r{0, (int32_t)n, 0, 0};
    std::memcpy(st, &r, sizeof(NodeRange));
    upload_stage(x, off_pch_ + sizeof(int32_t) * (size_t)t.num_pchunk);
    LGA_HIP_CHECK(hipMemcpyAsync(nodes_.get() + node_id, x.stage_dev.get(), sizeof(NodeRange), hipMemcpyDeviceToDevice, stream_));
    const SplitTask* dt = reinterpret_cast<const SplitTask*>(x.stage_dev.get() + off_tasks_);
    const int32_t* dpc = reinterpret_cast<const int32_t*>(x.stage_dev.get() + off_pch_);
    k::InlineTasks none{};
    k::partition(dt, dpc, none, t.num_pchunk, t.num_pchunk > 1, nodes_.get(), base_rows, nullptr, nullptr, out, nullptr,
                 N_, x.mask.get(), x.chunk_left.get(), x.task_left.get(), bp, nullptr, 0, stream_);
    int32_t nl = 0;
    LGA_HIP_CHECK(hipMemcpyAsync(&nl, x.task_left.get(), sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    sync();
    return nl;
  }

  void sync() { LGA_HIP_CHECK(hipStreamSynchronize(stream_)); }
  hipStream_t stream() const { return stream_; }

 private:
  int pick_hist_chunk(int64_t total_rows, int tiles) const {
    // aim at ~1024 workgroups (row chunks x feature tiles) when there is enough work; never exceed the LDS
    // accumulator range.  Every workgroup clears and flushes a 64 KiB histogram, so wide data wants few, long
    // chunks: 400 k rows x 63 tiles as 1024-row chunks are 24 k workgroups and 400 M global atomics per pass.
    const int64_t chunks = std::max<int64_t>(1, 1024 / std::max(tiles, 1));
    int64_t c = (total_rows + chunks - 1) / chunks;
    c = ((c + 511) / 512) * 512;
    if (c < 1024) c = 1024;
    if (c > max_unit_rows_) c = max_unit_rows_;
    return (int)c;
  }

  // Everything one round in flight owns: its stream, scratch for the partition, the staging block for
  // large task lists, and the host-mapped block its results and completion flag arrive in.
  struct RoundCtx {
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    hipEvent_t ev_ready = nullptr, ev_reduced = nullptr;   // hand-over to / from the collectives' stream
    DeviceBuffer<uint64_t> mask;
    // voting-parallel rounds (kernels.hpp: Vote)
    DeviceBuffer<double> vote_gains;
    DeviceBuffer<k::Vote> votes_local, votes_all;
    DeviceBuffer<int32_t> vote_chosen;
    DeviceBuffer<int64_t> vote_buf;
    DeviceBuffer<unsigned long long> part_cursor;  // single-pass partition: the tasks' reservation cursors (zero between rounds)
    DeviceBuffer<int32_t> chunk_left, task_left;
    DeviceBuffer<char> stage_dev;
    PinnedBuffer<char> stage_host;
    MappedBuffer host_block;
    DeviceBuffer<unsigned> counters;
    DeviceBuffer<k::Cand> block_best, best_local, best_all;
    DeviceBuffer<int64_t> scan_totals;
    DeviceBuffer<SplitInfo> results;
    unsigned seq = 0;
    // the round in flight
    bool busy = false, root = false;
    int direct = 0;        // the round's searches hand their results over one by one (ScanArgs::direct)
    int direct_blocks = 1; // ... direct == 2: feature blocks per searched child
    size_t blocks_off = 0; // ... and where their records start in the host block
    std::vector<SplitTask> tasks;
    std::chrono::steady_clock::time_point t_begin, t_launched;
    unsigned* done_counter() const { return counters.get(); }
    unsigned* scan_tickets() const { return counters.get() + 4; }
  };

  void upload_stage(RoundCtx& x, size_t bytes) {
    LGA_HIP_CHECK(hipMemcpyAsync(x.stage_dev.get(), x.stage_host.get(), bytes, hipMemcpyHostToDevice, x.stream));
  }

  // Node tables the split search reads (monotone ranges / path smoothing, interaction masks): the entries of
  // this round's children go up before the round's first kernel.
  void upload_node_tables(const std::vector<SplitTask>& tasks, hipStream_t st_) {
    if (!ranged_ && !masked_) return;
    int first = tasks[0].left_node, last = tasks[0].right_node;   // the children of a round's tasks have consecutive node ids
    for (const auto& t : tasks) { first = std::min(first, t.left_node); last = std::max(last, t.right_node); }
    if (ranged_)
      LGA_HIP_CHECK(hipMemcpyAsync(bounds_dev_.get() + kNodeInfoStride * (size_t)first,
                                   bounds_host_.get() + kNodeInfoStride * (size_t)first,
                                   kNodeInfoStride * sizeof(double) * (size_t)(last - first + 1), hipMemcpyHostToDevice, st_));
    if (masked_) {
      const size_t W = (size_t)sp_.mask_words;
      LGA_HIP_CHECK(hipMemcpyAsync(masks_dev_.get() + W * (size_t)first, masks_host_.get() + W * (size_t)first,
                                   W * sizeof(uint64_t) * (size_t)(last - first + 1), hipMemcpyHostToDevice, st_));
    }
  }
  void upload_cat_sets(const std::vector<SplitTask>& tasks, hipStream_t st_) {
    // splits on category sets: the sets of this round's parents go up before the partition reads them
    if (sorted_cat_.empty()) return;
    for (const auto& t : tasks)
      if (t.default_left == kCategoricalSetTask)
        LGA_HIP_CHECK(hipMemcpyAsync(cat_sets_dev_.get() + (size_t)t.parent_node * kCatSetWords,
                                     cat_sets_host_.get() + (size_t)t.parent_node * kCatSetWords,
                                     sizeof(uint64_t) * kCatSetWords, hipMemcpyHostToDevice, st_));
  }

  // Enqueues one growth round on the context's stream; results come back through host-mapped memory and a
  // completion flag.  root: no partition, a single scan task.
  void launch_round(RoundCtx& x, std::vector<SplitTask>& tasks, bool root) {
    if (x.busy) throw std::runtime_error("round context is still in flight");
    const int B = (int)tasks.size();
    if (B > max_batch_) throw std::runtime_error("split batch exceeds backend capacity");
    x.t_begin = std::chrono::steady_clock::now();
    if (!root) prof.host_between_ms += std::chrono::duration<double, std::milli>(x.t_begin - prof.last_end).count();
    if (++x.seq == 0) ++x.seq;   // never 0
    prof.mark(PhaseProfile::kUpload, x.stream);
    trace_push(root ? "lga.root_round" : "lga.split_round");
    if (!root) {
      upload_cat_sets(tasks, x.stream);
      upload_node_tables(tasks, x.stream);
    }
    const k::RoundSlots round{tasks[0].new_slot, root ? 1 : B};
    // a round's new slots are consecutive (the grower hands them out in task order)
    for (int i = 0; i < B; ++i)
      if (tasks[(size_t)i].new_slot != round.first_slot + i) throw std::runtime_error("round slots are not consecutive");
    slots_dirty_ = std::max(slots_dirty_, round.first_slot + round.n_slots);
    if (lines_) launch_lines(x, tasks, root, round);
    else launch_tiles(x, tasks, root, round);
    trace_pop();
    x.busy = true;
    x.root = root;
    x.tasks = tasks;
    x.t_launched = std::chrono::steady_clock::now();
    prof.host_launch_ms += std::chrono::duration<double, std::milli>(x.t_launched - x.t_begin).count();
  }

  bool distributed() const { return mode_ != ParallelMode::Serial; }

  // ---- fused rounds over row lines (one feature tile; serial and data-parallel) ----
  void launch_lines(RoundCtx& x, std::vector<SplitTask>& tasks, bool root, k::RoundSlots round) {
    const int B = (int)tasks.size();
    hipStream_t st_ = x.stream;
    // units: a leaf of up to solo_rows_ rows is one workgroup's from the sides of its rows to its child's histogram;
    // a larger leaf is partitioned by two launches of their own and the rows of its histogram child (at most
    // hist_rows of them on any rank) are cut into units of unit_rows consecutive rows, as are the root's
    auto hist_bound = [&](const SplitTask& t) {
      const int64_t cnt = root ? bag_n_ : ranges_host_[(size_t)t.parent_node].count;
      return root ? cnt : std::min<int64_t>(cnt, std::max(t.hist_rows, 0));
    };
    // (partition_scatter_kernel leaves a leaf of ONE block to the kernel that made its masks - the tile path's does
    // finish such leaves, grow_flags_kernel does not: a multi-unit leaf here always spans at least two blocks)
    static_assert(k::kSoloMaxRows >= k::kPartChunk, "leaves of up to one partition block must be solo leaves");
    const int64_t solo_limit = std::max<int64_t>(solo_rows_, k::kPartChunk);
    int64_t multi_rows = 0;
    for (const auto& t : tasks) {
      const int64_t cnt = root ? bag_n_ : ranges_host_[(size_t)t.parent_node].count;
      if (root || cnt > solo_limit) multi_rows += hist_bound(t);
    }
    int64_t ur = (multi_rows + target_units_ - 1) / target_units_;
    ur = ((ur + k::kSubRows - 1) / k::kSubRows) * k::kSubRows;
    ur = std::max<int64_t>(k::kSubRows, std::min<int64_t>(ur, (max_unit_rows_ / k::kSubRows) * k::kSubRows));
    const int unit_rows = (int)ur;
    int np = 0;
    if (single_pass_partition_ && !root) {
      // the blocks of one leaf reserve their room through one cursor: a round with a large leaf takes larger blocks
      // (fewer atomics on that one address), a round of many middling leaves smaller ones (more workgroups per leaf)
      int64_t largest = 0;
      for (const auto& t : tasks) largest = std::max<int64_t>(largest, ranges_host_[(size_t)t.parent_node].count);
      part_chunk_ = part_rows_forced_ ? part_rows_forced_ : largest > part_rows8k_above_ ? 8192 : largest > part_rows4k_above_ ? 4096 : 2048;
    }
    for (auto& t : tasks) {
      const int64_t cnt = root ? bag_n_ : ranges_host_[(size_t)t.parent_node].count;
      const bool multi = root || cnt > solo_limit;
      t.first_pchunk = np;
      t.num_pchunk = (multi && !root) ? ceil_div(cnt, part_chunk_) : 0;
      t.num_hchunk = multi ? std::max(1, ceil_div(hist_bound(t), unit_rows)) : 1;
      np += t.num_pchunk;
      if (np > max_pchunks_) throw std::runtime_error("chunk table overflow");
    }
    const int32_t* src0 = (root || tasks[0].parent_node == 0) ? bag_ : arena_[0].get();
    int units = 0;
    for (auto& t : tasks) { t.first_hchunk = units; units += t.num_hchunk; }
    // descriptors: inline in the kernel arguments, or - rounds above kInlineTasks leaves - staged in device memory
    k::InlineTasks inl{};
    const SplitTask* staged = nullptr;
    if (B <= k::kInlineTasks) {
      inl.count = B;
      std::memcpy(inl.t, tasks.data(), sizeof(SplitTask) * (size_t)B);
    } else {
      std::memcpy(x.stage_host.get() + off_tasks_, tasks.data(), sizeof(SplitTask) * (size_t)B);
      LGA_HIP_CHECK(hipMemcpyAsync(x.stage_dev.get() + off_tasks_, x.stage_host.get() + off_tasks_, sizeof(SplitTask) * (size_t)B,
                                   hipMemcpyHostToDevice, st_));
      staged = reinterpret_cast<const SplitTask*>(x.stage_dev.get() + off_tasks_);
    }
    prof.mark(PhaseProfile::kPartition, st_);
    k::GrowArgs a{};
    a.tasks = staged; a.n_staged = staged ? B : 0;
    a.nodes = nodes_.get();
    a.a0 = src0; a.a1 = arena_[1].get();
    a.a0_dst = arena_[0].get(); a.a1_dst = arena_[1].get();
    a.lines = lines_; a.bins_fm = mir_->bins_fm.get(); a.N = N_;
    a.pool = lay_; a.round = round;
    a.F = F_; a.unit_rows = unit_rows;
    a.mask = x.mask.get(); a.chunk_left = x.chunk_left.get(); a.task_left = x.task_left.get();
    a.cat_sets = cat_sets_dev_.get();
    if (root) a.quant = root_quant_;
    a.wide = wide_ ? 1 : 0;
    a.qs = qs_dev_.get();
    a.phase_ticks = (!root && grow_timing_) ? grow_ticks() : nullptr;
    if (np > 0 && single_pass_partition_) {
      a.part_cursor = x.part_cursor.get();
      a.part_block_rows = part_chunk_;
      k::grow_partition(inl, a, np, st_);
    } else if (np > 0) {
      k::grow_flags(inl, a, np, st_);
      k::grow_scatter(staged, a.n_staged, inl, np, nodes_.get(), src0, arena_[1].get(), arena_[0].get(), arena_[1].get(),
                      x.mask.get(), x.chunk_left.get(), x.task_left.get(), st_);
    }
    k::grow_round(inl, a, units, root, st_);
    prof.mark(PhaseProfile::kHist, st_);
    if (distributed()) { finish_distributed(x, tasks, root, round, staged, inl); return; }
    prof.mark(PhaseProfile::kComm, st_);
    prof.mark(PhaseProfile::kScan, st_);
    k::ScanArgs sa = scan_args(x, root, B, round);
    sa.tasks = staged;
    k::scan_blocks(inl, sa, root ? 1 : 2 * B, st_);
    prof.mark(PhaseProfile::kFetch, st_);
    prof.mark(PhaseProfile::kCount, st_);
  }

  // ---- rounds over feature-tile records (several tiles, or feature-parallel) ----
  void launch_tiles(RoundCtx& x, std::vector<SplitTask>& tasks, bool root, k::RoundSlots round) {
    const int B = (int)tasks.size();
    hipStream_t st_ = x.stream;
    const bool inline_tasks = B <= k::kInlineTasks;
    int64_t total_rows = 0;
    for (const auto& t : tasks) total_rows += root ? bag_n_ : ranges_host_[(size_t)t.parent_node].count;
    int tiles_here = T_;
    if (mode_ == ParallelMode::Feature) {
      const int P = comm_->size(), r = comm_->rank();
      tiles_here = std::max(1, r < T_ ? (T_ - r + P - 1) / P : 0);
    }
    const int hch = pick_hist_chunk(total_rows, tiles_here);
    char* st = x.stage_host.get();
    int32_t* pc = reinterpret_cast<int32_t*>(st + off_pch_);
    int32_t* hc = reinterpret_cast<int32_t*>(st + off_hch_);
    int np = 0, nh = 0;
    bool any_multi = false;
    for (int i = 0; i < B; ++i) {
      SplitTask& t = tasks[(size_t)i];
      int64_t cnt = root ? bag_n_ : ranges_host_[(size_t)t.parent_node].count;
      t.first_pchunk = np;
      t.num_pchunk = root ? 0 : std::max(1, ceil_div(cnt, k::kPartChunk));
      t.first_hchunk = nh;
      t.num_hchunk = std::max(1, ceil_div(cnt, hch));
      if (np + t.num_pchunk > max_pchunks_ || nh + t.num_hchunk > max_hchunks_)
        throw std::runtime_error("chunk table overflow");
      any_multi = any_multi || t.num_pchunk > 1;
      if (!inline_tasks) {
        for (int c = 0; c < t.num_pchunk; ++c) pc[np + c] = i;
        for (int c = 0; c < t.num_hchunk; ++c) hc[nh + c] = i;
      }
      np += t.num_pchunk;
      nh += t.num_hchunk;
    }
    k::InlineTasks inl{};
    const SplitTask* dt = nullptr;
    const int32_t* dpc = nullptr;
    const int32_t* dhc = nullptr;
    if (inline_tasks) {
      inl.count = B;
      std::memcpy(inl.t, tasks.data(), sizeof(SplitTask) * (size_t)B);
    } else {
      std::memcpy(st + off_tasks_, tasks.data(), sizeof(SplitTask) * (size_t)B);
      upload_stage(x, off_hch_ + sizeof(int32_t) * (size_t)nh);
      dt = reinterpret_cast<const SplitTask*>(x.stage_dev.get() + off_tasks_);
      dpc = reinterpret_cast<const int32_t*>(x.stage_dev.get() + off_pch_);
      dhc = reinterpret_cast<const int32_t*>(x.stage_dev.get() + off_hch_);
    }
    // The root's rows are the bag itself: node 0 (buffer 0) is read from there, so a tree starts without
    // copying the row list.  Only the root round and the round that splits the root ever touch node 0.
    const int32_t* src0 = (root || tasks[0].parent_node == 0) ? bag_ : arena_[0].get();
    prof.mark(PhaseProfile::kPartition, st_);
    k::GradArenas grads{};
    if (!root && gh_arena_[0].size() > 0) {
      grads.by_row = packed_;
      grads.arena[0] = gh_arena_[0].get();
      grads.arena[1] = gh_arena_[1].get();
    }
    if (!root) {
      k::BagParams part{};
      part.cat_sets = cat_sets_dev_.get();
      k::partition(dt, dpc, inl, np, any_multi, nodes_.get(), src0, arena_[1].get(), arena_[0].get(),
                   arena_[1].get(), mir_->bins_fm.get(), N_, x.mask.get(), x.chunk_left.get(), x.task_left.get(),
                   part, pool_.get() + (size_t)round.first_slot * slot_words_, (int64_t)slot_words_ * B, st_, grads);
    }
    prof.mark(PhaseProfile::kHist, st_);
    {
      int tile_first = 0, tile_step = 1, tile_count = T_;
      if (mode_ == ParallelMode::Feature) {
        int P = comm_->size(), r = comm_->rank();
        int owned = r < T_ ? (T_ - r + P - 1) / P : 0;
        if (owned > 0) { tile_first = r; tile_step = P; tile_count = owned; }
        else { tile_first = 0; tile_step = 1; tile_count = 1; }   // totals only
      }
      k::histogram(dt, dhc, inl, nh, hch, nodes_.get(), src0, arena_[1].get(), hist_src_, N_, packed_, lay_, round, F_,
                   tile_first, tile_step, tile_count, wide_ ? qs_dev_.get() : nullptr, st_, grads);
    }
    if (distributed()) { finish_distributed(x, tasks, root, round, dt, inl); return; }
    prof.mark(PhaseProfile::kComm, st_);
    prof.mark(PhaseProfile::kScan, st_);
    k::ScanArgs sa = scan_args(x, root, B, round);
    sa.tasks = dt;
    sa.host_out = x.host_block.device_ptr();
    k::scan_blocks(inl, sa, root ? 1 : 2 * B, st_);
    prof.mark(PhaseProfile::kFetch, st_);
    prof.mark(PhaseProfile::kCount, st_);
  }

  k::ScanArgs scan_args(RoundCtx& x, bool root, int B, k::RoundSlots round) {
    k::ScanArgs sa{};
    sa.task0 = 0;
    sa.n_scan = root ? 1 : 2 * B;
    sa.n_tasks = root ? 0 : B;
    sa.root = root ? 1 : 0;
    sa.meta = meta_dev_.get();
    sa.F = F_;
    sa.pool = lay_;
    sa.round = round;
    sa.range = scan_range_;
    if (mode_ != ParallelMode::Feature) {
      // feature blocks: one feature per wave (8 per workgroup) while the round is small - its search is pure
      // latency - growing to keep the launch near a thousand workgroups when it has hundreds of leaves
      const int owned = std::max(0, scan_range_.f_end - scan_range_.f_begin);
      const int want_blocks = std::max(1, scan_wg_target_ / std::max(1, sa.n_scan));
      int g = (owned + want_blocks - 1) / want_blocks;
      g = std::max(8, ((g + 7) / 8) * 8);
      // (ONE block per child for small rounds - no block ticket, no merge of the blocks' winners, a wave takes four
      // features in turn - measured slower: 2.97 -> 3.04-3.17 ms per tree at 10.5 M rows, 1.67 -> 1.72-1.80 at 1.3 M)
      sa.range.group_size = g;
      sa.range.n_blocks = std::max(1, (owned + g - 1) / g);
    }
    // wide data: the search is bound by the histogram bytes it moves - both children of a split in one workgroup
    // read the fresh row once instead of twice (LGA_SCAN_PAIRS=0 / 1 overrides the width rule)
    sa.pairs = (!root && scan_pairs_) ? 1 : 0;
    sa.my_rank = lay_.ranks > 1 ? comm_->rank() : 0;
    sa.node_totals = node_totals_.get();
    sa.scan_totals = x.scan_totals.get();
    sa.qs = qs_dev_.get();
    sa.sp = sp_;
    sa.block_best = x.block_best.get();
    sa.tickets = x.scan_tickets();
    sa.task_left = x.task_left.get();
    sa.results = x.results.get();
    sa.host_out = x.host_block.device_ptr();
    sa.done_counter = x.done_counter();
    sa.seq = x.seq;
    // (data-parallel ranks that have summed whole histograms search like the serial learner, each for itself)
    sa.direct = (!root && direct_results_ && (!distributed() || (mode_ == ParallelMode::Data && dp_allreduce_))) ? 1 : 0;
    if (sa.direct && direct_blocks_ && !sa.pairs && sa.n_scan * sa.range.n_blocks <= k::kDirectBlocksMax) sa.direct = 2;
    sa.host_blocks = x.host_block.device_ptr() + x.blocks_off;
    x.direct = sa.direct;
    x.direct_blocks = sa.range.n_blocks;
    return sa;
  }

  // Distributed tail of a round.  Data-parallel: ONE in-place reduce-scatter over the round's slots leaves
  // this rank with the summed histograms of the features it owns (and the summed totals); it searches those,
  // the ranks' winners (32 bytes per child) are all-gathered and every rank merges them the same way.
  // Feature-parallel: every rank already holds the full rows of its tiles; only the winners travel.
  // With two rounds in flight the collectives (and what sits between them) run on the one stream that
  // carries every collective, in submission order.
  void finish_distributed(RoundCtx& x, const std::vector<SplitTask>& tasks, bool root, k::RoundSlots round,
                          const SplitTask* staged, const k::InlineTasks& inl) {
    const int B = (int)tasks.size();
    hipStream_t st_ = x.stream;
    prof.mark(PhaseProfile::kComm, st_);
    hipStream_t cs = st_;
    if (comm_stream_) {
      LGA_HIP_CHECK(hipEventRecord(x.ev_ready, st_));
      LGA_HIP_CHECK(hipStreamWaitEvent(comm_stream_, x.ev_ready, 0));
      cs = comm_stream_;
    }
    if (mode_ == ParallelMode::Data) {
      int64_t* region = pool_.get() + (size_t)round.first_slot * slot_words_;
      const size_t block = (size_t)round.n_slots * (size_t)lay_.chunk_words;
      // features searched over category sets need their full rows on every rank: sum everything then
      if (sorted_cat_.empty() && !dp_allreduce_) comm_->reduce_scatter_sum_i64(region, block, true, cs);
      else comm_->allreduce_sum_i64(region, block * (size_t)lay_.ranks, true, cs);
    }
    prof.mark(PhaseProfile::kScan, cs);
    const int n_scan = root ? 1 : 2 * B;
    k::ScanArgs sa = scan_args(x, root, B, round);
    if (mode_ == ParallelMode::Voting) {
      // PV-Tree (kernels.hpp: Vote): local search and votes, the votes all-gathered, the most voted features' local
      // rows + the children's local totals all-reduced, the search in the summed rows
      const int P = comm_->size(), kk = top_k_, n_sel = std::min(F_, 2 * kk);
      const size_t per_child = (size_t)k::vote_words_per_child(n_sel);
      const size_t cap = (size_t)(2 * max_batch_ + 2);
      if (x.vote_gains.size() == 0) {
        x.vote_gains.resize(cap * (size_t)std::max(F_, 1));
        x.votes_local.resize(cap * (size_t)kk);
        x.votes_all.resize(cap * (size_t)kk * (size_t)P);
        x.vote_chosen.resize(cap * (size_t)n_sel);
        x.vote_buf.resize(cap * per_child);
      }
      sa.tasks = staged;
      sa.range = k::ScanRange{0, F_, 8, 0, 1, std::max(1, (F_ + 7) / 8)};
      sa.my_rank = 0;
      k::ScanArgs local = sa;   // the ranks' own searches: leaf-size limits scaled down by the number of ranks
      local.sp.min_data_in_leaf = std::max(1, (int)(sp_.min_data_in_leaf / P));
      local.sp.min_sum_hessian_in_leaf = sp_.min_sum_hessian_in_leaf / P;
      k::vote_gains(inl, local, n_scan, x.vote_gains.get(), cs);
      k::vote_topk(x.vote_gains.get(), n_scan, F_, kk, x.votes_local.get(), cs);
      comm_->allgather_bytes(x.votes_local.get(), x.votes_all.get(), sizeof(k::Vote) * (size_t)n_scan * (size_t)kk, true, cs);
      k::vote_select(inl, sa, x.votes_all.get(), P, n_scan, kk, n_sel, x.vote_chosen.get(), x.vote_buf.get(), cs);
      comm_->allreduce_sum_i64(x.vote_buf.get(), (size_t)n_scan * per_child, true, cs);
      k::vote_final(inl, sa, n_scan, x.vote_chosen.get(), x.vote_buf.get(), n_sel, x.best_local.get(), cs);
      k::merge_winners(x.best_local.get(), 1, n_scan, root ? 0 : B, x.scan_totals.get(), qs_dev_.get(), sp_,
                       x.task_left.get(), x.host_block.device_ptr(), x.seq, cs);
      if (comm_stream_) {
        LGA_HIP_CHECK(hipEventRecord(x.ev_reduced, comm_stream_));
        LGA_HIP_CHECK(hipStreamWaitEvent(st_, x.ev_reduced, 0));
      }
      prof.mark(PhaseProfile::kFetch, st_);
      prof.mark(PhaseProfile::kCount, st_);
      return;
    }
    if (mode_ == ParallelMode::Data && dp_allreduce_) {
      // every rank holds the summed histograms of every feature: the search and its hand-over are the serial ones,
      // and every rank arrives at the same splits by itself
      sa.tasks = staged;
      k::scan_blocks(inl, sa, n_scan, cs);
      if (comm_stream_) {
        LGA_HIP_CHECK(hipEventRecord(x.ev_reduced, comm_stream_));
        LGA_HIP_CHECK(hipStreamWaitEvent(st_, x.ev_reduced, 0));
      }
      prof.mark(PhaseProfile::kFetch, st_);
      prof.mark(PhaseProfile::kCount, st_);
      return;
    }
    sa.best_out = x.best_local.get();
    sa.tasks = staged;
    k::scan_blocks(inl, sa, n_scan, cs);
    comm_->allgather_bytes(x.best_local.get(), x.best_all.get(), sizeof(k::Cand) * (size_t)n_scan, true, cs);
    k::merge_winners(x.best_all.get(), comm_->size(), n_scan, root ? 0 : B, x.scan_totals.get(), qs_dev_.get(), sp_,
                     x.task_left.get(), x.host_block.device_ptr(), x.seq, cs);
    if (comm_stream_) {
      LGA_HIP_CHECK(hipEventRecord(x.ev_reduced, comm_stream_));
      LGA_HIP_CHECK(hipStreamWaitEvent(st_, x.ev_reduced, 0));
    }
    prof.mark(PhaseProfile::kFetch, st_);
    prof.mark(PhaseProfile::kCount, st_);
  }

  // Waits for the round of this context and hands its results over.
  void finish_round(RoundCtx& x, SplitInfo* infos) {
    if (!x.busy) throw std::runtime_error("no round in flight on this context");
    const auto t_wait = std::chrono::steady_clock::now();
    TraceRange wait_range("lga.wait_round");
    if (x.direct == 2 && !x.root) merge_direct_blocks(x, 2 * (int)x.tasks.size());
    else if (x.direct && !x.root) wait_direct(x, 2 * (int)x.tasks.size());
    else wait_flag(x);
    if (prof.on) {
      LGA_HIP_CHECK(hipStreamSynchronize(x.stream));
      if (comm_stream_) LGA_HIP_CHECK(hipStreamSynchronize(comm_stream_));
    }
    prof.host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wait).count();
    prof.collect(PhaseProfile::kCount);
    ++prof.rounds;
    const bool root = x.root;
    const int B = (int)x.tasks.size();
    const int n_scan = root ? 1 : 2 * B;
    if (!root) prof.tasks += B;
    const char* hres = x.host_block.get() + k::kHostDataOffset;
    std::memcpy(infos, hres, sizeof(SplitInfo) * (size_t)n_scan);
    const int32_t* tl = reinterpret_cast<const int32_t*>(hres + sizeof(SplitInfo) * (size_t)n_scan);
    if (root) {
      std::memcpy(&qs_host_, tl, sizeof(QuantScale));
    } else {
      if (x.direct)
        for (int i = 0; i < n_scan; ++i) infos[i].pad0 = 0;   // (the hand-over's sequence tag)
      for (int i = 0; i < B; ++i) {
        const SplitTask& t = x.tasks[(size_t)i];
        const NodeRange p = ranges_host_[(size_t)t.parent_node];
        prof.part_rows += p.count;
        prof.hist_rows += t.hist_child == 0 ? tl[i] : p.count - tl[i];
        ranges_host_[(size_t)t.left_node] = NodeRange{p.begin, tl[i], 1 - p.buf, 0};
        ranges_host_[(size_t)t.right_node] = NodeRange{p.begin + tl[i], p.count - tl[i], 1 - p.buf, 0};
      }
    }
    if (!sorted_cat_.empty()) category_set_pass(x, infos, n_scan);
    x.busy = false;
    prof.last_end = std::chrono::steady_clock::now();
    if (k::device_checks_enabled()) {   // debug build: a kernel of this round may have refused an index
      const std::string failure = k::take_check_failure(x.stream);
      if (!failure.empty()) throw std::runtime_error(failure);
    }
  }

  // Features searched over category sets (cat_search.hpp): the finished histogram rows of those features
  // come back for every node the round scanned, the search runs on the host - the same code the host
  // engine runs - and a set that beats the device scan's winner replaces it in `infos`.
  void category_set_pass(RoundCtx& x, SplitInfo* infos, int n_scan) {
    const int nf = (int)sorted_cat_.size();
    cat_items_host_.resize((size_t)n_scan * (size_t)nf * 2);
    cat_items_dev_.resize((size_t)n_scan * (size_t)nf * 2);
    cat_hist_dev_.resize((size_t)n_scan * (size_t)nf * 2 * kMaxBins);
    cat_hist_host_.resize((size_t)n_scan * (size_t)nf * 2 * kMaxBins);
    std::vector<int> node_of((size_t)n_scan);
    std::vector<uint32_t> key_of((size_t)n_scan);
    for (int s = 0; s < n_scan; ++s) {
      int slot = 0;
      if (x.root) { node_of[(size_t)s] = 0; key_of[(size_t)s] = kRootNodeKey; }
      else {
        const SplitTask& t = x.tasks[(size_t)(s >> 1)];
        const int side = s & 1;
        node_of[(size_t)s] = side == 0 ? t.left_node : t.right_node;
        key_of[(size_t)s] = child_node_key((uint32_t)t.parent_key, side);
        slot = side == t.hist_child ? t.new_slot : t.parent_slot;
      }
      for (int j = 0; j < nf; ++j) {
        cat_items_host_.get()[((size_t)s * (size_t)nf + (size_t)j) * 2] = slot;
        cat_items_host_.get()[((size_t)s * (size_t)nf + (size_t)j) * 2 + 1] = sorted_cat_[(size_t)j];
      }
    }
    const int n_items = n_scan * nf;
    LGA_HIP_CHECK(hipMemcpyAsync(cat_items_dev_.get(), cat_items_host_.get(), sizeof(int32_t) * 2 * (size_t)n_items,
                                 hipMemcpyHostToDevice, x.stream));
    k::gather_feature_hist(lay_, cat_items_dev_.get(), n_items, cat_hist_dev_.get(), x.stream);
    LGA_HIP_CHECK(hipMemcpyAsync(cat_hist_host_.get(), cat_hist_dev_.get(), sizeof(int64_t) * 2 * kMaxBins * (size_t)n_items,
                                 hipMemcpyDeviceToHost, x.stream));
    LGA_HIP_CHECK(hipStreamSynchronize(x.stream));
    const double gunit = unit_of(qs_host_.kg), hunit = unit_of(qs_host_.kh);
    const double kInf = std::numeric_limits<double>::infinity();
    SplitParams sp_host = sp_;          // the node tables of sp_ are device pointers: the host reads its own copies
    sp_host.node_bounds = nullptr;
    sp_host.node_masks = masked_ ? masks_host_.get() : nullptr;
    sp_host.feature_scale = gain_scale_host_;
    sp_host.feature_offset = gain_offset_host_;
    for (int s = 0; s < n_scan; ++s) {
      SplitInfo& o = infos[s];
      const int node = node_of[(size_t)s];
      const double* nb = ranged_ ? bounds_host_.get() + (size_t)kNodeInfoStride * (size_t)node : nullptr;
      const LeafScanCtx ctx = make_scan_ctx(o.total_g, o.total_h, o.total_cnt, gunit, hunit, sp_host, nb ? nb[0] : -kInf,
                                            nb ? nb[1] : kInf, nb ? nb[2] : std::numeric_limits<double>::quiet_NaN());
      CatSetSplit cs;
      cs.gain = o.feature >= 0 ? o.gain : -kInf;
      for (int j = 0; j < nf; ++j) {
        const int f = sorted_cat_[(size_t)j];
        if (!node_feature_on(sp_host, key_of[(size_t)s], f, F_) || !node_allows(sp_host, node, f)) continue;
        const FeatureMeta& m = meta_host_[(size_t)f];
        const int64_t* g = cat_hist_host_.get() + ((size_t)s * (size_t)nf + (size_t)j) * 2 * kMaxBins;
        categorical_set_search(g, g + kMaxBins, m.missing_bin >= 0 ? m.num_bin - 1 : m.num_bin, f, ctx, sp_host, cat_, &cs);
      }
      if (cs.feature < 0) continue;
      o.gain = cs.gain;
      o.feature = cs.feature;
      o.threshold = cs.num_cat;
      o.default_left = kCategoricalSetTask;
      o.left_g = cs.lg; o.left_h = cs.lh;
      o.left_cnt = est_count(cs.lh, ctx);
      for (int w = 0; w < kCatSetWords; ++w) cat_sets_host_.get()[(size_t)node * kCatSetWords + (size_t)w] = cs.bins[w];
    }
  }

  // Waits for the completion flag the round's last workgroup writes into host-mapped memory.  A short spin
  // catches the common case (a round takes tens of microseconds); after that the thread yields its core between
  // polls, so many boosters - cross-validation folds, grid rows - can wait on few cores.  The stream is
  // queried now and then so that a failed launch surfaces as an error instead of an endless wait.
  void wait_flag(RoundCtx& x) { wait_for(reinterpret_cast<volatile unsigned*>(x.host_block.get()), x.seq, x.stream); }
  // direct hand-over (ScanArgs::direct): every result carries the round's sequence number once it is complete
  void wait_direct(RoundCtx& x, int n_scan) {
    const SplitInfo* res = reinterpret_cast<const SplitInfo*>(x.host_block.get() + k::kHostDataOffset);
    for (int i = 0; i < n_scan; ++i)
      wait_for(reinterpret_cast<volatile unsigned*>(const_cast<int32_t*>(&res[i].pad0)), x.seq, x.stream);
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  // ScanArgs::direct == 2: waits for every feature block's record, merges a child's blocks with the search's own
  // total order and writes the SplitInfo records and left counts where the other hand-overs leave them.
  void merge_direct_blocks(RoundCtx& x, int n_scan) {
    const int nb = x.direct_blocks;
    const k::BlockResult* br = reinterpret_cast<const k::BlockResult*>(x.host_block.get() + x.blocks_off);
    for (int i = 0; i < n_scan * nb; ++i)
      wait_for(reinterpret_cast<volatile unsigned*>(const_cast<int32_t*>(&br[i].tag)), x.seq, x.stream);
    std::atomic_thread_fence(std::memory_order_acquire);
    SplitInfo* out = reinterpret_cast<SplitInfo*>(x.host_block.get() + k::kHostDataOffset);
    int32_t* tl = reinterpret_cast<int32_t*>(x.host_block.get() + k::kHostDataOffset + sizeof(SplitInfo) * (size_t)n_scan);
    const double gunit = unit_of(qs_host_.kg), hunit = unit_of(qs_host_.kh);
    const double ninf = -std::numeric_limits<double>::infinity();
    SplitParams sp_host = sp_;   // (the node tables of sp_ are device pointers; a plain context reads none of them)
    sp_host.node_bounds = nullptr;
    sp_host.node_masks = nullptr;
    sp_host.feature_scale = nullptr;
    sp_host.feature_offset = nullptr;
    for (int s = 0; s < n_scan; ++s) {
      const k::BlockResult* b0 = br + (size_t)s * (size_t)nb;
      double bg = b0->gain;
      uint64_t bkey = b0->key;
      int64_t blg = b0->lg, blh = b0->lh;
      for (int b = 1; b < nb; ++b)
        if (cand_better(b0[b].gain, b0[b].key, bg, bkey)) { bg = b0[b].gain; bkey = b0[b].key; blg = b0[b].lg; blh = b0[b].lh; }
      SplitInfo& o = out[s];
      o.total_cnt = b0->tc; o.total_g = b0->tg; o.total_h = b0->th; o.pad0 = 0;
      if (bg == ninf) {
        o.gain = ninf; o.feature = -1; o.threshold = 0; o.default_left = 0; o.left_g = 0; o.left_h = 0; o.left_cnt = 0;
      } else {
        o.gain = bg;
        o.feature = (int32_t)(bkey >> 16);
        o.threshold = (int32_t)((bkey >> 1) & 0x7fff);
        o.default_left = (int32_t)(bkey & 1);
        o.left_g = blg; o.left_h = blh;
        o.left_cnt = est_count(blh, make_scan_ctx(b0->tg, b0->th, b0->tc, gunit, hunit, sp_host));
      }
      if ((s & 1) == 0) tl[s >> 1] = b0->left;
    }
  }
  bool direct_done(const RoundCtx& x, int n_scan) const {
    if (x.direct == 2) {
      const k::BlockResult* br = reinterpret_cast<const k::BlockResult*>(x.host_block.get() + x.blocks_off);
      for (int i = 0; i < n_scan * x.direct_blocks; ++i)
        if (*reinterpret_cast<const volatile unsigned*>(&br[i].tag) != x.seq) return false;
      return true;
    }
    const SplitInfo* res = reinterpret_cast<const SplitInfo*>(x.host_block.get() + k::kHostDataOffset);
    for (int i = 0; i < n_scan; ++i)
      if (*reinterpret_cast<const volatile unsigned*>(&res[i].pad0) != x.seq) return false;
    return true;
  }
  static void wait_for(volatile unsigned* flag, unsigned seq, hipStream_t stream) {
    unsigned spins = 0;
    while (*flag != seq) {
      ++spins;
      // a round takes tens of microseconds, a chunk of boosting iterations tens of milliseconds: spin, then yield, then sleep
      if (spins > 100 * kSpinBeforeYield) std::this_thread::sleep_for(std::chrono::microseconds(30));
      else if (spins > kSpinBeforeYield) std::this_thread::yield();
      if ((spins & 4095u) == 0) {
        hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) {
          if (*flag == seq) break;
          throw std::runtime_error("device work finished without raising its completion flag");
        }
        if (q != hipErrorNotReady) LGA_HIP_CHECK(q);
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  }
  static constexpr unsigned kSpinBeforeYield = 2000;   // ~20-40 us of pause-spinning

  const Dataset* root_;
  DeviceMirror* mir_;
  Comm* comm_;
  ParallelMode mode_;
  hipStream_t stream_ = nullptr;
  hipStream_t comm_stream_ = nullptr;   // distributed, two rounds in flight: the one stream collectives run on
  int F_ = 0, T_ = 0;
  int64_t N_ = 0, slot_words_ = 0;
  int capacity_splits_ = 0, max_batch_ = 0, max_pchunks_ = 0, max_hchunks_ = 0;
  int slots_dirty_ = 1;       // slots written since the pool was last cleared
  int solo_rows_ = 2048;      // leaves up to this many rows are one workgroup's (LGA_SOLO_ROWS)
  int target_units_ = 512;    // units a round's multi-unit leaves are cut into (LGA_TARGET_UNITS)
  int max_unit_rows_ = kMaxUnitRows;   // rows a unit / histogram chunk may hold (LGA_MAX_UNIT_ROWS lowers it)
  k::PoolLayout lay_{};
  k::ScanRange scan_range_{};
  DeviceBuffer<int64_t> pool_, slot_table_;
  DeviceBuffer<NodeRange> nodes_;
  DeviceBuffer<int64_t> node_totals_;
  DeviceBuffer<int32_t> arena_[2];
  DeviceBuffer<int64_t> gh_arena_[2];   // the packed gradients in the order of the row arenas (data with several feature tiles)
  bool ordered_grads_ = !(std::getenv("LGA_ORDERED_GRADS") && std::atoi(std::getenv("LGA_ORDERED_GRADS")) == 0);
  bool dp_allreduce_ = false;   // data-parallel exchange: whole histograms by all-reduce (narrow data) instead of reduce-scatter by feature
  static constexpr int kMaxRoundsInFlight = 4;
  RoundCtx ctx_[kMaxRoundsInFlight];
  int n_ctx_ = 1;
  DeviceBuffer<FeatureMeta> meta_dev_;
  DeviceBuffer<QuantScale> qs_dev_;
  std::vector<FeatureMeta> meta_host_;
  std::vector<NodeRange> ranges_host_;
  size_t off_tasks_ = 0, off_pch_ = 0, off_hch_ = 0, stage_bytes_ = 0;
  SplitParams sp_{};
  bool ranged_ = false;                 // monotone constraints or path smoothing: node_bounds is in use
  PinnedBuffer<double> bounds_host_;    // [node][lo, hi, out]
  DeviceBuffer<double> bounds_dev_;
  const double* gain_scale_host_ = nullptr;   // the grower's tables behind SplitParams::feature_scale / _offset
  const double* gain_offset_host_ = nullptr;
  DeviceBuffer<double> gain_dev_;
  PinnedBuffer<double> gain_pinned_;
  std::vector<int> sorted_cat_;         // features of the current tree searched over category sets
  CatParams cat_;
  PinnedBuffer<uint64_t> cat_sets_host_;   // [node][kCatSetWords]
  DeviceBuffer<uint64_t> cat_sets_dev_;
  PinnedBuffer<int32_t> cat_items_host_;
  DeviceBuffer<int32_t> cat_items_dev_;
  PinnedBuffer<int64_t> cat_hist_host_;
  DeviceBuffer<int64_t> cat_hist_dev_;
  bool masked_ = false;                 // interaction constraints are active for the current tree
  PinnedBuffer<uint64_t> masks_host_;   // [node][words]
  DeviceBuffer<uint64_t> masks_dev_;
  QuantScale qs_host_{0, 0};
  const int32_t* bag_ = nullptr;
  int64_t bag_n_ = 0;
  const int64_t* packed_ = nullptr;
  const uint8_t* hist_src_ = nullptr;
  uint8_t* lines_ = nullptr;
  k::FusedQuant root_quant_{};
  // whole-tree kernel (kernels.hpp: TreeArgs)
  bool tree_kernel_on_ = true;
  bool tree_kernel_forced_ = false;   // whole_tree_max_rows was given explicitly
  bool in_company_ = false;           // the booster is advanced together with others (Engine::set_group_size)
  int64_t tree_kernel_rows_ = k::kTreeMaxRows;
  DeviceBuffer<unsigned long long> rec_gh_[2];
  DeviceBuffer<uint32_t> rec_bins_[2];
  DeviceBuffer<SplitInfo> tree_best_;
  MappedBuffer tree_out_, tree_desc_;
  k::TreeLayout tree_lay_{};
  unsigned tree_seq_ = 0;
  bool tree_prepared_ = false, tree_launched_ = false, tree_wide_valid_ = false;
  int tree_renew_ = 0;
  bool wide_ = false;                 // full-precision (36-bit) histograms
  bool scan_pairs_ = false;           // split search: both children of a split per workgroup (wide data)
  int part_chunk_ = k::kPartChunk;    // rows per partition workgroup of the row-lines rounds (single pass: chosen per round)
  int part_rows_forced_ = 0;          // LGA_PART_ROWS: rows per workgroup of the single-pass partition, 0 = by the round's largest leaf
  int64_t part_rows4k_above_ = 1 << 20, part_rows8k_above_ = 1 << 22;
  bool direct_results_ = !(std::getenv("LGA_DIRECT_RESULTS") && std::atoi(std::getenv("LGA_DIRECT_RESULTS")) == 0);   // see ScanArgs::direct
  bool direct_blocks_ = !(std::getenv("LGA_DIRECT_RESULTS") && std::atoi(std::getenv("LGA_DIRECT_RESULTS")) == 1);    // ... 1: never per block
  // Workgroups a split search aims for: the search kernel keeps 8 waves of ~170 registers per CU - ONE workgroup - so a
  // launch of more workgroups than CUs runs in shifts; rounds of many leaves give a wave several features in turn
  // instead (measured on the 10.5 M x 28 benchmark: 1024 -> 256 workgroups 2.43 -> 2.38 ms per tree; LGA_SCAN_WGS).
  // Data of several feature tiles searches hundreds of features per child and is bound by the histogram bytes it moves:
  // two workgroups' worth per CU measured best there (1 M x 968: 8.2 ms per tree at 256, 6.9 at 512, 7.1 at 1024).
  int scan_wg_target_ = 256;
  int top_k_ = 20;                    // voting-parallel: features every rank votes for per searched child
  bool single_pass_partition_ = true; // row lines: sides + row movement of the large leaves in one launch (grow_partition)
  int64_t whole_trees_ = 0;
  bool tree_timing_ = std::getenv("LGA_TREE_TIMING") != nullptr;
  // LGA_GROW_TIMING=1: phase profile of the non-root growth kernels, printed when the backend goes away
  bool grow_timing_ = std::getenv("LGA_GROW_TIMING") != nullptr;
  DeviceBuffer<unsigned long long> grow_ticks_;
  unsigned long long* grow_ticks() {
    if (grow_ticks_.size() == 0) {
      grow_ticks_.resize(16);
      LGA_HIP_CHECK(hipMemset(grow_ticks_.get(), 0, 16 * sizeof(unsigned long long)));
    }
    return grow_ticks_.get();
  }
  void report_grow_timing() {
    if (!grow_timing_ || grow_ticks_.size() == 0) return;
    unsigned long long v[16];
    if (hipDeviceSynchronize() != hipSuccess) return;
    if (hipMemcpy(v, grow_ticks_.get(), sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return;
    std::fprintf(stderr, "[grow timing] workgroup-ticks (10 ns): flags %llu, prefix %llu, hist rows %llu, masks/move %llu, sums %llu, "
                         "flush %llu (over %llu multi-unit workgroups), solo store %llu\n", v[0], v[1], v[2], v[3], v[4], v[5], v[7], v[6]);
    std::fprintf(stderr, "[grow timing] sides kernel (sampled workgroups): first row ids %llu, blocks %llu, counts %llu (over %llu workgroups)\n", v[8], v[9], v[10], v[12]);
  }
  double tree_ticks_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double tree_cycles_ = 0;
  int tree_timed_ = 0;
};

// ---------------------------------------------------------------------------
// Lets a host-only communicator (torch.distributed / gloo callbacks, the thread loop-back) serve the device
// engine: a device buffer travels through pinned host memory around each collective.  This is for rehearsing
// multi-process runs on a machine that has fewer GPUs than ranks; production runs reduce with RCCL.
// ---------------------------------------------------------------------------
class StagingComm : public Comm {
 public:
  explicit StagingComm(Comm* inner) : inner_(inner) {}
  int rank() const override { return inner_->rank(); }
  int size() const override { return inner_->size(); }
  void allreduce_sum_i64(int64_t* buf, size_t n, bool device, void* stream) override {
    staged(buf, n * sizeof(int64_t), device, stream, [&](void* h) { inner_->allreduce_sum_i64((int64_t*)h, n, false, nullptr); });
  }
  void allreduce_sum_f64(double* buf, size_t n, bool device, void* stream) override {
    staged(buf, n * sizeof(double), device, stream, [&](void* h) { inner_->allreduce_sum_f64((double*)h, n, false, nullptr); });
  }
  void allreduce_max_f32(float* buf, size_t n, bool device, void* stream) override {
    staged(buf, n * sizeof(float), device, stream, [&](void* h) { inner_->allreduce_max_f32((float*)h, n, false, nullptr); });
  }
  void allgather_bytes(const void* in, void* out, size_t nbytes, bool device, void* stream) override {
    if (!device) { inner_->allgather_bytes(in, out, nbytes, false, nullptr); return; }
    hipStream_t s = (hipStream_t)stream;
    const size_t total = nbytes * (size_t)size();
    host_.resize(nbytes + total);
    LGA_HIP_CHECK(hipMemcpyAsync(host_.get(), in, nbytes, hipMemcpyDeviceToHost, s));
    LGA_HIP_CHECK(hipStreamSynchronize(s));
    inner_->allgather_bytes(host_.get(), host_.get() + nbytes, nbytes, false, nullptr);
    LGA_HIP_CHECK(hipMemcpyAsync(out, host_.get() + nbytes, total, hipMemcpyHostToDevice, s));
    LGA_HIP_CHECK(hipStreamSynchronize(s));
  }

 private:
  template <typename F>
  void staged(void* buf, size_t bytes, bool device, void* stream, F reduce_host) {
    if (bytes == 0) return;
    if (!device) { reduce_host(buf); return; }
    hipStream_t s = (hipStream_t)stream;
    host_.resize(bytes);
    LGA_HIP_CHECK(hipMemcpyAsync(host_.get(), buf, bytes, hipMemcpyDeviceToHost, s));
    LGA_HIP_CHECK(hipStreamSynchronize(s));
    reduce_host(host_.get());
    LGA_HIP_CHECK(hipMemcpyAsync(buf, host_.get(), bytes, hipMemcpyHostToDevice, s));
    LGA_HIP_CHECK(hipStreamSynchronize(s));
  }
  Comm* inner_;
  PinnedBuffer<char> host_;
};

// ---------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------
class HipEngine : public Engine {
 public:
  HipEngine(std::shared_ptr<Dataset> train, const Config& cfg, Comm* comm, ParallelMode mode)
      : train_(train), cfg_(cfg), comm_(comm), grower_(cfg, train.get()) {
    if (train_->root_label().empty()) throw std::runtime_error("training data has no label");
    if (comm_ && comm_->host_only()) {
      staging_.reset(new StagingComm(comm_));
      comm_ = comm = staging_.get();
    }
    stream_ = current_stream();
    LGA_HIP_CHECK(hipGetDevice(&device_));
    mir_ = ensure_device_mirror(train_->root());
    N_ = train_->root_rows();
    upload_meta(train_->root(), mir_);
    n_base_ = train_->num_data();
    base_rows_.resize((size_t)std::max<int64_t>(n_base_, 1));
    // Locality order (locality.hpp): large tables of one feature tile, trained whole, get private copies of their
    // matrices in an order that keeps the rows of a deep leaf together.  From here on `mir_` is that private copy, a
    // "root row" inside this engine is a position in it, and base_rows_ (dataset row -> root row) carries the mapping.
    {
      bool want = cfg.locality_order && n_base_ >= kLocalityMinRows;
      if (const char* e = std::getenv("LGA_LOCALITY_ORDER")) want = std::atoi(e) != 0 && n_base_ > k::kTreeMaxRows;
      want = want && !train_->used_indices() && train_->num_tiles() == 1 && train_->query_boundaries().empty() &&
             mode != ParallelMode::Feature;
      if (want) {
        own_mir_.reset(new DeviceMirror());
        locality_ = build_locality_mirror(*mir_, train_->feature_meta(), own_mir_.get(), &order_, stream_);
        if (locality_) mir_ = own_mir_.get();
        else own_mir_.reset();
      }
    }
    if (locality_) {
      LGA_HIP_CHECK(hipMemcpyAsync(base_rows_.get(), order_.int_of_ext.get(), sizeof(int32_t) * (size_t)n_base_, hipMemcpyDeviceToDevice, stream_));
      iter_rows_.resize((size_t)n_base_);
      k::iota(iter_rows_.get(), n_base_, stream_);
    } else if (train_->used_indices())
      LGA_HIP_CHECK(hipMemcpyAsync(base_rows_.get(), train_->used_indices()->data(), sizeof(int32_t) * (size_t)n_base_,
                                   hipMemcpyHostToDevice, stream_));
    else
      k::iota(base_rows_.get(), n_base_, stream_);
    bag_rows_.resize((size_t)std::max<int64_t>(n_base_, 1));
    K_ = std::max(1, cfg.num_class);
    scores_.resize((size_t)N_ * (size_t)K_);
    grad_.resize((size_t)N_ * (size_t)K_);
    hess_.resize((size_t)N_ * (size_t)K_);
    maxbits_.resize(2);
    red_dev_.resize(k::kReduceDoubles);
    red_host_.resize(8);
    if (!train_->root_init_score().empty()) {
      init_score_dev_.upload(train_->root_init_score().data(), train_->root_init_score().size(), stream_);
      has_init_ = true;
      init_per_class_ = (int64_t)train_->root_init_score().size() == (int64_t)K_ * N_ && K_ > 1;
      if (locality_) {   // root order -> this engine's order, class by class
        const size_t vectors = train_->root_init_score().size() / (size_t)N_;
        DeviceBuffer<double> ordered(vectors * (size_t)N_);
        for (size_t v = 0; v < vectors; ++v)
          gather_rows_f64(init_score_dev_.get() + v * (size_t)N_, order_.ext_of_int.get(), N_, ordered.get() + v * (size_t)N_, stream_);
        init_score_dev_ = std::move(ordered);
      }
    }
    backend_.reset(new HipBackend(train_.get(), mir_, cfg, comm, mode, n_base_));
    if (backend_->mode() == ParallelMode::Data || backend_->mode() == ParallelMode::Voting) grower_.set_row_shards(comm->size());
    bag_ptr_ = all_rows();
    bag_n_ = n_base_;
    oob_n_ = 0;
    // Large single-tile datasets get a private copy of the bin records interleaved with the packed
    // gradients (one 128-byte line per three rows); everything else keeps the separate gradient array.
    const bool lines_possible = train_->num_tiles() == 1 && backend_->mode() != ParallelMode::Feature;
    use_lines_ = lines_possible;
    if (const char* e = std::getenv("LGA_ROW_LINES")) use_lines_ = lines_possible && std::string(e) != "0";   // 0 / 1 override the size rule
    if (use_lines_) {
      lines_.resize((size_t)k::row_line_bytes(N_));
      k::build_row_lines(mir_->bins_rt.get(), N_, lines_.get(), stream_);
    } else {
      packed_.resize((size_t)N_);
    }
    if (const char* e = std::getenv("LGA_SCORE_UPDATE"))
      score_update_ = std::string(e) == "leaf" ? 1 : (std::string(e) == "walk" ? 2 : 0);
    LGA_HIP_CHECK(hipStreamSynchronize(stream_));
  }

  // Pinned and mapped blocks go back to a process-wide cache when the members below are destroyed: nothing enqueued
  // on the stream may still read or write them by then.
  ~HipEngine() override {
    if (upload_ev_) { (void)hipEventSynchronize(upload_ev_); (void)hipEventDestroy(upload_ev_); }
    (void)hipStreamSynchronize(stream_);
  }
  const char* name() const override { return "hip"; }
  void bind_thread() override {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != device_) LGA_HIP_CHECK(hipSetDevice(device_));
  }
  int device() const { return device_; }

  int add_valid(std::shared_ptr<Dataset> ds) override {
    if (ds->num_total_features() != train_->num_total_features())
      throw std::runtime_error("validation data has a different number of features");
    if (ds->root_label().empty()) throw std::runtime_error("validation data has no label");
    std::unique_ptr<Valid> v(new Valid());
    v->ds = ds;
    v->mir = ensure_device_mirror(ds->root());
    upload_meta(ds->root(), v->mir);
    v->n = ds->num_data();
    v->N = ds->root_rows();
    v->rows.resize((size_t)std::max<int64_t>(v->n, 1));
    if (ds->used_indices())
      LGA_HIP_CHECK(hipMemcpyAsync(v->rows.get(), ds->used_indices()->data(), sizeof(int32_t) * (size_t)v->n,
                                   hipMemcpyHostToDevice, stream_));
    else
      k::iota(v->rows.get(), v->n, stream_);
    v->shared_scores = false;
    if (!v->shared_scores) v->scores.resize((size_t)v->N * (size_t)K_);
    if (!ds->root_init_score().empty() && !v->shared_scores) {
      v->init.upload(ds->root_init_score().data(), ds->root_init_score().size(), stream_);
      v->has_init = true;
      v->init_per_class = (int64_t)ds->root_init_score().size() == (int64_t)K_ * v->N && K_ > 1;
    }
    LGA_HIP_CHECK(hipStreamSynchronize(stream_));
    valid_.push_back(std::move(v));
    return (int)valid_.size() - 1;
  }
  int num_valid() const override { return (int)valid_.size(); }
  int64_t num_rows(int which) const override { return which < 0 ? n_base_ : valid_[(size_t)which]->n; }

  void label_stats(double* sum_wy, double* sum_w, double* sum_w_pos, bool sqrt_labels) override {
    k::label_sums(mir_->label.get(), weight_ptr(mir_), base_rows_.get(), n_base_, sqrt_labels, red_dev_.get(), stream_);
    red_dev_.download(red_host_.get(), 3, stream_);
    LGA_HIP_CHECK(hipStreamSynchronize(stream_));
    double b[3] = {red_host_[0], red_host_[1], red_host_[2]};
    if (dist()) comm_->allreduce_sum_f64(b, 3, false, stream_);
    *sum_wy = b[0]; *sum_w = b[1]; *sum_w_pos = b[2];
  }

  void reset_scores(double v) override {
    // initial scores: one vector for every class, or one per class ([class][root row])
    const double* own = has_init_ ? init_score_dev_.get() + (init_per_class_ ? (size_t)cls_ * (size_t)N_ : 0) : nullptr;
    ++score_version_;
    k::init_scores(sc(), own, base_or_null(), n_base_, v, stream_);
    for (auto& vs : valid_) {
      const double* vi = vs->has_init ? vs->init.get() + (vs->init_per_class ? (size_t)cls_ * (size_t)vs->N : 0)
                                      : (vs->shared_scores ? own : nullptr);
      k::init_scores(score_ptr(*vs), vi, vs->rows.get(), vs->n, v, stream_);
    }
  }
  void scale_scores(int which, double f) override {
    if (which < 0) { ++score_version_; k::scale_add(sc(), base_or_null(), n_base_, f, 0.0, stream_); }
    else { auto& v = *valid_[(size_t)which]; k::scale_add(score_ptr(v), v.rows.get(), v.n, f, 0.0, stream_); }
  }
  void add_constant(int which, double c) override {
    if (which < 0) { ++score_version_; k::scale_add(sc(), base_or_null(), n_base_, 1.0, c, stream_); }
    else { auto& v = *valid_[(size_t)which]; k::scale_add(score_ptr(v), v.rows.get(), v.n, 1.0, c, stream_); }
  }

  void select_class(int k) override {
    if (k < 0 || k >= K_) throw std::runtime_error("select_class: class out of range");
    cls_ = k;
  }
  // Labels are kept on the host as well; counted there, once per training run.
  void class_weights(int num_class, std::vector<double>* out) override {
    const auto& y = train_->root_label();
    const auto& w = train_->root_weight();
    const std::vector<int32_t>* idx = train_->used_indices();
    out->assign((size_t)num_class, 0.0);
    for (int64_t i = 0; i < n_base_; ++i) {
      const size_t r = idx ? (size_t)(*idx)[(size_t)i] : (size_t)i;
      const int c = (int)y[r];
      if (c < 0 || c >= num_class || (float)c != y[r])
        throw std::runtime_error("multiclass labels must be integers in [0, num_class)");
      (*out)[(size_t)c] += w.empty() ? 1.0 : (double)w[r];
    }
    if (dist()) comm_->allreduce_sum_f64(out->data(), out->size(), false, stream_);
  }

  void compute_gradients(const ObjectiveSpec& obj) override {
    TraceRange range("lga.gradients");
    if (is_multiclass(obj.kind)) {
      // all classes in one pass; the per-class maxima are taken when each class's tree is grown
      k::multiclass_gradients(obj, scores_.get(), N_, mir_->label.get(), weight_ptr(mir_), base_or_null(), n_base_,
                              grad_.get(), hess_.get(), stream_);
      have_max_ = false;
      return;
    }
    if (iteration_obj_ && grad_version_ == score_version_) {
      // taken together with the last tree's score update (add_trained_tree_to_train_scores); the scores are unchanged
      have_max_ = true;
      return;
    }
    k::gradients(obj, scores_.get(), mir_->label.get(), weight_ptr(mir_), base_or_null(), n_base_, grad_.get(),
                 hess_.get(), maxbits_.get(), stream_);
    have_max_ = true;
  }
  void set_iteration_objective(const ObjectiveSpec* obj) override { iteration_obj_ = obj; grad_version_ = 0; ++score_version_; }
  void invalidate_precomputed_gradients() override { grad_version_ = 0; ++score_version_; }
  bool ranking_gradients(const RankingSpec& spec, const std::vector<int64_t>& bounds,
                         const std::vector<double>& inv_max_dcg, uint64_t iter) override {
    if (!rank_ready_) {
      // per-query tables go up once; a query beyond the kernels' LDS budget leaves ranking to the host
      rank_ready_ = true;
      int64_t longest = 0;
      for (size_t q = 0; q + 1 < bounds.size(); ++q) longest = std::max(longest, bounds[q + 1] - bounds[q]);
      rank_on_device_ = longest <= k::kMaxDeviceQuery && !std::getenv("LGA_RANK_ON_HOST");
      if (rank_on_device_) {
        rank_max_query_ = (int)std::max<int64_t>(longest, 1);
        rank_bounds_.upload(bounds.data(), bounds.size(), stream_);
        std::vector<double> imd = inv_max_dcg;
        imd.resize(bounds.size(), 0.0);   // rank_xendcg has none
        rank_imd_.upload(imd.data(), imd.size(), stream_);
        rank_gain_.upload(spec.label_gain.data(), spec.label_gain.size(), stream_);
        std::vector<double> disc((size_t)rank_max_query_);
        for (int r = 0; r < rank_max_query_; ++r) disc[(size_t)r] = rank_discount(r);
        rank_disc_.upload(disc.data(), disc.size(), stream_);
        LGA_HIP_CHECK(hipStreamSynchronize(stream_));   // the host vectors above go out of scope
      }
    }
    if (!rank_on_device_) return false;
    TraceRange range("lga.gradients");
    k::RankArgs a{};
    a.score = sc();
    a.label = mir_->label.get();
    a.weight = weight_ptr(mir_);
    a.to_root = base_or_null();
    a.bounds = rank_bounds_.get();
    a.inv_max_dcg = rank_imd_.get();
    a.label_gain = rank_gain_.get();
    a.discount = rank_disc_.get();
    a.g = gr();
    a.h = he();
    a.nq = (int)bounds.size() - 1;
    a.max_query = rank_max_query_;
    a.sigmoid = spec.sigmoid;
    a.norm = spec.norm ? 1 : 0;
    a.trunc = spec.truncation_level;
    a.seed = spec.seed;
    a.iter = iter;
    if (spec.xendcg) k::xendcg_gradients(a, stream_);
    else k::lambdarank_gradients(a, stream_);
    have_max_ = false;
    return true;
  }
  bool ranking_metric_values(int which, const RankingSpec& spec, const std::vector<int64_t>& bounds,
                             const std::vector<int>& at, std::vector<double>* values) override {
    const int nq = (int)bounds.size() - 1;
    if (nq <= 0 || at.empty() || (int)at.size() > k::kMaxRankCutoffs || std::getenv("LGA_RANK_ON_HOST")) return false;
    if (spec.label_gain.size() > 64) return false;
    int64_t longest = 1;
    for (int q = 0; q < nq; ++q) longest = std::max(longest, bounds[(size_t)q + 1] - bounds[(size_t)q]);
    if (longest > k::kMaxDeviceQuery) return false;
    // per-dataset query offsets go up once; the gain and discount tables are shared and only ever grow
    auto& db = metric_bounds_[which];
    if (!db) {
      db.reset(new DeviceBuffer<int64_t>());
      db->upload(bounds.data(), bounds.size(), stream_);
      LGA_HIP_CHECK(hipStreamSynchronize(stream_));
    }
    if ((int64_t)metric_disc_len_ < longest) {
      std::vector<double> disc((size_t)longest);
      for (int64_t r = 0; r < longest; ++r) disc[(size_t)r] = rank_discount((int)r);
      metric_disc_.upload(disc.data(), disc.size(), stream_);
      std::vector<double> gain(64, 0.0);
      for (size_t g = 0; g < spec.label_gain.size(); ++g) gain[g] = spec.label_gain[g];
      metric_gain_.upload(gain.data(), gain.size(), stream_);
      LGA_HIP_CHECK(hipStreamSynchronize(stream_));
      metric_disc_len_ = (size_t)longest;
    }
    k::RankArgs a{};
    if (which < 0) { a.score = scores_.get(); a.label = mir_->label.get(); a.to_root = base_or_null(); }
    else {
      Valid& v = *valid_[(size_t)which];
      a.score = all_scores(v);
      a.label = v.mir->label.get();
      a.to_root = v.ds->used_indices() ? v.rows.get() : nullptr;
    }
    a.bounds = db->get();
    a.label_gain = metric_gain_.get();
    a.discount = metric_disc_.get();
    a.nq = nq;
    a.max_query = (int)longest;
    k::RankCutoffs cut{};
    cut.n = (int)at.size();
    for (int k = 0; k < cut.n; ++k) cut.k[k] = at[(size_t)k];
    const size_t n_out = (size_t)nq * 2 * at.size();
    tmp_f64_.resize(n_out);
    k::rank_metrics(a, cut, tmp_f64_.get(), stream_);
    values->resize(n_out);
    LGA_HIP_CHECK(hipMemcpyAsync(values->data(), tmp_f64_.get(), sizeof(double) * n_out, hipMemcpyDeviceToHost, stream_));
    LGA_HIP_CHECK(hipStreamSynchronize(stream_));
    return true;
  }
  void set_gradients(const float* g, const float* h, int64_t n) override {
    if (n != n_base_) throw std::runtime_error("custom gradients: wrong length");
    tmp_f32_.resize((size_t)n * 2);
    LGA_HIP_CHECK(hipMemcpyAsync(tmp_f32_.get(), g, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, stream_));
    LGA_HIP_CHECK(hipMemcpyAsync(tmp_f32_.get() + n, h, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, stream_));
    k::scatter_f32(tmp_f32_.get(), base_rows_.get(), n, gr(), stream_);
    k::scatter_f32(tmp_f32_.get() + n, base_rows_.get(), n, he(), stream_);
    LGA_HIP_CHECK(hipStreamSynchronize(stream_));   // g/h are caller memory
    have_max_ = false;
  }
  void get_gradients(float* g, float* h) override {
    tmp_f32_.resize((size_t)n_base_ * 2);
    for (int k = 0; k < K_; ++k) {
      k::gather_f32(grad_.get() + (size_t)k * (size_t)N_, base_rows_.get(), n_base_, tmp_f32_.get(), stream_);
      k::gather_f32(hess_.get() + (size_t)k * (size_t)N_, base_rows_.get(), n_base_, tmp_f32_.get() + n_base_, stream_);
      LGA_HIP_CHECK(hipMemcpyAsync(g + (size_t)k * (size_t)n_base_, tmp_f32_.get(), sizeof(float) * (size_t)n_base_, hipMemcpyDeviceToHost, stream_));
      LGA_HIP_CHECK(hipMemcpyAsync(h + (size_t)k * (size_t)n_base_, tmp_f32_.get() + n_base_, sizeof(float) * (size_t)n_base_, hipMemcpyDeviceToHost, stream_));
      LGA_HIP_CHECK(hipStreamSynchronize(stream_));
    }
  }

  void resample(double fraction, uint64_t seed, uint64_t iter, double neg_fraction) override {
    if (fraction >= 1.0 && (neg_fraction < 0.0 || neg_fraction >= 1.0)) { bag_ptr_ = all_rows(); bag_n_ = n_base_; oob_n_ = 0; return; }
    uint64_t salt = dist() ? ((uint64_t)comm_->rank() << 40) : 0;
    if (neg_fraction >= 0.0) {
      k::BagParams bp{};
      bp.seed = seed; bp.iter = iter; bp.salt = salt; bp.fraction = (float)fraction;
      bp.label = mir_->label.get();
      bp.neg_fraction = (float)neg_fraction;
      bp.ext_ids = ext_ids();
      bag_n_ = backend_->sample_partition(all_rows(), n_base_, bag_rows_.get(), -3, bp);
    } else {
      bag_n_ = backend_->bag_partition(all_rows(), n_base_, bag_rows_.get(), fraction, seed, iter, salt, ext_ids());
    }
    bag_ptr_ = bag_rows_.get();
    oob_n_ = n_base_ - bag_n_;
  }

  void goss_sample(double top_rate, double other_rate, uint64_t seed, uint64_t iter) override {
    if (n_base_ == 0) return;
    const GossPlan plan = goss_plan(n_base_, top_rate, other_rate);
    tmp_f32_.resize((size_t)n_base_ + 8);
    select_scratch_.resize(2048 + 8);
    float* w = tmp_f32_.get();
    float* threshold = tmp_f32_.get() + n_base_;
    k::goss_weights(grad_.get(), hess_.get(), K_, N_, base_or_null(), n_base_, w, stream_);
    k::kth_largest(w, n_base_, plan.top_k, select_scratch_.get(), threshold, stream_);
    k::BagParams bp{};
    bp.seed = seed; bp.iter = iter; bp.salt = dist() ? ((uint64_t)comm_->rank() << 40) : 0;
    bp.fraction = plan.keep_prob;
    bp.g = grad_.get(); bp.h = hess_.get(); bp.threshold = threshold; bp.num_class = K_; bp.stride = N_;
    bp.ext_ids = ext_ids();
    bag_n_ = backend_->sample_partition(all_rows(), n_base_, bag_rows_.get(), -2, bp);
    bag_ptr_ = bag_rows_.get();
    oob_n_ = n_base_ - bag_n_;
    k::goss_amplify(grad_.get(), hess_.get(), K_, N_, bag_ptr_, bag_n_, threshold, plan.multiplier, stream_);
    have_max_ = false;   // the amplified rows may exceed the maxima taken with the gradients
  }

  void set_tree_index(int64_t i) override { tree_index_ = i; grower_.set_tree_index(i); }
  void update_split_config(const Config& c) override { grower_.update_split_config(c); }
  void set_group_size(int n) override { backend_->set_in_company(n >= 2); }
  Tree train_tree(const std::vector<FeatureMeta>& meta) override {
    train_tree_begin(meta);
    return train_tree_finish();
  }
  bool train_tree_begin(const std::vector<FeatureMeta>& meta) override {
    trace_push("lga.train_tree");
    pending_meta_ = meta;
    debug_check();
    if (!have_max_) k::absmax(gr(), he(), base_or_null(), n_base_, maxbits_.get(), stream_);
    if (dist()) comm_->allreduce_max_f32(reinterpret_cast<float*>(maxbits_.get()), 2, true, stream_);
    backend_->set_bag(bag_ptr_, bag_n_);
    RoundingSpec rounding = rounding_spec(cfg_, tree_index_, dist() ? comm_->rank() : 0);
    rounding.ext_ids = ext_ids();
    if (use_lines_) {
      backend_->set_lines(lines_.get());
      backend_->set_root_quant(gr(), he(), maxbits_.get(), rounding);
    } else {
      k::quantize(gr(), he(), bag_ptr_ == all_rows() ? base_or_null() : bag_ptr_, bag_n_, maxbits_.get(),
                  backend_->qs_device(), packed_.get(), backend_->wide(), rounding, stream_);
      backend_->set_packed(packed_.get(), mir_->bins_rt.get());
    }
    tree_begun_ = true;
    leaf_ids_valid_ = false;
    const bool prepared = grower_.grow_begin(*backend_, pending_meta_);
    trace_pop();
    return prepared;
  }
  Tree train_tree_finish() override {
    if (!tree_begun_) throw std::runtime_error("train_tree_finish without train_tree_begin");
    tree_begun_ = false;
    TraceRange range("lga.train_tree");
    backend_->whole_tree_launch();   // no-op unless a prepared build is still waiting
    return grower_.grow_finish(*backend_, pending_meta_, &leaf_nodes_);
  }
  // The prepared whole-tree builds of a group of engines go out as ONE launch, one workgroup per tree, when the
  // engines share a stream (boosters created from one thread do); the descriptors are copied into a block this
  // engine owns, which stays untouched until the group's next launch.
  void launch_prepared_trees(Engine* const* group, int n) override {
    std::vector<HipEngine*> same;
    for (int i = 0; i < n; ++i) {
      HipEngine* e = dynamic_cast<HipEngine*>(group[i]);
      if (!e) continue;
      const k::TreeArgs* d = e->backend_->whole_tree_descriptor();
      if (!d) continue;
      if (e->stream_ == stream_ && e->train_->num_features() == train_->num_features()) same.push_back(e);
      else e->backend_->whole_tree_launch();
    }
    if (same.empty()) return;
    if (same.size() == 1) { same[0]->backend_->whole_tree_launch(); return; }
    TraceRange range("lga.whole_tree_batch");
    batch_desc_.resize(sizeof(k::TreeArgs) * same.size());
    k::TreeArgs* dst = reinterpret_cast<k::TreeArgs*>(batch_desc_.get());
    for (size_t i = 0; i < same.size(); ++i) dst[i] = *same[i]->backend_->whole_tree_descriptor();
    k::grow_trees(reinterpret_cast<const k::TreeArgs*>(batch_desc_.device_ptr()), (int)same.size(), train_->num_features(), stream_);
    for (HipEngine* e : same) e->backend_->whole_tree_launched();
  }

  // ---- chunks of iterations on the device (kernels.hpp: BoostArgs) ----
  bool train_chunk_begin(const ChunkPlan& plan) override {
    chunk_prepared_ = false;
    const int n_iters = (int)plan.iters.size();
    if (n_iters < 1 || K_ != 1 || dist() || !use_lines_) return false;
    if (!backend_->tree_kernel_possible(n_base_, cfg_.num_leaves - 1)) return false;
    if (plan.metrics.size() > 4 || (plan.eval_valid && valid_.size() != 1) || valid_.size() > 1) return false;
    if (!valid_.empty() && (valid_[0]->n > k::kTreeMaxRows || valid_[0]->shared_scores)) return false;
    for (const ChunkIterPlan& it : plan.iters) {
      if (!grower_.plain_trees(it.meta)) return false;
      for (const FeatureMeta& m : it.meta) if (m.categorical == kCategoricalSorted) return false;
    }
    grower_.set_tree_index(plan.iters[0].tree_index);
    const SplitParams sp0 = grower_.split_params();
    if (sp0.feature_scale) return false;        // per-feature gain adjustment changes from tree to tree on the host
    // the descriptor's feature table has every usable feature enabled; the iterations' masks narrow it
    std::vector<FeatureMeta> base_meta = plan.iters[0].meta;
    const std::vector<FeatureMeta> all = train_->feature_meta();
    for (size_t f = 0; f < base_meta.size(); ++f) base_meta[f].enabled = all[f].enabled;
    backend_->set_tree_params(base_meta, sp0);
    backend_->set_bag(bag_ptr_, bag_n_);
    backend_->set_lines(lines_.get());
    backend_->set_root_quant(gr(), he(), maxbits_.get(), RoundingSpec{});
    const int max_splits = cfg_.num_leaves - 1;
    const k::TreeLayout lay = k::tree_layout(max_splits);
    chunk_out_.resize((size_t)lay.bytes * (size_t)n_iters);
    chunk_iters_.resize(sizeof(k::BoostIter) * (size_t)n_iters);
    chunk_desc_.resize(sizeof(k::BoostArgs));
    reinterpret_cast<k::TreeHeader*>(chunk_out_.get())->flag = 0;
    k::BoostArgs& b = *reinterpret_cast<k::BoostArgs*>(chunk_desc_.get());
    b = k::BoostArgs{};
    backend_->fill_tree_args(b.t, max_splits, chunk_out_.device_ptr());
    b.t.renew = plan.renew ? 1 : 0;
    b.n_iters = n_iters;
    b.n_base = (int32_t)n_base_;
    b.base_rows = base_rows_.get();
    b.bag_rows = bag_rows_.get();
    ++score_version_;
    b.scores = sc();
    b.label = mir_->label.get();
    b.weight = weight_ptr(mir_);
    b.g = gr(); b.h = he(); b.maxbits = maxbits_.get();
    b.obj = plan.obj;
    b.bagging = plan.bagging;
    b.bag_fraction = (float)plan.bag_fraction;
    b.neg_fraction = (float)plan.neg_fraction;
    b.bag_seed = plan.bag_seed;
    b.learning_rate = plan.learning_rate;
    b.bins_fm = mir_->bins_fm.get();
    b.N = N_;
    if (!valid_.empty()) {
      Valid& v = *valid_[0];
      b.n_valid = (int32_t)v.n;
      b.valid_N = v.N;
      b.valid_rows = v.ds->used_indices() ? v.rows.get() : nullptr;
      b.valid_bins_fm = v.mir->bins_fm.get();
      b.valid_scores = score_ptr(v);
      b.valid_label = v.mir->label.get();
      b.valid_weight = weight_ptr(v.mir);
    }
    b.n_metrics = plan.eval_valid ? (int32_t)plan.metrics.size() : 0;
    for (size_t m = 0; m < plan.metrics.size() && m < 4; ++m) b.metrics[m] = plan.metrics[m];
    k::BoostIter* its = reinterpret_cast<k::BoostIter*>(chunk_iters_.get());
    for (int j = 0; j < n_iters; ++j) {
      const ChunkIterPlan& it = plan.iters[(size_t)j];
      grower_.set_tree_index(it.tree_index);
      const SplitParams spj = grower_.split_params();
      uint32_t mask = 0;
      for (size_t f = 0; f < it.meta.size(); ++f) if (it.meta[f].enabled) mask |= 1u << f;
      its[j] = k::BoostIter{mask, it.redraw ? 1u : 0u, it.bag_iter, spj};
    }
    b.iters = reinterpret_cast<const k::BoostIter*>(chunk_iters_.device_ptr());
    chunk_plan_ = plan;
    chunk_lay_ = lay;
    chunk_seq_ = b.t.seq;
    chunk_prepared_ = true;
    chunk_launched_ = false;
    return true;
  }
  void launch_prepared_chunks(Engine* const* group, int n) override {
    std::vector<HipEngine*> same;
    for (int i = 0; i < n; ++i) {
      HipEngine* e = dynamic_cast<HipEngine*>(group[i]);
      if (!e || !e->chunk_prepared_ || e->chunk_launched_) continue;
      if (e->stream_ == stream_ && e->train_->num_features() == train_->num_features()) same.push_back(e);
      else e->launch_own_chunk();
    }
    if (same.empty()) return;
    if (same.size() == 1) { same[0]->launch_own_chunk(); return; }
    TraceRange range("lga.boost_chunk_batch");
    batch_desc_.resize(sizeof(k::BoostArgs) * same.size());
    k::BoostArgs* dst = reinterpret_cast<k::BoostArgs*>(batch_desc_.get());
    for (size_t i = 0; i < same.size(); ++i) dst[i] = *reinterpret_cast<const k::BoostArgs*>(same[i]->chunk_desc_.get());
    k::boost_chunks(reinterpret_cast<const k::BoostArgs*>(batch_desc_.device_ptr()), (int)same.size(), train_->num_features(), stream_);
    for (HipEngine* e : same) e->chunk_launched_ = true;
  }
  void launch_own_chunk() {
    k::boost_chunks(reinterpret_cast<const k::BoostArgs*>(chunk_desc_.device_ptr()), 1, train_->num_features(), stream_);
    chunk_launched_ = true;
  }
  void train_chunk_finish(std::vector<ChunkTree>* out) override {
    if (!chunk_prepared_) throw std::runtime_error("train_chunk_finish without a prepared chunk");
    if (!chunk_launched_) launch_own_chunk();
    {
      TraceRange wait_range("lga.wait_chunk");
      HipBackend::wait_chunk(reinterpret_cast<volatile unsigned*>(chunk_out_.get()), chunk_seq_, stream_);
    }
    chunk_prepared_ = false;
    const int n_iters = (int)chunk_plan_.iters.size();
    out->clear();
    out->resize((size_t)n_iters);
    for (int j = 0; j < n_iters; ++j) {
      const char* block = chunk_out_.get() + (size_t)chunk_lay_.bytes * (size_t)j;
      TreeBackend::WholeTree wt;
      const k::TreeHeader& hd = HipBackend::read_tree_block(block, chunk_lay_, &wt);
      const NodeRange* ranges = reinterpret_cast<const NodeRange*>(block + chunk_lay_.off_range);
      const int64_t* wide = reinterpret_cast<const int64_t*>(block + chunk_lay_.off_wide);
      ChunkTree& ct = (*out)[(size_t)j];
      grower_.set_tree_index(chunk_plan_.iters[(size_t)j].tree_index);
      const QuantScale qs{hd.kg, hd.kh};
      std::vector<LeafPiece> pieces;
      ct.tree = grower_.assemble_whole(*backend_, chunk_plan_.iters[(size_t)j].meta, wt, qs, &pieces);
      ct.kg = hd.kg + kWideExtraBits;
      ct.kh = hd.kh + kWideExtraBits;
      if (chunk_plan_.renew) {
        ct.leaf_sums.assign((size_t)2 * (size_t)ct.tree.num_leaves, 0);
        for (const LeafPiece& p : pieces) {
          ct.leaf_sums[(size_t)2 * (size_t)p.leaf] += wide[2 * p.node];
          ct.leaf_sums[(size_t)2 * (size_t)p.leaf + 1] += wide[2 * p.node + 1];
        }
      }
      for (size_t m = 0; m < chunk_plan_.metrics.size() && chunk_plan_.eval_valid; ++m) {
        ct.metric_sums.push_back(hd.metric[m][0]);
        ct.metric_sums.push_back(hd.metric[m][1]);
      }
      if (j == n_iters - 1) {
        // the engine's state after the chunk: the bag, the last tree's row ranges and scales
        if (chunk_plan_.bagging) { bag_ptr_ = bag_rows_.get(); bag_n_ = hd.bag_n; oob_n_ = n_base_ - bag_n_; }
        backend_->set_quant_scale(qs);
        (void)ranges;
        leaf_nodes_.clear();
      }
      if (backend_->tree_timing()) backend_->note_tree_timing(hd, 1);
    }
    have_max_ = false;
    chunk_iters_total_ += n_iters;
  }

  // The new tree's leaf values reach the training scores either through the leaf row ranges the grower
  // left behind (a scattered read-modify-write per row) or by walking the tree over the binned rows in
  // storage order (coalesced scores, ~depth one-byte bin reads per row).  Large row sets favour the walk.
  void add_trained_tree_to_train_scores(const Tree& tree) override {
    TraceRange range("lga.score_update");
    const bool walk = score_update_ == 2 || (score_update_ == 0 && n_base_ >= kWalkMinRows);
    ++score_version_;
    // Leaf values that needed no renewal: nobody has walked the rows yet.  Then walk, score update and the next
    // iteration's gradients are one pass (kernels.hpp: walk_update_gradients) ...
    if (!leaf_ids_valid_ && leaf_values_from_exact_sums() && fused_walk_ && walk_pays(tree) && mir_->tiles == 1 && iteration_obj_ &&
        K_ == 1 && !is_multiclass(iteration_obj_->kind) && fuse_gradients_) {
      const TreeOnDevice td = upload_tree(tree);
      if (k::walk_update_gradients(*iteration_obj_, td.nodes, td.n_internal, td.leaf_value, td.num_leaves, mir_->bins_rt.get(), N_,
                                   base_or_null(), n_base_, sc(), mir_->label.get(), weight_ptr(mir_), grad_.get(), hess_.get(),
                                   maxbits_.get(), td.node_sets, stream_)) {
        grad_version_ = score_version_;
        return;
      }
    }
    // ... or, where that does not apply, a walk for the rows' leaves and a lookup pass
    if (!leaf_ids_valid_ && leaf_values_from_exact_sums()) walk_rows_to_leaves(tree);
    if (leaf_ids_valid_ && leaf_ids_leaves_ == tree.num_leaves) {
      // the leaf of every row is known from the walk that took the leaf sums: the tree's structure has not changed
      // since, only its leaf values
      leaf_ids_valid_ = false;
      const TreeOnDevice td = upload_tree(tree);
      if (iteration_obj_ && K_ == 1 && !is_multiclass(iteration_obj_->kind) && fuse_gradients_) {
        // ... and the next iteration's gradients come out of the same pass
        k::leaf_value_add_gradients(*iteration_obj_, leaf_of_row_.get(), td.leaf_value, td.num_leaves, base_or_null(), n_base_, sc(),
                                    mir_->label.get(), weight_ptr(mir_), grad_.get(), hess_.get(), maxbits_.get(), stream_);
        grad_version_ = score_version_;
        return;
      }
      k::leaf_value_add(leaf_of_row_.get(), td.leaf_value, td.num_leaves, base_or_null(), n_base_, sc(), stream_);
      return;
    }
    if (walk) {
      traverse(upload_tree(tree), mir_, N_, base_or_null(), n_base_, 1.0, sc());
      return;
    }
    const int L = upload_spans(&tree);
    k::leaf_add(span_dev_.get(), L, backend_->arena(0), backend_->arena(1), bag_n_, sc(), stream_);
    if (oob_n_ > 0) traverse(upload_tree(tree), mir_, N_, bag_ptr_ + bag_n_, oob_n_, 1.0, sc());
  }

  // Unweighted: the residuals are ranked on the device (one sort of (leaf, residual) keys, then one thread
  // per leaf reads its percentile off the sorted segment).  Weighted (row weights or MAPE's label scale): the
  // (leaf, residual, weight) triples go to the host, which ranks them like the host engine does.
  void leaf_residual_percentiles(const Tree& tree, double alpha, bool mape_scale,
                                 std::vector<double>* values) override {
    TraceRange range("lga.leaf_percentiles");
    const int L = tree.num_leaves;
    const std::vector<double> keep = *values;
    std::vector<int> has((size_t)L, 0);
    const int n_spans = upload_spans(nullptr);
    const bool weighted = mape_scale || mir_->has_weight;
    if (bag_n_ > 0 && !weighted) {
      const int64_t npad = sort_padded_size(bag_n_);
      sort_keys_.resize((size_t)npad);
      pct_values_.resize((size_t)L);
      pct_counts_.resize((size_t)L);
      k::residual_keys(span_dev_.get(), n_spans, backend_->arena(0), backend_->arena(1), bag_n_, sc(),
                       mir_->label.get(), sort_keys_.get(), npad, stream_);
      bitonic_sort_u64(sort_keys_.get(), npad, 1, stream_);
      k::leaf_percentiles(sort_keys_.get(), bag_n_, L, alpha, pct_values_.get(), pct_counts_.get(), stream_);
      std::vector<int32_t> counts((size_t)L);
      LGA_HIP_CHECK(hipMemcpyAsync(values->data(), pct_values_.get(), sizeof(double) * (size_t)L, hipMemcpyDeviceToHost, stream_));
      LGA_HIP_CHECK(hipMemcpyAsync(counts.data(), pct_counts_.get(), sizeof(int32_t) * (size_t)L, hipMemcpyDeviceToHost, stream_));
      LGA_HIP_CHECK(hipStreamSynchronize(stream_));
      for (int l = 0; l < L; ++l) has[(size_t)l] = counts[(size_t)l] > 0;
    } else if (bag_n_ > 0) {
      const size_t n = (size_t)bag_n_;
      item_leaf_.resize(n);
      tmp_f32_.resize(2 * n);
      k::residual_items(span_dev_.get(), n_spans, backend_->arena(0), backend_->arena(1), bag_n_, sc(),
                        mir_->label.get(), weight_ptr(mir_), mape_scale ? 1 : 0, item_leaf_.get(), tmp_f32_.get(),
                        tmp_f32_.get() + n, stream_);
      std::vector<int32_t> leaf(n);
      std::vector<float> val(n), wt(n);
      LGA_HIP_CHECK(hipMemcpyAsync(leaf.data(), item_leaf_.get(), sizeof(int32_t) * n, hipMemcpyDeviceToHost, stream_));
      LGA_HIP_CHECK(hipMemcpyAsync(val.data(), tmp_f32_.get(), sizeof(float) * n, hipMemcpyDeviceToHost, stream_));
      LGA_HIP_CHECK(hipMemcpyAsync(wt.data(), tmp_f32_.get() + n, sizeof(float) * n, hipMemcpyDeviceToHost, stream_));
      LGA_HIP_CHECK(hipStreamSynchronize(stream_));
      std::vector<std::vector<WeightedResidual>> items((size_t)L);
      for (size_t i = 0; i < n; ++i) items[(size_t)leaf[i]].push_back(WeightedResidual{val[i], wt[i]});
#pragma omp parallel for schedule(dynamic)
      for (int l = 0; l < L; ++l) {
        auto& v = items[(size_t)l];
        if (v.empty()) continue;
        has[(size_t)l] = 1;
        std::sort(v.begin(), v.end(), [](const WeightedResidual& a, const WeightedResidual& b) {
          return a.value != b.value ? a.value < b.value : a.weight < b.weight;
        });
        (*values)[(size_t)l] = percentile_weighted(v.data(), (int64_t)v.size(), alpha);
      }
    }
    merge_leaf_percentiles(comm_, *values, has, keep);
  }

  void add_tree_by_traversal(int which, const Tree& tree, double sign) override {
    if (tree.num_leaves > 1 && (tree.split_feature_inner.empty() || tree.split_feature_inner[0] < 0))
      throw std::runtime_error("tree has no bin thresholds (loaded from a model file); cannot traverse binned data");
    const TreeOnDevice td = upload_tree(tree);
    if (which < 0) {
      ++score_version_;
      traverse(td, mir_, N_, base_or_null(), n_base_, sign, sc());
    } else {
      auto& v = *valid_[(size_t)which];
      traverse(td, v.mir, v.N, v.ds->used_indices() ? v.rows.get() : nullptr, v.n, sign, score_ptr(v));
    }
  }

  std::vector<EvalResult> eval(int which, const std::vector<int32_t>& metrics, const ObjectiveSpec& obj) override {
    TraceRange range("lga.eval");
    debug_check();
    // class 0's vector; class k's follows at k * stride
    const double* sc; const DeviceMirror* m; const int32_t* rows; int64_t n; const Dataset* ds; int64_t stride;
    if (which < 0) { sc = scores_.get(); m = mir_; rows = base_rows_.get(); n = n_base_; ds = train_.get(); stride = N_; }
    else { auto& v = *valid_[(size_t)which]; sc = all_scores(v); m = v.mir; rows = v.rows.get(); n = v.n; ds = v.ds.get(); stride = v.N; }
    std::vector<EvalResult> out;
    for (int32_t mk : metrics) {
      EvalResult r;
      r.name = metric_name(mk);
      r.higher_better = metric_higher_better(mk);
      if (mk == kMetAuc && !dist() && !m->has_weight) {
        // rank statistic, sorted and scanned on the device
        r.value = auc_on_device(sc, m->label.get(), ds->used_indices() ? rows : nullptr, n, stream_);
      } else if (mk == kMetAuc) {
        // weighted or distributed: scores come back to the host (exact weighted sort / binned merge over ranks)
        std::vector<double> hs((size_t)n);
        tmp_f64_.resize((size_t)std::max<int64_t>(n, 1));
        k::gather_f64(sc, rows, n, tmp_f64_.get(), stream_);
        LGA_HIP_CHECK(hipMemcpyAsync(hs.data(), tmp_f64_.get(), sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream_));
        LGA_HIP_CHECK(hipStreamSynchronize(stream_));
        std::vector<float> lab = ds->local_label();
        std::vector<float> w = ds->local_weight();
        std::vector<int32_t> met{kMetAuc};
        auto rr = eval_metrics_host(hs.data(), lab.data(), w.empty() ? nullptr : w.data(), nullptr, n, met, obj, comm_);
        r.value = rr[0].value;
      } else {
        if (mk == kMetMultiLogloss || mk == kMetMultiError)
          k::multi_metric_sum(mk, obj, sc, stride, m->label.get(), weight_ptr(m), rows, n, red_dev_.get(), stream_);
        else
          k::metric_sum(mk, obj, sc, m->label.get(), weight_ptr(m), rows, n, red_dev_.get(), stream_);
        red_dev_.download(red_host_.get(), 2, stream_);
        LGA_HIP_CHECK(hipStreamSynchronize(stream_));
        double b[2] = {red_host_[0], red_host_[1]};
        if (dist()) comm_->allreduce_sum_f64(b, 2, false, stream_);
        double v = b[1] > 0.0 ? b[0] / b[1] : 0.0;
        if (mk == kMetRmse) v = std::sqrt(v);
        r.value = v;
      }
      out.push_back(r);
    }
    return out;
  }

  void get_scores(int which, double* out) override {
    const double* sc; const int32_t* rows; int64_t n, stride;
    if (which < 0) { sc = scores_.get(); rows = base_rows_.get(); n = n_base_; stride = N_; }
    else { auto& v = *valid_[(size_t)which]; sc = all_scores(v); rows = v.rows.get(); n = v.n; stride = v.N; }
    tmp_f64_.resize((size_t)std::max<int64_t>(n, 1) * (size_t)K_);
    for (int k = 0; k < K_; ++k)
      k::gather_f64(sc + (size_t)k * (size_t)stride, rows, n, tmp_f64_.get() + (size_t)k * (size_t)n, stream_);
    LGA_HIP_CHECK(hipMemcpyAsync(out, tmp_f64_.get(), sizeof(double) * (size_t)n * (size_t)K_, hipMemcpyDeviceToHost, stream_));
    LGA_HIP_CHECK(hipStreamSynchronize(stream_));
  }

  std::string profile_json() const override {
    std::string j = backend_->prof.json();
    j.insert(1, "\"chunk_iterations\":" + std::to_string(chunk_iters_total_) + ",\"whole_trees\":" +
                    std::to_string(backend_->whole_trees_grown()) + ",");
    return j;
  }
  GrowStats grow_stats() const override { return grower_.last_stats(); }
  void sum_over_ranks(double* v, size_t n) override {
    if (dist()) comm_->allreduce_sum_f64(v, n, false, stream_);
  }
  void sum_over_ranks_i64(int64_t* v, size_t n) override {   // feature-parallel ranks all hold every row
    if (dist() && (backend_->mode() == ParallelMode::Data || backend_->mode() == ParallelMode::Voting))
      comm_->allreduce_sum_i64(v, n, false, stream_);
  }
  // (LGA_RENEW_ALWAYS=1 keeps the renewal pass in full precision too: a check that it changes nothing)
  bool leaf_values_from_exact_sums() const override { return backend_->wide() && !renew_always_; }
  // Large data with every row in the bag: one walk over the rows in storage order yields every row's leaf - which turns
  // the score update into a lookup - and the leaves' 36-bit gradient sums.
  bool walk_pays(const Tree& tree) {
    // (a walk costs one dependent bin read per level; LGA_WALK_DEPTH caps the row-weighted mean leaf depth it is taken for)
    return score_update_ != 1 && n_base_ >= kWalkMinRows && oob_n_ == 0 && bag_n_ == n_base_ && tree.num_leaves > 1 &&
           !tree.split_feature_inner.empty() && tree.split_feature_inner[0] >= 0 && mean_leaf_depth(tree) <= walk_max_depth_;
  }
  bool walk_rows_to_leaves(const Tree& tree) {
    const int L = tree.num_leaves;
    if (!walk_pays(tree)) return false;
    leaf_acc_.resize((size_t)2 * (size_t)L);
    const TreeOnDevice td = upload_tree(tree);
    leaf_of_row_.resize((size_t)N_);
    const bool walked = k::walk_leaves(td.nodes, td.n_internal, td.num_leaves, mir_->bins_fm.get(),
                                       mir_->tiles == 1 ? mir_->bins_rt.get() : nullptr, N_, base_or_null(), n_base_, gr(), he(),
                                       backend_->qs_device(), backend_->renew_extra_bits(), leaf_of_row_.get(), leaf_acc_.get(), td.node_sets, stream_);
    leaf_ids_valid_ = walked;
    leaf_ids_leaves_ = L;
    return walked;
  }
  void leaf_gradient_sums(const Tree& tree, std::vector<int64_t>* sums, int* kg, int* kh) override {
    TraceRange range("lga.leaf_sums");
    const int L = tree.num_leaves;
    {
      // a tree grown whole comes with its leaves' sums
      int64_t wg = 0, wh = 0;
      if (!leaf_nodes_.empty() && backend_->whole_tree_leaf_sums(leaf_nodes_[0].node, &wg, &wh)) {
        sums->assign((size_t)2 * (size_t)L, 0);
        for (const LeafPiece& p : leaf_nodes_) {
          backend_->whole_tree_leaf_sums(p.node, &wg, &wh);
          (*sums)[(size_t)2 * (size_t)p.leaf] += wg;
          (*sums)[(size_t)2 * (size_t)p.leaf + 1] += wh;
        }
        const QuantScale qs = backend_->quant_scale();
        *kg = qs.kg + kWideExtraBits;
        *kh = qs.kh + kWideExtraBits;
        return;
      }
    }
    leaf_acc_.resize((size_t)2 * (size_t)L);
    // Large data with every row in the bag: one walk over the rows in storage order yields the sums AND every row's
    // leaf, which turns the score update that follows into a lookup (add_trained_tree_to_train_scores).
    leaf_ids_valid_ = false;
    const bool walked = walk_rows_to_leaves(tree);
    if (!walked) {
      const int n_spans = upload_spans(nullptr);
      k::leaf_sums(span_dev_.get(), n_spans, L, backend_->arena(0), backend_->arena(1), bag_n_, gr(), he(),
                   backend_->qs_device(), backend_->renew_extra_bits(), leaf_acc_.get(), stream_);
    }
    sums->assign((size_t)2 * (size_t)L, 0);
    LGA_HIP_CHECK(hipMemcpyAsync(sums->data(), leaf_acc_.get(), sizeof(int64_t) * 2 * (size_t)L, hipMemcpyDeviceToHost, stream_));
    LGA_HIP_CHECK(hipStreamSynchronize(stream_));
    const QuantScale qs = backend_->quant_scale();
    *kg = qs.kg + backend_->renew_extra_bits();
    *kh = qs.kh + backend_->renew_extra_bits();
  }

  // raw device pointers (used by the bindings for zero-copy access)
  double* scores_device() { return scores_.get(); }
  int64_t root_rows() const { return N_; }

 private:
  struct Valid {
    std::shared_ptr<Dataset> ds;
    DeviceMirror* mir = nullptr;
    int64_t n = 0, N = 0;
    DeviceBuffer<int32_t> rows;
    DeviceBuffer<double> scores, init;
    bool shared_scores = false, has_init = false, init_per_class = false;
  };
  bool dist() const { return comm_ && comm_->size() > 1; }
  // the selected class's vectors
  double* sc() { return scores_.get() + (size_t)cls_ * (size_t)N_; }
  float* gr() { return grad_.get() + (size_t)cls_ * (size_t)N_; }
  float* he() { return hess_.get() + (size_t)cls_ * (size_t)N_; }
  double* all_scores(Valid& v) { return v.shared_scores ? scores_.get() : v.scores.get(); }
  double* score_ptr(Valid& v) { return all_scores(v) + (size_t)cls_ * (size_t)v.N; }
  static const float* weight_ptr(const DeviceMirror* m) { return m->has_weight ? m->weight.get() : nullptr; }

  void upload_meta(const Dataset* root, DeviceMirror* m) {
    const auto& y = root->root_label();
    if (!y.empty()) m->label.upload(y.data(), y.size(), stream_);
    const auto& w = root->root_weight();
    m->has_weight = !w.empty();
    if (m->has_weight) m->weight.upload(w.data(), w.size(), stream_);
    LGA_HIP_CHECK(hipStreamSynchronize(stream_));
  }

  // Row ranges of the last grown tree's leaves -> span_dev_, sorted by begin (empty ranges first, so the last
  // entry with begin <= p is the one that owns position p).  Returns the number of spans.
  int upload_spans(const Tree* tree) {
    const int L = (int)leaf_nodes_.size();
    std::vector<k::LeafSpan> spans((size_t)L);
    for (int l = 0; l < L; ++l) {
      const NodeRange& r = backend_->range(leaf_nodes_[(size_t)l].node);
      const int leaf = leaf_nodes_[(size_t)l].leaf;
      spans[(size_t)l] = k::LeafSpan{r.begin, r.begin + r.count, r.buf, leaf, tree ? tree->leaf_value[(size_t)leaf] : 0.0};
    }
    std::sort(spans.begin(), spans.end(), [](const k::LeafSpan& a, const k::LeafSpan& b) {
      if (a.begin != b.begin) return a.begin < b.begin;
      return a.end < b.end;
    });
    wait_upload();
    span_host_.resize((size_t)L);
    std::memcpy(span_host_.get(), spans.data(), sizeof(k::LeafSpan) * (size_t)L);
    span_dev_.resize((size_t)L);
    LGA_HIP_CHECK(hipMemcpyAsync(span_dev_.get(), span_host_.get(), sizeof(k::LeafSpan) * (size_t)L, hipMemcpyHostToDevice, stream_));
    mark_upload();
    return L;
  }

  // Mean depth of the leaves, weighted by their row counts.
  static double mean_leaf_depth(const Tree& tree) {
    const int L = tree.num_leaves;
    if (L <= 1) return 0.0;
    std::vector<int> depth((size_t)(L - 1), 0);
    double sum = 0.0, rows = 0.0;
    for (int i = 0; i < L - 1; ++i) {      // children are created after their parents: a node's depth is known when it is reached
      for (int c : {tree.left_child[(size_t)i], tree.right_child[(size_t)i]}) {
        if (c >= 0) depth[(size_t)c] = depth[(size_t)i] + 1;
        else { sum += (double)(depth[(size_t)i] + 1) * (double)tree.leaf_count[(size_t)~c]; rows += (double)tree.leaf_count[(size_t)~c]; }
      }
    }
    return rows > 0.0 ? sum / rows : 0.0;
  }

  void debug_check() {
    if (!k::device_checks_enabled()) return;
    const std::string failure = k::take_check_failure(stream_);
    if (!failure.empty()) throw std::runtime_error(failure);
  }

  void wait_upload() { if (upload_pending_) { LGA_HIP_CHECK(hipEventSynchronize(upload_ev_)); upload_pending_ = false; } }
  void mark_upload() {
    if (!upload_ev_) LGA_HIP_CHECK(hipEventCreateWithFlags(&upload_ev_, hipEventDisableTiming));
    LGA_HIP_CHECK(hipEventRecord(upload_ev_, stream_));
    upload_pending_ = true;
  }

  // Device copy of a tree for binned traversal: [TreeNodeDev x (L-1) | leaf values x L].  The same tree is
  // usually applied to several score vectors in a row (out-of-bag rows, every validation set), so the
  // upload is skipped when the packed bytes have not changed.
  struct TreeOnDevice {
    const k::TreeNodeDev* nodes;
    const double* leaf_value;
    const uint64_t* node_sets;   // null unless a node splits on a set of categories
    int n_internal, num_leaves;
  };
  TreeOnDevice upload_tree(const Tree& tree) {
    const int L = tree.num_leaves, ni = L - 1;
    bool has_sets = false;
    for (int i = 0; i < ni; ++i)
      has_sets = has_sets || ((tree.decision_type[(size_t)i] & kCategoricalMask) && tree.threshold_bin[(size_t)i] < 0);
    const size_t doff = sizeof(k::TreeNodeDev) * (size_t)std::max(ni, 1);
    const size_t soff = doff + sizeof(double) * (size_t)L;
    const size_t bytes = soff + (has_sets ? sizeof(uint64_t) * kCatSetWords * (size_t)ni : 0);
    tree_pack_.assign(bytes, 0);
    k::TreeNodeDev* np = reinterpret_cast<k::TreeNodeDev*>(tree_pack_.data());
    uint64_t* sets = reinterpret_cast<uint64_t*>(tree_pack_.data() + soff);
    for (int i = 0; i < ni; ++i) {
      const int f = tree.split_feature_inner[(size_t)i];
      const int mb = train_->inner_mapper(f).missing_bin;
      const bool cat = (tree.decision_type[(size_t)i] & kCategoricalMask) != 0;
      const bool set = cat && tree.threshold_bin[(size_t)i] < 0;
      np[i].feature = f;
      np[i].meta = (set ? 0 : (tree.threshold_bin[(size_t)i] & 255)) | ((tree.decision_type[(size_t)i] & kDefaultLeftMask) ? 256 : 0) |
                   (cat && !set ? 512 : 0) | (set ? 1024 : 0) | ((mb + 1) << 16);
      np[i].left = tree.left_child[(size_t)i];
      np[i].right = tree.right_child[(size_t)i];
      if (set) {
        const size_t k = (size_t)tree.threshold[(size_t)i];
        for (int w = 0; w < kCatSetWords; ++w) sets[(size_t)i * kCatSetWords + (size_t)w] = tree.cat_bins[kCatSetWords * k + (size_t)w];
      }
    }
    std::memcpy(tree_pack_.data() + doff, tree.leaf_value.data(), sizeof(double) * (size_t)L);
    if (tree_pack_ != tree_uploaded_) {
      wait_upload();
      tree_host_.resize(bytes);
      tree_dev_.resize(bytes);
      std::memcpy(tree_host_.get(), tree_pack_.data(), bytes);
      LGA_HIP_CHECK(hipMemcpyAsync(tree_dev_.get(), tree_host_.get(), bytes, hipMemcpyHostToDevice, stream_));
      mark_upload();
      tree_uploaded_ = tree_pack_;
    }
    TreeOnDevice td;
    td.nodes = reinterpret_cast<const k::TreeNodeDev*>(tree_dev_.get());
    td.leaf_value = reinterpret_cast<const double*>(tree_dev_.get() + doff);
    td.node_sets = has_sets ? reinterpret_cast<const uint64_t*>(tree_dev_.get() + soff) : nullptr;
    td.n_internal = ni;
    td.num_leaves = L;
    return td;
  }
  void traverse(const TreeOnDevice& td, const DeviceMirror* m, int64_t N, const int32_t* rows, int64_t n, double sign,
                double* scores) {
    k::traverse_add(td.nodes, td.n_internal, td.leaf_value, td.num_leaves, m->bins_fm.get(), N, rows, n, sign, scores,
                    td.node_sets, stream_);
  }

  std::shared_ptr<Dataset> train_;
  Config cfg_;
  Comm* comm_;
  std::unique_ptr<StagingComm> staging_;
  TreeGrower grower_;
  hipStream_t stream_ = nullptr;
  int device_ = 0;                 // the GPU this engine was built on (bind_thread)
  DeviceMirror* mir_ = nullptr;
  std::unique_ptr<HipBackend> backend_;
  int64_t N_ = 0, n_base_ = 0, bag_n_ = 0, oob_n_ = 0;
  int K_ = 1, cls_ = 0;
  int64_t tree_index_ = 0;
  // rows == nullptr lets an elementwise kernel index rows directly (the training set is the whole root)
  const int32_t* base_or_null() const { return train_->used_indices() ? base_rows_.get() : nullptr; }
  // the training rows as a list for the kernels that walk them (the root's row list, the source of a bag): in the
  // engine's own order when it has one (base_rows_ then maps dataset rows to it and is for gathers to the outside)
  const int32_t* all_rows() const { return locality_ ? iter_rows_.get() : base_rows_.get(); }
  const int32_t* ext_ids() const { return locality_ ? order_.ext_of_int.get() : nullptr; }
  static constexpr int64_t kLocalityMinRows = 1 << 20;
  bool locality_ = false;
  std::unique_ptr<DeviceMirror> own_mir_;   // the private, reordered copies (mir_ points here then)
  LocalityOrder order_;
  DeviceBuffer<int32_t> iter_rows_;         // 0 .. n - 1
  const int32_t* bag_ptr_ = nullptr;
  DeviceBuffer<int32_t> base_rows_, bag_rows_;
  static constexpr int64_t kWalkMinRows = 1 << 20;
  bool use_lines_ = false;
  DeviceBuffer<uint8_t> lines_;
  int score_update_ = 0;   // 0 auto, 1 leaf ranges, 2 tree walk (LGA_SCORE_UPDATE=leaf|walk overrides)
  // gradients taken ahead of time (set_iteration_objective): valid while the training scores still are the ones they
  // were taken from - every writer of the training scores moves score_version_ on
  const ObjectiveSpec* iteration_obj_ = nullptr;
  uint64_t score_version_ = 1, grad_version_ = 0;
  bool fuse_gradients_ = std::getenv("LGA_FUSE_GRADIENTS") == nullptr || std::string(std::getenv("LGA_FUSE_GRADIENTS")) != "0";
  bool init_per_class_ = false;
  DeviceBuffer<double> scores_, init_score_dev_, red_dev_, tmp_f64_;
  DeviceBuffer<float> grad_, hess_, tmp_f32_;
  // ranking: per-query tables of the training data (ranking_gradients)
  bool rank_ready_ = false, rank_on_device_ = false;
  int rank_max_query_ = 0;
  DeviceBuffer<int64_t> rank_bounds_;
  DeviceBuffer<double> rank_imd_, rank_gain_, rank_disc_;
  std::map<int, std::unique_ptr<DeviceBuffer<int64_t>>> metric_bounds_;   // per dataset (which)
  DeviceBuffer<double> metric_gain_, metric_disc_;
  size_t metric_disc_len_ = 0;
  DeviceBuffer<int64_t> packed_, leaf_acc_;
  DeviceBuffer<uint16_t> leaf_of_row_;   // leaf of every root row as the last walk for leaf sums left it
  bool leaf_ids_valid_ = false;
  bool fused_walk_ = !(std::getenv("LGA_FUSED_WALK") && std::atoi(std::getenv("LGA_FUSED_WALK")) == 0);
  bool renew_always_ = std::getenv("LGA_RENEW_ALWAYS") && std::atoi(std::getenv("LGA_RENEW_ALWAYS")) != 0;
  int leaf_ids_leaves_ = 0;
  double walk_max_depth_ = std::getenv("LGA_WALK_DEPTH") ? std::atof(std::getenv("LGA_WALK_DEPTH")) : 1e9;   // measured: never worse than the per-leaf row lists
  DeviceBuffer<uint32_t> maxbits_, select_scratch_;
  DeviceBuffer<k::LeafSpan> span_dev_;
  DeviceBuffer<uint64_t> sort_keys_;
  DeviceBuffer<double> pct_values_;
  DeviceBuffer<int32_t> pct_counts_, item_leaf_;
  DeviceBuffer<char> tree_dev_;
  PinnedBuffer<double> red_host_;
  PinnedBuffer<k::LeafSpan> span_host_;
  PinnedBuffer<char> tree_host_;
  std::vector<char> tree_pack_, tree_uploaded_;
  hipEvent_t upload_ev_ = nullptr;
  bool upload_pending_ = false;
  bool has_init_ = false, have_max_ = false;
  std::vector<LeafPiece> leaf_nodes_;
  bool tree_begun_ = false;
  MappedBuffer batch_desc_;      // descriptors of a batched whole-tree / chunk launch
  MappedBuffer chunk_out_, chunk_iters_, chunk_desc_;
  ChunkPlan chunk_plan_;
  k::TreeLayout chunk_lay_{};
  unsigned chunk_seq_ = 0;
  bool chunk_prepared_ = false, chunk_launched_ = false;
  int64_t chunk_iters_total_ = 0;
  std::vector<std::unique_ptr<Valid>> valid_;
};

}  // namespace

std::unique_ptr<Engine> make_hip_engine(std::shared_ptr<Dataset> train, const Config& cfg, Comm* comm,
                                        ParallelMode mode) {
  return std::unique_ptr<Engine>(new HipEngine(train, cfg, comm, mode));
}

}  // namespace lga

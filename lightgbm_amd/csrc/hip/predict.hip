// Batched tree-ensemble prediction on raw feature values (the device side of
// `predict(booster, matrix)`, r/gridsearchCV.R:63,140, and of staged
// prediction with the first k trees, bagging_boosting.ipynb:136).
//
// Trees are packed into 24-byte nodes and staged tile by tile through LDS
// (as many trees per tile as fit 36 KiB, so four workgroups share a CU); a
// workgroup owns 1024 rows (4 per lane, walked side by side, sums in registers)
// and walks every staged tree for them, so each tile is read from L2 once per
// 1024 rows.
#include "hip_engine.hpp"
#include "../gbdt.hpp"
#include "../objective.hpp"
#include <cstring>
#include <type_traits>

namespace lga {

namespace {

struct PNode {
  double threshold;
  int32_t feature;
  int32_t left, right;
  int32_t dtype;
};
static_assert(sizeof(PNode) == 24, "PNode layout");

constexpr int kRowsPerLane = 4;
constexpr int kPredThreads = 256;
constexpr size_t kTileBytes = 36 * 1024;      // four 255-leaf trees; four workgroups (16 waves) share a CU's LDS
constexpr size_t kMaxTileBytes = 120 * 1024;  // a single larger tree may claim up to this much

// Categorical nodes: a single category sits in `threshold` itself; a larger set (dtype bit 6) is a bitset
// over category values in `cat_words`, `threshold` packing its first word (low 32 bits) and its length.
constexpr int kSetNodeBit = 64;
template <bool SETS = true>
__device__ __forceinline__ bool go_left_dev(double v, double thr, int dtype, const uint32_t* __restrict__ cat_words) {
  if (SETS && (dtype & kSetNodeBit)) {
    if (!(v >= 0.0)) return false;
    const long long c = (long long)v;
    const unsigned long long enc = (unsigned long long)thr;
    const unsigned long long first = enc & 0xffffffffull, words = enc >> 32;
    if ((unsigned long long)(c >> 5) >= words) return false;
    return (cat_words[first + (unsigned long long)(c >> 5)] >> (c & 31)) & 1u;
  }
  if (dtype & 1) return v >= 0.0 && (double)(long long)v == thr;
  const int mt = (dtype >> 2) & 3;
  if (v != v && mt != 2) v = 0.0;
  if ((mt == 1 && v > -1e-35 && v < 1e-35) || (mt == 2 && v != v)) return (dtype & 2) != 0;
  return v <= thr;
}

template <typename T>
__global__ __launch_bounds__(kPredThreads) void predict_kernel(const T* __restrict__ X, int64_t nrow, int ncol,
                                                               const PNode* __restrict__ nodes,
                                                               const double* __restrict__ leaves,
                                                               const int32_t* __restrict__ node_off,   // [trees+1]
                                                               const int32_t* __restrict__ leaf_off,   // [trees+1]
                                                               const int32_t* __restrict__ tile_begin, // [tiles+1]
                                                               int n_tiles, int n_trees, int average, int obj_kind,
                                                               double sigmoid, int raw, double* __restrict__ out,
                                                               const uint32_t* __restrict__ cat_words) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t r0 = (int64_t)blockIdx.x * (kPredThreads * kRowsPerLane) + threadIdx.x;
  double sum[kRowsPerLane];
#pragma unroll
  for (int j = 0; j < kRowsPerLane; ++j) sum[j] = 0.0;
  for (int tile = 0; tile < n_tiles; ++tile) {
    const int t0 = tile_begin[tile], t1 = tile_begin[tile + 1];
    const int n0 = node_off[t0], n1 = node_off[t1];
    const int l0 = leaf_off[t0], l1 = leaf_off[t1];
    const int nn = n1 - n0, nl = l1 - l0;
    PNode* sn = reinterpret_cast<PNode*>(smem);
    double* sl = reinterpret_cast<double*>(smem + (size_t)nn * sizeof(PNode));
    __syncthreads();
    {
      // 24-byte nodes as 8-byte words
      const double* src = reinterpret_cast<const double*>(nodes + n0);
      double* dst = reinterpret_cast<double*>(sn);
      for (int i = threadIdx.x; i < nn * 3; i += kPredThreads) dst[i] = src[i];
      for (int i = threadIdx.x; i < nl; i += kPredThreads) sl[i] = leaves[l0 + i];
    }
    __syncthreads();
    // the lane's rows walk a tree side by side: their node / feature loads are independent, so the memory
    // latency of one overlaps with the others'
    const T* x[kRowsPerLane];
    bool live[kRowsPerLane];
#pragma unroll
    for (int j = 0; j < kRowsPerLane; ++j) {
      const int64_t r = r0 + (int64_t)j * kPredThreads;
      live[j] = r < nrow;
      x[j] = X + (live[j] ? r : 0) * (int64_t)ncol;
    }
    for (int t = t0; t < t1; ++t) {
      const int nb = node_off[t] - n0, lb = leaf_off[t] - l0;
      const int tn = node_off[t + 1] - node_off[t];
      int node[kRowsPerLane];
#pragma unroll
      for (int j = 0; j < kRowsPerLane; ++j) node[j] = (tn > 0 && live[j]) ? 0 : ~0;
      bool any = tn > 0;
      while (any) {
        any = false;
#pragma unroll
        for (int j = 0; j < kRowsPerLane; ++j) {
          if (node[j] >= 0) {
            const PNode nd = sn[nb + node[j]];
            node[j] = go_left_dev((double)x[j][nd.feature], nd.threshold, nd.dtype, cat_words) ? nd.left : nd.right;
            any = any || node[j] >= 0;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < kRowsPerLane; ++j)
        if (live[j]) sum[j] += sl[lb + (~node[j])];
    }
  }
#pragma unroll
  for (int j = 0; j < kRowsPerLane; ++j) {
    const int64_t r = r0 + (int64_t)j * kPredThreads;
    if (r >= nrow) continue;
    double s = sum[j];
    if (average && n_trees > 0) s /= (double)n_trees;
    out[r] = raw ? s : transform_score(obj_kind, sigmoid, s);
  }
}


// ---------------------------------------------------------------------------
// Narrow matrices (the row tile fits LDS next to the tree tile): the workgroup first copies its rows into
// LDS transposed - xs[feature][row], so the lanes of a wave, each on its own row, read one feature value
// from 64 consecutive words whatever feature each lane asks for - and packs nodes into 16 bytes (one
// ds_read_b128 per step).  A step then touches LDS only; in the kernel above every step is a load of 64
// different cache lines through the vector cache.
// ---------------------------------------------------------------------------
struct QNode {
  double threshold;
  int16_t feature, left, right, dtype;
};
static_assert(sizeof(QNode) == 16, "QNode layout");
constexpr size_t kQTileBytes = 20 * 1024;   // three 255-leaf trees
constexpr int kQTileTrees = 64;             // trees a tile holds at most (their offsets wait in LDS too)

// SETS: the model has category-set nodes (the common case without them compiles the lookup away)
template <typename T, int R, bool SETS>
__global__ __launch_bounds__(kPredThreads) void predict_lds_kernel(const T* __restrict__ X, int64_t nrow, int ncol,
                                                                   int nfeat,   // columns the model reads (<= ncol)
                                                                   const QNode* __restrict__ nodes,
                                                                   const double* __restrict__ leaves,
                                                                   const int32_t* __restrict__ node_off,
                                                                   const int32_t* __restrict__ leaf_off,
                                                                   const int32_t* __restrict__ tile_begin,
                                                                   int n_tiles, int n_trees, int average, int obj_kind,
                                                                   double sigmoid, int raw, double* __restrict__ out,
                                                                   const uint32_t* __restrict__ cat_words) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_noff[kQTileTrees + 1], s_loff[kQTileTrees + 1];   // the tile's trees: first node / first leaf inside the tile
  constexpr int kRows = kPredThreads * R;
  T* xs = reinterpret_cast<T*>(smem);                                   // [nfeat][kRows]
  char* tile = smem + (((size_t)nfeat * kRows * sizeof(T) + 15) & ~(size_t)15);
  const int64_t base = (int64_t)blockIdx.x * kRows;
  // coalesced read of the row-major block, transposed store
  const int64_t block_vals = (int64_t)min((int64_t)kRows, nrow - base) * ncol;
  for (int64_t i = threadIdx.x; i < block_vals; i += kPredThreads) {
    const int r = (int)(i / ncol), c = (int)(i - (int64_t)r * ncol);
    if (c < nfeat) xs[c * kRows + r] = X[base * ncol + i];
  }
  double sum[R];
  bool live[R];
#pragma unroll
  for (int j = 0; j < R; ++j) { sum[j] = 0.0; live[j] = base + j * kPredThreads + threadIdx.x < nrow; }
  for (int tl = 0; tl < n_tiles; ++tl) {
    const int t0 = tile_begin[tl], t1 = tile_begin[tl + 1];
    const int n0 = node_off[t0], nn = node_off[t1] - n0;
    const int l0 = leaf_off[t0], nl = leaf_off[t1] - l0;
    QNode* sn = reinterpret_cast<QNode*>(tile);
    double* sl = reinterpret_cast<double*>(tile + (size_t)nn * sizeof(QNode));
    __syncthreads();   // the previous tile is done with (first pass: the rows are in place)
    {
      const double2* src = reinterpret_cast<const double2*>(nodes + n0);
      double2* dst = reinterpret_cast<double2*>(sn);
      for (int i = threadIdx.x; i < nn; i += kPredThreads) dst[i] = src[i];
      for (int i = threadIdx.x; i < nl; i += kPredThreads) sl[i] = leaves[l0 + i];
      for (int i = threadIdx.x; i <= t1 - t0; i += kPredThreads) { s_noff[i] = node_off[t0 + i] - n0; s_loff[i] = leaf_off[t0 + i] - l0; }
    }
    __syncthreads();
    // A lane walks the tile's trees one after the other WITHOUT waiting for the other lanes at every leaf: it adds the
    // leaf's value and starts on its next tree at once, so a wave runs as long as its slowest lane needs for all the
    // tile's trees together (sums of depths differ far less between rows than single depths do) instead of the
    // deepest leaf of every tree.  The trees of a row are still added in order: the sums are the same bit for bit.
    int tcur[R], node[R], nb[R], lb[R];
    auto enter = [&](int j) {
      // trees without a split are their one leaf
      while (tcur[j] < t1 && s_noff[tcur[j] - t0 + 1] == s_noff[tcur[j] - t0]) { sum[j] += sl[s_loff[tcur[j] - t0]]; ++tcur[j]; }
      if (tcur[j] < t1) { nb[j] = s_noff[tcur[j] - t0]; lb[j] = s_loff[tcur[j] - t0]; }
      node[j] = 0;
    };
    bool any = false;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      tcur[j] = live[j] ? t0 : t1;
      nb[j] = 0; lb[j] = 0;
      enter(j);
      any = any || tcur[j] < t1;
    }
    while (any) {
      any = false;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        if (tcur[j] < t1) {
          const QNode nd = sn[nb[j] + node[j]];
          const double v = (double)xs[(int)nd.feature * kRows + j * kPredThreads + threadIdx.x];
          node[j] = go_left_dev<SETS>(v, nd.threshold, nd.dtype, cat_words) ? (int)nd.left : (int)nd.right;
          if (node[j] < 0) {
            sum[j] += sl[lb[j] + (~node[j])];
            ++tcur[j];
            enter(j);
          }
          any = any || tcur[j] < t1;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    if (!live[j]) continue;
    double s = sum[j];
    if (average && n_trees > 0) s /= (double)n_trees;
    out[base + j * kPredThreads + threadIdx.x] = raw ? s : transform_score(obj_kind, sigmoid, s);
  }
}

// Wider matrices: 64 rows per workgroup wait in LDS (transposed like above: a wave's 64 lanes are the 64 rows), and the
// four waves walk FOUR DIFFERENT TREES of the tile for them at the same time - wave w takes the tile's trees w, w + 4,
// ... - so a row tile a quarter the size keeps as many walks in flight.  The leaf values pass through LDS and the first
// wave adds them to its rows' sums in tree order: the same sums, bit for bit, as one lane per row walking every tree.
constexpr int kWideRows = 64;
constexpr int kWideTileTrees = 16;   // trees per tile at most (their leaf values wait in LDS: 16 x 64 doubles)
template <typename T, bool SETS>
__global__ __launch_bounds__(kPredThreads) void predict_rows64_kernel(const T* __restrict__ X, int64_t nrow, int ncol, int nfeat,
                                                                      const QNode* __restrict__ nodes,
                                                                      const double* __restrict__ leaves,
                                                                      const int32_t* __restrict__ node_off,
                                                                      const int32_t* __restrict__ leaf_off,
                                                                      const int32_t* __restrict__ tile_begin,
                                                                      int n_tiles, int n_trees, int average, int obj_kind,
                                                                      double sigmoid, int raw, double* __restrict__ out,
                                                                      const uint32_t* __restrict__ cat_words) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_noff[kWideTileTrees + 1], s_loff[kWideTileTrees + 1];
  __shared__ double vals[kWideTileTrees][kWideRows];
  T* xs = reinterpret_cast<T*>(smem);                                   // [nfeat][64]
  char* tile = smem + (((size_t)nfeat * kWideRows * sizeof(T) + 15) & ~(size_t)15);
  const int64_t base = (int64_t)blockIdx.x * kWideRows;
  const int row = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t block_vals = (int64_t)min((int64_t)kWideRows, nrow - base) * ncol;
  for (int64_t i = threadIdx.x; i < block_vals; i += kPredThreads) {
    const int r = (int)(i / ncol), c = (int)(i - (int64_t)r * ncol);
    if (c < nfeat) xs[c * kWideRows + r] = X[base * ncol + i];
  }
  const bool live = base + row < nrow;
  double sum = 0.0;
  for (int tl = 0; tl < n_tiles; ++tl) {
    const int t0 = tile_begin[tl], t1 = tile_begin[tl + 1];
    const int n0 = node_off[t0], nn = node_off[t1] - n0;
    const int l0 = leaf_off[t0], nl = leaf_off[t1] - l0;
    QNode* sn = reinterpret_cast<QNode*>(tile);
    double* sl = reinterpret_cast<double*>(tile + (size_t)nn * sizeof(QNode));
    __syncthreads();   // the previous tile is done with (first pass: the rows are in place)
    {
      const double2* src = reinterpret_cast<const double2*>(nodes + n0);
      double2* dst = reinterpret_cast<double2*>(sn);
      for (int i = threadIdx.x; i < nn; i += kPredThreads) dst[i] = src[i];
      for (int i = threadIdx.x; i < nl; i += kPredThreads) sl[i] = leaves[l0 + i];
      for (int i = threadIdx.x; i <= t1 - t0; i += kPredThreads) { s_noff[i] = node_off[t0 + i] - n0; s_loff[i] = leaf_off[t0 + i] - l0; }
    }
    __syncthreads();
    // this wave's trees of the tile, every lane at its own pace
    int tcur = live ? t0 + wave : t1, node = 0, nb = 0, lb = 0;
    auto enter = [&]() {
      while (tcur < t1 && s_noff[tcur - t0 + 1] == s_noff[tcur - t0]) { vals[tcur - t0][row] = sl[s_loff[tcur - t0]]; tcur += 4; }
      if (tcur < t1) { nb = s_noff[tcur - t0]; lb = s_loff[tcur - t0]; }
      node = 0;
    };
    enter();
    while (tcur < t1) {
      const QNode nd = sn[nb + node];
      const double v = (double)xs[(int)nd.feature * kWideRows + row];
      node = go_left_dev<SETS>(v, nd.threshold, nd.dtype, cat_words) ? (int)nd.left : (int)nd.right;
      if (node < 0) {
        vals[tcur - t0][row] = sl[lb + (~node)];
        tcur += 4;
        enter();
      }
    }
    __syncthreads();
    if (wave == 0 && live)
      for (int t = 0; t < t1 - t0; ++t) sum += vals[t][row];
  }
  if (wave == 0 && live) {
    double s = sum;
    if (average && n_trees > 0) s /= (double)n_trees;
    out[base + row] = raw ? s : transform_score(obj_kind, sigmoid, s);
  }
}

// first parameter (const T* X) of a predictor kernel, for casting the untyped matrix pointer
template <typename T, typename... A>
const T* kernel_arg0(void (*)(const T*, A...));

// LDS bytes of the row tile + tree tile of predict_lds_kernel<T, R>
template <typename T, int R>
size_t lds_kernel_bytes(int nfeat, size_t tile_limit) {
  return (((size_t)nfeat * kPredThreads * R * sizeof(T) + 15) & ~(size_t)15) + tile_limit;
}

// Numerical node: its threshold; categorical node: the one category that goes left, or - for a larger set -
// the place of its bitset in `cat_words` (appended here), flagged in *dtype.
double node_threshold(const Tree& tr, int i, std::vector<uint32_t>* cat_words, int* dtype) {
  *dtype = tr.decision_type[(size_t)i];
  if (!(tr.decision_type[(size_t)i] & kCategoricalMask)) return tr.threshold[(size_t)i];
  const std::vector<int64_t> cats = tr.node_categories(i);
  if (cats.size() == 1) return (double)cats[0];
  const size_t k = (size_t)tr.threshold[(size_t)i];
  const size_t first = cat_words->size(), words = (size_t)(tr.cat_boundaries[k + 1] - tr.cat_boundaries[k]);
  if (first + words >= (1ull << 32) || words >= (1ull << 21))
    throw std::runtime_error("the model's category sets are too large for the device predictor; predict on the host");
  cat_words->insert(cat_words->end(), tr.cat_threshold.begin() + tr.cat_boundaries[k], tr.cat_threshold.begin() + tr.cat_boundaries[k + 1]);
  *dtype |= kSetNodeBit;
  return (double)(((unsigned long long)words << 32) | (unsigned long long)first);
}

}  // namespace

namespace {

// Multiclass: raw scores arrive class-major ([class][row]); one thread per row applies the link function
// and writes the row's num_class values next to each other.
__global__ __launch_bounds__(256) void multiclass_finish_kernel(const double* __restrict__ raw, int64_t n, int K,
                                                                int32_t kind, double sigmoid, int keep_raw,
                                                                double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (keep_raw || (kind != kObjMulticlass && kind != kObjMulticlassOva)) {
    for (int k = 0; k < K; ++k) out[i * K + k] = raw[(int64_t)k * n + i];
    return;
  }
  if (kind == kObjMulticlassOva) {
    for (int k = 0; k < K; ++k) out[i * K + k] = 1.0 / (1.0 + exp(-sigmoid * raw[(int64_t)k * n + i]));
    return;
  }
  double mx = raw[i];
  for (int k = 1; k < K; ++k) mx = fmax(mx, raw[(int64_t)k * n + i]);
  double sum = 0.0;
  for (int k = 0; k < K; ++k) sum += exp(raw[(int64_t)k * n + i] - mx);
  for (int k = 0; k < K; ++k) out[i * K + k] = exp(raw[(int64_t)k * n + i] - mx) / sum;
}

// Scores of ONE class of the model: class k's trees are trees[(iteration) * K + k].
void predict_class(const Booster& b, int K, int k, const void* data, bool on_device, int dtype, int64_t nrow, int ncol,
                   int start_iter, int num_iter, int type, double* out, bool out_on_device) {
  hipStream_t s = current_stream();
  const auto& all_trees = b.trees();
  int total = (int)all_trees.size() / K;
  start_iter = std::max(0, std::min(start_iter, total));
  int n_it = b.used_iterations(start_iter, num_iter);
  // the iteration's tree of this class, under the name the single-output code below uses
  struct ClassTrees {
    const std::vector<Tree>& all;
    int K, k;
    const Tree& operator[](size_t iteration) const { return all[iteration * (size_t)K + (size_t)k]; }
  };
  const ClassTrees trees{all_trees, K, k};
  ObjectiveSpec obj = b.objective_spec();
  const size_t esz = dtype == 0 ? 4 : 8;
  const int64_t chunk = on_device ? nrow : std::min<int64_t>(nrow, std::max<int64_t>(1, (int64_t)((512u << 20) / (esz * (size_t)ncol))));
  DeviceBuffer<char> stage;
  DeviceBuffer<double> d_out;
  if (!on_device) stage.resize((size_t)chunk * (size_t)ncol * esz);
  if (!out_on_device) d_out.resize((size_t)std::max<int64_t>(chunk, 1));
  // Runs `launch(x, n, o)` over the matrix in chunks (host matrices are staged through HBM).
  auto for_each_chunk = [&](auto launch) {
    for (int64_t r0 = 0; r0 < nrow; r0 += chunk) {
      const int64_t n = std::min(chunk, nrow - r0);
      const void* x;
      if (on_device) x = (const char*)data + (size_t)r0 * (size_t)ncol * esz;
      else {
        LGA_HIP_CHECK(hipMemcpyAsync(stage.get(), (const char*)data + (size_t)r0 * (size_t)ncol * esz,
                                     (size_t)n * (size_t)ncol * esz, hipMemcpyHostToDevice, s));
        x = stage.get();
      }
      double* o = out_on_device ? out + r0 : d_out.get();
      launch(x, n, o);
      LGA_HIP_CHECK(hipGetLastError());
      if (!out_on_device)
        LGA_HIP_CHECK(hipMemcpyAsync(out + r0, d_out.get(), sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
      LGA_HIP_CHECK(hipStreamSynchronize(s));
    }
    LGA_HIP_CHECK(hipStreamSynchronize(s));
  };

  // ---- narrow matrices: rows and trees both in LDS
  {
    const int nfeat = b.num_features();
    int max_leaves = 1;
    for (int t = 0; t < n_it; ++t) max_leaves = std::max(max_leaves, trees[(size_t)(start_iter + t)].num_leaves);
    const size_t biggest = (size_t)(max_leaves - 1) * sizeof(QNode) + (size_t)max_leaves * sizeof(double);
    // How many workgroups a CU holds decides the throughput (a walk is a chain of LDS round trips; measured on 500 trees of
    // 255 leaves over 28 columns: three workgroups of one row per lane 84 M rows/s, two workgroups of two rows per lane
    // 63 M, two of one row 65 M): one row per lane, and the tree tile as large as still lets three workgroups - or,
    // with a wide row tile, two, or one - share the CU's 160 KiB.  LGA_PRED_TILE_KB / LGA_PRED_ROWS override both.
    const size_t cu_lds = 160 * 1024, slack = 1536;   // (the kernel's static arrays, allocation granularity)
    const size_t xs1 = (((size_t)nfeat * kPredThreads * esz + 15) & ~(size_t)15);
    const size_t xs64 = (((size_t)nfeat * kWideRows * esz + 15) & ~(size_t)15), slack64 = slack + sizeof(double) * kWideTileTrees * kWideRows;
    // 256 rows per workgroup while THREE such workgroups fit a CU (measured, 300 trees of 255 leaves: 28 columns 98 M
    // rows/s against 77 M with the 64-row kernel, 45 columns 77 against 64, 60 columns - two workgroups - 63 against 78);
    // else 64 rows and a tree per wave (predict_rows64_kernel) while that fits (200 columns: 43 M rows/s against 18.5 M
    // with the feature values in global memory); else predict_kernel.  LGA_PRED_KERNEL=rows64 forces the second.
    int R = xs1 + biggest + slack <= cu_lds / 3 ? 1 : 0;
    if (const char* e = std::getenv("LGA_PRED_KERNEL")) { if (std::string(e) == "rows64") R = 0; }
    const bool rows64 = R == 0 && xs64 + biggest + slack64 <= cu_lds;
    if (R == 0 && !rows64 && xs1 + biggest + slack <= cu_lds) R = 1;
    size_t q_limit = biggest;
    int max_tile_trees = kQTileTrees;
    if (R > 0) {
      int wgs = 3;
      while (wgs > 1 && xs1 + biggest + slack > cu_lds / (size_t)wgs) --wgs;
      q_limit = std::max(biggest, std::min<size_t>(cu_lds / (size_t)wgs - xs1 - slack, 2 * kQTileBytes));
    } else if (rows64) {
      int wgs = 3;
      while (wgs > 1 && xs64 + biggest + slack64 > cu_lds / (size_t)wgs) --wgs;
      q_limit = std::max(biggest, std::min<size_t>(cu_lds / (size_t)wgs - xs64 - slack64, 2 * kQTileBytes));
      max_tile_trees = kWideTileTrees;
    }
    if (const char* e = std::getenv("LGA_PRED_ROWS")) {
      if (std::atoi(e) == 2 && 2 * xs1 + biggest + slack <= cu_lds) { R = 2; q_limit = std::max(biggest, kQTileBytes); }
    }
    if (const char* e = std::getenv("LGA_PRED_TILE_KB")) q_limit = std::max(biggest, (size_t)std::max(4, std::atoi(e)) * 1024);
    const bool fits = R > 0 ? q_limit + (size_t)R * xs1 + slack <= cu_lds : (rows64 && q_limit + xs64 + slack64 <= cu_lds);
    if (fits && max_leaves <= 32768 && nfeat <= 32767 && n_it > 0) {
      std::vector<QNode> nodes;
      std::vector<uint32_t> cat_words;
      std::vector<double> leaves;
      std::vector<int32_t> node_off(1, 0), leaf_off(1, 0), tile_begin(1, 0);
      size_t tile_bytes = 0;
      for (int t = 0; t < n_it; ++t) {
        const Tree& tr = trees[(size_t)(start_iter + t)];
        const int ni = tr.num_leaves - 1;
        const size_t bytes = (size_t)ni * sizeof(QNode) + (size_t)tr.num_leaves * sizeof(double);
        if (tile_bytes + bytes > q_limit || t - tile_begin.back() >= max_tile_trees) { tile_begin.push_back(t); tile_bytes = 0; }
        tile_bytes += bytes;
        for (int i = 0; i < ni; ++i) {
          QNode n;
          int dt;
          n.threshold = node_threshold(tr, i, &cat_words, &dt);
          n.feature = (int16_t)tr.split_feature[(size_t)i];
          n.left = (int16_t)tr.left_child[(size_t)i];
          n.right = (int16_t)tr.right_child[(size_t)i];
          n.dtype = (int16_t)dt;
          nodes.push_back(n);
        }
        for (int l = 0; l < tr.num_leaves; ++l) leaves.push_back(tr.leaf_value[(size_t)l]);
        node_off.push_back((int32_t)nodes.size());
        leaf_off.push_back((int32_t)leaves.size());
      }
      tile_begin.push_back(n_it);
      const int n_tiles = (int)tile_begin.size() - 1;
      DeviceBuffer<QNode> d_nodes(std::max<size_t>(nodes.size(), 1));
      DeviceBuffer<double> d_leaves(leaves.size());
      DeviceBuffer<int32_t> d_noff(node_off.size()), d_loff(leaf_off.size()), d_tb(tile_begin.size());
      if (!nodes.empty()) d_nodes.upload(nodes.data(), nodes.size(), s);
      d_leaves.upload(leaves.data(), leaves.size(), s);
      d_noff.upload(node_off.data(), node_off.size(), s);
      d_loff.upload(leaf_off.data(), leaf_off.size(), s);
      d_tb.upload(tile_begin.data(), tile_begin.size(), s);
      DeviceBuffer<uint32_t> d_cat(std::max<size_t>(cat_words.size(), 1));
      if (!cat_words.empty()) d_cat.upload(cat_words.data(), cat_words.size(), s);
      const int avg = b.average_output() ? 1 : 0, raw = type == kPredictRaw ? 1 : 0;
      auto run = [&](auto kernel, size_t lds) {
        LGA_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for_each_chunk([&](const void* x, int64_t n, double* o) {
          const int rows_per_wg = R > 0 ? kPredThreads * R : kWideRows;
          dim3 grid((unsigned)((n + rows_per_wg - 1) / rows_per_wg));
          using XT = typename std::remove_pointer<decltype(kernel_arg0(kernel))>::type;
          hipLaunchKernelGGL(kernel, grid, dim3(kPredThreads), lds, s, (XT*)x, n, ncol, nfeat, d_nodes.get(),
                             d_leaves.get(), d_noff.get(), d_loff.get(), d_tb.get(), n_tiles, n_it, avg, obj.kind,
                             obj.sigmoid, raw, o, d_cat.get());
        });
      };
      const bool sets = !cat_words.empty();
      if (R == 0) {
        const size_t lds = xs64 + q_limit;
        if (dtype == 0) { if (sets) run(predict_rows64_kernel<float, true>, lds); else run(predict_rows64_kernel<float, false>, lds); }
        else { if (sets) run(predict_rows64_kernel<double, true>, lds); else run(predict_rows64_kernel<double, false>, lds); }
      } else if (dtype == 0 && R == 2) {
        if (sets) run(predict_lds_kernel<float, 2, true>, lds_kernel_bytes<float, 2>(nfeat, q_limit));
        else run(predict_lds_kernel<float, 2, false>, lds_kernel_bytes<float, 2>(nfeat, q_limit));
      } else if (dtype == 0) {
        if (sets) run(predict_lds_kernel<float, 1, true>, lds_kernel_bytes<float, 1>(nfeat, q_limit));
        else run(predict_lds_kernel<float, 1, false>, lds_kernel_bytes<float, 1>(nfeat, q_limit));
      } else if (R == 2) {
        if (sets) run(predict_lds_kernel<double, 2, true>, lds_kernel_bytes<double, 2>(nfeat, q_limit));
        else run(predict_lds_kernel<double, 2, false>, lds_kernel_bytes<double, 2>(nfeat, q_limit));
      } else {
        if (sets) run(predict_lds_kernel<double, 1, true>, lds_kernel_bytes<double, 1>(nfeat, q_limit));
        else run(predict_lds_kernel<double, 1, false>, lds_kernel_bytes<double, 1>(nfeat, q_limit));
      }
      return;
    }
  }

  // ---- wide matrices: trees in LDS, feature values through the vector cache
  std::vector<PNode> nodes;
  std::vector<uint32_t> cat_words;
  std::vector<double> leaves;
  std::vector<int32_t> node_off(1, 0), leaf_off(1, 0), tile_begin(1, 0);
  size_t tile_bytes = 0, tile_limit = kTileBytes;
  for (int t = 0; t < n_it; ++t) {
    const Tree& tr = trees[(size_t)(start_iter + t)];
    tile_limit = std::max(tile_limit, (size_t)(tr.num_leaves - 1) * sizeof(PNode) + (size_t)tr.num_leaves * sizeof(double));
  }
  if (tile_limit > kMaxTileBytes) throw std::runtime_error("a tree is too large for the device predictor's LDS tile");
  for (int t = 0; t < n_it; ++t) {
    const Tree& tr = trees[(size_t)(start_iter + t)];
    int ni = tr.num_leaves - 1;
    size_t bytes = (size_t)ni * sizeof(PNode) + (size_t)tr.num_leaves * sizeof(double);
    if (tile_bytes + bytes > tile_limit) { tile_begin.push_back(t); tile_bytes = 0; }
    tile_bytes += bytes;
    for (int i = 0; i < ni; ++i) {
      PNode n;
      int dt;
      n.threshold = node_threshold(tr, i, &cat_words, &dt);
      n.feature = tr.split_feature[(size_t)i];
      n.left = tr.left_child[(size_t)i];
      n.right = tr.right_child[(size_t)i];
      n.dtype = dt;
      nodes.push_back(n);
    }
    for (int l = 0; l < tr.num_leaves; ++l) leaves.push_back(tr.leaf_value[(size_t)l]);
    node_off.push_back((int32_t)nodes.size());
    leaf_off.push_back((int32_t)leaves.size());
  }
  tile_begin.push_back(n_it);
  int n_tiles = (int)tile_begin.size() - 1;
  if (n_it == 0) n_tiles = 0;
  DeviceBuffer<PNode> d_nodes(std::max<size_t>(nodes.size(), 1));
  DeviceBuffer<double> d_leaves(std::max<size_t>(leaves.size(), 1));
  DeviceBuffer<int32_t> d_noff(node_off.size()), d_loff(leaf_off.size()), d_tb(tile_begin.size());
  if (!nodes.empty()) d_nodes.upload(nodes.data(), nodes.size(), s);
  if (!leaves.empty()) d_leaves.upload(leaves.data(), leaves.size(), s);
  d_noff.upload(node_off.data(), node_off.size(), s);
  d_loff.upload(leaf_off.data(), leaf_off.size(), s);
  d_tb.upload(tile_begin.data(), tile_begin.size(), s);
  DeviceBuffer<uint32_t> d_cat(std::max<size_t>(cat_words.size(), 1));
  if (!cat_words.empty()) d_cat.upload(cat_words.data(), cat_words.size(), s);
  static bool attr_set[2] = {false, false};
  if (!attr_set[dtype]) {
    if (dtype == 0) LGA_HIP_CHECK(hipFuncSetAttribute((const void*)predict_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxTileBytes));
    else LGA_HIP_CHECK(hipFuncSetAttribute((const void*)predict_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxTileBytes));
    attr_set[dtype] = true;
  }
  for_each_chunk([&](const void* x, int64_t n, double* o) {
    dim3 grid((unsigned)((n + kPredThreads * kRowsPerLane - 1) / (kPredThreads * kRowsPerLane)));
    if (dtype == 0)
      hipLaunchKernelGGL(predict_kernel<float>, grid, dim3(kPredThreads), tile_limit, s, (const float*)x, n, ncol,
                         d_nodes.get(), d_leaves.get(), d_noff.get(), d_loff.get(), d_tb.get(), n_tiles, n_it,
                         b.average_output() ? 1 : 0, obj.kind, obj.sigmoid, type == kPredictRaw ? 1 : 0, o, d_cat.get());
    else
      hipLaunchKernelGGL(predict_kernel<double>, grid, dim3(kPredThreads), tile_limit, s, (const double*)x, n, ncol,
                         d_nodes.get(), d_leaves.get(), d_noff.get(), d_loff.get(), d_tb.get(), n_tiles, n_it,
                         b.average_output() ? 1 : 0, obj.kind, obj.sigmoid, type == kPredictRaw ? 1 : 0, o, d_cat.get());
  });
}

}  // namespace

// out: one value per row; num_class values per row (row-major) for multiclass models.
void hip_predict(const Booster& b, const void* data, bool on_device, int dtype, int64_t nrow, int ncol, int start_iter,
                 int num_iter, int type, double* out, bool out_on_device) {
  if (ncol < b.num_features())
    throw std::runtime_error("predict: the data has " + std::to_string(ncol) + " features, the model needs " +
                             std::to_string(b.num_features()));
  if (type == kPredictLeaf) throw std::runtime_error("device prediction does not produce leaf indices");
  const int K = b.num_class();
  if (K == 1) {
    predict_class(b, 1, 0, data, on_device, dtype, nrow, ncol, start_iter, num_iter, type, out, out_on_device);
    return;
  }
  // One pass per class over row blocks that stay on the device between the passes, then the link function.
  hipStream_t s = current_stream();
  const ObjectiveSpec obj = b.objective_spec();
  const size_t esz = dtype == 0 ? 4 : 8;
  const int64_t block = std::min<int64_t>(std::max<int64_t>(nrow, 1),
                                          std::max<int64_t>(1, (int64_t)((1024ull << 20) / (esz * (size_t)ncol))));
  DeviceBuffer<char> stage;
  DeviceBuffer<double> raw((size_t)block * (size_t)K), fin;
  if (!on_device) stage.resize((size_t)block * (size_t)ncol * esz);
  if (!out_on_device) fin.resize((size_t)block * (size_t)K);
  for (int64_t r0 = 0; r0 < nrow; r0 += block) {
    const int64_t n = std::min(block, nrow - r0);
    const char* src = (const char*)data + (size_t)r0 * (size_t)ncol * esz;
    const void* x = src;
    if (!on_device) {
      LGA_HIP_CHECK(hipMemcpyAsync(stage.get(), src, (size_t)n * (size_t)ncol * esz, hipMemcpyHostToDevice, s));
      x = stage.get();
    }
    for (int k = 0; k < K; ++k)
      predict_class(b, K, k, x, true, dtype, n, ncol, start_iter, num_iter, kPredictRaw, raw.get() + (size_t)k * (size_t)n, true);
    double* o = out_on_device ? out + r0 * K : fin.get();
    hipLaunchKernelGGL(multiclass_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, raw.get(), n, K,
                       obj.kind, obj.sigmoid, type == kPredictRaw ? 1 : 0, o);
    LGA_HIP_CHECK(hipGetLastError());
    if (!out_on_device)
      LGA_HIP_CHECK(hipMemcpyAsync(out + r0 * K, fin.get(), sizeof(double) * (size_t)n * (size_t)K, hipMemcpyDeviceToHost, s));
    LGA_HIP_CHECK(hipStreamSynchronize(s));
  }
}

}  // namespace lga

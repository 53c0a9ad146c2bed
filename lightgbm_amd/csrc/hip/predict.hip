// Batched tree-ensemble prediction on raw feature values (the device side of
// `predict(booster, matrix)`, r/gridsearchCV.R:63,140, and of staged
// prediction with the first k trees, bagging_boosting.ipynb:136).
//
// Trees are packed into 24-byte nodes and staged tile by tile through LDS
// (as many trees per tile as fit ~120 KiB); a workgroup owns 1024 rows (4 per
// lane, sums in registers) and walks every staged tree for them, so each tile
// is read from L2 once per 1024 rows.
#include "hip_engine.hpp"
#include "../gbdt.hpp"
#include "../objective.hpp"
#include <cstring>

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
constexpr size_t kTileBytes = 120 * 1024;

__device__ __forceinline__ bool go_left_dev(double v, double thr, int dtype) {
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
                                                               double sigmoid, int raw, double* __restrict__ out) {
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
#pragma unroll
    for (int j = 0; j < kRowsPerLane; ++j) {
      const int64_t r = r0 + (int64_t)j * kPredThreads;
      if (r >= nrow) continue;
      const T* x = X + r * (int64_t)ncol;
      double s = 0.0;
      for (int t = t0; t < t1; ++t) {
        const int nb = node_off[t] - n0, lb = leaf_off[t] - l0;
        const int tn = node_off[t + 1] - node_off[t];
        int node = tn > 0 ? 0 : ~0;
        while (node >= 0) {
          const PNode nd = sn[nb + node];
          node = go_left_dev((double)x[nd.feature], nd.threshold, nd.dtype) ? nd.left : nd.right;
        }
        s += sl[lb + (~node)];
      }
      sum[j] += s;
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

}  // namespace

void hip_predict(const Booster& b, const void* data, bool on_device, int dtype, int64_t nrow, int ncol, int start_iter,
                 int num_iter, int type, double* out, bool out_on_device) {
  if (ncol < b.num_features())
    throw std::runtime_error("predict: the data has " + std::to_string(ncol) + " features, the model needs " +
                             std::to_string(b.num_features()));
  if (type == kPredictLeaf) throw std::runtime_error("device prediction does not produce leaf indices");
  hipStream_t s = current_stream();
  if (b.num_class() != 1) throw std::runtime_error("the device predictor serves single-output models; multiclass models predict on the host");
  const auto& trees = b.trees();
  int total = (int)trees.size();
  start_iter = std::max(0, std::min(start_iter, total));
  int n_it = b.used_iterations(start_iter, num_iter);
  // pack the model
  std::vector<PNode> nodes;
  std::vector<double> leaves;
  std::vector<int32_t> node_off(1, 0), leaf_off(1, 0), tile_begin(1, 0);
  size_t tile_bytes = 0;
  for (int t = 0; t < n_it; ++t) {
    const Tree& tr = trees[(size_t)(start_iter + t)];
    int ni = tr.num_leaves - 1;
    size_t bytes = (size_t)ni * sizeof(PNode) + (size_t)tr.num_leaves * sizeof(double);
    if (bytes > kTileBytes) throw std::runtime_error("a tree is too large for the device predictor's LDS tile");
    if (tile_bytes + bytes > kTileBytes) { tile_begin.push_back(t); tile_bytes = 0; }
    tile_bytes += bytes;
    for (int i = 0; i < ni; ++i) {
      PNode n;
      n.threshold = tr.threshold[(size_t)i];
      n.feature = tr.split_feature[(size_t)i];
      n.left = tr.left_child[(size_t)i];
      n.right = tr.right_child[(size_t)i];
      n.dtype = tr.decision_type[(size_t)i];
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
  ObjectiveSpec obj = b.objective_spec();
  const size_t esz = dtype == 0 ? 4 : 8;
  int64_t chunk = on_device ? nrow : std::min<int64_t>(nrow, std::max<int64_t>(1, (int64_t)((512u << 20) / (esz * (size_t)ncol))));
  DeviceBuffer<char> stage;
  DeviceBuffer<double> d_out;
  if (!on_device) stage.resize((size_t)chunk * (size_t)ncol * esz);
  if (!out_on_device) d_out.resize((size_t)std::max<int64_t>(chunk, 1));
  static bool attr_set[2] = {false, false};
  if (!attr_set[dtype]) {
    if (dtype == 0) LGA_HIP_CHECK(hipFuncSetAttribute((const void*)predict_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileBytes));
    else LGA_HIP_CHECK(hipFuncSetAttribute((const void*)predict_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileBytes));
    attr_set[dtype] = true;
  }
  for (int64_t r0 = 0; r0 < nrow; r0 += chunk) {
    int64_t n = std::min(chunk, nrow - r0);
    const void* x;
    if (on_device) x = (const char*)data + (size_t)r0 * (size_t)ncol * esz;
    else {
      LGA_HIP_CHECK(hipMemcpyAsync(stage.get(), (const char*)data + (size_t)r0 * (size_t)ncol * esz,
                                   (size_t)n * (size_t)ncol * esz, hipMemcpyHostToDevice, s));
      x = stage.get();
    }
    double* o = out_on_device ? out + r0 : d_out.get();
    dim3 grid((unsigned)((n + kPredThreads * kRowsPerLane - 1) / (kPredThreads * kRowsPerLane)));
    if (dtype == 0)
      hipLaunchKernelGGL(predict_kernel<float>, grid, dim3(kPredThreads), kTileBytes, s, (const float*)x, n, ncol,
                         d_nodes.get(), d_leaves.get(), d_noff.get(), d_loff.get(), d_tb.get(), n_tiles, n_it,
                         b.average_output() ? 1 : 0, obj.kind, obj.sigmoid, type == kPredictRaw ? 1 : 0, o);
    else
      hipLaunchKernelGGL(predict_kernel<double>, grid, dim3(kPredThreads), kTileBytes, s, (const double*)x, n, ncol,
                         d_nodes.get(), d_leaves.get(), d_noff.get(), d_loff.get(), d_tb.get(), n_tiles, n_it,
                         b.average_output() ? 1 : 0, obj.kind, obj.sigmoid, type == kPredictRaw ? 1 : 0, o);
    LGA_HIP_CHECK(hipGetLastError());
    if (!out_on_device)
      LGA_HIP_CHECK(hipMemcpyAsync(out + r0, d_out.get(), sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    LGA_HIP_CHECK(hipStreamSynchronize(s));
  }
  LGA_HIP_CHECK(hipStreamSynchronize(s));
}

}  // namespace lga

// Locality order of the training rows (device engine, private to one booster).
//
// A leaf deep in a tree holds every 100th or 1000th row of the table; while rows are stored in the caller's order its
// rows are spread evenly over the whole table and every access to one of them - the split feature's bin byte, the
// row's 128-byte line - pays a memory sector of its own.  Rows that fall into the same leaves are rows that agree on
// the bins of the features the trees split on, so the engine keeps its PRIVATE copies of the binned matrices (288 GB
// of HBM make a second copy free) in the order of a coarse key over the few features whose bins explain most of the
// label: rows that agree on those features' octiles become neighbours, and the rows of a deep leaf come in runs.
// Nothing a tree is made of depends on the order of the rows (histograms and leaf sums are integer sums), so the trees
// are the ones the caller's order gives, bit for bit; everything that names a row to the outside (scores, custom
// gradients, initial scores, the random streams of bagging / GOSS / stochastic rounding, which are keyed by row) goes
// through the two index lists built here.  Measured on the 10.5 M x 28 benchmark: 3.64 -> 3.2 ms per tree.
#include "hip_engine.hpp"
#include "kernels.hpp"
#include "locality.hpp"
#include <algorithm>
#include <cmath>
#include <vector>

namespace lga {
namespace {

constexpr int kLevels = 8;          // a key feature contributes its bin's octile: 3 bits
constexpr int kLevelBits = 3;
constexpr int kMaxKeyFeatures = 6;  // 18-bit keys: 262144 groups

// Per feature and octile: rows and label sum (doubles; the order of the adds only decides ties between features).
__global__ __launch_bounds__(256) void octile_label_stats_kernel(const uint4* __restrict__ recs, const float* __restrict__ label,
                                                                 int64_t N, int F, const int32_t* __restrict__ num_bin,
                                                                 double* __restrict__ out) {
  __shared__ double acc[kTileFeatures * kLevels * 2];
  __shared__ int nb[kTileFeatures];
  for (int i = threadIdx.x; i < kTileFeatures * kLevels * 2; i += blockDim.x) acc[i] = 0.0;
  for (int i = threadIdx.x; i < kTileFeatures; i += blockDim.x) nb[i] = i < F ? max(num_bin[i], 1) : 1;
  __syncthreads();
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
    const uint4 a = recs[r * 2], b = recs[r * 2 + 1];
    const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const double y = (double)label[r];
#pragma unroll
    for (int f = 0; f < kTileFeatures; ++f) {
      if (f < F) {
        const int bin = (int)((w[f >> 2] >> ((f & 3) * 8)) & 255u);
        const int lv = min(kLevels - 1, bin * kLevels / nb[f]);
        atomicAdd(&acc[(f * kLevels + lv) * 2], 1.0);
        atomicAdd(&acc[(f * kLevels + lv) * 2 + 1], y);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < F * kLevels * 2; i += blockDim.x)
    if (acc[i] != 0.0) atomicAdd(&out[i], acc[i]);
}

struct KeySpec {
  int32_t n;
  int32_t feature[kMaxKeyFeatures];
  int32_t num_bin[kMaxKeyFeatures];
};
__device__ __forceinline__ int row_key(const uint4* __restrict__ recs, int64_t r, const KeySpec& ks) {
  const uint4 a = recs[r * 2], b = recs[r * 2 + 1];
  const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  int key = 0;
  for (int k = 0; k < ks.n; ++k) {
    const int f = ks.feature[k];
    unsigned d = w[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) d = (f >> 2) == i ? w[i] : d;
    const int bin = (int)((d >> ((f & 3) * 8)) & 255u);
    key = (key << kLevelBits) | min(kLevels - 1, bin * kLevels / ks.num_bin[k]);
  }
  return key;
}
__global__ __launch_bounds__(256) void key_count_kernel(const uint4* __restrict__ recs, int64_t N, KeySpec ks, int32_t* __restrict__ key_of,
                                                        unsigned* __restrict__ counts) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
    const int key = row_key(recs, r, ks);
    key_of[r] = key;
    atomicAdd(&counts[key], 1u);
  }
}
// One workgroup: counts -> exclusive prefix sums, in place (n_keys is a multiple of 1024).
__global__ __launch_bounds__(1024) void key_offsets_kernel(unsigned* __restrict__ counts, int n_keys) {
  __shared__ unsigned part[1024];
  const int per = n_keys / 1024;
  unsigned sum = 0;
  for (int i = 0; i < per; ++i) sum += counts[threadIdx.x * per + i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const unsigned v = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned run = part[threadIdx.x] - sum;
  for (int i = 0; i < per; ++i) {
    const unsigned c = counts[threadIdx.x * per + i];
    counts[threadIdx.x * per + i] = run;
    run += c;
  }
}
// Rows take their places inside their key's range in the order they arrive (any order inside a group will do).
__global__ __launch_bounds__(256) void key_place_kernel(const int32_t* __restrict__ key_of, int64_t N, unsigned* __restrict__ cursor,
                                                        int32_t* __restrict__ ext_of_int, int32_t* __restrict__ int_of_ext) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
    const unsigned pos = atomicAdd(&cursor[key_of[r]], 1u);
    ext_of_int[pos] = (int32_t)r;
    int_of_ext[r] = (int32_t)pos;
  }
}
// The private copies: record j = the caller's record ext_of_int[j]; the feature-major columns are written from the
// records on the way (consecutive threads = consecutive rows: every column store is coalesced).
__global__ __launch_bounds__(256) void permute_rows_kernel(const uint4* __restrict__ recs, const float* __restrict__ label,
                                                           const float* __restrict__ weight, const int32_t* __restrict__ ext_of_int,
                                                           int64_t N, int F, uint4* __restrict__ recs_out, uint8_t* __restrict__ fm_out,
                                                           float* __restrict__ label_out, float* __restrict__ weight_out) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = ext_of_int[j];
    const uint4 a = recs[r * 2], b = recs[r * 2 + 1];
    recs_out[j * 2] = a;
    recs_out[j * 2 + 1] = b;
    const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int f = 0; f < kTileFeatures; ++f)
      if (f < F) fm_out[(int64_t)f * N + j] = (uint8_t)((w[f >> 2] >> ((f & 3) * 8)) & 255u);
    label_out[j] = label[r];
    if (weight) weight_out[j] = weight[r];
  }
}
__global__ __launch_bounds__(256) void gather_rows_f64_kernel(const double* __restrict__ src, const int32_t* __restrict__ ext_of_int,
                                                              int64_t N, double* __restrict__ dst) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += (int64_t)gridDim.x * blockDim.x) dst[j] = src[ext_of_int[j]];
}

inline int grid_of(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192)); }

}  // namespace

bool build_locality_mirror(const DeviceMirror& src, const std::vector<FeatureMeta>& meta, DeviceMirror* dst, LocalityOrder* ord,
                           hipStream_t s) {
  const int64_t N = src.rows;
  const int F = src.features;
  if (src.tiles != 1 || F < 1 || F > kTileFeatures || N < 2 || N > (int64_t)0x7fffffff) return false;
  const uint4* recs = reinterpret_cast<const uint4*>(src.bins_rt.get());
  // 1. which features' octiles explain the label best?
  std::vector<int32_t> nb((size_t)F);
  for (int f = 0; f < F; ++f) nb[(size_t)f] = std::max(1, meta[(size_t)f].num_bin);
  DeviceBuffer<int32_t> nb_dev;
  nb_dev.upload(nb.data(), nb.size(), s);
  DeviceBuffer<double> stats_dev((size_t)F * kLevels * 2);
  stats_dev.zero(s);
  hipLaunchKernelGGL(octile_label_stats_kernel, dim3(std::min(grid_of(N), 1024)), dim3(256), 0, s, recs, src.label.get(), N, F, nb_dev.get(),
                     stats_dev.get());
  LGA_HIP_CHECK(hipGetLastError());
  std::vector<double> stats((size_t)F * kLevels * 2);
  stats_dev.download(stats.data(), stats.size(), s);
  LGA_HIP_CHECK(hipStreamSynchronize(s));
  double tot_n = 0.0, tot_y = 0.0;
  for (int lv = 0; lv < kLevels; ++lv) { tot_n += stats[(size_t)lv * 2]; tot_y += stats[(size_t)lv * 2 + 1]; }
  const double mean = tot_n > 0.0 ? tot_y / tot_n : 0.0;
  std::vector<std::pair<double, int>> ranked;
  for (int f = 0; f < F; ++f) {
    if (nb[(size_t)f] < 2) continue;
    double sc = 0.0;
    for (int lv = 0; lv < kLevels; ++lv) {
      const double n = stats[((size_t)f * kLevels + (size_t)lv) * 2], sy = stats[((size_t)f * kLevels + (size_t)lv) * 2 + 1];
      if (n > 0.0) sc += n * (sy / n - mean) * (sy / n - mean);
    }
    ranked.emplace_back(-sc, f);
  }
  if (ranked.empty()) return false;
  std::sort(ranked.begin(), ranked.end());
  KeySpec ks{};
  ks.n = (int32_t)std::min<size_t>(ranked.size(), kMaxKeyFeatures);
  for (int k = 0; k < ks.n; ++k) { ks.feature[k] = ranked[(size_t)k].second; ks.num_bin[k] = nb[(size_t)ks.feature[k]]; }
  // 2. counting sort by key
  const int n_keys = std::max(1024, 1 << (kLevelBits * ks.n));
  DeviceBuffer<int32_t> key_of((size_t)N);
  DeviceBuffer<unsigned> counts((size_t)n_keys);
  counts.zero(s);
  hipLaunchKernelGGL(key_count_kernel, dim3(grid_of(N)), dim3(256), 0, s, recs, N, ks, key_of.get(), counts.get());
  hipLaunchKernelGGL(key_offsets_kernel, dim3(1), dim3(1024), 0, s, counts.get(), n_keys);
  ord->ext_of_int.resize((size_t)N);
  ord->int_of_ext.resize((size_t)N);
  hipLaunchKernelGGL(key_place_kernel, dim3(grid_of(N)), dim3(256), 0, s, key_of.get(), N, counts.get(), ord->ext_of_int.get(),
                     ord->int_of_ext.get());
  LGA_HIP_CHECK(hipGetLastError());
  // 3. the private copies
  dst->rows = N;
  dst->features = F;
  dst->tiles = 1;
  dst->bins_rt.resize((size_t)N * kTileFeatures);
  dst->bins_fm.resize((size_t)F * (size_t)N);
  dst->label.resize((size_t)N);
  dst->has_weight = src.has_weight;
  if (src.has_weight) dst->weight.resize((size_t)N);
  dst->meta_uploaded = true;
  hipLaunchKernelGGL(permute_rows_kernel, dim3(grid_of(N)), dim3(256), 0, s, recs, src.label.get(),
                     src.has_weight ? src.weight.get() : (const float*)nullptr, ord->ext_of_int.get(), N, F,
                     reinterpret_cast<uint4*>(dst->bins_rt.get()), dst->bins_fm.get(), dst->label.get(),
                     src.has_weight ? dst->weight.get() : (float*)nullptr);
  LGA_HIP_CHECK(hipGetLastError());
  LGA_HIP_CHECK(hipStreamSynchronize(s));   // the scratch buffers above go out of scope
  return true;
}

void gather_rows_f64(const double* src, const int32_t* ext_of_int, int64_t n, double* dst, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_rows_f64_kernel, dim3(grid_of(n)), dim3(256), 0, s, src, ext_of_int, n, dst);
  LGA_HIP_CHECK(hipGetLastError());
}

}  // namespace lga

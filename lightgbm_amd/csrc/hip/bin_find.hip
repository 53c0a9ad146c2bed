// Quantile bin construction on the device (used when the raw matrix already
// lives in HBM): per feature, sort the row sample, run-length encode it into
// distinct values + counts, and cut it into <= max_bin bins.
//
//   1. sample_to_keys      sample [ns][F] -> order-preserving 64-bit keys [F][npad], NaN / padding = max key
//   2. bitonic sort        (sort.hip)
//   3. rle_and_cut         one workgroup per feature: head flags + block scan -> distinct[], counts[], then
//                          prefix sums; lane 0 runs the cut-to-cut form of the host BinMapper's greedy routine
//                          (binning_core.hpp: greedy_bounds_jump, checked equal to the value-by-value form),
//                          so host and device boundaries are bit-identical
#include "hip_engine.hpp"
#include "sort.hpp"
#include "../binning_core.hpp"
#include <algorithm>
#include <cstring>

namespace lga {

namespace {

__device__ __forceinline__ uint64_t key_of(double v) {
  if (v != v) return ~0ull;
  uint64_t b = (uint64_t)__double_as_longlong(v);
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double value_of(uint64_t k) {
  uint64_t b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}
// key(+inf); anything above is NaN or padding
constexpr uint64_t kMaxValueKey = 0xfff0000000000000ull;

template <typename T>
__global__ __launch_bounds__(256) void sample_to_keys_kernel(const T* __restrict__ sample, int64_t ns, int ncol,
                                                             int f0, int nf, int64_t npad,
                                                             uint64_t* __restrict__ keys) {
  const int f = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t k = ~0ull;
    if (i < ns) {
      double v = (double)sample[i * ncol + f0 + f];
      if (v == 0.0) v = 0.0;   // -0.0 and +0.0 are one value
      k = key_of(v);
    }
    keys[(int64_t)f * npad + i] = k;
  }
}

struct FeatureBins {      // result record per feature
  int32_t n_bounds;       // inclusive upper bounds written (last = +inf)
  int32_t n_nan;          // NaN rows in the sample
  int32_t n_distinct;
  int32_t pad;
  double min_val, max_val;
};

__global__ __launch_bounds__(256) void rle_and_cut_kernel(const uint64_t* __restrict__ keys, int64_t npad, int64_t ns,
                                                          double* __restrict__ distinct, int32_t* __restrict__ counts,
                                                          int64_t* __restrict__ prefix, int64_t* __restrict__ small_pref,
                                                          int32_t* __restrict__ next_big,
                                                          int max_bin, int min_data_in_bin, int use_missing,
                                                          double* __restrict__ bounds, FeatureBins* __restrict__ out) {
  __shared__ int s_cnt[256];
  __shared__ int s_total;
  __shared__ long long s_sum[256], s_small[256];
  __shared__ int s_first_big[256];
  const int f = blockIdx.x, t = threadIdx.x;
  const uint64_t* k = keys + (int64_t)f * npad;
  double* dv = distinct + (int64_t)f * ns;
  int32_t* dc = counts + (int64_t)f * ns;
  // valid (non-NaN) prefix length: keys are sorted, NaN / padding sit at the end
  int local = 0;
  for (int64_t i = t; i < ns; i += 256) local += (k[i] <= kMaxValueKey) ? 1 : 0;
  s_cnt[t] = local;
  __syncthreads();
  if (t == 0) { int s = 0; for (int i = 0; i < 256; ++i) s += s_cnt[i]; s_total = s; }
  __syncthreads();
  const int64_t nv = s_total;
  __syncthreads();
  // heads per contiguous segment
  const int64_t seg = (nv + 255) / 256;
  const int64_t lo = (int64_t)t * seg, hi = lo + seg < nv ? lo + seg : nv;
  int heads = 0;
  for (int64_t i = lo; i < hi; ++i) heads += (i == 0 || k[i] != k[i - 1]) ? 1 : 0;
  s_cnt[t] = heads;
  __syncthreads();
  if (t == 0) { int s = 0; for (int i = 0; i < 256; ++i) { int c = s_cnt[i]; s_cnt[i] = s; s += c; } s_total = s; }
  __syncthreads();
  int pos = s_cnt[t];
  const int nd = s_total;
  // distinct values + the index of each head (stored in counts[] for now)
  for (int64_t i = lo; i < hi; ++i) {
    if (i == 0 || k[i] != k[i - 1]) { dv[pos] = value_of(k[i]); dc[pos] = (int32_t)i; ++pos; }
  }
  __threadfence_block();
  __syncthreads();
  // head positions -> run lengths (read the neighbour before anyone overwrites: two phases)
  for (int base = 0; base < nd; base += 256) {
    const int d = base + t;
    int32_t c = 0;
    if (d < nd) c = (d + 1 < nd ? dc[d + 1] : (int32_t)nv) - dc[d];
    __syncthreads();
    if (d < nd) dc[d] = c;
    __syncthreads();
  }
  const int n_nan = (int)(ns - nv);
  int value_bins = max_bin - ((use_missing && n_nan > 0) ? 1 : 0);
  if (value_bins < 1) value_bins = 1;
  int64_t* pf = prefix + (int64_t)f * (ns + 1);
  int64_t* sp = small_pref + (int64_t)f * (ns + 1);
  int32_t* nbg = next_big + (int64_t)f * (ns + 1);
  if (nd > value_bins) {
    // prefix sums of the counts (all values / the values below the mean bin size) and, for every
    // position, the next value that fills a bin on its own: inputs of the cut-to-cut search
    double mean = (double)nv / (double)value_bins;
    if (mean < (double)min_data_in_bin) mean = (double)min_data_in_bin;
    const int seg2 = (nd + 255) / 256;
    const int lo2 = min(t * seg2, nd), hi2 = min(lo2 + seg2, nd);
    long long sum = 0, sum_small = 0;
    int first_big = nd;
    for (int i = hi2 - 1; i >= lo2; --i) {
      const int c = dc[i];
      sum += c;
      if ((double)c >= mean) first_big = i; else sum_small += c;
    }
    s_sum[t] = sum; s_small[t] = sum_small; s_first_big[t] = first_big;
    __syncthreads();
    if (t == 0) {
      long long a = 0, b = 0;
      for (int i = 0; i < 256; ++i) { long long x = s_sum[i], y = s_small[i]; s_sum[i] = a; s_small[i] = b; a += x; b += y; }
      int nxt = nd;
      for (int i = 255; i >= 0; --i) { int x = s_first_big[i]; s_first_big[i] = nxt; if (x < nd) nxt = x; }
      pf[0] = 0; sp[0] = 0; nbg[nd] = nd;
    }
    __syncthreads();
    long long a = s_sum[t], b = s_small[t];
    for (int i = lo2; i < hi2; ++i) {
      const int c = dc[i];
      a += c;
      if (!((double)c >= mean)) b += c;
      pf[i + 1] = a;
      sp[i + 1] = b;
    }
    int nxt = s_first_big[t];   // first big value in the segments after this one
    for (int i = hi2 - 1; i >= lo2; --i) {
      if ((double)dc[i] >= mean) nxt = i;
      nbg[i] = nxt;
    }
    __threadfence_block();
    __syncthreads();
  }
  if (t == 0) {
    FeatureBins r;
    r.n_bounds = greedy_bounds_jump(dv, dc, pf, sp, nbg, nd, value_bins, nv, min_data_in_bin,
                                    bounds + (int64_t)f * (kMaxBins + 1));
    r.n_nan = n_nan;
    r.n_distinct = nd;
    r.pad = 0;
    r.min_val = nd > 0 ? dv[0] : 0.0;
    r.max_val = nd > 0 ? dv[nd - 1] : 0.0;
    out[f] = r;
  }
}

}  // namespace

// sample: device memory, row-major [ns][ncol]; returns one BinMapper per column.
std::vector<BinMapper> find_bins_on_device(const void* dev_sample, int dtype, int64_t ns, int ncol, const Config& cfg) {
  hipStream_t s = current_stream();
  std::vector<BinMapper> mappers((size_t)ncol);
  if (ns <= 0) return mappers;
  const int64_t npad = sort_padded_size(ns);
  const int batch = std::max(1, std::min(ncol, (int)std::max<int64_t>(1, (int64_t)(1ll << 27) / npad)));   // <= 1 GiB of keys
  DeviceBuffer<uint64_t> keys((size_t)batch * (size_t)npad);
  DeviceBuffer<double> distinct((size_t)batch * (size_t)ns);
  DeviceBuffer<int32_t> counts((size_t)batch * (size_t)ns);
  DeviceBuffer<int64_t> prefix((size_t)batch * (size_t)(ns + 1)), small_pref((size_t)batch * (size_t)(ns + 1));
  DeviceBuffer<int32_t> next_big((size_t)batch * (size_t)(ns + 1));
  DeviceBuffer<double> bounds((size_t)batch * (kMaxBins + 1));
  DeviceBuffer<FeatureBins> res((size_t)batch);
  std::vector<double> h_bounds((size_t)batch * (kMaxBins + 1));
  std::vector<FeatureBins> h_res((size_t)batch);
  for (int f0 = 0; f0 < ncol; f0 += batch) {
    const int nf = std::min(batch, ncol - f0);
    dim3 g1((unsigned)std::min<int64_t>((npad + 255) / 256, 1024), (unsigned)nf);
    if (dtype == 0)
      hipLaunchKernelGGL(sample_to_keys_kernel<float>, g1, dim3(256), 0, s, (const float*)dev_sample, ns, ncol, f0, nf, npad, keys.get());
    else
      hipLaunchKernelGGL(sample_to_keys_kernel<double>, g1, dim3(256), 0, s, (const double*)dev_sample, ns, ncol, f0, nf, npad, keys.get());
    LGA_HIP_CHECK(hipGetLastError());
    bitonic_sort_u64(keys.get(), npad, nf, s);
    hipLaunchKernelGGL(rle_and_cut_kernel, dim3((unsigned)nf), dim3(256), 0, s, keys.get(), npad, ns, distinct.get(),
                       counts.get(), prefix.get(), small_pref.get(), next_big.get(), cfg.max_bin, cfg.min_data_in_bin, cfg.use_missing ? 1 : 0, bounds.get(), res.get());
    LGA_HIP_CHECK(hipGetLastError());
    LGA_HIP_CHECK(hipMemcpyAsync(h_bounds.data(), bounds.get(), sizeof(double) * (size_t)nf * (kMaxBins + 1), hipMemcpyDeviceToHost, s));
    LGA_HIP_CHECK(hipMemcpyAsync(h_res.data(), res.get(), sizeof(FeatureBins) * (size_t)nf, hipMemcpyDeviceToHost, s));
    LGA_HIP_CHECK(hipStreamSynchronize(s));
    for (int i = 0; i < nf; ++i) {
      const FeatureBins& r = h_res[(size_t)i];
      BinMapper& m = mappers[(size_t)(f0 + i)];
      m.upper.assign(h_bounds.begin() + (size_t)i * (kMaxBins + 1), h_bounds.begin() + (size_t)i * (kMaxBins + 1) + r.n_bounds);
      int nv = (int)m.upper.size();
      m.missing = (cfg.use_missing && r.n_nan > 0) ? MissingType::NaN : MissingType::None;
      if (m.missing == MissingType::NaN) { m.missing_bin = nv; m.num_bin = nv + 1; }
      else { m.missing_bin = -1; m.num_bin = nv; }
      m.min_val = r.min_val;
      m.max_val = r.max_val;
      m.trivial = m.num_bin <= 1;
    }
  }
  return mappers;
}

}  // namespace lga

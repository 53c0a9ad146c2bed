// Ranking gradients on the device: one workgroup per query, the query's documents staged in LDS.
//
//   lambdarank   (score, row) keys are sorted in LDS (bitonic; the row number settles ties like the host's
//                stable sort), then every document sums its own pair terms by ascending partner rank -
//                the summation order ranking.hpp fixes for host and device alike.
//   rank_xendcg  softmax, noisy labels and the three per-query totals; the totals are added by one lane in
//                row order, again the host's order.
//
// The pair formula is the shared rank_pair() of ranking.hpp; exp() and log2() are the device library's, so
// results equal the host's up to the last bits those two functions differ in.
#include "device.hpp"
#include "kernels.hpp"
#include "../ranking.hpp"
#include <mutex>

namespace lga {
namespace k {

namespace {

constexpr int kRankThreads = 256;
constexpr int kWave = 64;
#define LGA_LAUNCH_CHECK() LGA_HIP_CHECK(hipGetLastError())

__device__ __forceinline__ uint64_t ordered_bits(double v) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double from_ordered(uint64_t o) {
  const uint64_t b = (o >> 63) ? (o & 0x7fffffffffffffffull) : ~o;
  return __longlong_as_double((long long)b);
}

__device__ __forceinline__ int64_t root_row(const RankArgs& a, int64_t local) {
  return a.to_root ? (int64_t)a.to_root[local] : local;
}

// DPT: documents per thread; the launch picks the smallest instantiation that covers the largest query.
template <int DPT>
__global__ __launch_bounds__(kRankThreads) void lambdarank_kernel(RankArgs a, int P) {
  extern __shared__ unsigned char smem[];
  uint64_t* key = reinterpret_cast<uint64_t*>(smem);          // sort key, later the score itself
  double* part = reinterpret_cast<double*>(key + P);          // per rank: pair terms with partners further down
  int32_t* idx = reinterpret_cast<int32_t*>(part + P);        // rank -> document of the query
  int32_t* grade = idx + P;                                   // rank -> relevance label
  __shared__ double factor_s;
  const int tid = threadIdx.x;
  const int q = blockIdx.x;
  const int64_t s0 = a.bounds[q];
  const int cnt = (int)(a.bounds[q + 1] - s0);
  const double imd = a.inv_max_dcg[q];
  if (cnt <= 1 || !(imd > 0.0) || cnt > DPT * kRankThreads) {   // the last case is excluded by the host
    for (int i = tid; i < cnt; i += kRankThreads) {
      const int64_t row = root_row(a, s0 + i);
      a.g[row] = 0.0f;
      a.h[row] = 0.0f;
    }
    return;
  }
  int n2 = 1;
  while (n2 < cnt) n2 <<= 1;
  for (int i = tid; i < n2; i += kRankThreads) {
    if (i < cnt) {
      key[i] = ~ordered_bits(a.score[root_row(a, s0 + i)] + 0.0);   // + 0.0: -0 and +0 are one score
      idx[i] = i;
    } else {
      key[i] = ~0ull;
      idx[i] = 0x7fffffff;
    }
  }
  __syncthreads();
  for (int k2 = 2; k2 <= n2; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n2; i += kRankThreads) {
        const int o = i ^ j;
        if (o > i) {
          const uint64_t ka = key[i], kb = key[o];
          const int ia = idx[i], ib = idx[o];
          const bool a_after_b = ka > kb || (ka == kb && ia > ib);
          const bool ascending = (i & k2) == 0;
          if (a_after_b == ascending) { key[i] = kb; key[o] = ka; idx[i] = ib; idx[o] = ia; }
        }
      }
      __syncthreads();
    }
  }
  double* score = reinterpret_cast<double*>(key);
  for (int r = tid; r < cnt; r += kRankThreads) {
    score[r] = from_ordered(~key[r]);
    grade[r] = (int)a.label[root_row(a, s0 + idx[r])];
  }
  __syncthreads();
  const bool normalise = a.norm && score[0] != score[cnt - 1];
  const int top = cnt < a.trunc ? cnt : a.trunc;
  double lam[DPT], hes[DPT];
#pragma unroll
  for (int kk = 0; kk < DPT; ++kk) {
    const int r = tid + kk * kRankThreads;
    lam[kk] = 0.0;
    hes[kk] = 0.0;
    if (r < cnt) {
      const int my_grade = grade[r];
      const double my_score = score[r], my_gain = a.label_gain[my_grade], my_disc = a.discount[r];
      const int partners = r < top ? cnt : top;   // a pair needs one member among the first `top` ranks
      double l = 0.0, hh = 0.0, pt = 0.0;
      for (int o = 0; o < partners; ++o) {
        const int og = grade[o];
        if (o == r || og == my_grade) continue;
        double pl, ph;
        if (my_grade > og) {
          rank_pair(my_score, score[o], my_gain, a.label_gain[og], my_disc, a.discount[o], imd, a.sigmoid, normalise, &pl, &ph);
          l += pl;
        } else {
          rank_pair(score[o], my_score, a.label_gain[og], my_gain, a.discount[o], my_disc, imd, a.sigmoid, normalise, &pl, &ph);
          l -= pl;
        }
        hh += ph;
        if (o > r) pt -= 2.0 * pl;
      }
      lam[kk] = l;
      hes[kk] = hh;
      part[r] = pt;
    }
  }
  __syncthreads();
  if (tid == 0) {
    double total = 0.0;
    for (int r = 0; r < cnt; ++r) total += part[r];
    factor_s = (a.norm && total > 0.0) ? log2(1.0 + total) / total : 1.0;
  }
  __syncthreads();
  const double f = factor_s;
#pragma unroll
  for (int kk = 0; kk < DPT; ++kk) {
    const int r = tid + kk * kRankThreads;
    if (r < cnt) {
      const int64_t row = root_row(a, s0 + idx[r]);
      float g = (float)(lam[kk] * f), h = (float)(hes[kk] * f);
      if (a.weight) { const float w = a.weight[row]; g *= w; h *= w; }
      a.g[row] = g;
      a.h[row] = h;
    }
  }
}

__global__ __launch_bounds__(kRankThreads) void xendcg_kernel(RankArgs a, int P) {
  extern __shared__ unsigned char smem[];
  double* rho = reinterpret_cast<double*>(smem);
  double* par = rho + P;
  __shared__ double total_s, red[kRankThreads / kWave];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.x;
  const int64_t s0 = a.bounds[q];
  const int cnt = (int)(a.bounds[q + 1] - s0);
  if (cnt <= 1 || cnt > P) {
    for (int i = tid; i < cnt; i += kRankThreads) {
      const int64_t row = root_row(a, s0 + i);
      a.g[row] = 0.0f;
      a.h[row] = 0.0f;
    }
    return;
  }
  // max is exact in any order
  double mx = -1.0e300;
  for (int i = tid; i < cnt; i += kRankThreads) mx = fmax(mx, a.score[root_row(a, s0 + i)]);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) mx = fmax(mx, __shfl_xor(mx, m, kWave));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w = 1; w < kRankThreads / kWave; ++w) mx = fmax(mx, red[w]);
  for (int i = tid; i < cnt; i += kRankThreads) {
    const int64_t row = root_row(a, s0 + i);
    rho[i] = exp(a.score[row] - mx);
    const double u = (double)rng_uniform(a.seed, a.iter, (uint64_t)(s0 + i));
    par[i] = ldexp(1.0, (int)a.label[row]) - u;
  }
  __syncthreads();
  // one lane adds in row order: the host's order
  auto row_order_sum = [&](const double* v) {
    if (tid == 0) {
      double t = 0.0;
      for (int i = 0; i < cnt; ++i) t += v[i];
      total_s = t;
    }
    __syncthreads();
    const double t = total_s;
    __syncthreads();
    return t;
  };
  const double z = row_order_sum(rho);
  const double sum_labels = row_order_sum(par);
  const double inv_denominator = 1.0 / fmax(sum_labels, 1e-15);
  // the three terms need every document's running lambda: kept in registers per stride position
  constexpr int kMaxPerThread = kMaxDeviceQuery / kRankThreads;
  double acc[kMaxPerThread];
#pragma unroll
  for (int kk = 0; kk < kMaxPerThread; ++kk) {
    const int i = tid + kk * kRankThreads;
    acc[kk] = 0.0;
    if (i < cnt) {
      rho[i] /= z;
      const double term = -par[i] * inv_denominator + rho[i];
      acc[kk] = term;
      par[i] = term / (1.0 - rho[i]);
    }
  }
  __syncthreads();
  const double sum_l1 = row_order_sum(par);
#pragma unroll
  for (int kk = 0; kk < kMaxPerThread; ++kk) {
    const int i = tid + kk * kRankThreads;
    if (i < cnt) {
      const double term = rho[i] * (sum_l1 - par[i]);
      acc[kk] += term;
      par[i] = term / (1.0 - rho[i]);
    }
  }
  __syncthreads();
  const double sum_l2 = row_order_sum(par);
#pragma unroll
  for (int kk = 0; kk < kMaxPerThread; ++kk) {
    const int i = tid + kk * kRankThreads;
    if (i < cnt) {
      const int64_t row = root_row(a, s0 + i);
      float g = (float)(acc[kk] + rho[i] * (sum_l2 - par[i]));
      float h = (float)(rho[i] * (1.0 - rho[i]));
      if (a.weight) { const float w = a.weight[row]; g *= w; h *= w; }
      a.g[row] = g;
      a.h[row] = h;
    }
  }
}


// ndcg@k and map@k of every query for up to kMaxRankCutoffs cut-offs: the same LDS sort as the lambdarank
// kernel, then one lane walks the top of the ranking - the additions run in the host's order
// (ranking.cpp: ndcg_sums / map_sums), so the values are the host's bit for bit.
// out: per query [ndcg x n_at | map x n_at].
__global__ __launch_bounds__(kRankThreads) void rank_metrics_kernel(RankArgs a, int P, RankCutoffs at, double* __restrict__ out) {
  extern __shared__ unsigned char smem[];
  uint64_t* key = reinterpret_cast<uint64_t*>(smem);
  int32_t* idx = reinterpret_cast<int32_t*>(key + P);
  int32_t* grade = idx + P;
  __shared__ int grade_count[64];
  const int tid = threadIdx.x;
  const int q = blockIdx.x;
  const int64_t s0 = a.bounds[q];
  const int cnt = (int)(a.bounds[q + 1] - s0);
  double* o = out + (int64_t)q * 2 * at.n;
  if (tid < 64) grade_count[tid] = 0;
  if (cnt <= 0 || cnt > P) {
    if (tid == 0)
      for (int k = 0; k < 2 * at.n; ++k) o[k] = 1.0;
    return;
  }
  int n2 = 1;
  while (n2 < cnt) n2 <<= 1;
  for (int i = tid; i < n2; i += kRankThreads) {
    if (i < cnt) { key[i] = ~ordered_bits(a.score[root_row(a, s0 + i)] + 0.0); idx[i] = i; }
    else { key[i] = ~0ull; idx[i] = 0x7fffffff; }
  }
  __syncthreads();
  for (int k2 = 2; k2 <= n2; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n2; i += kRankThreads) {
        const int p = i ^ j;
        if (p > i) {
          const uint64_t ka = key[i], kb = key[p];
          const int ia = idx[i], ib = idx[p];
          const bool a_after_b = ka > kb || (ka == kb && ia > ib);
          if (a_after_b == ((i & k2) == 0)) { key[i] = kb; key[p] = ka; idx[i] = ib; idx[p] = ia; }
        }
      }
      __syncthreads();
    }
  }
  for (int r = tid; r < cnt; r += kRankThreads) {
    const int g = (int)a.label[root_row(a, s0 + idx[r])];
    grade[r] = g;
    atomicAdd(&grade_count[g < 63 ? g : 63], 1);
  }
  __syncthreads();
  if (tid != 0) return;
  int relevant = 0;
  for (int g = 1; g < 64; ++g) relevant += grade_count[g];
  for (int k = 0; k < at.n; ++k) {
    const int top = cnt < at.k[k] ? cnt : at.k[k];
    // the ranking's DCG and, from the grade counts walked downwards, the best possible one
    double dcg = 0.0, best = 0.0, ap = 0.0;
    int hits = 0, g = 63, left = grade_count[63];
    for (int i = 0; i < top; ++i) {
      dcg += a.label_gain[grade[i]] * a.discount[i];
      while (left == 0 && g > 0) { --g; left = grade_count[g]; }
      best += a.label_gain[g] * a.discount[i];
      --left;
      if (grade[i] > 0) { ++hits; ap += (double)hits / (double)(i + 1); }
    }
    o[k] = best > 0.0 ? dcg / best : 1.0;
    o[at.n + k] = relevant > 0 ? ap / (double)(relevant < top ? relevant : top) : 1.0;
  }
}

// The LDS budget of every instantiation is raised once, to what the largest supported query needs
// (boosters on several host threads launch these kernels concurrently).
template <typename Kernel>
void launch_rank(Kernel kernel, std::once_flag& once, const RankArgs& a, size_t per_doc, hipStream_t s) {
  std::call_once(once, [&] {
    LGA_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(per_doc * (size_t)kMaxDeviceQuery)));
  });
  int P = 1;
  while (P < a.max_query) P <<= 1;
  hipLaunchKernelGGL(kernel, dim3((unsigned)a.nq), dim3(kRankThreads), per_doc * (size_t)P, s, a, P);
  LGA_LAUNCH_CHECK();
}

}  // namespace

void lambdarank_gradients(const RankArgs& a, hipStream_t s) {
  if (a.nq <= 0) return;
  if (a.max_query > kMaxDeviceQuery) throw std::runtime_error("lambdarank_gradients: query larger than the device kernel holds");
  constexpr size_t per_doc = sizeof(uint64_t) + sizeof(double) + 2 * sizeof(int32_t);
  static std::once_flag once[3];
  if (a.max_query <= kRankThreads) launch_rank(lambdarank_kernel<1>, once[0], a, per_doc, s);
  else if (a.max_query <= 4 * kRankThreads) launch_rank(lambdarank_kernel<4>, once[1], a, per_doc, s);
  else launch_rank(lambdarank_kernel<kMaxDeviceQuery / kRankThreads>, once[2], a, per_doc, s);
}

void rank_metrics(const RankArgs& a, const RankCutoffs& at, double* out, hipStream_t s) {
  if (a.nq <= 0) return;
  if (a.max_query > kMaxDeviceQuery) throw std::runtime_error("rank_metrics: query larger than the device kernel holds");
  constexpr size_t per_doc = sizeof(uint64_t) + 2 * sizeof(int32_t);
  static std::once_flag once;
  std::call_once(once, [&] {
    LGA_HIP_CHECK(hipFuncSetAttribute((const void*)rank_metrics_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(per_doc * (size_t)kMaxDeviceQuery)));
  });
  int P = 1;
  while (P < a.max_query) P <<= 1;
  hipLaunchKernelGGL(rank_metrics_kernel, dim3((unsigned)a.nq), dim3(kRankThreads), per_doc * (size_t)P, s, a, P, at, out);
  LGA_LAUNCH_CHECK();
}

void xendcg_gradients(const RankArgs& a, hipStream_t s) {
  if (a.nq <= 0) return;
  if (a.max_query > kMaxDeviceQuery) throw std::runtime_error("xendcg_gradients: query larger than the device kernel holds");
  static std::once_flag once;
  launch_rank(xendcg_kernel, once, a, 2 * sizeof(double), s);
}

}  // namespace k
}  // namespace lga

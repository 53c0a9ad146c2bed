// Exact (unweighted) AUC on the device: what `eval = "auc"` costs per evaluation when the scores live
// in HBM.  The host formulation (metric.cpp: auc_host) sorts (score, label) pairs; here
//   1. every row becomes one 64-bit key: the order-preserving bit pattern of its fp64 score with the
//      lowest bit replaced by the label, so equal scores sort negatives-before-positives and two scores
//      one ulp apart count as tied (a 2^-52 relative change of a score),
//   2. the keys are sorted ascending (sort.hip),
//   3. a three-kernel scan gives every positive the number of negatives below it (neg_lt) and
//      below-or-tied (neg_le); sum(neg_lt + neg_le) / 2 / (P * N) is the AUC.  Counts are integers, so
//      the result does not depend on the order of the summation.
#include "hip_engine.hpp"
#include "sort.hpp"

namespace lga {

namespace {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanBlock = kScanThreads * kScanItems;   // 2048 keys per workgroup

__device__ __forceinline__ uint64_t score_key(double v, bool positive) {
  uint64_t b = (uint64_t)__double_as_longlong(v);
  b = (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
  return (b & ~1ull) | (positive ? 1ull : 0ull);
}

__global__ __launch_bounds__(256) void auc_keys_kernel(const double* __restrict__ scores, const float* __restrict__ label,
                                                       const int32_t* __restrict__ rows, int64_t n, int64_t npad,
                                                       uint64_t* __restrict__ keys) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t k = ~0ull;
    if (i < n) {
      const int64_t r = rows ? rows[i] : i;
      double s = scores[r];
      if (s != s) s = 0.0;
      if (s == 0.0) s = 0.0;
      k = score_key(s, label[r] > 0.0f);
    }
    keys[i] = k;
  }
}

__device__ __forceinline__ long long shfl_up_ll(long long v, int d) {
  int lo = __shfl_up((int)(unsigned)((unsigned long long)v & 0xffffffffull), d, 64);
  int hi = __shfl_up((int)(unsigned)((unsigned long long)v >> 32), d, 64);
  return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}

// Per 2048-key block: negatives in the block, and the number of block-local negatives before its last
// group head (-1 when no group starts inside the block).
__global__ __launch_bounds__(kScanThreads) void auc_block_reduce_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                                        long long* __restrict__ block_neg,
                                                                        long long* __restrict__ block_head) {
  __shared__ long long s_neg[kScanThreads];
  __shared__ int s_head[kScanThreads];
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + (int64_t)threadIdx.x * kScanItems;
  long long neg = 0;
  int last_head = -1;          // item index (within the thread) of the thread's last head
  long long neg_before_head = 0;
  uint64_t prev = base > 0 && base <= n ? keys[base - 1] : 0ull;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t p = base + j;
    if (p < n) {
      const uint64_t k = keys[p];
      if (p == 0 || (k >> 1) != (prev >> 1)) { last_head = j; neg_before_head = neg; }
      neg += (k & 1ull) ? 0 : 1;
      prev = k;
    }
  }
  s_neg[threadIdx.x] = neg;
  s_head[threadIdx.x] = last_head;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long total = 0;
    for (int t = 0; t < kScanThreads; ++t) total += s_neg[t];
    block_neg[blockIdx.x] = total;
  }
  // last thread that owns a head publishes: negatives of the threads before it + its own count before the head
  __shared__ int s_last_owner;
  if (threadIdx.x == 0) {
    int owner = -1;
    for (int t = 0; t < kScanThreads; ++t) if (s_head[t] >= 0) owner = t;
    s_last_owner = owner;
    if (owner < 0) block_head[blockIdx.x] = -1;
  }
  __syncthreads();
  if ((int)threadIdx.x == s_last_owner) {
    long long before = 0;
    for (int t = 0; t < (int)threadIdx.x; ++t) before += s_neg[t];
    block_head[blockIdx.x] = before + neg_before_head;
  }
}

// Exclusive scan of one value per thread over a 1024-thread workgroup (sum or max).
template <bool MAX>
__device__ __forceinline__ long long block_exclusive_scan_1024(long long v, long long identity, long long* lds16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    long long o = shfl_up_ll(incl, d);
    if (lane >= d) incl = MAX ? (incl > o ? incl : o) : incl + o;
  }
  __syncthreads();
  if (lane == 63) lds16[wave] = incl;
  __syncthreads();
  long long off = identity;
  for (int w = 0; w < wave; ++w) off = MAX ? (off > lds16[w] ? off : lds16[w]) : off + lds16[w];
  long long before = shfl_up_ll(incl, 1);
  if (lane == 0) before = identity;
  return MAX ? (off > before ? off : before) : off + before;
}

// One workgroup: exclusive prefix of the blocks' negatives, and for every block the neg_lt value of
// the group that is open when the block begins.  Each thread owns a run of consecutive blocks.
__global__ __launch_bounds__(1024) void auc_block_scan_kernel(long long* __restrict__ block_neg,
                                                              long long* __restrict__ block_head, int n_blocks) {
  __shared__ long long lds16[16];
  const int per = (n_blocks + 1023) / 1024;
  const int lo = min((int)threadIdx.x * per, n_blocks), hi = min(lo + per, n_blocks);
  long long sum = 0;
  for (int b = lo; b < hi; ++b) sum += block_neg[b];
  long long run = block_exclusive_scan_1024<false>(sum, 0, lds16);
  long long last_head = -1;
  for (int b = lo; b < hi; ++b) {
    const long long cnt = block_neg[b], local_head = block_head[b];
    block_neg[b] = run;
    if (local_head >= 0) last_head = run + local_head;
    run += cnt;
  }
  long long open_group = block_exclusive_scan_1024<true>(last_head, -1, lds16);
  for (int b = lo; b < hi; ++b) {
    const long long local_head = block_head[b];
    block_head[b] = open_group;
    if (local_head >= 0) open_group = block_neg[b] + local_head;
  }
}

__global__ __launch_bounds__(kScanThreads) void auc_block_final_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                                       const long long* __restrict__ block_neg,
                                                                       const long long* __restrict__ block_head,
                                                                       unsigned long long* __restrict__ out3) {
  __shared__ long long s_pref[kScanThreads / 64];
  __shared__ long long s_headv[kScanThreads / 64];
  __shared__ unsigned long long s_red[2][kScanThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)blockIdx.x * kScanBlock + (int64_t)threadIdx.x * kScanItems;
  uint64_t k[kScanItems];
  bool valid[kScanItems], head[kScanItems];
  uint64_t prev = base > 0 && base <= n ? keys[base - 1] : 0ull;
  long long neg = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t p = base + j;
    valid[j] = p < n;
    k[j] = valid[j] ? keys[p] : ~0ull;
    head[j] = valid[j] && (p == 0 || (k[j] >> 1) != (prev >> 1));
    if (valid[j]) { neg += (k[j] & 1ull) ? 0 : 1; prev = k[j]; }
  }
  // exclusive prefix of the threads' negative counts inside the block
  long long incl = neg;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    long long o = shfl_up_ll(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) s_pref[wave] = incl;
  __syncthreads();
  long long wave_off = 0;
  for (int w = 0; w < wave; ++w) wave_off += s_pref[w];
  long long run = block_neg[blockIdx.x] + wave_off + incl - neg;   // negatives before this thread's first key
  // neg_lt of the group open at each key: value at the latest head (values never decrease, so a max-scan)
  long long my_head = -1;
  {
    long long r = run;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
      if (head[j]) my_head = r;
      if (valid[j]) r += (k[j] & 1ull) ? 0 : 1;
    }
  }
  long long hincl = my_head;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    long long o = shfl_up_ll(hincl, d);
    if (lane >= d) hincl = hincl > o ? hincl : o;
  }
  if (lane == 63) s_headv[wave] = hincl;
  __syncthreads();
  long long carry = block_head[blockIdx.x];
  for (int w = 0; w < wave; ++w) carry = carry > s_headv[w] ? carry : s_headv[w];
  long long before = shfl_up_ll(hincl, 1);
  if (lane > 0) carry = carry > before ? carry : before;
  // walk the thread's keys
  unsigned long long acc = 0, pos = 0;
  long long open_group = carry;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    if (valid[j]) {
      if (head[j]) open_group = run;
      if (k[j] & 1ull) { acc += (unsigned long long)(run + open_group); ++pos; }
      else ++run;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned lo = __shfl_xor((unsigned)(acc & 0xffffffffull), m, 64), hi = __shfl_xor((unsigned)(acc >> 32), m, 64);
    acc += ((unsigned long long)hi << 32) | lo;
    lo = __shfl_xor((unsigned)(pos & 0xffffffffull), m, 64); hi = __shfl_xor((unsigned)(pos >> 32), m, 64);
    pos += ((unsigned long long)hi << 32) | lo;
  }
  if (lane == 0) { s_red[0][wave] = acc; s_red[1][wave] = pos; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long a = 0, p = 0;
    for (int w = 0; w < kScanThreads / 64; ++w) { a += s_red[0][w]; p += s_red[1][w]; }
    atomicAdd(&out3[0], a);
    atomicAdd(&out3[1], p);
  }
}

}  // namespace

// AUC of scores[rows[i]] against label[rows[i]] > 0, i < n (rows == nullptr: identity).  Unweighted.
double auc_on_device(const double* scores, const float* label, const int32_t* rows, int64_t n, hipStream_t s) {
  if (n <= 0) return 1.0;
  const int64_t npad = sort_padded_size(n);
  const int n_blocks = (int)((n + kScanBlock - 1) / kScanBlock);
  DeviceBuffer<uint64_t> keys((size_t)npad);
  DeviceBuffer<long long> block_neg((size_t)n_blocks), block_head((size_t)n_blocks);
  DeviceBuffer<unsigned long long> out(4);
  PinnedBuffer<unsigned long long> host;
  host.resize(4);
  hipLaunchKernelGGL(auc_keys_kernel, dim3((unsigned)std::min<int64_t>((npad + 255) / 256, 4096)), dim3(256), 0, s,
                     scores, label, rows, n, npad, keys.get());
  LGA_HIP_CHECK(hipGetLastError());
  bitonic_sort_u64(keys.get(), npad, 1, s);
  out.zero(s);
  hipLaunchKernelGGL(auc_block_reduce_kernel, dim3((unsigned)n_blocks), dim3(kScanThreads), 0, s, keys.get(), n,
                     block_neg.get(), block_head.get());
  hipLaunchKernelGGL(auc_block_scan_kernel, dim3(1), dim3(1024), 0, s, block_neg.get(), block_head.get(), n_blocks);
  hipLaunchKernelGGL(auc_block_final_kernel, dim3((unsigned)n_blocks), dim3(kScanThreads), 0, s, keys.get(), n,
                     block_neg.get(), block_head.get(), out.get());
  LGA_HIP_CHECK(hipGetLastError());
  LGA_HIP_CHECK(hipMemcpyAsync(host.get(), out.get(), sizeof(unsigned long long) * 2, hipMemcpyDeviceToHost, s));
  LGA_HIP_CHECK(hipStreamSynchronize(s));
  const double twice = (double)host[0], P = (double)host[1], Nn = (double)n - P;
  if (P <= 0.0 || Nn <= 0.0) return 1.0;
  return (twice * 0.5) / (P * Nn);
}

}  // namespace lga

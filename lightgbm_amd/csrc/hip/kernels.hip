// Training kernels for gfx950 (CDNA4, wave64).
//
//  * histogram:  one lane = one row; the row's 32 bins of a feature tile arrive
//    as two 16-byte loads; (g,h) are added as ONE packed 64-bit integer with an
//    LDS atomic (ds_add_u64) into a 64 KiB per-workgroup histogram; the
//    workgroup then merges its non-zero cells into the int64 global slot.
//    Integer sums make the result independent of thread order.
//  * partition:  ballot bit masks + popcount ranks give a stable split of a
//    leaf's row list without sorting; two launches (count, scatter) per round
//    for ALL leaves of the round.
//  * scan:       one wave per (leaf, feature): 4 bins per lane, wave prefix sum
//    by shuffles, fp64 gains, wave arg-max; sibling histograms are derived in
//    place (parent - smaller child) on the way in.
#include "kernels.hpp"
#include <climits>
#include <sstream>
#include <type_traits>
#include <string>
#include "device.hpp"
#include "wave_scan.hpp"
#include "../percentile.hpp"

namespace lga {
namespace k {

namespace {

// Debug builds (-DLGA_DEBUG, lightgbm_amd/build.py --debug): a kernel verifies every index it is about to
// use for a scattered access - row ids, scatter targets, tree nodes - and, instead of faulting, records the
// first violation in a small device buffer and skips the access.  The host reads the record after the round
// (take_check_failure) and raises.  Release builds compile the checks away.
#ifdef LGA_DEBUG
__device__ int* g_check_sink = nullptr;   // [hits, code, block, value lo, value hi, bound lo, bound hi, -]
__device__ __noinline__ bool check_failed(int code, long long value, long long bound) {
  int* sink = g_check_sink;
  if (sink && atomicAdd(&sink[0], 1) == 0) {
    sink[1] = code;
    sink[2] = (int)blockIdx.x;
    sink[3] = (int)(value & 0xffffffffll); sink[4] = (int)(value >> 32);
    sink[5] = (int)(bound & 0xffffffffll); sink[6] = (int)(bound >> 32);
  }
  return false;
}
#define LGA_DEV_CHECK(cond, code, value, bound) ((cond) || check_failed((code), (long long)(value), (long long)(bound)))
#else
#define LGA_DEV_CHECK(cond, code, value, bound) (true)
#endif
enum CheckCode : int {
  kCheckHistRow = 1,          // histogram: row id outside [0, N)
  kCheckPartitionRow = 2,     // partition: row id outside [0, N)
  kCheckPartitionTarget = 3,  // partition: scatter position outside the leaf's range
  kCheckWalkRow = 4,          // tree walk: row id outside [0, N)
  kCheckWalkNode = 5,         // tree walk: child index outside the tree
};


// Same-address global atomics are serialised in the memory system (~10+ ns each), so reductions
// first combine a whole workgroup and, for a maximum, skip the atomic when it cannot change anything.
__device__ __forceinline__ void publish_max(uint32_t* target, float v) {
  const uint32_t bits = __float_as_uint(v);   // non-negative floats order like their bit patterns
  if (bits > __atomic_load_n(target, __ATOMIC_RELAXED)) atomicMax(target, bits);
}
// The two running maxima (|g|, h) of a 256-thread workgroup -> ONE atomic each per workgroup (thousands of waves
// publishing to the same two words at once queue up behind each other for tens of microseconds: the fused score
// update + gradients pass over a 1.3 M-row shard took 68 us with one atomic per wave, 30 us this way).
__device__ __forceinline__ void publish_block_max(uint32_t* maxbits, float mg, float mh) {
  __shared__ float wg_max[2][4];
  mg = wave_max_f32(mg);
  mh = wave_max_f32(mh);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { wg_max[0][wave] = mg; wg_max[1][wave] = mh; }
  __syncthreads();
  if (threadIdx.x == 0) {
    publish_max(&maxbits[0], fmaxf(fmaxf(wg_max[0][0], wg_max[0][1]), fmaxf(wg_max[0][2], wg_max[0][3])));
    publish_max(&maxbits[1], fmaxf(fmaxf(wg_max[1][0], wg_max[1][1]), fmaxf(wg_max[1][2], wg_max[1][3])));
  }
}
template <int WAVES>
__device__ __forceinline__ double block_sum_f64(double v, double* lds_scratch) {
  v = wave_sum_f64(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds_scratch[wave] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) t += lds_scratch[w];
  return t;
}

inline int grid_for(int64_t n, int threads, int max_blocks = 2048) {
  int64_t b = (n + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

// ---------------------------------------------------------------------------
// gradients / quantisation
// ---------------------------------------------------------------------------
// Four independent rows per lane and trip: all loads of a trip are issued before the (fp64) math,
// which keeps enough memory requests in flight at 8 waves per SIMD.
__global__ __launch_bounds__(256) void gradients_kernel(ObjectiveSpec obj, const double* __restrict__ scores,
                                                        const float* __restrict__ label,
                                                        const float* __restrict__ weight,
                                                        const int32_t* __restrict__ rows, int64_t n,
                                                        float* __restrict__ g, float* __restrict__ h,
                                                        uint32_t* maxbits) {
  constexpr int U = 4;
  float mg = 0.f, mh = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += stride * U) {
    int64_t r[U];
    double sc[U];
    float lb[U], wt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      r[u] = i < n ? (rows ? (int64_t)rows[i] : i) : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r[u] >= 0) { sc[u] = scores[r[u]]; lb[u] = label[r[u]]; wt[u] = weight ? weight[r[u]] : 1.0f; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r[u] >= 0) {
        float gg, hh;
        grad_hess(obj, sc[u], lb[u], wt[u], &gg, &hh);
        g[r[u]] = gg;
        h[r[u]] = hh;
        mg = fmaxf(mg, fabsf(gg));
        mh = fmaxf(mh, hh);
      }
    }
  }
  publish_block_max(maxbits, mg, mh);
}

__global__ __launch_bounds__(256) void multiclass_gradients_kernel(ObjectiveSpec obj, const double* __restrict__ scores,
                                                                   int64_t stride, const float* __restrict__ label,
                                                                   const float* __restrict__ weight,
                                                                   const int32_t* __restrict__ rows, int64_t n,
                                                                   float* __restrict__ g, float* __restrict__ h) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows ? rows[i] : i;
    multiclass_grad_hess(obj, scores, stride, r, label[r], weight ? weight[r] : 1.0f, g, h);
  }
}

// ---------------------------------------------------------------------------
// gradient-based one-side sampling
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void goss_weights_kernel(const float* __restrict__ g, const float* __restrict__ h,
                                                           int num_class, int64_t stride,
                                                           const int32_t* __restrict__ rows, int64_t n,
                                                           float* __restrict__ w) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    w[i] = goss_weight(g, h, rows ? rows[i] : i, num_class, stride);
}

__global__ __launch_bounds__(256) void goss_amplify_kernel(float* __restrict__ g, float* __restrict__ h, int num_class,
                                                           int64_t stride, const int32_t* __restrict__ rows, int64_t n,
                                                           const float* __restrict__ threshold, float multiplier) {
  const float thr = *threshold;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows ? rows[i] : i;
    if (goss_weight(g, h, r, num_class, stride) >= thr) continue;
    for (int k = 0; k < num_class; ++k) {
      g[(int64_t)k * stride + r] *= multiplier;
      h[(int64_t)k * stride + r] *= multiplier;
    }
  }
}

// Radix select of the k-th largest value.  state: [0] prefix bits found so far, [1] their mask, [2..3] the
// rank still to find inside the prefix (64 bit), after kSelectBins histogram words.
constexpr int kSelectBins = 2048;
__global__ __launch_bounds__(256) void select_init_kernel(uint32_t* scratch, unsigned long long k) {
  for (int i = threadIdx.x; i < kSelectBins + 8; i += blockDim.x) scratch[i] = 0u;
  __syncthreads();
  if (threadIdx.x == 0) *reinterpret_cast<unsigned long long*>(scratch + kSelectBins + 2) = k;
}
__global__ __launch_bounds__(256) void select_count_kernel(const float* __restrict__ v, int64_t n, int shift, int bits,
                                                           uint32_t* __restrict__ scratch) {
  __shared__ unsigned lds[kSelectBins];
  for (int i = threadIdx.x; i < kSelectBins; i += blockDim.x) lds[i] = 0u;
  __syncthreads();
  const uint32_t prefix = scratch[kSelectBins], mask = scratch[kSelectBins + 1];
  const uint32_t digit_mask = (1u << bits) - 1u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t key = __float_as_uint(v[i]);
    if ((key & mask) == prefix) atomicAdd(&lds[(key >> shift) & digit_mask], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kSelectBins; i += blockDim.x)
    if (lds[i]) atomicAdd(&scratch[i], lds[i]);
}
// One workgroup: walks the digit histogram from the top, fixes the digit that holds the wanted rank,
// clears the histogram for the next pass; the last pass writes the value.
__global__ __launch_bounds__(256) void select_pick_kernel(uint32_t* __restrict__ scratch, int shift, int bits, int last,
                                                          float* __restrict__ out) {
  __shared__ unsigned lds[kSelectBins];
  for (int i = threadIdx.x; i < kSelectBins; i += blockDim.x) { lds[i] = scratch[i]; scratch[i] = 0u; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long want = *reinterpret_cast<unsigned long long*>(scratch + kSelectBins + 2);
    unsigned long long seen = 0;
    int digit = 0;
    for (int b = (1 << bits) - 1; b >= 0; --b) {
      if (seen + lds[b] >= want) { digit = b; break; }
      seen += lds[b];
    }
    *reinterpret_cast<unsigned long long*>(scratch + kSelectBins + 2) = want - seen;
    const uint32_t prefix = scratch[kSelectBins] | ((uint32_t)digit << shift);
    scratch[kSelectBins] = prefix;
    scratch[kSelectBins + 1] |= ((1u << bits) - 1u) << shift;
    if (last) *out = __uint_as_float(prefix);
  }
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ g, const float* __restrict__ h,
                                                     const int32_t* __restrict__ rows, int64_t n, uint32_t* maxbits) {
  float mg = 0.f, mh = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = rows ? rows[i] : i;
    mg = fmaxf(mg, fabsf(g[r]));
    mh = fmaxf(mh, h[r]);
  }
  publish_block_max(maxbits, mg, mh);
}

// Full-precision histograms (WIDE): a row's 8-byte gradient word holds the float bit patterns themselves (gradient
// high, hessian low); every histogram pass turns them into 36-bit fixed point on the way (quantize_wide at the
// iteration's exponents + kWideExtraBits - the scale the kernel publishes), as the CPU engine's full-precision mode does.
__device__ __forceinline__ unsigned long long float_pair_bits(float g, float h) {
  return ((unsigned long long)__float_as_uint(g) << 32) | (unsigned long long)__float_as_uint(h);
}
__device__ __forceinline__ void wide_of(unsigned long long pk, int wkg, int wkh, long long& qg, long long& qh) {
  qg = quantize_wide(__uint_as_float((unsigned)(pk >> 32)), wkg);
  qh = quantize_wide(__uint_as_float((unsigned)(pk & 0xffffffffull)), wkh);
}

template <bool WIDE>
__global__ __launch_bounds__(256) void quantize_kernel(const float* __restrict__ g, const float* __restrict__ h,
                                                       const int32_t* __restrict__ rows, int64_t n,
                                                       const uint32_t* __restrict__ maxbits, QuantScale* qs,
                                                       int64_t* __restrict__ packed, RoundingSpec rounding) {
  const int kg = quant_exponent(__uint_as_float(maxbits[0]));
  const int kh = quant_exponent(__uint_as_float(maxbits[1]));
  if (blockIdx.x == 0 && threadIdx.x == 0) { qs->kg = kg + (WIDE ? kWideExtraBits : 0); qs->kh = kh + (WIDE ? kWideExtraBits : 0); }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = rows ? rows[i] : i;
    packed[r] = WIDE ? (int64_t)float_pair_bits(g[r], h[r])
                     : pack_gh(quantize_with(g[r], kg, rounding, 0, (uint64_t)r), quantize_with(h[r], kh, rounding, 1, (uint64_t)r));
  }
}

// ---------------------------------------------------------------------------
// partition
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool goes_left(const SplitTask& t, const uint8_t* __restrict__ bins_fm, int64_t N,
                                          int32_t row, const BagParams& bag) {
  if (t.feature < 0) {
    if (t.feature == -2 && goss_weight(bag.g, bag.h, row, bag.num_class, bag.stride) >= *bag.threshold) return true;
    const float keep = (t.feature == -3 && !(bag.label[row] > 0.0f)) ? bag.neg_fraction : bag.fraction;
    return rng_uniform(bag.seed, bag.iter, bag.salt + (uint64_t)(uint32_t)(bag.ext_ids ? bag.ext_ids[row] : row)) < keep;
  }
  int b = bins_fm[(int64_t)t.feature * N + row];
  if (t.default_left == kCategoricalTask) return b == t.threshold;
  if (t.default_left == kCategoricalSetTask)
    return (bag.cat_sets[(int64_t)t.parent_node * kCatSetWords + (b >> 6)] >> (b & 63)) & 1ull;
  return (b == t.missing_bin) ? (t.default_left != 0) : (b <= t.threshold);
}

// Task descriptors reach a kernel either through a staged device array (+ a chunk -> task table)
// or, for small rounds, inline in the kernel arguments (no host->device copy at all).
template <bool PART>
__device__ __forceinline__ int find_task(const InlineTasks& inl, int wg) {
  // last task whose first chunk is <= wg (first chunks increase strictly: every task owns >= 1 chunk)
  int lo = 0, hi = inl.count - 1;
#pragma unroll 1
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    const int first = PART ? inl.t[mid].first_pchunk : inl.t[mid].first_hchunk;
    if (wg >= first) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <bool PART>
__device__ __forceinline__ int find_task_staged(const SplitTask* __restrict__ tasks, int count, int wg) {
  int lo = 0, hi = count - 1;
#pragma unroll 1
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    const int first = PART ? tasks[mid].first_pchunk : tasks[mid].first_hchunk;
    if (wg >= first) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// A workgroup's task descriptor as a COPY in registers.  Through a reference the compiler may keep a field only
// until the next store - the descriptor sits in kernel-argument or global memory, which a store through some other
// pointer could have changed - so every use after a store fetches the field again and waits for it: in the sides
// kernel that was two dependent memory round trips per 64 rows.
__device__ __forceinline__ SplitTask load_task(const SplitTask* __restrict__ staged, const InlineTasks& inl, int ti) {
  SplitTask t;
  if (staged) t = staged[ti];
  else t = inl.t[ti];
  return t;
}

// Pass 1 of the partition: ballot masks + left counts per 2048-row chunk.  A task that fits ONE
// chunk is finished right here (ranks from the masks in LDS, scatter, child ranges), so rounds made
// of small leaves need no second launch.  The workgroups also clear the histogram slots of the round.
__global__ __launch_bounds__(kPartThreads) void partition_count_kernel(
    const SplitTask* __restrict__ tasks, const int32_t* __restrict__ chunk_task, InlineTasks inl,
    NodeRange* __restrict__ nodes, const int32_t* __restrict__ a0, const int32_t* __restrict__ a1,
    int32_t* __restrict__ a0_dst, int32_t* __restrict__ a1_dst, const uint8_t* __restrict__ bins_fm, int64_t N,
    uint64_t* __restrict__ mask, int32_t* __restrict__ chunk_left, int32_t* __restrict__ task_left, BagParams bag,
    int total_chunks, int64_t* __restrict__ zero_ptr, int64_t zero_words, GradArenas ga) {
  constexpr int kWords = kPartChunk / kWave;
  __shared__ unsigned long long smask[kWords];
  __shared__ int word_prefix[kWords + 1];
  const int wg = blockIdx.x;
  if (wg >= total_chunks) {
    // trailing workgroups only clear the round's new histogram slots (16 KiB each, 16-byte stores)
    longlong2* z = reinterpret_cast<longlong2*>(zero_ptr);
    const int64_t pairs = zero_words >> 1;
    const int64_t base = (int64_t)(wg - total_chunks) * (kPartThreads * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = base + j * kPartThreads + threadIdx.x;
      if (i < pairs) z[i] = make_longlong2(0, 0);
    }
    return;
  }
  const int ti = tasks ? chunk_task[wg] : find_task<true>(inl, wg);
  const SplitTask t = load_task(tasks, inl, ti);
  const int c = wg - t.first_pchunk;
  const NodeRange p = nodes[t.parent_node];
  const int32_t* src = (p.buf ? a1 : a0) + p.begin;
  const int base = c * kPartChunk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int32_t myrow[kPartChunk / kPartThreads];
#pragma unroll
  for (int kk = 0; kk < kPartChunk / kPartThreads; ++kk) {
    const int pos = base + kk * kPartThreads + threadIdx.x;
    bool left = false;
    myrow[kk] = 0;
    if (pos < p.count) {
      myrow[kk] = src[pos];
      if (t.feature < 0 || LGA_DEV_CHECK((unsigned)myrow[kk] < (unsigned long long)N, kCheckPartitionRow, myrow[kk], N))
        left = goes_left(t, bins_fm, N, myrow[kk], bag);
    }
    const unsigned long long m = __ballot(left);
    if (lane == 0) smask[kk * (kPartThreads / kWave) + wave] = m;
  }
  __syncthreads();
  if (threadIdx.x < kWave) {
    int pc = threadIdx.x < kWords ? __popcll(smask[threadIdx.x]) : 0;
    int incl = pc;
#pragma unroll
    for (int d = 1; d < kWords; d <<= 1) {
      int o = __shfl_up(incl, d, kWave);
      if (lane >= d) incl += o;
    }
    if (threadIdx.x < kWords) word_prefix[threadIdx.x] = incl - pc;
    if (threadIdx.x == kWords - 1) word_prefix[kWords] = incl;
  }
  __syncthreads();
  const int lefts = word_prefix[kWords];
  if (t.num_pchunk == 1) {
    int32_t* dst = (p.buf ? a0_dst : a1_dst) + p.begin;
    const bool with_grads = ga.arena[0] != nullptr && t.feature >= 0;
    const int64_t* gh_src = with_grads ? ga.arena[p.buf] + p.begin : nullptr;
    int64_t* gh_dst = with_grads ? ga.arena[1 - p.buf] + p.begin : nullptr;
#pragma unroll
    for (int kk = 0; kk < kPartChunk / kPartThreads; ++kk) {
      const int w = kk * (kPartThreads / kWave) + wave;
      const int in_chunk = w * kWave + lane;
      if (in_chunk < p.count) {
        const unsigned long long m = smask[w];
        const bool left = (m >> lane) & 1ull;
        const int lrank = word_prefix[w] + __popcll(m & ((1ull << lane) - 1ull));
        const int d = left ? lrank : lefts + (in_chunk - lrank);
        if (LGA_DEV_CHECK((unsigned)d < (unsigned)p.count, kCheckPartitionTarget, d, p.count)) {
          dst[d] = myrow[kk];
          if (with_grads) gh_dst[d] = t.parent_node == 0 ? ga.by_row[myrow[kk]] : gh_src[in_chunk];
        }
      }
    }
    if (threadIdx.x == 0) {
      nodes[t.left_node] = NodeRange{p.begin, lefts, 1 - p.buf, 0};
      nodes[t.right_node] = NodeRange{p.begin + lefts, p.count - lefts, 1 - p.buf, 0};
      task_left[ti] = lefts;
    }
  } else {
    if (threadIdx.x < kWords) mask[(int64_t)wg * kWords + threadIdx.x] = smask[threadIdx.x];
    if (threadIdx.x == 0) chunk_left[wg] = lefts;
  }
}

// Pass 2 (tasks that span several chunks): each chunk adds up the left counts of the chunks before
// it and scatters its rows - lefts to the front of the leaf's range, rights behind them, both in order.
__global__ __launch_bounds__(kPartThreads) void partition_scatter_kernel(
    const SplitTask* __restrict__ tasks, const int32_t* __restrict__ chunk_task, InlineTasks inl,
    NodeRange* __restrict__ nodes, const int32_t* __restrict__ a0_src, const int32_t* __restrict__ a1_src,
    int32_t* __restrict__ a0_dst, int32_t* __restrict__ a1_dst, const uint64_t* __restrict__ mask,
    const int32_t* __restrict__ chunk_left, int32_t* __restrict__ task_left, int chunk0, int n_staged, GradArenas ga) {
  constexpr int kWords = kPartChunk / kWave;   // 32 mask words per chunk
  __shared__ int red_before[kPartThreads / kWave], red_all[kPartThreads / kWave];
  __shared__ int word_prefix[kWords];
  const int wg = blockIdx.x + chunk0;
  const int ti = tasks ? (chunk_task ? chunk_task[wg] : find_task_staged<true>(tasks, n_staged, wg)) : find_task<true>(inl, wg);
  const SplitTask t = load_task(tasks, inl, ti);
  if (t.num_pchunk == 1) return;   // finished by pass 1
  const int c = wg - t.first_pchunk;
  const NodeRange p = nodes[t.parent_node];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lefts in earlier chunks of this task, and in all of them
  int before = 0, all = 0;
  for (int j = threadIdx.x; j < t.num_pchunk; j += kPartThreads) {
    int v = chunk_left[t.first_pchunk + j];
    all += v;
    if (j < c) before += v;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    before += __shfl_xor(before, m, kWave);
    all += __shfl_xor(all, m, kWave);
  }
  if (lane == 0) { red_before[wave] = before; red_all[wave] = all; }
  // exclusive prefix of the popcounts of this chunk's mask words
  if (threadIdx.x < kWave) {
    int pc = 0;
    if (threadIdx.x < kWords) pc = __popcll(mask[(int64_t)wg * kWords + threadIdx.x]);
    int incl = pc;
#pragma unroll
    for (int d = 1; d < kWords; d <<= 1) {
      int o = __shfl_up(incl, d, kWave);
      if (lane >= d) incl += o;
    }
    if (threadIdx.x < kWords) word_prefix[threadIdx.x] = incl - pc;
  }
  __syncthreads();
  before = 0; all = 0;
#pragma unroll
  for (int w = 0; w < kPartThreads / kWave; ++w) { before += red_before[w]; all += red_all[w]; }
  const int32_t* src = (p.buf ? a1_src : a0_src) + p.begin;
  int32_t* dst = (p.buf ? a0_dst : a1_dst) + p.begin;   // the OTHER buffer
  const int base = c * kPartChunk;
  const int right_before = base - before;
  const bool with_grads = ga.arena[0] != nullptr && t.feature >= 0;
  const int64_t* gh_src = with_grads ? ga.arena[p.buf] + p.begin : nullptr;
  int64_t* gh_dst = with_grads ? ga.arena[1 - p.buf] + p.begin : nullptr;
#pragma unroll
  for (int kk = 0; kk < kPartChunk / kPartThreads; ++kk) {
    int w = kk * (kPartThreads / kWave) + wave;
    int pos = base + w * kWave + lane;
    if (pos < p.count) {
      unsigned long long m = mask[(int64_t)wg * kWords + w];
      bool left = (m >> lane) & 1ull;
      int lrank = word_prefix[w] + __popcll(m & ((1ull << lane) - 1ull));
      int in_chunk = w * kWave + lane;
      int d = left ? (before + lrank) : (all + right_before + (in_chunk - lrank));
      if (LGA_DEV_CHECK((unsigned)d < (unsigned)p.count, kCheckPartitionTarget, d, p.count)) {
        const int32_t row = src[pos];
        dst[d] = row;
        if (with_grads) gh_dst[d] = t.parent_node == 0 ? ga.by_row[row] : gh_src[pos];
      }
    }
  }
  if (c == 0 && threadIdx.x == 0) {
    nodes[t.left_node] = NodeRange{p.begin, all, 1 - p.buf, 0};
    nodes[t.right_node] = NodeRange{p.begin + all, p.count - all, 1 - p.buf, 0};
    task_left[ti] = all;
  }
}

// ---------------------------------------------------------------------------
// histogram pool addressing (kernels.hpp: PoolLayout)
// ---------------------------------------------------------------------------
struct SlotRef {
  int64_t* base;     // the slot's chunk of rank 0
  int64_t rstride;   // words between the chunks of consecutive ranks
};
// A slot created by the round `r`.
__device__ __forceinline__ SlotRef new_slot_ref(const PoolLayout& L, const RoundSlots& r, int slot) {
  return SlotRef{L.pool + (int64_t)r.first_slot * L.slot_words + (int64_t)(slot - r.first_slot) * L.chunk_words,
                 (int64_t)r.n_slots * L.chunk_words};
}
// A slot created by an earlier launch.
__device__ __forceinline__ SlotRef old_slot_ref(const PoolLayout& L, int slot) {
  return SlotRef{L.pool + L.slot_table[2 * slot], L.slot_table[2 * slot + 1]};
}
__device__ __forceinline__ void publish_slot(const PoolLayout& L, int slot, const SlotRef& s) {
  L.slot_table[2 * slot] = (int64_t)(s.base - L.pool);
  L.slot_table[2 * slot + 1] = s.rstride;
}
// Feature f's block: 256 gradient words, then 256 hessian words.
__device__ __forceinline__ int64_t* feat_block(const PoolLayout& L, const SlotRef& s, int f) {
  const int r = L.ranks > 1 ? f / L.fpr : 0;
  return s.base + (int64_t)r * s.rstride + (int64_t)(f - r * L.fpr) * (2 * kMaxBins);
}
// Rank r's copy of the slot header (rows, sum g, sum h).
__device__ __forceinline__ int64_t* slot_header(const PoolLayout& L, const SlotRef& s, int r) {
  return s.base + (int64_t)r * s.rstride + (int64_t)L.fpr * (2 * kMaxBins);
}

// ---- the workgroup histogram in LDS ----
// Cell (feature f, bin b) sits at index b * 32 + f: the bank of an 8-byte cell is its index mod 16 (mod 32 for
// reads), i.e. the FEATURE picks the bank, and the kernels make the lanes of a wave work on different features at
// any moment - lane l takes the features of its row in the order l, l + 1, ... (mod 32) - so the lanes of every
// 16-lane group of an LDS atomic hit 16 different banks whatever bins their rows hold: no bank conflicts and no
// same-address collisions (a feature with three distinct values does not queue 64 lanes on three words).
__device__ __forceinline__ int hist_cell(int f, int bin) { return (bin << 5) + f; }
// WIDE (full precision): two planes of 64-bit cells, gradients at [0, kWidePlane), hessians behind them.
constexpr int kWidePlane = kTileFeatures * kMaxBins;

// The 32 bin bytes of a record (8 dwords) rotated left by d bytes: out byte j = in byte (j + d) mod 32.  A lane
// rotates its row's record so that byte j holds the bin of feature (j + lane) mod 32; the loops over j then use
// static register indices and shifts.
__device__ __forceinline__ void rotate_record(unsigned (&w)[8], unsigned d) {
  // three conditional stages (rotate by 4, 2, 1 dwords: 8 selects each), then one byte alignment per dword.
  // Written out on scalars with constant indices: as loops over the arrays the compiler turns "select of two loads"
  // into "load at a selected index" before it unrolls, and the arrays end up indexed dynamically - an eight-way
  // compare-and-select chain per dword, four times the instructions.
  const bool r4 = (d & 16u) != 0, r2 = (d & 8u) != 0, r1 = (d & 4u) != 0;
  const unsigned a0 = w[0], a1 = w[1], a2 = w[2], a3 = w[3], a4 = w[4], a5 = w[5], a6 = w[6], a7 = w[7];
  const unsigned b0 = r4 ? a4 : a0, b1 = r4 ? a5 : a1, b2 = r4 ? a6 : a2, b3 = r4 ? a7 : a3;
  const unsigned b4 = r4 ? a0 : a4, b5 = r4 ? a1 : a5, b6 = r4 ? a2 : a6, b7 = r4 ? a3 : a7;
  const unsigned c0 = r2 ? b2 : b0, c1 = r2 ? b3 : b1, c2 = r2 ? b4 : b2, c3 = r2 ? b5 : b3;
  const unsigned c4 = r2 ? b6 : b4, c5 = r2 ? b7 : b5, c6 = r2 ? b0 : b6, c7 = r2 ? b1 : b7;
  const unsigned e0 = r1 ? c1 : c0, e1 = r1 ? c2 : c1, e2 = r1 ? c3 : c2, e3 = r1 ? c4 : c3;
  const unsigned e4 = r1 ? c5 : c4, e5 = r1 ? c6 : c5, e6 = r1 ? c7 : c6, e7 = r1 ? c0 : c7;
  const unsigned s = d & 3u;
  w[0] = __builtin_amdgcn_alignbyte(e1, e0, s);
  w[1] = __builtin_amdgcn_alignbyte(e2, e1, s);
  w[2] = __builtin_amdgcn_alignbyte(e3, e2, s);
  w[3] = __builtin_amdgcn_alignbyte(e4, e3, s);
  w[4] = __builtin_amdgcn_alignbyte(e5, e4, s);
  w[5] = __builtin_amdgcn_alignbyte(e6, e5, s);
  w[6] = __builtin_amdgcn_alignbyte(e7, e6, s);
  w[7] = __builtin_amdgcn_alignbyte(e0, e7, s);
}

// Item i of a flush -> (feature, bin): the 64 lanes of a wave take an 8 x 8 tile (8 features x 8 consecutive bins),
// so their LDS reads spread over 8 banks (4-way instead of 32-way for one feature's consecutive bins) and their
// global accesses still form eight 64-byte segments.
__device__ __forceinline__ void flush_item(int i, int& f, int& b) {
  const int lo = i & 63, hi = i >> 6;
  f = ((hi & 3) << 3) + (lo & 7);
  b = ((hi >> 2) << 3) + (lo >> 3);
}

// Adds a workgroup's LDS histogram (features f0 .. f0 + nf - 1; packed g:int32 | h:uint32 cells, or - WIDE - two planes
// of 64-bit cells) to a slot.
// `stagger` (any number that differs between the workgroups of a slot, e.g. the unit index) shifts where a workgroup
// starts: the units of one leaf finish at about the same time, and without it all of them would add to the same
// few cache lines at any moment.
template <bool WIDE>
__device__ __forceinline__ void flush_lds_hist(const unsigned long long* lds, int nf, int f0, const PoolLayout& L,
                                               const SlotRef& s, int nthreads, unsigned stagger = 0) {
  constexpr int kItems = kTileFeatures * kMaxBins;
  const int shift = (int)((stagger * 1031u) & (unsigned)(kItems - 1)) & ~63;
  for (int i0 = threadIdx.x; i0 < kItems; i0 += nthreads) {
    int fe, b;
    flush_item((i0 + shift) & (kItems - 1), fe, b);
    if (fe >= nf) continue;
    const int cell = hist_cell(fe, b);
    if (WIDE) {
      const unsigned long long vg = lds[cell], vh = lds[kWidePlane + cell];
      if ((vg | vh) != 0ull) {
        unsigned long long* blk = reinterpret_cast<unsigned long long*>(feat_block(L, s, f0 + fe));
        if (vg != 0ull) atomicAdd(&blk[b], vg);
        if (vh != 0ull) atomicAdd(&blk[kMaxBins + b], vh);
      }
    } else {
      const unsigned long long v = lds[cell];
      if (v != 0ull) {
        unsigned long long* blk = reinterpret_cast<unsigned long long*>(feat_block(L, s, f0 + fe));
        atomicAdd(&blk[b], (unsigned long long)(long long)(int32_t)(v >> 32));
        atomicAdd(&blk[kMaxBins + b], v & 0xffffffffull);
      }
    }
  }
}
__device__ __forceinline__ void add_slot_header(const PoolLayout& L, const SlotRef& s, long long rows, long long tg, long long th) {
  for (int r = 0; r < L.ranks; ++r) {
    unsigned long long* hd = reinterpret_cast<unsigned long long*>(slot_header(L, s, r));
    atomicAdd(&hd[0], (unsigned long long)rows);
    atomicAdd(&hd[1], (unsigned long long)tg);
    atomicAdd(&hd[2], (unsigned long long)th);
  }
}

// ---------------------------------------------------------------------------
// histogram (feature-tile records + separate gradient array)
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned div3(unsigned r) { return (unsigned)(((unsigned long long)r * 0xAAAAAAABull) >> 33); }

struct QuantSource {
  const float* g;
  const float* h;
  const uint32_t* maxbits;
  QuantScale* qs_out;
};

template <bool WIDE>
__global__ __launch_bounds__(WIDE ? kHistThreadsWide : kHistThreads) void hist_kernel(
    const SplitTask* __restrict__ tasks, const int32_t* __restrict__ chunk_task, InlineTasks inl, int rows_per_chunk,
    const NodeRange* __restrict__ nodes, const int32_t* __restrict__ a0, const int32_t* __restrict__ a1,
    const uint8_t* __restrict__ bins_rt, int64_t N, const int64_t* __restrict__ packed, PoolLayout L, RoundSlots round,
    int F, int tile_first, int tile_step, const QuantScale* __restrict__ qs, GradArenas ga) {
  constexpr int kThreads = WIDE ? kHistThreadsWide : kHistThreads;
  __shared__ unsigned long long lds[(WIDE ? 2 : 1) * kTileFeatures * kMaxBins];   // 64 KiB (WIDE: 128 KiB, one workgroup of 16 waves per CU)
  __shared__ long long red[2][kThreads / kWave];
  const int x = blockIdx.x;
  const int tile = tile_first + (int)blockIdx.y * tile_step;
  const int ti = tasks ? chunk_task[x] : find_task<false>(inl, x);
  const SplitTask t = load_task(tasks, inl, ti);
  const int c = x - t.first_hchunk;
  const int node = t.hist_child ? t.right_node : t.left_node;
  const NodeRange r = nodes[node];
  const SlotRef slot = new_slot_ref(L, round, t.new_slot);
  if (c == 0 && blockIdx.y == 0 && threadIdx.x == 0) publish_slot(L, t.new_slot, slot);
  const int lo = c * rows_per_chunk;
  if (lo >= r.count) return;   // the child turned out smaller than the grid was sized for
  const int hi = min(lo + rows_per_chunk, r.count);
  const int nf = min(kTileFeatures, F - tile * kTileFeatures);
  for (int i = threadIdx.x; i < kTileFeatures * kMaxBins; i += kThreads) {
    lds[i] = 0ull;
    if (WIDE) lds[kWidePlane + i] = 0ull;
  }
  const unsigned lane_rot = threadIdx.x & 31u;   // this lane takes features lane_rot, lane_rot + 1, ... (mod 32)
  int wkg = 0, wkh = 0;
  if (WIDE) { wkg = qs->kg; wkh = qs->kh; }
  __syncthreads();
  const int32_t* rows = (r.buf ? a1 : a0) + r.begin;
  // the child's gradients in the order of its rows (GradArenas), or by row id
  const int64_t* ordered = (ga.arena[0] != nullptr && node != 0) ? ga.arena[r.buf] + r.begin : nullptr;
  const uint4* recs = reinterpret_cast<const uint4*>(bins_rt + (int64_t)tile * N * kTileFeatures);
  long long sg = 0, sh = 0;
  // U rows per lane and trip: every load of a trip is issued before its LDS atomics, so 4 waves per
  // SIMD keep ~U x 3 memory requests per lane in flight.
  constexpr int U = 4;
  // (the row ids of trip i + 1 are requested before trip i's records: one memory round trip per trip instead of two)
  int nxt[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = lo + (int)threadIdx.x + u * kThreads;
    nxt[u] = i < hi ? rows[i] : -1;
  }
  for (int i0 = lo + threadIdx.x; i0 < hi; i0 += kThreads * U) {
    int row[U];
    unsigned long long p[U];
    uint4 ra[U], rb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * kThreads;
      row[u] = nxt[u];
      const int in = i + kThreads * U;
      nxt[u] = in < hi ? rows[in] : -1;
      if (i < hi && !LGA_DEV_CHECK((unsigned)row[u] < (unsigned long long)N, kCheckHistRow, row[u], N)) row[u] = -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (row[u] >= 0) {
        p[u] = (unsigned long long)(ordered ? ordered[i0 + u * kThreads] : packed[row[u]]);
        ra[u] = recs[(int64_t)row[u] * 2];
        rb[u] = recs[(int64_t)row[u] * 2 + 1];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (row[u] >= 0) {
        long long qg = 0, qh = 0;
        if (WIDE) {
          wide_of(p[u], wkg, wkh, qg, qh);
          sg += qg;
          sh += qh;
        } else {
          sg += (long long)(int32_t)(p[u] >> 32);
          sh += (long long)(p[u] & 0xffffffffull);
        }
        unsigned w[8] = {ra[u].x, ra[u].y, ra[u].z, ra[u].w, rb[u].x, rb[u].y, rb[u].z, rb[u].w};
        rotate_record(w, lane_rot);   // byte j = the bin of feature (j + lane) mod 32
        auto add_features = [&](auto allf) {
#pragma unroll
          for (int j = 0; j < kTileFeatures; ++j) {
            const unsigned f = (j + lane_rot) & 31u;
            if (decltype(allf)::value || f < (unsigned)nf) {
              const unsigned bin = (w[j >> 2] >> ((j & 3) * 8)) & 255u;
              const unsigned cell = (bin << 5) + f;
              if (WIDE) {
                atomicAdd(&lds[cell], (unsigned long long)qg);
                atomicAdd(&lds[kWidePlane + cell], (unsigned long long)qh);
              } else {
                atomicAdd(&lds[cell], p[u]);
              }
            }
          }
        };
        // a nearly full tile skips the per-step "does this lane's feature exist" test (see grow_kernel's ALLF); two
        // copies of the loop behind one uniform branch - as a flag inside the loop the test stays per step
        if (nf > 24) add_features(std::true_type{});
        else add_features(std::false_type{});
      }
    }
  }
  __syncthreads();
  // A child that fits one chunk has this workgroup as the only writer of the tile's feature blocks: plain coalesced
  // stores instead of two global atomics per non-empty cell (rounds of many small leaves on wide data were bound by
  // those atomics: 16 k of them per workgroup whatever the row count).
  const bool own_blocks = t.num_hchunk == 1;
  if (own_blocks) {
    for (int i = threadIdx.x; i < kTileFeatures * kMaxBins; i += kThreads) {
      int fe, b;
      flush_item(i, fe, b);
      if (fe >= nf) continue;
      const int cell = hist_cell(fe, b);
      int64_t* blk = feat_block(L, slot, tile * kTileFeatures + fe);
      if (WIDE) {
        blk[b] = (int64_t)lds[cell];
        blk[kMaxBins + b] = (int64_t)lds[kWidePlane + cell];
      } else {
        const unsigned long long v = lds[cell];
        blk[b] = (int64_t)(int32_t)(v >> 32);
        blk[kMaxBins + b] = (int64_t)(v & 0xffffffffull);
      }
    }
  } else {
    flush_lds_hist<WIDE>(lds, nf, tile * kTileFeatures, L, slot, kThreads);
  }
  if (blockIdx.y == 0) {
    // header: rows, sum g, sum h of this child (the first tile's workgroups see every row once)
    sg = wave_sum_i64(sg);
    sh = wave_sum_i64(sh);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = sg; red[1][wave] = sh; }
    __syncthreads();
    if (threadIdx.x == 0) {
      long long tg = 0, th = 0;
#pragma unroll
      for (int w = 0; w < kThreads / kWave; ++w) { tg += red[0][w]; th += red[1][w]; }
      if (own_blocks) {
        for (int rk = 0; rk < L.ranks; ++rk) {
          int64_t* hdw = slot_header(L, slot, rk);
          hdw[0] = hi - lo; hdw[1] = tg; hdw[2] = th;
        }
      } else {
        add_slot_header(L, slot, hi - lo, tg, th);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// split scan
// ---------------------------------------------------------------------------

// Split search as its own launch: workgroup (s, b) searches feature block b of scan task s (one wave per
// feature at a time), the workgroup that finishes a task last merges the blocks' winners.
__global__ __launch_bounds__(512) void scan_blocks_kernel(InlineTasks inl, ScanArgs a) {
  __shared__ Cand wbest[8];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int ti = blockIdx.x >> 1, side = blockIdx.x & 1;
  const SplitTask t = load_task(a.tasks, inl, ti);
  const int s = 2 * a.task0 + blockIdx.x;   // index among the round's scan tasks
  const bool direct = a.root || side == t.hist_child;
  const int node = a.root ? 0 : (side == 0 ? t.left_node : t.right_node);
  const uint32_t key = a.root ? kRootNodeKey : child_node_key((uint32_t)t.parent_key, side);
  const SlotRef fresh = new_slot_ref(a.pool, a.round, t.new_slot);
  // Everything the search needs from memory is requested before the first wait: the child's totals (and the parent's),
  // the scales, the parent's slot address and the first feature's description, then that feature's cells - the leaf
  // context's fp64 arithmetic runs while the cells are on their way.  In program order (totals, context, description,
  // slot address, cells) a small round's search was four dependent memory round trips long; this way it is two.
  const int group = a.range.group_first + b * a.range.group_step;
  const int f0 = a.range.f_begin + group * a.range.group_size;
  const int f1 = min(f0 + a.range.group_size, a.range.f_end);
  const int f_first = f0 + wave;
  const int64_t* hd = slot_header(a.pool, fresh, a.my_rank);
  int64_t tc = hd[0], tg = hd[1], th = hd[2];
  int64_t pc = 0, pg = 0, ph = 0;
  if (!direct) {
    const int64_t* pt = a.node_totals + (int64_t)t.parent_node * 4;
    pc = pt[0]; pg = pt[1]; ph = pt[2];
  }
  const QuantScale q = *a.qs;
  const SlotRef self = direct ? fresh : old_slot_ref(a.pool, t.parent_slot);
  FeatureMeta m_first{}, m_next{};
  if (f_first < f1) m_first = a.meta[f_first];
  if (f_first + 8 < f1) m_next = a.meta[f_first + 8];
  int64_t g[4], h[4], cg[4], ch[4];
  const bool first_loaded = f_first < f1 && m_first.enabled;
  if (first_loaded) {
    const int bpl = bins_per_lane(m_first.num_bin);
    load_feat_cells(feat_block(a.pool, self, f_first), lane, bpl, g, h);
    if (!direct) load_feat_cells(feat_block(a.pool, fresh, f_first), lane, bpl, cg, ch);
  }
  if (!direct) { tc = pc - tc; tg = pg - tg; th = ph - th; }
  const LeafScanCtx ctx = node_scan_ctx(tg, th, tc, q, a.sp, node);
  const double ninf = -std::numeric_limits<double>::infinity();
  double bg = ninf;
  uint64_t bkey = ~0ull;
  int64_t blg = 0, blh = 0;
  // A wave that takes several features in turn (rounds of many leaves: few, long workgroups) requests the NEXT feature's
  // cells before it works on the current one, and the description of the one after: one memory round trip per wave
  // instead of one per feature.
  FeatureMeta m = m_first;
  for (int f = f_first; f < f1; f += 8) {
    const int fn = f + 8;
    const FeatureMeta mn = m_next;
    int64_t ng[4] = {0, 0, 0, 0}, nh[4] = {0, 0, 0, 0}, ncg[4] = {0, 0, 0, 0}, nch[4] = {0, 0, 0, 0};
    if (fn < f1 && mn.enabled) {
      const int nbpl = bins_per_lane(mn.num_bin);
      load_feat_cells(feat_block(a.pool, self, fn), lane, nbpl, ng, nh);
      if (!direct) load_feat_cells(feat_block(a.pool, fresh, fn), lane, nbpl, ncg, nch);
    }
    if (fn + 8 < f1) m_next = a.meta[fn + 8];
    if (m.enabled) {
      int64_t* blk = feat_block(a.pool, self, f);
      const int bpl = bins_per_lane(m.num_bin);
      if (!direct) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { g[j] -= cg[j]; h[j] -= ch[j]; }
        store_feat_cells(blk, lane, bpl, g, h);
      }
      if (node_feature_on(a.sp, key, f, a.F) && node_allows(a.sp, node, f) && m.categorical != kCategoricalSorted)
        scan_cells_wave(f, lane, bpl, g, h, m, ctx, a.sp, bg, bkey, blg, blh, key);
    }
    m = mn;
#pragma unroll
    for (int j = 0; j < 4; ++j) { g[j] = ng[j]; h[j] = nh[j]; cg[j] = ncg[j]; ch[j] = nch[j]; }
  }
  wave_best(bg, bkey, blg, blh);
  if (lane == 0) wbest[wave] = Cand{bg, bkey, blg, blh};
  __syncthreads();
  if (threadIdx.x == 0) {
    s_last = 0;
    for (int w = 1; w < 8; ++w) take_better(bg, bkey, blg, blh, wbest[w].gain, wbest[w].key, wbest[w].lg, wbest[w].lh);
    const int nb = a.range.n_blocks;
    if (b == 0) {
      int64_t* nt = a.node_totals + (int64_t)node * 4;
      nt[0] = tc; nt[1] = tg; nt[2] = th; nt[3] = 0;
      int64_t* st = a.scan_totals + (int64_t)s * 4;
      st[0] = tc; st[1] = tg; st[2] = th; st[3] = 0;
    }
    bool have_winner = true;
    if (a.direct == 2) {
      BlockResult* o = reinterpret_cast<BlockResult*>(a.host_blocks) + ((int64_t)s * nb + b);
      o->gain = bg; o->key = bkey; o->lg = blg; o->lh = blh;
      o->tc = tc; o->tg = tg; o->th = th;
      o->left = (b == 0 && side == 0 && !a.root) ? a.task_left[a.task0 + ti] : 0;
      __threadfence_system();
      *reinterpret_cast<volatile int32_t*>(&o->tag) = (int32_t)a.seq;
      have_winner = false;
    } else if (nb > 1) {
      a.block_best[(int64_t)s * nb + b] = Cand{bg, bkey, blg, blh};
      have_winner = release_and_ticket(&a.tickets[s]) == (unsigned)(nb - 1);
      if (have_winner) {
        acquire_after_ticket();
        a.tickets[s] = 0u;
        bg = ninf; bkey = ~0ull; blg = 0; blh = 0;
        for (int i = 0; i < nb; ++i) {
          const Cand c = a.block_best[(int64_t)s * nb + i];
          take_better(bg, bkey, blg, blh, c.gain, c.key, c.lg, c.lh);
        }
      }
    }
    if (have_winner) {
      if (a.best_out) {
        a.best_out[s] = Cand{bg, bkey, blg, blh};   // distributed: all-gather + merge_winners follow
      } else if (a.direct) {
        SplitInfo* o = reinterpret_cast<SplitInfo*>(a.host_out + kHostDataOffset) + s;
        write_split(o, bg, bkey, blg, blh, tc, tg, th, q, a.sp);
        if (side == 0) {   // the left child's record travels with the task's LOCAL left count
          int32_t* tl = reinterpret_cast<int32_t*>(a.host_out + kHostDataOffset + sizeof(SplitInfo) * (size_t)a.n_scan);
          tl[a.task0 + ti] = a.task_left[a.task0 + ti];
        }
        __threadfence_system();
        *reinterpret_cast<volatile int32_t*>(&o->pad0) = (int32_t)a.seq;
      } else {
        write_split(&a.results[s], bg, bkey, blg, blh, tc, tg, th, q, a.sp);
        if (release_and_ticket(a.done_counter) == (unsigned)(a.n_scan - 1)) {
          acquire_after_ticket();
          *a.done_counter = 0u;
          s_last = 1;
        }
      }
    }
  }
  __syncthreads();
  if (!s_last) return;
  // the workgroup that finished the round's last search hands everything to the host in one go: results,
  // left counts, quantisation scales, then - behind one system-scope fence - the flag the host polls
  {
    const int n_left = a.root ? 0 : a.n_tasks;
    int64_t* out = reinterpret_cast<int64_t*>(a.host_out + kHostDataOffset);
    const int64_t* res = reinterpret_cast<const int64_t*>(a.results);
    const int words = a.n_scan * (int)(sizeof(SplitInfo) / 8);
    for (int i = threadIdx.x; i < words; i += 512) out[i] = res[i];
    int32_t* tl = reinterpret_cast<int32_t*>(a.host_out + kHostDataOffset + sizeof(SplitInfo) * (size_t)a.n_scan);
    for (int i = threadIdx.x; i < n_left; i += 512) tl[i] = a.task_left[i];
    if (threadIdx.x == 0) *reinterpret_cast<QuantScale*>(tl + n_left) = q;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *reinterpret_cast<volatile unsigned*>(a.host_out) = a.seq;
  }
}

// The same search with BOTH children of a split in one workgroup (wide data, where the search is bound by the
// histogram bytes it moves): per feature the fresh child's row and the parent's row are read once, the sibling is
// derived in registers, written to the parent's slot (which it inherits) and both are searched - 3 slot-rows of
// traffic per split and feature instead of 4 (the one-child form reads the fresh row twice).  Workgroup (t, b) takes
// feature block b of task t; its results are scan tasks 2t + hist_child (the fresh child) and 2t + 1 - hist_child.
__global__ __launch_bounds__(512) void scan_pairs_kernel(InlineTasks inl, ScanArgs a) {
  __shared__ Cand wbest[2][8];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y;
  const int ti = blockIdx.x;
  const SplitTask t = load_task(a.tasks, inl, ti);
  const SlotRef fresh = new_slot_ref(a.pool, a.round, t.new_slot);
  const SlotRef parent = old_slot_ref(a.pool, t.parent_slot);
  // k = 0: the child whose histogram was built; k = 1: its sibling
  int64_t tc[2], tg[2], th[2];
  int node[2], sidx[2];
  uint32_t key[2];
  {
    const int64_t* hd = slot_header(a.pool, fresh, a.my_rank);
    const int64_t* pt = a.node_totals + (int64_t)t.parent_node * 4;
    tc[0] = hd[0]; tg[0] = hd[1]; th[0] = hd[2];
    tc[1] = pt[0] - tc[0]; tg[1] = pt[1] - tg[0]; th[1] = pt[2] - th[0];
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int side = k == 0 ? t.hist_child : 1 - t.hist_child;
    node[k] = side == 0 ? t.left_node : t.right_node;
    key[k] = child_node_key((uint32_t)t.parent_key, side);
    sidx[k] = 2 * (a.task0 + ti) + side;
  }
  const QuantScale q = *a.qs;
  const LeafScanCtx ctx0 = node_scan_ctx(tg[0], th[0], tc[0], q, a.sp, node[0]);
  const LeafScanCtx ctx1 = node_scan_ctx(tg[1], th[1], tc[1], q, a.sp, node[1]);
  const double ninf = -std::numeric_limits<double>::infinity();
  double bg[2] = {ninf, ninf};
  uint64_t bkey[2] = {~0ull, ~0ull};
  int64_t blg[2] = {0, 0}, blh[2] = {0, 0};
  const int group = a.range.group_first + b * a.range.group_step;
  const int f0 = a.range.f_begin + group * a.range.group_size;
  const int f1 = min(f0 + a.range.group_size, a.range.f_end);
  for (int f = f0 + wave; f < f1; f += 8) {
    const FeatureMeta m = a.meta[f];
    if (!m.enabled) continue;
    const int bpl = bins_per_lane(m.num_bin);
    int64_t cg[4], ch[4], sg[4], sh[4];
    int64_t* pblk = feat_block(a.pool, parent, f);
    load_feat_cells(feat_block(a.pool, fresh, f), lane, bpl, cg, ch);
    load_feat_cells(pblk, lane, bpl, sg, sh);
#pragma unroll
    for (int j = 0; j < 4; ++j) { sg[j] -= cg[j]; sh[j] -= ch[j]; }
    store_feat_cells(pblk, lane, bpl, sg, sh);
    if (m.categorical == kCategoricalSorted) continue;
    if (node_feature_on(a.sp, key[0], f, a.F) && node_allows(a.sp, node[0], f))
      scan_cells_wave(f, lane, bpl, cg, ch, m, ctx0, a.sp, bg[0], bkey[0], blg[0], blh[0], key[0]);
    if (node_feature_on(a.sp, key[1], f, a.F) && node_allows(a.sp, node[1], f))
      scan_cells_wave(f, lane, bpl, sg, sh, m, ctx1, a.sp, bg[1], bkey[1], blg[1], blh[1], key[1]);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    wave_best(bg[k], bkey[k], blg[k], blh[k]);
    if (lane == 0) wbest[k][wave] = Cand{bg[k], bkey[k], blg[k], blh[k]};
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s_last = 0;
    const int nb = a.range.n_blocks;
    for (int k = 0; k < 2; ++k) {
      for (int w = 1; w < 8; ++w) take_better(bg[k], bkey[k], blg[k], blh[k], wbest[k][w].gain, wbest[k][w].key, wbest[k][w].lg, wbest[k][w].lh);
      if (b == 0) {
        int64_t* nt = a.node_totals + (int64_t)node[k] * 4;
        nt[0] = tc[k]; nt[1] = tg[k]; nt[2] = th[k]; nt[3] = 0;
        int64_t* st = a.scan_totals + (int64_t)sidx[k] * 4;
        st[0] = tc[k]; st[1] = tg[k]; st[2] = th[k]; st[3] = 0;
      }
      if (nb > 1) a.block_best[(int64_t)sidx[k] * nb + b] = Cand{bg[k], bkey[k], blg[k], blh[k]};
    }
    bool have_winner = true;
    if (nb > 1) {
      // one ticket per task serves both children (they always arrive together)
      have_winner = release_and_ticket(&a.tickets[sidx[0]]) == (unsigned)(nb - 1);
      if (have_winner) {
        acquire_after_ticket();
        a.tickets[sidx[0]] = 0u;
        for (int k = 0; k < 2; ++k) {
          bg[k] = ninf; bkey[k] = ~0ull; blg[k] = 0; blh[k] = 0;
          for (int i = 0; i < nb; ++i) {
            const Cand c = a.block_best[(int64_t)sidx[k] * nb + i];
            take_better(bg[k], bkey[k], blg[k], blh[k], c.gain, c.key, c.lg, c.lh);
          }
        }
      }
    }
    if (have_winner) {
      if (a.best_out) {
        for (int k = 0; k < 2; ++k) a.best_out[sidx[k]] = Cand{bg[k], bkey[k], blg[k], blh[k]};   // distributed: all-gather + merge_winners follow
      } else if (a.direct) {
        SplitInfo* o = reinterpret_cast<SplitInfo*>(a.host_out + kHostDataOffset);
        for (int k = 0; k < 2; ++k) write_split(&o[sidx[k]], bg[k], bkey[k], blg[k], blh[k], tc[k], tg[k], th[k], q, a.sp);
        {
          int32_t* tl = reinterpret_cast<int32_t*>(a.host_out + kHostDataOffset + sizeof(SplitInfo) * (size_t)a.n_scan);
          tl[a.task0 + ti] = a.task_left[a.task0 + ti];
        }
        __threadfence_system();
        for (int k = 0; k < 2; ++k) *reinterpret_cast<volatile int32_t*>(&o[sidx[k]].pad0) = (int32_t)a.seq;
      } else {
        for (int k = 0; k < 2; ++k) write_split(&a.results[sidx[k]], bg[k], bkey[k], blg[k], blh[k], tc[k], tg[k], th[k], q, a.sp);
        if (release_and_ticket(a.done_counter) == (unsigned)(a.n_tasks - 1)) {
          acquire_after_ticket();
          *a.done_counter = 0u;
          s_last = 1;
        }
      }
    }
  }
  __syncthreads();
  if (!s_last) return;
  {
    int64_t* out = reinterpret_cast<int64_t*>(a.host_out + kHostDataOffset);
    const int64_t* res = reinterpret_cast<const int64_t*>(a.results);
    const int words = a.n_scan * (int)(sizeof(SplitInfo) / 8);
    for (int i = threadIdx.x; i < words; i += 512) out[i] = res[i];
    int32_t* tl = reinterpret_cast<int32_t*>(a.host_out + kHostDataOffset + sizeof(SplitInfo) * (size_t)a.n_scan);
    for (int i = threadIdx.x; i < a.n_tasks; i += 512) tl[i] = a.task_left[i];
    if (threadIdx.x == 0) *reinterpret_cast<QuantScale*>(tl + a.n_tasks) = q;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *reinterpret_cast<volatile unsigned*>(a.host_out) = a.seq;
  }
}

// Distributed learners: every rank's winners have been gathered ([rank][scan task]); each rank merges them the
// same way and hands the results to its host.  One workgroup.
__global__ __launch_bounds__(256) void merge_winners_kernel(const Cand* __restrict__ gathered, int ranks, int n_scan,
                                                            int n_left, const int64_t* __restrict__ scan_totals,
                                                            const QuantScale* __restrict__ qs, SplitParams sp,
                                                            const int32_t* __restrict__ task_left, char* host_out,
                                                            unsigned seq) {
  const QuantScale q = *qs;
  SplitInfo* out = reinterpret_cast<SplitInfo*>(host_out + kHostDataOffset);
  int32_t* tl = reinterpret_cast<int32_t*>(host_out + kHostDataOffset + sizeof(SplitInfo) * (size_t)n_scan);
  for (int s = threadIdx.x; s < n_scan; s += blockDim.x) {
    Cand b = gathered[s];
    for (int r = 1; r < ranks; ++r) {
      const Cand c = gathered[(int64_t)r * n_scan + s];
      if (cand_better(c.gain, c.key, b.gain, b.key)) b = c;
    }
    const int64_t* st = scan_totals + (int64_t)s * 4;
    write_split(&out[s], b.gain, b.key, b.lg, b.lh, st[0], st[1], st[2], q, sp);
  }
  for (int i = threadIdx.x; i < n_left; i += blockDim.x) tl[i] = task_left[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    QuantScale* qo = reinterpret_cast<QuantScale*>(tl + n_left);
    *qo = q;
    __threadfence_system();
    *reinterpret_cast<volatile unsigned*>(host_out) = seq;
  }
}

// ---------------------------------------------------------------------------
// voting-parallel search (kernels.hpp: Vote; the host form is host_engine.cpp: vote_and_search)
// ---------------------------------------------------------------------------
// What a searched child is, for every kernel of the vote: its node, its key, the slot that holds (or will hold) its
// histogram, and whether that histogram was built (direct) or is the parent's minus the built sibling's.
struct VoteChild {
  SplitTask t;
  int node;
  uint32_t key;
  bool direct;
  SlotRef fresh, self;
};
__device__ __forceinline__ VoteChild vote_child(const ScanArgs& a, const InlineTasks& inl, int s) {
  VoteChild c;
  const int ti = s >> 1, side = s & 1;
  c.t = load_task(a.tasks, inl, a.root ? 0 : ti);
  c.direct = a.root || side == c.t.hist_child;
  c.node = a.root ? 0 : (side == 0 ? c.t.left_node : c.t.right_node);
  c.key = a.root ? kRootNodeKey : child_node_key((uint32_t)c.t.parent_key, side);
  c.fresh = new_slot_ref(a.pool, a.round, c.t.new_slot);
  c.self = c.direct ? c.fresh : old_slot_ref(a.pool, c.t.parent_slot);
  return c;
}

__global__ __launch_bounds__(512) void vote_gains_kernel(InlineTasks inl, ScanArgs a, double* __restrict__ gains) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.x, b = blockIdx.y;
  const VoteChild c = vote_child(a, inl, s);
  const int64_t* hd = slot_header(a.pool, c.fresh, 0);
  int64_t tc = hd[0], tg = hd[1], th = hd[2];
  if (!c.direct) {
    const int64_t* pt = a.node_totals + (int64_t)c.t.parent_node * 4;
    tc = pt[0] - tc; tg = pt[1] - tg; th = pt[2] - th;
  }
  const QuantScale q = *a.qs;
  const LeafScanCtx ctx = node_scan_ctx(tg, th, tc, q, a.sp, c.node);
  const double ninf = -std::numeric_limits<double>::infinity();
  const int f0 = a.range.f_begin + b * a.range.group_size;
  const int f1 = min(f0 + a.range.group_size, a.range.f_end);
  for (int f = f0 + wave; f < f1; f += 8) {
    const FeatureMeta m = a.meta[f];
    double bg = ninf;
    uint64_t bkey = ~0ull;
    int64_t blg = 0, blh = 0;
    if (m.enabled) {
      int64_t* blk = feat_block(a.pool, c.self, f);
      const int bpl = bins_per_lane(m.num_bin);
      int64_t g[4], h[4];
      load_feat_cells(blk, lane, bpl, g, h);
      if (!c.direct) {
        int64_t cg[4], ch[4];
        load_feat_cells(feat_block(a.pool, c.fresh, f), lane, bpl, cg, ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) { g[j] -= cg[j]; h[j] -= ch[j]; }
        store_feat_cells(blk, lane, bpl, g, h);
      }
      if (node_feature_on(a.sp, c.key, f, a.F) && node_allows(a.sp, c.node, f) && m.categorical != kCategoricalSorted)
        scan_cells_wave(f, lane, bpl, g, h, m, ctx, a.sp, bg, bkey, blg, blh, c.key);
      wave_best(bg, bkey, blg, blh);
    }
    if (lane == 0) gains[(int64_t)s * a.F + f] = bg;
  }
  if (b == 0 && threadIdx.x == 0) {
    int64_t* nt = a.node_totals + (int64_t)c.node * 4;   // LOCAL totals: what the children's siblings are derived from
    nt[0] = tc; nt[1] = tg; nt[2] = th; nt[3] = 0;
    int64_t* st = a.scan_totals + (int64_t)s * 4;
    st[0] = tc; st[1] = tg; st[2] = th; st[3] = 0;
  }
}

// One wave per searched child: pick j is the best feature strictly AFTER pick j - 1 in the order (gain descending,
// index ascending), so nothing has to be marked as taken.
__global__ __launch_bounds__(64) void vote_topk_kernel(const double* __restrict__ gains, int F, int top_k, Vote* __restrict__ votes) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const double* g = gains + (int64_t)s * F;
  const double ninf = -std::numeric_limits<double>::infinity();
  double pg = std::numeric_limits<double>::infinity();
  int pf = -1;
  bool dry = false;
  for (int j = 0; j < top_k; ++j) {
    double bg = ninf;
    int bf = INT_MAX;
    if (!dry) {
      for (int f = lane; f < F; f += 64) {
        const double v = g[f];
        if (!(v > ninf)) continue;
        if (!(v < pg || (v == pg && f > pf))) continue;
        if (v > bg || (v == bg && f < bf)) { bg = v; bf = f; }
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        const double og = shfl_xor_f64(bg, d);
        const int of = __shfl_xor(bf, d, kWave);
        if (og > bg || (og == bg && of < bf)) { bg = og; bf = of; }
      }
      dry = bf == INT_MAX;
    }
    if (lane == 0) votes[(int64_t)s * top_k + j] = dry ? Vote{-1, 0.0f} : Vote{bf, (float)bg};
    pg = bg; pf = bf;
  }
}

__global__ __launch_bounds__(256) void vote_select_kernel(InlineTasks inl, ScanArgs a, const Vote* __restrict__ all, int ranks, int n_scan,
                                                          int top_k, int n_sel, int32_t* __restrict__ chosen, int64_t* __restrict__ buf) {
  __shared__ int count[kVoteMaxFeatures];
  __shared__ double weight[kVoteMaxFeatures];
  __shared__ int cand[kVoteMaxVotes];
  __shared__ int pick[kVoteMaxSelected];
  __shared__ int n_cand;
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  for (int f = tid; f < a.F; f += 256) { count[f] = 0; weight[f] = 0.0; }
  __syncthreads();
  if (tid == 0) {
    // rank by rank, vote by vote: the sums of the voted gains are floating point, their order is part of the result
    int nc = 0;
    for (int r = 0; r < ranks; ++r)
      for (int j = 0; j < top_k; ++j) {
        const Vote v = all[((int64_t)r * n_scan + s) * top_k + j];
        if (v.feature >= 0 && v.feature < a.F) {
          if (count[v.feature]++ == 0) cand[nc++] = v.feature;
          weight[v.feature] += (double)v.gain;
        }
      }
    n_cand = nc;
  }
  __syncthreads();
  if (tid < kWave) {
    // a before b: more votes, then the larger sum of gains, then the lower index
    auto before = [&](int x, int y) {
      if (count[x] != count[y]) return count[x] > count[y];
      if (weight[x] != weight[y]) return weight[x] > weight[y];
      return x < y;
    };
    int prev = -1;
    for (int j = 0; j < n_sel; ++j) {
      int bf = -1;
      for (int i = lane; i < n_cand; i += kWave) {
        const int f = cand[i];
        if (prev >= 0 && !before(prev, f)) continue;   // not after the previous pick (or the pick itself)
        if (bf < 0 || before(f, bf)) bf = f;
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        const int of = __shfl_xor(bf, d, kWave);
        if (of >= 0 && (bf < 0 || before(of, bf))) bf = of;
      }
      if (lane == 0) pick[j] = bf;
      if (bf < 0) {
        for (int jj = j + 1 + lane; jj < n_sel; jj += kWave) pick[jj] = -1;
        break;
      }
      prev = bf;
    }
  }
  __syncthreads();
  const VoteChild c = vote_child(a, inl, s);
  int64_t* dst = buf + (int64_t)s * vote_words_per_child(n_sel);
  if (tid < 4) dst[tid] = tid < 3 ? a.scan_totals[(int64_t)s * 4 + tid] : 0;
  for (int j = 0; j < n_sel; ++j) {
    const int f = pick[j];
    if (tid == 0) chosen[(int64_t)s * n_sel + j] = f;
    const int64_t* blk = f >= 0 ? feat_block(a.pool, c.self, f) : nullptr;
    for (int i = tid; i < 2 * kMaxBins; i += 256) dst[4 + (int64_t)j * 2 * kMaxBins + i] = blk ? blk[i] : 0;
  }
}

__global__ __launch_bounds__(512) void vote_final_kernel(InlineTasks inl, ScanArgs a, const int32_t* __restrict__ chosen,
                                                         const int64_t* __restrict__ buf, int n_sel, Cand* __restrict__ best) {
  __shared__ Cand wbest[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.x;
  const VoteChild c = vote_child(a, inl, s);
  const int64_t* src = buf + (int64_t)s * vote_words_per_child(n_sel);
  const int64_t tc = src[0], tg = src[1], th = src[2];   // summed over the ranks
  const QuantScale q = *a.qs;
  const LeafScanCtx ctx = node_scan_ctx(tg, th, tc, q, a.sp, c.node);
  const double ninf = -std::numeric_limits<double>::infinity();
  double bg = ninf;
  uint64_t bkey = ~0ull;
  int64_t blg = 0, blh = 0;
  for (int j = wave; j < n_sel; j += 8) {
    const int f = chosen[(int64_t)s * n_sel + j];
    if (f < 0) continue;
    const FeatureMeta m = a.meta[f];
    if (m.categorical == kCategoricalSorted) continue;
    const int bpl = bins_per_lane(m.num_bin);
    int64_t g[4], h[4];
    load_feat_cells(src + 4 + (int64_t)j * 2 * kMaxBins, lane, bpl, g, h);
    scan_cells_wave(f, lane, bpl, g, h, m, ctx, a.sp, bg, bkey, blg, blh, c.key);
  }
  wave_best(bg, bkey, blg, blh);
  if (lane == 0) wbest[wave] = Cand{bg, bkey, blg, blh};
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 8; ++w) take_better(bg, bkey, blg, blh, wbest[w].gain, wbest[w].key, wbest[w].lg, wbest[w].lh);
    best[s] = Cand{bg, bkey, blg, blh};
    int64_t* st = a.scan_totals + (int64_t)s * 4;
    st[0] = tc; st[1] = tg; st[2] = th; st[3] = 0;
  }
}

// ---------------------------------------------------------------------------
// fused growth round over row lines (kernels.hpp: GrowArgs)
// ---------------------------------------------------------------------------
// Row r's line slot holds its 32 bin bytes rotated by r: byte j = feature (j + r) mod 32.
__device__ __forceinline__ int line_bin(const unsigned (&w)[8], int feature, int32_t row) {
  const unsigned j = ((unsigned)feature - (unsigned)row) & 31u;
  const unsigned q = j >> 2;
  unsigned d = w[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) d = q == (unsigned)i ? w[i] : d;
  return (int)((d >> ((j & 3u) * 8)) & 255u);
}
__device__ __forceinline__ bool bin_goes_left(const SplitTask& t, int b, const uint64_t* __restrict__ cat_sets) {
  if (t.default_left == kCategoricalTask) return b == t.threshold;
  if (t.default_left == kCategoricalSetTask)
    return (cat_sets[(int64_t)t.parent_node * kCatSetWords + (b >> 6)] >> (b & 63)) & 1ull;
  return (b == t.missing_bin) ? (t.default_left != 0) : (b <= t.threshold);
}

// Sides of the rows of the multi-unit leaves: go-left mask words and left counts per block of kPartChunk rows for
// partition_scatter_kernel.  The grid has one workgroup per block, but only every kFlagBlocks-th one works: it takes
// kFlagBlocks consecutive blocks of its leaf, and fetches the row ids of the next block while the split feature's
// bytes of the current one (feature-major matrix, one dependent load per row) are on their way - two memory
// round trips per workgroup would otherwise be all a workgroup ever does.  Next to no LDS: a CU holds eight.
__global__ __launch_bounds__(kPartThreads) void grow_flags_kernel(InlineTasks inl, GrowArgs a) {
  constexpr int kWords = kPartChunk / kWave;
  constexpr int U = kPartChunk / kPartThreads;
  __shared__ int wave_left[kFlagBlocks][kPartThreads / kWave];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x;
  const int ti = a.tasks ? find_task_staged<true>(a.tasks, a.n_staged, wg) : find_task<true>(inl, wg);
  const SplitTask t = load_task(a.tasks, inl, ti);
  const int c0 = wg - t.first_pchunk;
  if (c0 >= t.num_pchunk || (c0 % kFlagBlocks) != 0) return;
  const bool timed = a.phase_ticks && (wg & 255) == 0;   // LGA_GROW_TIMING: a sample of the workgroups, see grow_kernel
  unsigned long long t_prev = timed ? wall_clock64() : 0ull;
  auto stamp = [&](int phase) {
    if (timed) {
      __builtin_amdgcn_s_waitcnt(0);
      if (tid == 0) {
        const unsigned long long now = wall_clock64();
        atomicAdd(&a.phase_ticks[phase], now - t_prev);
        t_prev = now;
      }
    }
  };
  const NodeRange p = a.nodes[t.parent_node];
  const int32_t* src = (p.buf ? a.a1 : a.a0) + p.begin;
  const uint8_t* fcol = a.bins_fm + (int64_t)t.feature * a.N;
  const int n_blocks = min(kFlagBlocks, t.num_pchunk - c0);
  int next[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const int pos = c0 * kPartChunk + k * kPartThreads + tid;
    next[k] = pos < p.count ? src[pos] : -1;
  }
  stamp(8);
  for (int j = 0; j < n_blocks; ++j) {
    int row[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      row[k] = next[k];
      if (row[k] >= 0 && !LGA_DEV_CHECK((unsigned)row[k] < (unsigned long long)a.N, kCheckPartitionRow, row[k], a.N)) row[k] = -1;
    }
    if (j + 1 < n_blocks) {
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int pos = (c0 + j + 1) * kPartChunk + k * kPartThreads + tid;
        next[k] = pos < p.count ? src[pos] : -1;
      }
    }
    int bin[U];
#pragma unroll
    for (int k = 0; k < U; ++k) bin[k] = row[k] >= 0 ? (int)fcol[row[k]] : 0;
    int lefts = 0;
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const bool left = row[k] >= 0 && bin_goes_left(t, bin[k], a.cat_sets);
      const unsigned long long m = __ballot(left);
      lefts += __popcll(m);
      if (lane == 0) a.mask[(int64_t)(wg + j) * kWords + k * (kPartThreads / kWave) + wave] = m;
    }
    if (lane == 0) wave_left[j][wave] = lefts;
  }
  stamp(9);
  __syncthreads();
  if (tid < n_blocks) {
    int all = 0;
#pragma unroll
    for (int w = 0; w < kPartThreads / kWave; ++w) all += wave_left[tid][w];
    a.chunk_left[wg + tid] = all;
  }
  stamp(10);
  if (timed && tid == 0) atomicAdd(&a.phase_ticks[12], 1ull);
}

// Single-pass partition of the multi-unit leaves (kernels.hpp: grow_partition).  A block of kPartChunk rows keeps its
// rows in registers, counts its lefts and rights and reserves room for both with ONE 64-bit atomic add on its leaf's
// cursor (lefts in the low word, rights in the high one): lefts fill the leaf's range from the front, rights from
// the END backwards, so no block ever waits for another.  The block whose reservation completes the leaf's row count
// knows the total of the lefts, writes the children's ranges and puts the cursor back to zero for the next round.
template <int U, int T>   // rows per thread, threads
__global__ __launch_bounds__(T) void grow_partition_kernel(InlineTasks inl, GrowArgs a) {
  constexpr int kPartThreads = T, kPartChunk = U * T;   // shadow the two-launch partition's shape
  constexpr int kWords = kPartChunk / kWave;
  constexpr int kWaves = kPartThreads / kWave;
  constexpr int kPerLane = (kWords + kWave - 1) / kWave;   // mask words a lane of the scanning wave sums
  static_assert(kWords % kPerLane == 0 && kWords / kPerLane <= kWave, "one wave scans the block's mask words");
  __shared__ unsigned long long smask[kWords];
  __shared__ int word_prefix[kWords + 1];
  __shared__ int s_before[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x;
  const int ti = a.tasks ? find_task_staged<true>(a.tasks, a.n_staged, wg) : find_task<true>(inl, wg);
  const SplitTask t = load_task(a.tasks, inl, ti);
  const int c = wg - t.first_pchunk;
  if (c >= t.num_pchunk) return;
  const NodeRange p = a.nodes[t.parent_node];
  const int32_t* src = (p.buf ? a.a1 : a.a0) + p.begin;
  int32_t* dst = (p.buf ? a.a0_dst : a.a1_dst) + p.begin;
  const uint8_t* fcol = a.bins_fm + (int64_t)t.feature * a.N;
  const int base = c * kPartChunk;
  const int in_block = min(kPartChunk, p.count - base);
  int row[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const int pos = base + k * kPartThreads + tid;
    row[k] = pos < p.count ? src[pos] : -1;
    // a row id outside the table still has to be PLACED (the counts must add up to the leaf's): it goes right
    if (row[k] >= 0) LGA_DEV_CHECK((unsigned)row[k] < (unsigned long long)a.N, kCheckPartitionRow, row[k], a.N);
  }
  int bin[U];
#pragma unroll
  for (int k = 0; k < U; ++k) bin[k] = (unsigned)row[k] < (unsigned)a.N ? (int)fcol[row[k]] : -1;
  unsigned long long mine[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const bool left = bin[k] >= 0 && bin_goes_left(t, bin[k], a.cat_sets);
    mine[k] = __ballot(left);
    if (lane == 0) smask[k * kWaves + wave] = mine[k];
  }
  __syncthreads();
  if (tid < kWave) {
    constexpr int kLanes = kWords / kPerLane;
    int pcs[kPerLane], pc = 0;
#pragma unroll
    for (int q = 0; q < kPerLane; ++q) { pcs[q] = tid < kLanes ? __popcll(smask[tid * kPerLane + q]) : 0; pc += pcs[q]; }
    int incl = pc;
#pragma unroll
    for (int d = 1; d < kLanes; d <<= 1) {
      const int o = __shfl_up(incl, d, kWave);
      if (lane >= d) incl += o;
    }
    if (tid < kLanes) {
      int run = incl - pc;
#pragma unroll
      for (int q = 0; q < kPerLane; ++q) { word_prefix[tid * kPerLane + q] = run; run += pcs[q]; }
    }
    if (tid == kLanes - 1) {
      const int own = incl, own_right = in_block - own;
      const unsigned long long before = atomicAdd(&a.part_cursor[ti * kPartCursorStride], (unsigned long long)own | ((unsigned long long)own_right << 32));
      const int lb = (int)(before & 0xffffffffull), rb = (int)(before >> 32);
      s_before[0] = lb;
      s_before[1] = rb;
      if (lb + rb + in_block == p.count) {   // the leaf's last reservation: this block knows the children's sizes
        const int all = lb + own;
        a.nodes[t.left_node] = NodeRange{p.begin, all, 1 - p.buf, 0};
        a.nodes[t.right_node] = NodeRange{p.begin + all, p.count - all, 1 - p.buf, 0};
        a.task_left[ti] = all;
        a.part_cursor[ti * kPartCursorStride] = 0ull;
      }
    }
  }
  __syncthreads();
  const int lefts_before = s_before[0], rights_before = s_before[1];
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    if (row[k] >= 0) {
      const int w = k * kWaves + wave;
      const int in_chunk = w * kWave + lane;
      const bool left = (mine[k] >> lane) & 1ull;
      const int lrank = word_prefix[w] + __popcll(mine[k] & below);
      const int d = left ? lefts_before + lrank : p.count - 1 - (rights_before + (in_chunk - lrank));
      if (LGA_DEV_CHECK((unsigned)d < (unsigned)p.count, kCheckPartitionTarget, d, p.count)) dst[d] = row[k];
    }
  }
}

// WIDE (full-precision histograms): the rows' gradient words are float bit patterns, converted to 36-bit fixed point in
// every pass and added as TWO 64-bit LDS atomics per (row, feature) into separate gradient and hessian planes - 128 KiB
// of LDS, so a CU holds one workgroup, which therefore has 16 waves instead of 8.  No cell can overflow.
// ALLF (a nearly full tile, F > 24): no per-step "does this lane's feature exist" test - the unused features' bin bytes
// are zero in the lines, their adds land in cells (feature >= F, bin 0) that nothing ever reads, and 32 predicated
// branches per row cost more than a few spare atomics (as a run-time flag the test stays in the loop: measured).
template <bool ROOT, bool QUANT, bool WIDE, bool ALLF>
__global__ __launch_bounds__(WIDE ? kGrowThreadsWide : kGrowThreads, 4) void grow_kernel(InlineTasks inl, GrowArgs a) {
  constexpr int kThreads = WIDE ? kGrowThreadsWide : kGrowThreads;
  constexpr int kWaves = kThreads / kWave;
  constexpr int kMaskWords = kSoloMaxRows / kWave;
  constexpr int U1 = kSoloMaxRows / kThreads;   // rows per lane of a (non-root) unit: all of them in one trip
  __shared__ unsigned long long lds[(WIDE ? 2 : 1) * kTileFeatures * kMaxBins];   // 64 KiB (WIDE: 128 KiB)
  __shared__ unsigned long long smask[kMaskWords];
  __shared__ int word_prefix[kMaskWords + 1];
  __shared__ int wave_tot[kWaves];
  __shared__ long long red[2][kWaves];
  __shared__ long long tot[3];
  __shared__ int32_t hist_rows[kGrowListRows];   // the histogram child's rows of this unit, compacted
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ti = a.tasks ? find_task_staged<false>(a.tasks, a.n_staged, blockIdx.x) : find_task<false>(inl, blockIdx.x);
  const SplitTask t = load_task(a.tasks, inl, ti);
  const int ridx = ti;                           // index among the round's tasks
  const int u = blockIdx.x - t.first_hchunk;
  const bool solo = !ROOT && t.num_pchunk == 0;  // the whole leaf is this workgroup's: sides, row movement, histogram
  // otherwise the workgroup histograms one unit of consecutive rows: of the root, or of the histogram child of a
  // multi-unit leaf, whose rows grow_flags_kernel + partition_scatter_kernel have moved already
  const bool dense = ROOT || !solo;
  const NodeRange p = a.nodes[ROOT ? 0 : (solo ? t.parent_node : (t.hist_child ? t.right_node : t.left_node))];
  const int32_t* src = ((ROOT || solo) ? (p.buf ? a.a1 : a.a0) : (p.buf ? a.a1_dst : a.a0_dst)) + p.begin;
  const int lo = solo ? 0 : u * a.unit_rows;
  const int hi = solo ? p.count : min(lo + a.unit_rows, p.count);
  if (dense && u > 0 && lo >= p.count) return;   // the child has fewer rows on this rank than the grid was sized for
  const int nf = a.F;
  const PoolLayout& L = a.pool;
  const SlotRef fresh = new_slot_ref(L, a.round, t.new_slot);
  if (u == 0 && tid == 0) publish_slot(L, t.new_slot, fresh);
  // phase profile (LGA_GROW_TIMING): thread 0 adds the time since its previous stamp to phase_ticks[phase]
  unsigned long long t_prev = a.phase_ticks ? wall_clock64() : 0ull;
  auto stamp = [&](int phase) {
    if (a.phase_ticks && tid == 0) {
      const unsigned long long now = wall_clock64();
      atomicAdd(&a.phase_ticks[phase], now - t_prev);
      t_prev = now;
    }
  };
  auto clear_lds = [&]() {
    for (int i = tid; i < kTileFeatures * kMaxBins; i += kThreads) {
      lds[i] = 0ull;
      if (WIDE) lds[kWidePlane + i] = 0ull;
    }
  };
  clear_lds();
  int kg = 0, kh = 0;   // WIDE: the exponents of the 36-bit form
  if (QUANT) {
    kg = quant_exponent(__uint_as_float(a.quant.maxbits[0])) + (WIDE ? kWideExtraBits : 0);
    kh = quant_exponent(__uint_as_float(a.quant.maxbits[1])) + (WIDE ? kWideExtraBits : 0);
    if (blockIdx.x == 0 && tid == 0) { a.quant.qs_out->kg = kg; a.quant.qs_out->kh = kh; }
  } else if (WIDE) {
    kg = a.qs->kg;
    kh = a.qs->kh;
  }
  const bool hist_left = t.hist_child == 0;
  const uint4* lines = reinterpret_cast<const uint4*>(a.lines);
  long long sg = 0, sh = 0;
  // One trip of histogram work: up to four rows per lane fetch their lines (bins + packed gradients), every load
  // issued before the LDS atomics.  row < 0: nothing to do for that slot.
  auto add_rows = [&](int (&row)[4]) {
    unsigned long long pk[4];
    uint4 ra[4], rb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (row[k] >= 0 && !LGA_DEV_CHECK((unsigned)row[k] < (unsigned long long)a.N, kCheckHistRow, row[k], a.N)) row[k] = -1;
      if (row[k] >= 0) {
        const unsigned line = div3((unsigned)row[k]), slot = (unsigned)row[k] - line * 3u;
        const uint4* Lp = lines + (int64_t)line * 8;
        ra[k] = Lp[slot * 2];
        rb[k] = Lp[slot * 2 + 1];
        if (QUANT) {
          pk[k] = WIDE ? float_pair_bits(a.quant.g[row[k]], a.quant.h[row[k]])
                       : (unsigned long long)pack_gh(quantize_with(a.quant.g[row[k]], kg, a.quant.rounding, 0, (uint64_t)row[k]),
                                                     quantize_with(a.quant.h[row[k]], kh, a.quant.rounding, 1, (uint64_t)row[k]));
          reinterpret_cast<unsigned long long*>(const_cast<uint4*>(Lp))[12 + slot] = pk[k];
        } else {
          pk[k] = reinterpret_cast<const unsigned long long*>(Lp)[12 + slot];
        }
      }
    }
    if (a.phase_ticks) { __builtin_amdgcn_s_waitcnt(0); stamp(1); }   // profile only: loads done
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (row[k] >= 0) {
        unsigned w[8] = {ra[k].x, ra[k].y, ra[k].z, ra[k].w, rb[k].x, rb[k].y, rb[k].z, rb[k].w};
        long long qg = 0, qh = 0;
        if (WIDE) {
          wide_of(pk[k], kg, kh, qg, qh);
          sg += qg;
          sh += qh;
        } else {
          sg += (long long)(int32_t)(pk[k] >> 32);
          sh += (long long)(pk[k] & 0xffffffffull);
        }
        // byte j of the stored record holds feature (j + row) mod 32; rotated by (lane - row) it holds feature
        // (j + lane) mod 32: the lanes of a wave then work on 32 different features at any moment, whatever rows
        // they hold, and the feature is the bank (hist_cell): conflict-free LDS atomics
        rotate_record(w, ((unsigned)lane - (unsigned)row[k]) & 31u);
#pragma unroll
        for (int j = 0; j < kTileFeatures; ++j) {
          const unsigned f = (j + (unsigned)lane) & 31u;
          if (ALLF || f < (unsigned)nf) {
            const unsigned bn = (w[j >> 2] >> ((j & 3) * 8)) & 255u;
            const unsigned cell = (bn << 5) + f;
            if (WIDE) {
              atomicAdd(&lds[cell], (unsigned long long)qg);
              atomicAdd(&lds[kWidePlane + cell], (unsigned long long)qh);
            } else {
              atomicAdd(&lds[cell], pk[k]);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // one row at a time: interleaving the rows' rotations costs registers (spills)
    }
  };
  // rows, sum g, sum h of what the workgroup put into its histogram -> tot[]
  auto reduce_sums = [&](int rows) {
    sg = wave_sum_i64(sg);
    sh = wave_sum_i64(sh);
    if (lane == 0) { red[0][wave] = sg; red[1][wave] = sh; }
    __syncthreads();
    if (tid == 0) {
      long long tg = 0, th = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) { tg += red[0][w]; th += red[1][w]; }
      tot[0] = rows;
      tot[1] = tg;
      tot[2] = th;
    }
    __syncthreads();
  };
  int lefts = 0;
  if (dense) {
    // The grid holds the units the child was EXPECTED to have (its row count is only estimated by the search and only
    // known on the device): workgroup u takes units u, u + num_hchunk, ... until the child's rows are used up, and
    // adds each unit's histogram to the (zeroed) slot before the 32-bit halves of its LDS cells could overflow.
    const int stride = ROOT ? (int)gridDim.x : t.num_hchunk;
    for (int unit = u; unit == u || unit * a.unit_rows < p.count; unit += stride) {
      const int ulo = unit * a.unit_rows, uhi = min(ulo + a.unit_rows, p.count);
      if (unit != u) {
        clear_lds();
        sg = 0; sh = 0;
      }
      __syncthreads();
      stamp(0);
      // (fetching trip i + 1 while trip i goes into the histogram was measured slower: 128 VGPRs with spills,
      // root pass 244 -> 260 us - the other waves of the CU already fill the gaps)
      // the row ids of trip i + 1 are requested before trip i's lines (four registers): a trip then waits for ONE
      // memory round trip, its lines, instead of two dependent ones
      int nxt[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int pos = ulo + k * kThreads + tid;
        nxt[k] = pos < uhi ? src[pos] : -1;
      }
      for (int sub = ulo; sub < uhi; sub += 4 * kThreads) {
        int row[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          row[k] = nxt[k];
          const int pos = sub + 4 * kThreads + k * kThreads + tid;
          nxt[k] = pos < uhi ? src[pos] : -1;
        }
        add_rows(row);
        if (a.phase_ticks) { __builtin_amdgcn_s_waitcnt(0); stamp(2); }
      }
      __syncthreads();
      stamp(3);
      reduce_sums(max(uhi - ulo, 0));
      stamp(4);
      flush_lds_hist<WIDE>(lds, nf, 0, L, fresh, kThreads, (unsigned)unit);
      if (tid == 0) add_slot_header(L, fresh, tot[0], tot[1], tot[2]);
      __syncthreads();
      stamp(5);
      if (a.phase_ticks && tid == 0) atomicAdd(&a.phase_ticks[7], 1ull);
    }
    return;
  } else {
    // Phase 1 - which side does every row take?  The split feature's byte comes from the feature-major matrix
    // (coalesced while the leaf is dense); all rows of the unit (up to sixteen per lane) are in flight at once,
    // because this phase is nothing but two dependent loads per row.  The go-left bits land in the LDS mask
    // words (word = position / 64).
    const uint8_t* fcol = a.bins_fm + (int64_t)t.feature * a.N;
    int row[U1];
#pragma unroll
    for (int k = 0; k < U1; ++k) {
      const int pos = lo + k * kThreads + tid;
      row[k] = pos < hi ? src[pos] : -1;
      if (pos < hi && !LGA_DEV_CHECK((unsigned)row[k] < (unsigned long long)a.N, kCheckPartitionRow, row[k], a.N)) row[k] = -1;
    }
    int bin[U1];
#pragma unroll
    for (int k = 0; k < U1; ++k) bin[k] = row[k] >= 0 ? (int)fcol[row[k]] : 0;
    unsigned long long mine[U1];   // the mask word each of this lane's rows sits in
#pragma unroll
    for (int k = 0; k < U1; ++k) {
      mine[k] = 0ull;
      if (lo + k * kThreads < hi) {   // uniform: the whole stripe of kThreads positions is outside otherwise
        const bool left = row[k] >= 0 && bin_goes_left(t, bin[k], a.cat_sets);
        mine[k] = __ballot(left);
        if (lane == 0) smask[k * kWaves + wave] = mine[k];
      }
    }
    __syncthreads();
    stamp(0);
    // exclusive prefix of the mask words' popcounts
    {
      const int n_words = (hi - lo + kWave - 1) / kWave;
      int pc = 0, incl = 0;
      if (tid < kMaskWords) {
        pc = tid < n_words ? __popcll(smask[tid]) : 0;
        incl = pc;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
          const int o = __shfl_up(incl, d, kWave);
          if (lane >= d) incl += o;
        }
        if (lane == kWave - 1) wave_tot[wave] = incl;
      }
      __syncthreads();
      if (tid < kMaskWords) {
        int off = 0;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        word_prefix[tid] = off + incl - pc;
        if (tid == kMaskWords - 1) word_prefix[kMaskWords] = off + incl;
      }
      __syncthreads();
      lefts = word_prefix[kMaskWords];
    }
    stamp(1);
    // Phase 2 - the histogram child's rows, compacted: every lane knows the rank of its rows among the child's rows
    // of this unit and drops them into the LDS list, a window of kGrowListRows ranks at a time; the list is then
    // worked off by full waves, whatever share of the unit the child takes.
    const int n_hist = hist_left ? lefts : (hi - lo) - lefts;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int w0 = 0; w0 < n_hist; w0 += kGrowListRows) {
#pragma unroll
      for (int k = 0; k < U1; ++k) {
        const int q = k * kThreads + tid;   // position inside the unit
        if (lo + q < hi) {
          const bool left = (mine[k] >> lane) & 1ull;
          if (left == hist_left) {
            const int lrank = word_prefix[k * kWaves + wave] + __popcll(mine[k] & below);
            const int rank = (hist_left ? lrank : q - lrank) - w0;
            if ((unsigned)rank < (unsigned)kGrowListRows) hist_rows[rank] = row[k];
          }
        }
      }
      __syncthreads();
      const int n_list = min(kGrowListRows, n_hist - w0);
      for (int i0 = 0; i0 < n_list; i0 += 4 * kThreads) {
        int r4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * kThreads + tid;
          r4[k] = i < n_list ? hist_rows[i] : -1;
        }
        add_rows(r4);
      }
      __syncthreads();
    }
    stamp(2);
  }
  if (!dense) {
    {
      // the leaf fits this workgroup: move its rows right away - lefts to the front, rights behind, in order
      int32_t* dst = (p.buf ? a.a0_dst : a.a1_dst) + p.begin;
      for (int pos = tid; pos < p.count; pos += kThreads) {
        const int w = pos >> 6, l = pos & 63;
        const unsigned long long m = smask[w];
        const bool left = (m >> l) & 1ull;
        const int lrank = word_prefix[w] + __popcll(m & ((1ull << l) - 1ull));
        const int d = left ? lrank : lefts + (pos - lrank);
        if (LGA_DEV_CHECK((unsigned)d < (unsigned)p.count, kCheckPartitionTarget, d, p.count)) dst[d] = src[pos];
      }
      if (tid == 0) {
        a.nodes[t.left_node] = NodeRange{p.begin, lefts, 1 - p.buf, 0};
        a.nodes[t.right_node] = NodeRange{p.begin + lefts, p.count - lefts, 1 - p.buf, 0};
        a.task_left[ridx] = lefts;
      }
    }
  }
  stamp(3);
  reduce_sums(hist_left ? lefts : (hi - lo) - lefts);
  // the whole leaf was this workgroup's: the histogram goes to the slot as it is (no atomics, no zeroing)
  for (int e = tid; e < kTileFeatures * kMaxBins; e += kThreads) {
    int fe, b;
    flush_item(e, fe, b);
    if (fe >= nf) continue;
    const int cell = hist_cell(fe, b);
    int64_t* blk = feat_block(L, fresh, fe);
    if (WIDE) {
      blk[b] = (int64_t)lds[cell];
      blk[kMaxBins + b] = (int64_t)lds[kWidePlane + cell];
    } else {
      const unsigned long long v = lds[cell];
      blk[b] = (int64_t)(int32_t)(v >> 32);
      blk[kMaxBins + b] = (int64_t)(v & 0xffffffffull);
    }
  }
  if (tid == 0)
    for (int r = 0; r < L.ranks; ++r) {
      int64_t* hd = slot_header(L, fresh, r);
      hd[0] = tot[0]; hd[1] = tot[1]; hd[2] = tot[2];
    }
  __syncthreads();
  stamp(6);
}

// ---------------------------------------------------------------------------
// score updates
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void leaf_add_kernel(const LeafSpan* __restrict__ leaves, int n_leaves,
                                                       const int32_t* __restrict__ a0, const int32_t* __restrict__ a1,
                                                       int64_t n_rows, double* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* begins = reinterpret_cast<int32_t*>(smem);
  for (int i = threadIdx.x; i < n_leaves; i += blockDim.x) begins[i] = leaves[i].begin;
  __syncthreads();
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_rows; p += (int64_t)gridDim.x * blockDim.x) {
    // last leaf with begin <= p (leaves are sorted by begin and tile [0, n_rows))
    int lo = 0, hi = n_leaves - 1;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if (begins[mid] <= (int32_t)p) lo = mid; else hi = mid - 1;
    }
    const LeafSpan l = leaves[lo];
    const int32_t row = (l.buf ? a1 : a0)[p];
    scores[row] += l.value;
  }
}

// ---------------------------------------------------------------------------
// exact per-leaf gradient sums (leaf-value renewal)
// ---------------------------------------------------------------------------
// Workgroup w owns positions [w * 4096, (w + 1) * 4096) of the bag's row arena.  The leaves' row ranges are
// contiguous there, so a workgroup meets a handful of them: for each it reduces the 36-bit fixed-point gradients
// of its share of the range and adds the two sums to the leaf's accumulators (integers: order does not matter).
__device__ __forceinline__ int span_of(const int32_t* begins, int n_spans, int32_t p);
constexpr int kLeafSumChunk = 4096;
__global__ __launch_bounds__(256) void leaf_sums_kernel(const LeafSpan* __restrict__ spans, int n_spans,
                                                        const int32_t* __restrict__ a0, const int32_t* __restrict__ a1,
                                                        int64_t n_rows, const float* __restrict__ g,
                                                        const float* __restrict__ h, const QuantScale* __restrict__ qs, int extra_bits,
                                                        unsigned long long* __restrict__ acc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* begins = reinterpret_cast<int32_t*>(smem);
  __shared__ long long red[2][4];
  for (int i = threadIdx.x; i < n_spans; i += blockDim.x) begins[i] = spans[i].begin;
  __syncthreads();
  const int kg = qs->kg + extra_bits, kh = qs->kh + extra_bits;
  const int64_t p0 = (int64_t)blockIdx.x * kLeafSumChunk;
  const int64_t p1 = p0 + kLeafSumChunk < n_rows ? p0 + kLeafSumChunk : n_rows;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = span_of(begins, n_spans, (int32_t)p0); i < n_spans; ++i) {
    const LeafSpan sp = spans[i];
    if ((int64_t)sp.begin >= p1) break;
    const int64_t lo = sp.begin > p0 ? sp.begin : p0, hi = sp.end < p1 ? sp.end : p1;
    if (lo >= hi) continue;   // empty ranges sort in front of the range that owns their position
    const int32_t* rows = sp.buf ? a1 : a0;
    long long sg = 0, sh = 0;
    for (int64_t p = lo + threadIdx.x; p < hi; p += blockDim.x) {
      const int32_t r = rows[p];
      sg += quantize_wide(g[r], kg);
      sh += quantize_wide(h[r], kh);
    }
    sg = wave_sum_i64(sg);
    sh = wave_sum_i64(sh);
    __syncthreads();
    if (lane == 0) { red[0][wave] = sg; red[1][wave] = sh; }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(&acc[2 * sp.leaf], (unsigned long long)(red[0][0] + red[0][1] + red[0][2] + red[0][3]));
      atomicAdd(&acc[2 * sp.leaf + 1], (unsigned long long)(red[1][0] + red[1][1] + red[1][2] + red[1][3]));
    }
  }
}

// ---------------------------------------------------------------------------
// percentile leaf outputs
// ---------------------------------------------------------------------------
__device__ __forceinline__ int span_of(const int32_t* begins, int n_spans, int32_t p) {
  int lo = 0, hi = n_spans - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (begins[mid] <= p) lo = mid; else hi = mid - 1;
  }
  return lo;
}
__device__ __forceinline__ float residual_of(const double* scores, const float* label, int32_t row) {
  float r = (float)((double)label[row] - scores[row]);
  return r == 0.0f ? 0.0f : r;   // one zero
}
__device__ __forceinline__ uint32_t float_order_bits(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_from_order_bits(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void residual_keys_kernel(const LeafSpan* __restrict__ spans, int n_spans,
                                                            const int32_t* __restrict__ a0, const int32_t* __restrict__ a1,
                                                            int64_t n_rows, const double* __restrict__ scores,
                                                            const float* __restrict__ label, uint64_t* __restrict__ keys,
                                                            int64_t npad) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* begins = reinterpret_cast<int32_t*>(smem);
  for (int i = threadIdx.x; i < n_spans; i += blockDim.x) begins[i] = spans[i].begin;
  __syncthreads();
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npad; p += (int64_t)gridDim.x * blockDim.x) {
    uint64_t key = ~0ull;
    if (p < n_rows) {
      const LeafSpan sp = spans[span_of(begins, n_spans, (int32_t)p)];
      const int32_t row = (sp.buf ? a1 : a0)[p];
      key = ((uint64_t)(uint32_t)sp.leaf << 32) | float_order_bits(residual_of(scores, label, row));
    }
    keys[p] = key;
  }
}

__global__ __launch_bounds__(256) void residual_items_kernel(const LeafSpan* __restrict__ spans, int n_spans,
                                                             const int32_t* __restrict__ a0, const int32_t* __restrict__ a1,
                                                             int64_t n_rows, const double* __restrict__ scores,
                                                             const float* __restrict__ label, const float* __restrict__ weight,
                                                             int mape_scale, int32_t* __restrict__ leaf_out,
                                                             float* __restrict__ value_out, float* __restrict__ weight_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int32_t* begins = reinterpret_cast<int32_t*>(smem);
  for (int i = threadIdx.x; i < n_spans; i += blockDim.x) begins[i] = spans[i].begin;
  __syncthreads();
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_rows; p += (int64_t)gridDim.x * blockDim.x) {
    const LeafSpan sp = spans[span_of(begins, n_spans, (int32_t)p)];
    const int32_t row = (sp.buf ? a1 : a0)[p];
    float wt = weight ? weight[row] : 1.0f;
    if (mape_scale) wt = (float)((double)wt / fmax(1.0, fabs((double)label[row])));
    leaf_out[p] = sp.leaf;
    value_out[p] = residual_of(scores, label, row);
    weight_out[p] = wt;
  }
}

struct SortedKeyValues {
  const uint64_t* keys;
  __device__ double operator()(int64_t i) const { return (double)float_from_order_bits((uint32_t)keys[i]); }
};

__global__ __launch_bounds__(256) void leaf_percentiles_kernel(const uint64_t* __restrict__ keys, int64_t n_rows,
                                                               int n_leaves, double alpha, double* __restrict__ values,
                                                               int32_t* __restrict__ counts) {
  for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < n_leaves; l += gridDim.x * blockDim.x) {
    // [first key of leaf l, first key of leaf l + 1)
    int64_t bound[2];
    for (int e = 0; e < 2; ++e) {
      const uint64_t want = (uint64_t)(uint32_t)(l + e) << 32;
      int64_t lo = 0, hi = n_rows;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < want) lo = mid + 1; else hi = mid;
      }
      bound[e] = lo;
    }
    const int64_t n = bound[1] - bound[0];
    counts[l] = (int32_t)n;
    values[l] = n > 0 ? percentile_unweighted(SortedKeyValues{keys + bound[0]}, n, alpha) : 0.0;
  }
}

// Adds a tree to the scores of `rows` by walking it over the binned data.  Nodes are 16-byte records
// staged in LDS (one ds_read_b128 per step); rows are visited in storage order, so the one-byte bin
// reads of neighbouring lanes share cache lines and the score update is a coalesced read-modify-write.
template <bool IN_LDS>
__global__ __launch_bounds__(256) void traverse_add_kernel(const TreeNodeDev* __restrict__ nodes, int n_internal,
                                                           const double* __restrict__ leaf_value, int num_leaves,
                                                           const uint8_t* __restrict__ bins_fm, int64_t N,
                                                           const int32_t* __restrict__ rows, int64_t n, double sign,
                                                           double* __restrict__ scores,
                                                           const uint64_t* __restrict__ node_sets) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TreeNodeDev* tn = nodes;
  const double* lv = leaf_value;
  if (IN_LDS) {
    int4* sn = reinterpret_cast<int4*>(smem);
    double* sv = reinterpret_cast<double*>(smem + (size_t)n_internal * sizeof(TreeNodeDev));
    const int4* gn = reinterpret_cast<const int4*>(nodes);
    for (int i = threadIdx.x; i < n_internal; i += blockDim.x) sn[i] = gn[i];
    for (int i = threadIdx.x; i < num_leaves; i += blockDim.x) sv[i] = leaf_value[i];
    __syncthreads();
    tn = reinterpret_cast<const TreeNodeDev*>(smem);
    lv = sv;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows ? rows[i] : i;
    if (!LGA_DEV_CHECK((unsigned long long)r < (unsigned long long)N, kCheckWalkRow, r, N)) continue;
    int node = num_leaves <= 1 ? ~0 : 0;
    while (node >= 0) {
      const TreeNodeDev t = tn[node];
      const int b = bins_fm[(int64_t)t.feature * N + r];
      const int thr = t.meta & 255, dl = (t.meta >> 8) & 1, mb = (t.meta >> 16) - 1;
      const bool left = (t.meta & 1024) ? ((node_sets[(int64_t)node * kCatSetWords + (b >> 6)] >> (b & 63)) & 1ull) != 0
                        : (t.meta & 512) ? (b == thr) : ((b == mb) ? (dl != 0) : (b <= thr));
      node = left ? t.left : t.right;
      if (!LGA_DEV_CHECK(node < n_internal && ~node < num_leaves, kCheckWalkNode, node, n_internal)) break;
    }
    if (node < 0) scores[r] += sign * lv[~node];
  }
}

// Leaf of every row + exact leaf sums in ONE walk over the binned rows in storage order (large data without
// out-of-bag rows): leaf_of_row[r] = the leaf row r ends in, acc[2 * leaf] / acc[2 * leaf + 1] += the 36-bit
// fixed-point gradient / hessian of the row (quantize_wide, exponents qs + kWideExtraBits).  A workgroup sums in LDS
// and adds its non-zero cells to global memory once.  The score update then is a coalesced lookup
// (leaf_value_add_kernel) instead of a second walk, and the sums cost no scattered pass over the leaves' row lists.
__global__ __launch_bounds__(256) void walk_leaves_kernel(const TreeNodeDev* __restrict__ nodes, int n_internal, int num_leaves,
                                                          const uint8_t* __restrict__ bins_fm, int64_t N,
                                                          const int32_t* __restrict__ rows, int64_t n,
                                                          const float* __restrict__ g, const float* __restrict__ h,
                                                          const QuantScale* __restrict__ qs, int extra_bits, uint16_t* __restrict__ leaf_of_row,
                                                          unsigned long long* __restrict__ acc,
                                                          const uint64_t* __restrict__ node_sets) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int4* sn = reinterpret_cast<int4*>(smem);
  unsigned long long* sacc = reinterpret_cast<unsigned long long*>(smem + (size_t)n_internal * sizeof(TreeNodeDev));
  const int4* gn = reinterpret_cast<const int4*>(nodes);
  for (int i = threadIdx.x; i < n_internal; i += blockDim.x) sn[i] = gn[i];
  for (int i = threadIdx.x; i < 2 * num_leaves; i += blockDim.x) sacc[i] = 0ull;
  __syncthreads();
  const TreeNodeDev* tn = reinterpret_cast<const TreeNodeDev*>(smem);
  const int kg = qs->kg + extra_bits, kh = qs->kh + extra_bits;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows ? rows[i] : i;
    if (!LGA_DEV_CHECK((unsigned long long)r < (unsigned long long)N, kCheckWalkRow, r, N)) continue;
    const float gv = g[r], hv = h[r];
    int node = num_leaves <= 1 ? ~0 : 0;
    while (node >= 0) {
      const TreeNodeDev t = tn[node];
      const int b = bins_fm[(int64_t)t.feature * N + r];
      const int thr = t.meta & 255, dl = (t.meta >> 8) & 1, mb = (t.meta >> 16) - 1;
      const bool left = (t.meta & 1024) ? ((node_sets[(int64_t)node * kCatSetWords + (b >> 6)] >> (b & 63)) & 1ull) != 0
                        : (t.meta & 512) ? (b == thr) : ((b == mb) ? (dl != 0) : (b <= thr));
      node = left ? t.left : t.right;
      if (!LGA_DEV_CHECK(node < n_internal && ~node < num_leaves, kCheckWalkNode, node, n_internal)) break;
    }
    if (node < 0) {
      const int leaf = ~node;
      leaf_of_row[r] = (uint16_t)leaf;
      atomicAdd(&sacc[2 * leaf], (unsigned long long)quantize_wide(gv, kg));
      atomicAdd(&sacc[2 * leaf + 1], (unsigned long long)quantize_wide(hv, kh));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * num_leaves; i += blockDim.x)
    if (sacc[i] != 0ull) atomicAdd(&acc[i], sacc[i]);
}

// The same walk for data of a single feature tile: a row's 32 bin bytes are fetched ONCE (two 16-byte loads from the
// row-major records, consecutive lanes = consecutive rows) and every step of the walk picks its byte from registers,
// so a step costs an LDS node read instead of a dependent global load.  Four rows per lane walk interleaved.
__device__ __forceinline__ int record_bin(const unsigned (&w)[8], int feature) {
  const unsigned q = (unsigned)feature >> 2;
  unsigned d = w[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) d = q == (unsigned)i ? w[i] : d;
  return (int)((d >> (((unsigned)feature & 3u) * 8)) & 255u);
}
__global__ __launch_bounds__(256) void walk_leaves_records_kernel(const TreeNodeDev* __restrict__ nodes, int n_internal, int num_leaves,
                                                                  const uint4* __restrict__ recs, int64_t N,
                                                                  const int32_t* __restrict__ rows, int64_t n,
                                                                  const float* __restrict__ g, const float* __restrict__ h,
                                                                  const QuantScale* __restrict__ qs, int extra_bits, uint16_t* __restrict__ leaf_of_row,
                                                                  unsigned long long* __restrict__ acc,
                                                                  const uint64_t* __restrict__ node_sets) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = 4;
  int4* sn = reinterpret_cast<int4*>(smem);
  unsigned long long* sacc = reinterpret_cast<unsigned long long*>(smem + (size_t)n_internal * sizeof(TreeNodeDev));
  const int4* gn = reinterpret_cast<const int4*>(nodes);
  for (int i = threadIdx.x; i < n_internal; i += blockDim.x) sn[i] = gn[i];
  for (int i = threadIdx.x; i < 2 * num_leaves; i += blockDim.x) sacc[i] = 0ull;
  __syncthreads();
  const TreeNodeDev* tn = reinterpret_cast<const TreeNodeDev*>(smem);
  const int kg = qs->kg + extra_bits, kh = qs->kh + extra_bits;
  for (int64_t base = (int64_t)blockIdx.x * (R * 256); base < n; base += (int64_t)gridDim.x * (R * 256)) {
    int64_t r[R];
    unsigned w[R][8];
    float gv[R], hv[R];
    int node[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int64_t i = base + k * 256 + threadIdx.x;
      r[k] = i < n ? (rows ? (int64_t)rows[i] : i) : -1;
      if (r[k] >= 0 && !LGA_DEV_CHECK((unsigned long long)r[k] < (unsigned long long)N, kCheckWalkRow, r[k], N)) r[k] = -1;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      node[k] = (r[k] >= 0 && num_leaves > 1) ? 0 : ~0;
      if (r[k] >= 0) {
        const uint4 a = recs[r[k] * 2], b = recs[r[k] * 2 + 1];
        w[k][0] = a.x; w[k][1] = a.y; w[k][2] = a.z; w[k][3] = a.w;
        w[k][4] = b.x; w[k][5] = b.y; w[k][6] = b.z; w[k][7] = b.w;
        gv[k] = g[r[k]];
        hv[k] = h[r[k]];
      }
    }
    bool more = true;
    while (more) {
      more = false;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (node[k] >= 0) {
          const TreeNodeDev t = tn[node[k]];
          const int b = record_bin(w[k], t.feature);
          const int thr = t.meta & 255, dl = (t.meta >> 8) & 1, mb = (t.meta >> 16) - 1;
          const bool left = (t.meta & 1024) ? ((node_sets[(int64_t)node[k] * kCatSetWords + (b >> 6)] >> (b & 63)) & 1ull) != 0
                            : (t.meta & 512) ? (b == thr) : ((b == mb) ? (dl != 0) : (b <= thr));
          node[k] = left ? t.left : t.right;
          if (!LGA_DEV_CHECK(node[k] < n_internal && ~node[k] < num_leaves, kCheckWalkNode, node[k], n_internal)) { node[k] = ~0; r[k] = -1; }
          more = more || node[k] >= 0;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (r[k] >= 0) {
        const int leaf = ~node[k];
        leaf_of_row[r[k]] = (uint16_t)leaf;
        atomicAdd(&sacc[2 * leaf], (unsigned long long)quantize_wide(gv[k], kg));
        atomicAdd(&sacc[2 * leaf + 1], (unsigned long long)quantize_wide(hv[k], kh));
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * num_leaves; i += blockDim.x)
    if (sacc[i] != 0ull) atomicAdd(&acc[i], sacc[i]);
}

// Variant with the rows' bin bytes in LDS (R rows per lane, 36-byte stride: nine dwords, coprime with the bank
// count, so the lanes of a wave read their bytes from different banks): a step of the walk is a node read and a
// one-byte read, both LDS, instead of a sixteen-instruction pick from registers.
template <int R>
__global__ __launch_bounds__(256) void walk_leaves_lds_kernel(const TreeNodeDev* __restrict__ nodes, int n_internal, int num_leaves,
                                                              const uint4* __restrict__ recs, int64_t N,
                                                              const int32_t* __restrict__ rows, int64_t n,
                                                              const float* __restrict__ g, const float* __restrict__ h,
                                                              const QuantScale* __restrict__ qs, int extra_bits, uint16_t* __restrict__ leaf_of_row,
                                                              unsigned long long* __restrict__ acc,
                                                              const uint64_t* __restrict__ node_sets) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kStride = 36;
  int4* sn = reinterpret_cast<int4*>(smem);
  unsigned long long* sacc = reinterpret_cast<unsigned long long*>(smem + (size_t)n_internal * sizeof(TreeNodeDev));
  unsigned char* sbins = reinterpret_cast<unsigned char*>(sacc + 2 * num_leaves);   // [R * 256 rows][36 bytes], 4-byte aligned
  const int4* gn = reinterpret_cast<const int4*>(nodes);
  for (int i = threadIdx.x; i < n_internal; i += blockDim.x) sn[i] = gn[i];
  for (int i = threadIdx.x; i < 2 * num_leaves; i += blockDim.x) sacc[i] = 0ull;
  __syncthreads();
  const TreeNodeDev* tn = reinterpret_cast<const TreeNodeDev*>(smem);
  const int kg = qs->kg + extra_bits, kh = qs->kh + extra_bits;
  for (int64_t base = (int64_t)blockIdx.x * (R * 256); base < n; base += (int64_t)gridDim.x * (R * 256)) {
    int64_t r[R];
    float gv[R], hv[R];
    int node[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int64_t i = base + k * 256 + threadIdx.x;
      r[k] = i < n ? (rows ? (int64_t)rows[i] : i) : -1;
      if (r[k] >= 0 && !LGA_DEV_CHECK((unsigned long long)r[k] < (unsigned long long)N, kCheckWalkRow, r[k], N)) r[k] = -1;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      node[k] = (r[k] >= 0 && num_leaves > 1) ? 0 : ~0;
      if (r[k] >= 0) {
        const uint4 a = recs[r[k] * 2], b = recs[r[k] * 2 + 1];
        unsigned* dst = reinterpret_cast<unsigned*>(sbins + (size_t)(k * 256 + (int)threadIdx.x) * kStride);   // only this lane reads it back
        dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
        dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
        gv[k] = g[r[k]];
        hv[k] = h[r[k]];
      }
    }
    bool more = true;
    while (more) {
      more = false;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (node[k] >= 0) {
          const TreeNodeDev t = tn[node[k]];
          const int b = sbins[(size_t)(k * 256 + (int)threadIdx.x) * kStride + ((unsigned)t.feature & 31u)];
          const int thr = t.meta & 255, dl = (t.meta >> 8) & 1, mb = (t.meta >> 16) - 1;
          const bool left = (t.meta & 1024) ? ((node_sets[(int64_t)node[k] * kCatSetWords + (b >> 6)] >> (b & 63)) & 1ull) != 0
                            : (t.meta & 512) ? (b == thr) : ((b == mb) ? (dl != 0) : (b <= thr));
          node[k] = left ? t.left : t.right;
          if (!LGA_DEV_CHECK(node[k] < n_internal && ~node[k] < num_leaves, kCheckWalkNode, node[k], n_internal)) { node[k] = ~0; r[k] = -1; }
          more = more || node[k] >= 0;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (r[k] >= 0) {
        const int leaf = ~node[k];
        leaf_of_row[r[k]] = (uint16_t)leaf;
        atomicAdd(&sacc[2 * leaf], (unsigned long long)quantize_wide(gv[k], kg));
        atomicAdd(&sacc[2 * leaf + 1], (unsigned long long)quantize_wide(hv[k], kh));
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * num_leaves; i += blockDim.x)
    if (sacc[i] != 0ull) atomicAdd(&acc[i], sacc[i]);
}

__global__ __launch_bounds__(256) void leaf_value_add_kernel(const uint16_t* __restrict__ leaf_of_row,
                                                             const double* __restrict__ leaf_value, int num_leaves,
                                                             const int32_t* __restrict__ rows, int64_t n,
                                                             double* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lv = reinterpret_cast<double*>(smem);
  for (int i = threadIdx.x; i < num_leaves; i += blockDim.x) lv[i] = leaf_value[i];
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows ? rows[i] : i;
    scores[r] += lv[leaf_of_row[r]];
  }
}

// The score update of a finished tree and the gradients of the NEXT iteration in one pass over the rows (large data,
// every row's leaf known from the walk): scores[r] += value of r's leaf, then gradients_kernel's work on the new
// score - one read of the scores instead of a write and a read, one launch instead of two.
__global__ __launch_bounds__(256) void leaf_value_add_gradients_kernel(ObjectiveSpec obj, const uint16_t* __restrict__ leaf_of_row,
                                                                       const double* __restrict__ leaf_value, int num_leaves,
                                                                       const int32_t* __restrict__ rows, int64_t n,
                                                                       double* __restrict__ scores, const float* __restrict__ label,
                                                                       const float* __restrict__ weight, float* __restrict__ g,
                                                                       float* __restrict__ h, uint32_t* maxbits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lv = reinterpret_cast<double*>(smem);
  for (int i = threadIdx.x; i < num_leaves; i += blockDim.x) lv[i] = leaf_value[i];
  __syncthreads();
  constexpr int U = 4;
  float mg = 0.f, mh = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += stride * U) {
    int64_t r[U];
    double sc[U];
    float lb[U], wt[U];
    int leaf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      r[u] = i < n ? (rows ? (int64_t)rows[i] : i) : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r[u] >= 0) {
        sc[u] = scores[r[u]]; lb[u] = label[r[u]]; wt[u] = weight ? weight[r[u]] : 1.0f;
        leaf[u] = min((int)leaf_of_row[r[u]], num_leaves - 1);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r[u] >= 0) {
        const double s_new = sc[u] + lv[leaf[u]];
        scores[r[u]] = s_new;
        float gg, hh;
        grad_hess(obj, s_new, lb[u], wt[u], &gg, &hh);
        g[r[u]] = gg;
        h[r[u]] = hh;
        mg = fmaxf(mg, fabsf(gg));
        mh = fmaxf(mh, hh);
      }
    }
  }
  publish_block_max(maxbits, mg, mh);
}

// walk_leaves_lds_kernel's walk with leaf_value_add_gradients_kernel's work at the leaf (kernels.hpp: walk_update_gradients).
template <int R>
__global__ __launch_bounds__(256) void walk_update_kernel(ObjectiveSpec obj, const TreeNodeDev* __restrict__ nodes, int n_internal,
                                                          const double* __restrict__ leaf_value, int num_leaves,
                                                          const uint4* __restrict__ recs, int64_t N, const int32_t* __restrict__ rows, int64_t n,
                                                          double* __restrict__ scores, const float* __restrict__ label,
                                                          const float* __restrict__ weight, float* __restrict__ g, float* __restrict__ h,
                                                          uint32_t* maxbits, const uint64_t* __restrict__ node_sets) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kStride = 36;
  int4* sn = reinterpret_cast<int4*>(smem);
  double* lv = reinterpret_cast<double*>(smem + (size_t)n_internal * sizeof(TreeNodeDev));
  unsigned char* sbins = reinterpret_cast<unsigned char*>(lv + num_leaves);   // [R * 256 rows][36 bytes], 4-byte aligned
  const int4* gn = reinterpret_cast<const int4*>(nodes);
  for (int i = threadIdx.x; i < n_internal; i += blockDim.x) sn[i] = gn[i];
  for (int i = threadIdx.x; i < num_leaves; i += blockDim.x) lv[i] = leaf_value[i];
  __syncthreads();
  const TreeNodeDev* tn = reinterpret_cast<const TreeNodeDev*>(smem);
  float mg = 0.f, mh = 0.f;
  for (int64_t base = (int64_t)blockIdx.x * (R * 256); base < n; base += (int64_t)gridDim.x * (R * 256)) {
    int64_t r[R];
    double sc[R];
    float lb[R], wt[R];
    int node[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int64_t i = base + k * 256 + threadIdx.x;
      r[k] = i < n ? (rows ? (int64_t)rows[i] : i) : -1;
      if (r[k] >= 0 && !LGA_DEV_CHECK((unsigned long long)r[k] < (unsigned long long)N, kCheckWalkRow, r[k], N)) r[k] = -1;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      node[k] = (r[k] >= 0 && num_leaves > 1) ? 0 : ~0;
      if (r[k] >= 0) {
        const uint4 a = recs[r[k] * 2], b = recs[r[k] * 2 + 1];
        unsigned* dst = reinterpret_cast<unsigned*>(sbins + (size_t)(k * 256 + (int)threadIdx.x) * kStride);   // only this lane reads it back
        dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
        dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
        sc[k] = scores[r[k]];
        lb[k] = label[r[k]];
        wt[k] = weight ? weight[r[k]] : 1.0f;
      }
    }
    bool more = true;
    while (more) {
      more = false;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (node[k] >= 0) {
          const TreeNodeDev t = tn[node[k]];
          const int b = sbins[(size_t)(k * 256 + (int)threadIdx.x) * kStride + ((unsigned)t.feature & 31u)];
          const int thr = t.meta & 255, dl = (t.meta >> 8) & 1, mb = (t.meta >> 16) - 1;
          const bool left = (t.meta & 1024) ? ((node_sets[(int64_t)node[k] * kCatSetWords + (b >> 6)] >> (b & 63)) & 1ull) != 0
                            : (t.meta & 512) ? (b == thr) : ((b == mb) ? (dl != 0) : (b <= thr));
          node[k] = left ? t.left : t.right;
          if (!LGA_DEV_CHECK(node[k] < n_internal && ~node[k] < num_leaves, kCheckWalkNode, node[k], n_internal)) { node[k] = ~0; r[k] = -1; }
          more = more || node[k] >= 0;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (r[k] >= 0) {
        const double s_new = sc[k] + lv[min(~node[k], num_leaves - 1)];
        scores[r[k]] = s_new;
        float gg, hh;
        grad_hess(obj, s_new, lb[k], wt[k], &gg, &hh);
        g[r[k]] = gg;
        h[r[k]] = hh;
        mg = fmaxf(mg, fabsf(gg));
        mh = fmaxf(mh, hh);
      }
    }
  }
  publish_block_max(maxbits, mg, mh);
}

__global__ __launch_bounds__(256) void scale_add_kernel(double* __restrict__ scores, const int32_t* __restrict__ rows,
                                                        int64_t n, double scale, double add) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows ? rows[i] : i;
    scores[r] = scores[r] * scale + add;
  }
}

__global__ __launch_bounds__(256) void init_scores_kernel(double* __restrict__ scores, const double* __restrict__ init,
                                                          const int32_t* __restrict__ rows, int64_t n, double v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows ? rows[i] : i;
    scores[r] = v + (init ? init[r] : 0.0);
  }
}

// ---------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------
// Sums are reduced in a fixed order so that a metric is bit-reproducible: every workgroup stores its
// partial sums, and the workgroup that finishes last adds them up (lane-strided, then a fixed shuffle tree).
// out: [3 results | arrival counter | 4 unused | 3 partials per workgroup ...].
template <int NV>
__device__ __forceinline__ void ordered_finish(const double (&v)[NV], double* out, double* lds_scratch) {
  __shared__ bool last;
  double t[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) t[j] = block_sum_f64<4>(v[j], lds_scratch);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) out[8 + (size_t)blockIdx.x * 3 + j] = t[j];
    __threadfence();
    const unsigned old = atomicAdd(reinterpret_cast<unsigned*>(out + 3), 1u);
    last = old == gridDim.x - 1;
  }
  __syncthreads();
  if (!last || threadIdx.x >= 64) return;
  __threadfence();
  double acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.0;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += 64) {
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += __builtin_nontemporal_load(&out[8 + (size_t)b * 3 + j]);
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = wave_sum_f64(acc[j]);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) out[j] = acc[j];
  }
}

__global__ __launch_bounds__(256) void metric_sum_kernel(int32_t metric, ObjectiveSpec obj,
                                                         const double* __restrict__ scores,
                                                         const float* __restrict__ label,
                                                         const float* __restrict__ weight,
                                                         const int32_t* __restrict__ rows, int64_t n, double* out) {
  __shared__ double scratch[4];
  double v[2] = {0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows ? rows[i] : i;
    const double w = weight ? (double)weight[r] : 1.0;
    const double raw = scores[r];
    const double pred = transform_score(obj.kind, obj.sigmoid, raw);
    v[0] += w * point_loss(metric, pred, raw, label[r], obj);
    v[1] += w;
  }
  ordered_finish<2>(v, out, scratch);
}

__global__ __launch_bounds__(256) void multi_metric_sum_kernel(int32_t metric, ObjectiveSpec obj,
                                                               const double* __restrict__ scores, int64_t stride,
                                                               const float* __restrict__ label,
                                                               const float* __restrict__ weight,
                                                               const int32_t* __restrict__ rows, int64_t n, double* out) {
  __shared__ double scratch[4];
  double v[2] = {0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows ? rows[i] : i;
    const double w = weight ? (double)weight[r] : 1.0;
    v[0] += w * multi_point_loss(metric, obj, scores, stride, r, label[r]);
    v[1] += w;
  }
  ordered_finish<2>(v, out, scratch);
}

__global__ __launch_bounds__(256) void label_sums_kernel(const float* __restrict__ label,
                                                         const float* __restrict__ weight,
                                                         const int32_t* __restrict__ rows, int64_t n, int sqrt_labels,
                                                         double* out) {
  __shared__ double scratch[4];
  double v[3] = {0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows ? rows[i] : i;
    const double w = weight ? (double)weight[r] : 1.0;
    v[0] += w * (double)(sqrt_labels ? sqrt_label(label[r]) : label[r]);
    v[1] += w;
    if (label[r] > 0.0f) v[2] += w;
  }
  ordered_finish<3>(v, out, scratch);
}

template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const T* __restrict__ src, const int32_t* __restrict__ rows,
                                                     int64_t n, T* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[rows ? rows[i] : i];
}
template <typename T>
__global__ __launch_bounds__(256) void scatter_kernel(const T* __restrict__ src, const int32_t* __restrict__ rows,
                                                      int64_t n, T* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[rows ? rows[i] : i] = src[i];
}
__global__ __launch_bounds__(256) void iota_kernel(int32_t* dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (int32_t)i;
}

}  // namespace

// ---------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------
#define LGA_LAUNCH_CHECK() LGA_HIP_CHECK(hipGetLastError())

#ifdef LGA_DEBUG
namespace {
int* g_check_sink_host = nullptr;   // device buffer the kernels' g_check_sink points at
void ensure_check_sink() {
  if (g_check_sink_host) return;
  LGA_HIP_CHECK(hipMalloc((void**)&g_check_sink_host, 8 * sizeof(int)));
  LGA_HIP_CHECK(hipMemset(g_check_sink_host, 0, 8 * sizeof(int)));
  LGA_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_check_sink), &g_check_sink_host, sizeof(int*)));
}
}  // namespace
#endif

bool device_checks_enabled() {
#ifdef LGA_DEBUG
  ensure_check_sink();
  return true;
#else
  return false;
#endif
}

std::string take_check_failure(hipStream_t s) {
#ifdef LGA_DEBUG
  ensure_check_sink();
  int rec[8];
  LGA_HIP_CHECK(hipStreamSynchronize(s));
  LGA_HIP_CHECK(hipMemcpy(rec, g_check_sink_host, sizeof(rec), hipMemcpyDeviceToHost));
  if (rec[0] == 0) return std::string();
  LGA_HIP_CHECK(hipMemset(g_check_sink_host, 0, 8 * sizeof(int)));
  static const char* what[] = {"?", "histogram: row id outside the dataset", "partition: row id outside the dataset",
                               "partition: scatter position outside the leaf's range",
                               "tree walk: row id outside the dataset", "tree walk: child index outside the tree"};
  const long long value = ((long long)rec[4] << 32) | (unsigned)rec[3], bound = ((long long)rec[6] << 32) | (unsigned)rec[5];
  std::ostringstream os;
  os << "device check failed (" << rec[0] << " hits): " << what[rec[1] >= 1 && rec[1] <= 5 ? rec[1] : 0] << " (value "
     << value << ", bound " << bound << ", workgroup " << rec[2] << ")";
  return os.str();
#else
  (void)s;
  return std::string();
#endif
}

void gradients(const ObjectiveSpec& obj, const double* scores, const float* label, const float* weight,
               const int32_t* rows, int64_t n, float* g, float* h, uint32_t* maxbits, hipStream_t s) {
  LGA_HIP_CHECK(hipMemsetAsync(maxbits, 0, 2 * sizeof(uint32_t), s));
  if (n <= 0) return;
  hipLaunchKernelGGL(gradients_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, obj, scores, label, weight, rows, n, g,
                     h, maxbits);
  LGA_LAUNCH_CHECK();
}

void goss_weights(const float* g, const float* h, int num_class, int64_t stride, const int32_t* rows, int64_t n,
                  float* w, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(goss_weights_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, g, h, num_class, stride, rows, n, w);
  LGA_LAUNCH_CHECK();
}

void kth_largest(const float* v, int64_t n, int64_t k, uint32_t* scratch, float* out, hipStream_t s) {
  if (n <= 0) return;
  k = std::max<int64_t>(1, std::min(k, n));
  hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, s, scratch, (unsigned long long)k);
  const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
  for (int p = 0; p < 3; ++p) {
    hipLaunchKernelGGL(select_count_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, v, n, shifts[p], bits[p], scratch);
    hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(256), 0, s, scratch, shifts[p], bits[p], p == 2 ? 1 : 0, out);
  }
  LGA_LAUNCH_CHECK();
}

void goss_amplify(float* g, float* h, int num_class, int64_t stride, const int32_t* rows, int64_t n,
                  const float* threshold, float multiplier, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(goss_amplify_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, g, h, num_class, stride, rows, n,
                     threshold, multiplier);
  LGA_LAUNCH_CHECK();
}

void absmax(const float* g, const float* h, const int32_t* rows, int64_t n, uint32_t* maxbits, hipStream_t s) {
  LGA_HIP_CHECK(hipMemsetAsync(maxbits, 0, 2 * sizeof(uint32_t), s));
  if (n <= 0) return;
  hipLaunchKernelGGL(absmax_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, g, h, rows, n, maxbits);
  LGA_LAUNCH_CHECK();
}

void quantize(const float* g, const float* h, const int32_t* rows, int64_t n, const uint32_t* maxbits, QuantScale* qs,
              int64_t* packed, bool wide, const RoundingSpec& rounding, hipStream_t s) {
  if (wide) hipLaunchKernelGGL(quantize_kernel<true>, dim3(grid_for(n, 256)), dim3(256), 0, s, g, h, rows, n, maxbits, qs, packed, rounding);
  else hipLaunchKernelGGL(quantize_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, s, g, h, rows, n, maxbits, qs, packed, rounding);
  LGA_LAUNCH_CHECK();
}

void partition(const SplitTask* tasks, const int32_t* chunk_task, const InlineTasks& inl, int total_chunks,
               bool any_multi_chunk, NodeRange* nodes, const int32_t* a0_src, const int32_t* a1_src, int32_t* a0_dst,
               int32_t* a1_dst, const uint8_t* bins_fm, int64_t N, uint64_t* mask, int32_t* chunk_left,
               int32_t* task_left, BagParams bag, int64_t* zero_ptr, int64_t zero_words, hipStream_t s, const GradArenas& grads) {
  if (total_chunks <= 0) return;
  if ((grads.arena[0] != nullptr) != (grads.arena[1] != nullptr) || (grads.arena[0] && !grads.by_row))
    throw std::runtime_error("partition: incomplete gradient arenas");
  // zero_words is even (slot sizes are); one trailing workgroup clears 1024 x 16 bytes
  const int zero_blocks = (int)(((zero_words >> 1) + kPartThreads * 4 - 1) / (kPartThreads * 4));
  hipLaunchKernelGGL(partition_count_kernel, dim3(total_chunks + zero_blocks), dim3(kPartThreads), 0, s, tasks,
                     chunk_task, inl, nodes, a0_src, a1_src, a0_dst, a1_dst, bins_fm, N, mask, chunk_left, task_left,
                     bag, total_chunks, zero_ptr, zero_words, grads);
  LGA_LAUNCH_CHECK();
  if (!any_multi_chunk) return;
  hipLaunchKernelGGL(partition_scatter_kernel, dim3(total_chunks), dim3(kPartThreads), 0, s, tasks, chunk_task, inl,
                     nodes, a0_src, a1_src, a0_dst, a1_dst, mask, chunk_left, task_left, 0, 0, grads);
  LGA_LAUNCH_CHECK();
}

void histogram(const SplitTask* tasks, const int32_t* chunk_task, const InlineTasks& inl, int total_chunks,
               int rows_per_chunk, const NodeRange* nodes, const int32_t* a0, const int32_t* a1,
               const uint8_t* bins_rt, int64_t N, const int64_t* packed, const PoolLayout& pool, RoundSlots round, int F,
               int tile_first, int tile_step, int tile_count, const QuantScale* wide_qs, hipStream_t s, const GradArenas& grads) {
  if (total_chunks <= 0 || tile_count <= 0) return;
  if (rows_per_chunk > (wide_qs ? kMaxUnitRowsWide : kMaxUnitRows)) throw std::runtime_error("histogram chunk exceeds the 32-bit LDS accumulator range");
  auto launch = [&](auto kernel, int threads) {
    hipLaunchKernelGGL(kernel, dim3(total_chunks, tile_count), dim3(threads), 0, s, tasks, chunk_task, inl,
                       rows_per_chunk, nodes, a0, a1, bins_rt, N, packed, pool, round, F, tile_first, tile_step, wide_qs, grads);
  };
  if (wide_qs) launch(hist_kernel<true>, kHistThreadsWide);
  else launch(hist_kernel<false>, kHistThreads);
  LGA_LAUNCH_CHECK();
}

void vote_gains(const InlineTasks& inl, const ScanArgs& a, int n_scan, double* gains, hipStream_t s) {
  if (n_scan <= 0) return;
  if (a.range.group_size != 8 || a.range.f_begin != 0 || a.range.f_end != a.F) throw std::runtime_error("vote: the local search covers every feature in groups of 8");
  hipLaunchKernelGGL(vote_gains_kernel, dim3(n_scan, std::max(1, (a.F + 7) / 8)), dim3(512), 0, s, inl, a, gains);
  LGA_LAUNCH_CHECK();
}
void vote_topk(const double* gains, int n_scan, int F, int top_k, Vote* votes, hipStream_t s) {
  if (n_scan <= 0) return;
  hipLaunchKernelGGL(vote_topk_kernel, dim3(n_scan), dim3(64), 0, s, gains, F, top_k, votes);
  LGA_LAUNCH_CHECK();
}
void vote_select(const InlineTasks& inl, const ScanArgs& a, const Vote* all, int ranks, int n_scan, int top_k, int n_sel,
                 int32_t* chosen, int64_t* buf, hipStream_t s) {
  if (n_scan <= 0) return;
  if (a.F > kVoteMaxFeatures || ranks * top_k > kVoteMaxVotes || n_sel > kVoteMaxSelected)
    throw std::runtime_error("vote: more features, votes or selected features than the kernel's tables hold");
  hipLaunchKernelGGL(vote_select_kernel, dim3(n_scan), dim3(256), 0, s, inl, a, all, ranks, n_scan, top_k, n_sel, chosen, buf);
  LGA_LAUNCH_CHECK();
}
void vote_final(const InlineTasks& inl, const ScanArgs& a, int n_scan, const int32_t* chosen, const int64_t* buf, int n_sel,
                Cand* best, hipStream_t s) {
  if (n_scan <= 0) return;
  hipLaunchKernelGGL(vote_final_kernel, dim3(n_scan), dim3(512), 0, s, inl, a, chosen, buf, n_sel, best);
  LGA_LAUNCH_CHECK();
}

void grow_round(const InlineTasks& inl, const GrowArgs& a, int total_units, bool root, hipStream_t s) {
  if (total_units <= 0) return;
  if (a.unit_rows <= 0 || a.unit_rows % kSubRows != 0 || a.unit_rows > (a.wide ? kMaxUnitRowsWide : kMaxUnitRows))
    throw std::runtime_error("growth round: bad unit size");
  if (a.F > kTileFeatures) throw std::runtime_error("row lines hold a single feature tile");
  if (a.wide && !a.quant.g && !a.qs) throw std::runtime_error("full-precision growth round without its quantisation scale");
  if (!root && a.quant.g) throw std::runtime_error("fused quantisation belongs to the root pass");
  auto launch = [&](auto kernel, int threads) { hipLaunchKernelGGL(kernel, dim3(total_units), dim3(threads), 0, s, inl, a); };
  const int tw = kGrowThreadsWide, tq = kGrowThreads;
  const bool allf = a.F > 24, quant = a.quant.g != nullptr, wide = a.wide != 0;
#define LGA_GROW_CASE(R, Q, W, A) if (root == R && quant == Q && wide == W && allf == A) launch(grow_kernel<R, Q, W, A>, W ? tw : tq);
  LGA_GROW_CASE(true, true, true, true) LGA_GROW_CASE(true, true, true, false)
  LGA_GROW_CASE(true, true, false, true) LGA_GROW_CASE(true, true, false, false)
  LGA_GROW_CASE(true, false, true, true) LGA_GROW_CASE(true, false, true, false)
  LGA_GROW_CASE(true, false, false, true) LGA_GROW_CASE(true, false, false, false)
  LGA_GROW_CASE(false, false, true, true) LGA_GROW_CASE(false, false, true, false)
  LGA_GROW_CASE(false, false, false, true) LGA_GROW_CASE(false, false, false, false)
#undef LGA_GROW_CASE
  LGA_LAUNCH_CHECK();
}

void grow_partition(const InlineTasks& inl, const GrowArgs& a, int total_chunks, hipStream_t s) {
  if (total_chunks <= 0) return;
  if (!a.part_cursor) throw std::runtime_error("single-pass partition without its cursors");
  switch (a.part_block_rows) {
    case 2048: hipLaunchKernelGGL((grow_partition_kernel<8, 256>), dim3(total_chunks), dim3(256), 0, s, inl, a); break;
    case 4096: hipLaunchKernelGGL((grow_partition_kernel<16, 256>), dim3(total_chunks), dim3(256), 0, s, inl, a); break;
    case 8192: hipLaunchKernelGGL((grow_partition_kernel<8, 1024>), dim3(total_chunks), dim3(1024), 0, s, inl, a); break;
    default: throw std::runtime_error("single-pass partition: blocks of 2048, 4096 or 8192 rows");
  }
  LGA_LAUNCH_CHECK();
}

void grow_flags(const InlineTasks& inl, const GrowArgs& a, int total_chunks, hipStream_t s) {
  if (total_chunks <= 0) return;
  hipLaunchKernelGGL(grow_flags_kernel, dim3(total_chunks), dim3(kPartThreads), 0, s, inl, a);
  LGA_LAUNCH_CHECK();
}

void grow_scatter(const SplitTask* staged, int n_staged, const InlineTasks& inl, int total_chunks, NodeRange* nodes,
                  const int32_t* a0_src, const int32_t* a1_src, int32_t* a0_dst, int32_t* a1_dst, const uint64_t* mask,
                  const int32_t* chunk_left, int32_t* task_left, hipStream_t s) {
  if (total_chunks <= 0) return;
  hipLaunchKernelGGL(partition_scatter_kernel, dim3(total_chunks), dim3(kPartThreads), 0, s, staged,
                     (const int32_t*)nullptr, inl, nodes, a0_src, a1_src, a0_dst, a1_dst, mask, chunk_left, task_left, 0,
                     n_staged, GradArenas{});
  LGA_LAUNCH_CHECK();
}

// Row r's 32 bin bytes are stored rotated by r: byte j of the line slot = feature (j + r) mod 32.
__global__ __launch_bounds__(256) void build_lines_kernel(const unsigned* __restrict__ recs, int64_t N, unsigned* __restrict__ lines) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
    unsigned src[8], dst[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) src[i] = recs[r * 8 + i];
    const unsigned rot = (unsigned)(r & 31);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = 0u;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const unsigned f = (j + rot) & 31u;                      // feature stored at byte j
      unsigned b = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) if ((f >> 2) == (unsigned)q) b = (src[q] >> ((f & 3u) * 8)) & 255u;
      dst[j >> 2] |= b << ((j & 3) * 8);
    }
    const int64_t line = r / 3;
    const int slot = (int)(r - line * 3);
#pragma unroll
    for (int i = 0; i < 8; ++i) lines[line * 32 + slot * 8 + i] = dst[i];
  }
}

__global__ __launch_bounds__(256) void store_line_gradients_kernel(unsigned long long* __restrict__ lines,
                                                                   const int64_t* __restrict__ packed, int64_t N) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < N; r += (int64_t)gridDim.x * blockDim.x) {
    const int64_t line = r / 3;
    lines[line * 16 + 12 + (r - line * 3)] = (unsigned long long)packed[r];
  }
}

void store_line_gradients(uint8_t* lines, const int64_t* packed, int64_t N, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(store_line_gradients_kernel, dim3(grid_for(N, 256, 8192)), dim3(256), 0, s,
                     reinterpret_cast<unsigned long long*>(lines), packed, N);
  LGA_LAUNCH_CHECK();
}

void build_row_lines(const uint8_t* bins_rt_tile0, int64_t N, uint8_t* lines, hipStream_t s) {
  if (N <= 0) return;
  LGA_HIP_CHECK(hipMemsetAsync(lines, 0, (size_t)row_line_bytes(N), s));
  hipLaunchKernelGGL(build_lines_kernel, dim3(grid_for(N, 256, 8192)), dim3(256), 0, s,
                     reinterpret_cast<const unsigned*>(bins_rt_tile0), N, reinterpret_cast<unsigned*>(lines));
  LGA_LAUNCH_CHECK();
}

void scan_blocks(const InlineTasks& inl, const ScanArgs& a, int launch_scan, hipStream_t s) {
  if (launch_scan <= 0) return;
  if (a.range.n_blocks < 1) throw std::runtime_error("split search needs at least one feature block");
  if (a.pairs && !a.root) {
    if (launch_scan != 2 * a.n_tasks || a.task0 != 0) throw std::runtime_error("paired split search takes a round's tasks whole");
    hipLaunchKernelGGL(scan_pairs_kernel, dim3(a.n_tasks, a.range.n_blocks), dim3(512), 0, s, inl, a);
  } else {
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(launch_scan, a.range.n_blocks), dim3(512), 0, s, inl, a);
  }
  LGA_LAUNCH_CHECK();
}

void merge_winners(const Cand* gathered, int ranks, int n_scan, int n_left, const int64_t* scan_totals, const QuantScale* qs,
                   const SplitParams& sp, const int32_t* task_left, char* host_out, unsigned seq, hipStream_t s) {
  if (n_scan <= 0) return;
  hipLaunchKernelGGL(merge_winners_kernel, dim3(1), dim3(256), 0, s, gathered, ranks, n_scan, n_left, scan_totals, qs, sp,
                     task_left, host_out, seq);
  LGA_LAUNCH_CHECK();
}

void leaf_add(const LeafSpan* leaves, int n_leaves, const int32_t* a0, const int32_t* a1, int64_t n_rows,
              double* scores, hipStream_t s) {
  if (n_rows <= 0 || n_leaves <= 0) return;
  size_t sm = ((size_t)n_leaves * sizeof(int32_t) + 15) & ~(size_t)15;
  hipLaunchKernelGGL(leaf_add_kernel, dim3(grid_for(n_rows, 256)), dim3(256), sm, s, leaves, n_leaves, a0, a1, n_rows,
                     scores);
  LGA_LAUNCH_CHECK();
}

void leaf_sums(const LeafSpan* spans, int n_spans, int n_leaves, const int32_t* a0, const int32_t* a1, int64_t n_rows,
               const float* g, const float* h, const QuantScale* qs, int extra_bits, int64_t* acc, hipStream_t s) {
  LGA_HIP_CHECK(hipMemsetAsync(acc, 0, sizeof(int64_t) * 2 * (size_t)n_leaves, s));
  if (n_rows <= 0 || n_spans <= 0) return;
  const size_t sm = ((size_t)n_spans * sizeof(int32_t) + 15) & ~(size_t)15;
  const int blocks = (int)((n_rows + kLeafSumChunk - 1) / kLeafSumChunk);
  hipLaunchKernelGGL(leaf_sums_kernel, dim3(blocks), dim3(256), sm, s, spans, n_spans, a0, a1, n_rows, g, h, qs, extra_bits,
                     reinterpret_cast<unsigned long long*>(acc));
  LGA_LAUNCH_CHECK();
}

void residual_keys(const LeafSpan* spans, int n_spans, const int32_t* a0, const int32_t* a1, int64_t n_rows,
                   const double* scores, const float* label, uint64_t* keys, int64_t npad, hipStream_t s) {
  if (npad <= 0 || n_spans <= 0) return;
  size_t sm = ((size_t)n_spans * sizeof(int32_t) + 15) & ~(size_t)15;
  hipLaunchKernelGGL(residual_keys_kernel, dim3(grid_for(npad, 256)), dim3(256), sm, s, spans, n_spans, a0, a1, n_rows,
                     scores, label, keys, npad);
  LGA_LAUNCH_CHECK();
}

void residual_items(const LeafSpan* spans, int n_spans, const int32_t* a0, const int32_t* a1, int64_t n_rows,
                    const double* scores, const float* label, const float* weight, int mape_scale,
                    int32_t* leaf_out, float* value_out, float* weight_out, hipStream_t s) {
  if (n_rows <= 0 || n_spans <= 0) return;
  size_t sm = ((size_t)n_spans * sizeof(int32_t) + 15) & ~(size_t)15;
  hipLaunchKernelGGL(residual_items_kernel, dim3(grid_for(n_rows, 256)), dim3(256), sm, s, spans, n_spans, a0, a1,
                     n_rows, scores, label, weight, mape_scale, leaf_out, value_out, weight_out);
  LGA_LAUNCH_CHECK();
}

void leaf_percentiles(const uint64_t* keys, int64_t n_rows, int n_leaves, double alpha, double* values,
                      int32_t* counts, hipStream_t s) {
  if (n_leaves <= 0) return;
  hipLaunchKernelGGL(leaf_percentiles_kernel, dim3((n_leaves + 255) / 256), dim3(256), 0, s, keys, n_rows, n_leaves,
                     alpha, values, counts);
  LGA_LAUNCH_CHECK();
}

bool walk_leaves(const TreeNodeDev* nodes, int n_internal, int num_leaves, const uint8_t* bins_fm, const uint8_t* records, int64_t N,
                 const int32_t* rows, int64_t n, const float* g, const float* h, const QuantScale* qs, int extra_bits,
                 uint16_t* leaf_of_row, int64_t* acc, const uint64_t* node_sets, hipStream_t s) {
  const size_t sm = (size_t)n_internal * sizeof(TreeNodeDev) + (size_t)2 * (size_t)num_leaves * sizeof(unsigned long long);
  if (sm > 48 * 1024 || num_leaves > 65535) return false;   // the caller takes the per-leaf row lists instead
  LGA_HIP_CHECK(hipMemsetAsync(acc, 0, sizeof(int64_t) * 2 * (size_t)num_leaves, s));
  if (n <= 0) return true;
  // rows per lane whose bin bytes wait in LDS (2 or 4); 0: the bytes are picked from registers.  Measured on the
  // benchmark: 2 -> 3.51, 4 -> 3.99 (45 KiB of LDS: three workgroups per CU), 0 -> 3.65 ms per tree (1 and 3 rows
  // per lane: 3.50 and 3.52).
  static const int lds_rows = std::getenv("LGA_WALK_LDS") ? std::atoi(std::getenv("LGA_WALK_LDS")) : 2;
  if (records && (lds_rows == 2 || lds_rows == 4)) {
    const size_t sm_rows = sm + (size_t)lds_rows * 256 * 36;
    // (every workgroup ends by adding its leaves' sums to the same 2 x num_leaves words, yet fewer, longer workgroups
    // measure slower: 4096 -> 3.16, 1024 -> 3.22, 512 -> 3.34 ms per tree on the 10.5 M x 28 benchmark)
    const dim3 grid(grid_for(n, 256 * lds_rows, 4096));
    auto launch = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, grid, dim3(256), sm_rows, s, nodes, n_internal, num_leaves, reinterpret_cast<const uint4*>(records), N,
                         rows, n, g, h, qs, extra_bits, leaf_of_row, reinterpret_cast<unsigned long long*>(acc), node_sets);
    };
    if (lds_rows == 2) launch(walk_leaves_lds_kernel<2>);
    else launch(walk_leaves_lds_kernel<4>);
    LGA_LAUNCH_CHECK();
    return true;
  }
  if (records) {
    hipLaunchKernelGGL(walk_leaves_records_kernel, dim3(grid_for(n, 1024, 2048)), dim3(256), sm, s, nodes, n_internal, num_leaves,
                       reinterpret_cast<const uint4*>(records), N, rows, n, g, h, qs, extra_bits, leaf_of_row,
                       reinterpret_cast<unsigned long long*>(acc), node_sets);
    LGA_LAUNCH_CHECK();
    return true;
  }
  hipLaunchKernelGGL(walk_leaves_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), sm, s, nodes, n_internal, num_leaves, bins_fm, N,
                     rows, n, g, h, qs, extra_bits, leaf_of_row, reinterpret_cast<unsigned long long*>(acc), node_sets);
  LGA_LAUNCH_CHECK();
  return true;
}

bool walk_update_gradients(const ObjectiveSpec& obj, const TreeNodeDev* nodes, int n_internal, const double* leaf_value, int num_leaves,
                           const uint8_t* records, int64_t N, const int32_t* rows, int64_t n, double* scores, const float* label,
                           const float* weight, float* g, float* h, uint32_t* maxbits, const uint64_t* node_sets, hipStream_t s) {
  constexpr int kRows = 2;
  const size_t sm = (size_t)n_internal * sizeof(TreeNodeDev) + (size_t)num_leaves * sizeof(double) + (size_t)kRows * 256 * 36;
  if (!records || sm > 60 * 1024 || num_leaves > 65535) return false;
  LGA_HIP_CHECK(hipMemsetAsync(maxbits, 0, 2 * sizeof(uint32_t), s));
  if (n <= 0) return true;
  hipLaunchKernelGGL(walk_update_kernel<kRows>, dim3(grid_for(n, 256 * kRows, 4096)), dim3(256), sm, s, obj, nodes, n_internal, leaf_value,
                     num_leaves, reinterpret_cast<const uint4*>(records), N, rows, n, scores, label, weight, g, h, maxbits, node_sets);
  LGA_LAUNCH_CHECK();
  return true;
}

void leaf_value_add(const uint16_t* leaf_of_row, const double* leaf_value, int num_leaves, const int32_t* rows, int64_t n,
                    double* scores, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(leaf_value_add_kernel, dim3(grid_for(n, 256)), dim3(256), (size_t)num_leaves * sizeof(double), s, leaf_of_row,
                     leaf_value, num_leaves, rows, n, scores);
  LGA_LAUNCH_CHECK();
}

void leaf_value_add_gradients(const ObjectiveSpec& obj, const uint16_t* leaf_of_row, const double* leaf_value, int num_leaves,
                              const int32_t* rows, int64_t n, double* scores, const float* label, const float* weight, float* g,
                              float* h, uint32_t* maxbits, hipStream_t s) {
  LGA_HIP_CHECK(hipMemsetAsync(maxbits, 0, 2 * sizeof(uint32_t), s));
  if (n <= 0) return;
  hipLaunchKernelGGL(leaf_value_add_gradients_kernel, dim3(grid_for(n, 256)), dim3(256), (size_t)num_leaves * sizeof(double), s, obj,
                     leaf_of_row, leaf_value, num_leaves, rows, n, scores, label, weight, g, h, maxbits);
  LGA_LAUNCH_CHECK();
}

void traverse_add(const TreeNodeDev* nodes, int n_internal, const double* leaf_value, int num_leaves,
                  const uint8_t* bins_fm, int64_t N, const int32_t* rows, int64_t n, double sign, double* scores, const uint64_t* node_sets,
                  hipStream_t s) {
  if (n <= 0) return;
  const size_t sm = (size_t)n_internal * sizeof(TreeNodeDev) + (size_t)num_leaves * sizeof(double);
  if (sm <= 48 * 1024)
    hipLaunchKernelGGL(traverse_add_kernel<true>, dim3(grid_for(n, 256)), dim3(256), sm, s, nodes, n_internal,
                       leaf_value, num_leaves, bins_fm, N, rows, n, sign, scores, node_sets);
  else
    hipLaunchKernelGGL(traverse_add_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, s, nodes, n_internal,
                       leaf_value, num_leaves, bins_fm, N, rows, n, sign, scores, node_sets);
  LGA_LAUNCH_CHECK();
}

namespace {
// One workgroup per (slot, feature) item: the feature's gradient and hessian histogram rows, next to each other.
__global__ __launch_bounds__(kMaxBins) void gather_feature_hist_kernel(PoolLayout L, const int32_t* __restrict__ items,
                                                                       int64_t* __restrict__ out) {
  const int slot = items[2 * blockIdx.x], f = items[2 * blockIdx.x + 1];
  const int64_t* blk = feat_block(L, old_slot_ref(L, slot), f);
  out[(int64_t)blockIdx.x * 2 * kMaxBins + threadIdx.x] = blk[threadIdx.x];
  out[(int64_t)blockIdx.x * 2 * kMaxBins + kMaxBins + threadIdx.x] = blk[kMaxBins + threadIdx.x];
}
}  // namespace

void gather_feature_hist(const PoolLayout& pool, const int32_t* items, int n_items, int64_t* out, hipStream_t s) {
  if (n_items <= 0) return;
  hipLaunchKernelGGL(gather_feature_hist_kernel, dim3(n_items), dim3(kMaxBins), 0, s, pool, items, out);
  LGA_LAUNCH_CHECK();
}

void scale_add(double* scores, const int32_t* rows, int64_t n, double scale, double add, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scale_add_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, scores, rows, n, scale, add);
  LGA_LAUNCH_CHECK();
}

void init_scores(double* scores, const double* init, const int32_t* rows, int64_t n, double v, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(init_scores_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, scores, init, rows, n, v);
  LGA_LAUNCH_CHECK();
}

void metric_sum(int32_t metric, const ObjectiveSpec& obj, const double* scores, const float* label,
                const float* weight, const int32_t* rows, int64_t n, double* out2, hipStream_t s) {
  LGA_HIP_CHECK(hipMemsetAsync(out2, 0, 4 * sizeof(double), s));
  if (n <= 0) return;
  hipLaunchKernelGGL(metric_sum_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, metric, obj, scores, label,
                     weight, rows, n, out2);
  LGA_LAUNCH_CHECK();
}

void multi_metric_sum(int32_t metric, const ObjectiveSpec& obj, const double* scores, int64_t stride, const float* label,
                      const float* weight, const int32_t* rows, int64_t n, double* out2, hipStream_t s) {
  LGA_HIP_CHECK(hipMemsetAsync(out2, 0, 4 * sizeof(double), s));
  if (n <= 0) return;
  hipLaunchKernelGGL(multi_metric_sum_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, metric, obj, scores, stride,
                     label, weight, rows, n, out2);
  LGA_LAUNCH_CHECK();
}

void multiclass_gradients(const ObjectiveSpec& obj, const double* scores, int64_t stride, const float* label,
                          const float* weight, const int32_t* rows, int64_t n, float* g, float* h, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(multiclass_gradients_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, obj, scores, stride, label,
                     weight, rows, n, g, h);
  LGA_LAUNCH_CHECK();
}

void label_sums(const float* label, const float* weight, const int32_t* rows, int64_t n, bool sqrt_labels, double* out3,
                hipStream_t s) {
  LGA_HIP_CHECK(hipMemsetAsync(out3, 0, 4 * sizeof(double), s));
  if (n <= 0) return;
  hipLaunchKernelGGL(label_sums_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, s, label, weight, rows, n,
                     sqrt_labels ? 1 : 0, out3);
  LGA_LAUNCH_CHECK();
}

void gather_f64(const double* src, const int32_t* rows, int64_t n, double* dst, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_kernel<double>, dim3(grid_for(n, 256)), dim3(256), 0, s, src, rows, n, dst);
  LGA_LAUNCH_CHECK();
}
void gather_f32(const float* src, const int32_t* rows, int64_t n, float* dst, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, s, src, rows, n, dst);
  LGA_LAUNCH_CHECK();
}
void scatter_f32(const float* src, const int32_t* rows, int64_t n, float* dst, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scatter_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, s, src, rows, n, dst);
  LGA_LAUNCH_CHECK();
}
void iota(int32_t* dst, int64_t n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, dst, n);
  LGA_LAUNCH_CHECK();
}

}  // namespace k
}  // namespace lga

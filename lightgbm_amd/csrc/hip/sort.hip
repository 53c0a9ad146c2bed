// Bitonic sort for gfx950: 2048-key blocks are sorted and merged in LDS (strides <= 1024, 1024 lanes,
// two keys each); larger strides run as global compare-exchange passes that take two strides at a time
// (four keys per lane), which halves the number of trips through HBM.
#include "sort.hpp"
#include "device.hpp"
#include <algorithm>

namespace lga {

namespace {

constexpr int kSortBlock = 2048;     // keys per workgroup in the LDS phases
constexpr int kSortThreads = 1024;

__device__ __forceinline__ void cmp_swap(uint64_t& a, uint64_t& b, bool up) {
  if ((a > b) == up) { uint64_t t = a; a = b; b = t; }
}

// Sorts every 2048-key block completely (k = 2 .. 2048) in LDS.
__global__ __launch_bounds__(kSortThreads) void bitonic_block_sort_kernel(uint64_t* __restrict__ keys, int64_t npad) {
  __shared__ uint64_t s[kSortBlock];
  uint64_t* base = keys + (int64_t)blockIdx.y * npad + (int64_t)blockIdx.x * kSortBlock;
  const int t = threadIdx.x;
  s[t] = base[t];
  s[t + kSortThreads] = base[t + kSortThreads];
  __syncthreads();
  const int64_t g0 = (int64_t)blockIdx.x * kSortBlock;
  for (int k = 2; k <= kSortBlock; k <<= 1) {
    for (int j = k >> 1; j >= 1; j >>= 1) {
      const int i = ((t / j) * 2 * j) + (t % j);      // lower index of the pair
      const bool up = (((g0 + i) & k) == 0);
      uint64_t a = s[i], b = s[i + j];
      cmp_swap(a, b, up);
      s[i] = a; s[i + j] = b;
      __syncthreads();
    }
  }
  base[t] = s[t];
  base[t + kSortThreads] = s[t + kSortThreads];
}

// One global compare-exchange step of stage k with stride j (j >= 2048).
__global__ __launch_bounds__(256) void bitonic_global_step_kernel(uint64_t* __restrict__ keys, int64_t npad, int64_t j,
                                                                  int64_t k) {
  uint64_t* base = keys + (int64_t)blockIdx.y * npad;
  const int64_t half = npad >> 1;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < half; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = ((t / j) * 2 * j) + (t % j);
    const bool up = ((i & k) == 0);
    uint64_t a = base[i], b = base[i + j];
    if ((a > b) == up) { base[i] = b; base[i + j] = a; }
  }
}

// Two steps of stage k in one pass: strides j and j/2 (j/2 >= 2048).  A lane owns the four keys
// i, i+j/2, i+j, i+3j/2, which are closed under both exchanges and share one sort direction.
__global__ __launch_bounds__(256) void bitonic_global_step2_kernel(uint64_t* __restrict__ keys, int64_t npad, int64_t j,
                                                                   int64_t k) {
  uint64_t* base = keys + (int64_t)blockIdx.y * npad;
  const int64_t quarter = npad >> 2, h = j >> 1;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < quarter; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = ((t / h) * 4 * h) + (t % h);
    const bool up = ((i & k) == 0);
    uint64_t a = base[i], b = base[i + h], c = base[i + j], d = base[i + j + h];
    cmp_swap(a, c, up);
    cmp_swap(b, d, up);
    cmp_swap(a, b, up);
    cmp_swap(c, d, up);
    base[i] = a; base[i + h] = b; base[i + j] = c; base[i + j + h] = d;
  }
}

// Tail of stage k: strides 1024 .. 1 inside each 2048-key block, in LDS.
__global__ __launch_bounds__(kSortThreads) void bitonic_block_merge_kernel(uint64_t* __restrict__ keys, int64_t npad,
                                                                           int64_t k) {
  __shared__ uint64_t s[kSortBlock];
  uint64_t* base = keys + (int64_t)blockIdx.y * npad + (int64_t)blockIdx.x * kSortBlock;
  const int t = threadIdx.x;
  s[t] = base[t];
  s[t + kSortThreads] = base[t + kSortThreads];
  __syncthreads();
  const int64_t g0 = (int64_t)blockIdx.x * kSortBlock;
  for (int j = kSortBlock >> 1; j >= 1; j >>= 1) {
    const int i = ((t / j) * 2 * j) + (t % j);
    const bool up = (((g0 + i) & k) == 0);
    uint64_t a = s[i], b = s[i + j];
    cmp_swap(a, b, up);
    s[i] = a; s[i + j] = b;
    __syncthreads();
  }
  base[t] = s[t];
  base[t + kSortThreads] = s[t + kSortThreads];
}

}  // namespace

int64_t sort_padded_size(int64_t n) {
  int64_t p = kSortBlock;
  while (p < n) p <<= 1;
  return p;
}

void bitonic_sort_u64(uint64_t* keys, int64_t npad, int batches, hipStream_t s) {
  if (batches <= 0) return;
  if (npad < kSortBlock || (npad & (npad - 1)) != 0) throw std::runtime_error("bitonic_sort_u64: bad padded size");
  dim3 gb((unsigned)(npad / kSortBlock), (unsigned)batches);
  hipLaunchKernelGGL(bitonic_block_sort_kernel, gb, dim3(kSortThreads), 0, s, keys, npad);
  LGA_HIP_CHECK(hipGetLastError());
  for (int64_t k = (int64_t)kSortBlock * 2; k <= npad; k <<= 1) {
    int64_t j = k >> 1;
    while (j >= kSortBlock) {
      if ((j >> 1) >= kSortBlock) {
        dim3 gg((unsigned)std::min<int64_t>((npad / 4 + 255) / 256, 4096), (unsigned)batches);
        hipLaunchKernelGGL(bitonic_global_step2_kernel, gg, dim3(256), 0, s, keys, npad, j, k);
        j >>= 2;
      } else {
        dim3 gg((unsigned)std::min<int64_t>((npad / 2 + 255) / 256, 4096), (unsigned)batches);
        hipLaunchKernelGGL(bitonic_global_step_kernel, gg, dim3(256), 0, s, keys, npad, j, k);
        j >>= 1;
      }
      LGA_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(bitonic_block_merge_kernel, gb, dim3(kSortThreads), 0, s, keys, npad, k);
    LGA_HIP_CHECK(hipGetLastError());
  }
}

}  // namespace lga

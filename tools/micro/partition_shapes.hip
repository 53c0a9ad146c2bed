// Micro-benchmark: the single-pass partition (rows in registers, one reservation atomic per workgroup, lefts forward /
// rights backward) at several workgroup shapes, on one 10.5 M-row "leaf" (a) with rows in order and (b) with the rows
// of 32 interleaved leaves (stride-32 row ids: every gathered byte is its own cache line).  Prints microseconds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// leaves: the n rows are split into `leaves` equal ranges, each with its own cursor
template <int U, int T, bool ATOMIC, int PAD = 1>
__global__ __launch_bounds__(T) void part(const int* __restrict__ rows, const uint8_t* __restrict__ col, int* __restrict__ dst,
                                          unsigned long long* cursor, int per_leaf, int blocks_per_leaf) {
  constexpr int kChunk = U * T, kWords = kChunk / 64, kWaves = T / 64;
  constexpr int kPerLane = (kWords + 63) / 64, kLanes = kWords / kPerLane;
  __shared__ unsigned long long smask[kWords];
  __shared__ int word_prefix[kWords + 1];
  __shared__ int s_before[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int leaf = blockIdx.x / blocks_per_leaf, c = blockIdx.x % blocks_per_leaf;
  const int* src = rows + (size_t)leaf * per_leaf;
  int* out = dst + (size_t)leaf * per_leaf;
  const int base = c * kChunk;
  const int in_block = min(kChunk, per_leaf - base);
  int row[U];
#pragma unroll
  for (int k = 0; k < U; ++k) { const int pos = base + k * T + tid; row[k] = pos < per_leaf ? src[pos] : -1; }
  int bin[U];
#pragma unroll
  for (int k = 0; k < U; ++k) bin[k] = row[k] >= 0 ? (int)col[row[k]] : 255;
  unsigned long long mine[U];
#pragma unroll
  for (int k = 0; k < U; ++k) { mine[k] = __ballot(bin[k] < 128); if (lane == 0) smask[k * kWaves + wave] = mine[k]; }
  __syncthreads();
  if (tid < 64) {
    int pcs[kPerLane], pc = 0;
#pragma unroll
    for (int q = 0; q < kPerLane; ++q) { pcs[q] = tid < kLanes ? __popcll(smask[tid * kPerLane + q]) : 0; pc += pcs[q]; }
    int incl = pc;
#pragma unroll
    for (int d = 1; d < kLanes; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (tid < kLanes) { int run = incl - pc;
#pragma unroll
      for (int q = 0; q < kPerLane; ++q) { word_prefix[tid * kPerLane + q] = run; run += pcs[q]; } }
    if (tid == kLanes - 1) {
      const int own = incl, own_right = in_block - own;
      unsigned long long before;
      if (ATOMIC) before = atomicAdd(&cursor[leaf * PAD], (unsigned long long)own | ((unsigned long long)own_right << 32));
      else before = (unsigned long long)(base / 2) | ((unsigned long long)(base / 2) << 32);   // no communication: not a partition, a timing floor
      s_before[0] = (int)(before & 0xffffffffull); s_before[1] = (int)(before >> 32);
      if (ATOMIC && s_before[0] + s_before[1] + in_block == per_leaf) cursor[leaf * PAD] = 0ull;
    }
  }
  __syncthreads();
  const int lb = s_before[0], rb = s_before[1];
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    if (row[k] >= 0) {
      const int w = k * kWaves + wave;
      const bool left = (mine[k] >> lane) & 1ull;
      const int lrank = word_prefix[w] + __popcll(mine[k] & below);
      int d = left ? lb + lrank : per_leaf - 1 - (rb + (w * 64 + lane - lrank));
      d = min(max(d, 0), per_leaf - 1);
      out[d] = row[k];
    }
  }
}

int main() {
  const int n = 10485760;   // 32 leaves x 327680
  int *rows, *rows32, *dst; uint8_t* col; unsigned long long* cursor;
  CK(hipMalloc(&rows, (size_t)n * 4)); CK(hipMalloc(&rows32, (size_t)n * 4)); CK(hipMalloc(&dst, (size_t)n * 4)); CK(hipMalloc(&col, n));
  CK(hipMalloc(&cursor, 8 * 64 * 512)); CK(hipMemset(cursor, 0, 8 * 64 * 512));
  std::vector<int> h(n), h32(n);
  const int per32 = n / 32;
  for (int i = 0; i < n; ++i) { h[i] = i; h32[i] = (i % per32) * 32 + i / per32; }
  std::vector<uint8_t> hc(n); for (int i = 0; i < n; ++i) hc[i] = (uint8_t)((i * 2654435761u) >> 24);
  CK(hipMemcpy(rows, h.data(), (size_t)n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rows32, h32.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(col, hc.data(), n, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %8.1f us\n", name, ms * 1000 / 20);
  };
#define RUN(U, T) { \
    const int b1 = (n + (U) * (T) - 1) / ((U) * (T)), b32 = (per32 + (U) * (T) - 1) / ((U) * (T)); \
    timeit("one leaf, rows in order   U=" #U " T=" #T " atomic", [&] { hipLaunchKernelGGL((part<U, T, true>), dim3(b1), dim3(T), 0, 0, rows, col, dst, cursor, n, b1); }); \
    timeit("one leaf, rows in order   U=" #U " T=" #T " no atomic", [&] { hipLaunchKernelGGL((part<U, T, false>), dim3(b1), dim3(T), 0, 0, rows, col, dst, cursor, n, b1); }); \
    timeit("32 leaves, stride-32 rows U=" #U " T=" #T " atomic", [&] { hipLaunchKernelGGL((part<U, T, true>), dim3(b32 * 32), dim3(T), 0, 0, rows32, col, dst, cursor, per32, b32); }); \
    timeit("32 leaves, stride-32 rows U=" #U " T=" #T " atomic, cursors 128 B apart", [&] { hipLaunchKernelGGL((part<U, T, true, 16>), dim3(b32 * 32), dim3(T), 0, 0, rows32, col, dst, cursor, per32, b32); }); \
    timeit("32 leaves, stride-32 rows U=" #U " T=" #T " atomic, cursors 4 KB apart", [&] { hipLaunchKernelGGL((part<U, T, true, 512>), dim3(b32 * 32), dim3(T), 0, 0, rows32, col, dst, cursor, per32, b32); }); \
    timeit("32 leaves, stride-32 rows U=" #U " T=" #T " no atomic", [&] { hipLaunchKernelGGL((part<U, T, false>), dim3(b32 * 32), dim3(T), 0, 0, rows32, col, dst, cursor, per32, b32); }); }
  RUN(8, 256) RUN(16, 256) RUN(8, 1024)
  return 0;
}

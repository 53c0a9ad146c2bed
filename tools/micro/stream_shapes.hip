// Micro-benchmark: how fast can "read row ids, gather one byte per row, ballot, write" shaped kernels go on this GPU,
// as a function of the workgroup shape?  (Orientation for the partition kernels; prints microseconds per variant.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U, int T>
__global__ __launch_bounds__(T) void flags_like(const int* __restrict__ rows, const uint8_t* __restrict__ col, unsigned long long* __restrict__ mask, int n) {
  const int base = blockIdx.x * (U * T);
  int r[U];
#pragma unroll
  for (int k = 0; k < U; ++k) { const int p = base + k * T + threadIdx.x; r[k] = p < n ? rows[p] : -1; }
  int b[U];
#pragma unroll
  for (int k = 0; k < U; ++k) b[k] = r[k] >= 0 ? col[r[k]] : 255;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const unsigned long long m = __ballot(b[k] < 128);
    if ((threadIdx.x & 63) == 0) mask[(size_t)(base + k * T + threadIdx.x) >> 6] = m;
  }
}
template <int U, int T>
__global__ __launch_bounds__(T) void flags_persistent(const int* __restrict__ rows, const uint8_t* __restrict__ col, unsigned long long* __restrict__ mask, int n) {
  for (int base = blockIdx.x * (U * T); base < n; base += gridDim.x * (U * T)) {
    int r[U];
#pragma unroll
    for (int k = 0; k < U; ++k) { const int p = base + k * T + threadIdx.x; r[k] = p < n ? rows[p] : -1; }
    int b[U];
#pragma unroll
    for (int k = 0; k < U; ++k) b[k] = r[k] >= 0 ? col[r[k]] : 255;
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const unsigned long long m = __ballot(b[k] < 128);
      if ((threadIdx.x & 63) == 0) mask[(size_t)(base + k * T + threadIdx.x) >> 6] = m;
    }
  }
}
template <int U, int T>
__global__ __launch_bounds__(T) void copy_like(const int* __restrict__ rows, int* __restrict__ out, int n) {
  const int base = blockIdx.x * (U * T);
  int r[U];
#pragma unroll
  for (int k = 0; k < U; ++k) { const int p = base + k * T + threadIdx.x; r[k] = p < n ? rows[p] : -1; }
#pragma unroll
  for (int k = 0; k < U; ++k) { const int p = base + k * T + threadIdx.x; if (p < n) out[p] = r[k]; }
}
// rows only (no gather): how much of the time is the dependent byte load?
template <int U, int T>
__global__ __launch_bounds__(T) void rows_only(const int* __restrict__ rows, unsigned long long* __restrict__ mask, int n) {
  const int base = blockIdx.x * (U * T);
  int r[U];
#pragma unroll
  for (int k = 0; k < U; ++k) { const int p = base + k * T + threadIdx.x; r[k] = p < n ? rows[p] : -1; }
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const unsigned long long m = __ballot((r[k] & 1) != 0);
    if ((threadIdx.x & 63) == 0) mask[(size_t)(base + k * T + threadIdx.x) >> 6] = m;
  }
}
// one atomic on ONE address per workgroup on top of flags_like
template <int U, int T>
__global__ __launch_bounds__(T) void flags_atomic(const int* __restrict__ rows, const uint8_t* __restrict__ col, unsigned long long* __restrict__ mask, unsigned long long* cursor, int n) {
  const int base = blockIdx.x * (U * T);
  int r[U];
#pragma unroll
  for (int k = 0; k < U; ++k) { const int p = base + k * T + threadIdx.x; r[k] = p < n ? rows[p] : -1; }
  int b[U];
#pragma unroll
  for (int k = 0; k < U; ++k) b[k] = r[k] >= 0 ? col[r[k]] : 255;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const unsigned long long m = __ballot(b[k] < 128);
    cnt += __popcll(m);
  }
  __shared__ unsigned long long s;
  if (threadIdx.x == 0) s = atomicAdd(cursor, (unsigned long long)cnt);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) mask[(size_t)(base + threadIdx.x) >> 6] = s;
}

int main() {
  const int n = 10500000;
  int* rows; uint8_t* col; unsigned long long* mask; int* out; unsigned long long* cursor;
  CK(hipMalloc(&rows, (size_t)n * 4)); CK(hipMalloc(&col, n)); CK(hipMalloc(&mask, (size_t)n / 8 + 4096)); CK(hipMalloc(&out, (size_t)n * 4));
  CK(hipMalloc(&cursor, 8)); CK(hipMemset(cursor, 0, 8));
  std::vector<int> h(n); for (int i = 0; i < n; ++i) h[i] = i;
  std::vector<uint8_t> hc(n); for (int i = 0; i < n; ++i) hc[i] = (uint8_t)((i * 2654435761u) >> 24);
  CK(hipMemcpy(rows, h.data(), (size_t)n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(col, hc.data(), n, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.1f us\n", name, ms * 1000 / 20);
    return 0;
  };
#define GRID(U, T) dim3((n + (U) * (T) - 1) / ((U) * (T)))
  timeit("flags_like U=8 T=256", [&] { hipLaunchKernelGGL((flags_like<8, 256>), GRID(8, 256), dim3(256), 0, 0, rows, col, mask, n); });
  timeit("flags_like U=4 T=256", [&] { hipLaunchKernelGGL((flags_like<4, 256>), GRID(4, 256), dim3(256), 0, 0, rows, col, mask, n); });
  timeit("flags_like U=16 T=256", [&] { hipLaunchKernelGGL((flags_like<16, 256>), GRID(16, 256), dim3(256), 0, 0, rows, col, mask, n); });
  timeit("flags_like U=32 T=256", [&] { hipLaunchKernelGGL((flags_like<32, 256>), GRID(32, 256), dim3(256), 0, 0, rows, col, mask, n); });
  timeit("flags_like U=8 T=1024", [&] { hipLaunchKernelGGL((flags_like<8, 1024>), GRID(8, 1024), dim3(1024), 0, 0, rows, col, mask, n); });
  timeit("flags_like U=8 T=64", [&] { hipLaunchKernelGGL((flags_like<8, 64>), GRID(8, 64), dim3(64), 0, 0, rows, col, mask, n); });
  timeit("flags_persistent U=8 T=256 grid 2048", [&] { hipLaunchKernelGGL((flags_persistent<8, 256>), dim3(2048), dim3(256), 0, 0, rows, col, mask, n); });
  timeit("flags_persistent U=8 T=256 grid 1024", [&] { hipLaunchKernelGGL((flags_persistent<8, 256>), dim3(1024), dim3(256), 0, 0, rows, col, mask, n); });
  timeit("flags_persistent U=4 T=256 grid 4096", [&] { hipLaunchKernelGGL((flags_persistent<4, 256>), dim3(4096), dim3(256), 0, 0, rows, col, mask, n); });
  timeit("rows_only U=8 T=256", [&] { hipLaunchKernelGGL((rows_only<8, 256>), GRID(8, 256), dim3(256), 0, 0, rows, mask, n); });
  timeit("copy_like U=8 T=256", [&] { hipLaunchKernelGGL((copy_like<8, 256>), GRID(8, 256), dim3(256), 0, 0, rows, out, n); });
  timeit("copy_like U=16 T=256", [&] { hipLaunchKernelGGL((copy_like<16, 256>), GRID(16, 256), dim3(256), 0, 0, rows, out, n); });
  timeit("flags_atomic U=8 T=256", [&] { hipLaunchKernelGGL((flags_atomic<8, 256>), GRID(8, 256), dim3(256), 0, 0, rows, col, mask, cursor, n); });
  timeit("flags_atomic U=16 T=256", [&] { hipLaunchKernelGGL((flags_atomic<16, 256>), GRID(16, 256), dim3(256), 0, 0, rows, col, mask, cursor, n); });
  timeit("flags_atomic U=32 T=256", [&] { hipLaunchKernelGGL((flags_atomic<32, 256>), GRID(32, 256), dim3(256), 0, 0, rows, col, mask, cursor, n); });
  timeit("hipMemcpyDtoD 42 MB", [&] { hipMemcpyAsync(out, rows, (size_t)n * 4, hipMemcpyDeviceToDevice, 0); });
  return 0;
}

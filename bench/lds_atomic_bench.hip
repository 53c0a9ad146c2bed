// Micro-benchmark behind the histogram kernel's design: throughput of LDS
// atomic adds on gfx950 for the access pattern of a gradient histogram
// (each lane adds to hist[feature k][random bin], k wave-uniform).
// Variants: one packed 64-bit add, two 32-bit integer adds, two fp32 adds;
// random bins vs conflict-free bins (lane == bin).
//   hipcc --offload-arch=gfx950 -O3 bench/lds_atomic_bench.hip -o /tmp/lds_bench && /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kThreads = 512;
constexpr int kFeatures = 32;
constexpr int kBins = 256;

template <int MODE>
__global__ __launch_bounds__(kThreads) void bench_kernel(const unsigned* __restrict__ bins, int iters, unsigned long long* out) {
  __shared__ unsigned long long lds[kFeatures * kBins];
  for (int i = threadIdx.x; i < kFeatures * kBins; i += kThreads) lds[i] = 0;
  __syncthreads();
  unsigned* lds32 = reinterpret_cast<unsigned*>(lds);
  float* ldsf = reinterpret_cast<float*>(lds);
  const unsigned long long v = 0x0000000100000003ull + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    // 8 dwords = 32 bins of one row
    const uint4 a = reinterpret_cast<const uint4*>(bins)[((size_t)blockIdx.x * iters + it) * kThreads * 2 + threadIdx.x * 2];
    const uint4 b = reinterpret_cast<const uint4*>(bins)[((size_t)blockIdx.x * iters + it) * kThreads * 2 + threadIdx.x * 2 + 1];
    const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < kFeatures; ++k) {
      const unsigned bin = (w[k >> 2] >> ((k & 3) * 8)) & 255u;
      if (MODE == 0) atomicAdd(&lds[k * kBins + bin], v);
      if (MODE == 1) { atomicAdd(&lds32[(k * kBins + bin) * 2], (unsigned)v); atomicAdd(&lds32[(k * kBins + bin) * 2 + 1], (unsigned)(v >> 32)); }
      if (MODE == 2) { atomicAdd(&ldsf[(k * kBins + bin) * 2], 1.0f); atomicAdd(&ldsf[(k * kBins + bin) * 2 + 1], 0.5f); }
      if (MODE == 3) { atomicAdd(&lds32[k * kBins + bin], (unsigned)v); }   // a single 32-bit add (lower bound)
    }
  }
  __syncthreads();
  unsigned long long s = 0;
  for (int i = threadIdx.x; i < kFeatures * kBins; i += kThreads) s += lds[i];
  if (s == 0x12345) out[0] = s;
}

template <int MODE>
double run(const unsigned* d_bins, int blocks, int iters, unsigned long long* d_out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(bench_kernel<MODE>, dim3(blocks), dim3(kThreads), 0, 0, d_bins, iters, d_out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(bench_kernel<MODE>, dim3(blocks), dim3(kThreads), 0, 0, d_bins, iters, d_out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5.0;
}

int main() {
  const int blocks = 1024, iters = 20;   // 1024 * 20 * 512 = 10.5 M rows
  const size_t rows = (size_t)blocks * iters * kThreads;
  std::vector<unsigned> h(rows * 8);
  unsigned long long* d_out;
  unsigned* d_bins;
  CHECK(hipMalloc(&d_out, 64));
  CHECK(hipMalloc(&d_bins, h.size() * 4));
  const char* names[4] = {"1 x ds_add_u64 (packed g|h)", "2 x ds_add_u32", "2 x ds_add_f32", "1 x ds_add_u32 (bound)"};
  for (int pattern = 0; pattern < 3; ++pattern) {
    // 0: uniform random bins, 1: conflict-free (bin = lane id mod 256), 2: skewed (16 hot bins)
    srand(1);
    for (size_t r = 0; r < rows; ++r)
      for (int k = 0; k < 32; ++k) {
        unsigned bin = pattern == 0 ? (rand() & 255) : pattern == 1 ? (unsigned)((r + k * 7) & 255) : (rand() & 15);
        unsigned& w = h[r * 8 + (k >> 2)];
        w = (w & ~(255u << ((k & 3) * 8))) | (bin << ((k & 3) * 8));
      }
    CHECK(hipMemcpy(d_bins, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const char* pn[3] = {"random bins", "conflict-free bins", "16 hot bins"};
    printf("pattern: %s (%zu rows x 32 features)\n", pn[pattern], rows);
    double t0 = run<0>(d_bins, blocks, iters, d_out);
    double t1 = run<1>(d_bins, blocks, iters, d_out);
    double t2 = run<2>(d_bins, blocks, iters, d_out);
    double t3 = run<3>(d_bins, blocks, iters, d_out);
    double ts[4] = {t0, t1, t2, t3};
    for (int m = 0; m < 4; ++m)
      printf("  %-30s %8.3f ms  %7.2f G row-features/s\n", names[m], ts[m], rows * 32.0 / ts[m] / 1e6);
  }
  return 0;
}

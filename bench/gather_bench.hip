// Micro-benchmark for the non-root histogram passes: a leaf's rows are a sorted, SPARSE subset of the
// dataset, so every row costs whole cache lines.  For row densities 1 .. 1/64 of a 10.5 M-row matrix it
// times, with the histogram kernel's launch shape (512 threads, 64 KiB LDS, 4 rows per lane and trip):
//   sep/load   : 32-byte bin record + 8-byte packed gradient from separate arrays, loads only
//   lines/load : the same bytes from 128-byte row lines (3 rows per line), loads only
//   sep/full   : separate arrays + 28 packed LDS atomics per row (random bins)
//   lines/full : row lines + LDS atomics
//   hipcc --offload-arch=gfx950 -O3 bench/gather_bench.hip -o bench/bin/gather_bench && bench/bin/gather_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kThreads = 512;
constexpr int kCells = 32 * 256;
constexpr int kFeat = 28;

__device__ __forceinline__ unsigned div3(unsigned r) { return (unsigned)(((unsigned long long)r * 0xAAAAAAABull) >> 33); }

// LAYOUT 0: separate record + gradient arrays; 1: 128-byte row lines; 2: 32-byte records whose last four
// bytes hold a 16:16 packed gradient (28 features).
template <int LAYOUT, bool ATOMICS>
__global__ __launch_bounds__(kThreads) void gather_kernel(const int* __restrict__ rows, int n, int rows_per_chunk,
                                                          const uint4* __restrict__ recs, const unsigned long long* __restrict__ packed,
                                                          unsigned long long* __restrict__ out) {
  __shared__ unsigned long long lds[kCells];
  for (int i = threadIdx.x; i < kCells; i += kThreads) lds[i] = 0ull;
  __syncthreads();
  const int lo = blockIdx.x * rows_per_chunk;
  const int hi = min(lo + rows_per_chunk, n);
  unsigned long long acc = 0;
  constexpr int U = 4;
  for (int i0 = lo + threadIdx.x; i0 < hi; i0 += kThreads * U) {
    int row[U];
    unsigned long long p[U];
    uint4 ra[U], rb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int i = i0 + u * kThreads; row[u] = i < hi ? rows[i] : -1; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (row[u] >= 0) {
        if (LAYOUT == 2) {
          ra[u] = recs[(int64_t)row[u] * 2]; rb[u] = recs[(int64_t)row[u] * 2 + 1];
          p[u] = ((unsigned long long)(long long)(short)(rb[u].w >> 16) << 32) | (rb[u].w & 0xffffu);
        } else if (LAYOUT == 3) {
          // 128-byte lines whose first and third row sit in ONE 64-byte half each (bins + gradient word), the second row
          // in what is left of both: 1.33 halves per row on average instead of 1.67
          const unsigned line = div3((unsigned)row[u]), slot = (unsigned)row[u] - line * 3u;
          const uint4* L = recs + (int64_t)line * 8;
          const unsigned ia = slot == 0 ? 0u : (slot == 1 ? 3u : 4u), ib = slot == 0 ? 1u : (slot == 1 ? 7u : 5u);
          const unsigned ip = slot == 0 ? 4u : (slot == 1 ? 13u : 12u);
          ra[u] = L[ia]; rb[u] = L[ib];
          p[u] = reinterpret_cast<const unsigned long long*>(L)[ip];
        } else if (LAYOUT == 1) {
          const unsigned line = div3((unsigned)row[u]), slot = (unsigned)row[u] - line * 3u;
          const uint4* L = recs + (int64_t)line * 8;
          ra[u] = L[slot * 2]; rb[u] = L[slot * 2 + 1];
          p[u] = reinterpret_cast<const unsigned long long*>(L)[12 + slot];
        } else {
          p[u] = packed[row[u]];
          ra[u] = recs[(int64_t)row[u] * 2]; rb[u] = recs[(int64_t)row[u] * 2 + 1];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (row[u] >= 0) {
        const unsigned w[8] = {ra[u].x, ra[u].y, ra[u].z, ra[u].w, rb[u].x, rb[u].y, rb[u].z, rb[u].w};
        if (ATOMICS) {
#pragma unroll
          for (int k = 0; k < kFeat; ++k) {
            const unsigned bin = (w[k >> 2] >> ((k & 3) * 8)) & 255u;
            atomicAdd(&lds[k * 256 + bin], p[u]);
          }
        } else {
          acc += p[u] + w[0] + w[1] + w[2] + w[3] + w[4] + w[5] + w[6] + w[7];
        }
      }
    }
  }
  __syncthreads();
  if (ATOMICS) { for (int i = threadIdx.x; i < kCells; i += kThreads) acc += lds[i]; }
  if (acc == 0x1234567ull) out[0] = acc;
}

template <int LAYOUT, bool ATOMICS>
double run(const int* rows, int n, int chunk, const void* recs, const unsigned long long* packed, unsigned long long* out) {
  const int blocks = (n + chunk - 1) / chunk;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((gather_kernel<LAYOUT, ATOMICS>), dim3(blocks), dim3(kThreads), 0, 0, rows, n, chunk, (const uint4*)recs, packed, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r)
    hipLaunchKernelGGL((gather_kernel<LAYOUT, ATOMICS>), dim3(blocks), dim3(kThreads), 0, 0, rows, n, chunk, (const uint4*)recs, packed, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.0 / 5.0;
}

int main() {
  const int64_t N = 10500000;
  std::vector<unsigned char> h_rec((size_t)N * 32);
  uint64_t s = 88172645463325252ull;
  for (auto& b : h_rec) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; b = (unsigned char)(s >> 32); }
  const int64_t n_lines = (N + 2) / 3;
  std::vector<unsigned char> h_lines((size_t)n_lines * 128, 0);
  std::vector<unsigned long long> h_packed((size_t)N);
  for (int64_t r = 0; r < N; ++r) {
    h_packed[(size_t)r] = ((unsigned long long)(r % 1000) << 32) | (unsigned long long)(r % 77 + 1);
    const int64_t line = r / 3; const int slot = (int)(r % 3);
    for (int b = 0; b < 32; ++b) h_lines[(size_t)line * 128 + slot * 32 + b] = h_rec[(size_t)r * 32 + b];
    reinterpret_cast<unsigned long long*>(&h_lines[(size_t)line * 128])[12 + slot] = h_packed[(size_t)r];
  }
  std::vector<unsigned char> h_lines2((size_t)n_lines * 128, 0);
  for (int64_t r = 0; r < N; ++r) {
    const int64_t line = r / 3; const int slot = (int)(r % 3);
    const int oa = slot == 0 ? 0 : (slot == 1 ? 48 : 64), ob = slot == 0 ? 16 : (slot == 1 ? 112 : 80), op = slot == 0 ? 32 : (slot == 1 ? 104 : 96);
    for (int b = 0; b < 16; ++b) {
      h_lines2[(size_t)line * 128 + oa + b] = h_rec[(size_t)r * 32 + b];
      h_lines2[(size_t)line * 128 + ob + b] = h_rec[(size_t)r * 32 + 16 + b];
    }
    *reinterpret_cast<unsigned long long*>(&h_lines2[(size_t)line * 128 + op]) = h_packed[(size_t)r];
  }
  void* d_lines2;
  CHECK(hipMalloc(&d_lines2, h_lines2.size()));
  CHECK(hipMemcpy(d_lines2, h_lines2.data(), h_lines2.size(), hipMemcpyHostToDevice));
  void *d_rec, *d_lines; unsigned long long *d_packed, *d_out; int* d_rows;
  CHECK(hipMalloc(&d_rec, h_rec.size())); CHECK(hipMalloc(&d_lines, h_lines.size()));
  CHECK(hipMalloc(&d_packed, h_packed.size() * 8)); CHECK(hipMalloc(&d_out, 64)); CHECK(hipMalloc(&d_rows, N * 4));
  CHECK(hipMemcpy(d_rec, h_rec.data(), h_rec.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_lines, h_lines.data(), h_lines.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_packed, h_packed.data(), h_packed.size() * 8, hipMemcpyHostToDevice));
  printf("%8s %10s | %10s %10s %10s %10s %10s %10s  (us)\n", "density", "rows", "sep/load", "lines/load", "rec32/load", "sep/full", "lines/full", "rec32/full");
  const int dens[] = {1, 2, 3, 4, 8, 16, 64};
  for (int d : dens) {
    std::vector<int> rows;
    uint64_t t = 0x9E3779B97F4A7C15ull;
    for (int64_t r = 0; r < N; ++r) { t = t * 6364136223846793005ull + 1442695040888963407ull; if ((t >> 33) % (uint64_t)d == 0) rows.push_back((int)r); }
    const int n = (int)rows.size();
    CHECK(hipMemcpy(d_rows, rows.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    int chunk = (n + 1023) / 1024; chunk = ((chunk + 511) / 512) * 512; if (chunk < 1024) chunk = 1024; if (chunk > 16384) chunk = 16384;
    double a = run<0, false>(d_rows, n, chunk, d_rec, d_packed, d_out);
    double b = run<1, false>(d_rows, n, chunk, d_lines, d_packed, d_out);
    double b2 = run<2, false>(d_rows, n, chunk, d_rec, d_packed, d_out);
    double c = run<0, true>(d_rows, n, chunk, d_rec, d_packed, d_out);
    double e = run<1, true>(d_rows, n, chunk, d_lines, d_packed, d_out);
    double e2 = run<2, true>(d_rows, n, chunk, d_rec, d_packed, d_out);
    double b3 = run<3, false>(d_rows, n, chunk, d_lines2, d_packed, d_out);
    double e3 = run<3, true>(d_rows, n, chunk, d_lines2, d_packed, d_out);
    printf("  1/%-5d %10d | %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f   rec32/full: %.1f ps/row | halves/load %.1f halves/full %.1f\n", d, n, a, b, b2, c, e, e2, e2 * 1e6 / n, b3, e3);
  }
  return 0;
}

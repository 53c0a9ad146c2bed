// Single-kernel entry points: the device gather used while sampling rows for
// bin finding, and thin drivers that run ONE training kernel on caller-provided
// device arrays so each kernel can be checked against a host oracle in isolation.
#include "hip_engine.hpp"
#include "kernels.hpp"
#include <stdexcept>
#include <string>

namespace lga {

namespace {
__global__ __launch_bounds__(256) void gather_rows_bytes_kernel(const char* __restrict__ src,
                                                                const int64_t* __restrict__ idx, int64_t n,
                                                                int64_t row_bytes, char* __restrict__ dst) {
  const int64_t total = n * row_bytes;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / row_bytes, b = i % row_bytes;
    dst[i] = src[idx[r] * row_bytes + b];
  }
}
}  // namespace

namespace {
int run_op(const std::string& name, void** p, int nptr, const int64_t* a, int nint, hipStream_t s);
}

int hip_run_op(const std::string& name, void** p, int nptr, const int64_t* a, int nint) {
  hipStream_t s = current_stream();
  const int rc = run_op(name, p, nptr, a, nint, s);
  if (rc == 0 && k::device_checks_enabled()) {
    const std::string failure = k::take_check_failure(s);
    if (!failure.empty()) throw std::runtime_error(failure);
  }
  return rc;
}

namespace {
int run_op(const std::string& name, void** p, int nptr, const int64_t* a, int nint, hipStream_t s) {
  if (name == "gather_rows_bytes") {
    // ptrs: src, idx(int64), dst ; ints: n, row_bytes
    if (nptr < 3 || nint < 2) return -1;
    int64_t n = a[0], rb = a[1];
    if (n <= 0) return 0;
    int64_t total = n * rb;
    int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(gather_rows_bytes_kernel, dim3(blocks), dim3(256), 0, s, (const char*)p[0], (const int64_t*)p[1],
                       n, rb, (char*)p[2]);
    LGA_HIP_CHECK(hipGetLastError());
    return 0;
  }
  if (name == "histogram") {
    // ptrs: tasks, chunk_task, nodes, a0, a1, bins_rt, packed, pool
    // ints: total_chunks, rows_per_chunk, N, slot_words, plane, F, tile_first, tile_step, tile_count
    if (nptr < 8 || nint < 9) return -1;
    k::InlineTasks none{};
    k::histogram((const SplitTask*)p[0], (const int32_t*)p[1], none, (int)a[0], (int)a[1], (const NodeRange*)p[2],
                 (const int32_t*)p[3], (const int32_t*)p[4], (const uint8_t*)p[5], a[2], (const int64_t*)p[6],
                 (int64_t*)p[7], a[3], a[4], (int)a[5], (int)a[6], (int)a[7], (int)a[8], nullptr, s);
    return 0;
  }
  if (name == "partition") {
    // ptrs: tasks, chunk_task, nodes, a0, a1, bins_fm, mask, chunk_left, task_left ; ints: total_chunks, N
    if (nptr < 9 || nint < 2) return -1;
    k::InlineTasks none{};
    k::partition((const SplitTask*)p[0], (const int32_t*)p[1], none, (int)a[0], a[0] > 1, (NodeRange*)p[2],
                 (const int32_t*)p[3], (const int32_t*)p[4], (int32_t*)p[3], (int32_t*)p[4], (const uint8_t*)p[5], a[1],
                 (uint64_t*)p[6], (int32_t*)p[7], (int32_t*)p[8], k::BagParams{}, nullptr, 0, s);
    return 0;
  }
  if (name == "scan") {
    // ptrs: scan tasks, meta, pool, node_totals, qs, split params (host), cands, out ; ints: n_tasks, F, slot_words, plane
    if (nptr < 8 || nint < 4) return -1;
    const SplitParams sp = *(const SplitParams*)p[5];
    k::scan((const k::ScanTaskDev*)p[0], (int)a[0], (const FeatureMeta*)p[1], (int)a[1], 0, 1, (int64_t*)p[2], a[2],
            a[3], (int64_t*)p[3], (const QuantScale*)p[4], sp, (k::Cand*)p[6], s);
    k::select((const k::ScanTaskDev*)p[0], (int)a[0], (int)a[1], (const k::Cand*)p[6], (const int64_t*)p[3],
              (const QuantScale*)p[4], sp, (SplitInfo*)p[7], nullptr, s);
    return 0;
  }
  return -1;
}
}  // namespace

}  // namespace lga

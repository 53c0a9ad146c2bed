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
    // ptrs: tasks, chunk_task, nodes, a0, a1, bins_rt, packed, pool, slot_table (2 words)
    //       [, wide qs: full-precision pass - `packed` holds float bit patterns (g high, h low), *qs their 36-bit exponents]
    // ints: total_chunks, rows_per_chunk, N, F, tile_first, tile_step, tile_count, tiles
    // pool: one slot of `tiles` x 32 feature blocks [g 256 | h 256] + header
    if (nptr < 9 || nint < 8) return -1;
    k::InlineTasks none{};
    k::PoolLayout L{};
    L.pool = (int64_t*)p[7];
    L.slot_table = (int64_t*)p[8];
    L.fpr = (int32_t)a[7] * kTileFeatures;
    L.ranks = 1;
    L.chunk_words = (int64_t)L.fpr * 2 * kMaxBins + kSlotHeader;
    L.slot_words = L.chunk_words;
    k::histogram((const SplitTask*)p[0], (const int32_t*)p[1], none, (int)a[0], (int)a[1], (const NodeRange*)p[2],
                 (const int32_t*)p[3], (const int32_t*)p[4], (const uint8_t*)p[5], a[2], (const int64_t*)p[6], L,
                 k::RoundSlots{0, 1}, (int)a[3], (int)a[4], (int)a[5], (int)a[6], nptr >= 10 ? (const QuantScale*)p[9] : nullptr, s);
    return 0;
  }
  if (name == "lines_histogram") {
    // Root pass of the fused growth kernel over row lines, histogram only (no split search).
    // ptrs: nodes (root range), rows, bins_rt (tile 0 records), packed, lines (scratch, row_line_bytes(N)), pool,
    //       slot_table, tickets (1 word, zero), g (float, for the fused quantisation) or null, h, maxbits, qs out
    // ints: n_rows, N, F, unit_rows [, wide: 1 = full-precision pass (packed = float bit patterns, qs = their 36-bit
    //       exponents: written by the fused quantisation, read otherwise)]
    if (nptr < 12 || nint < 4) return -1;
    const int64_t n = a[0], N = a[1];
    k::build_row_lines((const uint8_t*)p[2], N, (uint8_t*)p[4], s);
    k::InlineTasks inl{};
    inl.count = 1;
    SplitTask& t = inl.t[0];
    t.parent_node = -1;
    const int unit_rows = (int)a[3];
    t.num_hchunk = (int)std::max<int64_t>(1, (n + unit_rows - 1) / unit_rows);
    k::GrowArgs g{};
    g.nodes = (NodeRange*)p[0];
    g.a0 = (const int32_t*)p[1];
    g.a1 = (const int32_t*)p[1];
    g.lines = (uint8_t*)p[4];
    g.N = N;
    g.pool.pool = (int64_t*)p[5];
    g.pool.slot_table = (int64_t*)p[6];
    g.pool.fpr = kTileFeatures;
    g.pool.ranks = 1;
    g.pool.chunk_words = (int64_t)kTileFeatures * 2 * kMaxBins + kSlotHeader;
    g.pool.slot_words = g.pool.chunk_words;
    g.round = k::RoundSlots{0, 1};
    g.F = (int)a[2];
    g.unit_rows = unit_rows;
    g.wide = (nint >= 5 && a[4]) ? 1 : 0;
    g.qs = (const QuantScale*)p[11];
    if (p[8]) g.quant = k::FusedQuant{(const float*)p[8], (const float*)p[9], (const uint32_t*)p[10], (QuantScale*)p[11], RoundingSpec{}};
    else k::store_line_gradients((uint8_t*)p[4], (const int64_t*)p[3], N, s);
    k::grow_round(inl, g, t.num_hchunk, true, s);
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
  if (name == "grow_partition") {
    // Single-pass partition of ONE multi-block leaf (node 0 -> nodes 1, 2).
    // ptrs: task (one SplitTask), nodes, a0 (source rows), a1 (destination), bins_fm, cursor (>= 16 words, zero), task_left
    // ints: total_chunks, N, rows per workgroup (2048, 4096 or 8192)
    if (nptr < 7 || nint < 3) return -1;
    k::InlineTasks inl{};
    inl.count = 1;
    LGA_HIP_CHECK(hipMemcpyAsync(&inl.t[0], p[0], sizeof(SplitTask), hipMemcpyDeviceToHost, s));
    LGA_HIP_CHECK(hipStreamSynchronize(s));
    k::GrowArgs g{};
    g.nodes = (NodeRange*)p[1];
    g.a0 = (const int32_t*)p[2]; g.a1 = (const int32_t*)p[3];
    g.a0_dst = (int32_t*)p[2]; g.a1_dst = (int32_t*)p[3];
    g.bins_fm = (const uint8_t*)p[4];
    g.N = a[1];
    g.part_cursor = (unsigned long long*)p[5];
    g.part_block_rows = (int32_t)a[2];
    g.task_left = (int32_t*)p[6];
    k::grow_partition(inl, g, (int)a[0], s);
    return 0;
  }
  if (name == "scan") {
    // Split search of ONE leaf whose histogram sits in slot 0 of `pool` (feature blocks [g 256 | h 256] + header).
    // ptrs: meta, pool, slot_table (filled: {0, chunk_words}), node_totals, qs, split params (host), block_best,
    //       counters (>= 8 words, zero), scan_totals (4 words), out (device: 64-byte flag block + SplitInfo + QuantScale),
    //       results (device, one SplitInfo)
    // ints: F, features per slot (multiple of 32)
    if (nptr < 11 || nint < 2) return -1;
    const SplitParams sp = *(const SplitParams*)p[5];
    k::InlineTasks inl{};
    inl.count = 1;
    k::ScanArgs sa{};
    sa.n_scan = 1;
    sa.root = 1;
    sa.meta = (const FeatureMeta*)p[0];
    sa.F = (int)a[0];
    sa.pool.pool = (int64_t*)p[1];
    sa.pool.slot_table = (int64_t*)p[2];
    sa.pool.fpr = (int32_t)a[1];
    sa.pool.ranks = 1;
    sa.pool.chunk_words = (int64_t)sa.pool.fpr * 2 * kMaxBins + kSlotHeader;
    sa.pool.slot_words = sa.pool.chunk_words;
    sa.round = k::RoundSlots{0, 1};
    sa.range = k::ScanRange{0, (int)a[0], 64, 0, 1, std::max(1, ((int)a[0] + 63) / 64)};
    sa.node_totals = (int64_t*)p[3];
    sa.scan_totals = (int64_t*)p[8];
    sa.qs = (const QuantScale*)p[4];
    sa.sp = sp;
    sa.block_best = (k::Cand*)p[6];
    sa.tickets = (unsigned*)p[7] + 4;
    sa.done_counter = (unsigned*)p[7];
    sa.host_out = (char*)p[9];
    sa.results = (SplitInfo*)p[10];
    sa.seq = 1;
    k::scan_blocks(inl, sa, 1, s);
    return 0;
  }
  return -1;
}
}  // namespace

}  // namespace lga

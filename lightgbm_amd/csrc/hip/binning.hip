// Device binning: raw values -> uint8 bins in both layouts the learners read
// (row-tile-major 32-byte records and feature-major columns), plus the
// management of a dataset's device mirror.
//
// bin_apply: one workgroup = 256 rows x one 32-feature tile.  The tile's
// upper bounds sit in LDS (32 x 256 doubles = 64 KiB); a lane handles one
// (row, feature) value per step with a binary search in LDS; the bins of the
// block are staged in LDS so that both output layouts are written coalesced.
#include "hip_engine.hpp"
#include "../binning_core.hpp"
#include <cstring>

namespace lga {

namespace {

constexpr int kBinRows = 256;

template <typename T>
__global__ __launch_bounds__(256) void bin_apply_kernel(const T* __restrict__ raw, int64_t nrow, int ncol,
                                                        int64_t row_offset, int64_t total_rows,
                                                        const double* __restrict__ upper,      // [F][256]
                                                        const int32_t* __restrict__ nvalue,    // [F] value bins
                                                        const int32_t* __restrict__ missing,   // [F] NaN bin or -1
                                                        const int32_t* __restrict__ real_of,   // [F] real column
                                                        int F, uint8_t* __restrict__ bins_rt,
                                                        uint8_t* __restrict__ bins_fm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* ub = reinterpret_cast<double*>(smem);                                 // [32][256]
  uint8_t* tile_bins = reinterpret_cast<uint8_t*>(smem + sizeof(double) * kTileFeatures * kMaxBins);  // [32][256 rows]
  const int tile = blockIdx.y;
  const int f0 = tile * kTileFeatures;
  const int nf = min(kTileFeatures, F - f0);
  for (int i = threadIdx.x; i < nf * kMaxBins; i += 256) ub[i] = upper[(int64_t)f0 * kMaxBins + i];
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * kBinRows;
  const int rows_here = (int)min((int64_t)kBinRows, nrow - r0);
  // lane -> feature (fast index), so a wave reads 64 consecutive values of 2 rows when F is small
  for (int idx = threadIdx.x; idx < rows_here * kTileFeatures; idx += 256) {
    const int rr = idx / kTileFeatures, k = idx % kTileFeatures;
    uint8_t b = 0;
    if (k < nf) {
      const int f = f0 + k;
      double v = (double)raw[(r0 + rr) * (int64_t)ncol + real_of[f]];
      if (v != v) {
        if (missing[f] >= 0) b = (uint8_t)missing[f];
        else b = (uint8_t)search_bin(ub + k * kMaxBins, nvalue[f], 0.0);
      } else {
        b = (uint8_t)search_bin(ub + k * kMaxBins, nvalue[f], v);
      }
    }
    tile_bins[k * kBinRows + rr] = b;
  }
  __syncthreads();
  // row-tile-major: 32 bytes per row, rows consecutive -> one dword per lane, fully coalesced
  uint32_t* rt = reinterpret_cast<uint32_t*>(bins_rt + ((int64_t)tile * total_rows + row_offset + r0) * kTileFeatures);
  for (int idx = threadIdx.x; idx < rows_here * (kTileFeatures / 4); idx += 256) {
    const int rr = idx / (kTileFeatures / 4), q = idx % (kTileFeatures / 4);
    uint32_t w = (uint32_t)tile_bins[(q * 4 + 0) * kBinRows + rr] | ((uint32_t)tile_bins[(q * 4 + 1) * kBinRows + rr] << 8) |
                 ((uint32_t)tile_bins[(q * 4 + 2) * kBinRows + rr] << 16) | ((uint32_t)tile_bins[(q * 4 + 3) * kBinRows + rr] << 24);
    rt[idx] = w;
  }
  // feature-major: per feature 256 consecutive bytes
  for (int idx = threadIdx.x; idx < nf * kBinRows; idx += 256) {
    const int k = idx / kBinRows, rr = idx % kBinRows;
    if (rr < rows_here) bins_fm[(int64_t)(f0 + k) * total_rows + row_offset + r0 + rr] = tile_bins[k * kBinRows + rr];
  }
}

struct MapperTables {
  DeviceBuffer<double> upper;
  DeviceBuffer<int32_t> nvalue, missing, real_of;
};

void upload_mapper_tables(const Dataset* root, MapperTables* t, hipStream_t s) {
  int F = root->num_features();
  std::vector<double> up((size_t)std::max(F, 1) * kMaxBins, std::numeric_limits<double>::infinity());
  std::vector<int32_t> nv((size_t)std::max(F, 1), 1), ms((size_t)std::max(F, 1), -1), ro((size_t)std::max(F, 1), 0);
  for (int f = 0; f < F; ++f) {
    const BinMapper& m = root->inner_mapper(f);
    std::memcpy(&up[(size_t)f * kMaxBins], m.upper.data(), sizeof(double) * m.upper.size());
    nv[(size_t)f] = (int32_t)m.upper.size();
    ms[(size_t)f] = m.missing_bin;
    ro[(size_t)f] = root->real_feature(f);
  }
  t->upper.upload(up.data(), up.size(), s);
  t->nvalue.upload(nv.data(), nv.size(), s);
  t->missing.upload(ms.data(), ms.size(), s);
  t->real_of.upload(ro.data(), ro.size(), s);
  LGA_HIP_CHECK(hipStreamSynchronize(s));   // the host vectors go out of scope
}

DeviceMirror* new_mirror(const Dataset* root) {
  auto m = std::make_shared<DeviceMirror>();
  m->rows = root->root_rows();
  m->features = root->num_features();
  m->tiles = root->num_tiles();
  m->bins_rt.resize((size_t)std::max(m->tiles, 1) * (size_t)m->rows * kTileFeatures);
  m->bins_fm.resize((size_t)std::max(m->features, 1) * (size_t)m->rows);
  root->device = m;
  return m.get();
}

template <typename T>
void launch_bin_apply(const T* dev_raw, int64_t nrow, int ncol, int64_t row_offset, const Dataset* root,
                      DeviceMirror* m, const MapperTables& t, hipStream_t s) {
  if (nrow <= 0 || m->features == 0) return;
  size_t smem = sizeof(double) * kTileFeatures * kMaxBins + (size_t)kTileFeatures * kBinRows;
  dim3 grid((unsigned)((nrow + kBinRows - 1) / kBinRows), (unsigned)m->tiles);
  static bool attr_set = false;   // one flag per template instantiation
  if (!attr_set) {
    LGA_HIP_CHECK(hipFuncSetAttribute((const void*)bin_apply_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_set = true;
  }
  hipLaunchKernelGGL(bin_apply_kernel<T>, grid, dim3(256), smem, s, dev_raw, nrow, ncol, row_offset, m->rows,
                     t.upper.get(), t.nvalue.get(), t.missing.get(), t.real_of.get(), m->features, m->bins_rt.get(),
                     m->bins_fm.get());
  LGA_HIP_CHECK(hipGetLastError());
}

}  // namespace

DeviceMirror* ensure_device_mirror(const Dataset* ds) {
  const Dataset* root = ds->root();
  if (root->device) return root->device.get();
  if (!root->has_host_bins()) throw std::runtime_error("dataset has neither host nor device bins");
  DeviceMirror* m = new_mirror(root);
  hipStream_t s = current_stream();
  size_t rt_bytes = (size_t)m->tiles * (size_t)m->rows * kTileFeatures, fm_bytes = (size_t)m->features * (size_t)m->rows;
  if (rt_bytes) LGA_HIP_CHECK(hipMemcpyAsync(m->bins_rt.get(), root->bins_rt(), rt_bytes, hipMemcpyHostToDevice, s));
  if (fm_bytes) LGA_HIP_CHECK(hipMemcpyAsync(m->bins_fm.get(), root->bins_fm(), fm_bytes, hipMemcpyHostToDevice, s));
  LGA_HIP_CHECK(hipStreamSynchronize(s));
  return m;
}

void download_bins_to_host(Dataset* ds) {
  Dataset* root = ds->root_mut();
  if (root->has_host_bins()) return;
  if (!root->device) throw std::runtime_error("dataset has no bins at all");
  DeviceMirror* m = root->device.get();
  hipStream_t s = current_stream();
  std::vector<uint8_t> rt((size_t)m->tiles * (size_t)m->rows * kTileFeatures), fm((size_t)m->features * (size_t)m->rows);
  if (!rt.empty()) LGA_HIP_CHECK(hipMemcpyAsync(rt.data(), m->bins_rt.get(), rt.size(), hipMemcpyDeviceToHost, s));
  if (!fm.empty()) LGA_HIP_CHECK(hipMemcpyAsync(fm.data(), m->bins_fm.get(), fm.size(), hipMemcpyDeviceToHost, s));
  LGA_HIP_CHECK(hipStreamSynchronize(s));
  root->adopt_host_bins(std::move(rt), std::move(fm));
}

void bin_device_matrix(const Dataset* ds, const void* dev_data, int dtype, int64_t nrow, int ncol) {
  const Dataset* root = ds->root();
  if (nrow != root->root_rows()) throw std::runtime_error("bin_device_matrix: row count mismatch");
  DeviceMirror* m = root->device ? root->device.get() : new_mirror(root);
  hipStream_t s = current_stream();
  MapperTables t;
  upload_mapper_tables(root, &t, s);
  if (dtype == 0) launch_bin_apply((const float*)dev_data, nrow, ncol, 0, root, m, t, s);
  else launch_bin_apply((const double*)dev_data, nrow, ncol, 0, root, m, t, s);
  LGA_HIP_CHECK(hipStreamSynchronize(s));
}

void bin_host_matrix_on_device(const Dataset* ds, const void* data, int dtype, int64_t nrow, int ncol, bool row_major) {
  const Dataset* root = ds->root();
  if (nrow != root->root_rows()) throw std::runtime_error("bin_host_matrix_on_device: row count mismatch");
  DeviceMirror* m = root->device ? root->device.get() : new_mirror(root);
  hipStream_t s = current_stream();
  MapperTables t;
  upload_mapper_tables(root, &t, s);
  const size_t esz = dtype == 0 ? 4 : 8;
  // stream the matrix through a bounded staging buffer (<= 256 MiB)
  int64_t chunk = std::max<int64_t>(1, (int64_t)((256u << 20) / (esz * (size_t)ncol)));
  chunk = std::min(chunk, nrow);
  DeviceBuffer<char> stage((size_t)chunk * (size_t)ncol * esz);
  std::vector<char> rm_buf;
  for (int64_t r0 = 0; r0 < nrow; r0 += chunk) {
    int64_t n = std::min(chunk, nrow - r0);
    const char* src;
    if (row_major) {
      src = (const char*)data + (size_t)r0 * (size_t)ncol * esz;
    } else {
      // column-major input: transpose the block on the host
      rm_buf.resize((size_t)n * (size_t)ncol * esz);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i)
        for (int c = 0; c < ncol; ++c)
          std::memcpy(&rm_buf[((size_t)i * ncol + c) * esz], (const char*)data + ((size_t)c * nrow + (r0 + i)) * esz, esz);
      src = rm_buf.data();
    }
    LGA_HIP_CHECK(hipMemcpyAsync(stage.get(), src, (size_t)n * (size_t)ncol * esz, hipMemcpyHostToDevice, s));
    if (dtype == 0) launch_bin_apply((const float*)stage.get(), n, ncol, r0, root, m, t, s);
    else launch_bin_apply((const double*)stage.get(), n, ncol, r0, root, m, t, s);
    LGA_HIP_CHECK(hipStreamSynchronize(s));
  }
}

}  // namespace lga

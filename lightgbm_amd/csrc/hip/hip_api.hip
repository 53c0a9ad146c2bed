// C ABI additions of the HIP build: allocator / stream hooks, device-side
// dataset construction, device prediction, the RCCL communicator, and a small
// "op" entry point that runs single kernels for the kernel-level tests.
#include "../c_api.hpp"
#include "../gbdt.hpp"
#include "hip_engine.hpp"
#include "kernels.hpp"
#include "rccl_comm.hpp"
#include <cstring>
#include <string>
#include <vector>

using namespace lga;

namespace lga {
void hip_predict(const Booster& b, const void* data, bool on_device, int dtype, int64_t nrow, int ncol, int start_iter,
                 int num_iter, int type, double* out, bool out_on_device);
int hip_run_op(const std::string& name, void** ptrs, int nptr, const int64_t* ints, int nint);
}

namespace {
template <typename F>
int hguard(F f) {
  try { f(); return 0; }
  catch (const std::exception& e) { lga_set_last_error(e.what()); return -1; }
  catch (...) { lga_set_last_error("unknown error"); return -1; }
}
struct DatasetBox { std::shared_ptr<Dataset> ds; };
struct BoosterBox { std::unique_ptr<Booster> b; std::string scratch; };
}  // namespace

extern "C" {

int lga_hip_set_allocator(DeviceAllocFn a, DeviceFreeFn f) { return hguard([&] { set_device_allocator(a, f); }); }
int lga_hip_set_stream(void* stream) { return hguard([&] { set_current_stream((hipStream_t)stream); }); }

int lga_hip_device_info(int* cus, int64_t* mem_bytes, int* lds_bytes) {
  return hguard([&] {
    int dev = 0;
    LGA_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t p;
    LGA_HIP_CHECK(hipGetDeviceProperties(&p, dev));
    *cus = p.multiProcessorCount;
    *mem_bytes = (int64_t)p.totalGlobalMem;
    *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
  });
}

// Raw matrix already in device memory (row-major): sample, find bin boundaries and bin on the device
// (device_bin_find=false moves the boundary search to the host; both give the same boundaries).
int lga_hip_dataset_from_device_mat(const void* dev_data, int dtype, int64_t nrow, int ncol, const char* params,
                                    LgaDataset reference, LgaDataset* out) {
  return hguard([&] {
    Config cfg;
    cfg.set(std::string(params ? params : ""));
    hipStream_t s = current_stream();
    std::shared_ptr<Dataset> ds;
    if (reference) {
      // mappers from the reference: build a mapper-only dataset through a tiny host sample
      const Dataset* ref = ((DatasetBox*)reference)->ds->root();
      if (ref->num_total_features() != ncol) throw std::runtime_error("reference has a different number of features");
      ds = Dataset::from_reference(ref, nrow);
    } else {
      std::vector<int64_t> rows = Dataset::sample_rows(nrow, cfg);
      int64_t ns = (int64_t)rows.size();
      const size_t esz = dtype == 0 ? 4 : 8;
      std::vector<char> host_rows(cfg.device_bin_find ? 0 : (size_t)ns * (size_t)ncol * esz);
      // contiguous runs are rare in a random sample: copy row by row in batches through a strided 2-D copy
      // (hipMemcpy2D cannot gather), so gather on the device first
      DeviceBuffer<int64_t> d_idx((size_t)std::max<int64_t>(ns, 1));
      DeviceBuffer<char> d_rows((size_t)std::max<int64_t>(ns, 1) * (size_t)ncol * esz);
      d_idx.upload(rows.data(), (size_t)ns, s);
      void* ptrs[3] = {(void*)dev_data, d_idx.get(), d_rows.get()};
      int64_t ints[3] = {ns, (int64_t)ncol * (int64_t)esz, 0};
      if (hip_run_op("gather_rows_bytes", ptrs, 3, ints, 3) != 0) throw std::runtime_error("gather_rows_bytes failed");
      if (cfg.device_bin_find) {
        ds = Dataset::from_mappers(find_bins_on_device(d_rows.get(), dtype, ns, ncol, cfg), nrow);
      } else {
        LGA_HIP_CHECK(hipMemcpyAsync(host_rows.data(), d_rows.get(), host_rows.size(), hipMemcpyDeviceToHost, s));
        LGA_HIP_CHECK(hipStreamSynchronize(s));
        std::vector<double> sample((size_t)ns * (size_t)ncol);
        for (size_t i = 0; i < sample.size(); ++i)
          sample[i] = dtype == 0 ? (double)((const float*)host_rows.data())[i] : ((const double*)host_rows.data())[i];
        ds = Dataset::from_sample(sample.data(), ns, ncol, nrow, cfg);
      }
    }
    bin_device_matrix(ds.get(), dev_data, dtype, nrow, ncol);
    *out = new DatasetBox{ds};
  });
}

// Raw matrix in host memory, dataset on the device: the sampled rows go up first and the bin boundaries are
// found there (sort + run-length + cut per feature), then the matrix streams through the binning kernel.
int lga_hip_dataset_from_host_mat(const void* data, int dtype, int64_t nrow, int ncol, int row_major, const char* params,
                                  LgaDataset* out) {
  return hguard([&] {
    if (nrow <= 0 || ncol <= 0) throw std::runtime_error("Dataset: empty matrix");
    if (nrow > 2147483000LL) throw std::runtime_error("Dataset: more than 2^31 rows per device is not supported");
    Config cfg;
    cfg.set(std::string(params ? params : ""));
    hipStream_t s = current_stream();
    std::vector<int64_t> rows = Dataset::sample_rows(nrow, cfg);
    const int64_t ns = (int64_t)rows.size();
    const size_t esz = dtype == 0 ? 4 : 8;
    std::vector<char> sample((size_t)ns * (size_t)ncol * esz);
    const char* src = (const char*)data;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < ns; ++i) {
      const int64_t r = rows[(size_t)i];
      if (row_major) {
        std::memcpy(&sample[(size_t)i * (size_t)ncol * esz], src + (size_t)r * (size_t)ncol * esz, (size_t)ncol * esz);
      } else {
        for (int c = 0; c < ncol; ++c)
          std::memcpy(&sample[((size_t)i * (size_t)ncol + (size_t)c) * esz], src + ((size_t)c * (size_t)nrow + (size_t)r) * esz, esz);
      }
    }
    DeviceBuffer<char> d_sample(sample.size());
    LGA_HIP_CHECK(hipMemcpyAsync(d_sample.get(), sample.data(), sample.size(), hipMemcpyHostToDevice, s));
    LGA_HIP_CHECK(hipStreamSynchronize(s));
    std::shared_ptr<Dataset> ds = Dataset::from_mappers(find_bins_on_device(d_sample.get(), dtype, ns, ncol, cfg), nrow);
    d_sample.release();
    bin_host_matrix_on_device(ds.get(), data, dtype, nrow, ncol, row_major != 0);
    *out = new DatasetBox{ds};
  });
}

// Bins a host matrix on the device into the (mapper-only) dataset created by lga_dataset_from_mat(bin_on_host=0).
int lga_hip_dataset_upload_raw(LgaDataset h, const void* data, int dtype, int64_t nrow, int ncol, int row_major) {
  return hguard([&] {
    Dataset* ds = ((DatasetBox*)h)->ds.get();
    bin_host_matrix_on_device(ds, data, dtype, nrow, ncol, row_major != 0);
  });
}

// Bins a raw row-major DEVICE matrix into a dataset that only holds bin mappers so far.
int lga_hip_dataset_bin_device(LgaDataset h, const void* dev_data, int dtype, int64_t nrow, int ncol) {
  return hguard([&] {
    Dataset* ds = ((DatasetBox*)h)->ds.get();
    if (ncol != ds->num_total_features()) throw std::runtime_error("column count differs from the dataset's");
    bin_device_matrix(ds, dev_data, dtype, nrow, ncol);
  });
}

int lga_hip_predict(LgaBooster b, const void* data, int on_device, int dtype, int64_t nrow, int ncol, int start_iter,
                    int num_iter, int type, double* out, int out_on_device) {
  return hguard([&] {
    hip_predict(*((BoosterBox*)b)->b, data, on_device != 0, dtype, nrow, ncol, start_iter, num_iter, type, out,
                out_on_device != 0);
  });
}

int lga_hip_rccl_unique_id(void* out, int cap, int* len) { return hguard([&] { rccl_unique_id(out, cap, len); }); }
int lga_hip_comm_create_rccl(int rank, int size, const void* unique_id, int id_len, LgaComm* out) {
  return hguard([&] { *out = make_rccl_comm(rank, size, unique_id, id_len); });
}

int lga_hip_comm_create_loopback_group(int size, LgaComm* out) {
  return hguard([&] {
    std::vector<Comm*> v((size_t)size);
    make_loopback_group(size, v.data());
    for (int r = 0; r < size; ++r) out[r] = v[(size_t)r];
  });
}

int lga_hip_op(const char* name, void** ptrs, int nptr, const int64_t* ints, int nint) {
  return hguard([&] {
    int r = hip_run_op(name, ptrs, nptr, ints, nint);
    if (r != 0) throw std::runtime_error(std::string("unknown or failed op ") + name);
  });
}

}  // extern "C"

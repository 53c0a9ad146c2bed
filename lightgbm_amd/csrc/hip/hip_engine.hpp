// Device (MI355X) execution engine.
#pragma once
#include "../engine.hpp"
#include "../host_engine.hpp"
#include "device.hpp"

namespace lga {

std::unique_ptr<Engine> make_hip_engine(std::shared_ptr<Dataset> train, const Config& cfg, Comm* comm,
                                        ParallelMode mode);

// Creates / returns the device mirror of a root dataset (uploads host bins when the
// bins were built on the host).
DeviceMirror* ensure_device_mirror(const Dataset* root);
// Copies device bins back into the dataset's host matrices when those are missing.
void download_bins_to_host(Dataset* ds);
// Bins a raw matrix that already lives in device memory (row-major) into `ds`' mirror.
void bin_device_matrix(const Dataset* ds, const void* dev_data, int dtype, int64_t nrow, int ncol);
// Bin boundaries of every column of a row-major DEVICE sample [ns][ncol] (bin_find.hip); equal to
// BinMapper::find_bins on the same sample.
// Exact unweighted AUC of device-resident scores (auc.hip).
double auc_on_device(const double* scores, const float* label, const int32_t* rows, int64_t n, hipStream_t s);
std::vector<BinMapper> find_bins_on_device(const void* dev_sample, int dtype, int64_t ns, int ncol, const Config& cfg);
// Streams a host matrix through the device binning kernel into `ds`' mirror.
void bin_host_matrix_on_device(const Dataset* ds, const void* data, int dtype, int64_t nrow, int ncol, bool row_major);

}  // namespace lga

// Chooses the execution engine for a booster.  The host build only knows the
// OpenMP engine; the HIP build adds the device engine for device_type=gpu.
#pragma once
#include "engine.hpp"
#include "host_engine.hpp"

namespace lga {
std::unique_ptr<Engine> make_engine(std::shared_ptr<Dataset> train, const Config& cfg, Comm* comm);
const char* engine_build_name();
// Makes sure the dataset's host bin matrices exist (copies them back from the device if needed).
void engine_ensure_host_bins(Dataset* ds);
}  // namespace lga

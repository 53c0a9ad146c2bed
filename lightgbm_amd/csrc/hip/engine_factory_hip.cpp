#include "../engine_factory.hpp"
#include "hip_engine.hpp"
#include <stdexcept>

namespace lga {

std::unique_ptr<Engine> make_engine(std::shared_ptr<Dataset> train, const Config& cfg, Comm* comm) {
  ParallelMode mode = cfg.tree_learner == "feature" ? ParallelMode::Feature
                      : cfg.tree_learner == "data"  ? ParallelMode::Data
                      : cfg.tree_learner == "voting" ? ParallelMode::Voting
                                                    : ParallelMode::Serial;
  if (cfg.device_type == "gpu") return make_hip_engine(train, cfg, comm, mode);
  engine_ensure_host_bins(train.get());
  return make_host_engine(train, cfg, comm, mode);
}

const char* engine_build_name() { return "hip"; }

void engine_ensure_host_bins(Dataset* ds) {
  if (!ds->has_host_bins()) download_bins_to_host(ds);
}

}  // namespace lga

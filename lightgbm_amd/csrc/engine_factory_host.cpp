#include "engine_factory.hpp"
#include <stdexcept>

namespace lga {

std::unique_ptr<Engine> make_engine(std::shared_ptr<Dataset> train, const Config& cfg, Comm* comm) {
  if (cfg.device_type == "gpu")
    throw std::runtime_error("device_type=gpu needs the HIP build of the native core (no GPU library is loaded)");
  ParallelMode mode = cfg.tree_learner == "feature" ? ParallelMode::Feature
                      : cfg.tree_learner == "data"  ? ParallelMode::Data
                      : cfg.tree_learner == "voting" ? ParallelMode::Voting
                                                    : ParallelMode::Serial;
  return make_host_engine(train, cfg, comm, mode);
}

const char* engine_build_name() { return "host"; }
void engine_ensure_host_bins(Dataset*) {}

}  // namespace lga

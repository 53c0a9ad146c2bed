// CPU (OpenMP) engine: the path used when no GPU is requested and the oracle
// the HIP kernels are checked against.  It implements the same batched-round
// backend contract and the same integer histogram arithmetic as the device,
// so both grow identical trees.
#pragma once
#include "engine.hpp"
#include <memory>

namespace lga {

// Voting: rows sharded as in Data; only the histograms of the features the ranks vote for are summed (PV-Tree).
enum class ParallelMode { Serial, Data, Feature, Voting };

std::unique_ptr<Engine> make_host_engine(std::shared_ptr<Dataset> train, const Config& cfg, Comm* comm,
                                         ParallelMode mode);

// Quantisation helper shared with tests: packs (g,h) of the listed rows.
void quantize_rows_host(const float* g, const float* h, const int32_t* rows, int64_t n, QuantScale qs,
                        int64_t* packed, const RoundingSpec& rounding);

// How the 16-bit form of tree `tree_index`'s gradients is rounded (common.hpp: RoundingSpec), for both engines.
inline RoundingSpec rounding_spec(const Config& cfg, int64_t tree_index, int rank) {
  RoundingSpec r{};
  r.stochastic = cfg.stochastic_rounding ? 1 : 0;
  r.seed = mix64((uint64_t)(uint32_t)cfg.seed * 0x9e3779b97f4a7c15ull + 0x51edull);
  r.stream = (uint64_t)tree_index;
  r.salt = (uint64_t)rank << 40;
  r.ext_ids = nullptr;
  return r;
}

}  // namespace lga

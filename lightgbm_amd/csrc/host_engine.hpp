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
                        int64_t* packed);

}  // namespace lga

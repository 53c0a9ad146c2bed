// Learning to rank: the lambdarank and rank_xendcg objectives and the ndcg@k / map@k metrics, all over
// QUERIES - contiguous row ranges of a dataset given by Dataset::query_boundaries().
//
// Gradients of one query depend on the order of its documents' scores, so they are taken per query on the
// host (one OpenMP task per query) from the scores the engine hands out, and given back through the
// engine's caller-supplied-gradients entry; the trees are then grown by the engine (host or HIP) exactly as
// for any other objective.  A distributed run keeps every query on one rank: gradients need no
// communication, the metrics are sums over queries that the caller reduces.
//
// Summation order (so that the host code and the device kernels of csrc/hip/ranking.hip add the same doubles
// in the same order): a document's lambda / hessian is the sum of its pair terms by ascending RANK of the
// partner document; the normalisation total of a query is the sum, by ascending rank r, of document r's
// partial sum over its partners of higher rank number; rank_xendcg's per-query totals run in row order.
#pragma once
#include "common.hpp"
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace lga {

// Pairwise logistic loss weighted by the NDCG change of swapping the pair (Burges, "From RankNet to
// LambdaRank to LambdaMART").  hi is the more relevant document.  lambda is hi's gradient term (<= 0; lo
// receives -lambda), hessian the term both receive.
LGA_HD void rank_pair(double score_hi, double score_lo, double gain_hi, double gain_lo, double disc_hi, double disc_lo,
                      double inv_max_dcg, double sigmoid, bool normalise, double* lambda, double* hessian) {
  const double delta_score = score_hi - score_lo;
  double delta_ndcg = (gain_hi - gain_lo) * fabs(disc_hi - disc_lo) * inv_max_dcg;
  if (normalise) delta_ndcg /= 0.01 + fabs(delta_score);
  double p = 1.0 / (1.0 + exp(sigmoid * delta_score));   // probability of the wrong order
  *hessian = p * (1.0 - p) * (sigmoid * sigmoid * delta_ndcg);
  *lambda = p * (-sigmoid * delta_ndcg);
}

// Discount of a 0-based rank.
inline double rank_discount(int rank) { return 1.0 / std::log2(2.0 + (double)rank); }

struct RankingSpec {
  bool xendcg = false;                // false: lambdarank
  double sigmoid = 1.0;               // steepness of the pairwise logistic loss
  bool norm = true;                   // lambdarank_norm
  int truncation_level = 30;          // pairs need one member among the top `truncation_level` documents
  std::vector<double> label_gain;     // gain of an integer relevance label (default 2^l - 1)
  uint64_t seed = 5;                  // rank_xendcg draws one uniform number per document and iteration
};

// 2^l - 1 for l = 0..30.
std::vector<double> default_label_gain();

// Labels of ranking data are integers within the gain table; throws otherwise.
void check_rank_labels(const float* label, int64_t n, const std::vector<double>& label_gain);

// 1 / (best DCG over the top k documents) per query; 0 when every document of the query has gain 0.
std::vector<double> inverse_max_dcg(const float* label, const std::vector<int64_t>& bounds,
                                    const std::vector<double>& label_gain, int k);

// g, h: one value per row.  weight may be null.  iter only matters for rank_xendcg.
void ranking_gradients(const RankingSpec& spec, const double* score, const float* label, const float* weight,
                       const std::vector<int64_t>& bounds, const std::vector<double>& inv_max_dcg, uint64_t iter,
                       float* g, float* h);

// Sum over queries of (query weight * metric) for every cut-off in `at`, and the sum of query weights in
// sums[at.size()].  ndcg: a query without any gain counts as 1; map: a query without relevant documents counts
// as 1.  The query weight is the mean row weight of the query (1 without weights).
void ndcg_sums(const double* score, const float* label, const float* weight, const std::vector<int64_t>& bounds,
               const std::vector<double>& label_gain, const std::vector<int>& at, std::vector<double>* sums);
// The same sums from per-query values computed elsewhere (Engine::ranking_metric_values):
// values[q * stride + offset + k] is query q's metric at cut-off k.
void weighted_query_sums(const double* values, size_t stride, size_t offset, size_t n_at, const float* weight,
                         const std::vector<int64_t>& bounds, std::vector<double>* sums);
void map_sums(const double* score, const float* label, const float* weight, const std::vector<int64_t>& bounds,
              const std::vector<int>& at, std::vector<double>* sums);

}  // namespace lga

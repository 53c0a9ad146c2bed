// Host evaluation metrics.  `eval = "rmse"` (r/gridsearchCV.R:75) and the
// default l2 of the grid loop (r/gridsearchCV.R:108-115) are the two the
// reference exercises; auc / binary_logloss serve the binary configs.
#pragma once
#include "backend.hpp"
#include "engine.hpp"
#include "objective.hpp"
#include <string>
#include <vector>

namespace lga {

int32_t metric_kind_from_name(const std::string& canonical);
const char* metric_name(int32_t kind);
bool metric_higher_better(int32_t kind);

// scores/label/weight are root-indexed; rows lists the rows to evaluate
// (nullptr: 0..n-1).  weight may be null.  With comm != nullptr sums are
// reduced over ranks (AUC: exact over gathered (score,label,weight) triples is
// not attempted; each rank's rank-sum is combined through a global score grid).
std::vector<EvalResult> eval_metrics_host(const double* scores, const float* label, const float* weight,
                                          const int32_t* rows, int64_t n, const std::vector<int32_t>& metrics,
                                          const ObjectiveSpec& obj, Comm* comm, int64_t class_stride = 0);
// class_stride: multiclass metrics read class k's score of a row at scores[k * class_stride + row].

// Weighted average precision of (score, label>0) pairs: the mean over positives of the precision at their
// score threshold (tied scores share a threshold).  Also returns, through the pointers, the weight of the
// positives (what a mean over ranks is weighted with).  1 when a class is missing.
double average_precision_host(std::vector<std::pair<double, std::pair<float, float>>>& items, double* total_pos);

// Weighted AUC of (score, label>0) pairs, ties counted half.
double auc_host(std::vector<std::pair<double, std::pair<float, float>>>& items);

}  // namespace lga

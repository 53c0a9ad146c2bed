#include "metric.hpp"
#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace lga {

int32_t metric_kind_from_name(const std::string& n) {
  if (n == "l2") return kMetL2;
  if (n == "rmse") return kMetRmse;
  if (n == "l1") return kMetL1;
  if (n == "huber") return kMetHuber;
  if (n == "fair") return kMetFair;
  if (n == "poisson") return kMetPoisson;
  if (n == "binary_logloss") return kMetBinaryLogloss;
  if (n == "binary_error") return kMetBinaryError;
  if (n == "auc") return kMetAuc;
  if (n == "cross_entropy") return kMetCrossEntropy;
  if (n == "quantile") return kMetQuantile;
  if (n == "mape") return kMetMape;
  if (n == "gamma") return kMetGamma;
  if (n == "gamma_deviance") return kMetGammaDeviance;
  if (n == "tweedie") return kMetTweedie;
  if (n == "multi_logloss") return kMetMultiLogloss;
  if (n == "multi_error") return kMetMultiError;
  throw std::runtime_error("unknown metric " + n);
}

const char* metric_name(int32_t k) {
  switch (k) {
    case kMetL2: return "l2";
    case kMetRmse: return "rmse";
    case kMetL1: return "l1";
    case kMetHuber: return "huber";
    case kMetFair: return "fair";
    case kMetPoisson: return "poisson";
    case kMetBinaryLogloss: return "binary_logloss";
    case kMetBinaryError: return "binary_error";
    case kMetAuc: return "auc";
    case kMetCrossEntropy: return "cross_entropy";
    case kMetQuantile: return "quantile";
    case kMetMape: return "mape";
    case kMetGamma: return "gamma";
    case kMetGammaDeviance: return "gamma_deviance";
    case kMetTweedie: return "tweedie";
    case kMetMultiLogloss: return "multi_logloss";
    case kMetMultiError: return "multi_error";
  }
  return "?";
}

bool metric_higher_better(int32_t k) { return k == kMetAuc; }

double auc_host(std::vector<std::pair<double, std::pair<float, float>>>& items) {
  // sort by score descending; accumulate positives seen so far
  std::sort(items.begin(), items.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
  double sum_pos = 0.0, accum = 0.0, cur_pos = 0.0, cur_neg = 0.0;
  double total_pos = 0.0, total_neg = 0.0;
  size_t n = items.size();
  for (size_t i = 0; i < n; ++i) {
    double w = items[i].second.second;
    bool pos = items[i].second.first > 0.0f;
    if (i > 0 && items[i].first != items[i - 1].first) {
      accum += cur_neg * (sum_pos + cur_pos * 0.5);
      sum_pos += cur_pos;
      cur_pos = cur_neg = 0.0;
    }
    if (pos) { cur_pos += w; total_pos += w; } else { cur_neg += w; total_neg += w; }
  }
  accum += cur_neg * (sum_pos + cur_pos * 0.5);
  if (total_pos <= 0.0 || total_neg <= 0.0) return 1.0;
  return accum / (total_pos * total_neg);
}

double average_precision_host(std::vector<std::pair<double, std::pair<float, float>>>& items, double* total_pos) {
  std::sort(items.begin(), items.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
  double seen_pos = 0.0, seen_neg = 0.0, cur_pos = 0.0, cur_neg = 0.0, ap = 0.0;
  auto close_group = [&]() {
    seen_pos += cur_pos;
    seen_neg += cur_neg;
    if (cur_pos > 0.0) ap += cur_pos * seen_pos / (seen_pos + seen_neg);
    cur_pos = cur_neg = 0.0;
  };
  for (size_t i = 0; i < items.size(); ++i) {
    if (i > 0 && items[i].first != items[i - 1].first) close_group();
    const double w = items[i].second.second;
    if (items[i].second.first > 0.0f) cur_pos += w; else cur_neg += w;
  }
  close_group();
  if (total_pos) *total_pos = seen_pos;
  if (seen_pos <= 0.0 || seen_neg <= 0.0) return 1.0;
  return ap / seen_pos;
}

namespace {
// AUC over ranks: scores are mapped to a common 2^20-cell grid, per-cell
// positive / negative weights are summed over ranks, ties inside a cell count half.
double auc_binned(const double* scores, const float* label, const float* weight, const int32_t* rows, int64_t n,
                  Comm* comm) {
  float mm[2] = {-3.0e38f, -3.0e38f};   // max(score), max(-score)
  for (int64_t i = 0; i < n; ++i) {
    int64_t r = rows ? rows[i] : i;
    float s = (float)scores[r];
    mm[0] = std::max(mm[0], s);
    mm[1] = std::max(mm[1], -s);
  }
  comm->allreduce_max_f32(mm, 2, false, nullptr);
  double hi = mm[0], lo = -mm[1];
  const int cells = 1 << 20;
  double scale = hi > lo ? (cells - 1) / (hi - lo) : 0.0;
  std::vector<double> hist((size_t)cells * 2, 0.0);
  for (int64_t i = 0; i < n; ++i) {
    int64_t r = rows ? rows[i] : i;
    int c = (int)((scores[r] - lo) * scale);
    c = std::min(std::max(c, 0), cells - 1);
    double w = weight ? weight[r] : 1.0;
    hist[(size_t)c * 2 + (label[r] > 0.0f ? 1 : 0)] += w;
  }
  comm->allreduce_sum_f64(hist.data(), hist.size(), false, nullptr);
  double sum_pos = 0.0, accum = 0.0, tp = 0.0, tn = 0.0;
  for (int c = cells - 1; c >= 0; --c) {
    double neg = hist[(size_t)c * 2], pos = hist[(size_t)c * 2 + 1];
    accum += neg * (sum_pos + pos * 0.5);
    sum_pos += pos;
    tp += pos; tn += neg;
  }
  if (tp <= 0.0 || tn <= 0.0) return 1.0;
  return accum / (tp * tn);
}
}  // namespace

std::vector<EvalResult> eval_metrics_host(const double* scores, const float* label, const float* weight,
                                          const int32_t* rows, int64_t n, const std::vector<int32_t>& metrics,
                                          const ObjectiveSpec& obj, Comm* comm, int64_t class_stride) {
  std::vector<EvalResult> out;
  bool dist = comm && comm->size() > 1;
  for (int32_t m : metrics) {
    EvalResult r;
    r.name = metric_name(m);
    r.higher_better = metric_higher_better(m);
    if (m == kMetAuc) {
      if (dist) {
        r.value = auc_binned(scores, label, weight, rows, n, comm);
      } else {
        std::vector<std::pair<double, std::pair<float, float>>> items((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
          int64_t row = rows ? rows[i] : i;
          items[(size_t)i] = {scores[row], {label[row], weight ? weight[row] : 1.0f}};
        }
        r.value = auc_host(items);
      }
    } else {
      double sum = 0.0, sw = 0.0;
#pragma omp parallel for reduction(+ : sum, sw) schedule(static) if (n > 65536)
      for (int64_t i = 0; i < n; ++i) {
        int64_t row = rows ? rows[i] : i;
        double w = weight ? weight[row] : 1.0;
        if (m == kMetMultiLogloss || m == kMetMultiError) {
          sum += w * multi_point_loss(m, obj, scores, class_stride, row, label[row]);
        } else {
          double raw = scores[row];
          double pred = transform_score(obj.kind, obj.sigmoid, raw);
          sum += w * point_loss(m, pred, raw, label[row], obj);
        }
        sw += w;
      }
      if (dist) {
        double b[2] = {sum, sw};
        comm->allreduce_sum_f64(b, 2, false, nullptr);
        sum = b[0]; sw = b[1];
      }
      double v = sw > 0.0 ? sum / sw : 0.0;
      if (m == kMetRmse) v = std::sqrt(v);
      r.value = v;
    }
    out.push_back(r);
  }
  return out;
}

}  // namespace lga

// Objective (gradient / hessian) and pointwise metric formulas, shared by the
// host engine and the HIP kernels.  `objective = "regression"` is what every
// training call of the reference passes (r/gridsearchCV.R:60,74,112,137);
// `binary` serves the Higgs-shaped benchmark configs.
#pragma once
#include "common.hpp"

namespace lga {

enum ObjectiveKind : int32_t {
  kObjRegression = 0,
  kObjRegressionL1 = 1,
  kObjHuber = 2,
  kObjFair = 3,
  kObjPoisson = 4,
  kObjBinary = 5,
  kObjCrossEntropy = 6,
  kObjCustom = 7,
  kObjQuantile = 8,
  kObjMape = 9,
  kObjGamma = 10,
  kObjTweedie = 11,
  kObjMulticlass = 12,      // softmax over num_class score vectors
  kObjMulticlassOva = 13,   // one binary (sigmoid) model per class
  kObjLambdarank = 14,      // ranking objectives: gradients per query on the host (ranking.hpp)
  kObjXendcg = 15,
  kObjRegressionSqrt = 16,  // L2 on sign(y) sqrt|y| (reg_sqrt); predictions are squared back
};

enum MetricKind : int32_t {
  kMetL2 = 0,
  kMetRmse = 1,
  kMetL1 = 2,
  kMetHuber = 3,
  kMetFair = 4,
  kMetPoisson = 5,
  kMetBinaryLogloss = 6,
  kMetBinaryError = 7,
  kMetAuc = 8,
  kMetCrossEntropy = 9,
  kMetQuantile = 10,
  kMetMape = 11,
  kMetGamma = 12,
  kMetGammaDeviance = 13,
  kMetTweedie = 14,
  kMetMultiLogloss = 15,
  kMetMultiError = 16,
};

struct ObjectiveSpec {
  int32_t kind;
  int32_t num_class;         // 1 unless multiclass
  double sigmoid;            // binary
  double pos_weight;         // binary: weight multiplier of positive rows
  double neg_weight;
  double alpha;              // huber delta / quantile level
  double fair_c;
  double poisson_max_delta;  // poisson safety term
  double tweedie_rho;        // tweedie variance power, in [1, 2)
  int32_t multi_error_top_k; // multi_error: classes that may score at least as high as the true one (it included); < 1 reads as 1
  int32_t pad_;
};

// reg_sqrt: the label the model is fitted to (rounded to float like every label; upstream LightGBM:
// regression_objective.hpp, RegressionL2loss with sqrt_)
LGA_HD float sqrt_label(float y) {
  const double r = sqrt(fabs((double)y));
  return (float)(y < 0.0f ? -r : r);
}

// label is the raw label; for binary any label > 0 is positive.
LGA_HD void grad_hess(const ObjectiveSpec& o, double score, float label, float weight, float* g, float* h) {
  double gg, hh;
  switch (o.kind) {
    case kObjRegression: {
      gg = score - (double)label; hh = 1.0; break;
    }
    case kObjRegressionSqrt: {
      gg = score - (double)sqrt_label(label); hh = 1.0; break;
    }
    case kObjRegressionL1: {
      double d = score - (double)label;
      gg = d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0); hh = 1.0; break;
    }
    case kObjHuber: {
      double d = score - (double)label;
      if (fabs(d) <= o.alpha) gg = d; else gg = d > 0.0 ? o.alpha : -o.alpha;
      hh = 1.0; break;
    }
    case kObjFair: {
      double d = score - (double)label;
      double c = o.fair_c;
      gg = c * d / (fabs(d) + c);
      hh = c * c / ((fabs(d) + c) * (fabs(d) + c)); break;
    }
    case kObjPoisson: {
      double e = exp(score);
      gg = e - (double)label;
      hh = e * exp(o.poisson_max_delta); break;
    }
    case kObjBinary: {
      // y in {-1,+1}: g = -y*s/(1+exp(y*s*score)), h = |g|*(s-|g|)
      double y = label > 0.0f ? 1.0 : -1.0;
      double lw = label > 0.0f ? o.pos_weight : o.neg_weight;
      double resp = -y * o.sigmoid / (1.0 + exp(y * o.sigmoid * score));
      double ar = fabs(resp);
      gg = resp * lw; hh = ar * (o.sigmoid - ar) * lw; break;
    }
    case kObjCrossEntropy: {
      double p = 1.0 / (1.0 + exp(-score));
      gg = p - (double)label; hh = p * (1.0 - p); break;
    }
    case kObjQuantile: {
      double d = score - (double)label;
      gg = d >= 0.0 ? 1.0 - o.alpha : -o.alpha; hh = 1.0; break;
    }
    case kObjMape: {
      // absolute percentage error: the L1 gradient scaled by 1 / max(1, |label|)
      double d = score - (double)label;
      double lw = 1.0 / fmax(1.0, fabs((double)label));
      gg = (d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0)) * lw; hh = 1.0; break;
    }
    case kObjGamma: {
      double e = exp(-score);
      gg = 1.0 - (double)label * e; hh = (double)label * e; break;
    }
    case kObjTweedie: {
      double a = exp((1.0 - o.tweedie_rho) * score), b = exp((2.0 - o.tweedie_rho) * score);
      gg = -(double)label * a + b;
      hh = -(double)label * (1.0 - o.tweedie_rho) * a + (2.0 - o.tweedie_rho) * b; break;
    }
    default: gg = 0.0; hh = 1.0; break;
  }
  *g = (float)(gg * (double)weight);
  *h = (float)(hh * (double)weight);
}

LGA_HD bool is_multiclass(int32_t obj_kind) { return obj_kind == kObjMulticlass || obj_kind == kObjMulticlassOva; }

// Multiclass: class k's score of row r is scores[k * stride + r]; gradients are laid out the same way.
// Softmax: p = softmax(scores), g = p - [y == k], h = K / (K - 1) * p (1 - p).  One-vs-all: the binary
// formula per class with label [y == k].
LGA_HD void multiclass_grad_hess(const ObjectiveSpec& o, const double* scores, int64_t stride, int64_t r, float label,
                                 float weight, float* g, float* h) {
  const int K = o.num_class, y = (int)label;
  if (o.kind == kObjMulticlassOva) {
    for (int k = 0; k < K; ++k) {
      const double t = k == y ? 1.0 : -1.0;
      const double resp = -t * o.sigmoid / (1.0 + exp(t * o.sigmoid * scores[(int64_t)k * stride + r]));
      const double ar = fabs(resp);
      g[(int64_t)k * stride + r] = (float)(resp * (double)weight);
      h[(int64_t)k * stride + r] = (float)(ar * (o.sigmoid - ar) * (double)weight);
    }
    return;
  }
  double mx = scores[r];
  for (int k = 1; k < K; ++k) mx = fmax(mx, scores[(int64_t)k * stride + r]);
  double sum = 0.0;
  for (int k = 0; k < K; ++k) sum += exp(scores[(int64_t)k * stride + r] - mx);
  const double factor = (double)K / (double)(K - 1);
  for (int k = 0; k < K; ++k) {
    const double p = exp(scores[(int64_t)k * stride + r] - mx) / sum;
    g[(int64_t)k * stride + r] = (float)((p - (k == y ? 1.0 : 0.0)) * (double)weight);
    h[(int64_t)k * stride + r] = (float)(factor * p * (1.0 - p) * (double)weight);
  }
}

// multi_logloss: -log of the true class's probability (softmax, or that class's own sigmoid under
// one-vs-all); multi_error: 1 when multi_error_top_k classes besides the true one score at least as high as it.
LGA_HD double multi_point_loss(int32_t metric, const ObjectiveSpec& o, const double* scores, int64_t stride, int64_t r,
                               float label) {
  const int K = o.num_class, y = (int)label;
  const double sy = scores[(int64_t)y * stride + r];
  if (metric == kMetMultiError) {
    int at_least = 0;
    for (int k = 0; k < K; ++k) at_least += scores[(int64_t)k * stride + r] >= sy ? 1 : 0;
    return at_least > (o.multi_error_top_k > 1 ? o.multi_error_top_k : 1) ? 1.0 : 0.0;
  }
  double p;
  if (o.kind == kObjMulticlassOva) {
    p = 1.0 / (1.0 + exp(-o.sigmoid * sy));
  } else {
    double mx = sy;
    for (int k = 0; k < K; ++k) mx = fmax(mx, scores[(int64_t)k * stride + r]);
    double sum = 0.0;
    for (int k = 0; k < K; ++k) sum += exp(scores[(int64_t)k * stride + r] - mx);
    p = exp(sy - mx) / sum;
  }
  return -log(p > 1e-15 ? p : 1e-15);
}

// score -> prediction (link function)
LGA_HD double transform_score(int32_t obj_kind, double sigmoid, double score) {
  switch (obj_kind) {
    case kObjBinary: return 1.0 / (1.0 + exp(-sigmoid * score));
    case kObjCrossEntropy: return 1.0 / (1.0 + exp(-score));
    case kObjPoisson:
    case kObjGamma:
    case kObjTweedie: return exp(score);
    case kObjRegressionSqrt: return score < 0.0 ? -(score * score) : score * score;
    default: return score;
  }
}

// Pointwise loss of a metric (AUC is rank based and handled separately).
// `pred` is the transformed prediction, `raw` the raw score.
LGA_HD double point_loss(int32_t metric, double pred, double raw, float label, const ObjectiveSpec& o) {
  const double eps = 1e-15;
  switch (metric) {
    case kMetL2:
    case kMetRmse: { double d = pred - (double)label; return d * d; }
    case kMetL1: return fabs(pred - (double)label);
    case kMetHuber: {
      double d = fabs(pred - (double)label);
      return d <= o.alpha ? 0.5 * d * d : o.alpha * (d - 0.5 * o.alpha);
    }
    case kMetFair: {
      double x = fabs(pred - (double)label), c = o.fair_c;
      return c * x - c * c * log(1.0 + x / c);
    }
    case kMetPoisson: {
      double p = pred < 1e-10 ? 1e-10 : pred;
      return p - (double)label * log(p);
    }
    case kMetBinaryLogloss: {
      double p = pred;
      if (label > 0.0f) return -log(p > eps ? p : eps);
      return -log(1.0 - p > eps ? 1.0 - p : eps);
    }
    case kMetBinaryError: {
      bool pos = pred > 0.5;
      return (pos != (label > 0.0f)) ? 1.0 : 0.0;
    }
    case kMetCrossEntropy: {
      double p = pred, y = (double)label;
      double a = p > eps ? p : eps, b = 1.0 - p > eps ? 1.0 - p : eps;
      return -(y * log(a) + (1.0 - y) * log(b));
    }
    case kMetQuantile: {
      double d = (double)label - pred;
      return d >= 0.0 ? o.alpha * d : (o.alpha - 1.0) * d;
    }
    case kMetMape: return fabs((double)label - pred) / fmax(1.0, fabs((double)label));
    case kMetGamma: {
      // negative log-likelihood of a gamma distribution with unit shape: y / mu + log(mu)
      double p = pred < 1e-10 ? 1e-10 : pred;
      return (double)label / p + log(p);
    }
    case kMetGammaDeviance: {
      // 2 * (y / mu - log(y / mu) - 1), the factor applied here
      double t = (double)label / (pred + 1e-9);
      return 2.0 * (t - log(t) - 1.0);
    }
    case kMetTweedie: {
      double p = pred < 1e-10 ? 1e-10 : pred, rho = o.tweedie_rho;
      return -(double)label * exp((1.0 - rho) * log(p)) / (1.0 - rho) + exp((2.0 - rho) * log(p)) / (2.0 - rho);
    }
    default: (void)raw; return 0.0;
  }
}

}  // namespace lga

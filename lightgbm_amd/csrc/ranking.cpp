#include "ranking.hpp"
#include "common.hpp"
#include <algorithm>
#include <cmath>
#include <numeric>
#include <stdexcept>

namespace lga {

namespace {

inline double discount(int rank) { return rank_discount(rank); }

// Positions of a query's documents by descending score; equal scores keep their row order.
void order_by_score(const double* score, int64_t cnt, std::vector<int>* idx) {
  idx->resize((size_t)cnt);
  std::iota(idx->begin(), idx->end(), 0);
  std::stable_sort(idx->begin(), idx->end(), [&](int a, int b) { return score[a] > score[b]; });
}

// Lambdas of one query, every document summing its own pair terms by ascending partner rank (ranking.hpp).
void lambdarank_query(const RankingSpec& s, const double* score, const float* label, int64_t cnt, double inv_max_dcg,
                      float* g, float* h, std::vector<int>& idx, std::vector<double>& lam, std::vector<double>& hes) {
  lam.assign((size_t)cnt, 0.0);
  hes.assign((size_t)cnt, 0.0);
  if (cnt > 1 && inv_max_dcg > 0.0) {
    order_by_score(score, cnt, &idx);
    const double best = score[idx[0]], worst = score[idx[(size_t)cnt - 1]];
    const bool normalise = s.norm && best != worst;
    const int64_t top = std::min<int64_t>(cnt, s.truncation_level);
    double total = 0.0;
    for (int64_t r = 0; r < cnt; ++r) {
      const int me = idx[(size_t)r];
      const double my_gain = s.label_gain[(size_t)label[me]], my_disc = discount((int)r);
      // a pair needs one member among the first `top` ranks
      const int64_t partners = r < top ? cnt : top;
      double l = 0.0, hh = 0.0, part = 0.0;
      for (int64_t q = 0; q < partners; ++q) {
        const int other = idx[(size_t)q];
        if (q == r || label[other] == label[me]) continue;
        const double o_gain = s.label_gain[(size_t)label[other]], o_disc = discount((int)q);
        double pl, ph;
        if (label[me] > label[other]) {
          rank_pair(score[me], score[other], my_gain, o_gain, my_disc, o_disc, inv_max_dcg, s.sigmoid, normalise, &pl, &ph);
          l += pl;
        } else {
          rank_pair(score[other], score[me], o_gain, my_gain, o_disc, my_disc, inv_max_dcg, s.sigmoid, normalise, &pl, &ph);
          l -= pl;
        }
        hh += ph;
        if (q > r) part -= 2.0 * pl;
      }
      lam[(size_t)me] = l;
      hes[(size_t)me] = hh;
      total += part;
    }
    if (s.norm && total > 0.0) {
      const double f = std::log2(1.0 + total) / total;
      for (int64_t i = 0; i < cnt; ++i) { lam[(size_t)i] *= f; hes[(size_t)i] *= f; }
    }
  }
  for (int64_t i = 0; i < cnt; ++i) { g[i] = (float)lam[(size_t)i]; h[i] = (float)hes[(size_t)i]; }
}

// Cross-entropy surrogate of NDCG (Bruch, "An Alternative Cross Entropy Loss for Learning-to-Rank"): the
// softmax of the scores is fitted to the distribution of 2^label - u, u ~ U(0, 1) drawn per document.
void xendcg_query(const RankingSpec& s, const double* score, const float* label, int64_t cnt, int64_t first_row,
                  uint64_t iter, float* g, float* h, std::vector<double>& rho, std::vector<double>& par) {
  if (cnt <= 1) {
    for (int64_t i = 0; i < cnt; ++i) { g[i] = 0.0f; h[i] = 0.0f; }
    return;
  }
  rho.resize((size_t)cnt);
  par.resize((size_t)cnt);
  double mx = score[0];
  for (int64_t i = 1; i < cnt; ++i) mx = std::max(mx, score[i]);
  double z = 0.0;
  for (int64_t i = 0; i < cnt; ++i) { rho[(size_t)i] = std::exp(score[i] - mx); z += rho[(size_t)i]; }
  for (int64_t i = 0; i < cnt; ++i) rho[(size_t)i] /= z;
  double sum_labels = 0.0;
  for (int64_t i = 0; i < cnt; ++i) {
    const double u = (double)rng_uniform(s.seed, iter, (uint64_t)(first_row + i));
    par[(size_t)i] = std::ldexp(1.0, (int)label[i]) - u;
    sum_labels += par[(size_t)i];
  }
  const double inv_denominator = 1.0 / std::max(sum_labels, 1e-15);
  std::vector<double> lam((size_t)cnt);
  // first-order term, then two correction terms of the series for the inverse Hessian
  double sum_l1 = 0.0;
  for (int64_t i = 0; i < cnt; ++i) {
    const double term = -par[(size_t)i] * inv_denominator + rho[(size_t)i];
    lam[(size_t)i] = term;
    par[(size_t)i] = term / (1.0 - rho[(size_t)i]);
    sum_l1 += par[(size_t)i];
  }
  double sum_l2 = 0.0;
  for (int64_t i = 0; i < cnt; ++i) {
    const double term = rho[(size_t)i] * (sum_l1 - par[(size_t)i]);
    lam[(size_t)i] += term;
    par[(size_t)i] = term / (1.0 - rho[(size_t)i]);
    sum_l2 += par[(size_t)i];
  }
  for (int64_t i = 0; i < cnt; ++i) {
    lam[(size_t)i] += rho[(size_t)i] * (sum_l2 - par[(size_t)i]);
    g[i] = (float)lam[(size_t)i];
    h[i] = (float)(rho[(size_t)i] * (1.0 - rho[(size_t)i]));
  }
}

double query_weight(const float* weight, int64_t b, int64_t e) {
  if (!weight || e <= b) return 1.0;
  double s = 0.0;
  for (int64_t i = b; i < e; ++i) s += weight[i];
  return s / (double)(e - b);
}

}  // namespace

std::vector<double> default_label_gain() {
  std::vector<double> g(31);
  for (int i = 0; i < 31; ++i) g[(size_t)i] = (double)((1ll << i) - 1);
  return g;
}

void check_rank_labels(const float* label, int64_t n, const std::vector<double>& label_gain) {
  for (int64_t i = 0; i < n; ++i) {
    const float l = label[i];
    if (!(l >= 0.0f) || l != (float)(int)l)
      throw std::runtime_error("ranking labels must be non-negative integers (relevance grades), got " + std::to_string(l));
    if ((size_t)l >= label_gain.size())
      throw std::runtime_error("ranking label " + std::to_string((int)l) + " has no entry in label_gain (" +
                               std::to_string(label_gain.size()) + " grades)");
  }
}

std::vector<double> inverse_max_dcg(const float* label, const std::vector<int64_t>& bounds,
                                    const std::vector<double>& label_gain, int k) {
  const int64_t nq = (int64_t)bounds.size() - 1;
  std::vector<double> out((size_t)std::max<int64_t>(nq, 0), 0.0);
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t q = 0; q < nq; ++q) {
    std::vector<float> l(label + bounds[(size_t)q], label + bounds[(size_t)q + 1]);
    std::sort(l.begin(), l.end(), [](float a, float b) { return a > b; });
    double dcg = 0.0;
    const int64_t top = std::min<int64_t>((int64_t)l.size(), k);
    for (int64_t i = 0; i < top; ++i) dcg += label_gain[(size_t)l[(size_t)i]] * discount((int)i);
    out[(size_t)q] = dcg > 0.0 ? 1.0 / dcg : 0.0;
  }
  return out;
}

void ranking_gradients(const RankingSpec& spec, const double* score, const float* label, const float* weight,
                       const std::vector<int64_t>& bounds, const std::vector<double>& inv_max_dcg, uint64_t iter,
                       float* g, float* h) {
  const int64_t nq = (int64_t)bounds.size() - 1;
#pragma omp parallel
  {
    std::vector<int> idx;
    std::vector<double> a, b;
#pragma omp for schedule(dynamic, 16)
    for (int64_t q = 0; q < nq; ++q) {
      const int64_t s = bounds[(size_t)q], cnt = bounds[(size_t)q + 1] - s;
      if (spec.xendcg) xendcg_query(spec, score + s, label + s, cnt, s, iter, g + s, h + s, a, b);
      else lambdarank_query(spec, score + s, label + s, cnt, inv_max_dcg[(size_t)q], g + s, h + s, idx, a, b);
      if (weight)
        for (int64_t i = s; i < s + cnt; ++i) { g[i] *= weight[i]; h[i] *= weight[i]; }
    }
  }
}

void ndcg_sums(const double* score, const float* label, const float* weight, const std::vector<int64_t>& bounds,
               const std::vector<double>& label_gain, const std::vector<int>& at, std::vector<double>* sums) {
  const int64_t nq = (int64_t)bounds.size() - 1;
  const size_t K = at.size();
  sums->assign(K + 1, 0.0);
  std::vector<double> per_query((size_t)std::max<int64_t>(nq, 0) * (K + 1), 0.0);
#pragma omp parallel
  {
    std::vector<int> idx;
    std::vector<float> sorted;
#pragma omp for schedule(dynamic, 64)
    for (int64_t q = 0; q < nq; ++q) {
      const int64_t s = bounds[(size_t)q], cnt = bounds[(size_t)q + 1] - s;
      double* out = &per_query[(size_t)q * (K + 1)];
      const double w = query_weight(weight, s, s + cnt);
      out[K] = w;
      order_by_score(score + s, cnt, &idx);
      sorted.assign(label + s, label + s + cnt);
      std::sort(sorted.begin(), sorted.end(), [](float a, float b) { return a > b; });
      for (size_t k = 0; k < K; ++k) {
        const int64_t top = std::min<int64_t>(cnt, at[k]);
        double dcg = 0.0, best = 0.0;
        for (int64_t i = 0; i < top; ++i) {
          dcg += label_gain[(size_t)label[s + idx[(size_t)i]]] * discount((int)i);
          best += label_gain[(size_t)sorted[(size_t)i]] * discount((int)i);
        }
        out[k] = w * (best > 0.0 ? dcg / best : 1.0);
      }
    }
  }
  // fixed summation order: the result does not depend on the thread count
  for (int64_t q = 0; q < nq; ++q)
    for (size_t k = 0; k <= K; ++k) (*sums)[k] += per_query[(size_t)q * (K + 1) + k];
}

void weighted_query_sums(const double* values, size_t stride, size_t offset, size_t n_at, const float* weight,
                         const std::vector<int64_t>& bounds, std::vector<double>* sums) {
  const int64_t nq = (int64_t)bounds.size() - 1;
  sums->assign(n_at + 1, 0.0);
  for (int64_t q = 0; q < nq; ++q) {
    const double w = query_weight(weight, bounds[(size_t)q], bounds[(size_t)q + 1]);
    for (size_t k = 0; k < n_at; ++k) (*sums)[k] += w * values[(size_t)q * stride + offset + k];
    (*sums)[n_at] += w;
  }
}

void map_sums(const double* score, const float* label, const float* weight, const std::vector<int64_t>& bounds,
              const std::vector<int>& at, std::vector<double>* sums) {
  const int64_t nq = (int64_t)bounds.size() - 1;
  const size_t K = at.size();
  sums->assign(K + 1, 0.0);
  std::vector<double> per_query((size_t)std::max<int64_t>(nq, 0) * (K + 1), 0.0);
#pragma omp parallel
  {
    std::vector<int> idx;
#pragma omp for schedule(dynamic, 64)
    for (int64_t q = 0; q < nq; ++q) {
      const int64_t s = bounds[(size_t)q], cnt = bounds[(size_t)q + 1] - s;
      double* out = &per_query[(size_t)q * (K + 1)];
      const double w = query_weight(weight, s, s + cnt);
      out[K] = w;
      order_by_score(score + s, cnt, &idx);
      int64_t relevant = 0;
      for (int64_t i = 0; i < cnt; ++i) relevant += label[s + i] > 0.5f ? 1 : 0;
      for (size_t k = 0; k < K; ++k) {
        const int64_t top = std::min<int64_t>(cnt, at[k]);
        int64_t hits = 0;
        double ap = 0.0;
        for (int64_t i = 0; i < top; ++i)
          if (label[s + idx[(size_t)i]] > 0.5f) { ++hits; ap += (double)hits / (double)(i + 1); }
        out[k] = w * (relevant > 0 ? ap / (double)std::min<int64_t>(relevant, top) : 1.0);
      }
    }
  }
  for (int64_t q = 0; q < nq; ++q)
    for (size_t k = 0; k <= K; ++k) (*sums)[k] += per_query[(size_t)q * (K + 1) + k];
}

}  // namespace lga

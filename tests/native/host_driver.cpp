// Stand-alone exercise of the host core through its C ABI, meant to be built with
// -fsanitize=address,undefined (tools/sanitize_host.py): dataset construction with missing values and a
// constant column, zero-copy subsets, gbdt with bagging / feature sampling and a validation set, rf,
// custom gradients, rollback, model text round trip, prediction, and a two-rank data-parallel run over
// an in-process communicator.  Exit code 0 = every check held and the sanitizers stayed silent.
#include "c_api.hpp"
#include "binning_core.hpp"
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define OK(call)                                                                       \
  do {                                                                                 \
    if ((call) != 0) { std::fprintf(stderr, "FAILED %s: %s\n", #call, lga_last_error()); std::exit(1); } \
  } while (0)
#define REQUIRE(cond)                                                                  \
  do {                                                                                 \
    if (!(cond)) { std::fprintf(stderr, "check failed at line %d: %s\n", __LINE__, #cond); std::exit(1); } \
  } while (0)

static uint64_t rng_state = 12345;
static double uni() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (double)(rng_state >> 11) / 9007199254740992.0;
}
static double gauss() { return std::sqrt(-2.0 * std::log(uni() + 1e-300)) * std::cos(6.283185307179586 * uni()); }

// ---- a two-rank in-process communicator (barrier + shared slots) ----
struct Hub {
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0, generation = 0;
  std::vector<char> slot[2];
  void barrier() {
    std::unique_lock<std::mutex> l(m);
    int g = generation;
    if (++arrived == 2) { arrived = 0; ++generation; cv.notify_all(); }
    else cv.wait(l, [&] { return generation != g; });
  }
};
static Hub hub;
static thread_local int t_rank = 0;
static void allreduce_cb(void* buf, int64_t n, int dtype, int op) {
  size_t esz = dtype == 2 ? 4 : 8;
  hub.slot[t_rank].assign((char*)buf, (char*)buf + n * esz);
  hub.barrier();
  const char* a = hub.slot[0].data();
  const char* b = hub.slot[1].data();
  for (int64_t i = 0; i < n; ++i) {
    if (dtype == 0) { int64_t x, y; std::memcpy(&x, a + i * 8, 8); std::memcpy(&y, b + i * 8, 8); x += y; std::memcpy((char*)buf + i * 8, &x, 8); }
    else if (dtype == 1) { double x, y; std::memcpy(&x, a + i * 8, 8); std::memcpy(&y, b + i * 8, 8); x = op == 1 ? std::fmax(x, y) : x + y; std::memcpy((char*)buf + i * 8, &x, 8); }
    else { float x, y; std::memcpy(&x, a + i * 4, 4); std::memcpy(&y, b + i * 4, 4); x = op == 1 ? std::fmax(x, y) : x + y; std::memcpy((char*)buf + i * 4, &x, 4); }
  }
  hub.barrier();
}
static void allgather_cb(const void* in, void* out, int64_t nbytes) {
  hub.slot[t_rank].assign((const char*)in, (const char*)in + nbytes);
  hub.barrier();
  std::memcpy(out, hub.slot[0].data(), (size_t)nbytes);
  std::memcpy((char*)out + nbytes, hub.slot[1].data(), (size_t)nbytes);
  hub.barrier();
}

static std::string model_text(LgaBooster b) {
  const char* s = nullptr;
  int64_t len = 0;
  OK(lga_booster_save_string(b, 0, -1, 0, &s, &len));
  return std::string(s, (size_t)len);
}
static std::string trees_of(const std::string& m) {
  size_t a = m.find("Tree=0"), b = m.find("end of trees");
  return m.substr(a, b - a);
}

// The cut-to-cut form of the bin-boundary search must reproduce the value-by-value form exactly.
static void check_jump_cut() {
  for (int trial = 0; trial < 3000; ++trial) {
    const int n = 1 + (int)(uni() * (trial % 3 == 0 ? 40 : 3000));
    const int max_bin = 2 + (int)(uni() * 254);
    const int min_data = 1 + (int)(uni() * (trial % 5 == 0 ? 200 : 6));
    const int style = trial % 4;
    std::vector<double> distinct((size_t)n);
    std::vector<int> counts((size_t)n);
    double v = -5.0;
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
      v += 1e-3 + uni();
      distinct[(size_t)i] = v;
      int c = 1;
      if (style == 0) c = 1 + (int)(uni() * 3);
      else if (style == 1) c = uni() < 0.02 ? 1 + (int)(uni() * 5000) : 1 + (int)(uni() * 4);
      else if (style == 2) c = 1 + (int)(uni() * uni() * 400);
      else c = (i % 50 == 7) ? 3000 : 1;
      counts[(size_t)i] = c;
      total += c;
    }
    double mean = (double)total / (double)max_bin;
    if (mean < (double)min_data) mean = (double)min_data;
    std::vector<int64_t> prefix((size_t)n + 1, 0), small((size_t)n + 1, 0);
    std::vector<int> next_big((size_t)n + 1, n);
    for (int i = 0; i < n; ++i) {
      prefix[(size_t)i + 1] = prefix[(size_t)i] + counts[(size_t)i];
      small[(size_t)i + 1] = small[(size_t)i] + ((double)counts[(size_t)i] >= mean ? 0 : counts[(size_t)i]);
    }
    for (int i = n - 1; i >= 0; --i) next_big[(size_t)i] = (double)counts[(size_t)i] >= mean ? i : next_big[(size_t)i + 1];
    std::vector<double> a((size_t)max_bin + 2), b((size_t)max_bin + 2);
    const int na = lga::greedy_bounds_core(distinct.data(), counts.data(), n, max_bin, total, min_data, a.data());
    const int nb = lga::greedy_bounds_jump(distinct.data(), counts.data(), prefix.data(), small.data(), next_big.data(), n,
                                           max_bin, total, min_data, b.data());
    REQUIRE(na == nb);
    for (int i = 0; i < na; ++i) REQUIRE(a[(size_t)i] == b[(size_t)i]);
  }
}

int main() {
  check_jump_cut();
  const int64_t n = 6000;
  const int f = 37;   // two feature tiles
  std::vector<double> X((size_t)n * f);
  std::vector<float> y((size_t)n), w((size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    for (int j = 0; j < f; ++j) X[(size_t)(i * f + j)] = gauss();
    X[(size_t)(i * f + 5)] = 1.5;                                   // constant column
    X[(size_t)(i * f + 7)] = (double)(int)(uni() * 4);              // few distinct values
    if (i % 9 == 0) X[(size_t)(i * f + 3)] = NAN;
    double s = X[(size_t)(i * f)] * 1.2 - X[(size_t)(i * f + 1)] + std::sin(2 * X[(size_t)(i * f + 2)]) + 0.3 * gauss();
    y[(size_t)i] = (float)s;
    w[(size_t)i] = (float)(0.5 + uni());
  }
  const char* dparams = "max_bin=63 min_data_in_bin=3 verbose=-1 device_type=cpu";
  LgaDataset full = nullptr;
  OK(lga_dataset_from_mat(X.data(), 1, n, f, 1, dparams, nullptr, 1, &full));
  OK(lga_dataset_set_field(full, "label", y.data(), n, 0));
  OK(lga_dataset_set_field(full, "weight", w.data(), n, 0));
  int used = 0;
  OK(lga_dataset_num_used_feature(full, &used));
  REQUIRE(used == f - 1);

  // folds as zero-copy subsets
  std::vector<int32_t> tr, te;
  for (int32_t i = 0; i < n; ++i) (i % 5 == 0 ? te : tr).push_back(i);
  LgaDataset dtr = nullptr, dte = nullptr;
  OK(lga_dataset_subset(full, tr.data(), (int64_t)tr.size(), &dtr));
  OK(lga_dataset_subset(full, te.data(), (int64_t)te.size(), &dte));

  // gbdt with bagging + feature sampling + validation
  LgaBooster b = nullptr;
  OK(lga_booster_create(dtr, "objective=regression num_leaves=31 learning_rate=0.2 bagging_fraction=0.7 bagging_freq=2 "
                             "feature_fraction=0.8 lambda_l2=0.5 metric=rmse,l1 verbose=-1 device_type=cpu", nullptr, &b));
  OK(lga_booster_add_valid(b, dte, "valid"));
  double first = 0, last = 0;
  for (int it = 0; it < 15; ++it) {
    int fin = 0;
    OK(lga_booster_update(b, &fin));
    double ev[8];
    int ne = 0;
    OK(lga_booster_eval(b, 0, ev, 8, &ne));
    REQUIRE(ne == 2 && std::isfinite(ev[0]));
    if (it == 0) first = ev[0];
    last = ev[0];
  }
  REQUIRE(last < 0.7 * first);
  OK(lga_booster_rollback(b));
  int nt = 0;
  OK(lga_booster_num_trees(b, &nt));
  REQUIRE(nt == 14);

  // text round trip + prediction on raw rows
  std::string m1 = model_text(b);
  LgaBooster b2 = nullptr;
  OK(lga_booster_from_string(m1.c_str(), &b2));
  REQUIRE(model_text(b2) == m1);
  std::vector<double> p1((size_t)n), p2((size_t)n);
  OK(lga_booster_predict(b, X.data(), 1, n, f, 1, 0, -1, 0, p1.data()));
  OK(lga_booster_predict(b2, X.data(), 1, n, f, 1, 0, -1, 0, p2.data()));
  for (int64_t i = 0; i < n; ++i) REQUIRE(p1[(size_t)i] == p2[(size_t)i]);
  std::vector<double> imp((size_t)f);
  OK(lga_booster_feature_importance(b, -1, 0, imp.data()));
  REQUIRE(imp[5] == 0.0);

  // random forest + custom gradients
  LgaBooster rf = nullptr;
  OK(lga_booster_create(full, "boosting=rf objective=regression num_leaves=15 bagging_fraction=0.63 bagging_freq=1 "
                              "feature_fraction=0.8 verbose=-1 device_type=cpu", nullptr, &rf));
  for (int it = 0; it < 5; ++it) { int fin = 0; OK(lga_booster_update(rf, &fin)); }
  LgaBooster cu = nullptr;
  OK(lga_booster_create(full, "objective=custom num_leaves=15 verbose=-1 device_type=cpu", nullptr, &cu));
  std::vector<float> g((size_t)n), h((size_t)n, 1.0f);
  std::vector<double> sc((size_t)n);
  for (int it = 0; it < 3; ++it) {
    OK(lga_booster_inner_predict(cu, -1, 1, sc.data()));
    for (int64_t i = 0; i < n; ++i) g[(size_t)i] = (float)(sc[(size_t)i] - y[(size_t)i]);
    int fin = 0;
    OK(lga_booster_update_custom(cu, g.data(), h.data(), n, &fin));
  }

  // two-rank data-parallel == serial
  LgaBooster serial = nullptr;
  OK(lga_booster_create(full, "objective=regression num_leaves=31 verbose=-1 device_type=cpu", nullptr, &serial));
  for (int it = 0; it < 6; ++it) { int fin = 0; OK(lga_booster_update(serial, &fin)); }
  std::string serial_trees = trees_of(model_text(serial));
  std::string rank_trees[2];
  auto worker = [&](int r) {
    t_rank = r;
    const int64_t lo = n * r / 2, hi = n * (r + 1) / 2;
    LgaDataset shard = nullptr;
    OK(lga_dataset_from_reference(full, hi - lo, &shard));
    OK(lga_dataset_bin_matrix(shard, X.data() + lo * f, 1, hi - lo, f, 1));
    OK(lga_dataset_set_field(shard, "label", y.data() + lo, hi - lo, 0));
    OK(lga_dataset_set_field(shard, "weight", w.data() + lo, hi - lo, 0));
    LgaComm comm = nullptr;
    OK(lga_comm_create_callback(r, 2, allreduce_cb, allgather_cb, &comm));
    LgaBooster db = nullptr;
    OK(lga_booster_create(shard, "objective=regression num_leaves=31 tree_learner=data verbose=-1 device_type=cpu", comm, &db));
    for (int it = 0; it < 6; ++it) { int fin = 0; OK(lga_booster_update(db, &fin)); }
    rank_trees[r] = trees_of(model_text(db));
    OK(lga_booster_free(db));
    OK(lga_comm_free(comm));
    OK(lga_dataset_free(shard));
  };
  std::thread t0(worker, 0), t1(worker, 1);
  t0.join(); t1.join();
  // Split structure, thresholds, counts and gains are identical; leaf values carry the boost-from-average
  // start, a weighted mean whose floating-point sum is grouped differently over two ranks (last bits).
  auto structure = [](const std::string& t) {
    std::string out;
    size_t pos = 0;
    while (pos < t.size()) {
      size_t e = t.find('\n', pos);
      if (e == std::string::npos) e = t.size();
      std::string line = t.substr(pos, e - pos);
      if (line.rfind("leaf_value=", 0) != 0 && line.rfind("internal_value=", 0) != 0) out += line + "\n";
      pos = e + 1;
    }
    return out;
  };
  REQUIRE(rank_trees[0] == rank_trees[1]);
  REQUIRE(structure(rank_trees[0]) == structure(serial_trees));
  {
    LgaBooster s2 = nullptr;
    std::string whole = model_text(serial);
    std::string swapped = whole.substr(0, whole.find("Tree=0")) + rank_trees[0] + whole.substr(whole.find("end of trees"));
    OK(lga_booster_from_string(swapped.c_str(), &s2));
    OK(lga_booster_predict(serial, X.data(), 1, n, f, 1, 0, -1, 0, p1.data()));
    OK(lga_booster_predict(s2, X.data(), 1, n, f, 1, 0, -1, 0, p2.data()));
    for (int64_t i = 0; i < n; ++i) REQUIRE(std::fabs(p1[(size_t)i] - p2[(size_t)i]) < 1e-12);
    OK(lga_booster_free(s2));
  }

  // the later additions under the sanitizers: ranking (gradients per query, ndcg / map), category-set
  // splits with constraints and path smoothing, multiclass with GOSS, leaf refit
  {
    const int64_t m = 4000;
    const int fc = 6;
    std::vector<double> Xc((size_t)m * fc);
    std::vector<float> rel((size_t)m), cls((size_t)m), reg((size_t)m);
    std::vector<double> groups;
    for (int64_t i = 0; i < m; ++i) {
      for (int j = 0; j < fc; ++j) Xc[(size_t)(i * fc + j)] = gauss();
      const int cat = (int)(uni() * 40);
      Xc[(size_t)(i * fc + 4)] = (double)cat;
      if (i % 11 == 0) Xc[(size_t)(i * fc + 1)] = 0.0;
      const double s = Xc[(size_t)(i * fc)] + std::sin((double)cat) + 0.3 * gauss();
      reg[(size_t)i] = (float)s;
      rel[(size_t)i] = (float)std::min(4, std::max(0, (int)std::lround(s + 1.5)));
      cls[(size_t)i] = (float)std::min(2, std::max(0, (int)std::lround(s + 1.0)));
    }
    for (int64_t left = m; left > 0;) { int64_t q = std::min<int64_t>(left, 7 + (int64_t)(uni() * 60)); groups.push_back((double)q); left -= q; }
    const char* dp = "max_bin=63 verbose=-1 device_type=cpu categorical_feature=4 zero_as_missing=true";
    LgaDataset dc = nullptr;
    OK(lga_dataset_from_mat(Xc.data(), 1, m, fc, 1, dp, nullptr, 1, &dc));
    OK(lga_dataset_set_field(dc, "label", rel.data(), m, 0));
    OK(lga_dataset_set_field(dc, "group", groups.data(), (int64_t)groups.size(), 1));
    for (const char* obj : {"lambdarank", "rank_xendcg"}) {
      LgaBooster rb = nullptr;
      std::string ps = std::string("objective=") + obj + " num_leaves=15 min_data_in_leaf=5 metric=ndcg,map eval_at=1,5 "
                       "bagging_fraction=0.7 bagging_freq=1 verbose=-1 device_type=cpu categorical_feature=4 zero_as_missing=true";
      OK(lga_booster_create(dc, ps.c_str(), nullptr, &rb));
      double ev[8];
      int ne = 0;
      for (int it = 0; it < 4; ++it) { int fin = 0; OK(lga_booster_update(rb, &fin)); }
      OK(lga_booster_eval(rb, -1, ev, 8, &ne));
      REQUIRE(ne == 4 && ev[1] > 0.5 && ev[1] <= 1.0);
      OK(lga_booster_free(rb));
    }
    OK(lga_dataset_set_field(dc, "group", nullptr, 0, 1));
    OK(lga_dataset_set_field(dc, "label", reg.data(), m, 0));
    LgaBooster cb = nullptr;
    OK(lga_booster_create(dc, "objective=regression num_leaves=31 min_data_in_leaf=5 path_smooth=5 min_data_per_group=20 "
                              "monotone_constraints=1,0,0,0,0,0 interaction_constraints=[0,1,4],[2,3,5] "
                              "verbose=-1 device_type=cpu categorical_feature=4 zero_as_missing=true", nullptr, &cb));
    for (int it = 0; it < 6; ++it) { int fin = 0; OK(lga_booster_update(cb, &fin)); }
    std::string cm = model_text(cb);
    REQUIRE(cm.find("cat_threshold=") != std::string::npos);
    std::vector<double> sc2((size_t)m), pr2((size_t)m);
    OK(lga_booster_inner_predict(cb, -1, 1, sc2.data()));
    OK(lga_booster_predict(cb, Xc.data(), 1, m, fc, 1, 0, -1, 1, pr2.data()));
    for (int64_t i = 0; i < m; ++i) REQUIRE(std::fabs(sc2[(size_t)i] - pr2[(size_t)i]) < 1e-9);
    OK(lga_booster_refit(cb, Xc.data(), 1, m, fc, 1, reg.data(), nullptr, 0.5, ""));
    OK(lga_booster_free(cb));
    OK(lga_dataset_set_field(dc, "label", cls.data(), m, 0));
    LgaBooster mb = nullptr;
    OK(lga_booster_create(dc, "objective=multiclass num_class=3 num_leaves=15 data_sample_strategy=goss learning_rate=0.5 "
                              "metric=multi_logloss,auc_mu verbose=-1 device_type=cpu categorical_feature=4 zero_as_missing=true",
                          nullptr, &mb));
    for (int it = 0; it < 5; ++it) { int fin = 0; OK(lga_booster_update(mb, &fin)); }
    double ev[4];
    int ne = 0;
    OK(lga_booster_eval(mb, -1, ev, 4, &ne));
    REQUIRE(ne == 2 && ev[1] > 0.6);
    OK(lga_booster_free(mb));
    OK(lga_dataset_free(dc));
  }

  // error path: message, no crash
  LgaBooster bad = nullptr;
  REQUIRE(lga_booster_create(full, "boosting=rf verbose=-1 device_type=cpu", nullptr, &bad) != 0);
  REQUIRE(std::strlen(lga_last_error()) > 0);

  for (LgaBooster x : {b, b2, rf, cu, serial}) OK(lga_booster_free(x));
  for (LgaDataset d : {dtr, dte, full}) OK(lga_dataset_free(d));
  std::printf("host driver: all checks passed\n");
  return 0;
}

#include "c_api.hpp"
#include "engine_factory.hpp"
#include "gbdt.hpp"
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>

using namespace lga;

namespace {
thread_local std::string g_error;
thread_local std::string g_str;

struct DatasetBox { std::shared_ptr<Dataset> ds; };
struct BoosterBox { std::unique_ptr<Booster> b; std::string scratch; };

class CallbackComm : public Comm {
 public:
  CallbackComm(int rank, int size, lga_allreduce_fn ar, lga_allgather_fn ag) : rank_(rank), size_(size), ar_(ar), ag_(ag) {}
  int rank() const override { return rank_; }
  int size() const override { return size_; }
  bool host_only() const override { return true; }
  void allreduce_sum_i64(int64_t* buf, size_t n, bool device, void*) override { need_host(device); ar_(buf, (int64_t)n, 0, 0); }
  void allreduce_sum_f64(double* buf, size_t n, bool device, void*) override { need_host(device); ar_(buf, (int64_t)n, 1, 0); }
  void allreduce_max_f32(float* buf, size_t n, bool device, void*) override { need_host(device); ar_(buf, (int64_t)n, 2, 1); }
  void allgather_bytes(const void* in, void* out, size_t nbytes, bool device, void*) override {
    need_host(device);
    ag_(in, out, (int64_t)nbytes);
  }

 private:
  static void need_host(bool device) {
    if (device) throw std::runtime_error("callback communicator cannot reduce device memory");
  }
  int rank_, size_;
  lga_allreduce_fn ar_;
  lga_allgather_fn ag_;
};

template <typename F>
int guard(F f) {
  try { f(); return 0; }
  catch (const std::exception& e) { g_error = e.what(); return -1; }
  catch (...) { g_error = "unknown error"; return -1; }
}
Dataset* D(LgaDataset h) { if (!h) throw std::runtime_error("null dataset handle"); return ((DatasetBox*)h)->ds.get(); }
std::shared_ptr<Dataset> DS(LgaDataset h) { if (!h) throw std::runtime_error("null dataset handle"); return ((DatasetBox*)h)->ds; }
Booster* B(LgaBooster h) { if (!h) throw std::runtime_error("null booster handle"); return ((BoosterBox*)h)->b.get(); }
}  // namespace

extern "C" {

const char* lga_last_error() { return g_error.c_str(); }
void lga_set_last_error(const char* msg) { g_error = msg ? msg : ""; }
const char* lga_build_info() { return engine_build_name(); }
const char* lga_alias_table_json() {
  g_str = Config::alias_table_json();
  return g_str.c_str();
}
int lga_resolve_alias(const char* name, char* out, int out_len) {
  return guard([&] {
    std::string r = Config::resolve_alias(name);
    if ((int)r.size() + 1 > out_len) throw std::runtime_error("buffer too small");
    std::strcpy(out, r.c_str());
  });
}

int lga_dataset_from_mat(const void* data, int dtype, int64_t nrow, int ncol, int row_major, const char* params,
                         LgaDataset reference, int bin_on_host, LgaDataset* out) {
  return guard([&] {
    Config cfg;
    cfg.set(std::string(params ? params : ""));
    auto ds = Dataset::from_matrix(data, dtype, nrow, ncol, row_major != 0, cfg, reference ? D(reference) : nullptr,
                                   bin_on_host != 0);
    *out = new DatasetBox{ds};
  });
}

int lga_dataset_from_sample(const double* sample, int64_t nsample, int ncol, int64_t nrow, const char* params,
                            LgaDataset* out) {
  return guard([&] {
    Config cfg;
    cfg.set(std::string(params ? params : ""));
    *out = new DatasetBox{Dataset::from_sample(sample, nsample, ncol, nrow, cfg)};
  });
}

int lga_dataset_bin_matrix(LgaDataset ds, const void* data, int dtype, int64_t nrow, int ncol, int row_major) {
  return guard([&] {
    Dataset* d = D(ds);
    if (d->is_subset()) throw std::runtime_error("cannot bin into a subset");
    if (nrow != d->root_rows()) throw std::runtime_error("row count differs from the dataset's");
    if (ncol != d->num_total_features()) throw std::runtime_error("column count differs from the dataset's");
    std::vector<uint8_t> rt((size_t)d->num_tiles() * (size_t)nrow * kTileFeatures, 0), fm((size_t)d->num_features() * (size_t)nrow, 0);
    d->bin_rows(data, dtype, nrow, ncol, row_major != 0, rt.data(), fm.data(), nrow, 0);
    d->adopt_host_bins(std::move(rt), std::move(fm));
  });
}

int lga_dataset_from_reference(LgaDataset ref, int64_t nrow, LgaDataset* out) {
  return guard([&] { *out = new DatasetBox{Dataset::from_reference(D(ref), nrow)}; });
}

int lga_dataset_subset(LgaDataset ds, const int32_t* idx, int64_t n, LgaDataset* out) {
  return guard([&] { *out = new DatasetBox{D(ds)->subset(idx, n)}; });
}

int lga_dataset_set_field(LgaDataset ds, const char* name, const void* data, int64_t n, int dtype) {
  return guard([&] {
    std::string f = name;
    Dataset* d = D(ds);
    auto as_float = [&]() {
      std::vector<float> v((size_t)n);
      for (int64_t i = 0; i < n; ++i) v[(size_t)i] = dtype == 0 ? ((const float*)data)[i] : (float)((const double*)data)[i];
      return v;
    };
    if (f == "label") { auto v = as_float(); d->set_label(v.data(), n); }
    else if (f == "weight") { auto v = as_float(); d->set_weight(v.data(), n); }
    else if (f == "init_score") {
      std::vector<double> v((size_t)n);
      for (int64_t i = 0; i < n; ++i) v[(size_t)i] = dtype == 0 ? (double)((const float*)data)[i] : ((const double*)data)[i];
      d->set_init_score(v.data(), n);
    } else if (f == "group") {
      std::vector<int32_t> v((size_t)n);
      for (int64_t i = 0; i < n; ++i) {
        const double x = dtype == 0 ? (double)((const float*)data)[i] : ((const double*)data)[i];
        if (!(x >= 0.0) || x != (double)(int32_t)x) throw std::runtime_error("group sizes must be non-negative integers");
        v[(size_t)i] = (int32_t)x;
      }
      d->set_group(v.data(), n);
    } else throw std::runtime_error("unknown field " + f);
  });
}

int lga_dataset_get_field(LgaDataset ds, const char* name, double* out, int64_t cap, int64_t* n) {
  return guard([&] {
    std::string f = name;
    Dataset* d = D(ds);
    std::vector<double> v;
    if (f == "label") { for (float x : d->local_label()) v.push_back(x); }
    else if (f == "weight") { for (float x : d->local_weight()) v.push_back(x); }
    else if (f == "init_score") {
      const auto& is = d->root_init_score();
      if (!is.empty()) {
        if (d->used_indices()) {
          const size_t classes = is.size() / (size_t)d->root_rows();
          for (size_t k = 0; k < classes; ++k)
            for (int32_t r : *d->used_indices()) v.push_back(is[k * (size_t)d->root_rows() + (size_t)r]);
        } else v = is;
      }
    } else if (f == "group") {
      const auto& b = d->query_boundaries();
      for (size_t q = 0; q + 1 < b.size(); ++q) v.push_back((double)(b[q + 1] - b[q]));
    } else throw std::runtime_error("unknown field " + f);
    *n = (int64_t)v.size();
    if (out) {
      if (cap < *n) throw std::runtime_error("buffer too small");
      std::memcpy(out, v.data(), v.size() * sizeof(double));
    }
  });
}

int lga_dataset_num_data(LgaDataset ds, int64_t* out) { return guard([&] { *out = D(ds)->num_data(); }); }
int lga_dataset_num_feature(LgaDataset ds, int* out) { return guard([&] { *out = D(ds)->num_total_features(); }); }
int lga_dataset_num_used_feature(LgaDataset ds, int* out) { return guard([&] { *out = D(ds)->num_features(); }); }

int lga_dataset_set_feature_names(LgaDataset ds, const char* joined) {
  return guard([&] {
    std::vector<std::string> names;
    std::istringstream is(joined);
    std::string l;
    while (std::getline(is, l)) names.push_back(l);
    D(ds)->set_feature_names(names);
  });
}

int lga_dataset_get_feature_names(LgaDataset ds, char* out, int64_t cap, int64_t* needed) {
  return guard([&] {
    std::string s;
    for (const auto& n : D(ds)->names()) { if (!s.empty()) s += "\n"; s += n; }
    *needed = (int64_t)s.size() + 1;
    if (out) {
      if (cap < *needed) throw std::runtime_error("buffer too small");
      std::strcpy(out, s.c_str());
    }
  });
}

int lga_dataset_bin_info(LgaDataset ds, int real_feature, int* num_bin, int* missing_bin, double* upper, int cap,
                         int* n_upper) {
  return guard([&] {
    const auto& ms = D(ds)->mappers();
    if (real_feature < 0 || real_feature >= (int)ms.size()) throw std::runtime_error("feature index out of range");
    const BinMapper& m = ms[(size_t)real_feature];
    *num_bin = m.num_bin;
    *missing_bin = m.missing_bin;
    *n_upper = (int)m.upper.size();
    if (upper) {
      if (cap < *n_upper) throw std::runtime_error("buffer too small");
      std::memcpy(upper, m.upper.data(), m.upper.size() * sizeof(double));
    }
  });
}

int lga_dataset_get_bins(LgaDataset ds, uint8_t* fm_out, uint8_t* rt_out) {
  return guard([&] {
    Dataset* d = D(ds);
    engine_ensure_host_bins(d);
    if (!d->has_host_bins()) throw std::runtime_error("dataset has no host bins");
    size_t n = (size_t)d->root_rows();
    if (fm_out) std::memcpy(fm_out, d->bins_fm(), (size_t)d->num_features() * n);
    if (rt_out) std::memcpy(rt_out, d->bins_rt(), (size_t)d->num_tiles() * n * kTileFeatures);
  });
}

int lga_dataset_save_binary(LgaDataset ds, const char* path) {
  return guard([&] {
    Dataset* d = D(ds);
    engine_ensure_host_bins(d);
    d->root()->save_binary(path);
  });
}
int lga_dataset_load_binary(const char* path, LgaDataset* out) {
  return guard([&] { *out = new DatasetBox{Dataset::load_binary(path)}; });
}

int lga_dataset_free(LgaDataset ds) { return guard([&] { delete (DatasetBox*)ds; }); }

int lga_comm_create_callback(int rank, int size, lga_allreduce_fn ar, lga_allgather_fn ag, LgaComm* out) {
  return guard([&] { *out = new CallbackComm(rank, size, ar, ag); });
}
int lga_comm_free(LgaComm c) { return guard([&] { delete (Comm*)c; }); }

int lga_booster_create(LgaDataset train, const char* params, LgaComm comm, LgaBooster* out) {
  return guard([&] {
    Comm* c = (Comm*)comm;
    EngineFactory factory = [c](std::shared_ptr<Dataset> ds, const Config& cfg) { return make_engine(ds, cfg, c); };
    auto box = new BoosterBox();
    try {
      box->b.reset(new Booster(DS(train), params ? params : "", factory));
    } catch (...) { delete box; throw; }
    *out = box;
  });
}

int lga_booster_from_string(const char* model, LgaBooster* out) {
  return guard([&] {
    auto box = new BoosterBox();
    try { box->b.reset(new Booster(std::string(model))); } catch (...) { delete box; throw; }
    *out = box;
  });
}

int lga_booster_free(LgaBooster b) { return guard([&] { delete (BoosterBox*)b; }); }
int lga_booster_add_valid(LgaBooster b, LgaDataset ds, const char* name) {
  return guard([&] { B(b)->add_valid(DS(ds), name ? name : "valid"); });
}
int lga_booster_update(LgaBooster b, int* finished) { return guard([&] { *finished = B(b)->update_one_iter() ? 1 : 0; }); }
// n_iters iterations of n boosters in lockstep (the folds of a cross-validation, the rows of a parameter grid):
// which == 0 evaluates every booster's first validation set after every iteration, which < 0 nothing.  values:
// [iteration][booster][metric] with `stride` doubles per booster.  The loop ends early after an iteration in which
// NO booster found anything to split; finished[iteration * n + b] = booster b found nothing to split in that
// iteration, *iters_done = iterations run.  (Booster::update_group_chunk: on the device engine small boosters run the whole
// chunk without the host in between.)
int lga_booster_group_update(LgaBooster* boosters, int n, int n_iters, int which, double* values, int stride,
                             int* finished, int* iters_done) {
  return guard([&] {
    if (which > 0) throw std::runtime_error("group updates evaluate validation set 0 or nothing");
    std::vector<Booster*> bs((size_t)n);
    for (int i = 0; i < n; ++i) bs[(size_t)i] = B(boosters[i]);
    *iters_done = Booster::update_group_chunk(bs.data(), n, n_iters, which == 0, values, stride, finished);
  });
}
int lga_booster_update_custom(LgaBooster b, const float* grad, const float* hess, int64_t n, int* finished) {
  return guard([&] { *finished = B(b)->update_one_iter_custom(grad, hess, n) ? 1 : 0; });
}
int lga_booster_rollback(LgaBooster b) { return guard([&] { B(b)->rollback_one_iter(); }); }
int lga_booster_refit(LgaBooster b, const void* data, int dtype, int64_t nrow, int ncol, int row_major, const float* label,
                      const float* weight, double decay, const char* params) {
  return guard([&] { B(b)->refit(data, dtype, nrow, ncol, row_major != 0, label, weight, decay, params ? params : ""); });
}
int lga_booster_set_leaf_value(LgaBooster b, int tree, int leaf, double value) {
  return guard([&] { B(b)->set_leaf_value(tree, leaf, value); });
}
int lga_booster_shuffle(LgaBooster b, int start_iter, int end_iter, int64_t seed) {
  return guard([&] { B(b)->shuffle_iterations(start_iter, end_iter, (uint64_t)seed); });
}
int lga_booster_num_trees(LgaBooster b, int* out) { return guard([&] { *out = B(b)->num_trees(); }); }
int lga_booster_num_class(LgaBooster b, int* out) { return guard([&] { *out = B(b)->num_class(); }); }
int lga_booster_num_feature(LgaBooster b, int* out) { return guard([&] { *out = B(b)->num_features(); }); }

int lga_booster_eval_names(LgaBooster b, char* out, int cap) {
  return guard([&] {
    std::string s;
    for (const auto& m : B(b)->eval_names()) {
      if (!s.empty()) s += ",";
      s += m.first + ":" + (m.second ? "1" : "0");
    }
    if ((int)s.size() + 1 > cap) throw std::runtime_error("buffer too small");
    std::strcpy(out, s.c_str());
  });
}

int lga_booster_eval(LgaBooster b, int which, double* out, int cap, int* n) {
  return guard([&] {
    auto r = B(b)->eval(which);
    *n = (int)r.size();
    if (cap < *n) throw std::runtime_error("buffer too small");
    for (size_t i = 0; i < r.size(); ++i) out[i] = r[i].value;
  });
}

int lga_booster_inner_nrows(LgaBooster b, int which, int64_t* out) { return guard([&] { *out = B(b)->inner_num_rows(which); }); }
int lga_booster_inner_predict(LgaBooster b, int which, int raw, double* out) {
  return guard([&] { B(b)->inner_predict(which, raw != 0, out); });
}
int lga_booster_get_gradients(LgaBooster b, float* g, float* h) { return guard([&] { B(b)->get_gradients(g, h); }); }

int lga_booster_predict(LgaBooster b, const void* data, int dtype, int64_t nrow, int ncol, int row_major,
                        int start_iter, int num_iter, int type, double* out) {
  return guard([&] { B(b)->predict(data, dtype, nrow, ncol, row_major != 0, start_iter, num_iter, type, out); });
}
int lga_booster_used_iterations(LgaBooster b, int start_iter, int num_iter, int* out) {
  return guard([&] { *out = B(b)->used_iterations(start_iter, num_iter); });
}

int lga_booster_save_string(LgaBooster b, int start_iter, int num_iter, int importance_type, const char** out,
                            int64_t* len) {
  return guard([&] {
    auto box = (BoosterBox*)b;
    box->scratch = B(b)->save_model_to_string(start_iter, num_iter, importance_type);
    *out = box->scratch.c_str();
    *len = (int64_t)box->scratch.size();
  });
}
int lga_booster_dump_json(LgaBooster b, int start_iter, int num_iter, const char** out, int64_t* len) {
  return guard([&] {
    auto box = (BoosterBox*)b;
    box->scratch = B(b)->dump_json(start_iter, num_iter);
    *out = box->scratch.c_str();
    *len = (int64_t)box->scratch.size();
  });
}
int lga_booster_feature_importance(LgaBooster b, int num_iter, int type, double* out) {
  return guard([&] {
    auto v = B(b)->feature_importance(num_iter, type);
    std::memcpy(out, v.data(), v.size() * sizeof(double));
  });
}
int lga_booster_reset_parameter(LgaBooster b, const char* params) { return guard([&] { B(b)->reset_parameter(params); }); }
int lga_booster_prepend(LgaBooster b, LgaBooster first) { return guard([&] { B(b)->prepend_trees(*B(first)); }); }
int lga_booster_stats(LgaBooster b, int* rounds, int* materialized, int* used) {
  return guard([&] {
    GrowStats s = B(b)->last_grow_stats();
    *rounds = s.rounds; *materialized = s.materialized_splits; *used = s.used_splits;
  });
}
int lga_booster_profile_json(LgaBooster b, const char** out, int64_t* len) {
  return guard([&] {
    auto box = (BoosterBox*)b;
    box->scratch = B(b)->last_profile();
    *out = box->scratch.c_str();
    *len = (int64_t)box->scratch.size();
  });
}
int lga_booster_info(LgaBooster b, const char** objective, int* average_output, double* sigmoid) {
  return guard([&] {
    *objective = B(b)->objective_string().c_str();
    *average_output = B(b)->average_output() ? 1 : 0;
    *sigmoid = B(b)->objective_spec().sigmoid;
  });
}

}  // extern "C"

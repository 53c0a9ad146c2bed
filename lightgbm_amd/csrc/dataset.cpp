#include "dataset.hpp"
#include "binning_core.hpp"
#include <algorithm>
#include <stdexcept>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace lga {

namespace {
inline double load_value(const void* data, int dtype, int64_t nrow, int ncol, bool row_major, int64_t r, int c) {
  int64_t off = row_major ? r * (int64_t)ncol + c : (int64_t)c * nrow + r;
  return dtype == 0 ? (double)((const float*)data)[off] : ((const double*)data)[off];
}
}  // namespace

void Dataset::finish_mappers() {
  used_feature_.clear();
  inner_of_real_.assign(mappers_.size(), -1);
  for (size_t f = 0; f < mappers_.size(); ++f) {
    if (!mappers_[f].trivial) {
      inner_of_real_[f] = (int32_t)used_feature_.size();
      used_feature_.push_back((int32_t)f);
    }
  }
  if (feature_names.size() != mappers_.size()) {
    feature_names.resize(mappers_.size());
    for (size_t f = 0; f < mappers_.size(); ++f) feature_names[f] = "Column_" + std::to_string(f);
  }
}

namespace {
// forcedbins_filename: a JSON list of {"feature": <column>, "bin_upper_bound": [<ascending numbers>]} objects
// (upstream LightGBM: examples/regression/forced_bins.json).  Returns one list of bounds per column.
std::vector<std::vector<double>> load_forced_bins(const std::string& path, int ncol) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("forcedbins_filename: cannot read " + path);
  std::stringstream buf;
  buf << in.rdbuf();
  const std::string t = buf.str();
  size_t i = 0;
  auto bad = [&](const char* what) { throw std::runtime_error(std::string("forcedbins_filename: ") + what + " near offset " + std::to_string(i)); };
  auto ws = [&]() { while (i < t.size() && (t[i] == ' ' || t[i] == '\n' || t[i] == '\t' || t[i] == '\r')) ++i; };
  auto eat = [&](char c) { ws(); if (i < t.size() && t[i] == c) { ++i; return true; } return false; };
  auto number = [&]() {
    ws();
    char* end = nullptr;
    const double v = std::strtod(t.c_str() + i, &end);
    if (end == t.c_str() + i) bad("expected a number");
    i = (size_t)(end - t.c_str());
    return v;
  };
  std::vector<std::vector<double>> out((size_t)ncol);
  if (!eat('[')) bad("expected [");
  if (!eat(']')) {
    do {
      if (!eat('{')) bad("expected {");
      int feature = -1;
      std::vector<double> bounds;
      do {
        ws();
        if (i >= t.size() || t[i] != '"') bad("expected a quoted key");
        const size_t e = t.find('"', i + 1);
        if (e == std::string::npos) bad("unterminated string");
        const std::string key = t.substr(i + 1, e - i - 1);
        i = e + 1;
        if (!eat(':')) bad("expected :");
        if (key == "feature") feature = (int)number();
        else if (key == "bin_upper_bound") {
          if (!eat('[')) bad("expected [");
          if (!eat(']')) {
            do bounds.push_back(number()); while (eat(','));
            if (!eat(']')) bad("expected ]");
          }
        } else bad("unknown key (feature, bin_upper_bound)");
      } while (eat(','));
      if (!eat('}')) bad("expected }");
      if (feature < 0 || feature >= ncol) throw std::runtime_error("forcedbins_filename: feature " + std::to_string(feature) + " does not exist");
      std::sort(bounds.begin(), bounds.end());
      out[(size_t)feature] = bounds;
    } while (eat(','));
    if (!eat(']')) bad("expected ]");
  }
  ws();
  if (i != t.size()) bad("trailing text");
  return out;
}
}  // namespace

std::shared_ptr<Dataset> Dataset::from_sample(const double* sample, int64_t nsample, int ncol, int64_t nrow,
                                              const Config& cfg) {
  auto ds = std::make_shared<Dataset>();
  ds->root_rows_ = nrow;
  ds->mappers_.resize(ncol);
  if (!cfg.max_bin_by_feature.empty() && (int)cfg.max_bin_by_feature.size() != ncol)
    throw std::runtime_error("max_bin_by_feature lists " + std::to_string(cfg.max_bin_by_feature.size()) +
                             " values, the data has " + std::to_string(ncol) + " columns");
  std::vector<std::vector<double>> forced;
  if (!cfg.forcedbins_filename.empty()) forced = load_forced_bins(cfg.forcedbins_filename, ncol);
#pragma omp parallel for schedule(dynamic)
  for (int f = 0; f < ncol; ++f) {
    std::vector<double> col((size_t)nsample);
    for (int64_t i = 0; i < nsample; ++i) col[(size_t)i] = sample[i * ncol + f];
    const bool cat = std::find(cfg.categorical_feature.begin(), cfg.categorical_feature.end(), f) != cfg.categorical_feature.end();
    const int bins = cfg.max_bin_by_feature.empty() ? cfg.max_bin : cfg.max_bin_by_feature[(size_t)f];
    if (cat) ds->mappers_[f].find_bins_categorical(col, bins, cfg.min_data_in_bin);
    else ds->mappers_[f].find_bins(col, bins, cfg.min_data_in_bin, cfg.use_missing, cfg.zero_as_missing,
                                   forced.empty() ? nullptr : &forced[(size_t)f]);
  }
  ds->finish_mappers();
  return ds;
}

std::vector<int64_t> Dataset::sample_rows(int64_t nrow, const Config& cfg) {
  int64_t want = cfg.bin_construct_sample_cnt;
  std::vector<int64_t> rows;
  if (nrow <= want) {
    rows.resize((size_t)nrow);
    for (int64_t i = 0; i < nrow; ++i) rows[(size_t)i] = i;
  } else {
    float p = (float)((double)want / (double)nrow);
    rows.reserve((size_t)(want + want / 8));
    for (int64_t i = 0; i < nrow; ++i)
      if (rng_uniform((uint64_t)cfg.data_random_seed, 0x5a17u, (uint64_t)i) < p) rows.push_back(i);
  }
  return rows;
}

std::shared_ptr<Dataset> Dataset::from_matrix(const void* data, int dtype, int64_t nrow, int ncol, bool row_major,
                                              const Config& cfg, const Dataset* reference, bool bin_on_host) {
  if (nrow <= 0 || ncol <= 0) throw std::runtime_error("Dataset: empty matrix");
  if (nrow > 2147483000LL) throw std::runtime_error("Dataset: more than 2^31 rows per device is not supported");
  std::shared_ptr<Dataset> ds;
  if (reference) {
    const Dataset* r = reference->root();
    if (r->num_total_features() != ncol)
      throw std::runtime_error("Dataset: reference has a different number of features");
    ds = std::make_shared<Dataset>();
    ds->root_rows_ = nrow;
    ds->mappers_ = r->mappers_;
    ds->feature_names = r->feature_names;
    ds->finish_mappers();
  } else {
    // row sample for bin finding
    std::vector<int64_t> rows = sample_rows(nrow, cfg);
    int64_t ns = (int64_t)rows.size();
    std::vector<double> sample((size_t)ns * ncol);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < ns; ++i)
      for (int c = 0; c < ncol; ++c)
        sample[(size_t)(i * ncol + c)] = load_value(data, dtype, nrow, ncol, row_major, rows[(size_t)i], c);
    ds = from_sample(sample.data(), ns, ncol, nrow, cfg);
  }
  if (bin_on_host) {
    int F = ds->num_features();
    int T = ds->num_tiles();
    std::vector<uint8_t> rt((size_t)T * nrow * kTileFeatures, 0), fm((size_t)F * nrow, 0);
    ds->bin_rows(data, dtype, nrow, ncol, row_major, rt.data(), fm.data(), nrow, 0);
    ds->adopt_host_bins(std::move(rt), std::move(fm));
  }
  return ds;
}

std::shared_ptr<Dataset> Dataset::from_reference(const Dataset* ref, int64_t nrow) {
  const Dataset* r = ref->root();
  auto ds = std::make_shared<Dataset>();
  ds->root_rows_ = nrow;
  ds->mappers_ = r->mappers_;
  ds->feature_names = r->feature_names;
  ds->finish_mappers();
  return ds;
}

std::shared_ptr<Dataset> Dataset::from_mappers(std::vector<BinMapper>&& mappers, int64_t nrow) {
  auto ds = std::make_shared<Dataset>();
  ds->root_rows_ = nrow;
  ds->mappers_ = std::move(mappers);
  ds->finish_mappers();
  return ds;
}

void Dataset::bin_rows(const void* data, int dtype, int64_t nrow, int ncol, bool row_major, uint8_t* rt, uint8_t* fm,
                       int64_t total_rows, int64_t row_offset) const {
  const Dataset* r = root();
  int F = r->num_features();
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < nrow; ++i) {
    int64_t row = row_offset + i;
    for (int f = 0; f < F; ++f) {
      const BinMapper& m = r->mappers_[r->used_feature_[f]];
      double v = load_value(data, dtype, nrow, ncol, row_major, i, r->used_feature_[f]);
      uint8_t b = (uint8_t)m.value_to_bin(v);
      int t = f / kTileFeatures, k = f % kTileFeatures;
      rt[((size_t)t * total_rows + row) * kTileFeatures + k] = b;
      fm[(size_t)f * total_rows + row] = b;
    }
  }
}

void Dataset::adopt_host_bins(std::vector<uint8_t>&& rt, std::vector<uint8_t>&& fm) {
  if (parent_) throw std::runtime_error("adopt_host_bins on a subset");
  bins_rt_ = std::move(rt);
  bins_fm_ = std::move(fm);
}

std::shared_ptr<Dataset> Dataset::subset(const int32_t* indices, int64_t n) const {
  auto ds = std::make_shared<Dataset>();
  std::shared_ptr<const Dataset> rootp = parent_ ? parent_ : shared_from_this();
  ds->parent_ = rootp;
  ds->root_rows_ = rootp->root_rows_;
  auto idx = std::make_shared<std::vector<int32_t>>((size_t)n);
  int64_t local_n = num_data();
  for (int64_t i = 0; i < n; ++i) {
    int32_t v = indices[i];
    if (v < 0 || v >= local_n) throw std::runtime_error("subset: index out of range");
    (*idx)[(size_t)i] = used_indices_ ? (*used_indices_)[(size_t)v] : v;
  }
  // the learners keep row lists ascending (stable partitions preserve it)
  std::sort(idx->begin(), idx->end());
  if (std::adjacent_find(idx->begin(), idx->end()) != idx->end())
    throw std::runtime_error("subset: duplicate row index");
  ds->used_indices_ = idx;
  return ds;
}

std::vector<FeatureMeta> Dataset::feature_meta() const {
  int F = num_features();
  std::vector<FeatureMeta> m((size_t)F);
  for (int f = 0; f < F; ++f) {
    const BinMapper& bm = inner_mapper(f);
    m[(size_t)f].num_bin = bm.num_bin;
    m[(size_t)f].missing_bin = bm.missing_bin;
    m[(size_t)f].enabled = 1;
    m[(size_t)f].monotone = 0;
    m[(size_t)f].categorical = bm.categorical ? 1 : 0;
  }
  return m;
}

namespace {
template <typename T>
void scatter_field(std::vector<T>& dst, int64_t root_rows, const std::vector<int32_t>* used, const T* v, int64_t n,
                   int64_t local_n, const char* what) {
  if (n == 0) { if (!used) dst.clear(); return; }
  if (n != local_n) throw std::runtime_error(std::string("length of ") + what + " differs from the number of rows");
  if (!used) { dst.assign(v, v + n); return; }
  if (dst.empty()) dst.assign((size_t)root_rows, (T)0);
  for (int64_t i = 0; i < n; ++i) dst[(size_t)(*used)[(size_t)i]] = v[i];
}
}  // namespace

void Dataset::set_label(const float* v, int64_t n) {
  scatter_field(root_mut()->label_, root_rows_, used_indices_.get(), v, n, num_data(), "label");
}
void Dataset::set_weight(const float* v, int64_t n) {
  for (int64_t i = 0; i < n; ++i)
    if (!(v[i] >= 0.0f)) throw std::runtime_error("weights must be non-negative");
  scatter_field(root_mut()->weight_, root_rows_, used_indices_.get(), v, n, num_data(), "weight");
}
// n == num_data(), or num_class * num_data() values laid out [class][row] (multiclass); the root then holds
// [class][root row].
void Dataset::set_init_score(const double* v, int64_t n) {
  const int64_t local = num_data();
  if (n == 0 || n == local || local == 0 || n % local != 0) {
    scatter_field(root_mut()->init_score_, root_rows_, used_indices_.get(), v, n, local, "init_score");
    return;
  }
  const int64_t K = n / local;
  std::vector<double>& dst = root_mut()->init_score_;
  if ((int64_t)dst.size() != K * root_rows_) dst.assign((size_t)(K * root_rows_), 0.0);
  for (int64_t k = 0; k < K; ++k)
    for (int64_t i = 0; i < local; ++i) {
      const int64_t r = used_indices_ ? (int64_t)(*used_indices_)[(size_t)i] : i;
      dst[(size_t)(k * root_rows_ + r)] = v[k * local + i];
    }
}

void Dataset::set_group(const int32_t* sizes, int64_t n) {
  query_bounds_.clear();
  if (n == 0) return;
  std::vector<int64_t> b((size_t)n + 1, 0);
  for (int64_t q = 0; q < n; ++q) {
    if (sizes[q] < 0) throw std::runtime_error("group sizes must be non-negative");
    b[(size_t)q + 1] = b[(size_t)q] + sizes[q];
  }
  if (b.back() != num_data())
    throw std::runtime_error("group sizes add up to " + std::to_string(b.back()) + ", the data has " +
                             std::to_string(num_data()) + " rows");
  query_bounds_ = std::move(b);
}

std::vector<float> Dataset::local_label() const {
  const auto& l = root_label();
  if (!used_indices_) return l;
  std::vector<float> out(used_indices_->size());
  if (l.empty()) { out.clear(); return out; }
  for (size_t i = 0; i < out.size(); ++i) out[i] = l[(size_t)(*used_indices_)[i]];
  return out;
}
std::vector<float> Dataset::local_weight() const {
  const auto& l = root_weight();
  if (!used_indices_ || l.empty()) return l;
  std::vector<float> out(used_indices_->size());
  for (size_t i = 0; i < out.size(); ++i) out[i] = l[(size_t)(*used_indices_)[i]];
  return out;
}

namespace {
constexpr char kBinaryMagic[8] = {'L', 'G', 'A', 'B', 'I', 'N', '0', '1'};
struct Writer {
  FILE* f;
  void raw(const void* p, size_t n) { if (n && std::fwrite(p, 1, n, f) != n) throw std::runtime_error("dataset cache: write failed"); }
  template <typename T> void pod(const T& v) { raw(&v, sizeof(T)); }
  template <typename T> void vec(const std::vector<T>& v) { pod<uint64_t>(v.size()); raw(v.data(), v.size() * sizeof(T)); }
  void str(const std::string& s) { pod<uint64_t>(s.size()); raw(s.data(), s.size()); }
};
struct Reader {
  FILE* f;
  void raw(void* p, size_t n) { if (n && std::fread(p, 1, n, f) != n) throw std::runtime_error("dataset cache: file is truncated"); }
  template <typename T> T pod() { T v; raw(&v, sizeof(T)); return v; }
  template <typename T> void vec(std::vector<T>& v, uint64_t limit) {
    uint64_t n = pod<uint64_t>();
    if (n > limit) throw std::runtime_error("dataset cache: implausible array length");
    v.resize((size_t)n);
    raw(v.data(), (size_t)n * sizeof(T));
  }
  std::string str() {
    uint64_t n = pod<uint64_t>();
    if (n > (1u << 20)) throw std::runtime_error("dataset cache: implausible string length");
    std::string s((size_t)n, '\0');
    raw(&s[0], (size_t)n);
    return s;
  }
};
struct FileCloser { FILE* f; ~FileCloser() { if (f) std::fclose(f); } };
}  // namespace

void Dataset::save_binary(const std::string& path) const {
  if (parent_) throw std::runtime_error("save_binary: save the root dataset, not a subset");
  if (!has_host_bins()) throw std::runtime_error("save_binary: the dataset has no host copy of its bins");
  const std::string tmp = path + ".tmp";
  {
    FileCloser fc{std::fopen(tmp.c_str(), "wb")};
    if (!fc.f) throw std::runtime_error("save_binary: cannot open " + tmp);
    Writer w{fc.f};
    w.raw(kBinaryMagic, sizeof(kBinaryMagic));
    w.pod<int64_t>(root_rows_);
    w.pod<int32_t>((int32_t)mappers_.size());
    for (size_t i = 0; i < mappers_.size(); ++i) {
      const BinMapper& m = mappers_[i];
      w.vec(m.upper);
      w.pod<int32_t>(m.num_bin);
      w.pod<int32_t>((int32_t)m.missing);
      w.pod<int32_t>(m.missing_bin);
      w.pod<double>(m.min_val);
      w.pod<double>(m.max_val);
      w.pod<int32_t>((m.trivial ? 1 : 0) | (m.categorical ? 2 : 0));
      w.str(feature_names[i]);
    }
    w.vec(bins_fm_);
    w.vec(label_);
    w.vec(weight_);
    w.vec(init_score_);
    w.vec(query_bounds_);   // trailing section (files written before it existed simply end here)
    if (std::fflush(fc.f) != 0) throw std::runtime_error("save_binary: flush failed");
  }
  if (std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("save_binary: cannot move the file into place");
}

std::shared_ptr<Dataset> Dataset::load_binary(const std::string& path) {
  FileCloser fc{std::fopen(path.c_str(), "rb")};
  if (!fc.f) throw std::runtime_error("load_binary: cannot open " + path);
  Reader r{fc.f};
  char magic[8];
  r.raw(magic, sizeof(magic));
  if (std::memcmp(magic, kBinaryMagic, sizeof(magic)) != 0) throw std::runtime_error("load_binary: not a dataset cache file");
  auto ds = std::make_shared<Dataset>();
  ds->root_rows_ = r.pod<int64_t>();
  if (ds->root_rows_ <= 0 || ds->root_rows_ > 2147483000LL) throw std::runtime_error("load_binary: bad row count");
  int32_t nf = r.pod<int32_t>();
  if (nf <= 0 || nf > (1 << 24)) throw std::runtime_error("load_binary: bad feature count");
  ds->mappers_.resize((size_t)nf);
  ds->feature_names.resize((size_t)nf);
  for (int32_t i = 0; i < nf; ++i) {
    BinMapper& m = ds->mappers_[(size_t)i];
    r.vec(m.upper, kMaxBins + 1);
    m.num_bin = r.pod<int32_t>();
    m.missing = (MissingType)r.pod<int32_t>();
    m.missing_bin = r.pod<int32_t>();
    m.min_val = r.pod<double>();
    m.max_val = r.pod<double>();
    const int32_t flags = r.pod<int32_t>();
    m.trivial = (flags & 1) != 0;
    m.categorical = (flags & 2) != 0;
    m.build_cat_lookup();
    if (m.num_bin < 1 || m.num_bin > kMaxBins || m.upper.empty()) throw std::runtime_error("load_binary: bad bin mapper");
    ds->feature_names[(size_t)i] = r.str();
  }
  ds->finish_mappers();
  const uint64_t n = (uint64_t)ds->root_rows_;
  const int F = (int)ds->used_feature_.size();
  r.vec(ds->bins_fm_, (uint64_t)F * n);
  if (ds->bins_fm_.size() != (size_t)F * n) throw std::runtime_error("load_binary: bin matrix has the wrong size");
  r.vec(ds->label_, n);
  if (ds->label_.size() != n && !ds->label_.empty()) throw std::runtime_error("load_binary: label has the wrong length");
  r.vec(ds->weight_, n);
  if (!ds->weight_.empty() && ds->weight_.size() != n) throw std::runtime_error("load_binary: weight has the wrong length");
  // one initial score per row, or one per class and row ([class][row]; at most 4096 classes)
  r.vec(ds->init_score_, n * 4096);
  if (ds->init_score_.size() % n != 0) throw std::runtime_error("load_binary: init_score has the wrong length");
  {
    const int c = std::fgetc(r.f);
    if (c != EOF) {
      std::ungetc(c, r.f);
      r.vec(ds->query_bounds_, n + 1);
      if (!ds->query_bounds_.empty() && (ds->query_bounds_.front() != 0 || ds->query_bounds_.back() != (int64_t)n))
        throw std::runtime_error("load_binary: query boundaries do not cover the rows");
      for (size_t q = 1; q < ds->query_bounds_.size(); ++q)
        if (ds->query_bounds_[q] < ds->query_bounds_[q - 1]) throw std::runtime_error("load_binary: query boundaries decrease");
    }
  }
  // a truncated or damaged cache must not reach the histogram kernels with bins outside a feature's range
  for (int f = 0; f < F; ++f) {
    const int nb = ds->mappers_[(size_t)ds->used_feature_[(size_t)f]].num_bin;
    const uint8_t* col = ds->bins_fm_.data() + (size_t)f * n;
    int worst = 0;
#pragma omp parallel for reduction(max : worst) schedule(static)
    for (int64_t row = 0; row < (int64_t)n; ++row) worst = std::max(worst, (int)col[row]);
    if (worst >= nb) throw std::runtime_error("load_binary: a bin value lies outside its feature's range");
  }
  // the row-tile-major copy is derived
  const int T = ds->num_tiles();
  ds->bins_rt_.assign((size_t)T * n * kTileFeatures, 0);
#pragma omp parallel for schedule(static)
  for (int64_t row = 0; row < (int64_t)n; ++row)
    for (int f = 0; f < F; ++f)
      ds->bins_rt_[((size_t)(f / kTileFeatures) * n + (size_t)row) * kTileFeatures + (size_t)(f % kTileFeatures)] =
          ds->bins_fm_[(size_t)f * n + (size_t)row];
  return ds;
}

void Dataset::set_feature_names(const std::vector<std::string>& names) {
  if ((int)names.size() != num_total_features()) throw std::runtime_error("feature_name length mismatch");
  root_mut()->feature_names = names;
}

}  // namespace lga

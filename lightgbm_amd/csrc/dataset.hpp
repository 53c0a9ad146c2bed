// Binned dataset: the engine-side object behind `lgb.Dataset(matrix, label=)`
// (r/gridsearchCV.R:52-53).  Bins are kept twice:
//   * row-tile-major  [tile][row][32]  one 32-byte record per (row, 32-feature tile) -
//     a histogram lane reads its row's bins with two 16-byte loads;
//   * feature-major   [feature][row]   a partition reads one byte per row of one feature.
// CV folds are zero-copy subsets: a child dataset holds a sorted row-index
// list into its root, and shares the root's bin mappers and bin storage.
#pragma once
#include "bin_mapper.hpp"
#include "common.hpp"
#include "config.hpp"
#include <memory>
#include <string>
#include <vector>

namespace lga {

struct DeviceMirror;  // defined by the HIP build only

class Dataset : public std::enable_shared_from_this<Dataset> {
 public:
  // ---- construction ----
  // Dense matrix -> binned dataset.  dtype: 0 float32, 1 float64.
  // reference != nullptr: reuse its bin mappers (validation data).
  // bin_on_host=false leaves the host bin matrices empty (the device build fills its own).
  static std::shared_ptr<Dataset> from_matrix(const void* data, int dtype, int64_t nrow, int ncol, bool row_major,
                                              const Config& cfg, const Dataset* reference, bool bin_on_host);
  // Mappers only (used by the device path and by data-parallel boundary agreement):
  // finds bins from `sample` (row-major float64, nsample x ncol).
  static std::shared_ptr<Dataset> from_sample(const double* sample, int64_t nsample, int ncol, int64_t nrow,
                                              const Config& cfg);
  // Mapper-only dataset with `nrow` rows that shares `ref`'s bin boundaries (bins are filled later).
  static std::shared_ptr<Dataset> from_reference(const Dataset* ref, int64_t nrow);
  // Rows whose values the bin boundaries are derived from: all rows up to bin_construct_sample_cnt,
  // otherwise a seeded Bernoulli sample of about that many (ascending).
  static std::vector<int64_t> sample_rows(int64_t nrow, const Config& cfg);
  // Mapper-only dataset from bin boundaries found elsewhere (the device bin finder).
  static std::shared_ptr<Dataset> from_mappers(std::vector<BinMapper>&& mappers, int64_t nrow);
  std::shared_ptr<Dataset> subset(const int32_t* indices, int64_t n) const;
  // Binary cache of a ROOT dataset: bin boundaries, feature names, the feature-major bin matrix and the
  // label / weight / init-score vectors and the query boundaries.  Loading skips sampling, bin finding and binning.
  void save_binary(const std::string& path) const;
  static std::shared_ptr<Dataset> load_binary(const std::string& path);

  // ---- shape ----
  int64_t num_data() const { return used_indices_ ? (int64_t)used_indices_->size() : root_rows_; }
  int64_t root_rows() const { return root_rows_; }
  int num_total_features() const { return (int)mappers().size(); }
  int num_features() const { return (int)root()->used_feature_.size(); }   // non-trivial features
  int num_tiles() const { return (num_features() + kTileFeatures - 1) / kTileFeatures; }
  const Dataset* root() const { return parent_ ? parent_.get() : this; }
  Dataset* root_mut() { return parent_ ? const_cast<Dataset*>(parent_.get()) : this; }
  bool is_subset() const { return (bool)parent_; }
  const std::vector<int32_t>* used_indices() const { return used_indices_.get(); }

  const std::vector<BinMapper>& mappers() const { return root()->mappers_; }
  int real_feature(int inner) const { return root()->used_feature_[inner]; }
  int inner_feature(int real) const { return root()->inner_of_real_[real]; }
  const BinMapper& inner_mapper(int inner) const { return mappers()[real_feature(inner)]; }
  std::vector<FeatureMeta> feature_meta() const;

  // ---- host bins (root only) ----
  const uint8_t* bins_rt() const { return root()->bins_rt_.data(); }   // [tiles][rows][32]
  const uint8_t* bins_fm() const { return root()->bins_fm_.data(); }   // [features][rows]
  bool has_host_bins() const { return !root()->bins_fm_.empty() || num_features() == 0; }
  // Bin a block of raw rows into caller-provided storage (used by the host path and by tests).
  void bin_rows(const void* data, int dtype, int64_t nrow, int ncol, bool row_major, uint8_t* rt, uint8_t* fm,
                int64_t total_rows, int64_t row_offset) const;
  void adopt_host_bins(std::vector<uint8_t>&& rt, std::vector<uint8_t>&& fm);

  // ---- metadata (indexed by LOCAL row of this dataset; subsets read through to the root) ----
  void set_label(const float* v, int64_t n);
  void set_weight(const float* v, int64_t n);      // n == 0 clears
  void set_init_score(const double* v, int64_t n); // n == 0 clears; num_class * num_data values are [class][row]
  // Root-indexed storage (length root_rows; init_score: root_rows per class); empty when unset.
  const std::vector<float>& root_label() const { return root()->label_; }
  const std::vector<float>& root_weight() const { return root()->weight_; }
  const std::vector<double>& root_init_score() const { return root()->init_score_; }
  std::vector<float> local_label() const;
  std::vector<float> local_weight() const;
  // Ranking: sizes of the queries, in row order; they must add up to num_data().  Kept per dataset (a
  // subset carries its own).  n == 0 clears.  query_boundaries() holds num_queries + 1 LOCAL row offsets.
  void set_group(const int32_t* sizes, int64_t n);
  const std::vector<int64_t>& query_boundaries() const { return query_bounds_; }

  std::vector<std::string> feature_names;   // per real feature
  void set_feature_names(const std::vector<std::string>& names);
  const std::vector<std::string>& names() const { return root()->feature_names; }

  // Device-side mirror of the bin matrices (HIP build); owned by the root.
  mutable std::shared_ptr<DeviceMirror> device;

 private:
  int64_t root_rows_ = 0;
  std::vector<BinMapper> mappers_;           // per real feature
  std::vector<int32_t> used_feature_;        // inner -> real
  std::vector<int32_t> inner_of_real_;       // real -> inner or -1
  std::vector<uint8_t> bins_rt_, bins_fm_;
  std::vector<float> label_, weight_;
  std::vector<double> init_score_;
  std::vector<int64_t> query_bounds_;         // of THIS dataset's rows (not read through to the root)
  std::shared_ptr<const Dataset> parent_;
  std::shared_ptr<std::vector<int32_t>> used_indices_;

  void finish_mappers();
};

}  // namespace lga

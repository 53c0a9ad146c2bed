// C ABI of the native core (loaded from Python through ctypes).  Every call
// returns 0 on success and -1 on failure; lga_last_error() holds the message.
#pragma once
#include <cstdint>

extern "C" {

typedef void* LgaDataset;
typedef void* LgaBooster;
typedef void* LgaComm;

const char* lga_last_error();
void lga_set_last_error(const char* msg);
const char* lga_build_info();          // "host" or "hip"
const char* lga_alias_table_json();
int lga_resolve_alias(const char* name, char* out, int out_len);

// ---- datasets ----
int lga_dataset_from_mat(const void* data, int dtype, int64_t nrow, int ncol, int row_major, const char* params,
                         LgaDataset reference, int bin_on_host, LgaDataset* out);
int lga_dataset_from_sample(const double* sample, int64_t nsample, int ncol, int64_t nrow, const char* params,
                            LgaDataset* out);
// Bins a raw host matrix into a dataset that only holds bin mappers so far (from_sample / reference).
int lga_dataset_bin_matrix(LgaDataset ds, const void* data, int dtype, int64_t nrow, int ncol, int row_major);
int lga_dataset_from_reference(LgaDataset ref, int64_t nrow, LgaDataset* out);
int lga_dataset_subset(LgaDataset ds, const int32_t* idx, int64_t n, LgaDataset* out);
int lga_dataset_set_field(LgaDataset ds, const char* name, const void* data, int64_t n, int dtype);
int lga_dataset_get_field(LgaDataset ds, const char* name, double* out, int64_t cap, int64_t* n);
int lga_dataset_num_data(LgaDataset ds, int64_t* out);
int lga_dataset_num_feature(LgaDataset ds, int* out);
int lga_dataset_num_used_feature(LgaDataset ds, int* out);
int lga_dataset_set_feature_names(LgaDataset ds, const char* joined);     // '\n' separated
int lga_dataset_get_feature_names(LgaDataset ds, char* out, int64_t cap, int64_t* needed);
int lga_dataset_bin_info(LgaDataset ds, int real_feature, int* num_bin, int* missing_bin, double* upper, int cap,
                         int* n_upper);
int lga_dataset_get_bins(LgaDataset ds, uint8_t* fm_out, uint8_t* rt_out);
int lga_dataset_save_binary(LgaDataset ds, const char* path);
int lga_dataset_load_binary(const char* path, LgaDataset* out);
int lga_dataset_free(LgaDataset ds);

// ---- communicator (callbacks implemented by the caller, e.g. torch.distributed/gloo) ----
typedef void (*lga_allreduce_fn)(void* buf, int64_t n, int dtype /*0 i64, 1 f64, 2 f32*/, int op /*0 sum, 1 max*/);
typedef void (*lga_allgather_fn)(const void* in, void* out, int64_t nbytes);
int lga_comm_create_callback(int rank, int size, lga_allreduce_fn ar, lga_allgather_fn ag, LgaComm* out);
int lga_comm_free(LgaComm c);

// ---- boosters ----
int lga_booster_create(LgaDataset train, const char* params, LgaComm comm, LgaBooster* out);
int lga_booster_from_string(const char* model, LgaBooster* out);
int lga_booster_free(LgaBooster b);
int lga_booster_add_valid(LgaBooster b, LgaDataset ds, const char* name);
int lga_booster_update(LgaBooster b, int* finished);
int lga_booster_group_update(LgaBooster* boosters, int n, int n_iters, int which, double* values, int stride,
                             int* finished, int* iters_done);
int lga_booster_update_custom(LgaBooster b, const float* grad, const float* hess, int64_t n, int* finished);
int lga_booster_rollback(LgaBooster b);
int lga_booster_set_leaf_value(LgaBooster b, int tree, int leaf, double value);
int lga_booster_shuffle(LgaBooster b, int start_iter, int end_iter, int64_t seed);
// Refits the leaf values of a (prediction-only) booster in place; weight may be null.
int lga_booster_refit(LgaBooster b, const void* data, int dtype, int64_t nrow, int ncol, int row_major, const float* label,
                      const float* weight, double decay, const char* params);
int lga_booster_num_trees(LgaBooster b, int* out);
int lga_booster_num_feature(LgaBooster b, int* out);
int lga_booster_eval_names(LgaBooster b, char* out, int cap);    // "l2:0,auc:1" (name:higher_better)
int lga_booster_eval(LgaBooster b, int which, double* out, int cap, int* n);
int lga_booster_num_class(LgaBooster b, int* out);   // trees per iteration
int lga_booster_inner_nrows(LgaBooster b, int which, int64_t* out);
int lga_booster_inner_predict(LgaBooster b, int which, int raw, double* out);
int lga_booster_get_gradients(LgaBooster b, float* g, float* h);
int lga_booster_predict(LgaBooster b, const void* data, int dtype, int64_t nrow, int ncol, int row_major,
                        int start_iter, int num_iter, int type, double* out);
int lga_booster_used_iterations(LgaBooster b, int start_iter, int num_iter, int* out);
int lga_booster_save_string(LgaBooster b, int start_iter, int num_iter, int importance_type, const char** out,
                            int64_t* len);
int lga_booster_dump_json(LgaBooster b, int start_iter, int num_iter, const char** out, int64_t* len);
int lga_booster_feature_importance(LgaBooster b, int num_iter, int type, double* out);
int lga_booster_reset_parameter(LgaBooster b, const char* params);
int lga_booster_prepend(LgaBooster b, LgaBooster first);
int lga_booster_stats(LgaBooster b, int* rounds, int* materialized, int* used);
int lga_booster_profile_json(LgaBooster b, const char** out, int64_t* len);
int lga_booster_info(LgaBooster b, const char** objective, int* average_output, double* sigmoid);
}

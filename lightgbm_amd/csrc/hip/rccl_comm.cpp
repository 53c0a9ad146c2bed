// RCCL communicator for the distributed learners.  librccl is bound at run
// time (dlopen of the copy already loaded by PyTorch, so one RCCL instance
// serves the process); every collective is enqueued on the learner's HIP
// stream, so there is no host synchronisation per reduction.  The unique id
// is created on rank 0 and distributed by the caller (torch.distributed).
#include "rccl_comm.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>
#include <stdexcept>
#include <string>

namespace lga {

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi& api() {
  static RcclApi a;
  if (a.handle) return a;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (a.handle) break;
  }
  if (!a.handle) throw std::runtime_error(std::string("cannot load librccl: ") + dlerror());
  auto sym = [&](const char* s) {
    void* p = dlsym(a.handle, s);
    if (!p) throw std::runtime_error(std::string("librccl lacks symbol ") + s);
    return p;
  };
  a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
  a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
  a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
  a.ReduceScatter = (decltype(a.ReduceScatter))sym("ncclReduceScatter");
  a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
  return a;
}

void check(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw std::runtime_error(std::string("RCCL ") + what + ": " + api().GetErrorString(r));
}

class RcclComm : public Comm {
 public:
  RcclComm(int rank, int size, const void* unique_id) : rank_(rank), size_(size) {
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    check(api().CommInitRank(&comm_, size, id, rank), "CommInitRank");
    scratch_.resize(1 << 16);
  }
  ~RcclComm() override { if (comm_) (void)api().CommDestroy(comm_); }
  int rank() const override { return rank_; }
  int size() const override { return size_; }

  void allreduce_sum_i64(int64_t* buf, size_t n, bool device, void* stream) override {
    reduce(buf, n, sizeof(int64_t), ncclInt64, ncclSum, device, (hipStream_t)stream);
  }
  void allreduce_sum_f64(double* buf, size_t n, bool device, void* stream) override {
    reduce(buf, n, sizeof(double), ncclFloat64, ncclSum, device, (hipStream_t)stream);
  }
  void allreduce_max_f32(float* buf, size_t n, bool device, void* stream) override {
    reduce(buf, n, sizeof(float), ncclFloat32, ncclMax, device, (hipStream_t)stream);
  }
  void reduce_scatter_sum_i64(int64_t* buf, size_t block, bool device, void* stream) override {
    if (block == 0) return;
    if (!device) { allreduce_sum_i64(buf, block * (size_t)size_, false, stream); return; }
    // in place: the receive buffer is this rank's block of the send buffer
    check(api().ReduceScatter(buf, buf + (size_t)rank_ * block, block, ncclInt64, ncclSum, comm_, (hipStream_t)stream),
          "ReduceScatter");
  }
  void allgather_bytes(const void* in, void* out, size_t nbytes, bool device, void* stream) override {
    hipStream_t s = (hipStream_t)stream;
    if (device) {
      check(api().AllGather(in, out, nbytes, ncclUint8, comm_, s), "AllGather");
      return;
    }
    size_t total = nbytes * (size_t)size_;
    scratch_.resize(nbytes + total);
    LGA_HIP_CHECK(hipMemcpyAsync(scratch_.get(), in, nbytes, hipMemcpyHostToDevice, s));
    check(api().AllGather(scratch_.get(), scratch_.get() + nbytes, nbytes, ncclUint8, comm_, s), "AllGather");
    LGA_HIP_CHECK(hipMemcpyAsync(out, scratch_.get() + nbytes, total, hipMemcpyDeviceToHost, s));
    LGA_HIP_CHECK(hipStreamSynchronize(s));
  }

 private:
  void reduce(void* buf, size_t n, size_t esz, ncclDataType_t dt, ncclRedOp_t op, bool device, hipStream_t s) {
    if (n == 0) return;
    if (device) {
      check(api().AllReduce(buf, buf, n, dt, op, comm_, s), "AllReduce");
      return;
    }
    // host buffer: stage through device scratch (small control-plane reductions only)
    scratch_.resize(n * esz);
    LGA_HIP_CHECK(hipMemcpyAsync(scratch_.get(), buf, n * esz, hipMemcpyHostToDevice, s));
    check(api().AllReduce(scratch_.get(), scratch_.get(), n, dt, op, comm_, s), "AllReduce");
    LGA_HIP_CHECK(hipMemcpyAsync(buf, scratch_.get(), n * esz, hipMemcpyDeviceToHost, s));
    LGA_HIP_CHECK(hipStreamSynchronize(s));
  }
  int rank_, size_;
  ncclComm_t comm_ = nullptr;
  DeviceBuffer<char> scratch_;
};

}  // namespace

void rccl_unique_id(void* out, int cap, int* len) {
  if (cap < (int)sizeof(ncclUniqueId)) throw std::runtime_error("unique id buffer too small");
  ncclUniqueId id;
  check(api().GetUniqueId(&id), "GetUniqueId");
  std::memcpy(out, &id, sizeof(id));
  *len = (int)sizeof(id);
}

Comm* make_rccl_comm(int rank, int size, const void* unique_id, int id_len) {
  if (id_len != (int)sizeof(ncclUniqueId)) throw std::runtime_error("bad RCCL unique id length");
  return new RcclComm(rank, size, unique_id);
}

}  // namespace lga

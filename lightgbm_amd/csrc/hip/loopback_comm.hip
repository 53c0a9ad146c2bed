// In-process loop-back communicator for DEVICE buffers: P learners run in P
// host threads of one process (all on the same GPU and stream) and reduce
// through staging buffers, in fixed rank order.  It exists so that the HIP
// data-parallel / feature-parallel code paths can be verified on a single
// GPU; production runs use RCCL (rccl_comm.cpp).
#include "rccl_comm.hpp"
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

namespace lga {

namespace {

struct Hub {
  int size;
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0;
  long generation = 0;
  std::vector<DeviceBuffer<char>> stage;
  std::vector<std::vector<char>> host;
  explicit Hub(int n) : size(n), stage((size_t)n), host((size_t)n) {}
  void barrier() {
    std::unique_lock<std::mutex> lk(m);
    long g = generation;
    if (++waiting == size) { waiting = 0; ++generation; cv.notify_all(); }
    else if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != g; }))
      throw std::runtime_error("loop-back communicator: a rank did not arrive within 120 s");
  }
};

template <typename T, bool MAX>
__global__ __launch_bounds__(256) void reduce_kernel(T* __restrict__ out, const T* const* __restrict__ parts, int P,
                                                     size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    T acc = parts[0][i];
    for (int r = 1; r < P; ++r) {
      T v = parts[r][i];
      acc = MAX ? (v > acc ? v : acc) : acc + v;
    }
    out[i] = acc;
  }
}

class LoopbackComm : public Comm {
 public:
  LoopbackComm(std::shared_ptr<Hub> hub, int rank) : hub_(hub), rank_(rank) {}
  int rank() const override { return rank_; }
  int size() const override { return hub_->size; }
  void allreduce_sum_i64(int64_t* b, size_t n, bool dev, void* s) override { reduce<int64_t, false>(b, n, dev, (hipStream_t)s); }
  void allreduce_sum_f64(double* b, size_t n, bool dev, void* s) override { reduce<double, false>(b, n, dev, (hipStream_t)s); }
  void allreduce_max_f32(float* b, size_t n, bool dev, void* s) override { reduce<float, true>(b, n, dev, (hipStream_t)s); }
  // A true reduce-scatter: only this rank's block is reduced, the others keep the local partial sums -
  // what RCCL leaves behind as well, so a learner that reads a block it does not own shows up in the tests.
  void reduce_scatter_sum_i64(int64_t* buf, size_t block, bool dev, void* stream) override {
    if (block == 0) return;
    if (!dev) { reduce<int64_t, false>(buf, block * (size_t)hub_->size, false, (hipStream_t)stream); return; }
    hipStream_t s = (hipStream_t)stream;
    const int P = hub_->size;
    const size_t total = block * (size_t)P;
    hub_->stage[(size_t)rank_].resize(total * sizeof(int64_t));
    LGA_HIP_CHECK(hipMemcpyAsync(hub_->stage[(size_t)rank_].get(), buf, total * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    LGA_HIP_CHECK(hipStreamSynchronize(s));
    hub_->barrier();
    std::vector<const int64_t*> parts((size_t)P);
    for (int r = 0; r < P; ++r) parts[(size_t)r] = (const int64_t*)hub_->stage[(size_t)r].get() + (size_t)rank_ * block;
    ptrs_.resize((size_t)P * sizeof(void*));
    LGA_HIP_CHECK(hipMemcpyAsync(ptrs_.get(), parts.data(), (size_t)P * sizeof(void*), hipMemcpyHostToDevice, s));
    int blocks = (int)std::min<size_t>((block + 255) / 256, 2048);
    hipLaunchKernelGGL((reduce_kernel<int64_t, false>), dim3(blocks), dim3(256), 0, s, buf + (size_t)rank_ * block,
                       (const int64_t* const*)ptrs_.get(), P, block);
    LGA_HIP_CHECK(hipGetLastError());
    LGA_HIP_CHECK(hipStreamSynchronize(s));
    hub_->barrier();
  }
  void allgather_bytes(const void* in, void* out, size_t nbytes, bool dev, void* stream) override {
    hipStream_t s = (hipStream_t)stream;
    if (dev) {
      hub_->stage[(size_t)rank_].resize(nbytes);
      LGA_HIP_CHECK(hipMemcpyAsync(hub_->stage[(size_t)rank_].get(), in, nbytes, hipMemcpyDeviceToDevice, s));
      LGA_HIP_CHECK(hipStreamSynchronize(s));
      hub_->barrier();
      for (int r = 0; r < hub_->size; ++r)
        LGA_HIP_CHECK(hipMemcpyAsync((char*)out + (size_t)r * nbytes, hub_->stage[(size_t)r].get(), nbytes, hipMemcpyDeviceToDevice, s));
      LGA_HIP_CHECK(hipStreamSynchronize(s));
      hub_->barrier();
    } else {
      hub_->host[(size_t)rank_].assign((const char*)in, (const char*)in + nbytes);
      hub_->barrier();
      for (int r = 0; r < hub_->size; ++r) std::memcpy((char*)out + (size_t)r * nbytes, hub_->host[(size_t)r].data(), nbytes);
      hub_->barrier();
    }
  }

 private:
  template <typename T, bool MAX>
  void reduce(T* buf, size_t n, bool dev, hipStream_t s) {
    if (n == 0) return;
    const int P = hub_->size;
    if (!dev) {
      hub_->host[(size_t)rank_].assign((const char*)buf, (const char*)buf + n * sizeof(T));
      hub_->barrier();
      for (size_t i = 0; i < n; ++i) {
        T acc = ((const T*)hub_->host[0].data())[i];
        for (int r = 1; r < P; ++r) {
          T v = ((const T*)hub_->host[(size_t)r].data())[i];
          acc = MAX ? (v > acc ? v : acc) : acc + v;
        }
        buf[i] = acc;
      }
      hub_->barrier();
      return;
    }
    hub_->stage[(size_t)rank_].resize(n * sizeof(T));
    LGA_HIP_CHECK(hipMemcpyAsync(hub_->stage[(size_t)rank_].get(), buf, n * sizeof(T), hipMemcpyDeviceToDevice, s));
    LGA_HIP_CHECK(hipStreamSynchronize(s));
    hub_->barrier();
    std::vector<const T*> parts((size_t)P);
    for (int r = 0; r < P; ++r) parts[(size_t)r] = (const T*)hub_->stage[(size_t)r].get();
    ptrs_.resize((size_t)P * sizeof(void*));
    LGA_HIP_CHECK(hipMemcpyAsync(ptrs_.get(), parts.data(), (size_t)P * sizeof(void*), hipMemcpyHostToDevice, s));
    int blocks = (int)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL((reduce_kernel<T, MAX>), dim3(blocks), dim3(256), 0, s, buf, (const T* const*)ptrs_.get(), P, n);
    LGA_HIP_CHECK(hipGetLastError());
    LGA_HIP_CHECK(hipStreamSynchronize(s));
    hub_->barrier();
  }
  std::shared_ptr<Hub> hub_;
  int rank_;
  DeviceBuffer<char> ptrs_;
};

}  // namespace

void make_loopback_group(int size, Comm** out) {
  auto hub = std::make_shared<Hub>(size);
  for (int r = 0; r < size; ++r) out[r] = new LoopbackComm(hub, r);
}

}  // namespace lga

// Device-side plumbing of the HIP build: error checks, the memory hook that
// lets PyTorch's caching allocator own every device buffer, pinned staging,
// the stream all kernels and collectives are issued on, and the device
// mirror of a binned dataset.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>
#include "../common.hpp"

namespace lga {

#define LGA_HIP_CHECK(expr)                                                                       \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + __FILE__ + ":" + \
                               std::to_string(__LINE__) + " (" #expr ")");                         \
  } while (0)

typedef void* (*DeviceAllocFn)(int64_t nbytes);
typedef void (*DeviceFreeFn)(void* p);

// Allocation goes through the registered hook (torch.empty under the hood) when
// one is installed, through hipMalloc otherwise (stand-alone C++ use).
void set_device_allocator(DeviceAllocFn a, DeviceFreeFn f);
void* device_alloc(size_t nbytes);
void device_free(void* p);
void set_current_stream(hipStream_t s);
hipStream_t current_stream();

template <typename T>
class DeviceBuffer {
 public:
  DeviceBuffer() {}
  explicit DeviceBuffer(size_t n) { resize(n); }
  ~DeviceBuffer() { release(); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
  DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
    if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
    return *this;
  }
  // Contents are NOT preserved.
  void resize(size_t n) {
    if (n <= n_) return;
    release();
    p_ = (T*)device_alloc(n * sizeof(T));
    n_ = n;
  }
  void release() { if (p_) device_free(p_); p_ = nullptr; n_ = 0; }
  T* get() const { return p_; }
  size_t size() const { return n_; }
  void upload(const T* host, size_t n, hipStream_t s) {
    resize(n);
    LGA_HIP_CHECK(hipMemcpyAsync(p_, host, n * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void download(T* host, size_t n, hipStream_t s) const {
    LGA_HIP_CHECK(hipMemcpyAsync(host, p_, n * sizeof(T), hipMemcpyDeviceToHost, s));
  }
  void zero(hipStream_t s) { if (p_) LGA_HIP_CHECK(hipMemsetAsync(p_, 0, n_ * sizeof(T), s)); }

 private:
  T* p_ = nullptr;
  size_t n_ = 0;
};

// Page-locked host memory is expensive to obtain (hipHostMalloc maps it into the device's address space: hundreds of
// microseconds) and boosters come and go by the hundred in a grid search, so released blocks are kept in a
// process-wide cache by size class (powers of two) and handed out again.  mapped = fine-grained, coherent memory the
// device can write directly.
void* host_block_acquire(size_t bytes, bool mapped, size_t* capacity);
void host_block_release(void* p, size_t capacity, bool mapped);

template <typename T>
class PinnedBuffer {
 public:
  PinnedBuffer() {}
  ~PinnedBuffer() { if (p_) host_block_release(p_, cap_, false); }
  PinnedBuffer(const PinnedBuffer&) = delete;
  PinnedBuffer& operator=(const PinnedBuffer&) = delete;
  void resize(size_t n) {
    if (n <= n_) return;
    if (p_) host_block_release(p_, cap_, false);
    p_ = (T*)host_block_acquire(n * sizeof(T), false, &cap_);
    n_ = n;
  }
  T* get() const { return p_; }
  T& operator[](size_t i) { return p_[i]; }
  size_t size() const { return n_; }

 private:
  T* p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

// Host memory the device can write directly (fine-grained, mapped): results of a growth round land
// here without a copy command.
class MappedBuffer {
 public:
  MappedBuffer() {}
  ~MappedBuffer() { if (p_) host_block_release(p_, cap_, true); }
  MappedBuffer(const MappedBuffer&) = delete;
  MappedBuffer& operator=(const MappedBuffer&) = delete;
  void resize(size_t n) {
    if (n <= n_) return;
    if (p_) host_block_release(p_, cap_, true);
    p_ = (char*)host_block_acquire(n, true, &cap_);
    LGA_HIP_CHECK(hipHostGetDevicePointer((void**)&d_, p_, 0));
    n_ = n;
  }
  char* get() const { return p_; }
  char* device_ptr() const { return d_; }
  size_t size() const { return n_; }

 private:
  char* p_ = nullptr;
  char* d_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

// Named ranges for rocprofv3 --marker-trace (roctx).  The library is bound at run time and only when
// profiling was asked for (profile=true or LGA_ROCTX=1); otherwise a range costs one branch.
void trace_enable(bool on);
void trace_push(const char* name);
void trace_pop();
struct TraceRange {
  explicit TraceRange(const char* name) { trace_push(name); }
  ~TraceRange() { trace_pop(); }
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
};

// Device copy of a root dataset's bin matrices.
struct DeviceMirror {
  int64_t rows = 0;
  int features = 0;   // used features
  int tiles = 0;
  DeviceBuffer<uint8_t> bins_rt;   // [tiles][rows][32]
  DeviceBuffer<uint8_t> bins_fm;   // [features][rows]
  DeviceBuffer<float> label;       // [rows]
  DeviceBuffer<float> weight;      // [rows] or empty
  bool has_weight = false;
  bool meta_uploaded = false;
};

}  // namespace lga

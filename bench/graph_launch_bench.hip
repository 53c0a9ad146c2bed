// Launch-latency experiment behind the growth rounds: a round is 4 short dependent kernels whose arguments
// change every round, and the host waits for a flag the last kernel writes.  Compared here:
//   direct   4 x hipLaunchKernelGGL on one stream
//   graph    one hipGraphLaunch of a 4-node graph whose kernel-node parameters are rewritten before every launch
// Both report the time per round (launch of round r+1 right after the host saw round r's flag).
//   hipcc --offload-arch=gfx950 -O3 bench/graph_launch_bench.hip -o /tmp/glb && /tmp/glb
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

struct Args { int v[64]; };   // by-value argument block, like the rounds' inline task lists

__global__ void work(Args a, int* buf, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] += a.v[i & 63];
}
__global__ void last(Args a, int* buf, int n, volatile unsigned* flag, unsigned seq) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] += a.v[i & 63];
  __threadfence_system();
  if (i == 0) *flag = seq;
}

int main() {
  const int n = 4096, rounds = 3000;
  int* buf;
  CK(hipMalloc(&buf, n * sizeof(int)));
  CK(hipMemset(buf, 0, n * sizeof(int)));
  unsigned* flag_host;
  CK(hipHostMalloc((void**)&flag_host, 64, hipHostMallocMapped));
  *flag_host = 0;
  unsigned* flag_dev;
  CK(hipHostGetDevicePointer((void**)&flag_dev, flag_host, 0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  Args a{};
  auto wait = [&](unsigned seq) { while (*(volatile unsigned*)flag_host != seq) {} };
  unsigned seq = 0;
  // direct launches
  for (int pass = 0; pass < 2; ++pass) {
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < rounds; ++r) {
      a.v[0] = r;
      ++seq;
      hipLaunchKernelGGL(work, dim3(16), dim3(256), 0, s, a, buf, n);
      hipLaunchKernelGGL(work, dim3(16), dim3(256), 0, s, a, buf, n);
      hipLaunchKernelGGL(work, dim3(16), dim3(256), 0, s, a, buf, n);
      hipLaunchKernelGGL(last, dim3(16), dim3(256), 0, s, a, buf, n, (volatile unsigned*)flag_dev, seq);
      wait(seq);
    }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    if (pass) std::printf("direct launches : %.1f us per round of 4 kernels\n", us);
  }
  CK(hipStreamSynchronize(s));
  // graph with per-round parameter updates
  hipGraph_t g;
  CK(hipGraphCreate(&g, 0));
  hipGraphNode_t node[4];
  int nn = n;
  volatile unsigned* fd = flag_dev;
  unsigned sq = 0;
  void* wargs[3] = {&a, &buf, &nn};
  void* largs[5] = {&a, &buf, &nn, &fd, &sq};
  hipKernelNodeParams kp[4];
  for (int k = 0; k < 4; ++k) {
    kp[k] = hipKernelNodeParams{};
    kp[k].func = k < 3 ? (void*)work : (void*)last;
    kp[k].gridDim = dim3(16);
    kp[k].blockDim = dim3(256);
    kp[k].sharedMemBytes = 0;
    kp[k].kernelParams = k < 3 ? wargs : largs;
    kp[k].extra = nullptr;
    CK(hipGraphAddKernelNode(&node[k], g, k ? &node[k - 1] : nullptr, k ? 1 : 0, &kp[k]));
  }
  hipGraphExec_t ge;
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int pass = 0; pass < 2; ++pass) {
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < rounds; ++r) {
      a.v[0] = r;
      sq = ++seq;
      for (int k = 0; k < 4; ++k) CK(hipGraphExecKernelNodeSetParams(ge, node[k], &kp[k]));
      CK(hipGraphLaunch(ge, s));
      wait(seq);
    }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    if (pass) std::printf("graph + updates : %.1f us per round of 4 kernels\n", us);
  }
  CK(hipStreamSynchronize(s));
  int h0;
  CK(hipMemcpy(&h0, buf, sizeof(int), hipMemcpyDeviceToHost));
  std::printf("checksum %d\n", h0);
  return 0;
}

#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)
struct Args { float* p; int n; float v; };
__global__ void k(const Args* a) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < a->n) a->p[i] += a->v; }
int main() {
  const int NK = 12, N = 1 << 20;
  float* buf; CK(hipMalloc(&buf, N * 4)); CK(hipMemset(buf, 0, N * 4));
  Args* dargs; CK(hipMalloc(&dargs, NK * sizeof(Args)));
  Args* hargs; CK(hipHostMalloc(&hargs, NK * sizeof(Args) * 8));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int i = 0; i < NK * 8; i++) hargs[i] = Args{buf, N, 1.0f};
  auto eager = [&](int iters) {
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; it++) {
      (void)hipMemcpyAsync(dargs, hargs + (it % 8) * NK, NK * sizeof(Args), hipMemcpyHostToDevice, s);
      for (int j = 0; j < NK; j++) hipLaunchKernelGGL(k, dim3(N / 256), dim3(256), 0, s, dargs + j);
    }
    (void)hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
  };
  printf("eager: %.1f us/iter\n", eager(200)); printf("eager: %.1f us/iter\n", eager(2000));
  // capture
  hipGraph_t g; hipGraphExec_t ge;
  std::vector<hipEvent_t> ev(2 * NK);
  for (auto& e : ev) CK(hipEventCreate(&e));
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int j = 0; j < NK; j++) {
    hipError_t e1 = hipEventRecord(ev[2 * j], s);
    hipLaunchKernelGGL(k, dim3(N / 256), dim3(256), 0, s, dargs + j);
    hipError_t e2 = hipEventRecord(ev[2 * j + 1], s);
    if (j == 0) printf("event record in capture: %s %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
  }
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  auto graph = [&](int iters) {
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; it++) {
      (void)hipMemcpyAsync(dargs, hargs + (it % 8) * NK, NK * sizeof(Args), hipMemcpyHostToDevice, s);
      (void)hipGraphLaunch(ge, s);
    }
    (void)hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
  };
  printf("graph: %.1f us/iter\n", graph(200)); printf("graph: %.1f us/iter\n", graph(2000));
  float ms = -1; hipError_t ee = hipEventElapsedTime(&ms, ev[0], ev[1]);
  printf("elapsed in graph: %s %.3f us\n", hipGetErrorString(ee), ms * 1e3);
  ee = hipEventElapsedTime(&ms, ev[0], ev[2 * NK - 1]);
  printf("elapsed whole graph: %s %.3f us\n", hipGetErrorString(ee), ms * 1e3);
  float h0; CK(hipMemcpy(&h0, buf, 4, hipMemcpyDeviceToHost)); printf("buf[0]=%.0f (expect %d)\n", h0, NK * (2200 + 2200));
  return 0;
}

// launch_floor - per-launch cost of back-to-back dependent launches on one stream (the "floor" every
// kernel of the runtime pays): empty kernel, by grid size / LDS size / launch API.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Args { const void *a, *b; void *c; const void *d; long long l[5]; int i[7]; };
__global__ void empty_kernel(Args a) { if (a.i[0] == 12345) ((int *)a.c)[0] = 1; }
__global__ void empty_lds(Args a) { extern __shared__ char lds[]; if (a.i[0] == 12345) ((int *)a.c)[0] = lds[threadIdx.x]; }
int main() {
  hipStream_t s; CHECK(hipStreamCreate(&s));
  Args a{}; void *buf; CHECK(hipMalloc(&buf, 4096)); a.c = buf;
  CHECK(hipFuncSetAttribute((const void *)empty_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  const int N = 2000;
  auto run = [&](const char *name, auto launch) {
    for (int i = 0; i < 200; ++i) launch();
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto t0 = std::chrono::high_resolution_clock::now();
    hipEventRecord(e0, s);
    for (int i = 0; i < N; ++i) launch();
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    auto t1 = std::chrono::high_resolution_clock::now();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.2f us/launch (events)  %7.2f us/launch (wall)\n", name, ms * 1e3 / N,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
  };
  run("empty <<<1,64>>>", [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, a); });
  run("empty <<<256,256>>>", [&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, a); });
  run("empty <<<(8,8,4),256>>>", [&] { hipLaunchKernelGGL(empty_kernel, dim3(8, 8, 4), dim3(256), 0, s, a); });
  run("empty <<<256,384>>> 128 KiB LDS", [&] { hipLaunchKernelGGL(empty_lds, dim3(256), dim3(384), 131072, s, a); });
  run("empty <<<2048,256>>>", [&] { hipLaunchKernelGGL(empty_kernel, dim3(2048), dim3(256), 0, s, a); });
  run("hipExtLaunch any-order <<<256,256>>>", [&] {
    hipExtLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, a); });
  // graph of 10 dependent launches
  hipGraph_t g; hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, a);
  CHECK(hipStreamEndCapture(s, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  {
    for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < N / 10; ++i) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.2f us/launch (events)\n", "hipGraph of 10 dependent <<<256,256>>>", ms * 1e3 / N);
  }
  return 0;
}

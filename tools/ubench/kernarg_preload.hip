// kernarg_preload.hip - what the first scalar load of a kernel (its arguments) costs a back-to-back launch on gfx950, and whether
// preloading the leading scalar arguments into SGPRs (-mllvm -amdgpu-kernarg-preload-count=14) removes it.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/kernarg_preload.hip -o /tmp/kp_plain
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=14 tools/ubench/kernarg_preload.hip -o /tmp/kp_preload
// Each binary times N dependent launches of (a) a kernel whose only work needs its arguments (one global load through a pointer
// argument, one store) and (b) a kernel that spins without looking at its arguments, on one stream.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Pad { long v[12]; };
// the work (a 5 us spin) starts only when the arguments have arrived AND a load through a pointer argument has returned: the
// launches are GPU-bound (the host runs ahead), so the time per launch = spin + everything between two dependent kernels
__global__ void uses_args(int n, int m, const int *src, int *dst, Pad p) {
  const int v = src[blockIdx.x & 63] + m; // (= 1)
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 500ull * (unsigned)v) __builtin_amdgcn_s_sleep(2);
  if (threadIdx.x == 0) dst[blockIdx.x] = v + n + (int)p.v[3];
}
__global__ void empty_kernel(int n, int m, const int *src, int *dst, Pad p) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 500ull) __builtin_amdgcn_s_sleep(2);
}
int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 4000, grid = argc > 2 ? atoi(argv[2]) : 256;
  int *src, *dst;
  CHECK(hipMalloc(&src, 4096));
  CHECK(hipMalloc(&dst, 4096 * 4));
  CHECK(hipMemset(src, 0, 4096));
  Pad p{};
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 3; ++rep) {
      for (int i = 0; i < 200; ++i) {
        if (which) hipLaunchKernelGGL(uses_args, dim3(grid), dim3(64), 0, s, i, 1, src, dst, p);
        else hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(64), 0, s, i, 1, src, dst, p);
      }
      CHECK(hipStreamSynchronize(s));
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        if (which) hipLaunchKernelGGL(uses_args, dim3(grid), dim3(64), 0, s, i, 1, src, dst, p);
        else hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(64), 0, s, i, 1, src, dst, p);
      }
      CHECK(hipStreamSynchronize(s));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
      printf("%-12s grid %d: %.3f us per launch\n", which ? "uses_args" : "empty", grid, us);
    }
  }
  return 0;
}

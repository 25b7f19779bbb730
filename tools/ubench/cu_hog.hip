// cu_hog - TEST HELPER (tests/test_chain_starved_gpu.py): a kernel that HOLDS compute units - `blocks` workgroups of 256 threads with
// `lds_bytes` of LDS each spin until *flag != 0 (a word in pinned host memory the test sets) or `max_ms` have passed (every spin is
// bounded: a forgotten flag cannot hang the GPU). With 120 KiB of LDS per workgroup a hogged CU cannot take one of the persistent
// chain kernel's 160 KiB workgroups: the chain launch is starved, which is what the test wants to see handled.
// Built by tpp-mlir_amd/build.py build_tools() into tools/cu_hog.so; never loaded by the product.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void cu_hog_kernel(const volatile unsigned *flag, unsigned long long max_ticks, unsigned *started) {
  extern __shared__ unsigned char hog_lds[];
  if (threadIdx.x == 0) {
    hog_lds[0] = 1; // (the LDS allocation is what occupies the CU; touch it so that it is not optimised away)
    atomicAdd(started, 1u);
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < max_ticks) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
    __builtin_amdgcn_s_sleep(32);
  }
}

// returns 0 on success; started: device word the workgroups count themselves into
extern "C" int cu_hog_launch(void *stream, int blocks, int lds_bytes, const void *flag, int max_ms, void *started) {
  if (hipFuncSetAttribute((const void *)cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return 1;
  hipLaunchKernelGGL(cu_hog_kernel, dim3((unsigned)blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (const volatile unsigned *)flag,
                     (unsigned long long)max_ms * 100000ull, (unsigned *)started);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

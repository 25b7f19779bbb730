// Read + write streaming on one MI355X: what does a copy-like eltwise kernel (relu: 4 B in, 4 B out per element) reach, by launch shape?
// Variants: U loads in flight per thread before the stores (1 / 2 / 4 / 8), nontemporal or plain accesses, grid-stride over all
// blocks or one contiguous span per block, blocks per CU. Prints TB/s (read + written bytes) for 256 MiB in + 256 MiB out.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool SPAN>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ in, f4* __restrict__ out, size_t n4) {
  const size_t T = (size_t)gridDim.x * 256;
  size_t i, step, end;
  if (SPAN) { // block b owns [b * per, (b + 1) * per): consecutive lanes, then consecutive rounds
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    i = (size_t)blockIdx.x * per + threadIdx.x; end = (size_t)(blockIdx.x + 1) * per; if (end > n4) end = n4; step = 256;
  } else { i = (size_t)blockIdx.x * 256 + threadIdx.x; end = n4; step = T; }
  for (; i + (U - 1) * step < end; i += U * step) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(in + i + u * step) : in[i + u * step];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f4 r = v[u];
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.0f ? r[e] : 0.0f;
      if (NT) __builtin_nontemporal_store(r, out + i + u * step); else out[i + u * step] = r;
    }
  }
  for (; i < end; i += step) { f4 r = in[i]; for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.0f ? r[e] : 0.0f; out[i] = r; }
}

template <int U, bool NT, bool SPAN> void run(const char* name, const f4* in, f4* out, size_t n4, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<U, NT, SPAN>), dim3(blocks), dim3(256), 0, 0, in, out, n4);
  hipEventRecord(e0, 0);
  const int it = 20;
  for (int w = 0; w < it; ++w) hipLaunchKernelGGL((k<U, NT, SPAN>), dim3(blocks), dim3(256), 0, 0, in, out, n4);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s blocks %6d: %7.1f us  %5.2f TB/s\n", name, blocks, ms * 1e3 / it, 2.0 * n4 * 16 * it / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t n4 = (size_t)8192 * 8192 / 4;
  f4 *in, *out; hipMalloc(&in, n4 * 16); hipMalloc(&out, n4 * 16); hipMemset(in, 1, n4 * 16);
  for (int blocks : {2048, 4096, 8192, 16384, 65536}) {
    run<1, false, false>("U=1 plain grid-stride", in, out, n4, blocks);
    run<4, false, false>("U=4 plain grid-stride", in, out, n4, blocks);
    run<4, true, false>("U=4 nontemporal grid-stride", in, out, n4, blocks);
    run<8, true, false>("U=8 nontemporal grid-stride", in, out, n4, blocks);
    run<4, true, true>("U=4 nontemporal span-per-block", in, out, n4, blocks);
    run<2, true, false>("U=2 nontemporal grid-stride", in, out, n4, blocks);
  }
  // context: a device-to-device memcpy of the same bytes
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemcpy(out, in, n4 * 16, hipMemcpyDeviceToDevice);
  hipEventRecord(e0, 0); for (int w = 0; w < 10; ++w) hipMemcpyAsync(out, in, n4 * 16, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s              : %7.1f us  %5.2f TB/s\n", "hipMemcpyAsync device-to-device", ms * 1e3 / 10, 2.0 * n4 * 16 * 10 / (ms * 1e-3) / 1e12);
  return 0;
}

// What does one s_barrier round cost a workgroup (gfx950)? One workgroup per CU (LDS request forces it), NW waves, each iteration:
// `work` back-to-back v_mfma_f32_32x32x16_bf16 (32 cycles each on its SIMD) in the first NM waves, nothing in the others (the
// loader waves' role), then s_barrier. Reports shader cycles per iteration minus the MFMA time = the exposed cost of the round.
// (profiles/r04_bf16_128x128_structure_ab.txt measures ~0.13 us = ~290 cycles per barrier round INSIDE the 128x128 bf16 kernel with
// loads and math switched off; this isolates the barrier itself.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int WORK>
__global__ __launch_bounds__(1024) void k(int iters, int nm, unsigned long long* out, float* sink) {
  extern __shared__ char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
  const bool mf = wave < nm;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (mf) {
#pragma unroll
      for (int w = 0; w < WORK; ++w) acc[w & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[w & 3], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) sink[0] = s;
  (void)smem;
}

template <int WORK> void run(int nw, int nm, unsigned long long* dout, float* dsink) {
  const int iters = 4000, nblk = 256;
  hipFuncSetAttribute((const void*)k<WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<WORK>), dim3(nblk), dim3(64 * nw), 98304, 0, iters, nm, dout, dsink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(nblk * 16);
  hipMemcpy(h.data(), dout, nblk * 128, hipMemcpyDeviceToHost);
  double cyc = 0;
  for (int b = 0; b < nblk; ++b) cyc += (double)h[b * 16];
  cyc /= nblk * (double)iters;
  // MFMA time per iteration on the busiest SIMD: waves are dealt round-robin over 4 SIMDs
  const int per_simd = (nm + 3) / 4;
  printf("%2d waves (%d with %2d MFMAs each): %7.1f cycles per round, %6.1f beyond the MFMA time of the busiest SIMD (%d)\n", nw, nm, WORK, cyc,
         cyc - 32.0 * WORK * per_simd, 32 * WORK * per_simd);
}

int main() {
  unsigned long long* dout; float* dsink;
  hipMalloc(&dout, 256 * 128); hipMalloc(&dsink, 64);
  for (int nw : {1, 2, 4, 6, 8, 10, 12, 16}) run<0>(nw, 0, dout, dsink);
  for (int nw : {4, 6, 10}) run<8>(nw, 4, dout, dsink);    // 4 MFMA waves x 8 MFMAs (half a 128x128x64 chunk) + idle "loaders"
  for (int nw : {4, 6, 10}) run<16>(nw, 4, dout, dsink);   // a whole chunk between barriers
  run<8>(10, 8, dout, dsink);                              // 8 MFMA waves (two per SIMD) + 2 idle
  run<4>(10, 8, dout, dsink);
  return 0;
}

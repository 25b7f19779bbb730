// fill_pattern.hip - L2 -> LDS fill rate of a CU by ACCESS PATTERN and wave structure (LDS-DMA, 1 KiB per wave instruction, L2-resident
// window shared by the CUs of an XCD, 256 workgroups = one per CU): what the loader waves of brgemm_bf16_lw's 128x128 tile can get.
//   pattern 0: a piece = 1 KiB contiguous (tools/ubench/fillmix.hip)
//   pattern 1: a piece = 8 rows x 128 B, row stride 2 KiB   (the A panel of a 1024-wide bf16 layer: 64 k of 8 rows)
//   pattern 2: a piece = 2 rows x 512 B, row stride 4 KiB   (the VNNI-2 B panel of a 1024-wide layer: 128 columns of 2 pair-rows)
//   pattern 3: even waves pattern 1, odd waves pattern 2 - the loader pair of the tile kernel
// nw loader waves issue BURST pieces each, wait until all but that burst have landed (counted vmcnt) and optionally meet - together
// with `idle` waves that do nothing else - at a workgroup barrier (the kernel's per-chunk barrier). GB/s per CU by s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;

template <int BURST>
__global__ __launch_bounds__(512) void k(const char *base, size_t win, int iters, unsigned long long *out, int pattern_, int barrier, int nw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= nw) { // (the MFMA waves of a tile kernel with nothing to do: they only take part in the barriers)
    if (barrier)
      for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
    return;
  }
  const int pattern = pattern_ == 3 ? 1 + (wave & 1) : pattern_;
  const char *p = base + (size_t)(blockIdx.x & 7) * win;
  unsigned voff, piece_stride;
  if (pattern == 1) { voff = (lane >> 3) * 2048 + (lane & 7) * 16; piece_stride = 8 * 2048; }
  else if (pattern == 2) { voff = (lane >> 5) * 4096 + (lane & 31) * 16; piece_stride = 2 * 4096; }
  else { voff = lane * 16; piece_stride = 1024; }
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 0x7fffffff, 0x00020000);
  char *my = smem + wave * (2 * BURST * 1024);
  // pattern 1 walks ALONG its rows from burst to burst (the next 64 k of the same 8 * BURST rows), the others to the next rows / pieces
  const unsigned start = wave * BURST * piece_stride + ((blockIdx.x >> 3) & 15) * 128 * (pattern == 0 ? 8 : 1);
  const unsigned step = pattern == 1 ? 128u : piece_stride * BURST * (unsigned)nw;
  const unsigned wrap = pattern == 1 ? 2048u - 16 * 128 : (unsigned)win - 2 * step - 65536;
  unsigned soff = start, walked = 0;
  const unsigned long long c0 = __builtin_readcyclecounter(), t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < BURST; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)(my + ((it & 1) * BURST + u) * 1024), 16, voff, soff + u * piece_stride, 0, 0);
    soff += step;
    walked += step;
    if (walked >= wrap) { soff = start; walked = 0; }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BURST) : "memory");
    if (barrier) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long c1 = __builtin_readcyclecounter(), t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = c1 - c0; out[(blockIdx.x * 8 + wave) * 2 + 1] = t1 - t0; }
  if (smem[threadIdx.x] == 0x7f && iters < 0) out[0] = 1;
}

template <int BURST> void run(int nw, int pattern, int barrier, int idle, const char *d, unsigned long long *dout) {
  const int iters = 2000, nblk = 256;
  const size_t win = (size_t)4 << 20;
  (void)hipFuncSetAttribute((const void *)k<BURST>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  (void)hipMemset(dout, 0, nblk * 128);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<BURST>, dim3(nblk), dim3(64 * (nw + idle)), 131072, 0, d, win, iters, dout, pattern, barrier, nw);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(nblk * 16);
  (void)hipMemcpy(h.data(), dout, nblk * 128, hipMemcpyDeviceToHost);
  double cyc = 0, ticks = 0;
  for (int b = 0; b < nblk; ++b) {
    unsigned long long mc = 0, mt = 0;
    for (int w = 0; w < nw; ++w) {
      mc = h[(b * 8 + w) * 2] > mc ? h[(b * 8 + w) * 2] : mc;
      mt = h[(b * 8 + w) * 2 + 1] > mt ? h[(b * 8 + w) * 2 + 1] : mt;
    }
    cyc += (double)mc;
    ticks += (double)mt;
  }
  cyc /= nblk;
  ticks /= nblk;
  const double bytes = (double)iters * BURST * 1024 * nw;
  printf("pattern %d  %d loader wave(s) + %d idle, burst %2d, barrier %d: %6.1f B/clk/CU  %6.1f GB/s/CU  %.3f us per 32 KiB  (%.2f GHz)\n", pattern, nw, idle,
         BURST, barrier, bytes / cyc, bytes / (ticks * 10e-9) / 1e9, 32768.0 / (bytes / (ticks * 10e-9)) * 1e6, cyc / (ticks * 10.0));
}

int main() {
  char *d;
  unsigned long long *dout;
  (void)hipMalloc(&d, (size_t)64 << 20);
  (void)hipMemset(d, 1, (size_t)64 << 20);
  (void)hipMalloc(&dout, 256 * 128);
  for (int pattern = 0; pattern < 3; ++pattern)
    for (int nw = 1; nw <= 4; nw *= 2) {
      run<16>(nw, pattern, 0, 0, d, dout);
      run<16>(nw, pattern, 1, 0, d, dout);
    }
  // the loader pair of the 128x128 bf16 tile: one wave on an A panel, one on a B panel, 16 pieces each per barrier, 4 idle waves
  run<16>(2, 3, 1, 0, d, dout);
  run<16>(2, 3, 1, 4, d, dout);
  run<16>(2, 1, 1, 4, d, dout);
  run<16>(2, 2, 1, 4, d, dout);
  run<8>(4, 3, 1, 4, d, dout);
  return 0;
}

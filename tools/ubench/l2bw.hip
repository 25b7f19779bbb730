// L2 -> CU streaming bandwidth microbenchmark (gfx950): every workgroup re-reads a window of
// `win` bytes (L2 resident) with buffer_load_dwordx4 into VGPRs or with LDS-DMA, 4 waves/WG,
// one WG per CU (LDS request forces it). Reports B/clk/CU using wall_clock64 (100 MHz) and an
// assumed clock from s_memtime deltas.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

template <int MODE, int ROWSTRIDE>  // MODE 0: VGPR loads, 1: LDS-DMA ; ROWSTRIDE: bytes between 128-B rows (0 = contiguous 1 KiB)
__global__ __launch_bounds__(256) void k(const char* base, size_t win, int iters, unsigned long long* out, int shared_window) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // each WG gets its own window (shared_window=0) or XCD-shared window (1)
  const size_t wg_off = shared_window ? (size_t)(blockIdx.x & 7) * win : (size_t)blockIdx.x * win;
  const char* p = base + wg_off;
  unsigned voff;
  if (ROWSTRIDE == 0) voff = lane * 16;
  else voff = (lane >> 3) * ROWSTRIDE + (lane & 7) * 16;
  const unsigned instr_bytes = ROWSTRIDE == 0 ? 1024 : 8 * ROWSTRIDE;
  u32x4 acc = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
  unsigned soff = wave * instr_bytes;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        acc += v;
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)(smem + wave * 8192 + u * 1024), 16, voff, soff, 0, 0);
      }
      soff += 4 * instr_bytes;
      if (soff >= win) soff = wave * instr_bytes;
    }
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (MODE == 1) acc[0] += smem[threadIdx.x * 16];
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = w1 - w0; }
  if (acc[0] == 0x12345678u && acc[1] == 77u) out[0] = acc[2];
}

template <int MODE, int RS>
void run(const char* name, const char* d, size_t win, int shared, unsigned long long* dout, int nblk) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k<MODE, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, RS>), dim3(nblk), dim3(256), 98304, 0, d, win, iters, dout, shared);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(nblk * 4);
  hipMemcpy(h.data(), dout, nblk * 32, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int b = 0; b < nblk; ++b) { cyc += h[b * 4]; wall += h[b * 4 + 1]; }
  cyc /= nblk; wall /= nblk;
  const double bytes = (double)iters * 8 * 4 * 1024;  // per WG: 4 waves x 8 instr x 1 KiB per iter
  printf("%-44s win %6zu KiB %s: %6.1f B/clk/CU  (%.2f GHz, %.2f TB/s over %d CUs)\n", name, win >> 10, shared ? "XCD-shared" : "private  ",
         bytes / cyc, cyc / (wall * 10.0) / 1e3 * 1e3 / 1e3, bytes * nblk / (wall * 10e-9) / 1e12, nblk);
}

int main() {
  const int nblk = 256;
  char* d; unsigned long long* dout;
  const size_t total = (size_t)512 << 20;
  hipMalloc(&d, total); hipMemset(d, 1, total); hipMalloc(&dout, nblk * 32);
  for (size_t win : {(size_t)64 << 10, (size_t)1 << 20}) {
    run<0, 0>("VGPR loads, contiguous 1 KiB/instr", d, win, 0, dout, nblk);
    run<1, 0>("LDS-DMA,    contiguous 1 KiB/instr", d, win, 0, dout, nblk);
    run<0, 2048>("VGPR loads, 8 rows x 128 B (stride 2 KiB)", d, win, 0, dout, nblk);
    run<1, 2048>("LDS-DMA,    8 rows x 128 B (stride 2 KiB)", d, win, 0, dout, nblk);
  }
  run<0, 0>("VGPR loads, contiguous", d, (size_t)2 << 20, 1, dout, nblk);
  run<1, 0>("LDS-DMA,    contiguous", d, (size_t)2 << 20, 1, dout, nblk);
  run<0, 2048>("VGPR loads, 8 rows x 128 B", d, (size_t)2 << 20, 1, dout, nblk);
  run<1, 2048>("LDS-DMA,    8 rows x 128 B", d, (size_t)2 << 20, 1, dout, nblk);
  return 0;
}

// L2 -> LDS fill paths of one CU (gfx950): can LDS-DMA (buffer_load ... lds) and register staging (buffer_load_dwordx4 ->
// ds_write_b128) run side by side faster than either alone? One workgroup per CU (LDS request forces it), NW active waves, each
// streaming 1 KiB pieces of an L2-resident window: mode D = LDS-DMA, R = register staging (8 loads in flight, then 8 LDS writes),
// V = loads into VGPRs only (no LDS write). Reports B/clk/CU (shader cycles by s_memtime).
// Result (profiles/r03_fill_paths_ubench.txt): every path saturates at ~59 B/clk/CU - the vector L1's 64 B/clk - so the fill rate
// of a tile kernel is a property of the CU, not of the instruction used.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

// roles: 2 bits per wave (0 idle, 1 D, 2 R, 3 V), wave w at bits [2w, 2w+1]
__global__ __launch_bounds__(512) void k(const char* base, size_t win, int iters, unsigned long long* out, unsigned roles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int role = (roles >> (2 * wave)) & 3;
  if (role == 0) return;
  const char* p = base + (size_t)(blockIdx.x & 7) * win;  // XCD-shared window: L2 hits after the first pass
  const unsigned voff = lane * 16;
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
  unsigned soff = wave * 8192;
  char* my = smem + wave * 16384;
  if (role == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)(my + ((it & 1) * 8 + u) * 1024), 16, voff, soff + u * 1024, 0, 0);
      soff += 65536;
      if (soff >= win) soff = wave * 8192;
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  } else if (role == 2) {
    u32x4 v[2][8];
    const unsigned lbase = wave * 16384 + lane * 16; // dynamic LDS starts at 0 (no static LDS); asm: the compiler drops plain dead LDS stores
#pragma unroll
    for (int u = 0; u < 8; ++u) v[0][u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff + u * 1024, 0);
    for (int it = 0; it < iters; it += 2) {
      soff += 65536;
      if (soff >= win) soff = wave * 8192;
#pragma unroll
      for (int u = 0; u < 8; ++u) v[1][u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff + u * 1024, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_write_b128 %0, %1" ::"v"(lbase + u * 1024), "v"(v[0][u]) : "memory");
      soff += 65536;
      if (soff >= win) soff = wave * 8192;
#pragma unroll
      for (int u = 0; u < 8; ++u) v[0][u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff + u * 1024, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_write_b128 %0, %1" ::"v"(lbase + 8192 + u * 1024), "v"(v[1][u]) : "memory");
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[0][u];
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff + u * 1024, 0);
      soff += 65536;
      if (soff >= win) soff = wave * 8192;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  acc[0] += smem[threadIdx.x * 16];
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (acc[0] == 0x12345678u && acc[1] == 77u) out[0] = acc[2];
}

void run(const char* name, unsigned roles, const char* d, unsigned long long* dout) {
  const int iters = 2000, nblk = 256;
  const size_t win = (size_t)2 << 20;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipMemset(dout, 0, nblk * 64);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(nblk), dim3(512), 131072, 0, d, win, iters, dout, roles);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(nblk * 8);
  hipMemcpy(h.data(), dout, nblk * 64, hipMemcpyDeviceToHost);
  double cyc = 0;
  int nw = 0;
  for (int w = 0; w < 8; ++w) nw += ((roles >> (2 * w)) & 3) != 0;
  for (int b = 0; b < nblk; ++b) {
    unsigned long long mx = 0;
    for (int w = 0; w < 8; ++w) mx = h[b * 8 + w] > mx ? h[b * 8 + w] : mx;
    cyc += (double)mx;
  }
  cyc /= nblk;
  const double bytes = (double)iters * 8 * 1024 * nw;
  printf("%-64s %d waves: %6.1f B/clk/CU  (%.1f cycles per 1 KiB piece per CU)\n", name, nw, bytes / cyc, cyc / (bytes / 1024));
}

int main() {
  char* d; unsigned long long* dout;
  hipMalloc(&d, (size_t)64 << 20); hipMemset(d, 1, (size_t)64 << 20); hipMalloc(&dout, 256 * 64);
  // roles: wave w at bits 2w: 1 = D (LDS-DMA), 2 = R (loads + ds_write_b128), 3 = V (loads only)
  run("1 x D", 0x1, d, dout);
  run("2 x D", 0x5, d, dout);
  run("4 x D", 0x55, d, dout);
  run("8 x D", 0x5555, d, dout);
  run("1 x R", 0x2, d, dout);
  run("2 x R", 0xA, d, dout);
  run("4 x R", 0xAA, d, dout);
  run("8 x R", 0xAAAA, d, dout);
  run("2 x V", 0xF, d, dout);
  run("4 x V", 0xFF, d, dout);
  run("8 x V", 0xFFFF, d, dout);
  run("1 D + 1 R", 0x9, d, dout);
  run("2 D + 2 R", 0xA5, d, dout);
  run("4 D + 4 R", 0xAA55, d, dout);
  run("2 D + 2 V", 0xF5, d, dout);
  run("2 D + 4 R", 0xAAA5, d, dout);
  return 0;
}

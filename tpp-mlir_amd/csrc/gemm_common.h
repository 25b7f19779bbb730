// gemm_common.h - device-side helpers shared by the BRGEMM kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "xsmm_desc.h"

namespace tpp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// EP_* epilogue bits: xsmm_desc.h

struct GemmArgs {
  const void *A;
  const void *B;
  void *C;
  const void *D;       // bias row (length n) when EP_BIAS
  int64_t lda, ldb, ldc, stride_a, stride_b; // elements
  int m, n, k, br;
  int ep;              // EP_* bits
  int tiles_m, tiles_n;
  int vf;              // VNNI blocking factor of B (2 or 4) when the operand is VNNI-packed
  int xn_shift;        // brgemm_f32_lw, XCD-blocked grid: log2 of the XCD blocks along N (set by its launcher; blockIdx.x = M block << xn_shift | N block)
  // SPLIT kernels (the batch-reduce range of one output tile over `split` workgroups, set by their launchers): partial tiles
  // [tile][split][BM * BN] and one arrival counter per tile, both in the launch stream's scratch block (split_scratch.h)
  int split;
  float *scratch;
  unsigned *split_cnt;
  int b_trans;         // generic kernel: B[k][j] = B_ptr[j * ldb + k] (GemmDesc::b_trans)
};

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}
// round-to-nearest-even, NaN -> quiet NaN (same bit recipe as the oracle)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// global-address-space views: operand pointers that come out of a work list are generic pointers loaded
// from memory; without the explicit address space every access becomes a flat_load / flat_store
typedef __attribute__((address_space(1))) const void g_cvoid;
typedef __attribute__((address_space(1))) void g_void;

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float load(const void *p, int64_t i) { return ((const float *)p)[i]; }
  static __device__ __forceinline__ void store(void *p, int64_t i, float v) { ((float *)p)[i] = v; }
  static __device__ __forceinline__ float load(g_cvoid *p, int64_t i) {
    return ((__attribute__((address_space(1))) const float *)p)[i];
  }
  static __device__ __forceinline__ void store(g_void *p, int64_t i, float v) { ((__attribute__((address_space(1))) float *)p)[i] = v; }
};
template <> struct Elem<unsigned short> {
  static __device__ __forceinline__ float load(const void *p, int64_t i) {
    return bf16_bits_to_f32(((const unsigned short *)p)[i]);
  }
  static __device__ __forceinline__ void store(void *p, int64_t i, float v) {
    ((unsigned short *)p)[i] = f32_to_bf16_bits(v);
  }
  static __device__ __forceinline__ float load(g_cvoid *p, int64_t i) {
    return bf16_bits_to_f32(((__attribute__((address_space(1))) const unsigned short *)p)[i]);
  }
  static __device__ __forceinline__ void store(g_void *p, int64_t i, float v) {
    ((__attribute__((address_space(1))) unsigned short *)p)[i] = f32_to_bf16_bits(v);
  }
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: set it once per (call site, device). `done` is
// the call site's own bit mask of devices (bit = device id mod 64), updated atomically - concurrent first callers
// may both set the attribute (harmless), nobody skips it.
} // namespace tpp
#include <atomic>
namespace tpp {
static inline hipError_t ensure_dynamic_lds(const void *kernel, int bytes, std::atomic<unsigned long long> &done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}
// Cache policy (buffer-instruction aux bits) of the stores of an output tile that this launch does not read again: 16 = sc1,
// write-through. A line that is already in memory when the kernel ends is one the end-of-kernel write-back of the L2s does not have
// to move, and the launch boundary behind a kernel that leaves B dirty bytes costs ~B / 6 TB/s (0.7 us for the 4 MiB of C2).
// Same box, C2 17.57 -> 17.34 us, C3 9.94 -> 9.85 (profiles/r03_f32_lw_laps_ab.txt); sc0+sc1, sc1+nt measured the same, nt alone
// helps C2 but not C3. ONLY where one store instruction writes whole 128-byte lines (8 lanes x 16 bytes or 32 lanes x 4 bytes of
// one row): write-through of partial lines costs - the reg-staged 64x64 bf16 kernel (32 contiguous bytes per row and instruction)
// measured 4-7 % slower with sc1, the VNNI-2 pack kernel (two 16-byte stores per lane) 6.5 -> 3.8 TB/s; both keep plain stores.
// -DTPP_C_STORE_AUX=n: side builds for A/B runs.
#ifndef TPP_C_STORE_AUX
#define TPP_C_STORE_AUX 16
#endif
constexpr int C_STORE_AUX = TPP_C_STORE_AUX;

// compute units of the current device (queried once per process; tile heuristics use it)
static inline int device_cu_count() {
  static std::atomic<int> cus{0};
  int v = cus.load(std::memory_order_relaxed);
  if (v > 0) return v;
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) v = n;
  else {
    (void)hipGetLastError(); // no device visible (dispatch on a CPU-only host still plans for an MI355X)
    v = 256;
  }
  cus.store(v, std::memory_order_relaxed);
  return v;
}
} // namespace tpp

// eltwise.hip - the HBM-bound members of the xsmm op set for gfx950:
//   unary  IDENTITY / ZERO / RELU with row / column / scalar broadcast
//          (xsmm.unary, XsmmOps.td:30-70; XsmmUtils.cpp:105-126,254-288)
//   unary  TRANSPOSE (m, n = INPUT dims, ConvertLinalgToXsmm.cpp:147-148)
//   unary  VNNI2 pack [m][n] -> [m/2][n][2]  (ConvertLinalgToXsmm.cpp:1036-1083)
//   binary ADD / MUL / SUB / DIV with per-operand broadcast
//          (xsmm.binary, XsmmOps.td:72-120; XsmmUtils.cpp:193-252,290-352)
// These kernels move 16 bytes per lane per access whenever shape/stride/alignment
// allow (coalesced 1 KiB per wave-instruction) and fall back to element accesses
// otherwise. bf16 arithmetic is done in f32 with one RNE rounding at the store;
// IDENTITY / ZERO / TRANSPOSE / VNNI2 are bit-exact moves in the storage type
// (XsmmRunnerUtils.cpp:29-59).
#include "gemm_common.h"
#include "xsmm_desc.h"

namespace tpp {

typedef unsigned int u32x4_e __attribute__((ext_vector_type(4)));

enum : int64_t { U_IDENTITY = 1, U_ZERO = 2, U_RELU = 5, U_VNNI2 = 28, U_TRANSPOSE = 29 };
enum : int64_t { UF_ROW = 2, UF_COL = 4, UF_SCALAR = 8 };
enum : int64_t { B_ADD = 1, B_MUL = 2, B_SUB = 3, B_DIV = 4 };
enum : int64_t { BF_ROW0 = 1, BF_ROW1 = 2, BF_COL0 = 4, BF_COL1 = 8, BF_SC0 = 16, BF_SC1 = 32 };
enum : int { BC_NONE = 0, BC_ROW = 1, BC_COL = 2, BC_SCALAR = 3 };

template <typename T> struct Bits;
template <> struct Bits<float> {
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Bits<unsigned short> {
  static __device__ __forceinline__ float to_f32(unsigned short v) { return bf16_bits_to_f32(v); }
  static __device__ __forceinline__ unsigned short from_f32(float v) { return f32_to_bf16_bits(v); }
};

template <typename T, int VEC> struct alignas(sizeof(T) * VEC) Pack { T v[VEC]; };

template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_operand(const T *p, int bc, int64_t i, int64_t j, int64_t ld) {
  Pack<T, VEC> r;
  if (bc == BC_NONE) {
    r = *(const Pack<T, VEC> *)(p + i * ld + j);
  } else if (bc == BC_COL) {
    r = *(const Pack<T, VEC> *)(p + j);
  } else {
    const T s = bc == BC_ROW ? p[i * ld] : p[0];
#pragma unroll
    for (int e = 0; e < VEC; ++e) r.v[e] = s;
  }
  return r;
}

// Streaming accesses of a 16-byte pack: nontemporal (the operands of these ops are read once and written once - keeping them out
// of the caches' way measured +6..+20 % on read + write streams, tools/ubench/stream_rw.hip)
template <typename T, int VEC> __device__ __forceinline__ Pack<T, VEC> load_stream(const T *p) {
  if constexpr (sizeof(Pack<T, VEC>) == 16) return __builtin_bit_cast(Pack<T, VEC>, __builtin_nontemporal_load((const u32x4_e *)p));
  else return *(const Pack<T, VEC> *)p;
}
// 16-byte WRITE-THROUGH store (sc1): the line is in memory when the store retires instead of sitting dirty in the L2 until it is
// evicted or until the end-of-kernel write-back. Measured by size (profiles/r03_eltwise_store_policy.txt, 2048^2 .. 16384^2, same
// box, GB/s against the nontemporal / plain stores used before): transposes f32 +0..31 %, bf16 +0..49 %, binary add +0..14 % (one
// size -3 %), relu f32 +1..23 %; relu bf16 is mixed (-7 % at 8192^2, +16 % at 2048^2): the bf16 unary streams keep nontemporal
// stores. (sc1 nt is better still for f32 transposes up to 8192^2 and worse beyond: the repeated benchmark input then stays in the
// 256 MiB Infinity Cache - not a property a single transpose has, so it is not used.) Inline asm: the builtins reach sc1 only
// through buffer stores, whose 32-bit offsets do not span these operands.
template <typename T, int VEC> __device__ __forceinline__ void store_wt(T *p, const Pack<T, VEC> &v) {
  static_assert(sizeof(Pack<T, VEC>) == 16, "16-byte packs");
  const u32x4_e x = __builtin_bit_cast(u32x4_e, v);
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}
template <typename T, int VEC, bool WT = false> __device__ __forceinline__ void store_stream(T *p, const Pack<T, VEC> &v) {
  if constexpr (sizeof(Pack<T, VEC>) == 16 && WT) store_wt<T, VEC>(p, v);
  else if constexpr (sizeof(Pack<T, VEC>) == 16) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_e, v), (u32x4_e *)p);
  else *(Pack<T, VEC> *)p = v;
}
// Work distribution of the eltwise kernels: block b owns ONE contiguous span of the index space (rounded to whole rounds of 256
// lanes) and walks it four rounds at a time - four independent 16-byte loads in flight per lane, then the four stores. Measured
// on a 256 MiB + 256 MiB relu-like stream (tools/ubench/stream_rw.hip): 6.4-6.5 TB/s against 5.3 for the grid-stride loop with one
// load in flight that these kernels used before (a device-to-device hipMemcpy: 5.4).
constexpr int EW_U = 4;
__device__ __forceinline__ void span_of_block(int64_t total, int64_t &idx, int64_t &end) {
  const int64_t per = (((total + gridDim.x - 1) / gridDim.x) + 255) & ~(int64_t)255;
  idx = (int64_t)blockIdx.x * per + threadIdx.x;
  end = (int64_t)(blockIdx.x + 1) * per;
  if (end > total) end = total;
}

// IDENTITY / ZERO / RELU. Each thread owns VEC consecutive columns of one row.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void unary_kernel(int op, int bc, int64_t m, int64_t n, int64_t ldi, int64_t ldo,
                                                    const T *in, T *out, float scalar,
                                                    int use_scalar) {
  const int64_t nv = n / VEC, total = m * nv;
  const bool small = total < ((int64_t)1 << 31); // 32-bit index division (a 64-bit one is ~100 instructions per element)
  const T sc = use_scalar ? Bits<T>::from_f32(scalar) : T(0);
  const bool stream_in = bc == BC_NONE && op != (int)U_ZERO && !use_scalar;
  auto fetch = [&](int64_t idx, int64_t &o) __attribute__((always_inline)) {
    // (m == 1: the launcher flattened a contiguous operand - no division)
    const int64_t i = m == 1 ? 0 : small ? (int64_t)((unsigned)idx / (unsigned)nv) : idx / nv, j = (idx - i * nv) * VEC;
    o = i * ldo + j;
    Pack<T, VEC> x;
    if (op == (int)U_ZERO) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) x.v[e] = T(0);
    } else if (use_scalar) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) x.v[e] = sc;
    } else if (stream_in) {
      x = load_stream<T, VEC>(in + i * ldi + j);
    } else {
      x = load_operand<T, VEC>(in, bc, i, j, ldi);
    }
    return x;
  };
  auto finish = [&](Pack<T, VEC> x, int64_t o) __attribute__((always_inline)) {
    if (op == (int)U_RELU) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float f = Bits<T>::to_f32(x.v[e]);
        x.v[e] = Bits<T>::from_f32(f > 0.0f ? f : 0.0f);
      }
    }
    // (a pure fill - zero / scalar broadcast - keeps plain stores: nontemporal ones measured 7.4 -> 5.8 TB/s on a 256 MiB fill)
    if (op == (int)U_ZERO || use_scalar) *(Pack<T, VEC> *)(out + o) = x;
    else store_stream<T, VEC, sizeof(T) == 4>(out + o, x);
  };
  int64_t idx, end;
  span_of_block(total, idx, end);
  for (; idx + (EW_U - 1) * 256 < end; idx += EW_U * 256) {
    Pack<T, VEC> x[EW_U];
    int64_t o[EW_U];
#pragma unroll
    for (int u = 0; u < EW_U; ++u) x[u] = fetch(idx + u * 256, o[u]);
#pragma unroll
    for (int u = 0; u < EW_U; ++u) finish(x[u], o[u]);
  }
  for (; idx < end; idx += 256) {
    int64_t o;
    const Pack<T, VEC> x = fetch(idx, o);
    finish(x, o);
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void binary_kernel(int op, int bc0, int bc1, int64_t m, int64_t n, int64_t ldl,
                                                     int64_t ldr, int64_t ldo, const T *lhs, const T *rhs, T *out) {
  const int64_t nv = n / VEC, total = m * nv;
  const bool small = total < ((int64_t)1 << 31); // 32-bit index division
  auto fetch = [&](int64_t idx, Pack<T, VEC> &l, Pack<T, VEC> &r, int64_t &o) __attribute__((always_inline)) {
    const int64_t i = m == 1 ? 0 : small ? (int64_t)((unsigned)idx / (unsigned)nv) : idx / nv, j = (idx - i * nv) * VEC;
    o = i * ldo + j;
    l = bc0 == BC_NONE ? load_stream<T, VEC>(lhs + i * ldl + j) : load_operand<T, VEC>(lhs, bc0, i, j, ldl);
    r = bc1 == BC_NONE ? load_stream<T, VEC>(rhs + i * ldr + j) : load_operand<T, VEC>(rhs, bc1, i, j, ldr);
  };
  auto finish = [&](const Pack<T, VEC> &l, const Pack<T, VEC> &r, int64_t o) __attribute__((always_inline)) {
    Pack<T, VEC> x;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float a = Bits<T>::to_f32(l.v[e]), b = Bits<T>::to_f32(r.v[e]);
      float c;
      switch (op) {
      case (int)B_ADD: c = a + b; break;
      case (int)B_MUL: c = a * b; break;
      case (int)B_SUB: c = a - b; break;
      default: c = a / b; break;
      }
      x.v[e] = Bits<T>::from_f32(c);
    }
    store_stream<T, VEC, true>(out + o, x);
  };
  int64_t idx, end;
  span_of_block(total, idx, end);
  for (; idx + (EW_U - 1) * 256 < end; idx += EW_U * 256) {
    Pack<T, VEC> l[EW_U], r[EW_U];
    int64_t o[EW_U];
#pragma unroll
    for (int u = 0; u < EW_U; ++u) fetch(idx + u * 256, l[u], r[u], o[u]);
#pragma unroll
    for (int u = 0; u < EW_U; ++u) finish(l[u], r[u], o[u]);
  }
  for (; idx < end; idx += 256) {
    Pack<T, VEC> l, r;
    int64_t o;
    fetch(idx, l, r, o);
    finish(l, r, o);
  }
}

// out[j][i] = in[i][j], i < m, j < n. 64x64 tiles through LDS (row padded by one
// element: column reads are bank-conflict free); both global sides are coalesced.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(int64_t m, int64_t n, int64_t ldi, int64_t ldo,
                                                        const T *__restrict__ in, T *__restrict__ out) {
  __shared__ T tile[64][65];
  const int64_t tiles_n = (n + 63) / 64;
  const int64_t i0 = (blockIdx.x / tiles_n) * 64, j0 = (blockIdx.x % tiles_n) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // 64 x 4
#pragma unroll 4
  for (int r = ty; r < 64; r += 4)
    if (i0 + r < m && j0 + tx < n) tile[r][tx] = in[(i0 + r) * ldi + j0 + tx];
  __syncthreads();
#pragma unroll 4
  for (int r = ty; r < 64; r += 4)
    if (j0 + r < n && i0 + tx < m) out[(j0 + r) * ldo + i0 + tx] = tile[tx][r];
}

// VNNI-2 pack of 16-bit elements: out[(i/2)*(2*ldo) + 2*j + (i&1)] = in[i*ldi + j].
// Vector path: a thread takes 8 columns of a row pair (2 x 16 B in, 2 x 16 B out).
template <int VEC>
__global__ __launch_bounds__(256) void vnni2_kernel(int64_t m, int64_t n, int64_t ldi, int64_t ldo,
                                                    const unsigned short *__restrict__ in,
                                                    unsigned short *__restrict__ out) {
  const int64_t nv = n / VEC, total = (m / 2) * nv;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / nv, j = (idx - r * nv) * VEC;
    const Pack<unsigned short, VEC> e = *(const Pack<unsigned short, VEC> *)(in + (2 * r) * ldi + j);
    const Pack<unsigned short, VEC> o = *(const Pack<unsigned short, VEC> *)(in + (2 * r + 1) * ldi + j);
    Pack<unsigned int, VEC> w;
#pragma unroll
    for (int q = 0; q < VEC; ++q) w.v[q] = (unsigned int)e.v[q] | ((unsigned int)o.v[q] << 16);
    unsigned int *dst = (unsigned int *)(out + r * (2 * ldo) + 2 * j);
    if constexpr (VEC == 8) {
      *(Pack<unsigned int, 4> *)dst = *(const Pack<unsigned int, 4> *)&w.v[0];
      *(Pack<unsigned int, 4> *)(dst + 4) = *(const Pack<unsigned int, 4> *)&w.v[VEC / 2];
    } else {
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        out[r * (2 * ldo) + 2 * (j + q)] = e.v[q];
        out[r * (2 * ldo) + 2 * (j + q) + 1] = o.v[q];
      }
    }
  }
}

// Row-pair variant of the 16-byte path for matrices whose row pairs fit the grid's y dimension (every shape the
// reference packs: weights of a layer): blockIdx.y = row pair, 256 lanes x 8 columns per block along x. No
// grid-stride loop and no 64-bit division - at the C5 size (2048^2 bf16, 16 MiB moved in ~3 us) the kernel is a
// single wave of blocks and the address arithmetic of the generic loop was a measurable part of it.
__global__ __launch_bounds__(256) void vnni2_rows_kernel(int n8, int64_t ldi, int64_t ldo,
                                                         const unsigned short *__restrict__ in,
                                                         unsigned short *__restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x; // 8-column piece
  if (c >= n8) return;
  const int64_t r = blockIdx.y;
  const u32x4_e e = *(const u32x4_e *)(in + (2 * r) * ldi + 8 * (int64_t)c);
  const u32x4_e o = *(const u32x4_e *)(in + (2 * r + 1) * ldi + 8 * (int64_t)c);
  u32x4_e w0, w1; // dword q of the output = (even row element q, odd row element q)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    w0[2 * q] = (e[q] & 0xffffu) | (o[q] << 16);
    w0[2 * q + 1] = (e[q] >> 16) | (o[q] & 0xffff0000u);
    w1[2 * q] = (e[q + 2] & 0xffffu) | (o[q + 2] << 16);
    w1[2 * q + 1] = (e[q + 2] >> 16) | (o[q + 2] & 0xffff0000u);
  }
  u32x4_e *dst = (u32x4_e *)(out + r * (2 * ldo) + 16 * (int64_t)c);
  dst[0] = w0;
  dst[1] = w1;
}

// The same with ONE 16-byte output piece per lane (4 columns x 2 rows): two 8-byte loads, one 16-byte store - a wave's store
// instruction writes 1 KiB of whole lines (the 8-column variant's two stores per lane each write half of every line), and twice as
// many lanes are in flight for the one round trip the kernel consists of. TPP_HIP_PACK_PIECE=8 selects the 8-column variant (A/B).
typedef unsigned int u32x2_e __attribute__((ext_vector_type(2)));
template <int POLICY> // 0 plain, 1 nontemporal loads + stores, 2 nontemporal loads + write-through (sc1) stores
__global__ __launch_bounds__(256) void vnni2_rows4_kernel(int n4, int64_t ldi, int64_t ldo, const unsigned short *__restrict__ in,
                                                          unsigned short *__restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x; // 4-column piece
  if (c >= n4) return;
  const int64_t r = blockIdx.y;
  const u32x2_e *pe = (const u32x2_e *)(in + (2 * r) * ldi + 4 * (int64_t)c), *po = (const u32x2_e *)(in + (2 * r + 1) * ldi + 4 * (int64_t)c);
  const u32x2_e e = POLICY ? __builtin_nontemporal_load(pe) : *pe;
  const u32x2_e o = POLICY ? __builtin_nontemporal_load(po) : *po;
  u32x4_e w;
  w[0] = (e[0] & 0xffffu) | (o[0] << 16);
  w[1] = (e[0] >> 16) | (o[0] & 0xffff0000u);
  w[2] = (e[1] & 0xffffu) | (o[1] << 16);
  w[3] = (e[1] >> 16) | (o[1] & 0xffff0000u);
  u32x4_e *dst = (u32x4_e *)(out + r * (2 * ldo) + 8 * (int64_t)c);
  if (POLICY == 1) __builtin_nontemporal_store(w, dst);
  else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(w) : "memory");
  else *dst = w;
}

// ---- grouped variants (tile queue): ONE block per queued invoke of one small-tile descriptor ----
// The compiler lowers tensor.pack / unpack and bias broadcasts to hundreds of unary / binary invokes
// on <= 64x64 tiles (LowerPacksAndUnpacks.cpp:45-121); the queue runs them as one launch. Same
// element arithmetic as the kernels above (bit-identical results), element-granular accesses with
// consecutive lanes on consecutive columns.
template <typename T>
__global__ __launch_bounds__(256) void unary_grouped_kernel(int op, int bc, int m, int n, int64_t ldi, int64_t ldo,
                                                            const WorkItem *__restrict__ items) {
  __shared__ T tile[64][65];
  const T *in = (const T *)items[blockIdx.x].A;
  T *out = (T *)items[blockIdx.x].C;
  const int t = threadIdx.x;
  if (op == (int)U_TRANSPOSE) {
    for (int idx = t; idx < m * n; idx += 256) {
      const int i = idx / n, j = idx - i * n;
      tile[i][j] = in[i * ldi + j];
    }
    __syncthreads();
    for (int idx = t; idx < m * n; idx += 256) {
      const int j = idx / m, i = idx - j * m;
      out[j * ldo + i] = tile[i][j];
    }
    return;
  }
  if (op == (int)U_VNNI2) {
    const int pairs = m / 2;
    for (int idx = t; idx < pairs * 2 * n; idx += 256) {
      const int r = idx / (2 * n), rem = idx - r * 2 * n;
      out[r * (2 * ldo) + rem] = in[(2 * r + (rem & 1)) * ldi + (rem >> 1)];
    }
    return;
  }
  for (int idx = t; idx < m * n; idx += 256) {
    const int i = idx / n, j = idx - i * n;
    T x = T(0);
    if (op != (int)U_ZERO) {
      x = load_operand<T, 1>(in, bc, i, j, ldi).v[0];
      if (op == (int)U_RELU) {
        const float f = Bits<T>::to_f32(x);
        x = Bits<T>::from_f32(f > 0.0f ? f : 0.0f);
      }
    }
    out[i * ldo + j] = x;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void binary_grouped_kernel(int op, int bc0, int bc1, int m, int n, int64_t ldl,
                                                             int64_t ldr, int64_t ldo,
                                                             const WorkItem *__restrict__ items) {
  const T *lhs = (const T *)items[blockIdx.x].A, *rhs = (const T *)items[blockIdx.x].B;
  T *out = (T *)items[blockIdx.x].C;
  for (int idx = threadIdx.x; idx < m * n; idx += 256) {
    const int i = idx / n, j = idx - i * n;
    const float a = Bits<T>::to_f32(load_operand<T, 1>(lhs, bc0, i, j, ldl).v[0]);
    const float b = Bits<T>::to_f32(load_operand<T, 1>(rhs, bc1, i, j, ldr).v[0]);
    float c;
    switch (op) {
    case (int)B_ADD: c = a + b; break;
    case (int)B_MUL: c = a * b; break;
    case (int)B_SUB: c = a - b; break;
    default: c = a / b; break;
    }
    out[i * ldo + j] = Bits<T>::from_f32(c);
  }
}

static inline int unary_bc(int64_t flags) {
  return (flags & UF_SCALAR) ? BC_SCALAR : (flags & UF_ROW) ? BC_ROW : (flags & UF_COL) ? BC_COL : BC_NONE;
}
static inline int binary_bc(int64_t f, int64_t row, int64_t col, int64_t sc) {
  return (f & sc) ? BC_SCALAR : (f & row) ? BC_ROW : (f & col) ? BC_COL : BC_NONE;
}

hipError_t launch_unary_grouped(const UnaryDesc &d, const WorkItem *items, int n_items, hipStream_t s) {
  if (n_items <= 0 || d.m <= 0 || d.n <= 0) return hipSuccess;
  if (d.m > 64 || d.n > 64) return hipErrorInvalidValue;
  if (d.dtype == DT_F32)
    hipLaunchKernelGGL((unary_grouped_kernel<float>), dim3((unsigned)n_items), dim3(256), 0, s, (int)d.op,
                       unary_bc(d.flags), (int)d.m, (int)d.n, d.ldi, d.ldo, items);
  else
    hipLaunchKernelGGL((unary_grouped_kernel<unsigned short>), dim3((unsigned)n_items), dim3(256), 0, s, (int)d.op,
                       unary_bc(d.flags), (int)d.m, (int)d.n, d.ldi, d.ldo, items);
  return hipGetLastError();
}

hipError_t launch_binary_grouped(const BinaryDesc &d, const WorkItem *items, int n_items, hipStream_t s) {
  if (n_items <= 0 || d.m <= 0 || d.n <= 0) return hipSuccess;
  const int bc0 = binary_bc(d.flags, BF_ROW0, BF_COL0, BF_SC0), bc1 = binary_bc(d.flags, BF_ROW1, BF_COL1, BF_SC1);
  if (d.dtype == DT_F32)
    hipLaunchKernelGGL((binary_grouped_kernel<float>), dim3((unsigned)n_items), dim3(256), 0, s, (int)d.op, bc0, bc1,
                       (int)d.m, (int)d.n, d.ldi_lhs, d.ldi_rhs, d.ldo, items);
  else
    hipLaunchKernelGGL((binary_grouped_kernel<unsigned short>), dim3((unsigned)n_items), dim3(256), 0, s, (int)d.op,
                       bc0, bc1, (int)d.m, (int)d.n, d.ldi_lhs, d.ldi_rhs, d.ldo, items);
  return hipGetLastError();
}

static inline int grid_for(int64_t total) {
  int64_t blocks = (total + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 16384) blocks = 16384; // one contiguous span per block (span_of_block): 8192-16384 blocks measured best on large streams
  return (int)blocks;
}
static inline bool aligned(const void *p, size_t a) { return (((uintptr_t)p) & (a - 1)) == 0; }

// Vector variant for aligned full tiles: 16 bytes per lane on BOTH global sides. A 64x64 tile is
// read with 16-byte row pieces into LDS (row pitch 64+PAD elements), then every lane gathers
// VEC elements of one input column (VEC small LDS reads) and stores them as 16 contiguous bytes
// of an output row; consecutive lanes cover one contiguous output row segment.
template <typename T, int TSR, int TSC>
__global__ __launch_bounds__(256) void transpose_vec_kernel(int64_t m, int64_t n, int64_t ldi, int64_t ldo,
                                                            const T *__restrict__ in, T *__restrict__ out) {
  // TSR x TSC tile of the INPUT (rows x columns): row pieces of TSC elements are read, row pieces of TSR elements are written.
  // 128 x 128 when the shape allows (bf16 with 64-wide tiles moved 128-byte pieces 32 KiB apart: 4.3 TB/s at 16384^2; f32: see the
  // launcher), else 64 x 64.
  constexpr int VEC = 16 / sizeof(T), TPR = TSC / VEC, RPP = 256 / TPR; // threads per input row piece, input rows per pass
  constexpr int TPO = TSR / VEC;                                        // threads per output row piece
  constexpr int PITCH = TSC + (sizeof(T) == 4 ? 1 : 2);
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[]; // TSR * PITCH elements (dynamic: 128 x 128 f32 is 64.5 KiB)
  T *tile = (T *)tile_raw;
  // (a G x G super-block tile order was measured for DRAM page / TLB locality at 16384^2: no effect)
  const int64_t tiles_n = n / TSC;
  // DIAGONAL tile order: blocks that run at the same time (consecutive ids) then write to different column offsets
  // of the output AND different row groups - with the row-major order they all wrote row segments a whole number of
  // tile rows (a power of two of bytes for the usual shapes) apart, i.e. into the same memory channels. Measured
  // (profiles/r02_eltwise_bw.txt): f32 8192^2 4.33 -> 4.97 TB/s, 16384^2 4.53 -> 4.98, bf16 16384^2 4.14 -> 4.60.
  const int64_t ti = blockIdx.x / tiles_n, tj = (blockIdx.x % tiles_n + ti) % tiles_n;
  const int64_t i0 = ti * TSR, j0 = tj * TSC;
  const int t = threadIdx.x;
#pragma unroll
  for (int r = t / TPR; r < TSR; r += RPP) {
    const Pack<T, VEC> v = *(const Pack<T, VEC> *)(in + (i0 + r) * ldi + j0 + (t % TPR) * VEC);
#pragma unroll
    for (int e = 0; e < VEC; ++e) tile[r * PITCH + (t % TPR) * VEC + e] = v.v[e];
  }
  __syncthreads();
  // output row = input column c (TSC of them), VEC consecutive output columns = input rows r0..r0+VEC-1
#pragma unroll
  for (int q = t; q < TSC * TPO; q += 256) {
    const int c = q / TPO, r0 = (q % TPO) * VEC; // TPO consecutive lanes = one contiguous output row segment
    Pack<T, VEC> v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) v.v[e] = tile[(r0 + e) * PITCH + c];
    store_wt<T, VEC>(out + (j0 + c) * ldo + i0 + r0, v); // (write-through: see store_wt; nontemporal stores measured 4.8 -> 4.3 TB/s here)
  }
}

template <typename T>
static hipError_t launch_unary_t(const UnaryDesc &d, const void *in, float scalar, bool use_scalar, void *out,
                                 hipStream_t s) {
  constexpr int V = 16 / sizeof(T);
  const int op = (int)d.op;
  int bc = BC_NONE;
  if (d.flags & UF_SCALAR) bc = BC_SCALAR;
  else if (d.flags & UF_ROW) bc = BC_ROW;
  else if (d.flags & UF_COL) bc = BC_COL;
  if (op == (int)U_TRANSPOSE) {
    const int64_t tiles = ((d.m + 63) / 64) * ((d.n + 63) / 64);
    const bool vec_ok = d.ldi % V == 0 && d.ldo % V == 0 && aligned(in, 16) && aligned(out, 16);
    static const int alt = [] { const char *e = getenv("TPP_HIP_TRANSPOSE_TILE"); return e ? atoi(e) : 0; }(); // A/B runs
#define TPP_TRANSPOSE_LAUNCH(R, C)                                                                                        \
  do {                                                                                                                    \
    constexpr int lds_ = (R) * ((C) + (sizeof(T) == 4 ? 1 : 2)) * (int)sizeof(T);                                         \
    static std::atomic<unsigned long long> set_{0};                                                                       \
    if (hipError_t e_ = ensure_dynamic_lds((const void *)transpose_vec_kernel<T, R, C>, lds_, set_); e_ != hipSuccess) return e_; \
    hipLaunchKernelGGL((transpose_vec_kernel<T, R, C>), dim3((unsigned)((d.m / (R)) * (d.n / (C)))), dim3(256), lds_, s, d.m, d.n, d.ldi, \
                       d.ldo, (const T *)in, (T *)out);                                                                   \
  } while (0)
    // 128 x 128 tiles when the shape allows (f32: 512-byte row pieces on both sides, 64.5 KiB of LDS, two workgroups per CU - 8192^2
    // 4.9 -> 5.4 TB/s, 16384^2 5.2 -> 5.4 against 64 x 64; 64 x 128 / 128 x 64 landed in between; bf16 tiles of 256 x 128, 128 x 256
    // and 256 x 256 measured 5-35 % SLOWER than 128 x 128). TPP_HIP_TRANSPOSE_TILE=64 forces the small tile for A/B runs.
    if (vec_ok && alt != 64 && d.m % 128 == 0 && d.n % 128 == 0) TPP_TRANSPOSE_LAUNCH(128, 128);
    else if (vec_ok && d.m % 64 == 0 && d.n % 64 == 0) TPP_TRANSPOSE_LAUNCH(64, 64);
#undef TPP_TRANSPOSE_LAUNCH
    else
      hipLaunchKernelGGL((transpose_kernel<T>), dim3((unsigned)tiles), dim3(256), 0, s, d.m, d.n, d.ldi, d.ldo,
                         (const T *)in, (T *)out);
    return hipGetLastError();
  }
  const bool reads = op != (int)U_ZERO && !use_scalar;
  bool vec = d.n % V == 0 && d.ldo % V == 0 && aligned(out, 16);
  if (reads && (bc == BC_NONE || bc == BC_COL)) vec = vec && aligned(in, 16) && (bc == BC_COL || d.ldi % V == 0);
  if (vec) {
    // contiguous operands are one long row: no index division in the kernel
    const bool flat = d.ldo == d.n && (!reads || (bc == BC_NONE && d.ldi == d.n) || bc == BC_SCALAR);
    const int64_t m_ = flat ? 1 : d.m, n_ = flat ? d.m * d.n : d.n;
    hipLaunchKernelGGL((unary_kernel<T, V>), dim3(grid_for(d.m * (d.n / V))), dim3(256), 0, s, op, bc, m_, n_,
                       flat ? n_ : d.ldi, flat ? n_ : d.ldo, (const T *)in, (T *)out, scalar, (int)use_scalar);
  } else
    hipLaunchKernelGGL((unary_kernel<T, 1>), dim3(grid_for(d.m * d.n)), dim3(256), 0, s, op, bc, d.m, d.n, d.ldi,
                       d.ldo, (const T *)in, (T *)out, scalar, (int)use_scalar);
  return hipGetLastError();
}

hipError_t launch_unary(const UnaryDesc &d, const void *in, float scalar, bool use_scalar, void *out,
                        hipStream_t s) {
  if (d.m <= 0 || d.n <= 0) return hipSuccess;
  if (d.op == U_VNNI2) {
    const bool vec = d.n % 8 == 0 && d.ldi % 8 == 0 && (2 * d.ldo) % 8 == 0 && aligned(in, 16) && aligned(out, 16);
    static const int piece = [] {
      const char *e = getenv("TPP_HIP_PACK_PIECE");
      return e ? atoi(e) : 4;
    }();
    if (vec && piece == 4 && d.m / 2 <= 65535 && d.n / 4 < (1 << 30))
    {
      // store / load policy by size, measured (profiles/r04_vnni2_pack_ab.txt, same box): nontemporal loads + write-through (sc1)
      // stores win up to 4096^2 (64 MiB moved: 10.5 -> 9.7 us) and lose beyond (8192^2: 37.4 -> 38.7 us); nontemporal STORES lose
      // everywhere (-6 .. -11 %). TPP_HIP_PACK_POLICY=0|1|2 forces one (A/B runs).
      static const int forced = [] {
        const char *e = getenv("TPP_HIP_PACK_POLICY");
        return e ? atoi(e) : -1;
      }();
      const int policy = forced >= 0 ? forced : ((double)d.m * (double)d.n * 4.0 <= 64.0 * 1024 * 1024 ? 2 : 0);
      const dim3 g((unsigned)((d.n / 4 + 255) / 256), (unsigned)(d.m / 2));
      if (policy == 1) hipLaunchKernelGGL(vnni2_rows4_kernel<1>, g, dim3(256), 0, s, (int)(d.n / 4), d.ldi, d.ldo, (const unsigned short *)in, (unsigned short *)out);
      else if (policy == 2) hipLaunchKernelGGL(vnni2_rows4_kernel<2>, g, dim3(256), 0, s, (int)(d.n / 4), d.ldi, d.ldo, (const unsigned short *)in, (unsigned short *)out);
      else hipLaunchKernelGGL(vnni2_rows4_kernel<0>, g, dim3(256), 0, s, (int)(d.n / 4), d.ldi, d.ldo, (const unsigned short *)in, (unsigned short *)out);
    }
    else if (vec && d.m / 2 <= 65535 && d.n / 8 < (1 << 30))
      hipLaunchKernelGGL(vnni2_rows_kernel, dim3((unsigned)((d.n / 8 + 255) / 256), (unsigned)(d.m / 2)), dim3(256), 0, s,
                         (int)(d.n / 8), d.ldi, d.ldo, (const unsigned short *)in, (unsigned short *)out);
    else if (vec)
      hipLaunchKernelGGL((vnni2_kernel<8>), dim3(grid_for((d.m / 2) * (d.n / 8))), dim3(256), 0, s, d.m, d.n, d.ldi,
                         d.ldo, (const unsigned short *)in, (unsigned short *)out);
    else
      hipLaunchKernelGGL((vnni2_kernel<1>), dim3(grid_for((d.m / 2) * d.n)), dim3(256), 0, s, d.m, d.n, d.ldi, d.ldo,
                         (const unsigned short *)in, (unsigned short *)out);
    return hipGetLastError();
  }
  if (d.dtype == DT_F32) return launch_unary_t<float>(d, in, scalar, use_scalar, out, s);
  return launch_unary_t<unsigned short>(d, in, scalar, use_scalar, out, s);
}

template <typename T>
static hipError_t launch_binary_t(const BinaryDesc &d, const void *lhs, const void *rhs, void *out, hipStream_t s) {
  constexpr int V = 16 / sizeof(T);
  auto mode = [](int64_t f, int64_t row, int64_t col, int64_t sc) {
    return (f & sc) ? BC_SCALAR : (f & row) ? BC_ROW : (f & col) ? BC_COL : BC_NONE;
  };
  const int bc0 = mode(d.flags, BF_ROW0, BF_COL0, BF_SC0), bc1 = mode(d.flags, BF_ROW1, BF_COL1, BF_SC1);
  auto ok = [&](const void *p, int bc, int64_t ld) {
    if (bc == BC_ROW || bc == BC_SCALAR) return true;
    return aligned(p, 16) && (bc == BC_COL || ld % V == 0);
  };
  const bool vec = d.n % V == 0 && d.ldo % V == 0 && aligned(out, 16) && ok(lhs, bc0, d.ldi_lhs) && ok(rhs, bc1, d.ldi_rhs);
  if (vec) {
    auto contig = [&](int bc, int64_t ld) { return bc == BC_SCALAR || (bc == BC_NONE && ld == d.n); };
    const bool flat = d.ldo == d.n && contig(bc0, d.ldi_lhs) && contig(bc1, d.ldi_rhs); // one long row: no index division
    const int64_t m_ = flat ? 1 : d.m, n_ = flat ? d.m * d.n : d.n;
    hipLaunchKernelGGL((binary_kernel<T, V>), dim3(grid_for(d.m * (d.n / V))), dim3(256), 0, s, (int)d.op, bc0, bc1,
                       m_, n_, flat ? n_ : d.ldi_lhs, flat ? n_ : d.ldi_rhs, flat ? n_ : d.ldo, (const T *)lhs, (const T *)rhs, (T *)out);
  } else
    hipLaunchKernelGGL((binary_kernel<T, 1>), dim3(grid_for(d.m * d.n)), dim3(256), 0, s, (int)d.op, bc0, bc1, d.m,
                       d.n, d.ldi_lhs, d.ldi_rhs, d.ldo, (const T *)lhs, (const T *)rhs, (T *)out);
  return hipGetLastError();
}

hipError_t launch_binary(const BinaryDesc &d, const void *lhs, const void *rhs, void *out, hipStream_t s) {
  if (d.m <= 0 || d.n <= 0) return hipSuccess;
  if (d.dtype == DT_F32) return launch_binary_t<float>(d, lhs, rhs, out, s);
  return launch_binary_t<unsigned short>(d, lhs, rhs, out, s);
}

} // namespace tpp

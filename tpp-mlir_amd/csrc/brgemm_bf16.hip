// brgemm_bf16.hip - bf16 batch-reduce GEMM with a VNNI-2 B operand for gfx950 on
// v_mfma_f32_32x32x16_bf16 (f32 accumulate, one RNE rounding at the store).
//
// Semantics: as brgemm_f32.hip, with A row-major bf16 [m][k] and B in the layout the
// reference compiler packs it to: [k/2][n][2] (VNNIUtils.cpp:75-77; ldb is the row
// stride / 2, ConvertLinalgToXsmm.cpp:1144). A (k, k+1) pair of one column is one
// dword, so 4 consecutive pair-rows of a column are exactly the 8 consecutive k a
// lane feeds to one MFMA.
//
// Structure: workgroup tile (32*WM*TM) x (32*WN*TN), WM*WN waves, each wave TM x TN
// accumulator tiles of 32x32; K chunk = 64; same 3-slot LDS ring / one mid-chunk
// barrier pipeline as the f32 kernel.
//   A in LDS: [row][8 x 16 B], 16-byte column index XOR ((row>>1)&7)  -> the per-lane
//             ds_read_b128 (8 consecutive k) is bank-conflict free.
//   B in LDS: transposed while staging to [g = pair-row/4][column][4 dwords] (group
//             rows padded by one 16-B slot) so a lane's whole B fragment is ONE
//             ds_read_b128; the 4x4 dword transpose happens in registers between the
//             global_load_dwordx4 and the ds_write_b128.
#include "gemm_common.h"
#include "xsmm_desc.h"
#include <type_traits>

namespace tpp {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

#ifndef TPP_ABLATE
#define TPP_ABLATE 0
#endif
constexpr int HABL_NO_GLOAD = 1, HABL_NO_SWRITE = 2, HABL_NO_BARRIER = 4, HABL_NO_FRAG = 8, HABL_NO_TRANSPOSE = 16, HABL_STAMP = 32;

constexpr int BKH = 64;     // k per chunk
constexpr int NSTAGE_H = 3;
constexpr int NSET_H = 3;  // staging register sets (chunks of global loads in flight per lane)

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(64 * WM * WN) void brgemm_bf16_fast(GemmArgs p) {
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, NT = 64 * WM * WN;
  constexpr int A_STAGE = BM * BKH * 2;            // bytes
  constexpr int B_GROW = (BN + 1) * 16;            // bytes per pair-row group (padded)
  constexpr int B_STAGE = 8 * B_GROW;              // bytes
  constexpr int A_ITEMS = BM * 8, B_ITEMS = 8 * (BN / 4);
  constexpr int LA = (A_ITEMS + NT - 1) / NT, LB = (B_ITEMS + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  unsigned char *As = smem_h;
  unsigned char *Bs = smem_h + NSTAGE_H * A_STAGE;

  unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
  if (TPP_ABLATE & HABL_STAMP) { stamp[0] = __builtin_readcyclecounter(); stamp[4] = wall_clock64(); }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  // tile from the 3-D grid (see brgemm_f32.hip): (8, bn, bm) XCD-blocked or (1, tiles_n, tiles_m)
  const int tm = (int)(blockIdx.x >> 1) * p.tiles_m + (int)blockIdx.z;
  const int tn = (int)(blockIdx.x & 1) * p.tiles_n + (int)blockIdx.y;
  const int m0 = tm * BM, n0 = tn * BN;

  const unsigned short *__restrict__ A = (const unsigned short *)p.A;
  const unsigned short *__restrict__ B = (const unsigned short *)p.B;
  unsigned short *__restrict__ C = (unsigned short *)p.C;
  const int kchunks = p.k / BKH;
  const int T = p.br * kchunks;

  f32x16 acc[TM][TN]; // initialised after the first loads are on their way

  // staging: A pieces are 16-byte row segments; a B piece is a 4x4 dword block (4 pair-rows
  // x 4 columns) that is transposed in registers on its way to LDS. Buffer loads with
  // per-lane constant byte offsets; the wave-uniform panel bases advance by scalar adds.
  u32x4 ra[NSET_H][LA], rb[NSET_H][LB][4];
  unsigned voffA[LA], voffB[LB];
#pragma unroll
  for (int u = 0; u < LA; ++u) {
    const int q = tid + u * NT, row = q >> 3, c = q & 7;
    voffA[u] = (unsigned)((row * (int)p.lda + 8 * c) * 2);
  }
#pragma unroll
  for (int u = 0; u < LB; ++u) {
    const int q = tid + u * NT, g = q & 7, jq = q >> 3;
    voffB[u] = (unsigned)((4 * g * 2 * (int)p.ldb + 8 * jq) * 2);
  }
  const unsigned rowB = (unsigned)(2 * (int)p.ldb * 2); // bytes between pair-rows
  const unsigned short *gA = A + (int64_t)m0 * p.lda, *gB = B + 2 * (int64_t)n0;
  int kc = 0;
  const int64_t dA_wrap = p.stride_a - (int64_t)(kchunks - 1) * BKH;
  const int64_t dB_in = (int64_t)(BKH / 2) * 2 * p.ldb, dB_wrap = p.stride_b - (int64_t)(kchunks - 1) * dB_in;
  constexpr int NLOAD = LA + 4 * LB, NWRITE = LA + 4 * LB; // instruction counts per chunk per lane
  auto gload_item = [&](int set, int it) __attribute__((always_inline)) {
    if (it < LA) {
      if (A_ITEMS % NT == 0 || tid + it * NT < A_ITEMS) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)gA, 0, 0x7fffffff, 0x00020000);
        ra[set][it] = __builtin_amdgcn_raw_buffer_load_b128(r, voffA[it], 0, 0);
      }
    } else {
      const int u = (it - LA) >> 2, rr = (it - LA) & 3;
      if (B_ITEMS % NT == 0 || tid + u * NT < B_ITEMS) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)gB, 0, 0x7fffffff, 0x00020000);
        rb[set][u][rr] = __builtin_amdgcn_raw_buffer_load_b128(r, voffB[u] + rr * rowB, 0, 0);
      }
    }
  };
  auto swrite_item = [&](int stage, int it) __attribute__((always_inline)) {
    unsigned char *as = As + stage * A_STAGE, *bs = Bs + stage * B_STAGE;
    if (it < LA) {
      const int q = tid + it * NT, row = q >> 3, c = q & 7;
      if (A_ITEMS % NT == 0 || q < A_ITEMS) *(u32x4 *)(as + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = ra[stage][it];
    } else {
      const int u = (it - LA) >> 2, e = (it - LA) & 3;
      const int q = tid + u * NT, g = q & 7, jq = q >> 3;
      if (B_ITEMS % NT == 0 || q < B_ITEMS) { // column 4*jq+e: its 4 pair-rows, transposed in registers
        u32x4 v = {rb[stage][u][0][e], rb[stage][u][1][e], rb[stage][u][2][e], rb[stage][u][3][e]};
        if (TPP_ABLATE & HABL_NO_TRANSPOSE) v = rb[stage][u][e];
        *(u32x4 *)(bs + g * B_GROW + ((4 * jq + e) << 4)) = v;
      }
    }
  };
  bf16x8_t af[2][TM], bfr[2][TN];
  auto frag_load = [&](int buf, int stage, int ks) __attribute__((always_inline)) {
    const unsigned char *as = As + stage * A_STAGE;
    const unsigned char *bs = Bs + stage * B_STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = (wm * TM + i) * 32 + li;
      af[buf][i] = *(const bf16x8_t *)(as + row * 128 + (((2 * ks + lh) ^ ((row >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = (wn * TN + j) * 32 + li;
      bfr[buf][j] = *(const bf16x8_t *)(bs + (2 * ks + lh) * B_GROW + (col << 4));
    }
  };
  // one chunk: 4 k-steps of TM*TN MFMAs; same pipeline as the f32 kernel (ring slot known
  // at compile time, writes of chunk t+1 in the first half + ONE barrier, loads of chunk
  // t+2 in the second half, fragments of step q+1 read while step q multiplies).
  constexpr int SLOTS = 2 * TM * TN; // MFMA slots per half chunk
  auto gadvance = [&]() __attribute__((always_inline)) {
    if (++kc == kchunks) {
      kc = 0;
      gA += dA_wrap;
      gB += dB_wrap;
    } else {
      gA += BKH;
      gB += dB_in;
    }
  };
  auto chunk = [&](auto stage_c, auto has_next, auto has_load) __attribute__((always_inline)) {
    constexpr int STAGE = decltype(stage_c)::value, NSTG = (STAGE + 1) % NSTAGE_H;
    constexpr bool HAS_NEXT = decltype(has_next)::value, HAS_LOAD = decltype(has_load)::value;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int cur = q & 1, nxt = cur ^ 1;
      if (!(TPP_ABLATE & HABL_NO_FRAG)) {
        if (q + 1 < 4) frag_load(nxt, STAGE, q + 1);
        else if (HAS_NEXT) frag_load(nxt, NSTG, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][i], bfr[cur][j], acc[i][j], 0, 0, 0);
          const int slot = (q & 1) * TM * TN + i * TN + j; // slot inside the half chunk
          if (HAS_LOAD && q == 1 && i == TM - 1 && j == TN - 1) { // next panel base: scalar work
            if (++kc == kchunks) {
              kc = 0;
              gA += dA_wrap;
              gB += dB_wrap;
            } else {
              gA += BKH;
              gB += dB_in;
            }
          }
#pragma unroll
          for (int it = 0; it < NWRITE; ++it)
            if (HAS_NEXT && !(TPP_ABLATE & HABL_NO_SWRITE) && q < 2 && (it * SLOTS) / NWRITE == slot) swrite_item(NSTG, it);
#pragma unroll
          for (int it = 0; it < NLOAD; ++it)
            if (HAS_LOAD && !(TPP_ABLATE & HABL_NO_GLOAD) && q >= 2 && (it * SLOTS) / NLOAD == slot) gload_item(NSTG, it);
          __builtin_amdgcn_sched_barrier(0);
        }
      if (q == 1 && !(TPP_ABLATE & HABL_NO_BARRIER)) __syncthreads();
    }
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  static_assert(NSET_H == NSTAGE_H && NSTAGE_H == 3, "schedule written for 3 slots / 3 sets");

  // prologue: chunk 0 -> set 0 -> slot 0; chunks 1, 2, 3 -> sets 1, 2, 0 (in flight)
  if (T > 0) { // first thing the kernel does: get chunk 0 moving
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) gload_item(0, it);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  // C tile through buffer ops: tile base in the descriptor, per-lane offset constant, the
  // (tile row i, register r) row as a scalar offset, the tile column j as an immediate
  const int ccol0 = n0 + wn * 32 * TN + li;
  const __amdgpu_buffer_rsrc_t rsrcC =
      __builtin_amdgcn_make_buffer_rsrc((void *)(C + (int64_t)m0 * p.ldc + n0), 0, 0x7fffffff, 0x00020000);
  const unsigned voffC = (unsigned)(((wm * 32 * TM + 4 * lh) * (int)p.ldc + wn * 32 * TN + li) * 2);
  const unsigned ldcb = (unsigned)((int)p.ldc * 2);
  if (!(p.ep & EP_BETA0)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc[i][j][r] = bf16_bits_to_f32(__builtin_amdgcn_raw_buffer_load_b16(
              rsrcC, voffC + 64 * j, (unsigned)(32 * i + (r & 3) + 8 * (r >> 2)) * ldcb, 0));
  }

  if (T > 0) {
#pragma unroll
    for (int c = 1; c <= NSET_H; ++c) {
      if (c == NSET_H) {
#pragma unroll
        for (int it = 0; it < NWRITE; ++it) swrite_item(0, it);
      }
      if (c < T) {
        gadvance();
#pragma unroll
        for (int it = 0; it < NLOAD; ++it) gload_item(c % NSET_H, it);
      }
    }
  }
  __syncthreads();
  if (T > 0) frag_load(0, 0, 0);
  if (TPP_ABLATE & HABL_STAMP) stamp[1] = __builtin_readcyclecounter();
  int t = 0;
  for (; t + 2 + NSET_H + 1 < T; t += 3) {
    chunk(S0{}, yes{}, yes{});
    chunk(S1{}, yes{}, yes{});
    chunk(S2{}, yes{}, yes{});
  }
  auto tail = [&](auto stage_c) __attribute__((always_inline)) {
    const int left = T - t;
    if (left > NSET_H + 1) chunk(stage_c, yes{}, yes{});
    else if (left >= 2) chunk(stage_c, yes{}, no{});
    else chunk(stage_c, no{}, no{});
    ++t;
  };
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (t < T) tail(S0{});
    if (t < T) tail(S1{});
    if (t < T) tail(S2{});
  }

  if (TPP_ABLATE & HABL_STAMP) stamp[2] = __builtin_readcyclecounter();
  if (TPP_ABLATE & ~HABL_STAMP) {
#pragma unroll
    for (int st = 0; st < NSET_H; ++st) {
#pragma unroll
      for (int u = 0; u < LA; ++u) asm volatile("" ::"v"(ra[st][u]));
#pragma unroll
      for (int u = 0; u < LB; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(rb[st][u][r]));
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(af[0][i]), "v"(af[1][i]));
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bfr[0][j]), "v"(bfr[1][j]));
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = ccol0 + 32 * j;
    const float bias = (p.ep & EP_BIAS) ? bf16_bits_to_f32(((const unsigned short *)p.D)[col]) : 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] + bias;
        if (p.ep & EP_RELU) v = v > 0.0f ? v : 0.0f;
        __builtin_amdgcn_raw_buffer_store_b16(f32_to_bf16_bits(v), rsrcC, voffC + 64 * j,
                                              (unsigned)(32 * i + (r & 3) + 8 * (r >> 2)) * ldcb, 0);
      }
  }
  if ((TPP_ABLATE & HABL_STAMP) && p.D && !(p.ep & EP_BIAS) && tid == 0) {
    stamp[3] = __builtin_readcyclecounter();
    const size_t lin = blockIdx.x + gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
    unsigned long long *dbg = (unsigned long long *)p.D + lin * 8;
    for (int e = 0; e < 5; ++e) dbg[e] = stamp[e];
    dbg[5] = wall_clock64();
  }
}

template <int WM, int WN, int TM, int TN>
static hipError_t launch_bf16(const GemmArgs &a, hipStream_t s) {
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, NT = 64 * WM * WN;
  constexpr size_t lds = (size_t)NSTAGE_H * (BM * BKH * 2 + 8 * (BN + 1) * 16);
  static bool attr_set = false;
  auto kern = brgemm_bf16_fast<WM, WN, TM, TN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  GemmArgs args = a;
  const int tiles_m = a.m / BM, tiles_n = a.n / BN;
  dim3 grid;
  if ((tiles_m & 3) == 0 && (tiles_n & 1) == 0 && tiles_m / 4 <= 65535 && tiles_n / 2 <= 65535) {
    args.tiles_m = tiles_m / 4; // XCD-blocked: 4 (M) x 2 (N) XCD blocks of tiles_m/4 x tiles_n/2 tiles
    args.tiles_n = tiles_n / 2;
    grid = dim3(8, args.tiles_n, args.tiles_m);
  } else {
    args.tiles_m = args.tiles_n = 0;
    if (tiles_m > 65535 || tiles_n > 65535) return hipErrorInvalidValue;
    grid = dim3(1, tiles_n, tiles_m);
  }
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, args);
  return hipGetLastError();
}

bool bf16_fast_eligible(const GemmDesc &d) {
  if (d.dtype != DT_BF16 || !d.vnni_b) return false;
  if (d.k <= 0 || d.k % BKH) return false;
  if (d.m % 64 || d.n % 64) return false;
  if ((d.lda & 7) || (d.ldb & 3) || (d.stride_a & 7) || (d.stride_b & 7)) return false;
  if (d.lda >= (1 << 22) || d.ldb >= (1 << 21) || d.ldc >= (1 << 22)) return false; // 32-bit lane offsets
  return true;
}

hipError_t launch_gemm_bf16_fast(const GemmDesc &d, const GemmArgs &a, hipStream_t s) {
  const int64_t t128 = (d.m % 128 == 0 && d.n % 128 == 0) ? (d.m / 128) * (d.n / 128) : 0;
  if (t128 >= 192) return launch_bf16<2, 2, 2, 2>(a, s);
  return launch_bf16<2, 2, 1, 1>(a, s);
}

} // namespace tpp

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
#include <stdlib.h>

namespace tpp {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// Timing-only ablation masks: SIDE builds only (tools/sessions/ablate_bf16.sh compiles this file with -DTPP_ABLATE=mask into
// build/libabl_*.so); the shipped library is built with 0 and every `if (TPP_ABLATE & ...)` below folds away.
#ifndef TPP_ABLATE
#define TPP_ABLATE 0
#endif
constexpr int HABL_NO_GLOAD = 1, HABL_NO_SWRITE = 2, HABL_NO_BARRIER = 4, HABL_NO_FRAG = 8, HABL_NO_TRANSPOSE = 16,
              HABL_NO_BFRAG = 64 /* dma128: no B fragment reads */, HABL_NO_BDMA = 128 /* dma128: the B loader wave fetches nothing */,
              HABL_NO_ADMA = 256 /* dma128: the A loader wave fetches nothing */, HABL_LOADERS_ONLY = 512 /* dma128: the MFMA waves leave at once */;

constexpr int BKH = 64;     // k per chunk
constexpr int NSTAGE_H = 3;
constexpr int NSET_H = 3;  // staging register sets (chunks of global loads in flight per lane)

// items != nullptr: grouped mode (tile queue), grid (items, tiles_n, tiles_m) - see brgemm_f32.hip
// VF = 4 (round 5; grouped launches of --vnni=4 tile invokes): B is [k/4][n][4] - a B piece is still a 4-column x 8-k block fetched
// with four 16-byte loads (k-group rows 2g, 2g + 1 x column pairs) and written as four per-column 16-byte LDS pieces; only the
// source offsets and the register selection of the in-register transpose differ, the LDS image and everything behind it are the same.
template <int WM, int WN, int TM, int TN, int VF = 2>
__global__ __launch_bounds__(64 * WM * WN) void brgemm_bf16_fast(GemmArgs p, const WorkItem *__restrict__ items) {
  if (items) {
    const WorkItem it = items[blockIdx.x];
    p.A = it.A; p.B = it.B; p.C = it.C; p.D = it.D; p.br = (int)it.br;
  }
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, NT = 64 * WM * WN;
  constexpr int A_STAGE = BM * BKH * 2;            // bytes
  constexpr int B_GROW = (BN + 1) * 16;            // bytes per pair-row group (padded)
  constexpr int B_STAGE = 8 * B_GROW;              // bytes
  constexpr int A_ITEMS = BM * 8, B_ITEMS = 8 * (BN / 4);
  constexpr int LA = (A_ITEMS + NT - 1) / NT, LB = (B_ITEMS + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  unsigned char *As = smem_h;
  unsigned char *Bs = smem_h + NSTAGE_H * A_STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  // tile from the 3-D grid (see brgemm_f32.hip): (8, bn, bm) XCD-blocked or (1, tiles_n, tiles_m)
  const int tm = items ? (int)blockIdx.z : (int)(blockIdx.x >> 1) * p.tiles_m + (int)blockIdx.z;
  const int tn = items ? (int)blockIdx.y : (int)(blockIdx.x & 1) * p.tiles_n + (int)blockIdx.y;
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
    voffB[u] = VF == 4 ? (unsigned)((2 * g * 4 * (int)p.ldb + 16 * jq) * 2)  // k-group row 2g, columns 4 jq ..: 8 bytes per column
                       : (unsigned)((4 * g * 2 * (int)p.ldb + 8 * jq) * 2);
  }
  const unsigned rowB = VF == 4 ? (unsigned)(4 * (int)p.ldb * 2) : (unsigned)(2 * (int)p.ldb * 2); // bytes between k-group rows / pair-rows
  const unsigned short *gA = A + (int64_t)m0 * p.lda, *gB = B + VF * (int64_t)n0;
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
        // VNNI-2: pair-row 4g + rr, 4 columns; VNNI-4: k-group row 2g + rr / 2, column pair rr % 2
        rb[set][u][rr] = __builtin_amdgcn_raw_buffer_load_b128(r, VF == 4 ? voffB[u] + (rr >> 1) * rowB + (rr & 1) * 16u : voffB[u] + rr * rowB, 0, 0);
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
        if constexpr (VF == 4) { // column 4 jq + e = column e % 2 of pair e / 2: its k 0..3 from row 2g, k 4..7 from row 2g + 1
          const int c2 = e >> 1, o = 2 * (e & 1);
          v = u32x4{rb[stage][u][c2][o], rb[stage][u][c2][o + 1], rb[stage][u][2 + c2][o], rb[stage][u][2 + c2][o + 1]};
        }
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
          // operands swapped (D = B^T A^T tile): lane&31 = output ROW, registers = columns - see the epilogue
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cur][j], af[cur][i], acc[i][j], 0, 0, 0);
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
  // Output fragment layout (MFMA operands are swapped: the B fragment is the "A" operand):
  // lane (li, lh) owns output row wm*32*TM + 32*i + li of tile (i, j) and, in registers
  // 4g..4g+3, the four consecutive columns 32*j + 8*g + 4*lh + (0..3). The C tile is
  // addressed through a buffer descriptor (tile base) + per-lane byte offset (row) + an
  // immediate/scalar column offset.
  const __amdgpu_buffer_rsrc_t rsrcC =
      __builtin_amdgcn_make_buffer_rsrc((void *)(C + (int64_t)m0 * p.ldc + n0), 0, 0x7fffffff, 0x00020000);
  const unsigned ldcb = (unsigned)((int)p.ldc * 2);
  const unsigned voffC = (unsigned)(wm * 32 * TM + li) * ldcb + (unsigned)((wn * 32 * TN + 4 * lh) * 2);
  if (!(p.ep & EP_BETA0)) { // beta = 1: start the accumulator chain from C (8-byte loads, 4 columns each)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 c2 = __builtin_amdgcn_raw_buffer_load_b64(rsrcC, voffC + (unsigned)((32 * j + 8 * g) * 2),
                                                              (unsigned)(32 * i) * ldcb, 0);
          acc[i][j][4 * g + 0] = __uint_as_float(c2[0] << 16);
          acc[i][j][4 * g + 1] = __uint_as_float(c2[0] & 0xffff0000u);
          acc[i][j][4 * g + 2] = __uint_as_float(c2[1] << 16);
          acc[i][j][4 * g + 3] = __uint_as_float(c2[1] & 0xffff0000u);
        }
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

  if (TPP_ABLATE) {
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
  // epilogue: (+bias[col]) (relu) -> bf16 (RNE, v_cvt_pk_bf16_f32) -> 16-byte stores. A lane
  // holds 4 consecutive columns per register quad and its partner lane (lh ^ 1) the next 4;
  // one v_permlane32_swap per dword gives the lower half-wave columns 8g..8g+7 of quad g and
  // the upper half-wave the same of quad g+1: 2 stores of 16 bytes per 32x32 tile per lane.
  typedef unsigned int u32x2e __attribute__((ext_vector_type(2)));
  const bool relu = (p.ep & EP_RELU) != 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    float bias[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x2e b2 = {0u, 0u};
      if (p.ep & EP_BIAS)
        b2 = *(const u32x2e *)((const unsigned short *)p.D + n0 + wn * 32 * TN + 32 * j + 8 * g + 4 * lh);
      bias[g][0] = __uint_as_float(b2[0] << 16);
      bias[g][1] = __uint_as_float(b2[0] & 0xffff0000u);
      bias[g][2] = __uint_as_float(b2[1] << 16);
      bias[g][3] = __uint_as_float(b2[1] & 0xffff0000u);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      unsigned int pk[4][2]; // quad g -> two dwords of packed bf16
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          float v0 = acc[i][j][4 * g + 2 * h2] + bias[g][2 * h2];
          float v1 = acc[i][j][4 * g + 2 * h2 + 1] + bias[g][2 * h2 + 1];
          if (relu) { // wave-uniform; max(x, 0) == (x > 0 ? x : 0) incl. NaN -> 0
            v0 = __builtin_fmaxf(v0, 0.0f);
            v1 = __builtin_fmaxf(v1, 0.0f);
          }
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          typedef float f32x2_t __attribute__((ext_vector_type(2)));
          const f32x2_t vv = {v0, v1};
          pk[g][h2] = __builtin_bit_cast(unsigned int, __builtin_convertvector(vv, bf16x2_t)); // one v_cvt_pk_bf16_f32 (RNE)
        }
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
        const u32x4 out = {s0[0], s1[0], s0[1], s1[1]};
        // lower half: columns 32j + 8g .. +7 ; upper half: columns 32j + 8(g+1) .. +7
        const unsigned colb = (unsigned)((32 * j + 8 * g) * 2) + (lh ? 16u - 8u : 0u);
        // (plain stores here: two lanes cover 32 contiguous bytes of a row per instruction, and write-through of PARTIAL lines is slow -
        // this kernel measured 4-7 % slower with sc1, the VNNI-2 pack kernel, 32 bytes per lane, 40 %: gemm_common.h)
        __builtin_amdgcn_raw_buffer_store_b128(out, rsrcC, voffC + colb, (unsigned)(32 * i) * ldcb, 0);
      }
    }
  }
}

template <int WM, int WN, int TM, int TN>
static hipError_t launch_bf16(const GemmArgs &a, hipStream_t s) {
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, NT = 64 * WM * WN;
  constexpr size_t lds = (size_t)NSTAGE_H * (BM * BKH * 2 + 8 * (BN + 1) * 16);
  auto kern = brgemm_bf16_fast<WM, WN, TM, TN>;
  static std::atomic<unsigned long long> lds_set{0};
  if (hipError_t e = ensure_dynamic_lds((const void *)kern, (int)lds, lds_set); e != hipSuccess) return e;
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
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, args, (const WorkItem *)nullptr);
  return hipGetLastError();
}

// grouped launch of the 64x64 bf16 tile: one workgroup per (item, 64x64 tile of the item)
hipError_t launch_bf16_grouped64(const GemmArgs &a, const WorkItem *items, int n_items, hipStream_t s) {
  constexpr size_t lds = (size_t)NSTAGE_H * (64 * BKH * 2 + 8 * (64 + 1) * 16);
  GemmArgs args = a;
  args.tiles_m = args.tiles_n = 0;
  if (a.vf == 4) { // VNNI-4 B operands (xsmm_hip_set_vnni_factor(4))
    auto kern4 = brgemm_bf16_fast<2, 2, 1, 1, 4>;
    static std::atomic<unsigned long long> lds_set4{0};
    if (hipError_t e = ensure_dynamic_lds((const void *)kern4, (int)lds, lds_set4); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern4, dim3((unsigned)n_items, a.n / 64, a.m / 64), dim3(256), lds, s, args, items);
    return hipGetLastError();
  }
  auto kern = brgemm_bf16_fast<2, 2, 1, 1>;
  static std::atomic<unsigned long long> lds_set{0};
  if (hipError_t e = ensure_dynamic_lds((const void *)kern, (int)lds, lds_set); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)n_items, a.n / 64, a.m / 64), dim3(256), lds, s, args, items);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// brgemm_bf16_dma128: 128x128 tile, 4 waves (2x2) of 64x64, every panel byte goes
// HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write, no
// transpose moves. Measured on the register-staged kernel above, the ds_write path (13
// cycles per 16 bytes per wave-instruction, 32 per chunk per CU) was the critical
// resource of the first half of every chunk; the DMA path does not use it.
//   ring : 4 slots of (A 16 KiB | B 16 KiB); the DMA of chunk t+3 is issued right after the
//          mid-chunk barrier of chunk t (2.5 chunks of MFMAs ahead of its first use).
//   A    : LDS image [128 rows][8 x 16 B], lane-linear per DMA instruction (8 rows x 128 B);
//          the bank swizzle (16-byte column XOR ((row>>1)&7)) is applied to the SOURCE
//          address, the fragment read applies the same XOR (involution).
//   B    : LDS image = the VNNI-2 rows as they are, [32 pair-rows][128 dwords]; a lane's
//          fragment (8 consecutive k of one column) is 4 dwords one row apart, read by two
//          ds_read2st64_b32 (row stride 512 B = 2 x 64 dwords).
//   sync : per chunk ONE s_waitcnt vmcnt(N) (own DMA of chunk t+1 landed) + ONE raw s_barrier,
//          in the middle of the chunk's MFMAs; never __syncthreads (it would drain the DMA).
//   out  : accumulators (lane = row, registers = columns, operands swapped) -> bias/relu ->
//          bf16 -> per-wave LDS tile -> row-contiguous 16-byte reads -> fully coalesced
//          128-byte-per-row stores.
typedef __attribute__((address_space(3))) void lds_void_t;

// NLW > 0 adds NLW LOADER waves (the first half stream A, the others B: 32 / NLW DMA instructions per
// chunk each) so the four MFMA waves never stall on vector-memory issue; the per-chunk barrier is
// shared by all six waves (the loaders wait for their own DMA to land before they arrive).
template <int NLW>
__global__ __launch_bounds__(256 + 64 * NLW) void brgemm_bf16_dma128(GemmArgs p) {
  constexpr bool LW = NLW > 0;
  constexpr int PPL = LW ? 32 / NLW : 0; // DMA instructions per loader wave per chunk
  constexpr int BM = 128, BN = 128, NSLOT = 5, TM = 2, TN = 2; // 5 x 32 KiB = all 160 KiB of LDS
  constexpr int A_SLOT = BM * BKH * 2, B_SLOT = (BKH / 2) * BN * 4, SLOT = A_SLOT + B_SLOT;
  constexpr int DMA_PER_CHUNK = 8; // per wave: 4 x 1 KiB of A + 4 x 1 KiB of B
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_d[];

  const int tid = threadIdx.x, lane = tid & 63;
#ifndef TPP_BF16_LOADERS_FIRST
#define TPP_BF16_LOADERS_FIRST 1
#endif
  // the loader waves are the FIRST hardware waves of the workgroup (waves start in order: the first chunks are requested before
  // the MFMA waves have been launched); `wave` is the role index: MFMA waves 0-3, loaders 4 .. 4 + NLW - 1
  const int hw_wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = (LW && TPP_BF16_LOADERS_FIRST) ? (hw_wave < NLW ? 4 + hw_wave : hw_wave - NLW) : hw_wave;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int tm = (int)(blockIdx.x >> 1) * p.tiles_m + (int)blockIdx.z;
  const int tn = (int)(blockIdx.x & 1) * p.tiles_n + (int)blockIdx.y;
  const int m0 = tm * BM, n0 = tn * BN;
  const unsigned short *__restrict__ A = (const unsigned short *)p.A;
  const unsigned short *__restrict__ B = (const unsigned short *)p.B;
  unsigned short *__restrict__ C = (unsigned short *)p.C;
  const int kchunks = p.k / BKH;
  const int T = p.br * kchunks;

  // per-lane source offsets of this wave's 8 DMA instructions (constant for the kernel)
  unsigned voffA[4], voffB[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int v = wave * 4 + u;
    const int row = 8 * v + (lane >> 3), pos = lane & 7;
    voffA[u] = (unsigned)(row * (int)p.lda * 2 + ((pos ^ ((row >> 1) & 7)) << 4));
    const int rowb = 2 * v + (lane >> 5);
    voffB[u] = (unsigned)(rowb * (int)p.ldb * 4 + ((lane & 31) << 4));
  }
  // panel base of the chunk being fetched (wave-uniform) and its position inside the batch element
  const unsigned short *gA = A + (int64_t)m0 * p.lda, *gB = B + 2 * (int64_t)n0;
  int kc = 0;
  const int64_t dA_wrap = p.stride_a - (int64_t)(kchunks - 1) * BKH;
  const int64_t dB_in = (int64_t)(BKH / 2) * 2 * p.ldb, dB_wrap = p.stride_b - (int64_t)(kchunks - 1) * dB_in;
  // one of this wave's 8 DMA instructions of the chunk at (gA, gB): pieces 0-3 A, 4-7 B
  auto dma_piece = [&](int slot, int u) __attribute__((always_inline)) {
    unsigned char *base = smem_d + slot * SLOT + wave * 4096;
    if (u < 4) {
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)gA, 0, 0x7fffffff, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t *)(base + u * 1024), 16, voffA[u], 0, 0, 0);
    } else {
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)gB, 0, 0x7fffffff, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t *)(base + A_SLOT + (u - 4) * 1024), 16, voffB[u - 4], 0, 0, 0);
    }
  };
  auto dma_chunk = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 8; ++u) dma_piece(slot, u);
  };
#define TPP_DMA_ADVANCE()      \
  do {                         \
    if (++kc == kchunks) {     \
      kc = 0;                  \
      gA += dA_wrap;           \
      gB += dB_wrap;           \
    } else {                   \
      gA += BKH;               \
      gB += dB_in;             \
    }                          \
  } while (0)

  if (LW && wave >= 4) {
    // ---- loader waves ------------------------------------------------------------------
    // loader wave lw of NLW: the first NLW/2 stream A, the others B; each takes a contiguous share of
    // the 16 DMA instructions of its panel (NLW = 4 halves the per-wave issue burst; measured: no gain).
    const int lw = wave - 4;
    const bool isA = lw < NLW / 2;
    const int v0 = (lw % (NLW > 1 ? NLW / 2 : 1)) * PPL; // first instruction of this wave's share
    // A: instruction v covers rows 8v..8v+7; swizzle term ((row>>1)&7) = 4*(v&1) + (lane>>4)
    const unsigned rowoffA = (unsigned)((lane >> 3) * (int)p.lda * 2);
    const unsigned voA0 = rowoffA + (unsigned)((((lane & 7) ^ (lane >> 4))) << 4);
    const unsigned voA1 = rowoffA + (unsigned)((((lane & 7) ^ (4 + (lane >> 4)))) << 4);
    const unsigned voB = (unsigned)((lane >> 5) * (int)p.ldb * 4 + ((lane & 31) << 4));
    const unsigned stepA = (unsigned)(8 * (int)p.lda * 2), stepB = (unsigned)(2 * (int)p.ldb * 4);
    auto issue = [&](int slot) __attribute__((always_inline)) {
      unsigned char *base = smem_d + slot * SLOT + (isA ? 0 : A_SLOT) + v0 * 1024;
      if (isA) {
        if (!(TPP_ABLATE & HABL_NO_ADMA)) {
          const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)gA, 0, 0x7fffffff, 0x00020000);
#pragma unroll
          for (int v = 0; v < PPL; ++v)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)(base + v * 1024), 16, (v & 1) ? voA1 : voA0, (v0 + v) * stepA, 0, 0);
        }
      } else if (!(TPP_ABLATE & HABL_NO_BDMA)) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)gB, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int v = 0; v < PPL; ++v)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t *)(base + v * 1024), 16, voB, (v0 + v) * stepB, 0, 0);
      }
      TPP_DMA_ADVANCE();
    };
    // Ring of 5 slots: chunk t+4 is requested while chunk t multiplies. Measured (profiles/r02_bf16_dma128_ablation.txt):
    // the loop waits for data, not for the LDS or the barrier - each kernel starts with cold L2s and the workgroups that
    // share a panel run in lockstep, so every chunk is a first-touch miss (~1.4 us to the Infinity Cache / HBM) and the
    // chunk time is that latency divided by the chunks in flight. Chunks 0 and 1 go first and chunk 0 is published as
    // soon as it has landed; the rest of the ring fills behind the barrier.
    if (T > 0) issue(0);
    if (T > 1) issue(1);
    if (T > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPL) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); // chunk 0 published
    if (T > 2) issue(2);
    if (T > 3) issue(3);
    for (int t = 0; t + 1 < T; ++t) {
      // chunk t+1 has landed; up to two younger chunks (t+2, t+3) may still fly
      if (t + 3 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPL) : "memory");
      else if (t + 2 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPL) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(TPP_ABLATE & HABL_NO_BARRIER)) __builtin_amdgcn_s_barrier(); // = the MFMA waves' mid-chunk barrier of chunk t: chunk t+1 published, slot of chunk t-1 retired
      if (t + 4 < T) issue((t + 4) % NSLOT);
    }
    return; // ended waves do not take part in later barriers
  }

  if (TPP_ABLATE & HABL_LOADERS_ONLY) return; // timing only: how fast can the loader waves alone fill the ring?
  f32x16 acc[TM][TN];
  constexpr int NFB = 4; // fragment buffers: step q+2 is read while step q multiplies
  bf16x8_t af[NFB][TM];
  u32x4 bw[NFB][TN]; // B fragments as dwords: a fragment is filled by two 2-dword reads
  int b_lane[TN]; // dword index of this lane's column in tile j, row 4*lh of a k-step
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    b_lane[j] = (4 * lh) * BN + (wn * TN + j) * 32 + li;
    asm volatile("" : "+v"(b_lane[j]));
  }
  // one of the 6 LDS reads of a fragment set, in the order the MFMAs need them: A0, B0 (two halves:
  // one base VGPR per column tile, made opaque, so that rows r, r+1 pair up as ds_read2st64_b32
  // straight into consecutive registers), B1, A1
  auto frag_piece = [&](int buf, int slot, int ks, int idx) __attribute__((always_inline)) {
    const unsigned char *as = smem_d + slot * SLOT;
    const unsigned int *bs = (const unsigned int *)(as + A_SLOT);
    if (idx == 0 || idx == 5) {
      const int i = idx == 0 ? 0 : 1;
      const int row = (wm * TM + i) * 32 + li;
      af[buf][i] = *(const bf16x8_t *)(as + row * 128 + (((2 * ks + lh) ^ ((row >> 1) & 7)) << 4));
    } else {
      if (TPP_ABLATE & HABL_NO_BFRAG) return;
      const int j = (idx - 1) >> 1, h = (idx - 1) & 1;
      const unsigned int *bp = bs + b_lane[j] + (8 * ks) * BN;
      bw[buf][j][2 * h] = bp[(2 * h) * BN];
      bw[buf][j][2 * h + 1] = bp[(2 * h + 1) * BN];
    }
  };
  auto frag_load = [&](int buf, int slot, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int idx = 0; idx < 6; ++idx) frag_piece(buf, slot, ks, idx);
  };
#ifndef TPP_BF16_SPREAD_READS
#define TPP_BF16_SPREAD_READS 0 // A/B measured: no gain at this tile (not issue-bound: DESIGN.md 4.2); the 256x256 kernel needs it
#endif
  // one chunk in ring slot S. H1/H2/H3: chunk t+1 / t+2 / t+3 exist. Step q multiplies the
  // fragments in buffer q (4 steps per chunk, 4 buffers) while the fragments of step q+2 are
  // read (steps 2, 3 read the first two steps of chunk t+1, published by the mid barrier), the six
  // reads spread over the step's four MFMAs (a burst from four waves at once fills the LDS queue and
  // holds up the MFMA issue behind it). The 8 DMA instructions of chunk t+3 ride one per MFMA in
  // steps 2 and 3.
  // One chunk in ring slot S. STEADY: chunks t+1 .. t+3 exist, S is a literal (every LDS address is a
  // base VGPR + immediate) and nothing is conditional. Otherwise one of the last <= 6 chunks, with
  // run-time slot and flags h1 / h2 / h3 (chunk t+1 / t+2 / t+3 exists). Exactly TWO code instances: the
  // tail used to be a chain of specialised variants, each executed once - four instruction-cache cold
  // misses (~600 cycles each) in a kernel of 16 chunks.
  // Step q multiplies the fragments in buffer q (4 steps per chunk, 4 buffers) while the fragments of
  // step q+2 are read (steps 2, 3 read the first two steps of chunk t+1, published by the mid barrier).
  // Without loader waves the 8 DMA instructions of chunk t+3 ride one per MFMA in steps 2 and 3.
  auto chunk = [&](auto steady_c, int S, bool h1, bool h2, bool h3) __attribute__((always_inline)) {
    constexpr bool STEADY = decltype(steady_c)::value;
    const bool H1 = STEADY || h1, H2 = STEADY || h2, H3 = STEADY || h3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool reads = !(TPP_ABLATE & HABL_NO_FRAG) && (q + 2 < 4 || H1);
      const int rbuf = q + 2 < 4 ? q + 2 : q - 2, rslot = q + 2 < 4 ? S : (S + 1) % NSLOT, rks = rbuf;
      if (reads && !(STEADY && TPP_BF16_SPREAD_READS)) frag_load(rbuf, rslot, rks);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bw[q][j]), af[q][i], acc[i][j], 0, 0, 0);
          if (STEADY && TPP_BF16_SPREAD_READS && reads) {
            const int m = i * TN + j; // after MFMA 0: A0, B0 lo | 1: B0 hi, B1 lo | 2: B1 hi | 3: A1
            frag_piece(rbuf, rslot, rks, m == 0 ? 0 : m == 1 ? 2 : m == 2 ? 4 : 5);
            if (m < 2) frag_piece(rbuf, rslot, rks, m == 0 ? 1 : 3);
          }
          if (!LW && STEADY && q >= 2 && !(TPP_ABLATE & HABL_NO_GLOAD)) {
            dma_piece((S + 3) % NSLOT, (q - 2) * 4 + i * TN + j); // a slot no wave reads any more (NLW = 0 keeps 3 chunks in flight)
            if (q == 3 && i == TM - 1 && j == TN - 1) TPP_DMA_ADVANCE();
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      if (q == 1 && H1) {
        // chunk t+1: this wave's DMA has landed (chunk t+2's may still fly), then everybody's
        if (!LW) {
          if (H2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!(TPP_ABLATE & HABL_NO_BARRIER)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!LW && !STEADY && H3 && !(TPP_ABLATE & HABL_NO_GLOAD)) { // the last chunks: one burst
          dma_chunk((S + 3) % NSLOT);
          TPP_DMA_ADVANCE();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;

  // prologue: chunks 0, 1, 2 in flight at once
  if (!LW) {
    if (T > 0) {
      dma_chunk(0);
      TPP_DMA_ADVANCE();
    }
    if (T > 1) {
      dma_chunk(1);
      TPP_DMA_ADVANCE();
    }
    if (T > 2) {
      dma_chunk(2);
      TPP_DMA_ADVANCE();
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  // bias: fetched here (8 bytes = 4 columns per register quad), used in the epilogue - its latency hides under the K loop
  typedef unsigned int u32x2b __attribute__((ext_vector_type(2)));
  u32x2b biasw[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      biasw[j][g] = u32x2b{0u, 0u};
      if (p.ep & EP_BIAS) biasw[j][g] = *(const u32x2b *)((const unsigned short *)p.D + n0 + wn * 64 + 32 * j + 8 * g + 4 * lh);
    }
  if (!LW) {
    if (T > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_CHUNK) : "memory");
    else if (T > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (T > 0) {
    frag_load(0, 0, 0);
    frag_load(1, 0, 1);
  }
  int t = 0;
  for (; t + 7 < T; t += NSLOT) { // steady state (t % 5 == 0 here: ring slots are literals)
    chunk(yes{}, 0, true, true, true);
    chunk(yes{}, 1, true, true, true);
    chunk(yes{}, 2, true, true, true);
    chunk(yes{}, 3, true, true, true);
    chunk(yes{}, 4, true, true, true);
  }
  for (; t < T; ++t) chunk(no{}, t % NSLOT, t + 1 < T, t + 2 < T, t + 3 < T); // the last <= 7 chunks

  if (TPP_ABLATE & HABL_NO_FRAG) {
#pragma unroll
    for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(af[0][i]), "v"(af[1][i]), "v"(af[2][i]), "v"(af[3][i]));
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bw[0][j]), "v"(bw[1][j]), "v"(bw[2][j]), "v"(bw[3][j]));
  }
  // ---- epilogue ------------------------------------------------------------------------
  // lane (li, lh) owns row 32*i + li of wave-tile row block i and, in registers 4g..4g+3 of
  // tile (i, j), columns 32*j + 8*g + 4*lh + (0..3)
  typedef unsigned int u32x2d __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t rsrcC = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(C + (int64_t)(m0 + wm * 64) * p.ldc + n0 + wn * 64), 0, 0x7fffffff, 0x00020000);
  const unsigned ldcb = (unsigned)((int)p.ldc * 2);
  if (!(p.ep & EP_BETA0)) { // beta = 1: add C before the single rounding (8-byte loads, rare path)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u32x2d c2 = __builtin_amdgcn_raw_buffer_load_b64(
              rsrcC, (unsigned)(32 * i + li) * ldcb + (unsigned)((32 * j + 8 * g + 4 * lh) * 2), 0, 0);
          acc[i][j][4 * g + 0] += __uint_as_float(c2[0] << 16);
          acc[i][j][4 * g + 1] += __uint_as_float(c2[0] & 0xffff0000u);
          acc[i][j][4 * g + 2] += __uint_as_float(c2[1] << 16);
          acc[i][j][4 * g + 3] += __uint_as_float(c2[1] & 0xffff0000u);
        }
  }
  __syncthreads(); // every wave is done with the ring: reuse it as per-wave output tiles
  const bool relu = (p.ep & EP_RELU) != 0;
  constexpr int ES = 144; // bytes per staged row: 64 bf16 + 16 B pad (conflict-free 16-byte accesses)
  unsigned char *ot = smem_d + wave * (64 * ES);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    float bias[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u32x2b b2 = biasw[j][g];
      bias[g][0] = __uint_as_float(b2[0] << 16);
      bias[g][1] = __uint_as_float(b2[0] & 0xffff0000u);
      bias[g][2] = __uint_as_float(b2[1] << 16);
      bias[g][3] = __uint_as_float(b2[1] & 0xffff0000u);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      unsigned int pk[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          float v0 = acc[i][j][4 * g + 2 * h2] + bias[g][2 * h2];
          float v1 = acc[i][j][4 * g + 2 * h2 + 1] + bias[g][2 * h2 + 1];
          if (relu) { // wave-uniform; max(x, 0) == (x > 0 ? x : 0) incl. NaN -> 0
            v0 = __builtin_fmaxf(v0, 0.0f);
            v1 = __builtin_fmaxf(v1, 0.0f);
          }
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          typedef float f32x2_t __attribute__((ext_vector_type(2)));
          const f32x2_t vv = {v0, v1};
          pk[g][h2] = __builtin_bit_cast(unsigned int, __builtin_convertvector(vv, bf16x2_t)); // one v_cvt_pk_bf16_f32 (RNE)
        }
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
        const u32x4 out = {s0[0], s1[0], s0[1], s1[1]};
        // lower half-wave: columns 32j + 8g .. +7 ; upper: 32j + 8(g+1) .. +7
        *(u32x4 *)(ot + (32 * i + li) * ES + (32 * j + 8 * g + 8 * lh) * 2) = out;
      }
    }
  }
  // same wave reads its tile back row-contiguously: 8 lanes x 16 B = one 128-byte row
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + (lane >> 3), ch = lane & 7;
    const u32x4 v = *(const u32x4 *)(ot + row * ES + ch * 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrcC, (unsigned)(lane >> 3) * ldcb + (unsigned)(ch * 16),
                                           (unsigned)(it * 8) * ldcb, C_STORE_AUX);
  }
}

template <int LW> static hipError_t launch_bf16_dma128(const GemmArgs &a, hipStream_t s) {
  constexpr size_t lds = 5 * 32768;
  static std::atomic<unsigned long long> lds_set{0};
  if (hipError_t e = ensure_dynamic_lds((const void *)brgemm_bf16_dma128<LW>, (int)lds, lds_set); e != hipSuccess) return e;
  GemmArgs args = a;
  const int tiles_m = a.m / 128, tiles_n = a.n / 128;
  dim3 grid;
  if ((tiles_m & 3) == 0 && (tiles_n & 1) == 0 && tiles_m / 4 <= 65535 && tiles_n / 2 <= 65535) {
    args.tiles_m = tiles_m / 4;
    args.tiles_n = tiles_n / 2;
    grid = dim3(8, args.tiles_n, args.tiles_m);
  } else {
    args.tiles_m = args.tiles_n = 0;
    if (tiles_m > 65535 || tiles_n > 65535) return hipErrorInvalidValue;
    grid = dim3(1, tiles_n, tiles_m);
  }
  hipLaunchKernelGGL(brgemm_bf16_dma128<LW>, grid, dim3(256 + 64 * LW), lds, s, args);
  return hipGetLastError();
}

bool bf16_fast_eligible(const GemmDesc &d) {
  if (d.dtype != DT_BF16 || !d.vnni_b || d.vnni_factor != 2) return false; // (VNNI-4 operands: bf16_vnni4_eligible, brgemm_f32.hip)
  if (d.k <= 0 || d.k % BKH) return false;
  if (d.m % 64 || d.n % 64) return false;
  if ((d.lda & 7) || (d.ldb & 3) || (d.ldc & 7) || (d.stride_a & 7) || (d.stride_b & 7)) return false;
  if (d.lda >= (1 << 22) || d.ldb >= (1 << 21) || d.ldc >= (1 << 22)) return false; // 32-bit lane offsets
  return true;
}

hipError_t launch_bf16_dma256(const GemmArgs &a, hipStream_t s); // brgemm_bf16_dma256.hip

// tile choice for an eligible descriptor: 0 = 64x64 register-staged, 1 = 128x128 DMA, 2 = 256x256 DMA.
//  * 256 x 256 (LDS / L2 traffic per flop halves: measured 1.31-1.37 vs 0.88-1.03 PFLOP/s on 4096^3 ..
//    8192^3) when its tile waves fill the 256 CUs well enough to keep that 1.4x: tiles run one per CU,
//    so a grid of t tiles takes ceil(t / 256) rounds;
//  * 128 x 128 as soon as the 64 x 64 family would need a second round of workgroups (more than 256 tiles of
//    64 x 64 = more than 64 of 128 x 128): measured (n = 1024, K = 1024) the DMA kernel takes 9.2-9.4 us from 64
//    to 256 tiles while the 64 x 64 family jumps from 9.1 to 12.8 us past one tile per CU (tools/sessions/mid_probe.py);
//  * 64 x 64 below that, so that more CUs have work.
int pick_bf16_tile(const GemmDesc &d) {
  constexpr int64_t t256_min = 240, t128_min = 65; // crossovers measured in profiles/r01_sweep_shapes.txt
  const int64_t t256 = (d.m % 256 == 0 && d.n % 256 == 0) ? (d.m / 256) * (d.n / 256) : 0;
  const int64_t t128 = (d.m % 128 == 0 && d.n % 128 == 0) ? (d.m / 128) * (d.n / 128) : 0;
  auto fill = [](int64_t t) { return (double)t / (double)(((t + 255) / 256) * 256); }; // CU occupancy over the rounds
  if (t256 >= t256_min && 1.4 * fill(t256) >= fill(t128)) return 2;
  if (t128 >= t128_min) return 1;
  return 0;
}

hipError_t launch_gemm_bf16_fast(int tile, const GemmArgs &a, hipStream_t s) {
  if (tile == 2) return launch_bf16_dma256(a, s);
  // two loader waves; none / four were measured equal (profiles/r02_bf16_dma128_ablation.txt: the loop is bound by the
  // fill path, not by the DMA issue rate of a wave). -DTPP_BF16_NLW=0|4 builds those variants for A/B runs.
#ifndef TPP_BF16_NLW
#define TPP_BF16_NLW 2
#endif
  if (tile == 1) return launch_bf16_dma128<TPP_BF16_NLW>(a, s);
  return launch_bf16<2, 2, 1, 1>(a, s);
}

} // namespace tpp

// brgemm_bf16_dma256.hip - the large-shape member of the bf16 VNNI-2 BRGEMM family for gfx950:
// 256 x 256 workgroup tiles, four waves of 128 x 128 (4 x 4 accumulator tiles of
// v_mfma_f32_32x32x16_bf16 each), operands streamed global -> LDS by buffer_load ... lds (no
// staging registers).
//
// Why this tile: a CU pulls at most 50-64 B/clk from the L2 into the LDS (TA busy ~650 cycles per 32 KiB),
// and the 128 x 128 kernel needs 32 KiB of fill per 512 MFMA cycles - the fill path alone bounds it at
// ~80 % of the matrix-core rate and in practice it runs at ~50 %. A 256 x 256 workgroup
// tile halves the fill per flop (32 B/clk), and its 128 x 128 wave tiles read half the fragment bytes per
// MFMA ((4+4) fragments per 16 MFMAs instead of (2+2) per 4). Used when the output has about one such
// tile per CU (pick_bf16_tile); smaller outputs keep the 128 x 128 tiles so that every CU has work.
//
// Pipeline (per workgroup, 256 threads, 1 wave per SIMD so each lane may use 512 registers: 256
// accumulators + 2 fragment sets of 32):
//   ring : 4 slots x 32 KiB; a slot = one K chunk of 32: A [256 rows][64 B] | B 16 VNNI pair-rows x
//          [256 columns] dwords. Chunk t+3 is fetched (8 DMA instructions per wave, one every second
//          MFMA) during the second K step of chunk t, into the slot chunk t-1 just left.
//   A    : LDS image [256 rows][4 x 16 B]; 16-byte piece index XOR ((row>>2)&3), applied to the SOURCE
//          address of the DMA and again by the fragment read: ds_read_b128 conflict-free.
//   B    : the VNNI-2 pair-rows as they are; a lane's fragment (8 consecutive k of one column) is 4
//          dwords one pair-row (1 KiB) apart: two ds_read2st64_b32.
//   sync : per chunk ONE s_waitcnt vmcnt (own DMA of chunk t+1 landed) + ONE raw s_barrier between its
//          two K steps; never __syncthreads inside the loop (it would drain the DMA queue).
//   out  : as the 128 x 128 kernel - operands swapped so a lane owns a row, bias / relu / one RNE
//          rounding (v_cvt_pk_bf16_f32), permlane32 swap, per-wave LDS tile, coalesced 16-byte stores.
#include "gemm_common.h"
#include "xsmm_desc.h"
#include <type_traits>

namespace tpp {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;

#ifndef TPP_ABLATE256
#define TPP_ABLATE256 0 // timing experiments (results are wrong): 1 no DMA in the loop, 2 no fragment reads, 4 no barrier
#endif

__global__ __launch_bounds__(256) void brgemm_bf16_dma256(GemmArgs p) {
  constexpr int BM = 256, BN = 256, BK2 = 32, NSLOT = 4, TM = 4, TN = 4;
  constexpr int A_SLOT = BM * BK2 * 2, B_SLOT = (BK2 / 2) * BN * 4, SLOT = A_SLOT + B_SLOT;
  constexpr int DMA_PER_CHUNK = 8; // per wave: 4 x 1 KiB of A + 4 x 1 KiB of B
  static_assert(SLOT == 32768, "ring slot");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_x[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int tm = (int)(blockIdx.x >> 1) * p.tiles_m + (int)blockIdx.z;
  const int tn = (int)(blockIdx.x & 1) * p.tiles_n + (int)blockIdx.y;
  const int m0 = tm * BM, n0 = tn * BN;
  const unsigned short *__restrict__ A = (const unsigned short *)p.A;
  const unsigned short *__restrict__ B = (const unsigned short *)p.B;
  unsigned short *__restrict__ C = (unsigned short *)p.C;
  const int kchunks = p.k / BK2;
  const int T = p.br * kchunks;

  // per-lane source offsets of this wave's 8 DMA instructions (constant for the kernel):
  // A instruction v fills rows 16v .. 16v+15 (lane -> row 16v + lane/4, LDS piece lane%4),
  // B instruction v fills pair-row v (lane -> columns 4*lane .. 4*lane+3)
  unsigned voffA[4], voffB[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int v = wave * 4 + u;
    const int row = 16 * v + (lane >> 2), pos = lane & 3;
    voffA[u] = (unsigned)(row * (int)p.lda * 2 + ((pos ^ ((row >> 2) & 3)) << 4));
    voffB[u] = (unsigned)(v * (int)p.ldb * 4 + (lane << 4));
  }
  const unsigned short *gA = A + (int64_t)m0 * p.lda, *gB = B + 2 * (int64_t)n0;
  int kc = 0;
  const int64_t dA_wrap = p.stride_a - (int64_t)(kchunks - 1) * BK2;
  const int64_t dB_in = (int64_t)(BK2 / 2) * 2 * p.ldb, dB_wrap = p.stride_b - (int64_t)(kchunks - 1) * dB_in;
  auto dma_piece = [&](int slot, int u) __attribute__((always_inline)) {
    unsigned char *base = smem_x + slot * SLOT + wave * 4096;
    if (u < 4) {
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)gA, 0, 0x7fffffff, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t *)(base + u * 1024), 16, voffA[u], 0, 0, 0);
    } else {
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)gB, 0, 0x7fffffff, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t *)(base + A_SLOT + (u - 4) * 1024), 16, voffB[u - 4], 0, 0, 0);
    }
  };
#define TPP_DMA256_ADVANCE()    \
  do {                          \
    if (++kc == kchunks) {      \
      kc = 0;                   \
      gA += dA_wrap;            \
      gB += dB_wrap;            \
    } else {                    \
      gA += BK2;                \
      gB += dB_in;              \
    }                           \
  } while (0)

  f32x16 acc[TM][TN];
  bf16x8_t af[2][TM];
  u32x4 bw[2][TN]; // B fragments as dwords: a fragment is filled by two 2-dword reads
  // lane bases of the fragment reads (byte offsets inside a slot). One opaque base per column tile so
  // that the four pair-row reads of a fragment pair up as ds_read2st64_b32 (rows r, r+1) straight into
  // consecutive registers instead of being paired across tiles.
  int a_lane[2], b_lane[TN];
  a_lane[0] = (wm * 128 + li) * 64 + ((lh ^ ((li >> 2) & 3)) << 4); // K step 0: pieces lh
  a_lane[1] = a_lane[0] ^ 32;                                       // K step 1: pieces 2 + lh
  asm volatile("" : "+v"(a_lane[0]), "+v"(a_lane[1]));
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    b_lane[j] = A_SLOT + ((4 * lh) * BN + wn * 128 + j * 32 + li) * 4;
    asm volatile("" : "+v"(b_lane[j]));
  }
  // One of the 12 LDS reads of a fragment set, in the order the MFMAs need them: A0, B0 (two halves),
  // B1, B2, B3, A1, A2, A3. They are issued ONE PER MFMA (a burst of 12 reads from four waves at once
  // fills the LDS queue and stalls the MFMA issue behind it: measured 217 cycles per K step).
  auto frag_piece = [&](int buf, int slot, int ks, int idx) __attribute__((always_inline)) {
    if (TPP_ABLATE256 & 2) return;
    const unsigned char *s = smem_x + slot * SLOT;
    if (idx == 0 || idx >= 9) {
      const int i = idx == 0 ? 0 : idx - 8;
      af[buf][i] = *(const bf16x8_t *)(s + a_lane[ks] + i * (32 * 64));
    } else {
      const int j = (idx - 1) >> 1, h = (idx - 1) & 1;
      const unsigned int *bp = (const unsigned int *)(s + b_lane[j] + (8 * ks) * BN * 4);
      bw[buf][j][2 * h] = bp[(2 * h) * BN];
      bw[buf][j][2 * h + 1] = bp[(2 * h + 1) * BN];
    }
  };
  auto frag_load = [&](int buf, int slot, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int idx = 0; idx < 12; ++idx) frag_piece(buf, slot, ks, idx);
  };
  // One chunk in ring slot `slot`. STEADY: chunks t+1..t+3 exist (everything unconditional, slot a
  // literal); otherwise one of the last <= 6 chunks: h1 / h2 / h3 say whether chunks t+1 / t+2 / t+3
  // exist. Two code paths only (the 4-chunk steady body, the run-time tail): all 256 accumulator
  // registers are live across every path, so there is no room for the copies that merging many
  // specialised paths would need.
  auto chunk = [&](auto steady_c, int slot, bool h1, bool h2, bool h3) __attribute__((always_inline)) {
    constexpr bool STEADY = decltype(steady_c)::value;
    // ---- K step 0 (fragments in set 0); set 1 <- K step 1 of this chunk, one read per MFMA
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bw[0][j]), af[0][i], acc[i][j], 0, 0, 0);
        if (i * TN + j < 12) frag_piece(1, slot, 1, i * TN + j);
        __builtin_amdgcn_sched_barrier(0);
      }
    const bool next = STEADY || h1;
    if (next) {
      // chunk t+1: this wave's DMA has landed (chunk t+2's may still fly), then everybody's
      if (STEADY || h2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(TPP_ABLATE256 & 4)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (!STEADY) { // the last chunks: plain bursts, a branch per MFMA would cost more
        frag_load(0, (slot + 1) & (NSLOT - 1), 0);
        if (h3) {
#pragma unroll
          for (int u = 0; u < 8; ++u) dma_piece((slot + 3) & (NSLOT - 1), u);
          TPP_DMA256_ADVANCE();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- K step 1 (set 1); set 0 <- K step 0 of chunk t+1; the DMA of chunk t+3 rides along, into
    // the slot chunk t-1 has left
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bw[1][j]), af[1][i], acc[i][j], 0, 0, 0);
        if (STEADY && i * TN + j < 12) frag_piece(0, (slot + 1) & (NSLOT - 1), 0, i * TN + j);
        if (STEADY && ((i * TN + j) & 1) && !(TPP_ABLATE256 & 1)) {
          dma_piece((slot + 3) & (NSLOT - 1), (i * TN + j) >> 1);
          if (i == TM - 1 && j == TN - 1) TPP_DMA256_ADVANCE();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;

  // prologue: chunks 0, 1, 2 in flight at once
#pragma unroll
  for (int c = 0; c < 3; ++c)
    if (T > c) {
#pragma unroll
      for (int u = 0; u < 8; ++u) dma_piece(c, u);
      TPP_DMA256_ADVANCE();
    }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  if (T > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_CHUNK) : "memory");
  else if (T > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_CHUNK) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (T > 0) frag_load(0, 0, 0);
  if (TPP_ABLATE256 & 2) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[q][i] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        bw[q][i] = u32x4{0, 0, 0, 0};
        asm volatile("" : "+v"(af[q][i]), "+v"(bw[q][i]));
      }
  }
  int t = 0;
  for (; t + 6 < T; t += 4) { // steady state: ring slots are compile-time constants (t % 4 == 0 here)
    chunk(yes{}, 0, true, true, true);
    chunk(yes{}, 1, true, true, true);
    chunk(yes{}, 2, true, true, true);
    chunk(yes{}, 3, true, true, true);
  }
  for (; t < T; ++t) chunk(no{}, t & (NSLOT - 1), t + 1 < T, t + 2 < T, t + 3 < T); // <= 6 chunks

  // ---- epilogue ------------------------------------------------------------------------
  // lane (li, lh) owns row 32*i + li of wave-tile row block i and, in registers 4g..4g+3 of
  // tile (i, j), columns 32*j + 8*g + 4*lh + (0..3)
  typedef unsigned int u32x2d __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t rsrcC = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(C + (int64_t)(m0 + wm * 128) * p.ldc + n0 + wn * 128), 0, 0x7fffffff, 0x00020000);
  const unsigned ldcb = (unsigned)((int)p.ldc * 2);
  if (!(p.ep & EP_BETA0)) { // beta = 1: add C before the single rounding
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
  __syncthreads(); // every wave is done with the ring (all DMA landed): reuse it as per-wave output tiles
  const bool relu = (p.ep & EP_RELU) != 0;
  constexpr int ES = 272; // bytes per staged row: 128 bf16 + 16 B pad (conflict-free 16-byte accesses)
  unsigned char *ot = smem_x + wave * (64 * ES);
#pragma unroll
  for (int ih = 0; ih < 2; ++ih) { // 64 output rows of the wave at a time
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float bias[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2d b2 = {0u, 0u};
        if (p.ep & EP_BIAS)
          b2 = *(const u32x2d *)((const unsigned short *)p.D + n0 + wn * 128 + 32 * j + 8 * g + 4 * lh);
        bias[g][0] = __uint_as_float(b2[0] << 16);
        bias[g][1] = __uint_as_float(b2[0] & 0xffff0000u);
        bias[g][2] = __uint_as_float(b2[1] << 16);
        bias[g][3] = __uint_as_float(b2[1] & 0xffff0000u);
      }
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * ih + ii;
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
            pk[g][h2] = __builtin_bit_cast(unsigned int, __builtin_convertvector(vv, bf16x2_t)); // v_cvt_pk_bf16_f32 (RNE)
          }
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          const auto s0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
          const u32x4 out = {s0[0], s1[0], s0[1], s1[1]};
          // lower half-wave: columns 32j + 8g .. +7 ; upper: 32j + 8(g+1) .. +7
          *(u32x4 *)(ot + (32 * ii + li) * ES + (32 * j + 8 * g + 8 * lh) * 2) = out;
        }
      }
    }
    // the same wave reads its 64 x 128 tile back row-contiguously: 16 lanes x 16 B = one 256-byte row
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = it * 4 + (lane >> 4), ch = lane & 15;
      const u32x4 v = *(const u32x4 *)(ot + row * ES + ch * 16);
      __builtin_amdgcn_raw_buffer_store_b128(v, rsrcC, (unsigned)(lane >> 4) * ldcb + (unsigned)(ch * 16),
                                             (unsigned)(ih * 64 + it * 4) * ldcb, C_STORE_AUX);
    }
  }
}

hipError_t launch_bf16_dma256(const GemmArgs &a, hipStream_t s) {
  constexpr size_t lds = 4 * 32768;
  static std::atomic<unsigned long long> lds_set{0};
  if (hipError_t e = ensure_dynamic_lds((const void *)brgemm_bf16_dma256, (int)lds, lds_set); e != hipSuccess) return e;
  GemmArgs args = a;
  const int tiles_m = a.m / 256, tiles_n = a.n / 256;
  dim3 grid;
  if ((tiles_m & 3) == 0 && (tiles_n & 1) == 0 && tiles_m / 4 <= 65535 && tiles_n / 2 <= 65535) {
    args.tiles_m = tiles_m / 4; // XCD-blocked, as the other fast kernels
    args.tiles_n = tiles_n / 2;
    grid = dim3(8, args.tiles_n, args.tiles_m);
  } else {
    args.tiles_m = args.tiles_n = 0;
    if (tiles_m > 65535 || tiles_n > 65535) return hipErrorInvalidValue;
    grid = dim3(1, tiles_n, tiles_m);
  }
  hipLaunchKernelGGL(brgemm_bf16_dma256, grid, dim3(256), lds, s, args);
  return hipGetLastError();
}

} // namespace tpp

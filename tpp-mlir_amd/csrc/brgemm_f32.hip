// brgemm_f32.hip - f32 batch-reduce GEMM for gfx950 on v_mfma_f32_32x32x2_f32.
//
// Semantics (reference: xsmm.brgemm / xsmm.fused_brgemm, XsmmOps.td:128-181,281-308;
// runtime/Xsmm/XsmmRunnerUtils.cpp:288-457):
//   C[m x n] = relu?( (beta0 ? 0 : C) + sum_{b<br} A_b[m x k] * B_b[k x n]  (+ bias[n]) )
// row-major, A_b = A + b*stride_a, B_b = B + b*stride_b (elements).
//
// Kernel structure (one workgroup per output tile; no split-K across workgroups,
// so results are deterministic and each element is an f32 fma chain):
//   * workgroup tile (32*WM) x (32*WN), WM*WN*WK waves; each wave owns one 32x32
//     accumulator tile (16 VGPRs) and 1/WK of every K chunk; WK>1 partial sums are
//     combined once through LDS at the end.
//   * the batch-reduce loop IS the K loop: chunk t = (batch t / (k/64), 64 columns of
//     k). A/B panels go HBM -> VGPR (global_load_dwordx4, issued 1.5 chunks ahead)
//     -> LDS (ds_write_b128) -> MFMA fragments, through a 3-slot LDS ring.
//   * ONE barrier per chunk, placed in the MIDDLE of the chunk's MFMA stream: chunk
//     t+1 is published while chunk t still has MFMAs to issue, so no wave ever
//     waits for data at a chunk boundary.
//   * A is stored in LDS as [row][64 k] with the 16-byte column index XOR (row&15):
//     the per-lane ds_read_b128 of 4 consecutive k is bank-conflict free. A lane's
//     4 values feed 4 successive MFMAs; lane-half h therefore covers k = 8q+4h+s at
//     step s (a fixed permutation of k inside each block of 8 - any order is a valid
//     summation order). B is stored [k][n] linear and read with ds_read_b32
//     (conflict free: 32 consecutive columns per lane group).
#include "gemm_common.h"
#include "xsmm_desc.h"
#include "chain_args.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace tpp {

// Timing-only ablation switches for kernel work (tools/ablate.sh builds side libraries
// with -DTPP_ABLATE=mask; the shipped library is always built with 0 = full kernel).
#ifndef TPP_ABLATE
#define TPP_ABLATE 0
#endif
#ifndef TPP_NACC
#define TPP_NACC 1 // accumulator chains per wave: 1 measured +0.9 % on C2 over 2 (and it is the oracle's summation order: one chain)
#endif
constexpr int ABL_NO_GLOAD = 1, ABL_NO_SWRITE = 2, ABL_NO_BARRIER = 4, ABL_NO_FRAG = 8;

constexpr int BK = 64;     // k columns per chunk
constexpr int NSTAGE = 3;  // LDS ring slots
constexpr int NSET = 3;    // staging register sets = chunks of global loads in flight per lane

typedef __attribute__((address_space(3))) void lds_void_f;

// One LDS-DMA instruction: 64 lanes x 16 bytes from (panel base, per-lane offset) to 1 KiB of the
// dynamic LDS at byte offset lds_off. A plain function on purpose: called from the kernel TEMPLATE with
// template-dependent operands, the builtin made hipcc drop the kernels' host stubs without a diagnostic.
static __device__ __forceinline__ void lds_dma_16B(const void *panel, unsigned lds_off, unsigned voff) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds_f32[];
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)panel, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_f *)(dyn_lds_f32 + lds_off), 16, voff, 0, 0, 0);
}

// DMA = true: panels go HBM -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction):
// no staging registers and no ds_write_b128 (13 LDS cycles each) competing with the fragment reads.
// Chunk t+2 is fetched during the second half of chunk t into the ring slot chunk t-1 has left; the
// A swizzle is applied to the source address. Everything else (fragments, barrier placement,
// epilogue) is shared with the register-staged path.
// items != nullptr: GROUPED mode (tile queue) - the grid is (items, tiles_n, tiles_m) and workgroup
// (x, y, z) computes tile (z, y) of queued invoke x, whose operand pointers and batch count come from
// items[x]; the descriptor fields (m, n, k, leading dimensions, strides, epilogue) are shared.
template <int WM, int WN, int WK, int NACC, bool DMA>
__global__ __launch_bounds__(64 * WM * WN * WK) void brgemm_f32_fast(GemmArgs p, const WorkItem *__restrict__ items) {
  if (items) { // wave-uniform: overwrite the per-invoke fields of the (by-value) argument block
    const WorkItem it = items[blockIdx.x];
    p.A = it.A; p.B = it.B; p.C = it.C; p.D = it.D; p.br = (int)it.br;
  }
  constexpr int BM = 32 * WM, BN = 32 * WN, NT = 64 * WM * WN * WK;
  constexpr int A_STAGE = BM * BK, B_STAGE = BK * BN; // floats
  constexpr int LA = (BM * BK / 4) / NT, LB = (BK * BN / 4) / NT;
  constexpr int KB_PER_WAVE = 8 / WK;                  // k-blocks (of 8) per wave per chunk
  constexpr int KB_HALF = KB_PER_WAVE / 2;
  constexpr int NP = LA + LB;                          // staging pieces per thread per chunk
  // MFMA slots that carry the LDS writes: they end one step before the barrier so the
  // barrier's lgkmcnt(0) finds them retired
  constexpr int WSLOTS = KB_HALF >= 2 ? 4 * (KB_HALF - 1) : 4 * KB_HALF;
  static_assert(LA >= 1 && LB >= 1 && KB_HALF >= 1, "tile too small for this thread count");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *As = smem;
  float *Bs = smem + NSTAGE * A_STAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn / WN, wn = wmn % WN;
  const int li = lane & 31, lh = lane >> 5;
  // Output tile from the 3-D grid, no divisions (launch_fast picks the shape): XCD-blocked
  // grids are (8, bn, bm): blockIdx.x is the XCD slot (workgroups go to XCDs round-robin in
  // linear-id order, x fastest), and each XCD owns a compact bm x bn block of tiles so the
  // A row-panels / B column-panels it streams are shared in its private L2. Plain grids are
  // (1, tiles_n, tiles_m) with the same formula (blockIdx.x == 0).
  const int tm = items ? (int)blockIdx.z : (int)(blockIdx.x >> 1) * p.tiles_m + (int)blockIdx.z;
  const int tn = items ? (int)blockIdx.y : (int)(blockIdx.x & 1) * p.tiles_n + (int)blockIdx.y;
  const int m0 = tm * BM, n0 = tn * BN;

  const float *__restrict__ A = (const float *)p.A;
  const float *__restrict__ B = (const float *)p.B;
  float *__restrict__ C = (float *)p.C;
  const int kchunks = p.k / BK;
  const int T = p.br * kchunks;

  f32x16 acc[NACC]; // initialised after the first loads are on their way

  // staging registers of the chunk in flight (HBM -> VGPR -> LDS). Pieces 0..LA-1 are
  // A, LA..LA+LB-1 are B (16 bytes per lane each). The LDS writes of a chunk are spread
  // over the MFMA slots of the FIRST half of the previous chunk's steps and its global
  // loads over the SECOND half, so neither the LDS nor the vector-memory path sees a
  // burst. Loads are buffer loads: wave-uniform 64-bit panel base in the descriptor
  // (advanced by scalar adds once per chunk), per-lane byte offset constant for the
  // whole kernel -> one instruction per 16 bytes, no per-load vector address math.
  f32x4 rs[DMA ? 1 : NSET][NP];
  unsigned voff[NP];
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    if (DMA) {
      // DMA instruction v of this panel fills 1 KiB of the LDS image linearly: lane -> (row, piece)
      // of the image; the source piece of A is XOR-ed with (row & 15) (the read applies it again)
      if (u < LA) {
        const int v = wave * LA + u, row = 4 * v + (lane >> 4), c = lane & 15;
        voff[u] = (unsigned)((row * (int)p.lda + 4 * (c ^ (row & 15))) * 4);
      } else {
        constexpr int RPI = 256 / BN; // B rows per instruction
        const int v = wave * LB + (u - LA), krow = RPI * v + lane / (BN / 4), c = lane % (BN / 4);
        voff[u] = (unsigned)((krow * (int)p.ldb + 4 * c) * 4);
      }
    } else if (u < LA) {
      const int q = tid + u * NT, row = q >> 4, c = q & 15;
      voff[u] = (unsigned)((row * (int)p.lda + 4 * c) * 4);
    } else {
      const int q = tid + (u - LA) * NT, krow = q / (BN / 4), c = q % (BN / 4);
      voff[u] = (unsigned)((krow * (int)p.ldb + 4 * c) * 4);
    }
  }
  // panel base of the chunk being loaded (wave-uniform) and its position in the batch
  const float *gA = A + (int64_t)m0 * p.lda, *gB = B + n0;
  int kc = 0; // chunk index inside the current batch element
  const int64_t dA_wrap = p.stride_a - (int64_t)(kchunks - 1) * BK;
  const int64_t dB_in = (int64_t)BK * p.ldb, dB_wrap = p.stride_b - (int64_t)(kchunks - 1) * BK * p.ldb;
  auto gadvance = [&]() __attribute__((always_inline)) { // next chunk: +64 k inside a batch element, else next batch element
    if (++kc == kchunks) {
      kc = 0;
      gA += dA_wrap;
      gB += dB_wrap;
    } else {
      gA += BK;
      gB += dB_in;
    }
  };
  auto gload_piece = [&](int set, int u) __attribute__((always_inline)) {
    // descriptor built from wave-uniform scalars right at the load (kept in SGPRs)
    const __amdgpu_buffer_rsrc_t r =
        __builtin_amdgcn_make_buffer_rsrc((void *)(u < LA ? gA : gB), 0, 0x7fffffff, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff[u], 0, 0);
    rs[set][u] = __builtin_bit_cast(f32x4, v);
  };
  auto dma_piece = [&](int stage, int u) __attribute__((always_inline)) {
    if (u < LA) lds_dma_16B(gA, (unsigned)((stage * A_STAGE + (wave * LA + u) * 256) * 4), voff[u]);
    else lds_dma_16B(gB, (unsigned)((NSTAGE * A_STAGE + stage * B_STAGE + (wave * LB + (u - LA)) * 256) * 4), voff[u]);
  };
  auto swrite_piece = [&](int stage, int u) __attribute__((always_inline)) {
    float *as = As + stage * A_STAGE, *bs = Bs + stage * B_STAGE;
    if (u < LA) {
      const int q = tid + u * NT, row = q >> 4, c = q & 15;
      *(f32x4 *)(as + row * BK + ((c ^ (row & 15)) << 2)) = rs[stage][u];
    } else {
      const int q = tid + (u - LA) * NT, krow = q / (BN / 4), c = q % (BN / 4);
      *(f32x4 *)(bs + krow * BN + 4 * c) = rs[stage][u];
    }
  };
  // MFMA fragments of one k-block (8 k): 4 A values (one ds_read_b128) and 4 B values
  // per lane; double-buffered so block q+1 is read while block q multiplies.
  f32x4 fa[2];
  float fb[2][4];
  const int a_off = (wm * 32 + li) * BK, b_off = wn * 32 + li;
  auto frag_load = [&](int buf, int stage, int kb) __attribute__((always_inline)) {
    const float *as = As + stage * A_STAGE + a_off;
    const float *bs = Bs + stage * B_STAGE + b_off;
    fa[buf] = *(const f32x4 *)(as + (((2 * kb + lh) ^ (li & 15)) << 2));
#pragma unroll
    for (int s = 0; s < 4; ++s) fb[buf][s] = bs[(8 * kb + 4 * lh + s) * BN];
  };

  const int kbw = wk * KB_PER_WAVE;
  // One chunk = KB_PER_WAVE k-block steps of 4 MFMAs, ring slot STAGE known at compile
  // time (every LDS address is base VGPR + immediate). Register set s holds the chunk
  // that goes to ring slot s (NSET == NSTAGE). HAS_NEXT: chunk t+1 exists: set STAGE+1
  // is written to slot STAGE+1 during the first half of the steps, then ONE barrier
  // publishes it. HAS_LOAD: chunk t+1+NSET exists: its global loads refill the set just
  // written, during the second half - they have NSET-0.5 chunks of MFMAs to land. The
  // last step prefetches the first fragments of chunk t+1.
  auto chunk = [&](auto stage_c, auto has_next, auto has_load) __attribute__((always_inline)) {
    constexpr int STAGE = decltype(stage_c)::value, NSTG = (STAGE + 1) % NSTAGE;
    constexpr bool HAS_NEXT = decltype(has_next)::value, HAS_LOAD = decltype(has_load)::value;
#pragma unroll
    for (int q = 0; q < KB_PER_WAVE; ++q) {
      const int cur = q & 1, nxt = cur ^ 1;
      if (!(TPP_ABLATE & ABL_NO_FRAG)) {
        if (q + 1 < KB_PER_WAVE) frag_load(nxt, STAGE, kbw + q + 1);
        else if (HAS_NEXT) frag_load(nxt, NSTG, kbw);
      }
      // pin the issue order: the fragment reads of step q+1 stay ABOVE the MFMAs of
      // step q, and each piece of staging work sits in the shadow of one MFMA (the
      // wave is in-order: it idles at the next MFMA until the matrix pipe frees up).
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], fb[cur][s], acc[s % NACC], 0, 0, 0);
        if (!DMA && HAS_LOAD && !(TPP_ABLATE & ABL_NO_GLOAD) && q == 0 && s == 0) {
          // panel base of the next chunk to load: scalar work in the shadow of the MFMA above
          if (++kc == kchunks) {
            kc = 0;
            gA += dA_wrap;
            gB += dB_wrap;
          } else {
            gA += BK;
            gB += dB_in;
          }
        }
#pragma unroll
        for (int u = 0; u < NP; ++u) {
          if (!DMA && HAS_NEXT && !(TPP_ABLATE & ABL_NO_SWRITE) && (u * WSLOTS) / NP == q * 4 + s) swrite_piece(NSTG, u);
          if (HAS_LOAD && !(TPP_ABLATE & ABL_NO_GLOAD) && (u * 4 * (KB_PER_WAVE - KB_HALF)) / NP == (q - KB_HALF) * 4 + s) {
            if (DMA) dma_piece((STAGE + 2) % NSTAGE, u); // chunk t+2 into the slot chunk t-1 has left (after the barrier)
            else gload_piece(NSTG, u);
          }
        }
        if (DMA && HAS_LOAD && !(TPP_ABLATE & ABL_NO_GLOAD) && q == KB_PER_WAVE - 1 && s == 3) {
          if (++kc == kchunks) { // panel base of the chunk after the one just requested
            kc = 0;
            gA += dA_wrap;
            gB += dB_wrap;
          } else {
            gA += BK;
            gB += dB_in;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (q == KB_HALF - 1 && !(TPP_ABLATE & ABL_NO_BARRIER)) {
        // DMA: this wave's pieces of chunk t+1 have landed in the LDS (the fence of __syncthreads
        // does not wait for LDS-DMA), then everybody's
        if (DMA && HAS_NEXT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  static_assert(NSET == NSTAGE && NSTAGE == 3, "the chunk schedule below is written for 3 slots / 3 sets");

  // prologue: chunk 0 -> set 0 -> slot 0; chunks 1, 2, 3 -> sets 1, 2, 0 (in flight)
  if (T > 0) { // first thing the kernel does: get chunk 0 moving
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      if (DMA) dma_piece(0, u);
      else gload_piece(0, u);
    }
    if (DMA && T > 1) { // chunk 1 right behind it; afterwards the panel base points at chunk 2
      gadvance();
#pragma unroll
      for (int u = 0; u < NP; ++u) dma_piece(1, u);
      gadvance();
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // accumulators: wk == 0 starts from C (beta = 1) so the chain is C + sum, as in
  // the reference; other K groups start from zero.
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
  // C tile through buffer ops: wave-uniform tile base in the descriptor, per-lane offset
  // constant, the row of accumulator register r as a scalar offset -> one instruction per
  // register, no 64-bit vector address math in the epilogue.
  const int ccol = n0 + wn * 32 + li;
  const __amdgpu_buffer_rsrc_t rsrcC =
      __builtin_amdgcn_make_buffer_rsrc((void *)(C + (int64_t)m0 * p.ldc + n0), 0, 0x7fffffff, 0x00020000);
  const unsigned voffC = (unsigned)(((wm * 32 + 4 * lh) * (int)p.ldc + wn * 32 + li) * 4);
  const unsigned ldcb = (unsigned)((int)p.ldc * 4);
  if (!(p.ep & EP_BETA0) && wk == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      acc[0][r] = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(rsrcC, voffC, (unsigned)((r & 3) + 8 * (r >> 2)) * ldcb, 0));
  }

  if (!DMA && T > 0) {
#pragma unroll
    for (int c = 1; c <= NSET; ++c) {
      if (c == NSET) { // set 0 is reused for chunk NSET: chunk 0 must be in LDS first
#pragma unroll
        for (int u = 0; u < NP; ++u) swrite_piece(0, u);
      }
      if (c < T) {
        gadvance();
#pragma unroll
        for (int u = 0; u < NP; ++u) gload_piece(c % NSET, u);
      }
    }
  }
  if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (T > 0) frag_load(0, 0, kbw);
  int t = 0;
  constexpr int AHEAD = DMA ? 2 : NSET + 1; // chunk t + AHEAD is the one fetched during chunk t
  for (; t + 2 + AHEAD < T; t += 3) { // steady state: three chunks per trip, ring slots 0, 1, 2
    chunk(S0{}, yes{}, yes{});
    chunk(S1{}, yes{}, yes{});
    chunk(S2{}, yes{}, yes{});
  }
  auto tail = [&](auto stage_c) __attribute__((always_inline)) { // last chunks: same bodies minus what no longer exists
    const int left = T - t;
    if (left > AHEAD) chunk(stage_c, yes{}, yes{});
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

  if (TPP_ABLATE & (ABL_NO_SWRITE | ABL_NO_FRAG)) { // keep ablated producers alive
#pragma unroll
    for (int u = 0; u < LA + LB; ++u) asm volatile("" ::"v"(rs[0][u]), "v"(rs[DMA ? 0 : 1][u]), "v"(rs[DMA ? 0 : 2][u]));
    asm volatile("" ::"v"(fa[0]), "v"(fa[1]));
  }
#pragma unroll
  for (int a = 1; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] += acc[a][r];

  if constexpr (WK > 1) {
    // combine the K groups through LDS: group g>0 parks its 32x32 partial, group 0 adds
    __syncthreads();
    float *red = smem; // (WK-1) * WM*WN * 16 * 64 floats, fits in the ring
    if (wk > 0) {
      float *dst = red + ((wk - 1) * (WM * WN) + wmn) * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64] = acc[0][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int g = 1; g < WK; ++g) {
      const float *src = red + ((g - 1) * (WM * WN) + wmn) * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] += src[r * 64];
    }
  }

  const float bias = (p.ep & EP_BIAS) ? ((const float *)p.D)[ccol] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = acc[0][r] + bias;
    if (p.ep & EP_RELU) v = v > 0.0f ? v : 0.0f;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrcC, voffC,
                                          (unsigned)((r & 3) + 8 * (r >> 2)) * ldcb, C_STORE_AUX);
  }
}

// ---- grouped kernel: many invokes of one small-tile descriptor in ONE launch ----------
// The compiler's native call pattern is hundreds of invokes per layer on 32x32x32 tiles with
// batch 32 from OpenMP workers (test/Passes/pass-convert-mlp-to-parallel-tile.mlir:80-88); one
// GPU launch per invoke would be pure launch latency. The runtime's tile queue (runtime.cpp)
// collects such invokes and runs them here: grid = (32x32 tiles per item, items), one
// workgroup of 4 waves per tile; wave w takes the K chunks (32 k of one batch element)
// c = w, w+4, ... so all four SIMDs of a CU work on the tile, each with a private
// double-buffered LDS panel pair (no workgroup barrier in the K loop); the four partial
// accumulators are combined through LDS once. It is also the GENERIC kernel of the runtime
// (items == nullptr: one invoke described by the arguments themselves): any m/n/k/ld and
// alignment, f32 or bf16 storage (elements are widened to f32 on the way into LDS, bf16 x
// bf16 products are exact in f32, accumulation is f32 - the reference's "f32 compute for
// bf16" rule, XsmmRunnerUtils.cpp:127-129 - one rounding at the store), flat or VNNI-2 B.
// VEC selects 16-byte loads when shape, strides and pointers allow.
constexpr int GK = 32; // k per chunk of the grouped kernel

template <typename T, bool VNNI, bool VEC, int VF = 2>
__global__ __launch_bounds__(256) void brgemm_grouped(GemmArgs p, const WorkItem *__restrict__ items) {
  extern __shared__ __attribute__((aligned(16))) float smem_g[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const WorkItem it = items ? items[blockIdx.y] : WorkItem{p.A, p.B, p.C, p.D, (int64_t)p.br};
  g_cvoid *gA = (g_cvoid *)it.A, *gB = (g_cvoid *)it.B, *gD = (g_cvoid *)it.D; // global, not flat, accesses
  g_void *gC = (g_void *)it.C;


  const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x % p.tiles_n;
  const int m0 = tm * 32, n0 = tn * 32;
  const int kchunks = (p.k + GK - 1) / GK;
  const int nchunk = (int)it.br * kchunks;
  // per-wave LDS: 2 buffers x (A 32x32 + B 32x32) floats
  float *wl = smem_g + wave * (2 * 2 * 32 * GK);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

  // staging registers: A tile rows [m0, m0+32) x k [kk0, kk0+32); B tile k x n
  // two staging register sets: 32 x 32 floats = 256 pieces of 16 B per panel, 4 per lane and panel
  f32x4 rs[2][2][4];
  // f32 with 16-byte loads (the compiler-native packed 32x32x32 tiles): per-lane byte offsets inside a chunk, fixed for the kernel.
  // Rows / 4-column pieces beyond a ragged edge (m not a multiple of 32, n only of 4, e.g. --tiles=64,48,64) are loaded from
  // a CLAMPED address and left as they are: an output element depends on its own A row and B column only, and the
  // epilogue stores nothing beyond the edge.
  unsigned voffA[4] = {0, 0, 0, 0}, voffB[4] = {0, 0, 0, 0};
  if constexpr (VEC && sizeof(T) == 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = lane + 64 * u, row = q >> 3, c4 = q & 7; // 32 rows x 8 pieces of 4 floats
      // offsets are relative to the TILE (the 64-bit descriptor base carries m0 * lda and n0): 32 rows x ld x 4 B stays below
      // 2^31 for every ld the launcher admits (< 2^24), whatever m is - an absolute row here would wrap past 2 GiB of A
      const int a_row = m0 + row < p.m ? row : p.m - 1 - m0, b_col = n0 + 4 * c4 < p.n ? 4 * c4 : 0;
      voffA[u] = (unsigned)((a_row * (int)p.lda + 4 * c4) * 4);
      voffB[u] = (unsigned)((row * (int)p.ldb + b_col) * 4);
    }
  }
  // live = false: the chunk does not exist. The f32 16-byte path then still ISSUES its loads, switched off through the buffer
  // descriptor (num_records = 0, no memory traffic): a branch around a load makes hipcc wait with vmcnt(0) at the next use of
  // ANY staged register, which would serialise the two chunks kept in flight.
  auto gload = [&](int c, int set, bool live) __attribute__((always_inline)) {
    f32x4(&ra)[4] = rs[set][0];
    f32x4(&rb)[4] = rs[set][1];
    const int b = c / kchunks, kk0 = (c - b * kchunks) * GK;
    const int64_t abase = (int64_t)b * p.stride_a, bbase = (int64_t)b * p.stride_b;
    if constexpr (VEC && sizeof(T) == 4) {
      const int nrec = __builtin_amdgcn_readfirstlane(live ? 0x7fffffff : 0);
      const __amdgpu_buffer_rsrc_t rA =
          __builtin_amdgcn_make_buffer_rsrc((void *)((const float *)it.A + abase + (int64_t)m0 * p.lda + kk0), 0, nrec, 0x00020000);
      const __amdgpu_buffer_rsrc_t rB =
          __builtin_amdgcn_make_buffer_rsrc((void *)((const float *)it.B + bbase + (int64_t)kk0 * p.ldb + n0), 0, nrec, 0x00020000);
      // k need only be a multiple of 4: in the last chunk of a batch element the 16-byte pieces at or beyond k (A: k piece c4, B: k
      // row) are requested at an offset past the descriptor's end and come back as zeros - no branch, no select on loaded data
      const int klim = p.k - kk0; // >= 32 in every chunk but a ragged last one
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = lane + 64 * u;
        const unsigned oa = 4 * (q & 7) < klim ? voffA[u] : 0x80000000u, ob = (q >> 3) < klim ? voffB[u] : 0x80000000u;
        ra[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, oa, 0, 0));
        rb[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, ob, 0, 0));
      }
      return;
    }
    if (!live) return;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = lane + 64 * u, row = q >> 3, c4 = q & 7; // 32 rows x 8 pieces of 4 floats
      { // element-wise loads: any shape, stride and alignment
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int gr = m0 + row, gk = kk0 + 4 * c4 + e;
          ra[u][e] = (gr < p.m && gk < p.k) ? Elem<T>::load(gA, abase + (int64_t)gr * p.lda + gk) : 0.0f;
          const int bk = kk0 + row, bj = n0 + 4 * c4 + e;
          float v = 0.0f;
          if (bk < p.k && bj < p.n) {
            // VNNI-v B [k/v][ldb][v] (v = p.vf: 2 or 4; the oracle's b_index)
            // (b_trans: the operand is the SOURCE of a transpose the runtime folded into this gemm - element [bk][bj] of B is
            // element [bj][bk] of the source)
            const int64_t idx = VNNI ? (int64_t)(bk / p.vf) * (p.vf * p.ldb) + p.vf * (int64_t)bj + (bk % p.vf)
                                     : p.b_trans ? (int64_t)bj * p.ldb + bk : (int64_t)bk * p.ldb + bj;
            v = Elem<T>::load(gB, bbase + idx);
          }
          rb[u][e] = v;
        }
      }
    }
  };
  // piece u (one A quad + one B quad per lane) of staging set `set` into LDS buffer `buf`
  auto swrite_piece = [&](int buf, int set, int u) __attribute__((always_inline)) {
    f32x4(&ra)[4] = rs[set][0];
    f32x4(&rb)[4] = rs[set][1];
    float *as = wl + buf * (2 * 32 * GK), *bs = as + 32 * GK;
    const int q = lane + 64 * u;
    const int row = q >> 3, krow = row, c4 = q & 7, c4b = c4;
    *(f32x4 *)(as + row * GK + ((c4 ^ ((row >> 1) & 7)) << 2)) = ra[u]; // 128-byte rows: XOR on (row>>1)
    *(f32x4 *)(bs + krow * 32 + 4 * c4b) = rb[u];
  };
  auto swrite = [&](int buf, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) swrite_piece(buf, set, u);
  };
  // The 16 MFMAs of the chunk in LDS buffer `buf`, software-pipelined: the fragments of k-block kb + 1 are read while the four
  // MFMAs of k-block kb run, and (stage) the staging set `set` = the NEXT chunk of this wave goes into the other LDS buffer one
  // piece per k-block - a wave is alone on its SIMD here, so whatever is not overlapped inside the wave is idle matrix-core time
  // (before: 2300 cycles per chunk for 1024 cycles of MFMA).
  auto compute = [&](int buf, bool stage, int set) __attribute__((always_inline)) {
    const float *as = wl + buf * (2 * 32 * GK) + li * GK, *bs = wl + buf * (2 * 32 * GK) + 32 * GK + li;
    f32x4 a4[2];
    float b4[2][4];
    auto frag = [&](int kb, int to) __attribute__((always_inline)) {
      a4[to] = *(const f32x4 *)(as + (((2 * kb + lh) ^ ((li >> 1) & 7)) << 2));
#pragma unroll
      for (int s = 0; s < 4; ++s) b4[to][s] = bs[(8 * kb + 4 * lh + s) * 32];
    };
    frag(0, 0);
#pragma unroll
    for (int kb = 0; kb < GK / 8; ++kb) {
      if (kb + 1 < GK / 8) frag(kb + 1, (kb + 1) & 1);
      if (stage) swrite_piece(buf ^ 1, set, kb);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kb & 1][s], b4[kb & 1][s], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // wave-private pipeline, TWO chunks of loads in flight: while chunk c multiplies out of LDS buffer i & 1, chunk c + 4 sits in
  // (or is on its way to) register set (i + 1) & 1 and the loads of chunk c + 8 are issued into set i & 1. With one chunk in
  // flight a wave waited out most of an L2 round trip per chunk (0.5 us of MFMA against 0.6-0.8 us of latency): the tile-queue
  // launches of the reference's 32x32x32 pattern took 12.7 us per 256 tiles against 7.4 us for the whole-layer kernel.
  // The loop runs whole PAIRS of chunks (one per staging set / LDS buffer) with ONE exit and no branch inside: loads of chunks
  // that do not exist are switched off (f32 16-byte path) and staging them moves zeros / stale registers into an LDS
  // buffer nobody computes from. (With an exit per chunk hipcc keeps the accumulator in two register sets and moves it -
  // 16 v_accvgpr_read + 16 v_accvgpr_write behind every chunk's last MFMA; a branch around loads defeats the counted waits.)
  const int mine = nchunk > wave ? (nchunk - wave + 3) >> 2 : 0; // chunks of this wave: wave, wave + 4, ...
  int c = wave;
  if constexpr (sizeof(T) == 2 && VNNI && VEC) {
    // bf16 + VNNI-2 B with 16-byte loads (ragged bf16 tiles, e.g. mlir-gen --tiles=64,48,64 --vnni=2): the panels stay bf16 in
    // LDS and the chunk is TWO v_mfma_f32_32x32x16_bf16 (f32 accumulate, the arithmetic of the other bf16 kernels) instead of
    // sixteen f32 MFMAs on widened operands. Same pipeline: two chunks of loads in flight, switched off past the stream's end.
    //   A image [32 rows][32 k] bf16, row pitch 80 B (16-byte fragment reads of 16 consecutive rows hit 16 different bank quads);
    //   B image = the 16 VNNI pair-rows x 32 columns as they are (a lane's fragment: 4 dwords one pair-row apart).
    typedef __bf16 bf16x8_g __attribute__((ext_vector_type(8)));
    constexpr int APITCH = 80, ABYTES = 32 * APITCH, BUFB = ABYTES + 16 * 128;
    unsigned char *wlb = (unsigned char *)smem_g + wave * 16384;
    unsigned vA[2], vB[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int q = lane + 64 * v;
      // tile-relative (see the f32 path): the descriptor base carries m0 * lda and 2 * n0
      // B: a 16-byte piece = 4 columns x 2 k (VNNI-2: 16 pair-rows of 128 B) or 2 columns x 4 k (VNNI-4: 8 k-group rows of 256 B)
      constexpr int CPP = 8 / VF, PPR = 32 / CPP; // columns per piece, pieces per row of the 32-column panel
      const int a_row = m0 + (q >> 2) < p.m ? (q >> 2) : p.m - 1 - m0, b_col = n0 + CPP * (q % PPR) < p.n ? CPP * (q % PPR) : 0;
      vA[v] = (unsigned)((a_row * (int)p.lda + 8 * (q & 3)) * 2);
      vB[v] = (unsigned)(((q / PPR) * VF * (int)p.ldb + VF * b_col) * 2);
    }
    u32x4 sa[2][2], sb[2][2];
    auto gload_bf = [&](int cc, int set, bool live) __attribute__((always_inline)) {
      const int b = cc / kchunks, kk0 = (cc - b * kchunks) * GK;
      const int nrec = __builtin_amdgcn_readfirstlane(live ? 0x7fffffff : 0);
      const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
          (void *)((const unsigned short *)it.A + (int64_t)b * p.stride_a + (int64_t)m0 * p.lda + kk0), 0, nrec, 0x00020000);
      const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
          (void *)((const unsigned short *)it.B + (int64_t)b * p.stride_b + (int64_t)(kk0 / VF) * (VF * p.ldb) + VF * (int64_t)n0), 0, nrec, 0x00020000);
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        sa[set][v] = __builtin_amdgcn_raw_buffer_load_b128(rA, vA[v], 0, 0);
        sb[set][v] = __builtin_amdgcn_raw_buffer_load_b128(rB, vB[v], 0, 0);
      }
    };
    auto swrite_bf = [&](int buf, int set, int v) __attribute__((always_inline)) {
      const int q = lane + 64 * v;
      unsigned char *ab = wlb + buf * BUFB;
      *(u32x4 *)(ab + (q >> 2) * APITCH + (q & 3) * 16) = sa[set][v];
      *(u32x4 *)(ab + ABYTES + q * 16) = sb[set][v]; // (the rows follow each other: [16][128 B] or [8][256 B], both = piece q at 16 q)
    };
    auto compute_bf = [&](int buf, bool stage, int set) __attribute__((always_inline)) {
      const unsigned char *ab = wlb + buf * BUFB;
      bf16x8_g af[2];
      u32x4 bfr[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        af[ks] = *(const bf16x8_g *)(ab + li * APITCH + (2 * ks + lh) * 16);
        if constexpr (VF == 4) { // k-groups 4 ks + 2 lh and + 1 of this lane's column: two 8-byte pieces
          typedef unsigned int u32x2_g __attribute__((ext_vector_type(2)));
          const u32x2_g q0 = *(const u32x2_g *)(ab + ABYTES + (4 * ks + 2 * lh) * 256 + li * 8);
          const u32x2_g q1 = *(const u32x2_g *)(ab + ABYTES + (4 * ks + 2 * lh + 1) * 256 + li * 8);
          bfr[ks] = u32x4{q0[0], q0[1], q1[0], q1[1]};
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) bfr[ks][t] = *(const unsigned int *)(ab + ABYTES + (8 * ks + 4 * lh + t) * 128 + li * 4);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (stage) swrite_bf(buf ^ 1, set, ks);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], __builtin_bit_cast(bf16x8_g, bfr[ks]), acc, 0, 0, 0);
      }
    };
    gload_bf(c, 0, mine > 0);
    gload_bf(c + 4, 1, mine > 1);
    if (mine > 0) {
      swrite_bf(0, 0, 0);
      swrite_bf(0, 0, 1);
    }
    for (int i = 0; i + 1 < mine; i += 2) {
      gload_bf(c + 8, 0, i + 2 < mine);
      compute_bf(0, true, 1);
      gload_bf(c + 12, 1, i + 3 < mine);
      compute_bf(1, true, 0);
      c += 8;
    }
    if (mine & 1) compute_bf(0, false, 0);
  } else {
    gload(c, 0, mine > 0);
    gload(c + 4, 1, mine > 1);
    if (mine > 0) swrite(0, 0);
    for (int i = 0; i + 1 < mine; i += 2) {
      gload(c + 8, 0, i + 2 < mine);
      compute(0, true, 1);
      gload(c + 12, 1, i + 3 < mine);
      compute(1, true, 0);
      c += 8;
    }
    if (mine & 1) compute(0, false, 0);
  }
  // combine the four waves' partial sums: every wave parks all of its 16 accumulator registers, then wave w finishes the four
  // registers 4w .. 4w+3 (rows 8w + 4 lh + 0..3 of the tile): a quarter of the reads and stores per wave (one wave doing all of
  // it kept the other three idle for the whole tail). Summation order as before: wave 0 + wave 1 + wave 2 + wave 3, (+ C), + bias.
  __syncthreads();
  float *red = smem_g;
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[r];
  __syncthreads();
  const int col = n0 + li;
  const float bias = ((p.ep & EP_BIAS) && col < p.n) ? Elem<T>::load(gD, col) : 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 4 * wave + j;
    const int row = m0 + 4 * lh + j + 8 * wave;
    float v = red[r * 64 + lane];
#pragma unroll
    for (int g = 1; g < 4; ++g) v += red[g * 1024 + r * 64 + lane];
    if (row < p.m && col < p.n) {
      // C element (row, col): row-major, or VNNI-2 [m/2][n][2] (wire flag 8192, see the oracle's c_index)
      const int64_t ci = (p.ep & EP_VNNI_C) ? (int64_t)(row >> 1) * (2 * p.ldc) + 2 * (int64_t)col + (row & 1)
                                            : (int64_t)row * p.ldc + col;
      if (!(p.ep & EP_BETA0)) v += Elem<T>::load((g_cvoid *)gC, ci);
      v += bias;
      if (p.ep & EP_RELU) v = v > 0.0f ? v : 0.0f;
      Elem<T>::store(gC, ci, v);
    }
  }
}

// ---- host side ---------------------------------------------------------------
enum GemmVariant : int {
  V_F32_64x64 = 0,   // 4 waves 2x2x1
  V_F32_64x32K2 = 1, // 4 waves 2x1x2
  V_F32_32x32K4 = 2, // 4 waves 1x1x4
  V_F32_128x64 = 3,  // 8 waves 4x2x1
  V_F32_64x64K2 = 4, // 8 waves 2x2x2 (two waves per SIMD share every K chunk)
  V_F32_LW_64x64 = 5,     // brgemm_f32_lw.hip: 4 MFMA waves 2x2x1 + 2 loader waves
  V_F32_LW_64x64K2 = 6,   // 8 MFMA waves 2x2x2 + 2 loader waves
  V_F32_LW_64x32K2 = 7,   // 8 MFMA waves 2x1x4 + 2 loader waves (K split over four groups since round 3; the name of the constant stayed)
  V_GENERIC = 8,     // chosen per invoke when the fast preconditions fail
  V_F32_LW_32x32K4 = 9,   // 4 MFMA waves 1x1x4 + 2 loader waves
  V_F32_LW_128x64 = 10,   // 8 MFMA waves 4x2x1 + 2 x 2 loader waves, 3-slot ring (large outputs)
  V_F32_LW16_32x16 = 11,  // brgemm_f32_lw16.hip: 32x16 tiles on v_mfma_f32_16x16x4_f32, 4 MFMA waves (K split) + 3 loader waves: outputs of at most one 32x16 tile per CU
  V_BF16_FAST = 16,  // brgemm_bf16.hip: 64x64 register-staged
  V_BF16_DMA128 = 17, // brgemm_bf16.hip: 128x128, LDS-DMA + loader waves
  V_BF16_DMA256 = 18, // brgemm_bf16_dma256.hip: 256x256, LDS-DMA
  V_BF16_SMALL32 = 19, // brgemm_bf16_small.hip: 32x32 tiles, 4 waves split K, fragments straight from global memory
  V_BF16_LW_32x64 = 20,   // brgemm_bf16_lw.hip: loader-wave tiles for mid-size outputs (one workgroup per CU), 32x64 + K2
  V_BF16_LW_64x64 = 21,
  V_BF16_LW_64x128 = 22,
  V_BF16_LW_128x128 = 23,
  V_BF16_LWF_32x64 = 24,  // the same tiles for a FLAT bf16 B operand (no VNNI flag): the pair-row interleave happens in the B loader
  V_BF16_LWF_64x64 = 25,
  V_BF16_LWF_64x128 = 26,
  V_BF16_LWF_128x128 = 27,
  V_BF16_LW4_32x64 = 28,  // the same tiles for a VNNI-4 B operand [k/4][n][4] (xsmm_hip_set_vnni_factor(4)): a fragment is two 8-byte reads
  V_BF16_LW4_64x64 = 29,
  V_BF16_LW4_64x128 = 30,
  V_BF16_LW4_128x128 = 31,
};

template <int WM, int WN, int WK, int NACC, bool DMA>
static hipError_t launch_fast_t(const GemmArgs &a, hipStream_t s) {
  constexpr int BM = 32 * WM, BN = 32 * WN, NT = 64 * WM * WN * WK;
  constexpr size_t lds = (size_t)NSTAGE * (BM * BK + BK * BN) * sizeof(float);
  static std::atomic<unsigned long long> lds_set{0};
  if (hipError_t e = ensure_dynamic_lds((const void *)brgemm_f32_fast<WM, WN, WK, NACC, DMA>, (int)lds, lds_set); e != hipSuccess) return e;
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
  hipLaunchKernelGGL((brgemm_f32_fast<WM, WN, WK, NACC, DMA>), grid, dim3(NT), lds, s, args, (const WorkItem *)nullptr);
  return hipGetLastError();
}

// grouped launch of a fast tile family: one workgroup per (item, tile of the item)
template <int WM, int WN, int WK, int NACC, bool DMA>
static hipError_t launch_fast_grouped_t(const GemmArgs &a, const WorkItem *items, int n_items, hipStream_t s) {
  constexpr int BM = 32 * WM, BN = 32 * WN, NT = 64 * WM * WN * WK;
  constexpr size_t lds = (size_t)NSTAGE * (BM * BK + BK * BN) * sizeof(float);
  static std::atomic<unsigned long long> lds_set{0};
  if (hipError_t e = ensure_dynamic_lds((const void *)brgemm_f32_fast<WM, WN, WK, NACC, DMA>, (int)lds, lds_set); e != hipSuccess) return e;
  GemmArgs args = a;
  args.tiles_m = args.tiles_n = 0;
  hipLaunchKernelGGL((brgemm_f32_fast<WM, WN, WK, NACC, DMA>), dim3((unsigned)n_items, a.n / BN, a.m / BM), dim3(NT), lds, s, args,
                     items);
  return hipGetLastError();
}

template <int WM, int WN, int WK, int NACC, bool DMA>
static hipError_t launch_fast(const GemmArgs &a, hipStream_t s) {
  return launch_fast_t<WM, WN, WK, NACC, DMA>(a, s);
}

hipError_t launch_gemm_bf16_fast(int tile, const GemmArgs &a, hipStream_t s); // brgemm_bf16.hip
hipError_t launch_f32_lw(int tile, const GemmArgs &a, hipStream_t s);         // brgemm_f32_lw.hip (tile 4: 128x64, forced variant 10 only - it measures within 2 % of brgemm_f32_fast<128x64>)
hipError_t launch_f32_lw_grouped(int tile, const GemmArgs &a, const WorkItem *items, int n_items, int split, hipStream_t s);
hipError_t launch_f32_lw_split(int tile, const GemmArgs &a, int split, hipStream_t s); // hipErrorOutOfMemory / InvalidValue: not launched
hipError_t launch_f32_lw16(int tile, const GemmArgs &a, const WorkItem *items, int n_items, bool grouped, hipStream_t s); // brgemm_f32_lw16.hip: tile 0 = 32x16
int pick_bf16_tile(const GemmDesc &d);
bool bf16_fast_eligible(const GemmDesc &d);

template <typename T, bool VNNI, bool VEC, int VF = 2>
static hipError_t launch_grouped_t(const GemmArgs &a, const WorkItem *items, int n_items, hipStream_t s) {
  constexpr size_t lds = 4 * 2 * 2 * 32 * GK * sizeof(float); // 64 KiB: 4 waves x 2 buffers x (A + B)
  auto kern = brgemm_grouped<T, VNNI, VEC, VF>;
  static std::atomic<unsigned long long> lds_set{0};
  if (hipError_t e = ensure_dynamic_lds((const void *)kern, (int)lds, lds_set); e != hipSuccess) return e;
  GemmArgs args = a;
  args.tiles_m = (a.m + 31) / 32;
  args.tiles_n = (a.n + 31) / 32;
  for (int done = 0; done < n_items; done += 65535) { // gridDim.y limit
    const int n = n_items - done < 65535 ? n_items - done : 65535;
    hipLaunchKernelGGL(kern, dim3(args.tiles_m * args.tiles_n, n), dim3(256), lds, s, args, items ? items + done : items);
  }
  return hipGetLastError();
}

#define g_num_cus device_cu_count() /* compute units of the current device (gemm_common.h) */
constexpr int SPLIT_MAX_WG = 16; // = SPLIT_MAX of split_scratch.h

hipError_t launch_bf16_grouped64(const GemmArgs &a, const WorkItem *items, int n_items, hipStream_t s); // brgemm_bf16.hip
hipError_t launch_bf16_small32(const GemmArgs &a, const WorkItem *items, int n_items, hipStream_t s, int split = 1); // brgemm_bf16_small.hip
// flat-B bf16 for the loader-wave tiles: 16-byte row pieces of A, B and C, 64-k chunks, 32-bit lane offsets
static bool bf16_flat_eligible(const GemmDesc &d) {
  return d.dtype == DT_BF16 && !d.vnni_b && !d.vnni_c && d.k > 0 && d.k % 64 == 0 && d.m % 32 == 0 && d.n % 64 == 0 &&
         !((d.lda | d.ldb | d.ldc | d.stride_a | d.stride_b) & 7) && d.lda < (1 << 22) && d.ldb < (1 << 21) && d.ldc < (1 << 22);
}
// VNNI-4 B for the loader-wave tiles: k-group rows of 8 * ldb bytes in 16-byte pieces, 64-k chunks, 32-bit lane offsets
static bool bf16_vnni4_eligible(const GemmDesc &d) {
  return d.dtype == DT_BF16 && d.vnni_b && d.vnni_factor == 4 && !d.vnni_c && d.k > 0 && d.k % 64 == 0 && d.m % 32 == 0 && d.n % 64 == 0 &&
         !((d.lda | d.ldc | d.stride_a | d.stride_b) & 7) && !(d.ldb & 1) && d.lda < (1 << 22) && d.ldb < (1 << 20) && d.ldc < (1 << 22);
}
static bool bf16_small_eligible(const GemmDesc &d) {
  return d.dtype == DT_BF16 && d.vnni_b && d.vnni_factor == 2 && d.m % 32 == 0 && d.n % 32 == 0 && d.k > 0 && d.k % 16 == 0 && !(d.lda & 7) &&
         !(d.stride_a & 7) && !(d.stride_b & 1) && !(d.ldc & 3);
}

// the f32 chain tile (brgemm_f32_lw.hip, launch_f32_chain) a whole-layer f32 descriptor was planned on: 1 / 2 / 3, or -1 (another
// kernel family, a VNNI operand, k not in 64-k chunks ... - whatever pick_f32_variant sent elsewhere)
int f32_chain_tile(const GemmDesc &d) {
  if (d.dtype != DT_F32 || d.vnni_b || d.vnni_c || d.k <= 0 || d.k % BK) return -1;
  // (the 32x32 + K4 tile - the reference's batch-256 layers, 3.4 us of MFMA work per tile - is NOT chained: measured 21.9 us per
  // three-layer step as one launch against 20.6 as three, profiles/r04_f32_chain.txt: a seam is four dependent memory round trips
  // - store drain, counter add, poll, A fetch - and 32 producers + 32 pollers share one counter line; the 64-row tiles gain 1-2.5 %)
  switch (d.variant) {
  case V_F32_LW_64x64K2: return 1;
  case V_F32_LW_64x32K2: return 2;
  default: return -1;
  }
}

int bf16_lw_b_kind(const GemmDesc &d) {
  if (d.dtype != DT_BF16 || d.vnni_c) return -1;
  if (d.vnni_b && d.vnni_factor == 4) return bf16_vnni4_eligible(d) ? 4 : -1;
  if (d.vnni_b) return bf16_fast_eligible(d) ? 0 : -1;
  return bf16_flat_eligible(d) ? 2 : -1;
}

// How many workgroups share the batch-reduce range of ONE output tile (SPLIT kernels of brgemm_f32_lw.hip) - 1 = no split.
// tile: 1 = 64x64 + K2, 2 = 64x32 + K4, 3 = 32x32 + K4; tiles: output tiles of the whole launch; chunks: 64-k chunks per tile.
// Fitted to profiles/r05_split_sweep.txt (whole-layer calls of the reference's skinny benchmark shapes over tile x split count):
//  * a workgroup alone on a CU needs c(tile) us per chunk (the matrix pipes' rate: 0.213 us for a 32x32x64 chunk);
//  * a split costs R = 2.6-2.7 us whatever the count - three dependent trips to the memory side: partial tile written through
//    and acknowledged, arrival counter, the other partials read back - so it pays only where it takes more than that off the K loop;
//  * more workgroups than CUs never paid: 128 x 1024 x 4096 on 32x32 tiles 17.1 us unsplit, 12.9 (S = 2: 256 workgroups), 13.6 (S = 4:
//    512), 15.6 (6), 20.6 (8) - every further round of workgroups pays its own prologue and hand-off.
// Hence: the largest S with tiles * S <= CUs, if the K-loop time it saves exceeds R by a margin. The answer depends on the
// descriptor, the batch count and the number of tiles in the launch only: the same call pattern always adds in the same order.
// xsmm_hip_force_split / TPP_HIP_SPLIT: 0 / 1 = never split, n > 1 = always n (clamped to the chunks), -1 = this model.
static std::atomic<int> g_forced_split{[] {
  const char *e = getenv("TPP_HIP_SPLIT");
  return e ? atoi(e) : -1;
}()};
int force_gemm_split(int v) { return g_forced_split.exchange(v < -1 ? -1 : v); }
static std::atomic<int> g_strict_kernels{0};
int set_strict_kernels(int on) { return g_strict_kernels.exchange(on != 0); }
bool strict_kernels() { return g_strict_kernels.load(std::memory_order_relaxed) != 0; }
static int choose_f32_split(int tile, long long tiles, long long chunks) {
  const int forced = g_forced_split.load(std::memory_order_relaxed);
  if (tile < 1 || tile > 3 || tiles <= 0 || chunks < 2) return 1;
  const long long smax = chunks < SPLIT_MAX_WG ? chunks : SPLIT_MAX_WG;
  if (forced >= 0) return forced <= 1 ? 1 : (int)(forced < smax ? forced : smax);
  const double c = tile == 1 ? 0.92 : tile == 2 ? 0.46 : 0.213;
  long long S = g_num_cus / tiles;
  if (S > smax) S = smax;
  if (S > chunks / 4) S = chunks / 4; // at least four chunks per workgroup
  if (S < 2) return 1;
  const long long per = (chunks + S - 1) / S;
  const double saved = c * (double)(chunks - per);
  return saved > 2.7 + 0.8 ? (int)S : 1;
}

static bool lw16_on() {
  static const bool on = [] {
    const char *e = getenv("TPP_HIP_F32_LW16"); // A/B runs: 0 = never the 32x16 tiles
    return !e || atoi(e) != 0;
  }();
  return on;
}
static int64_t bf16_long_k() {
  static const int64_t v = [] {
    const char *e = getenv("TPP_HIP_BF16_LONG_K"); // A/B runs: the reduction length from which small bf16 outputs leave the 32x32 K-split kernel
    return e ? (int64_t)atoll(e) : (int64_t)1536;
  }();
  return v;
}
static bool bf16_lw32_on() {
  static const bool on = [] {
    const char *e = getenv("TPP_HIP_BF16_LW32"); // A/B runs: 0 = long-reduction bf16 layers stay on the 32x64 tile
    return !e || atoi(e) != 0;
  }();
  return on;
}
static bool k32_pairs_on() {
  static const bool on = [] {
    const char *e = getenv("TPP_HIP_GROUPED_K32_PAIRS"); // A/B runs: 0 = 32-k tiles on the generic grouped kernel, as before round 4
#ifdef TPP_GROUPED_FAST
    (void)e;
    return false;
#else
    return !e || atoi(e) != 0;
#endif
  }();
  return on;
}
// name of the kernel family of the most recent grouped GEMM launch (xsmm_hip_last_grouped_kernel: tests and tools/tpp_replay
// report which kernel a tile-queue group ran on - the descriptor's own name is what a SINGLE invoke would run on)
static std::atomic<const char *> g_last_grouped{""};
const char *last_grouped_kernel() { return g_last_grouped.load(std::memory_order_relaxed); }
// the same for single launches whose kernel was refined at INVOKE time (the batch count arrives with the invoke): "" = the
// descriptor's own kernel (xsmm_hip_kernel_name) ran
static std::atomic<const char *> g_last_refined{""};
const char *last_refined_kernel() { return g_last_refined.load(std::memory_order_relaxed); }
#define note_grouped(name, ...) (g_last_grouped.store(name, std::memory_order_relaxed), (__VA_ARGS__))
hipError_t launch_gemm_grouped(const GemmDesc &d, const WorkItem *items, int n_items, bool vec_ok, bool out_ok, bool pair_ok,
                               int64_t br_hint, hipStream_t stream) {
  if (d.m <= 0 || d.n <= 0 || n_items <= 0) return hipSuccess;
  // n_dec: the number of items every size-dependent DECISION below is taken for. Normally the group's - the group is what fills the
  // chip. In strict mode 1: a single invoke, the first pass of a queued group and its replays then all run on the same kernel.
  const int64_t n_dec = strict_kernels() ? 1 : n_items;
  GemmArgs a;
  a.A = a.B = a.D = nullptr; a.C = nullptr;
  a.lda = d.lda; a.ldb = d.ldb; a.ldc = d.ldc; a.stride_a = d.stride_a; a.stride_b = d.stride_b;
  a.m = (int)d.m; a.n = (int)d.n; a.k = (int)d.k; a.br = 0;
  a.ep = (d.beta0 ? EP_BETA0 : 0) | (d.bias ? EP_BIAS : 0) | (d.relu ? EP_RELU : 0) | (d.vnni_c ? EP_VNNI_C : 0);
  a.tiles_m = a.tiles_n = 0;
  a.vf = d.vnni_factor ? d.vnni_factor : 2;
  a.split = 0; a.scratch = nullptr; a.split_cnt = nullptr;
  a.b_trans = d.b_trans;
  if (d.b_trans) { // B read transposed (a folded xsmm.unary transpose): the generic kernel's element-wise loads
    if (d.dtype != DT_F32 || d.vnni_b) return hipErrorInvalidValue;
    return note_grouped("brgemm_grouped<f32>, B read transposed", launch_grouped_t<float, false, false>(a, items, n_items, stream));
  }
  const bool tiles_ok = vec_ok && d.n % 4 == 0 && d.k % GK == 0; // 16-byte pieces; ragged m / n edges are predicated
  const bool vec = vec_ok && d.n % 4 == 0 && d.k % 4 == 0 && d.dtype == DT_F32 && !d.vnni_b && !((d.lda | d.ldb | d.stride_a | d.stride_b) & 3) &&
                   d.lda < (1 << 24) && d.ldb < (1 << 24); // (32-bit tile-relative lane offsets: 32 rows x ld x 4 B < 2^31)
  // f32 tiles with k a multiple of 64 (mlir-gen --tiles=64,64,64, the most common setting of the reference's
  // benchmark configs): the fast tile families in grouped mode, the largest tile that still yields about one
  // workgroup per CU over the whole work list (the same rule as pick_f32_variant)
  // ... and 32-k tiles (--tiles=32,32,32, the reference's MLP benchmark) when every batch count is even: the loader waves build a
  // 64-k chunk from the blocks of two batch elements (brgemm_f32_lw.hip, pair mode)
  const bool k_pairs = k32_pairs_on() && d.k == 32 && pair_ok && d.stride_a >= 0 && d.stride_b >= 0 && d.stride_a < (1 << 26) && d.stride_b < (1 << 26);
  // (n that is not a multiple of 32 - the reference's --tiles=64,48,64 / 32,48,32 configs: the last 32-column tile of an item is
  // ragged, the loader-wave kernels clamp its loads and mask its stores; needs the 16-byte output pieces of out_ok and ldc % 4)
  // skinny groups - at most one 32x16 tile per CU over the whole work list: the half-width tiles (brgemm_f32_lw16.hip), every CU a
  // workgroup without a hand-off. 64-k tiles, or 32-k tiles whose n is not a multiple of 32 with even batch counts (--tiles=32,48,32);
  // plain 32x32x32 tiles stay on the pair kernel whatever the group size (a single invoke and its group add in the same order:
  // what tools/queue_fuzz.py checks bit for bit)
  {
    const int l16_tile = 0;
    const int64_t t16 = (d.m % 32 == 0 && d.n % 16 == 0) ? n_dec * (d.m / 32) * (d.n / 16) : 0;
    const bool k_ok = (d.k % BK == 0 && d.k > 0) || (k_pairs && d.n % 32 != 0);
    static const char *const l16_names[1][2] = {{"brgemm_f32_lw16<32x16,k4> grouped", "brgemm_f32_lw16<32x16,k4> grouped, 32-k pairs"}};
    if (vec && out_ok && lw16_on() && !d.generic_forced && t16 > 0 && t16 <= g_num_cus && k_ok && d.ldc % 4 == 0 && n_items <= 65535 && d.lda < (1 << 22) &&
        d.ldb < (1 << 22) && d.ldc < (1 << 22) && (!d.bias || out_ok) &&
        !((d.k == 32 ? br_hint / 2 : br_hint * (d.k / BK)) >= 48 && d.n % 32 == 0 && d.k % BK == 0)) // (long reductions: the split 32x32 tiles below, as launch_gemm)
      return note_grouped(l16_names[l16_tile][d.k == 32], launch_f32_lw16(l16_tile, a, items, n_items, true, stream));
  }
  const bool n_ragged = d.n % 32 != 0;
  const bool fam_ok = n_ragged ? (!d.generic_forced && d.n > 32 && (d.k % BK == 0 || k_pairs)) // (plan_gemm knows no tile for such an n: variant = generic)
                               : ((d.k % BK == 0 && d.variant != V_GENERIC) || (k_pairs && !d.generic_forced));
  if (vec && d.m % 32 == 0 && fam_ok && d.lda < (1 << 22) && d.ldb < (1 << 22) && d.ldc < (1 << 22)) {
    const int64_t t64 = (d.m % 64 == 0 && d.n % 64 == 0) ? n_dec * (d.m / 64) * (d.n / 64) : 0;
    const int64_t t6432 = (d.m % 64 == 0) ? n_dec * (d.m / 64) * ((d.n + 31) / 32) : 0;
    if (n_items <= 65535 * 2) { // grid.x carries the item index (x split)
      auto nm = [&](const char *plain, const char *pairs) { return d.k == 32 ? pairs : plain; };
      (void)nm;
      // the loader-wave kernels (brgemm_f32_lw.hip) in grouped mode; TPP_GROUPED_FAST builds the round-1 register-staged family for A/B runs
#ifdef TPP_GROUPED_FAST
      if (t64 >= g_num_cus) return note_grouped("brgemm_f32_fast<64x64> grouped", launch_fast_grouped_t<2, 2, 1, TPP_NACC, true>(a, items, n_items, stream));
      if (t6432 >= g_num_cus) return note_grouped("brgemm_f32_fast<64x32,k2> grouped", launch_fast_grouped_t<2, 1, 2, TPP_NACC, true>(a, items, n_items, stream));
      return note_grouped("brgemm_f32_fast<32x32,k4> grouped", launch_fast_grouped_t<1, 1, 4, TPP_NACC, false>(a, items, n_items, stream));
#else
      if (t64 >= g_num_cus) return note_grouped(t64 >= 2 * g_num_cus ? nm("brgemm_f32_lw<64x64> grouped", "brgemm_f32_lw<64x64> grouped, 32-k pairs") : nm("brgemm_f32_lw<64x64,k2> grouped", "brgemm_f32_lw<64x64,k2> grouped, 32-k pairs"), launch_f32_lw_grouped(t64 >= 2 * g_num_cus ? 0 : 1, a, items, n_items, 1, stream));
      const int64_t t32 = n_dec * (d.m / 32) * ((d.n + 31) / 32);
      // (rounds of workgroups x per-chunk time, as pick_f32_variant: 1.5 rounds of 64x32 tiles lose to 3 half-rounds of 32x32 tiles)
      if (t6432 >= g_num_cus && !(0.23 * 1.05 * (double)((t32 + g_num_cus - 1) / g_num_cus) < 0.46 * (double)((t6432 + g_num_cus - 1) / g_num_cus)))
        return note_grouped(nm("brgemm_f32_lw<64x32,k4> grouped", "brgemm_f32_lw<64x32,k4> grouped, 32-k pairs"), launch_f32_lw_grouped(2, a, items, n_items, 1, stream));
      // skinny groups (fewer 64x32 tiles than CUs): 32x32 tiles, and the batch-reduce range of a tile over several workgroups
      // when the model says so (choose_f32_split: from the descriptor, the first item's batch count and the group's size)
      const int64_t chunks = d.k == 32 ? br_hint / 2 : br_hint * (d.k / BK);
      const int S = choose_f32_split(3, t32, chunks);
      static const char *const split_names[2] = {"brgemm_f32_lw<32x32,k4> grouped, split", "brgemm_f32_lw<32x32,k4> grouped, 32-k pairs, split"};
      if (S > 1) return note_grouped(split_names[d.k == 32], launch_f32_lw_grouped(3, a, items, n_items, S, stream));
      return note_grouped(nm("brgemm_f32_lw<32x32,k4> grouped", "brgemm_f32_lw<32x32,k4> grouped, 32-k pairs"), launch_f32_lw_grouped(3, a, items, n_items, 1, stream));
#endif
    }
  }
  // bf16 + VNNI-2 B with 16-byte loads: 8-element A pieces, pair-rows of B 16-byte aligned
  const bool vec16x = d.dtype == DT_BF16 && d.vnni_b && !((d.lda | d.stride_a | d.stride_b) & 7) && d.lda < (1 << 21) && d.ldb < (1 << 21);
  const bool vec16_4 = vec16x && d.vnni_factor == 4 && vec_ok && d.n % 2 == 0 && d.k % GK == 0 && !(d.ldb & 1); // VNNI-4: 16-byte pieces of 2 columns
  const bool vec16 = tiles_ok && d.dtype == DT_BF16 && d.vnni_b && d.vnni_factor == 2 && !((d.lda | d.stride_a | d.stride_b) & 7) && !(d.ldb & 3) &&
                     d.lda < (1 << 21) && d.ldb < (1 << 21); // (32-bit lane offsets)
  // bf16 tile invokes with k a multiple of 64 and n a multiple of 64 (the reference's --tiles=64,64,64 / 32,64,64 bf16 rows): the
  // LOADER-WAVE tiles in grouped mode (round 6, brgemm_bf16_lw.hip launch_bf16_lw_grouped) - what the same layer runs on as one
  // whole-layer call. Tile by the model of pick_bf16_lw_tile over the group's workgroups: 32x64 + K2 (two workgroups per 64-row
  // item) or 64x64. Against the two older grouped kernels (profiles/r06_bf16_sweep_before.txt, forced-variant rows): the loader-wave
  // tiles win whenever the group fills 3/4 of the chip with 64x64 tiles (1024 x 1024 x 512: 5.1 us against 6.7) or the reduction is
  // long (16 chunks or more: 128 x 4096 x 1024 5.3 against 9.4) or half the chip gets a 32x64 tile of at least 8 chunks (128 x 3072 x 768: 4.9 against 6.3; 1024 x 512 x 256, 4 chunks: 5.6 against 5.1);
  // short reductions of small groups stay on the K-split kernel (128 x 768 x 768: 4.3 against 4.7). TPP_HIP_BF16_LW_GROUPED=0 switches the path off (A/B runs).
  {
    static const bool lwg_on = [] {
      const char *e = getenv("TPP_HIP_BF16_LW_GROUPED");
      return !e || atoi(e) != 0;
    }();
    const bool v2 = d.vnni_factor == 2, v4 = d.vnni_factor == 4;
    const bool shape_ok = d.dtype == DT_BF16 && d.vnni_b && (v2 || v4) && !d.vnni_c && !d.generic_forced && !d.variant_forced && d.k > 0 && d.k % BK == 0 &&
                          d.m % 32 == 0 && d.n % 64 == 0 && !((d.lda | d.stride_a | d.stride_b | d.ldc) & 7) && !(d.ldb & (v2 ? 3 : 1)) &&
                          d.lda < (1 << 21) && d.ldb < (1 << 20) && d.ldc < (1 << 22) && d.stride_a >= 0 && d.stride_b >= 0;
    // RAGGED n (round 6): items whose n is 16 more than a multiple of 32 - the reference's --tiles=64,48,64 rows (fc / matmul 128x768x2304)
    // - on the 32x32 + K2 instance: ceil(n / 32) column tiles per item, the last moved left to end at column n (it recomputes the 16
    // columns it shares with its neighbour and stores its own 16: brgemm_bf16_lw.hip skip_cols). Skinny groups with a long reduction
    // only, like the instance's other uses; everything else with such an n stays on the K-split kernel below.
    {
      static const bool ragged_on = [] {
        const char *e = getenv("TPP_HIP_BF16_LW_RAGGED");
        return !e || atoi(e) != 0;
      }();
      const bool ragged_ok = ragged_on && d.dtype == DT_BF16 && d.vnni_b && (v2 || v4) && !d.vnni_c && !d.generic_forced && !d.variant_forced && d.k > 0 && d.k % BK == 0 &&
                             d.m % 32 == 0 && d.n % 32 == 16 && d.n >= 48 && !((d.lda | d.stride_a | d.stride_b | d.ldc) & 7) && !(d.ldb & (v2 ? 3 : 1)) &&
                             d.lda < (1 << 21) && d.ldb < (1 << 20) && d.ldc < (1 << 22) && d.stride_a >= 0 && d.stride_b >= 0;
      if (lwg_on && ragged_ok && vec_ok && out_ok && br_hint >= 1 && g_forced_split.load(std::memory_order_relaxed) < 0) {
        const int64_t chunks = br_hint * (d.k / BK);
        const int64_t wg4 = n_dec * (d.m / 32) * ((d.n + 31) / 32);
        if (wg4 <= (int64_t)g_num_cus && chunks >= 16) {
          ChainArgs c;
          memset(&c, 0, sizeof(c));
          c.lda = d.lda;
          c.m = (int)d.m;
          c.n = (int)d.n;
          c.nlayers = 1;
          c.L[0] = ChainLayer{nullptr, nullptr, nullptr, d.ldb, d.ldc, d.stride_a, d.stride_b, (int)d.k, (int)br_hint, a.ep, 0};
          const bool even = ((d.k / BK) % 2 == 0) || pair_ok;
          return note_grouped(v4 ? "brgemm_bf16_lw_vnni4<32x32,k2> grouped, ragged n" : "brgemm_bf16_lw<32x32,k2> grouped, ragged n",
                              launch_bf16_lw_grouped(4, v4 ? 4 : 0, c, items, n_items, even, stream));
        }
      }
    }
    if (lwg_on && shape_ok && vec_ok && out_ok && br_hint >= 1 && g_forced_split.load(std::memory_order_relaxed) < 0) { // (a forced split count: the K-split kernel below)
      const int64_t chunks = br_hint * (d.k / BK);
      const int64_t t64 = d.m % 64 == 0 ? n_dec * (d.m / 64) * (d.n / 64) : 0;
      const int64_t wg0 = n_dec * (d.m / 32) * (d.n / 64);
      if (t64 * 4 >= 3 * (int64_t)g_num_cus || chunks >= 16 || (wg0 * 2 >= (int64_t)g_num_cus && chunks >= 8)) {
        static const double ca[2] = {3.56, 3.75}, cb[2] = {0.098, 0.135};
        const double c0 = (double)((wg0 + g_num_cus - 1) / g_num_cus) * (ca[0] + cb[0] * (double)chunks);
        const double c1 = t64 > 0 ? (double)((t64 + g_num_cus - 1) / g_num_cus) * (ca[1] + cb[1] * (double)chunks) : 1e30;
        int tile = c1 <= c0 ? 1 : 0;
        // 32x32 + K2 (VNNI-2): twice the workgroups of the 32x64 tile pulling panels - for skinny groups with a long reduction, as
        // launch_gemm does for the whole-layer call (one round of workgroups at most, 16 chunks or more; (a, b) = (3.52, 0.072) from
        // 128 x 1024 x 1024 / x 4096 on that tile)
        const int64_t wg4 = n_dec * (d.m / 32) * (d.n / 32);
        static const bool t4_v4 = [] { // (A/B runs: TPP_HIP_BF16_LW_T4_VNNI4=0 keeps VNNI-4 groups on the 32x64 / 64x64 tiles)
          const char *e = getenv("TPP_HIP_BF16_LW_T4_VNNI4");
          return !e || atoi(e) != 0;
        }();
        if ((v2 || (v4 && t4_v4)) && wg4 <= (int64_t)g_num_cus && chunks >= 16 && 3.52 + 0.072 * (double)chunks < (c1 < c0 ? c1 : c0)) tile = 4;
        ChainArgs c;
        memset(&c, 0, sizeof(c));
        c.lda = d.lda;
        c.m = (int)d.m;
        c.n = (int)d.n;
        c.nlayers = 1;
        c.L[0] = ChainLayer{nullptr, nullptr, nullptr, d.ldb, d.ldc, d.stride_a, d.stride_b, (int)d.k, (int)br_hint, a.ep, 0};
        const bool even = ((d.k / BK) % 2 == 0) || pair_ok;
        static const char *const names[2][2] = {{"brgemm_bf16_lw<32x64,k2> grouped", "brgemm_bf16_lw<64x64> grouped"},
                                                {"brgemm_bf16_lw_vnni4<32x64,k2> grouped", "brgemm_bf16_lw_vnni4<64x64> grouped"}};
        return note_grouped(tile == 4 ? (v4 ? "brgemm_bf16_lw_vnni4<32x32,k2> grouped" : "brgemm_bf16_lw<32x32,k2> grouped") : names[v4 ? 1 : 0][tile],
                            launch_bf16_lw_grouped(tile, v4 ? 4 : 0, c, items, n_items, even, stream));
      }
    }
  }
  // bf16 tiles of 64x64 with k a multiple of 64: the 64x64 bf16 family in grouped mode (it stores 16-byte
  // row pieces and reads the bias 8 bytes at a time: checked per item by the queue through out_ok)
  // (whatever a SINGLE invoke of the handle would run on - a lone 64x64 tile is planned on the 32x32 K-split kernel -, the GROUP is
  // what fills the chip: round 5, the reference's fc / matmul shapes as 64,64,64 tile invokes: 1024 x 2560 x 1024 30.4 us on 32x32
  // tiles against 15 us whole-layer)
  if (vec16 && out_ok && d.variant >= V_BF16_FAST && (d.variant != V_BF16_SMALL32 || !d.variant_forced) && !d.generic_forced && bf16_fast_eligible(d) &&
      n_dec * (d.m / 64) * (d.n / 64) >= (3 * g_num_cus) / 4)
    return note_grouped("brgemm_bf16_fast<64x64> grouped", launch_bf16_grouped64(a, items, n_items, stream));
  // ... and the same family on a VNNI-4 B operand (--vnni=4 tile invokes: benchmarks/config/*/*_dp4_*; the generic kernel's MFMA path
  // took 30 us for 1024 x 2560 x 1024 against 19.5 on VNNI-2)
  if (vec16_4 && out_ok && !d.generic_forced && !d.vnni_c && d.k % BK == 0 && d.m % 64 == 0 && d.n % 64 == 0 && !((d.ldc | d.stride_b) & 7) && d.ldc < (1 << 22) &&
      d.lda < (1 << 22) && d.ldb < (1 << 20) && n_dec * (d.m / 64) * (d.n / 64) >= (3 * g_num_cus) / 4)
    return note_grouped("brgemm_bf16_fast_vnni4<64x64> grouped", launch_bf16_grouped64(a, items, n_items, stream));
  // (VNNI-4 tile invokes - the compiler-native 32x32x32 tiles of a --vnni=4 pipeline, small groups of 64x64x64 tiles - on the same
  // kernel: its B fragment is then two 8-byte loads; a single invoke of such a handle stays on the generic kernel's MFMA path.
  // And tiles whose n is a multiple of 4 but not of 32 (--tiles=64,48,64): a masked last column tile instead of the generic kernel.)
  const bool small_base = d.dtype == DT_BF16 && d.vnni_b && !d.vnni_c && !d.generic_forced && d.m % 32 == 0 && d.n >= 32 && d.n % 4 == 0 && d.k > 0 &&
                          d.k % 16 == 0 && !(d.lda & 7) && !(d.stride_a & 7) && !(d.ldc & 3);
  const bool small4 = small_base && d.vnni_factor == 4 && !(d.stride_b & 3);
  const bool small_ragged = small_base && d.vnni_factor == 2 && d.n % 32 != 0 && !(d.stride_b & 1);
  if (vec_ok && out_ok && ((d.variant != V_GENERIC && bf16_small_eligible(d)) || small_ragged || small4)) {
    // skinny groups with a long reduction: the K steps of a tile over several workgroups (the kernel is a latency-bound stream: 0.047 us
    // per 16-k step of a workgroup). Measured (profiles/r05_bf16_skinny_small_vs_lw.txt): it pays only while every workgroup still has
    // a CU to itself - 128 x 1024 x 4096 as 64x64x64 tile invokes 12.0 -> 9.8 us at S = 2 (10.3 at 4, 13.1 at 8), 256 x 1024 x 4096
    // 12.3 -> 14.2 at S = 2. Hence the largest count with tiles x S <= CUs and at least 32 steps per workgroup, if it saves more
    // than the hand-off costs. xsmm_hip_force_split overrides.
    const long long t32 = (long long)n_dec * (d.m / 32) * ((d.n + 31) / 32), steps = (long long)br_hint * (d.k / 16);
    int S = 1;
    const int forced = g_forced_split.load(std::memory_order_relaxed);
    if (forced >= 0) S = forced <= 1 ? 1 : (int)(forced < 16 ? forced : 16);
    else if (t32 > 0) {
      long long c = (long long)g_num_cus / t32;
      if (c > 16) c = 16;
      if (c > steps / 32) c = steps / 32;
      if (c >= 2 && 0.047 * (double)(steps - (steps + c - 1) / c) > 2.7 + 0.8) S = (int)c;
    }
    if (S > (int)steps) S = steps > 1 ? (int)steps : 1;
    if (S > 1) return note_grouped(small4 ? "brgemm_bf16_small32_vnni4 grouped, split" : "brgemm_bf16_small32 grouped, split", launch_bf16_small32(a, items, n_items, stream, S));
    return note_grouped(small4 ? "brgemm_bf16_small32_vnni4 grouped" : "brgemm_bf16_small32 grouped", launch_bf16_small32(a, items, n_items, stream));
  }
  if (d.dtype == DT_F32) return note_grouped("brgemm_grouped<f32>", vec ? launch_grouped_t<float, false, true>(a, items, n_items, stream)
                                                                        : launch_grouped_t<float, false, false>(a, items, n_items, stream));
  if (d.vnni_b && vec16_4) return note_grouped("brgemm_grouped<bf16,vnni4>", launch_grouped_t<unsigned short, true, true, 4>(a, items, n_items, stream)); // VNNI-4 on the bf16 MFMA path
  if (d.vnni_b) return note_grouped("brgemm_grouped<bf16,vnni2>", vec16 ? launch_grouped_t<unsigned short, true, true>(a, items, n_items, stream)
                                                                        : launch_grouped_t<unsigned short, true, false>(a, items, n_items, stream));
  return note_grouped("brgemm_grouped<bf16,flat>", launch_grouped_t<unsigned short, false, false>(a, items, n_items, stream));
}

// QUADS (round 6; xsmm_desc.h QuadItem, brgemm_bf16_lw.hip GRP = 2): a group of 64x64 bf16 tile invokes that forms a grid of item rows and
// item columns runs as 2 x 2 blocks on the 128x128 loader-wave tile when the tile model says so - the kernel the same layer gets as
// ONE whole-layer call once it is large enough (pick_bf16_lw_tile). 1024 x 2560 x 1024 as 640 invokes: 3 rounds of 64x64 tiles
// (15.3 us) against 160 workgroups of 128x128 (whole-layer call 10.2 us). TPP_HIP_BF16_QUADS=0: off (A/B runs).
static bool quads_shape_ok(const GemmDesc &d) {
  const bool v2 = d.vnni_factor == 2 || d.vnni_factor == 0, v4 = d.vnni_factor == 4;
  return d.dtype == DT_BF16 && d.vnni_b && (v2 || v4) && !d.vnni_c && !d.b_trans && !d.generic_forced && !d.variant_forced && d.m == 64 && d.n == 64 && d.k > 0 &&
         d.k % BK == 0 && !((d.lda | d.stride_a | d.stride_b | d.ldc) & 7) && !(d.ldb & (v4 ? 1 : 3)) && d.lda < (1 << 21) && d.ldb < (1 << 20) &&
         d.ldc < (1 << 22) && d.stride_a >= 0 && d.stride_b >= 0;
}
bool gemm_quads_pay(const GemmDesc &d, int n_items, int64_t br) {
  static const bool on = [] {
    const char *e = getenv("TPP_HIP_BF16_QUADS");
    return !e || atoi(e) != 0;
  }();
  if (!on || strict_kernels() || !quads_shape_ok(d) || br < 1 || n_items < 4 || (n_items & 3) || g_forced_split.load(std::memory_order_relaxed) >= 0) return false;
  const double chunks = (double)(br * (d.k / BK));
  const int64_t cus = g_num_cus;
  // (a, b) of the tile model: 64x64 (3.75, 0.135), 32x64 + K2 (3.56, 0.098) - what the grouped path would pick from - and 128x128 (6.06, 0.236)
  const double c64 = (double)((n_items + cus - 1) / cus) * (3.75 + 0.135 * chunks), c32 = (double)((2 * (int64_t)n_items + cus - 1) / cus) * (3.56 + 0.098 * chunks);
  const double cq = (double)((n_items / 4 + cus - 1) / cus) * (6.06 + 0.236 * chunks);
  return cq * 1.05 < (c64 < c32 ? c64 : c32);
}
hipError_t launch_gemm_quads(const GemmDesc &d, const QuadItem *quads, int n_quads, int64_t br, hipStream_t stream) {
  if (!quads_shape_ok(d) || br < 1 || n_quads <= 0) return hipErrorInvalidValue;
  ChainArgs c;
  memset(&c, 0, sizeof(c));
  c.lda = d.lda;
  c.m = 128;
  c.n = 128;
  c.nlayers = 1;
  const int ep = (d.beta0 ? EP_BETA0 : 0) | (d.bias ? EP_BIAS : 0) | (d.relu ? EP_RELU : 0);
  c.L[0] = ChainLayer{nullptr, nullptr, nullptr, d.ldb, d.ldc, d.stride_a, d.stride_b, (int)d.k, (int)br, ep, 0};
  const bool v4 = d.vnni_factor == 4;
  return note_grouped(v4 ? "brgemm_bf16_lw_vnni4<128x128> quads" : "brgemm_bf16_lw<128x128> quads", launch_bf16_lw_quads(v4 ? 4 : 0, c, quads, n_quads, stream));
}


static int pick_f32_variant(const GemmDesc &d) {
  if (d.k <= 0 || d.k % BK) return V_GENERIC;
  if ((d.lda | d.ldb | d.stride_a | d.stride_b) & 3) return V_GENERIC;
  if (d.lda >= (1 << 22) || d.ldb >= (1 << 22) || d.ldc >= (1 << 22)) return V_GENERIC; // 32-bit lane offsets
  const int64_t m = d.m, n = d.n;
  auto tiles = [&](int bm, int bn) { return (m % bm == 0 && n % bn == 0) ? (m / bm) * (n / bn) : 0; };
  // 64-row tiles run on the loader-wave kernels (brgemm_f32_lw.hip). Measured on C2 (256 tiles of 64x64, uniform
  // [-1, 1) inputs, profiles/r02_f32_variants.txt): 64x64 with the K chunks split over two wave groups (two MFMA
  // waves per SIMD cover each other's barrier stalls) 18.3 us, one group 18.7 us, the round-1 kernel 19.7-20.4 us.
  // Outputs with at least one 64x64 tile per CU: 64x64 or 128x64 tiles, whichever needs less time over its rounds
  // of workgroups (one per CU at a time). A 128x64 round takes ~1.85x a 64x64 round (measured, K = 1024: 32.7 vs
  // 17.6 us), so 128x64 wins at 1280-2048 x 1024 (one round instead of two) and for large outputs (0.93x), and
  // loses e.g. at 3072 x 1024 (two rounds against three; tools/sessions/mid_probe.py).
  // Skinny outputs - at most one 32x16 tile per CU (the reference's M = 128 shapes: 128 x 1024 = 256 tiles, 128 x 768 = 192): the
  // half-width tiles of brgemm_f32_lw16.hip put a workgroup on every CU where 32x32 tiles would leave half the chip idle, and need
  // no hand-off between workgroups (the SPLIT launches pay 2.6-2.7 us for one); profiles/r05_lw16_vs_split.txt
  // (a 16x48 tile of the same family - 256 x 768 outputs are exactly 256 of them - was built and measured: 256 x 768 x 768 5.59 us
  // against 5.48 on 192 tiles of 32x32, x 3072 14.9 against 13.7: 16 KiB of panel per 98 kflop chunk, the launch is bound by the
  // L2 -> LDS traffic of all CUs together, ~16 TB/s; removed. profiles/r05_lw16_vs_split.txt)
  if (d.ldc % 4 == 0 && lw16_on() && tiles(32, 16) > 0 && tiles(32, 16) <= g_num_cus) return V_F32_LW16_32x16;
  if (tiles(64, 64) >= g_num_cus) {
    const int64_t r64 = (tiles(64, 64) + g_num_cus - 1) / g_num_cus, r128 = (tiles(128, 64) + g_num_cus - 1) / g_num_cus;
    if (tiles(128, 64) > 0 && 1.85 * (double)r128 < (double)r64) return V_F32_128x64;
    return V_F32_LW_64x64K2;
  }
  if (tiles(64, 32) >= g_num_cus) {
    // one 64x32 workgroup per CU at a time (96 KiB of LDS), two 32x32 ones (64 KiB): rounds x per-chunk time. 256 x 3072 (384 tiles of
    // 64x32 = 1.5 rounds) measured 15.9 us against 12.6 on 768 tiles of 32x32; where the rounds tie (C3: 256 / 512 tiles, 128 x 4096)
    // the larger tile stays - same time, half the LDS traffic (profiles/r05_split_sweep.txt)
    const int64_t r6432 = (tiles(64, 32) + g_num_cus - 1) / g_num_cus, r32 = (tiles(32, 32) + g_num_cus - 1) / g_num_cus;
    if (tiles(32, 32) > 0 && 0.23 * 1.05 * (double)r32 < 0.46 * (double)r6432) return V_F32_LW_32x32K4;
    return V_F32_LW_64x32K2;
  }
  if (tiles(32, 32) > 0 && tiles(32, 32) >= tiles(64, 64) * 2 && tiles(64, 32) < g_num_cus) return V_F32_LW_32x32K4;
  if (tiles(64, 64) > 0) return V_F32_LW_64x64K2;
  if (tiles(64, 32) > 0) return V_F32_LW_64x32K2;
  if (tiles(32, 32) > 0) return V_F32_LW_32x32K4;
  return V_GENERIC;
}

// Mid-size bf16 outputs: the loader-wave family (brgemm_bf16_lw.hip), the largest tile that still gives at least 3/4 of the CUs
// a workgroup (one workgroup per CU: 160 KiB of LDS). Outputs too small for that even with 32x64 tiles stay with the 32x32 K-split
// family (m = 256, n = 1024, K = 1024: 4.9 us against 5.7 on 128 tiles of 32x64, profiles/r03_sweep_shapes.txt).
// Returns the tile index (0 .. 3) or -1. TPP_HIP_BF16_LW=0 switches the family off (A/B runs).
static int pick_bf16_lw_tile(const GemmDesc &d) {
  static const int enabled = [] {
    const char *e = getenv("TPP_HIP_BF16_LW");
    return e ? atoi(e) : 1;
  }();
  if (!enabled) return -1;
  // Gate (unchanged since round 3): some tile of the family gives at least 3/4 of the CUs a workgroup. Which tile, round 6 - fitted to the
  // sweep of the reference's whole shape set over every tile (tools/bf16_sweep.py, profiles/r06_bf16_sweep.txt): a launch costs
  // rounds x (a + b x chunks), rounds = ceil(tiles / CUs) (one workgroup per CU: a second round is a second kernel's worth), with
  // (a, b) in us from the K = 1024 / K = 4096 pairs of the sweep: 32x64 (3.56, 0.098), 64x64 (3.75, 0.135), 64x128 (4.66, 0.204), 128x128
  // (6.06, 0.236). The old rule - the LARGEST tile that still reaches 3/4 of the CUs - put 1024 x 2560 on 320 tiles of 64x128 (two
  // rounds: 15.1 us) instead of 160 tiles of 128x128 (one round: 10.2 us). The batch count arrives with the invoke: priced at 16 chunks
  // (K = 1024; the order of two candidates flips with K only when their round counts differ AND the sums are within a few percent).
  static const int legacy = [] {
    const char *e = getenv("TPP_HIP_BF16_LW_PICK");
    return e ? atoi(e) == 0 : 0; // TPP_HIP_BF16_LW_PICK=0: the round-3 rule (A/B runs)
  }();
  static const double ca[4] = {3.56, 3.75, 4.66, 6.06}, cb[4] = {0.098, 0.135, 0.204, 0.236};
  bool gate = false;
  int best = -1;
  double best_t = 0;
  for (int t = 3; t >= 0; --t) {
    int bm, bn;
    blw_tile_dims(t, &bm, &bn);
    if (d.m % bm || d.n % bn) continue;
    const int64_t tiles = (d.m / bm) * (d.n / bn);
    if (tiles * 4 >= 3 * (int64_t)g_num_cus) {
      if (legacy) return t;
      gate = true;
    }
    const double cost = (double)((tiles + g_num_cus - 1) / g_num_cus) * (ca[t] + cb[t] * 16.0);
    if (best < 0 || cost < best_t) best = t, best_t = cost;
  }
  return gate ? best : -1;
}

static const char *variant_name(int v) {
  switch (v) {
  case V_BF16_LW_32x64: return "brgemm_bf16_lw<32x64,k2>";
  case V_BF16_LW_64x64: return "brgemm_bf16_lw<64x64>";
  case V_BF16_LW_64x128: return "brgemm_bf16_lw<64x128>";
  case V_BF16_LW_128x128: return "brgemm_bf16_lw<128x128>";
  case V_BF16_LWF_32x64: return "brgemm_bf16_lw_flatb<32x64,k2>";
  case V_BF16_LWF_64x64: return "brgemm_bf16_lw_flatb<64x64>";
  case V_BF16_LWF_64x128: return "brgemm_bf16_lw_flatb<64x128>";
  case V_BF16_LWF_128x128: return "brgemm_bf16_lw_flatb<128x128>";
  case V_BF16_LW4_32x64: return "brgemm_bf16_lw_vnni4<32x64,k2>";
  case V_BF16_LW4_64x64: return "brgemm_bf16_lw_vnni4<64x64>";
  case V_BF16_LW4_64x128: return "brgemm_bf16_lw_vnni4<64x128>";
  case V_BF16_LW4_128x128: return "brgemm_bf16_lw_vnni4<128x128>";
  case V_F32_64x64: return "brgemm_f32_fast<64x64,k1>";
  case V_F32_64x32K2: return "brgemm_f32_fast<64x32,k2>";
  case V_F32_32x32K4: return "brgemm_f32_fast<32x32,k4>";
  case V_F32_128x64: return "brgemm_f32_fast<128x64,k1>";
  case V_F32_64x64K2: return "brgemm_f32_fast<64x64,k2>";
  case V_F32_LW_64x64: return "brgemm_f32_fast_lw<64x64,k1>";
  case V_F32_LW_64x64K2: return "brgemm_f32_fast_lw<64x64,k2>";
  case V_F32_LW_64x32K2: return "brgemm_f32_fast_lw<64x32,k4>";
  case V_F32_LW_32x32K4: return "brgemm_f32_fast_lw<32x32,k4>";
  case V_F32_LW_128x64: return "brgemm_f32_fast_lw<128x64,k1>";
  case V_F32_LW16_32x16: return "brgemm_f32_lw16<32x16,k4>";
  case V_BF16_FAST: return "brgemm_bf16_fast<64x64>";
  case V_BF16_DMA128: return "brgemm_bf16_dma<128x128>";
  case V_BF16_DMA256: return "brgemm_bf16_dma<256x256>";
  case V_BF16_SMALL32: return "brgemm_bf16_small<32x32,k4>";
  default: return "brgemm_grouped(generic)";
  }
}

bool plan_gemm(GemmDesc &d, int forced_variant) {
  int v = V_GENERIC;
  if (d.vnni_c) forced_variant = V_GENERIC; // VNNI-2 C store: the generic kernel's epilogue only
  if (d.dtype == DT_F32 && !d.vnni_b) v = pick_f32_variant(d);
  else if (d.dtype == DT_BF16 && d.vnni_b && d.vnni_factor == 4) {
    // VNNI-4 B ([k/4][n][4]: benchmarks/config/omp/mlir-bf16.json:68-100 `--vnni=4`): the loader-wave tiles with the VNNI-4 image
    // (the largest tile that still gives 3/4 of the CUs a workgroup, else the smallest the shape divides); everything else - ragged
    // shapes, k not a multiple of 64 (the compiler-native 32x32x32 tiles) - on the generic kernel's element path
    if (bf16_vnni4_eligible(d)) {
      int t = pick_bf16_lw_tile(d);
      for (int c = 0; t < 0 && c < 4; ++c) {
        int bm, bn;
        blw_tile_dims(c, &bm, &bn);
        if (d.m % bm == 0 && d.n % bn == 0) t = c;
      }
      if (t >= 0) v = V_BF16_LW4_32x64 + t;
      if (forced_variant >= V_BF16_LW4_32x64 && forced_variant <= V_BF16_LW4_128x128) {
        int bm, bn;
        blw_tile_dims(forced_variant - V_BF16_LW4_32x64, &bm, &bn);
        if (d.m % bm == 0 && d.n % bn == 0) v = forced_variant;
      }
    }
  } else if (d.dtype == DT_BF16 && bf16_fast_eligible(d)) {
    v = V_BF16_FAST + pick_bf16_tile(d);
    // small outputs (e.g. the reference's --batch=256 layers): 32x32 tiles with K split over the waves give
    // every CU a workgroup. Measured crossover with the 64x64 family (n = 1024, K = 1024): 5.1 vs 8.4 us at 64
    // tiles of 64x64, 7.5 vs 8.5 at 128, 12.3 vs 8.9 at 256 (profiles/r01_sweep_shapes.txt)
    if (v == V_BF16_FAST && bf16_small_eligible(d) && (d.m / 64) * (d.n / 64) < (3 * g_num_cus) / 4) v = V_BF16_SMALL32;
    // mid-size outputs (the 64x64 / 32x32 families, or 128x128 tiles for fewer than 3/4 of the CUs): one loader-wave workgroup
    // per CU. Measured (profiles/r03_sweep_shapes.txt, 1024-wide layer, K = 1024): see DESIGN.md 4.2.
    const int64_t t128 = (d.m / 128) * (d.n / 128);
    if (v == V_BF16_FAST || v == V_BF16_SMALL32 || (v == V_BF16_DMA128 && t128 * 4 < 3 * (int64_t)g_num_cus)) {
      const int lw = pick_bf16_lw_tile(d);
      // (round 6: the 128x128 loader-wave tile also where brgemm_bf16_dma128 used to stay - fewer than 3/4 of the CUs busy: 1024 x 2560 x
      // 1024 = 160 tiles runs 10.2 us on it against 11.5 on dma128, profiles/r06_bf16_sweep_before.txt)
      if (lw >= 0) v = V_BF16_LW_32x64 + lw;
    } else if (v == V_BF16_DMA128 && t128 <= (int64_t)g_num_cus && pick_bf16_lw_tile(d) == 3) {
      // ONE round of 128x128 tiles (the C4 layer 4096 x 1024, C5 2048 x 2048): since the end of round 3 the loader-wave tile is
      // at least as fast as brgemm_bf16_dma128 there (same box, profiles/r03_write_through_c_stores.txt: C5 18.2 against 18.7 us,
      // the C4 layer 10.4 against 10.6-11.5) - and it is the tile the 4096-row chain runs on. Several rounds: dma128 (not re-measured).
      v = V_BF16_LW_128x128;
    }
    const int tile = forced_variant - V_BF16_FAST; // a forced bf16 tile is honoured if the shape divides it
    if (tile >= 0 && tile <= 2 && d.m % (64 << tile) == 0 && d.n % (64 << tile) == 0) v = forced_variant;
    if (forced_variant == V_BF16_SMALL32 && bf16_small_eligible(d)) v = forced_variant;
    if (forced_variant >= V_BF16_LW_32x64 && forced_variant <= V_BF16_LW_128x128) {
      int bm, bn;
      blw_tile_dims(forced_variant - V_BF16_LW_32x64, &bm, &bn);
      if (d.m % bm == 0 && d.n % bn == 0) v = forced_variant;
    }
  } else if (d.dtype == DT_BF16 && bf16_flat_eligible(d)) {
    // flat B ([k][n] row-major, what xsmm.unary pack would have turned into VNNI-2): the loader-wave tiles with the interleave
    // in the B loader. The largest tile that still gives 3/4 of the CUs a workgroup, else the smallest the shape divides
    // (launch_gemm falls back to the generic kernel when an operand is not 16-byte aligned).
    int t = pick_bf16_lw_tile(d);
    for (int c = 0; t < 0 && c < 4; ++c) {
      int bm, bn;
      blw_tile_dims(c, &bm, &bn);
      if (d.m % bm == 0 && d.n % bn == 0) t = c;
    }
    if (t >= 0) v = V_BF16_LWF_32x64 + t;
    if (forced_variant >= V_BF16_LWF_32x64 && forced_variant <= V_BF16_LWF_128x128) {
      int bm, bn;
      blw_tile_dims(forced_variant - V_BF16_LWF_32x64, &bm, &bn);
      if (d.m % bm == 0 && d.n % bn == 0) v = forced_variant;
    }
  } else if (d.dtype == DT_BF16 && bf16_small_eligible(d)) {
    v = V_BF16_SMALL32; // k a multiple of 16 only (e.g. the compiler-native 32x32x32 tile), m or n a multiple of 32 only
    if (forced_variant == V_BF16_LW_32x64) { // the 32x64 loader-wave tile needs m % 32 only (bf16_fast_eligible asks for 64)
      GemmDesc e = d;
      e.m = (d.m + 63) / 64 * 64;
      if (d.m % 32 == 0 && d.n % 64 == 0 && bf16_fast_eligible(e)) v = forced_variant;
    }
  }
  if (forced_variant >= 0 && d.dtype == DT_F32 && v != V_GENERIC) {
    // honour the forced tile only if the shape divides it
    const int bm[] = {64, 64, 32, 128, 64, 64, 64, 64}, bn[] = {64, 32, 32, 64, 64, 64, 64, 32};
    if (forced_variant <= 7 && d.m % bm[forced_variant] == 0 && d.n % bn[forced_variant] == 0) v = forced_variant;
    if (forced_variant == V_F32_LW_32x32K4 && d.m % 32 == 0 && d.n % 32 == 0) v = forced_variant;
    if (forced_variant == V_F32_LW_128x64 && d.m % 128 == 0 && d.n % 64 == 0) v = forced_variant;
    if (forced_variant == V_F32_LW16_32x16 && d.m % 32 == 0 && d.n % 16 == 0 && d.ldc % 4 == 0) v = forced_variant;
    if (forced_variant == V_GENERIC) v = V_GENERIC;
  } else if (forced_variant == V_GENERIC) {
    v = V_GENERIC;
  }
  d.variant = v;
  d.generic_forced = forced_variant == V_GENERIC;
  d.variant_forced = forced_variant >= 0 && v == forced_variant;
  strncpy(d.name, variant_name(v), sizeof(d.name) - 1);
  d.name[sizeof(d.name) - 1] = 0;
  return true;
}

hipError_t launch_gemm(const GemmDesc &d, const void *A, const void *B, void *C, const void *D, int64_t br,
                       hipStream_t stream) {
  if (d.m <= 0 || d.n <= 0) return hipSuccess;
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.D = D;
  a.lda = d.lda; a.ldb = d.ldb; a.ldc = d.ldc; a.stride_a = d.stride_a; a.stride_b = d.stride_b;
  a.m = (int)d.m; a.n = (int)d.n; a.k = (int)d.k; a.br = (int)(br < 0 ? 0 : br);
  a.ep = (d.beta0 ? EP_BETA0 : 0) | (d.bias ? EP_BIAS : 0) | (d.relu ? EP_RELU : 0) | (d.vnni_c ? EP_VNNI_C : 0);
  a.tiles_m = a.tiles_n = 0;
  a.vf = d.vnni_factor ? d.vnni_factor : 2;
  a.split = 0; a.scratch = nullptr; a.split_cnt = nullptr;
  a.b_trans = d.b_trans;
  int v = d.variant;
  g_last_refined.store("", std::memory_order_relaxed);
  if (d.b_trans) {
    if (d.dtype != DT_F32 || d.vnni_b) return hipErrorInvalidValue;
    return launch_grouped_t<float, false, false>(a, nullptr, 1, stream);
  }
  const bool aligned16 = ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0;
  if (v != V_GENERIC && !aligned16) v = V_GENERIC;
  // the bf16 kernel stores 16-byte row pieces and reads the bias 8 bytes at a time
  if (v >= V_BF16_FAST && v != V_BF16_SMALL32 && ((((uintptr_t)C) & 15) || (d.bias && (((uintptr_t)D) & 7)))) v = V_GENERIC;
  if (v == V_BF16_SMALL32 && ((((uintptr_t)C) & 7) || (d.bias && (((uintptr_t)D) & 7)))) v = V_GENERIC;
  // Small bf16 outputs with a LONG reduction: the 32x32 K-split kernel (fragments straight from global memory, two groups of loads
  // in flight per wave) is latency-bound there - 128 x 1024 x 4096: 15.4 us against 9.8 on 64 loader-wave tiles of 32x64 and 8.1 on
  // 128 tiles of 32x32 + K2 (the same kernel, twice the workgroups pulling panels). Crossovers (profiles/r05_bf16_skinny_small_vs_lw.txt):
  // against the 32x32 + K2 instance - usable when the output is at most one 32x32 tile per CU - at K = 1024 (256 x 1024 x 1024: 4.78
  // against 5.05 us, the reference's bs = 256 bf16 MLP as whole-layer calls 14.3 against 15.7; at K = 768 the K-split kernel still
  // wins by 0.06-0.2 us), against the 32x64 tile between 1024 and 2048. The batch count arrives with the invoke, so this choice is
  // made here and not at dispatch.
  bool bf16_lw_32x32 = false;
  if (v == V_BF16_SMALL32 && d.variant == V_BF16_SMALL32 && !d.generic_forced && d.k % 64 == 0 && d.m % 32 == 0 && d.n % 64 == 0 && !(((uintptr_t)C) & 15) &&
      !(d.bias && (((uintptr_t)D) & 7)) && !d.variant_forced) {
    const bool t32 = (d.m / 32) * (d.n / 32) <= (int64_t)g_num_cus && bf16_lw32_on();
    const int64_t thr = t32 && bf16_long_k() > 1024 ? 1024 : bf16_long_k();
    GemmDesc e = d;
    e.m = (d.m + 63) / 64 * 64; // (bf16_fast_eligible asks for m % 64; these tiles need m % 32 only)
    if ((int64_t)a.br * d.k >= thr && bf16_fast_eligible(e)) {
      v = V_BF16_LW_32x64;
      bf16_lw_32x32 = t32;
      g_last_refined.store(t32 ? "brgemm_bf16_lw<32x32,k2> (long reduction)" : "brgemm_bf16_lw<32x64,k2> (long reduction)", std::memory_order_relaxed);
    }
  }
  if (v >= V_BF16_LW_32x64 && v <= V_BF16_LW4_128x128 && a.br < 1) v = V_GENERIC; // empty batch (C = epilogue of nothing): the loader-wave kernels assume a chunk
  switch (v) {
  // LDS-DMA panels for every tile but the smallest: measured C2 +3 %, C3 +8 %, 4096^3 +3 %, 3 x 1024 MLP
  // at batch 512 / 1024 +5 % / +3 % over register staging. 32x32 tiles with 4 K-split waves have
  // 512-cycle chunks and short kernels: in a chain of dependent layers (3 x 1024 MLP, batch 256) the DMA
  // path's 2-chunk lead and slower prologue (an LDS-DMA instruction takes ~60 cycles to issue) cost more
  // than the ds_write path saves (32.4 vs 29.2 us); deeper rings (5-6 slots) made the prologue worse.
  case V_F32_64x64: return launch_fast<2, 2, 1, TPP_NACC, true>(a, stream);
  case V_F32_64x32K2: return launch_fast<2, 1, 2, TPP_NACC, true>(a, stream);
  case V_F32_32x32K4: return launch_fast<1, 1, 4, TPP_NACC, false>(a, stream);
  case V_F32_128x64: return launch_fast<4, 2, 1, TPP_NACC, true>(a, stream);
  case V_F32_64x64K2: return launch_fast<2, 2, 2, TPP_NACC, true>(a, stream);
  case V_F32_LW_64x64: return launch_f32_lw(0, a, stream);
  case V_F32_LW_64x64K2:
  case V_F32_LW_64x32K2:
  case V_F32_LW_32x32K4: {
    // skinny outputs (fewer tiles than CUs, a long batch-reduce): several workgroups per tile (choose_f32_split)
    const int tile = v == V_F32_LW_32x32K4 ? 3 : v - V_F32_LW_64x64;
    const int bm = tile == 3 ? 32 : 64, bn = tile == 1 ? 64 : 32;
    const int S = choose_f32_split(tile, (long long)(d.m / bm) * (d.n / bn), (long long)a.br * (d.k / BK));
    if (S > 1) {
      const hipError_t e = launch_f32_lw_split(tile, a, S, stream);
      if (e != hipErrorOutOfMemory && e != hipErrorInvalidValue) {
        g_last_refined.store(tile == 1 ? "brgemm_f32_lw<64x64,k2>, split" : tile == 2 ? "brgemm_f32_lw<64x32,k4>, split" : "brgemm_f32_lw<32x32,k4>, split", std::memory_order_relaxed);
        return e;
      }
      (void)hipGetLastError(); // no scratch block: the unsplit launch
    }
    return launch_f32_lw(tile, a, stream);
  }
  case V_F32_LW_128x64: return launch_f32_lw(4, a, stream);
  case V_F32_LW16_32x16: {
    if (((uintptr_t)C) & 15 || (d.bias && (((uintptr_t)D) & 15))) break; // 16-byte pieces of C and of the bias row: else the generic kernel below
    // LONG reductions (K >= 3072): every XCD streams all of A besides its share of B on the half-width tiles and their chunk time
    // rises by 40 % (128 x 1024 x 4096: 14.6 us); the 32x32 tiles with the k range shared between XCD-aligned workgroups fetch every
    // byte once (12.8 us). The batch count arrives with the invoke, so this is decided here. (profiles/r05_lw16_vs_split.txt)
    const long long chunks = (long long)a.br * (d.k / BK);
    if (chunks >= 48 && !d.variant_forced && d.n % 32 == 0) {
      const int S = choose_f32_split(3, (long long)(d.m / 32) * (d.n / 32), chunks);
      if (S > 1) {
        const hipError_t e = launch_f32_lw_split(3, a, S, stream);
        if (e != hipErrorOutOfMemory && e != hipErrorInvalidValue) {
          g_last_refined.store("brgemm_f32_lw<32x32,k4>, split (long reduction)", std::memory_order_relaxed);
          return e;
        }
        (void)hipGetLastError();
      }
    }
    return launch_f32_lw16(0, a, nullptr, 1, false, stream);
  }
  case V_BF16_FAST:
  case V_BF16_DMA128:
  case V_BF16_DMA256: return launch_gemm_bf16_fast(v - V_BF16_FAST, a, stream);
  case V_BF16_SMALL32: return launch_bf16_small32(a, nullptr, 1, stream);
  case V_BF16_LW_32x64:
  case V_BF16_LW_64x64:
  case V_BF16_LW_64x128:
  case V_BF16_LW_128x128: {
    ChainArgs c;
    c.A = a.A; c.lda = a.lda; c.cnt = nullptr; c.err = nullptr; c.target = 0;
    c.m = a.m; c.n = a.n; c.nlayers = 1; c.tiles_m = c.tiles_n = 0; c.xm = 0; c.stamps = nullptr;
    c.dbg = chain_ablation_bits();
    c.L[0] = ChainLayer{a.B, a.D, a.C, a.ldb, a.ldc, a.stride_a, a.stride_b, a.k, a.br, a.ep, 0};
    return launch_bf16_lw(bf16_lw_32x32 ? 4 : v - V_BF16_LW_32x64, c, stream);
  }
  case V_BF16_LW4_32x64:
  case V_BF16_LW4_64x64:
  case V_BF16_LW4_64x128:
  case V_BF16_LW4_128x128: {
    ChainArgs c;
    c.A = a.A; c.lda = a.lda; c.cnt = nullptr; c.err = nullptr; c.target = 0;
    c.m = a.m; c.n = a.n; c.nlayers = 1; c.tiles_m = c.tiles_n = 0; c.xm = 0; c.stamps = nullptr;
    c.dbg = chain_ablation_bits();
    c.L[0] = ChainLayer{a.B, a.D, a.C, a.ldb, a.ldc, a.stride_a, a.stride_b, a.k, a.br, a.ep, 0};
    // (round 6) skinny outputs with a long reduction: the 32x32 + K2 instance, as for VNNI-2 operands above - at most one 32x32 tile
    // per CU and K >= 1024: twice the workgroups of the 32x64 tile pulling panels (128 x 1024 x 4096: 9.3 -> 7.6 us)
    if (v == V_BF16_LW4_32x64 && !d.variant_forced && !d.generic_forced && bf16_lw32_on() && d.m % 32 == 0 && d.n % 32 == 0 &&
        (d.m / 32) * (d.n / 32) <= (int64_t)g_num_cus && (int64_t)a.br * d.k >= 1024) {
      g_last_refined.store("brgemm_bf16_lw_vnni4<32x32,k2> (long reduction)", std::memory_order_relaxed);
      return launch_bf16_lw_vnni4(4, c, stream);
    }
    return launch_bf16_lw_vnni4(v - V_BF16_LW4_32x64, c, stream);
  }
  case V_BF16_LWF_32x64:
  case V_BF16_LWF_64x64:
  case V_BF16_LWF_64x128:
  case V_BF16_LWF_128x128: {
    ChainArgs c;
    c.A = a.A; c.lda = a.lda; c.cnt = nullptr; c.err = nullptr; c.target = 0;
    c.m = a.m; c.n = a.n; c.nlayers = 1; c.tiles_m = c.tiles_n = 0; c.xm = 0; c.stamps = nullptr;
    c.dbg = chain_ablation_bits();
    c.L[0] = ChainLayer{a.B, a.D, a.C, a.ldb, a.ldc, a.stride_a, a.stride_b, a.k, a.br, a.ep, 0};
    return launch_bf16_lw_flatb(v - V_BF16_LWF_32x64, c, stream);
  }
  default: break;
  }
  // everything else: the grouped kernel with a single, inline work item
  const bool tiles_ok = aligned16 && d.n % 4 == 0 && d.k % GK == 0; // ragged m / n edges are predicated
  const bool vec = aligned16 && d.n % 4 == 0 && d.k % 4 == 0 && d.dtype == DT_F32 && !d.vnni_b && !((d.lda | d.ldb | d.stride_a | d.stride_b) & 3) &&
                   d.lda < (1 << 24) && d.ldb < (1 << 24); // (32-bit tile-relative lane offsets)
  const bool vec16x = d.dtype == DT_BF16 && d.vnni_b && !((d.lda | d.stride_a | d.stride_b) & 7) && d.lda < (1 << 21) && d.ldb < (1 << 21);
  const bool vec16_4 = vec16x && d.vnni_factor == 4 && aligned16 && d.n % 2 == 0 && d.k % GK == 0 && !(d.ldb & 1); // VNNI-4: 16-byte pieces of 2 columns
  const bool vec16 = tiles_ok && d.dtype == DT_BF16 && d.vnni_b && d.vnni_factor == 2 && !((d.lda | d.stride_a | d.stride_b) & 7) && !(d.ldb & 3) &&
                     d.lda < (1 << 21) && d.ldb < (1 << 21); // (32-bit lane offsets)
  // a SINGLE invoke of a 32-k f32 tile with an even batch count: the kernel its group would run on in the tile queue (the
  // loader-wave pair mode, tile chosen as launch_gemm_grouped does for one item) - queue on and queue off then add in the same order
  if (vec && k32_pairs_on() && !d.generic_forced && d.k == 32 && a.br >= 2 && !(a.br & 1) && d.m % 32 == 0 && d.n % 32 == 0 && d.stride_a >= 0 &&
      d.stride_b >= 0 && d.stride_a < (1 << 26) && d.stride_b < (1 << 26) && d.lda < (1 << 22) && d.ldb < (1 << 22) && d.ldc < (1 << 22)) {
    const int64_t t64 = (d.m % 64 == 0 && d.n % 64 == 0) ? (d.m / 64) * (d.n / 64) : 0;
    const int64_t t6432 = (d.m % 64 == 0) ? (d.m / 64) * (d.n / 32) : 0;
    if (t64 >= g_num_cus) return launch_f32_lw_grouped(t64 >= 2 * g_num_cus ? 0 : 1, a, nullptr, 1, 1, stream);
    if (t6432 >= g_num_cus) return launch_f32_lw_grouped(2, a, nullptr, 1, 1, stream);
    return launch_f32_lw_grouped(3, a, nullptr, 1, choose_f32_split(3, (d.m / 32) * (d.n / 32), a.br / 2), stream);
  }
  // a SINGLE invoke of a tile whose n ends inside a 32-column block (--tiles=64,48,64): the loader-wave kernel its group runs on
  if (vec && !d.generic_forced && d.k % BK == 0 && a.br >= 1 && d.m % 32 == 0 && d.n > 32 && d.n % 32 != 0 && d.lda < (1 << 22) && d.ldb < (1 << 22) &&
      d.ldc < (1 << 22))
    return launch_f32_lw_grouped(3, a, nullptr, 1, choose_f32_split(3, (d.m / 32) * ((d.n + 31) / 32), (long long)a.br * (d.k / BK)), stream);
  if (d.dtype == DT_F32) return vec ? launch_grouped_t<float, false, true>(a, nullptr, 1, stream)
                                    : launch_grouped_t<float, false, false>(a, nullptr, 1, stream);
  if (d.vnni_b && vec16_4) return launch_grouped_t<unsigned short, true, true, 4>(a, nullptr, 1, stream);
  if (d.vnni_b) return vec16 ? launch_grouped_t<unsigned short, true, true>(a, nullptr, 1, stream)
                             : launch_grouped_t<unsigned short, true, false>(a, nullptr, 1, stream);
  return launch_grouped_t<unsigned short, false, false>(a, nullptr, 1, stream);
}

} // namespace tpp

// brgemm_f32_lw.hip - f32 batch-reduce GEMM on v_mfma_f32_32x32x2_f32 with LOADER WAVES.
//
// Same arithmetic and LDS images as brgemm_f32_fast (brgemm_f32.hip: one workgroup per output tile,
// each MFMA wave one 32x32 accumulator = one k-ordered f32 fma chain per element, panels HBM -> LDS by
// LDS-DMA, A swizzled through the source address). What differs is WHO issues the DMA: measured on the
// 64x64 tile (profiles/r01_f32_c2_timeline_cycles.txt) a chunk costs 2302 cycles against 2048 of pure
// MFMA issue, and ~200 of the difference is the eight `buffer_load ... lds` instructions each MFMA wave
// issues per chunk: a wave is in-order, an LDS-DMA instruction takes 25-60 cycles to be accepted by
// the vector-memory path, and every such cycle delays the next MFMA of the SAME wave. A different wave on
// the same SIMD issues its VMEM instruction in parallel with that MFMA. So:
//   * WM*WN*WK MFMA waves: ds_read fragments + MFMA + ONE raw s_barrier per chunk. No vector memory
//     instruction in the loop at all.
//   * 2 loader waves (wave NMW streams A, wave NMW+1 streams B): all LDS-DMA of the workgroup, three
//     chunks ahead through a 4-slot ring; they wait for their own DMA (counted vmcnt) and meet the MFMA
//     waves at the per-chunk barrier, which publishes chunk t+1 and retires the slot of chunk t-1.
//   * one loop body per ring slot (slot offsets are immediates) and NO specialised tail bodies: the loop
//     is uniform, the end of the chunk stream only switches off a barrier (the first version's tail
//     variants each ran once per launch: instruction-cache cold misses in a 16-chunk kernel).
#include "gemm_common.h"
#include "xsmm_desc.h"
#include "chain_args.h"
#include "split_scratch.h"
#include <type_traits>

namespace tpp {

constexpr int LW_BK = 64;   // k per chunk
constexpr int LW_NSLOT = 4; // LDS ring slots
constexpr int LW_C_AUX = C_STORE_AUX; // write-through C stores (gemm_common.h)

typedef __attribute__((address_space(3))) void lds_void_lw;
typedef __attribute__((address_space(1))) unsigned int g_u32_lw;

// NSLOT: ring depth (4; 3 for the 128x64 tile, whose 48 KiB slots would not fit four times)
// NL loader waves for the A panel, NLB (default NL) for the B panel
// SPLIT (round 5): the batch-reduce dimension of ONE output tile is split over p.split workgroups - skinny outputs (the reference's
// M = 128 / 256 benchmark shapes: 32 .. 128 tiles on a 256-CU chip) otherwise leave most CUs idle. Workgroup (tile, s) owns the
// contiguous chunk range [T s / S, T (s + 1) / S), parks its partial tile in a scratch block (write-through stores), drains them
// and adds 1 to the tile's arrival counter; the workgroup whose add returns S - 1 (the last to arrive, whichever it is) sums the S
// partials IN SPLIT ORDER 0 .. S-1 - a fixed order of additions: results are bit-reproducible run to run, no float atomics -, adds
// C (beta = 1), bias, relu, stores, and resets the counter. No workgroup ever waits for another one (no co-residency assumption).
// GROUPED or SPLIT kernels also take n that is not a multiple of the tile width (the reference's --tiles=64,48,64 / 32,48,32):
// the B loader clamps the column pieces, the epilogue masks its loads and stores; the plain kernel (C2, C3) carries none of this.
template <int WM, int WN, int WK, bool GROUPED, int NL = 1, int NSLOT = LW_NSLOT, int NLB = NL, bool SPLIT = false>
__global__ __launch_bounds__(64 * (WM * WN * WK + NL + NLB)) void brgemm_f32_lw(GemmArgs p, const WorkItem *__restrict__ items) {
  constexpr int NMW = WM * WN * WK; // MFMA waves
  constexpr int BM = 32 * WM, BN = 32 * WN;
  constexpr int A_STAGE = BM * LW_BK, B_STAGE = LW_BK * BN, SLOT = A_STAGE + B_STAGE; // floats
  constexpr int NA = BM / 4;       // DMA instructions (1 KiB each) per chunk of A: 4 rows x 64 k
  constexpr int RPI = 256 / BN;    // B rows per DMA instruction
  constexpr int NB = LW_BK / RPI;  // DMA instructions per chunk of B
  constexpr int KB_PER_WAVE = 8 / WK, KB_HALF = KB_PER_WAVE / 2;
  static_assert(KB_HALF >= 1 && NA / NL <= 31 && NB / NLB <= 31, "tile outside the schedule's limits (vmcnt is 6 bits)");
  static_assert(NSLOT == 3 || NSLOT == 4, "ring depth");
  extern __shared__ __attribute__((aligned(16))) float smem_lw[];

  const int tid = threadIdx.x, lane = tid & 63;
#ifndef TPP_LW_LOADERS_FIRST
#define TPP_LW_LOADERS_FIRST 1
#endif
  // the two loader waves are the FIRST two hardware waves of the workgroup (waves start in order: the panels' first
  // chunks are requested before the MFMA waves have been launched); `wave` is the role index: MFMA waves 0 .. NMW-1, loaders NMW, NMW+1
  const int hw_wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = TPP_LW_LOADERS_FIRST ? (hw_wave < NL + NLB ? NMW + hw_wave : hw_wave - (NL + NLB)) : hw_wave;
  // XCD-blocked (8, bn, bm) or plain (1, tiles_n, tiles_m) grid: see brgemm_f32.hip. GROUPED (tile queue): grid (items,
  // tiles_n, tiles_m), workgroup = one tile of one queued invoke, operands and batch count from its item. A template
  // parameter, not a run-time test: the plain kernel is the headline kernel and must not carry a second mode (measured: 0.6 %).
  static_assert(!SPLIT || WK > 1, "the split epilogue is the K-split tiles' float4 epilogue");
  constexpr bool RAGN = GROUPED || SPLIT; // n may end inside the tile's last 32-column block (a multiple of 4)
  WorkItem it{p.A, p.B, p.C, p.D, (int64_t)p.br};
  int tm, tn, sp = 0, tile_id = 0;
  if constexpr (SPLIT && GROUPED) {
    // grid (items * S, tiles_n, tiles_m): x = item * S + s (the two block rows of a layer that share a B panel then meet on one XCD)
    const int item = (int)blockIdx.x / p.split;
    sp = (int)blockIdx.x - item * p.split;
    if (items) it = items[item];
    tm = (int)blockIdx.z, tn = (int)blockIdx.y;
    tile_id = (item * (int)gridDim.y + tn) * (int)gridDim.z + tm;
  } else if constexpr (SPLIT) {
    // linear grid of S * tiles units in [s][tn][tm] order; XCD x (= workgroup id mod 8) takes the x-th eighth of the list, so that
    // one XCD's L2 sees ONE k range of A and B (every byte of the operands then comes out of HBM about once)
    const int tiles = p.tiles_m * p.tiles_n, total = tiles * p.split;
    int u = (int)blockIdx.x;
    if ((total & 7) == 0) u = (u & 7) * (total >> 3) + (u >> 3);
    sp = u / tiles;
    tile_id = u - sp * tiles;
    tn = tile_id / p.tiles_m;
    tm = tile_id - tn * p.tiles_m;
  } else {
    if constexpr (GROUPED) {
      if (items) it = items[blockIdx.x]; // (no list: a single invoke of the grouped kernel, operands in the arguments)
    }
    tm = GROUPED ? (int)blockIdx.z : (int)(blockIdx.x >> p.xn_shift) * p.tiles_m + (int)blockIdx.z;
    tn = GROUPED ? (int)blockIdx.y : (int)(blockIdx.x & ((1u << p.xn_shift) - 1)) * p.tiles_n + (int)blockIdx.y;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int nvalid = RAGN ? p.n - n0 : BN; // columns of this tile that exist (>= BN everywhere but in a ragged last tile)
  const float *__restrict__ A = (const float *)it.A;
  const float *__restrict__ B = (const float *)it.B;
  float *__restrict__ C = (float *)it.C;
  // GROUPED, k = 32 (the compiler's 32^3 tiles, mlir-gen --tiles=32,32,32): a 64-k chunk is the 32-k blocks of TWO consecutive
  // batch elements (even batch counts only: launch_gemm_grouped checks) - k 0..31 of the chunk image from element 2t, k 32..63
  // from element 2t+1. Only the loaders' source offsets know; the LDS images and the MFMA waves are the same.
  const bool pair = GROUPED && p.k == 32;
  const int kchunks = pair ? 1 : p.k / LW_BK;
  const int Tall = pair ? (int)it.br / 2 : (int)it.br * kchunks;
  // SPLIT: this workgroup's chunks [t_first, t_first + T) of the tile's Tall
  const int t_first = SPLIT ? (int)(((long long)Tall * sp) / p.split) : 0;
  const int T = SPLIT ? (int)(((long long)Tall * (sp + 1)) / p.split) - t_first : Tall;

  if (wave >= NMW) {
    // ---- loader waves --------------------------------------------------------------------
    const bool isA = (wave - NMW) < NL; // NL loader waves per panel: wave `part` issues the instructions part, part + NL, ...
    const int part = isA ? wave - NMW : wave - NMW - NL;
    static_assert(NA % NL == 0 && NB % NLB == 0, "panel instructions divide over the loader waves");
    // per-lane source offsets, constant for the whole kernel. A instruction v covers rows 4v .. 4v+3
    // (lane -> row 4v + lane/16, 16-byte piece lane%16, XOR-ed with row&15 = 4(v&3) + lane/16: the
    // fragment read applies the same XOR); the 16-row group v>>2 goes into the scalar offset.
    unsigned voA[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * j + (lane >> 4), pc = (lane & 15) ^ r; // 16-byte piece of the chunk's row: k 4 pc .. 4 pc + 3
      voA[j] = (unsigned)((r * (int)p.lda + (pair ? (pc >> 3) * (int)p.stride_a + 4 * (pc & 7) : 4 * pc)) * 4);
    }
    unsigned voA2[2]; // NL = 2: this wave's instructions have v & 3 = part and part + 2 (NL = 4: always part)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 4 * (part + 2 * j) + (lane >> 4), pc = (lane & 15) ^ r;
      voA2[j] = (unsigned)((r * (int)p.lda + (pair ? (pc >> 3) * (int)p.stride_a + 4 * (pc & 7) : 4 * pc)) * 4);
    }
    // (a ragged last tile: the 16-byte column pieces beyond n re-read the tile's last valid piece - in bounds, and the columns
    // they feed are never stored)
    const int pieceB = RAGN && nvalid < BN ? (lane % (BN / 4) < nvalid / 4 ? lane % (BN / 4) : nvalid / 4 - 1) : lane % (BN / 4);
    const unsigned voB = (unsigned)(((lane / (BN / 4)) * (int)p.ldb + 4 * pieceB) * 4);
    const unsigned stepA = (unsigned)(16 * (int)p.lda * 4), stepB = (unsigned)(RPI * (int)p.ldb * 4);
    const float *g = isA ? A + (int64_t)m0 * p.lda : B + n0; // panel base of the chunk being fetched
    int kc = 0;
    const int64_t d_in = isA ? (int64_t)LW_BK : (int64_t)LW_BK * p.ldb;
    const int64_t d_wrap = (isA ? p.stride_a : p.stride_b) * (pair ? 2 : 1) - (int64_t)(kchunks - 1) * d_in;
    if constexpr (SPLIT) { // start at chunk t_first: batch element t_first / kchunks (pair mode: the pair t_first), k block t_first % kchunks
      const int b0 = t_first / kchunks;
      kc = t_first - b0 * kchunks;
      g += (int64_t)b0 * (isA ? p.stride_a : p.stride_b) * (pair ? 2 : 1) + (int64_t)kc * d_in;
    }
    const unsigned pairB = pair ? (unsigned)(((int)p.stride_b - 32 * (int)p.ldb) * 4) : 0u; // rows 32.. of a pair chunk: the second element
    auto issue = [&](int slot) __attribute__((always_inline)) {
      float *base = smem_lw + slot * SLOT + (isA ? 0 : A_STAGE);
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, 0x7fffffff, 0x00020000);
      if (isA) {
#pragma unroll
        for (int i = 0; i < NA / NL; ++i) {
          const int v = part + NL * i; // (NL = 2: v & 3 is part or part + 2 - both live in voA)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_lw *)(base + v * 256), 16, NL == 1 ? voA[i & 3] : NL == 2 ? voA2[i & 1] : voA2[0],
                                                   (unsigned)(v >> 2) * stepA, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NB / NLB; ++i) {
          const int v = part + NLB * i;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_lw *)(base + v * 256), 16, voB,
                                                   (unsigned)v * stepB + (GROUPED && v * RPI >= 32 ? pairB : 0u), 0, 0);
        }
      }
      if (++kc == kchunks) {
        kc = 0;
        g += d_wrap;
      } else {
        g += d_in;
      }
    };
    // s_waitcnt vmcnt(n chunks of this wave's DMA may still be in flight)
    auto wait_left = [&](int chunks) __attribute__((always_inline)) {
      if (chunks == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (isA) {
        if (chunks == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA / NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NA / NL) : "memory");
      } else {
        if (chunks == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB / NLB) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB / NLB) : "memory");
      }
    };
    // Prologue: chunks 0 and 1 are requested, chunk 0 is PUBLISHED as soon as it has landed, chunk 2 follows behind the barrier.
    // (Round 2 requested all three first: an LDS-DMA instruction takes ~25 ns to issue, a chunk is 16 of them per loader - the MFMA
    // waves waited ~0.4 us for requests they would not need for two chunk times; a chunk is 0.92 us of MFMA here, so the loader has
    // all the time it needs behind the barrier. Stamped on the bf16 twin of this kernel: profiles/r03_chain_anatomy.txt.)
    if (T > 0) issue(0);
    if (T > 1) issue(1);
    wait_left(T > 1 ? 1 : 0);
    __builtin_amdgcn_s_barrier(); // chunk 0 published
    if (NSLOT > 3 && T > 2) issue(2); // (a 3-slot ring holds chunks t, t+1, t+2: nothing more before chunk 0 has been retired)
    for (int t = 0; t + 1 < T; ++t) {
      wait_left(NSLOT > 3 && t + 2 < T ? 1 : 0); // chunk t+1 has landed (4 slots: chunk t+2 may still fly)
      __builtin_amdgcn_s_barrier();               // = the MFMA waves' mid-chunk barrier of chunk t
      if (t + NSLOT - 1 < T) issue((t + NSLOT - 1) % NSLOT); // the slot of chunk t-1: every MFMA wave is past it
    }
    return; // ended waves do not take part in later barriers
  }

  // ---- MFMA waves --------------------------------------------------------------------------
  const int wk = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn / WN, wn = wmn % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int ccol = n0 + wn * 32 + li;
  const __amdgpu_buffer_rsrc_t rsrcC =
      __builtin_amdgcn_make_buffer_rsrc((void *)(C + (int64_t)m0 * p.ldc + n0), 0, 0x7fffffff, 0x00020000);
  const unsigned voffC = (unsigned)(((wm * 32 + 4 * lh) * (int)p.ldc + wn * 32 + li) * 4);
  const unsigned ldcb = (unsigned)((int)p.ldc * 4);
  // the accumulator chain of K group 0 starts from C (beta = 1), as in the reference; the bias is fetched
  // here so that its latency is not exposed in the epilogue
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float bias = 0.0f;
  if constexpr (WK == 1) { // (the K-split tiles' epilogue works on float4 pieces: bias4 below)
    if ((p.ep & EP_BIAS) && (!RAGN || ccol < p.n)) bias = ((const float *)it.D)[ccol];
  }
  // ... and the K-split tiles' float4 epilogue: its four bias values per lane (16-byte column piece lane & 7). With ONE MFMA wave
  // per SIMD (32x32 + K4) they come in here as well - behind the combine's second barrier the load was an exposed round trip:
  // the batch-256 layer of the reference's MLP 7.05 -> 6.82 us per launch, its tile-queue iteration 19.47 -> 19.25 us (same box,
  // profiles/r04_c3_loaders_and_launch_knobs.txt (4)); with two MFMA waves per SIMD the other wave hides it and the early fetch
  // measured 0.02 us SLOWER on C3, so those tiles keep the load in the epilogue
  constexpr bool BIAS_EARLY = WK > 1 && WM * WN * WK <= 4;
  f32x4 bias4 = {0.0f, 0.0f, 0.0f, 0.0f};
  const bool piece_ok = !RAGN || wn * 32 + 4 * (lane & 7) < nvalid; // this lane's 16-byte column piece of the float4 epilogue exists
  if constexpr (BIAS_EARLY) {
    if ((p.ep & EP_BIAS) && piece_ok) bias4 = *(const f32x4 *)((const float *)it.D + n0 + wn * 32 + 4 * (lane & 7));
  }
  if (wk == 0 && !SPLIT) { // (SPLIT: C joins the ordered sum of the partials in the last workgroup's epilogue)
    if (!(p.ep & EP_BETA0) && (!RAGN || ccol < p.n)) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc[r] = __builtin_bit_cast(
            float, __builtin_amdgcn_raw_buffer_load_b32(rsrcC, voffC, (unsigned)((r & 3) + 8 * (r >> 2)) * ldcb, 0));
    }
  }

  // MFMA fragments of one k-block (8 k): 4 A values (one ds_read_b128) and 4 B values per lane;
  // double-buffered so block q+1 is read while block q multiplies (brgemm_f32.hip has the layout notes)
  f32x4 fa[2];
  float fb[2][4];
  const int a_off = (wm * 32 + li) * LW_BK, b_off = wn * 32 + li;
  auto frag_load = [&](int buf, int slot, int kb) __attribute__((always_inline)) {
    const float *as = smem_lw + slot * SLOT + a_off;
    const float *bs = smem_lw + slot * SLOT + A_STAGE + b_off;
    fa[buf] = *(const f32x4 *)(as + (((2 * kb + lh) ^ (li & 15)) << 2));
#pragma unroll
    for (int s = 0; s < 4; ++s) fb[buf][s] = bs[(8 * kb + 4 * lh + s) * BN];
  };
  const int kbw = wk * KB_PER_WAVE;
  // hn_c: does another chunk follow (= does this chunk carry the barrier)? 1: yes, a compile-time fact - the steady-state lap below
  // is then ONE basic block (a conditional barrier or an exit test between two chunks is a block boundary: a branch, and a point
  // where the compiler waits for every LDS read in flight); 2: decided at run time (the last lap)
  auto chunk = [&](auto slot_c, auto hn_c, bool has_next_rt) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value, NS = (S + 1) % NSLOT;
    const bool has_next = decltype(hn_c)::value == 1 ? true : has_next_rt;
#pragma unroll
    for (int q = 0; q < KB_PER_WAVE; ++q) {
      const int cur = q & 1, nxt = cur ^ 1;
      if (q + 1 < KB_PER_WAVE) frag_load(nxt, S, kbw + q + 1);
      else frag_load(nxt, NS, kbw); // first block of chunk t+1 (published by this chunk's barrier; unused after the last chunk)
      __builtin_amdgcn_sched_barrier(0); // the reads of step q+1 stay above the MFMAs of step q
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], fb[cur][s], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (q == KB_HALF - 1 && has_next) {
        __builtin_amdgcn_s_barrier(); // chunk t+1 published by the loaders; the slot of chunk t-1 retired
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the prefetched fragments of chunk t+1 are dead on the loop's exit path: without this the compiler sinks
    // their reads into the next chunk's head, where the first MFMA then waits for them (the wait this
    // implies sits behind the last step's four MFMAs: the reads are long back)
    // (not inside the steady-state laps: there the next chunk follows in the same basic block, the sched_barriers keep the order,
    // and the pin would only make the compiler wait for the prefetched fragments at the end of every chunk)
    constexpr int PF = KB_PER_WAVE & 1;
    if constexpr (decltype(hn_c)::value != 1)
      asm volatile("" : "+v"(fa[PF]), "+v"(fb[PF][0]), "+v"(fb[PF][1]), "+v"(fb[PF][2]), "+v"(fb[PF][3]));
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;

  __builtin_amdgcn_s_barrier(); // chunk 0 published
  __builtin_amdgcn_sched_barrier(0);
  if (T > 0) {
    frag_load(0, 0, kbw);
    using HY = std::integral_constant<int, 1>;
    using HR = std::integral_constant<int, 2>;
    int t = 0;
    for (; t + NSLOT < T; t += NSLOT) { // whole laps of the ring that are followed by at least one more chunk
      chunk(S0{}, HY{}, true);
      chunk(S1{}, HY{}, true);
      chunk(S2{}, HY{}, true);
      if constexpr (NSLOT > 3) chunk(S3{}, HY{}, true);
    }
    for (;;) { // the last lap: 1 .. NSLOT chunks
      chunk(S0{}, HR{}, t + 1 < T);
      if (++t == T) break;
      chunk(S1{}, HR{}, t + 1 < T);
      if (++t == T) break;
      chunk(S2{}, HR{}, t + 1 < T);
      if (++t == T) break;
      if constexpr (NSLOT > 3) {
        chunk(S3{}, HR{}, t + 1 < T);
        if (++t == T) break;
      }
    }
  }

  if constexpr (WK > 1) {
    // combine the K groups through LDS: EVERY group parks its 32x32 partial, then group g finishes the accumulator registers
    // [g * 16 / WK, (g + 1) * 16 / WK) of its tile - sum in group order (group 0 carries C when beta = 1), bias, relu, store.
    // (With group 0 finishing alone the other groups' waves idled through 16 LDS reads + 16 stores per lane.)
    __syncthreads();
    float *red = smem_lw; // WK * WM*WN * 1024 floats, fits in the ring
    {
      float *dst = red + (wk * (WM * WN) + wmn) * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64] = acc[r];
    }
    __syncthreads();
    // The parked partials are [register r][lane = column li + 32 * lh]: four consecutive columns of one output row (r, lh) are
    // 16 contiguous bytes. A lane finishes float4 pieces - 8 lanes x 16 B = one 128-byte row of the tile, a wave instruction 8
    // rows - so a tile is 4 x 16-byte stores per lane split over the K groups, instead of 16 dword stores (round 2).
    constexpr int IPG = 4 / WK; // store instructions per lane per group
    const int c4 = lane & 7, rsel = lane >> 3; // 16-byte column piece, row within the instruction's 8 rows
    if constexpr (!BIAS_EARLY) {
      if ((p.ep & EP_BIAS) && piece_ok) bias4 = *(const f32x4 *)((const float *)it.D + n0 + wn * 32 + 4 * c4);
    }
    f32x4 part[IPG];
#pragma unroll
    for (int j = 0; j < IPG; ++j) {
      const int q = 8 * (wk * IPG + j) + rsel;  // row of the 32x32 tile: q = (r & 3) + 4 * lh + 8 * (r >> 2)
      const int r = (q & 3) + 4 * (q >> 3), lh2 = (q >> 2) & 1;
      const float *src = red + wmn * 1024 + r * 64 + lh2 * 32 + 4 * c4;
      f32x4 v = *(const f32x4 *)src;
#pragma unroll
      for (int g = 1; g < WK; ++g) v += *(const f32x4 *)(src + g * (WM * WN) * 1024);
      part[j] = v;
    }
    if constexpr (SPLIT) {
      // park the partial tile: block [tile][split][piece], piece = (j * NMW + MFMA wave) * 64 + lane - every wave instruction writes
      // 1 KiB contiguous; the last workgroup reads the S blocks with the same lane mapping
      constexpr int TILE = BM * BN; // floats per partial
      const int S = p.split;
      float *scr = p.scratch + (size_t)tile_id * S * TILE;
      const __amdgpu_buffer_rsrc_t rsrcS = __builtin_amdgcn_make_buffer_rsrc((void *)scr, 0, 0x7fffffff, 0x00020000);
      const unsigned pvo = (unsigned)((wave * 64 + lane) * 16);
#pragma unroll
      for (int j = 0; j < IPG; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, part[j]), rsrcS, pvo, (unsigned)((sp * TILE + j * NMW * 256) * 4), 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the write-through stores of this wave have been acknowledged
      __syncthreads();                                  // ... and those of every MFMA wave (red is free again)
      unsigned *flag = (unsigned *)smem_lw;
      if (wave == 0 && lane == 0)
        *flag = __hip_atomic_fetch_add((g_u32_lw *)(p.split_cnt + tile_id), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (*flag != (unsigned)(S - 1)) return; // not the last to arrive: done
      if (wave == 0 && lane == 0) __hip_atomic_store((g_u32_lw *)(p.split_cnt + tile_id), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // for the next launch
      // the ordered sum: partial 0 + partial 1 + ... (sc1 loads: the blocks were written by workgroups behind other L2s)
#pragma unroll
      for (int j = 0; j < IPG; ++j) {
        f32x4 acc4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcS, pvo, (unsigned)((j * NMW * 256) * 4), 16));
        int s2 = 1;
        for (; s2 + 4 <= S; s2 += 4) { // four loads in flight, added in order
          f32x4 t[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            t[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcS, pvo, (unsigned)(((s2 + e) * TILE + j * NMW * 256) * 4), 16));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc4 += t[e];
        }
        for (; s2 < S; ++s2)
          acc4 += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcS, pvo, (unsigned)((s2 * TILE + j * NMW * 256) * 4), 16));
        part[j] = acc4;
      }
    }
#pragma unroll
    for (int j = 0; j < IPG; ++j) {
      const int q = 8 * (wk * IPG + j) + rsel;
      f32x4 v = part[j];
      const unsigned co = (unsigned)(((wm * 32 + q) * (int)p.ldc + wn * 32 + 4 * c4) * 4);
      if constexpr (SPLIT) {
        if (!(p.ep & EP_BETA0) && piece_ok) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcC, co, 0, 0));
      }
      v += bias4;
      if (p.ep & EP_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
      }
      if (piece_ok) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrcC, co, 0, LW_C_AUX);
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = acc[r] + bias;
    if (p.ep & EP_RELU) v = v > 0.0f ? v : 0.0f;
    if (!RAGN || ccol < p.n)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrcC, voffC,
                                            (unsigned)((r & 3) + 8 * (r >> 2)) * ldcb, LW_C_AUX);
  }
}

template <int WM, int WN, int WK, int NL = 1, int NSLOT = LW_NSLOT, int NLB = NL> static hipError_t launch_lw_t(const GemmArgs &a, hipStream_t s) {
  constexpr int BM = 32 * WM, BN = 32 * WN, NT = 64 * (WM * WN * WK + NL + NLB);
  constexpr size_t lds = (size_t)NSLOT * (BM * LW_BK + LW_BK * BN) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static std::atomic<unsigned long long> lds_set{0};
  if (hipError_t e = ensure_dynamic_lds((const void *)brgemm_f32_lw<WM, WN, WK, false, NL, NSLOT, NLB>, (int)lds, lds_set); e != hipSuccess) return e;
  GemmArgs args = a;
  const int tiles_m = a.m / BM, tiles_n = a.n / BN;
  dim3 grid;
  // XCD-blocked grid: xm x xn = 8 XCD blocks of tiles_m/xm x tiles_n/xn tiles (blockIdx.x = the XCD: workgroups go to XCDs round
  // robin). Each XCD's L2 then fetches m/xm rows of A and n/xn columns of B: the split that minimises m/xm + n/xn - 4 x 2 for
  // square outputs (C2; ties keep it: rounds 1-3 had only this one), 2 x 4 for C3's 512 x 1024 and the batch-256 layers (-20 / -33 % of
  // the L2 fill). TPP_HIP_F32_LW_XM forces xm (A/B runs).
  static const int forced_xm = [] {
    const char *e = getenv("TPP_HIP_F32_LW_XM");
    return e ? atoi(e) : 0;
  }();
  int xm = 0;
  long long best = -1;
  for (int c : {4, 2, 8, 1}) {
    const int xn = 8 / c;
    if (tiles_m % c || tiles_n % xn || tiles_m / c > 65535 || tiles_n / xn > 65535) continue;
    const long long cost = (long long)a.m / c + (long long)a.n / xn;
    if (forced_xm ? c == forced_xm : (best < 0 || cost < best)) {
      best = cost;
      xm = c;
    }
  }
  args.xn_shift = 0;
  if (xm) {
    const int xn = 8 / xm;
    args.tiles_m = tiles_m / xm;
    args.tiles_n = tiles_n / xn;
    args.xn_shift = xn == 8 ? 3 : xn == 4 ? 2 : xn == 2 ? 1 : 0;
    grid = dim3(8, args.tiles_n, args.tiles_m);
  } else {
    args.tiles_m = args.tiles_n = 0;
    if (tiles_m > 65535 || tiles_n > 65535) return hipErrorInvalidValue;
    grid = dim3(1, tiles_n, tiles_m);
  }
  hipLaunchKernelGGL((brgemm_f32_lw<WM, WN, WK, false, NL, NSLOT, NLB>), grid, dim3(NT), lds, s, args, (const WorkItem *)nullptr);
  return hipGetLastError();
}

// grouped launch (tile queue): one workgroup per (item, tile of the item); m, n multiples of the tile, k of 64 - or k = 32 with
// even batch counts and 0 <= stride < 2^26 elements (the kernel's pair mode)
template <int WM, int WN, int WK>
static hipError_t launch_lw_grouped_t(const GemmArgs &a, const WorkItem *items, int n_items, hipStream_t s) {
  constexpr int BM = 32 * WM, BN = 32 * WN, NT = 64 * (WM * WN * WK + 2);
  constexpr size_t lds = (size_t)LW_NSLOT * (BM * LW_BK + LW_BK * BN) * sizeof(float);
  static std::atomic<unsigned long long> lds_set{0};
  if (hipError_t e = ensure_dynamic_lds((const void *)brgemm_f32_lw<WM, WN, WK, true>, (int)lds, lds_set); e != hipSuccess) return e;
  GemmArgs args = a;
  args.tiles_m = args.tiles_n = 0;
  args.xn_shift = 0;
  hipLaunchKernelGGL((brgemm_f32_lw<WM, WN, WK, true>), dim3((unsigned)n_items, (a.n + BN - 1) / BN, a.m / BM), dim3(NT), lds, s, args, items);
  return hipGetLastError();
}

// ---- SPLIT launches: S workgroups per output tile (kernel comment; scratch: split_scratch.h) ----------------------------------
// whole-layer call: linear grid of S * tiles workgroups
template <int WM, int WN, int WK> static hipError_t launch_lw_split_t(const GemmArgs &a, int S, hipStream_t s) {
  constexpr int BM = 32 * WM, BN = 32 * WN, NT = 64 * (WM * WN * WK + 2);
  constexpr size_t lds = (size_t)LW_NSLOT * (BM * LW_BK + LW_BK * BN) * sizeof(float);
  const long long tiles = (long long)(a.m / BM) * ((a.n + BN - 1) / BN);
  const SplitScratch *sc = split_scratch_for(s, tiles, tiles * S * BM * BN);
  if (!sc) return hipErrorOutOfMemory; // the caller falls back to the unsplit launch
  static std::atomic<unsigned long long> lds_set{0};
  if (hipError_t e = ensure_dynamic_lds((const void *)brgemm_f32_lw<WM, WN, WK, false, 1, LW_NSLOT, 1, true>, (int)lds, lds_set); e != hipSuccess) return e;
  GemmArgs args = a;
  args.tiles_m = a.m / BM;
  args.tiles_n = (a.n + BN - 1) / BN;
  args.xn_shift = 0;
  args.split = S;
  args.scratch = sc->partial;
  args.split_cnt = sc->cnt;
  hipLaunchKernelGGL((brgemm_f32_lw<WM, WN, WK, false, 1, LW_NSLOT, 1, true>), dim3((unsigned)(tiles * S)), dim3(NT), lds, s, args, (const WorkItem *)nullptr);
  return hipGetLastError();
}
// tile queue group: grid (items * S, tiles_n, tiles_m)
template <int WM, int WN, int WK>
static hipError_t launch_lw_grouped_split_t(const GemmArgs &a, const WorkItem *items, int n_items, int S, hipStream_t s) {
  constexpr int BM = 32 * WM, BN = 32 * WN, NT = 64 * (WM * WN * WK + 2);
  constexpr size_t lds = (size_t)LW_NSLOT * (BM * LW_BK + LW_BK * BN) * sizeof(float);
  const int tiles_m = a.m / BM, tiles_n = (a.n + BN - 1) / BN;
  const long long tiles = (long long)n_items * tiles_m * tiles_n;
  const SplitScratch *sc = split_scratch_for(s, tiles, tiles * S * BM * BN);
  if (!sc) return hipErrorOutOfMemory;
  static std::atomic<unsigned long long> lds_set{0};
  if (hipError_t e = ensure_dynamic_lds((const void *)brgemm_f32_lw<WM, WN, WK, true, 1, LW_NSLOT, 1, true>, (int)lds, lds_set); e != hipSuccess) return e;
  GemmArgs args = a;
  args.tiles_m = args.tiles_n = 0;
  args.xn_shift = 0;
  args.split = S;
  args.scratch = sc->partial;
  args.split_cnt = sc->cnt;
  hipLaunchKernelGGL((brgemm_f32_lw<WM, WN, WK, true, 1, LW_NSLOT, 1, true>), dim3((unsigned)(n_items * S), tiles_n, tiles_m), dim3(NT), lds, s, args, items);
  return hipGetLastError();
}

// tile as in launch_f32_lw; split > 1: that many workgroups per output tile (K-split tiles 1 .. 3 only); n may end inside the last tile
hipError_t launch_f32_lw_grouped(int tile, const GemmArgs &a, const WorkItem *items, int n_items, int split, hipStream_t s) {
  if (split > 1) {
    hipError_t e = hipErrorInvalidValue;
    switch (tile) {
    case 1: e = launch_lw_grouped_split_t<2, 2, 2>(a, items, n_items, split, s); break;
    case 2: e = launch_lw_grouped_split_t<2, 1, 4>(a, items, n_items, split, s); break;
    case 3: e = launch_lw_grouped_split_t<1, 1, 4>(a, items, n_items, split, s); break;
    default: break;
    }
    if (e != hipErrorOutOfMemory && e != hipErrorInvalidValue) return e;
    (void)hipGetLastError(); // no scratch block (or a tile without a split instance): the unsplit launch
  }
  switch (tile) {
  case 0: return launch_lw_grouped_t<2, 2, 1>(a, items, n_items, s);
  case 1: return launch_lw_grouped_t<2, 2, 2>(a, items, n_items, s);
  case 2: return launch_lw_grouped_t<2, 1, 4>(a, items, n_items, s);
  case 3: return launch_lw_grouped_t<1, 1, 4>(a, items, n_items, s);
  default: return hipErrorInvalidValue;
  }
}
// whole-layer call on S workgroups per tile; hipErrorOutOfMemory / hipErrorInvalidValue: not launched, use launch_f32_lw
hipError_t launch_f32_lw_split(int tile, const GemmArgs &a, int split, hipStream_t s) {
  switch (tile) {
  case 1: return launch_lw_split_t<2, 2, 2>(a, split, s);
  case 2: return launch_lw_split_t<2, 1, 4>(a, split, s);
  case 3: return launch_lw_split_t<1, 1, 4>(a, split, s);
  default: return hipErrorInvalidValue;
  }
}

// ---- a CHAIN of whole-layer f32 BRGEMMs in one launch ------------------------------------------------------------------------
// The reference's MLP benchmark (mlir-gen --batch=256 --layers=1024,1024,1024,1024, fp32, benchmarks/config/base/base.json:74-80)
// lowers to one whole-layer xsmm_fused_brgemm_invoke per layer; a layer of 256 x 1024 x 1024 is 3.4 us of MFMA work behind a
// 2.5 us launch. xsmm_hip_fused_brgemm_chain_invoke (runtime.cpp) runs the layers of such a step as ONE launch of this kernel: the
// same tile, the same loader-wave structure, the same order of additions as brgemm_f32_lw<WM, WN, WK> (results bit-identical to
// the separate launches), one workgroup per output tile, all co-resident (tiles <= CUs), and between two layers the hand-off of
// the bf16 chain (brgemm_bf16_lw.hip): a tile is stored with 16-byte write-through stores, every storing wave drains them, the
// workgroup adds 1 to the arrival counter of its ROW BLOCK; the A loaders of the next layer wait until all tiles_n tiles of their
// row block have arrived and fetch the rows with sc1 loads (another XCD's L2 may hold them). Counters only grow (target = epoch x
// tiles_n, runtime.cpp); every spin is bounded (CHAIN_TIMEOUT_TICKS) and reports through p.err instead of hanging the GPU.
// The ring restarts at slot 0 with every layer (the K-group combine parks its partials in slot 0): the B loader of layer l+1
// requests its first chunks right behind the seam barrier, while the A loaders still poll.
//   barriers per layer, every wave: P (chunk 0 published), T - 1 mid-chunk barriers, R1 + R2 (combine), S1 (tile stored and
//   drained; not after the last layer)
typedef __attribute__((address_space(1))) unsigned int g_u32_f32c;

template <int WM, int WN, int WK, int NL>
__global__ __launch_bounds__(64 * (WM * WN * WK + 2 * NL)) void brgemm_f32_lw_chain(ChainArgs p) {
  static_assert(WK > 1, "the hand-off needs the 16-byte stores of the K-split tiles' epilogue");
  constexpr int NSLOT = LW_NSLOT;
  constexpr int NMW = WM * WN * WK;
  constexpr int BM = 32 * WM, BN = 32 * WN;
  constexpr int A_STAGE = BM * LW_BK, B_STAGE = LW_BK * BN, SLOT = A_STAGE + B_STAGE; // floats
  constexpr int NA = BM / 4, RPI = 256 / BN, NB = LW_BK / RPI;
  constexpr int KB_PER_WAVE = 8 / WK, KB_HALF = KB_PER_WAVE / 2;
  static_assert(KB_HALF >= 1 && NA / NL <= 31 && NB / NL <= 31 && NA % NL == 0 && NB % NL == 0, "tile outside the schedule's limits");
  extern __shared__ __attribute__((aligned(16))) float smem_lw[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int hw_wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = hw_wave < 2 * NL ? NMW + hw_wave : hw_wave - 2 * NL; // loaders first, as in brgemm_f32_lw
  const int tm = (int)blockIdx.x / p.tiles_n, tn = (int)blockIdx.x % p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int L = p.nlayers;

  if (wave >= NMW) {
    // ---- loader waves ------------------------------------------------------------------------------------------
    const bool isA = (wave - NMW) < NL;
    const int part = (wave - NMW) % NL;
    for (int l = 0; l < L; ++l) {
      const ChainLayer &Y = p.L[l];
      const int lda = (int)(l == 0 ? p.lda : p.L[l > 0 ? l - 1 : 0].ldc), ldb = (int)Y.ldb;
      const float *Asrc = (const float *)(l == 0 ? p.A : p.L[l > 0 ? l - 1 : 0].C);
      const int kchunks = Y.k / LW_BK;
      const int T = Y.br * kchunks;
      unsigned voA[4], voA2[2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * j + (lane >> 4);
        voA[j] = (unsigned)((r * lda + 4 * ((lane & 15) ^ r)) * 4);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = 4 * (part + 2 * j) + (lane >> 4);
        voA2[j] = (unsigned)((r * lda + 4 * ((lane & 15) ^ r)) * 4);
      }
      const unsigned voB = (unsigned)(((lane / (BN / 4)) * ldb + 4 * (lane % (BN / 4))) * 4);
      const unsigned stepA = (unsigned)(16 * lda * 4), stepB = (unsigned)(RPI * ldb * 4);
      const float *g = isA ? Asrc + (int64_t)m0 * lda : (const float *)Y.B + n0;
      int kc = 0;
      const int64_t d_in = isA ? (int64_t)LW_BK : (int64_t)LW_BK * ldb;
      const int64_t d_wrap = (isA ? Y.stride_a : Y.stride_b) - (int64_t)(kchunks - 1) * d_in;
      const bool sc1 = isA && l > 0; // rows written by other workgroups of THIS launch
      auto issue = [&](int slot) __attribute__((always_inline)) {
        float *base = smem_lw + slot * SLOT + (isA ? 0 : A_STAGE);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, 0x7fffffff, 0x00020000);
        if (isA && sc1) {
#pragma unroll
          for (int i = 0; i < NA / NL; ++i) {
            const int v = part + NL * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_lw *)(base + v * 256), 16, NL == 1 ? voA[i & 3] : NL == 2 ? voA2[i & 1] : voA2[0],
                                                     (unsigned)(v >> 2) * stepA, 0, 16);
          }
        } else if (isA) {
#pragma unroll
          for (int i = 0; i < NA / NL; ++i) {
            const int v = part + NL * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_lw *)(base + v * 256), 16, NL == 1 ? voA[i & 3] : NL == 2 ? voA2[i & 1] : voA2[0],
                                                     (unsigned)(v >> 2) * stepA, 0, 0);
          }
        } else {
#pragma unroll
          for (int i = 0; i < NB / NL; ++i) {
            const int v = part + NL * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_lw *)(base + v * 256), 16, voB, (unsigned)v * stepB, 0, 0);
          }
        }
        if (++kc == kchunks) {
          kc = 0;
          g += d_wrap;
        } else {
          g += d_in;
        }
      };
      auto wait_left = [&](int chunks) __attribute__((always_inline)) {
        if (chunks == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (isA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA / NL) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB / NL) : "memory");
      };
      if (isA && l > 0) {
        // every producer tile of row block tm of layer l-1 has been stored (write-through) and drained
        g_u32_f32c *c = (g_u32_f32c *)(p.cnt + ((size_t)(l - 1) * p.tiles_m + tm) * CHAIN_CNT_STRIDE);
        const unsigned target = p.target;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
          const unsigned v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((int)(v - target) >= 0) break;
          if (__builtin_amdgcn_s_memrealtime() - t0 > CHAIN_TIMEOUT_TICKS) { // never hang the GPU: flag it and go on
            if (lane == 0) __hip_atomic_store((g_u32_f32c *)p.err, 1u + (unsigned)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
      }
      issue(0);
      if (T > 1) issue(1);
      wait_left(T > 1 ? 1 : 0);
      __builtin_amdgcn_s_barrier(); // P: chunk 0 published
      if (T > 2) issue(2);
      for (int t = 0; t + 1 < T; ++t) {
        wait_left(t + 2 < T ? 1 : 0);
        __builtin_amdgcn_s_barrier();
        if (t + NSLOT - 1 < T) issue((t + NSLOT - 1) % NSLOT);
      }
      __builtin_amdgcn_s_barrier(); // R1
      __builtin_amdgcn_s_barrier(); // R2
      if (l + 1 < L) __builtin_amdgcn_s_barrier(); // S1
    }
    return;
  }

  // ---- MFMA waves ------------------------------------------------------------------------------------------------
  const int wk = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn / WN, wn = wmn % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int a_off = (wm * 32 + li) * LW_BK, b_off = wn * 32 + li;
  const int kbw = wk * KB_PER_WAVE;
  for (int l = 0; l < L; ++l) {
    const ChainLayer &Y = p.L[l];
    const int T = Y.br * (Y.k / LW_BK);
    float *__restrict__ C = (float *)Y.C;
    const int ldc = (int)Y.ldc;
    const __amdgpu_buffer_rsrc_t rsrcC = __builtin_amdgcn_make_buffer_rsrc((void *)(C + (int64_t)m0 * ldc + n0), 0, 0x7fffffff, 0x00020000);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    f32x4 fa[2];
    float fb[2][4];
    auto frag_load = [&](int buf, int slot, int kb) __attribute__((always_inline)) {
      const float *as = smem_lw + slot * SLOT + a_off;
      const float *bs = smem_lw + slot * SLOT + A_STAGE + b_off;
      fa[buf] = *(const f32x4 *)(as + (((2 * kb + lh) ^ (li & 15)) << 2));
#pragma unroll
      for (int s = 0; s < 4; ++s) fb[buf][s] = bs[(8 * kb + 4 * lh + s) * BN];
    };
    auto chunk = [&](auto slot_c, auto hn_c, bool has_next_rt) __attribute__((always_inline)) {
      constexpr int S = decltype(slot_c)::value, NS = (S + 1) % NSLOT;
      const bool has_next = decltype(hn_c)::value == 1 ? true : has_next_rt;
#pragma unroll
      for (int q = 0; q < KB_PER_WAVE; ++q) {
        const int cur = q & 1, nxt = cur ^ 1;
        if (q + 1 < KB_PER_WAVE) frag_load(nxt, S, kbw + q + 1);
        else frag_load(nxt, NS, kbw);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], fb[cur][s], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (q == KB_HALF - 1 && has_next) {
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      constexpr int PF = KB_PER_WAVE & 1;
      if constexpr (decltype(hn_c)::value != 1)
        asm volatile("" : "+v"(fa[PF]), "+v"(fb[PF][0]), "+v"(fb[PF][1]), "+v"(fb[PF][2]), "+v"(fb[PF][3]));
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    using HY = std::integral_constant<int, 1>;
    using HR = std::integral_constant<int, 2>;

    __builtin_amdgcn_s_barrier(); // P
    __builtin_amdgcn_sched_barrier(0);
    frag_load(0, 0, kbw);
    {
      int t = 0;
      for (; t + NSLOT < T; t += NSLOT) {
        chunk(S0{}, HY{}, true);
        chunk(S1{}, HY{}, true);
        chunk(S2{}, HY{}, true);
        chunk(S3{}, HY{}, true);
      }
      for (;;) {
        chunk(S0{}, HR{}, t + 1 < T);
        if (++t == T) break;
        chunk(S1{}, HR{}, t + 1 < T);
        if (++t == T) break;
        chunk(S2{}, HR{}, t + 1 < T);
        if (++t == T) break;
        chunk(S3{}, HR{}, t + 1 < T);
        if (++t == T) break;
      }
    }
    // combine the K groups and store, exactly as brgemm_f32_lw does (group order; bias; relu; 16-byte write-through stores)
    __syncthreads(); // R1
    float *red = smem_lw;
    {
      float *dst = red + (wk * (WM * WN) + wmn) * 1024 + lane;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64] = acc[r];
    }
    __syncthreads(); // R2
    constexpr int IPG = 4 / WK;
    const int c4 = lane & 7, rsel = lane >> 3;
    f32x4 bias4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (Y.ep & EP_BIAS) bias4 = *(const f32x4 *)((const float *)Y.D + n0 + wn * 32 + 4 * c4);
#pragma unroll
    for (int j = 0; j < IPG; ++j) {
      const int q = 8 * (wk * IPG + j) + rsel;
      const int r = (q & 3) + 4 * (q >> 3), lh2 = (q >> 2) & 1;
      const float *src = red + wmn * 1024 + r * 64 + lh2 * 32 + 4 * c4;
      f32x4 v = *(const f32x4 *)src;
#pragma unroll
      for (int g = 1; g < WK; ++g) v += *(const f32x4 *)(src + g * (WM * WN) * 1024);
      v += bias4;
      if (Y.ep & EP_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrcC, (unsigned)(((wm * 32 + q) * ldc + wn * 32 + 4 * c4) * 4), 0, 16);
    }
    if (l + 1 == L) break;
    // ---- seam: publish this tile to the row block's consumers ---------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // every storing wave drains its write-through stores (and is done with `red`)
    __builtin_amdgcn_s_barrier();                                // S1
    if (wave == 0 && lane == 0)
      __hip_atomic_fetch_add((g_u32_f32c *)(p.cnt + ((size_t)l * p.tiles_m + tm) * CHAIN_CNT_STRIDE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int WM, int WN, int WK, int NL> static hipError_t launch_f32_chain_t(const ChainArgs &a, hipStream_t s) {
  constexpr int BM = 32 * WM, BN = 32 * WN, NT = 64 * (WM * WN * WK + 2 * NL);
  constexpr size_t lds = (size_t)LW_NSLOT * (BM * LW_BK + LW_BK * BN) * sizeof(float);
  static std::atomic<unsigned long long> lds_set{0};
  if (hipError_t e = ensure_dynamic_lds((const void *)brgemm_f32_lw_chain<WM, WN, WK, NL>, (int)lds, lds_set); e != hipSuccess) return e;
  ChainArgs args = a;
  args.tiles_m = a.m / BM;
  args.tiles_n = a.n / BN;
  const long long tiles = (long long)args.tiles_m * args.tiles_n;
  if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL((brgemm_f32_lw_chain<WM, WN, WK, NL>), dim3((unsigned)tiles), dim3(NT), lds, s, args);
  return hipGetLastError();
}

// tile as in launch_f32_lw: 1 = 64x64 + K2, 2 = 64x32 + K4 (K-split tiles: 16-byte stores; 32x32 + K4 measured slower than three launches)
bool f32_chain_tile_dims(int tile, int *bm, int *bn) {
  switch (tile) {
  case 1: *bm = 64, *bn = 64; return true;
  case 2: *bm = 64, *bn = 32; return true;
  default: return false;
  }
}
hipError_t launch_f32_chain(int tile, const ChainArgs &a, hipStream_t s) {
  switch (tile) {
  case 1: return launch_f32_chain_t<2, 2, 2, 2>(a, s);
  case 2: return launch_f32_chain_t<2, 1, 4, 1>(a, s);
  default: return hipErrorInvalidValue;
  }
}

// The 64x32 tile's loader waves: 21 (default) = two for the A panel (16 requests per chunk) and one for B (8) - every loader issues 8;
// C3 9.912 -> 9.881 us, four alternating pairs on one box (profiles/r04_c3_loaders_and_launch_knobs.txt); 11 = one per panel
// (rounds 2-3), 22 = two each (9.888). TPP_HIP_F32_LW_C3_LOADERS for A/B runs.
static int f32_lw_c3_loaders() {
  static const int v = [] {
    const char *e = getenv("TPP_HIP_F32_LW_C3_LOADERS");
    return e ? atoi(e) : 21;
  }();
  return v;
}
// tile: 0 = 64x64 (4 MFMA waves), 1 = 64x64 with K split over 2 wave groups (8 MFMA waves, two per
// SIMD), 2 = 64x32 with K split over 4 (8 MFMA waves), 3 = 32x32 with K split over 4 (4 MFMA waves)
hipError_t launch_f32_lw(int tile, const GemmArgs &a, hipStream_t s) {
  switch (tile) {
  case 0: return launch_lw_t<2, 2, 1>(a, s);
  // two loader waves per panel for the 8-wave tile (C2): the 16 + 16 requests of a chunk - above all of chunk 0, which every MFMA
  // wave waits for - go out in half the time; same-box A/B 18.10 -> 17.97 us. The 64x32 tile (C3) measured 1 % slower with them.
  case 1: return launch_lw_t<2, 2, 2, 2>(a, s); // (four per panel: 18.25 us)
  // 64x32 with K split over FOUR wave groups: 8 MFMA waves = two per SIMD, like the 64x64 k2 tile - one wave's fragment reads and
  // barrier waits hide behind the other's MFMAs. C3 (512 x 1024 x 1024): 10.52 -> 10.21 us same-box against the K2 split (4 waves).
  case 2: return f32_lw_c3_loaders() == 11 ? launch_lw_t<2, 1, 4>(a, s) : f32_lw_c3_loaders() == 22 ? launch_lw_t<2, 1, 4, 2>(a, s) : launch_lw_t<2, 1, 4, 2, LW_NSLOT, 1>(a, s);
  // (the same tile with its K split over EIGHT wave groups - two MFMA waves per SIMD, one k-block per wave and chunk, the barrier behind
  // the block - measured 3-4.5 % slower on the reference's batch-256 layers: profiles/r04_c3_loaders_and_launch_knobs.txt (5))
  case 3: return launch_lw_t<1, 1, 4>(a, s);
  // 128x64 for large outputs: 8 MFMA waves (4 x 2 tiles of 32x32), two loader waves per panel, a 3-slot ring (48 KiB per slot)
  case 4: return launch_lw_t<4, 2, 1, 2, 3>(a, s);
  default: return hipErrorInvalidValue;
  }
}

} // namespace tpp

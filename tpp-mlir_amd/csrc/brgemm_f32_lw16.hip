// brgemm_f32_lw16.hip - f32 batch-reduce GEMM on 32 x 16 output tiles built from 16 x 16 blocks (v_mfma_f32_16x16x4_f32);
// loader-wave structure.
//
// Why (round 5, the reference's own benchmark shape set - benchmarks/config/matmul/*.json, fc/*.json): eight of its seventeen
// shapes have M = 128, i.e. 96 .. 128 output tiles of 32 x 32 on a 256-CU chip. A CU's matrix pipes run a 32 x 32 x 64 chunk in
// 512 cycles whatever else happens, so such a layer's K loop takes as long as a 256-tile layer's while half the chip idles; sharing
// a tile's batch-reduce range between workgroups (the SPLIT kernels of brgemm_f32_lw.hip) costs 2.6-2.7 us for the hand-off and
// pays only for K >= ~2000. Half-width tiles put a workgroup on every CU WITHOUT any hand-off: 128 x 1024 -> 256 tiles of 32 x 16.
// The 32x32x2 MFMA cannot make a 16-wide tile (its B operand is 32 columns); v_mfma_f32_16x16x4_f32 can, at the same rate
// (64 flop / cycle / SIMD: 32 cycles per instruction, 40 dependent - two accumulators per wave alternate, so the pipe stays full).
//
// (The kernel is written for (16 RB) x (16 CB) tiles; 16 x 48 - 256 x 768 outputs are exactly 256 of them - was measured and lost
// to 192 tiles of 32 x 32: 16 KiB of panel per 98 kflop chunk on every CU is more than the L2s deliver. Only 32 x 16 is instantiated.)
//
// Structure (as brgemm_f32_lw.hip), tile = (16 RB) x (16 CB): 4 MFMA waves, each one 16-k quarter of every 64-k chunk with RB x CB
// accumulators of 16 x 16, combined once through LDS; RB loader waves for the A panel and CB for the B panel (4 LDS-DMA instructions
// of 1 KiB per chunk and wave), a 4-slot ring three chunks ahead, ONE raw s_barrier per chunk in the middle of the chunk's MFMAs.
//   A image [32 rows][64 k], 16-byte pieces XOR-swizzled with the row (source address and fragment read): lane (i = l & 15, g = l >> 4)
//     of k-quarter wk reads ONE ds_read_b128 per row block = k 16 wk + 4 g .. + 3 of its row: at MFMA step s every lane group g
//     holds a different k (16 wk + 4 g + s) - any order of the k values inside a chunk is a valid order of additions.
//   B image: LDS row R of a 16-row block holds the chunk's k row 4 (R % 4) + R / 4 (the 4 x 4 transpose of the row order is applied
//     to the DMA's SOURCE addresses): at step s lane group g reads LDS row 4 s + g = k row 4 g + s, the same k as its A value, and
//     the four groups' rows are 64 bytes apart = four different bank quarters (rows 4 g + s would all share one: 4-way conflicts).
// One workgroup = 4 + RB + CB waves, 16 (RB + CB) KiB of LDS. GROUPED: tile-queue groups (grid (items, n / 16, m / 32)); k = 32 tiles with even batch
// counts build a chunk from two batch elements like brgemm_f32_lw's pair mode.
#include "gemm_common.h"
#include "xsmm_desc.h"
#include <type_traits>

namespace tpp {

constexpr int L16_BK = 64, L16_NSLOT = 4;
typedef __attribute__((address_space(3))) void lds_void_l16;

template <int RB, int CB, bool GROUPED>
__global__ __launch_bounds__(64 * (4 + RB + CB)) void brgemm_f32_lw16(GemmArgs p, const WorkItem *__restrict__ items) {
  constexpr int BM = 16 * RB, BN = 16 * CB;
  constexpr int A_ST = BM * L16_BK, B_ST = L16_BK * BN, SLOT = A_ST + B_ST; // floats
  constexpr int NLW = RB + CB;                                              // loader waves: A loaders 0 .. RB-1, B loaders RB .. RB+CB-1
  constexpr int PPR = 4 * CB;                                               // 16-byte pieces per B row
  extern __shared__ __attribute__((aligned(16))) float smem_l16[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int hw_wave = __builtin_amdgcn_readfirstlane(tid >> 6); // loaders first (waves start in order), then the MFMA waves (k quarters 0 .. 3)
  WorkItem it{p.A, p.B, p.C, p.D, (int64_t)p.br};
  int tm, tn;
  if constexpr (GROUPED) {
    if (items) it = items[blockIdx.x];
    tm = (int)blockIdx.z, tn = (int)blockIdx.y;
  } else {
    // XCD x (= workgroup id mod 8) owns a contiguous eighth of the column tiles and every row block: each byte of B leaves HBM once
    const int id = (int)blockIdx.x;
    if (p.xn_shift) {
      const int per = p.tiles_n >> 3, j = id >> 3;
      tn = (id & 7) * per + j % per;
      tm = j / per;
    } else {
      tn = id % p.tiles_n;
      tm = id / p.tiles_n;
    }
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const float *__restrict__ A = (const float *)it.A;
  const float *__restrict__ B = (const float *)it.B;
  float *__restrict__ C = (float *)it.C;
  const bool pair = GROUPED && p.k == 32; // a chunk = the 32-k blocks of two consecutive batch elements (even batch counts)
  const int kchunks = pair ? 1 : p.k / L16_BK;
  const int T = pair ? (int)it.br / 2 : (int)it.br * kchunks;

  if (hw_wave < NLW) {
    // ---- loader waves ------------------------------------------------------------------------------------------------
    const bool isA = hw_wave < RB;
    const int part = isA ? hw_wave : hw_wave - RB;
    unsigned vo[4];
    if (isA) {
      // A loader `part` owns row block part: instruction i = rows 16 part + 4 i .. + 3 (lane -> row + lane / 16, piece lane % 16,
      // XOR-ed with the row's low four bits)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 16 * part + 4 * i + (lane >> 4), pc = (lane & 15) ^ (r & 15);
        vo[i] = (unsigned)((r * (int)p.lda + (pair ? (pc >> 3) * (int)p.stride_a + 4 * (pc & 7) : 4 * pc)) * 4);
      }
    } else {
      // B loader `part`: instructions v = 4 part + i fill the LDS pieces 64 v .. 64 v + 63 of the chunk image [64 rows][PPR pieces];
      // LDS row R holds the chunk's k row 16 (R / 16) + 4 (R % 4) + (R % 16) / 4 (the 4 x 4 transpose of every 16-row block)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int P = 64 * (4 * part + i) + lane, R = P / PPR, pc = P % PPR;
        const int k = 16 * (R >> 4) + 4 * (R & 3) + ((R & 15) >> 2);
        vo[i] = (unsigned)((k * (int)p.ldb + 4 * pc) * 4) + (pair && k >= 32 ? (unsigned)(((int)p.stride_b - 32 * (int)p.ldb) * 4) : 0u);
      }
    }
    const float *g = isA ? A + (int64_t)m0 * p.lda : B + n0;
    int kc = 0;
    const int64_t d_in = isA ? (int64_t)L16_BK : (int64_t)L16_BK * p.ldb;
    const int64_t d_wrap = (isA ? p.stride_a : p.stride_b) * (pair ? 2 : 1) - (int64_t)(kchunks - 1) * d_in;
    auto issue = [&](int slot) __attribute__((always_inline)) {
      float *base = smem_l16 + slot * SLOT + (isA ? 0 : A_ST) + part * 1024; // (4 instructions x 256 floats per loader wave)
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_l16 *)(base + i * 256), 16, vo[i], 0, 0, 0);
      if (++kc == kchunks) {
        kc = 0;
        g += d_wrap;
      } else {
        g += d_in;
      }
    };
    auto wait_left = [&](int chunks) __attribute__((always_inline)) { // this wave's DMA of all but the `chunks` youngest chunks has landed (4 instructions per chunk)
      if (chunks == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (chunks == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    };
    if (T > 0) issue(0);
    if (T > 1) issue(1);
    wait_left(T > 1 ? 1 : 0);
    __builtin_amdgcn_s_barrier(); // chunk 0 published
    if (T > 2) issue(2);
    for (int t = 0; t + 1 < T; ++t) {
      wait_left(t + 2 < T ? 1 : 0); // chunk t + 1 has landed (chunk t + 2 may still fly)
      __builtin_amdgcn_s_barrier();  // = the MFMA waves' mid-chunk barrier of chunk t: publishes t + 1, retires the slot of t - 1
      if (t + 3 < T) issue((t + 3) % L16_NSLOT);
    }
    return; // ended waves do not take part in later barriers
  }

  // ---- MFMA waves ----------------------------------------------------------------------------------------------------
  const int wk = hw_wave - NLW;
  const int li = lane & 15, lg = lane >> 4;
  f32x4 acc[RB][CB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 fa[2][RB];
  float fb[2][CB][4];
  const int a_off = li * L16_BK + (((4 * wk + lg) ^ li) << 2); // row li of a row block: the swizzle term is row & 15 = li for every block
  const int b_off = (16 * wk + lg) * BN + li;                  // LDS row 16 wk + 4 s + lg at step s, column li of a column block
  auto frag_load = [&](int buf, int slot) __attribute__((always_inline)) {
    const float *s_ = smem_l16 + slot * SLOT;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) fa[buf][rb] = *(const f32x4 *)(s_ + a_off + rb * 16 * L16_BK);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int s = 0; s < 4; ++s) fb[buf][cb][s] = s_[A_ST + b_off + 4 * s * BN + 16 * cb];
  };
  __builtin_amdgcn_s_barrier(); // chunk 0 published
  __builtin_amdgcn_sched_barrier(0);
  if (T > 0) {
    frag_load(0, 0);
    int slot = 0;
    auto mul = [&](int buf, int s) __attribute__((always_inline)) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[buf][rb][s], fb[buf][cb][s], acc[rb][cb], 0, 0, 0);
    };
    auto chunk = [&](auto cur_c, bool has_next) __attribute__((always_inline)) {
      constexpr int CUR = decltype(cur_c)::value, NXT = CUR ^ 1;
      const int ns = slot + 1 == L16_NSLOT ? 0 : slot + 1;
      mul(CUR, 0);
      mul(CUR, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) {
        __builtin_amdgcn_s_barrier(); // chunk t + 1 published by the loaders; the slot of chunk t - 1 retired
        __builtin_amdgcn_sched_barrier(0);
        frag_load(NXT, ns);
      }
      __builtin_amdgcn_sched_barrier(0);
      mul(CUR, 2);
      mul(CUR, 3);
      __builtin_amdgcn_sched_barrier(0);
      slot = ns;
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 2 < T; t += 2) {
      chunk(B0{}, true);
      chunk(B1{}, true);
    }
    if (t + 1 < T) {
      chunk(B0{}, true);
      chunk(B1{}, false);
    } else {
      chunk(B0{}, false);
    }
  }

  // ---- combine the four k quarters through LDS, finish: (+ C) + bias, relu, 16-byte stores ---------------------------------
  // parked as [quarter][block rb * CB + cb][register r][lane]: element (row 16 rb + 4 g + r, column 16 cb + i) sits at lane 16 g + i,
  // so four consecutive columns of a row are 16 contiguous bytes
  constexpr int NBLK = RB * CB;
  __syncthreads();
  float *red = smem_l16;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wk * NBLK + rb * CB + cb) * 4 + r) * 64 + lane] = acc[rb][cb][r];
  __syncthreads();
  const int q = wk * 64 + lane; // 16-byte piece of the tile: row q / PPR, columns 4 (q % PPR) ..
  if (q >= BM * PPR) return;
  const int row = q / PPR, cpp = q % PPR;
  const int rb = row >> 4, g = (row >> 2) & 3, r = row & 3, cb = cpp >> 2, cp = cpp & 3;
  const float *src = red + ((rb * CB + cb) * 4 + r) * 64 + g * 16 + 4 * cp;
  f32x4 v = *(const f32x4 *)src;
#pragma unroll
  for (int w = 1; w < 4; ++w) v += *(const f32x4 *)(src + w * NBLK * 256);
  const __amdgpu_buffer_rsrc_t rsrcC = __builtin_amdgcn_make_buffer_rsrc((void *)(C + (int64_t)m0 * p.ldc + n0), 0, 0x7fffffff, 0x00020000);
  const unsigned co = (unsigned)((row * (int)p.ldc + 4 * cpp) * 4);
  if (!(p.ep & EP_BETA0)) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcC, co, 0, 0));
  if (p.ep & EP_BIAS) v += *(const f32x4 *)((const float *)it.D + n0 + 4 * cpp);
  if (p.ep & EP_RELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
  }
  // (a row of the tile is 64 - 192 bytes, not whole cache lines: plain stores; write-through pays only for whole lines, gemm_common.h)
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrcC, co, 0, 0);
}

template <int RB, int CB> static hipError_t launch_l16_t(const GemmArgs &a, const WorkItem *items, int n_items, bool grouped, hipStream_t s) {
  constexpr int BM = 16 * RB, BN = 16 * CB, NT = 64 * (4 + RB + CB);
  constexpr size_t lds = (size_t)L16_NSLOT * (BM * L16_BK + L16_BK * BN) * sizeof(float);
  static_assert(4 * RB * CB * 256 <= L16_NSLOT * (BM * L16_BK + L16_BK * BN), "the parked partials fit into the ring");
  GemmArgs args = a;
  args.tiles_m = a.m / BM;
  args.tiles_n = a.n / BN;
  args.xn_shift = 0;
  if (grouped) {
    static std::atomic<unsigned long long> lds_set{0};
    if (hipError_t e = ensure_dynamic_lds((const void *)brgemm_f32_lw16<RB, CB, true>, (int)lds, lds_set); e != hipSuccess) return e;
    if (args.tiles_n > 65535 || args.tiles_m > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL((brgemm_f32_lw16<RB, CB, true>), dim3((unsigned)n_items, args.tiles_n, args.tiles_m), dim3(NT), lds, s, args, items);
  } else {
    static std::atomic<unsigned long long> lds_set{0};
    if (hipError_t e = ensure_dynamic_lds((const void *)brgemm_f32_lw16<RB, CB, false>, (int)lds, lds_set); e != hipSuccess) return e;
    const long long tiles = (long long)args.tiles_m * args.tiles_n;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    args.xn_shift = (args.tiles_n & 7) == 0 ? 1 : 0; // (flag: the XCD-blocked tile order)
    hipLaunchKernelGGL((brgemm_f32_lw16<RB, CB, false>), dim3((unsigned)tiles), dim3(NT), lds, s, args, (const WorkItem *)nullptr);
  }
  return hipGetLastError();
}

// tile 0 = 32 x 16. Preconditions (checked by the callers): f32, no VNNI operand, m and n multiples of the tile,
// k % 64 == 0 (grouped also k == 32 with even batch counts), lda / ldb / ldc / strides multiples of 4 and below 2^22, A, B, C and the
// bias row 16-byte aligned
hipError_t launch_f32_lw16(int tile, const GemmArgs &a, const WorkItem *items, int n_items, bool grouped, hipStream_t s) {
  switch (tile) {
  case 0: return launch_l16_t<2, 1>(a, items, n_items, grouped, s);
  default: return hipErrorInvalidValue;
  }
}

} // namespace tpp

// brgemm_f32_lw16.hip - f32 batch-reduce GEMM on 32 x 16 output tiles (v_mfma_f32_16x16x4_f32), loader-wave structure.
//
// Why (round 5, the reference's own benchmark shape set - benchmarks/config/matmul/*.json, fc/*.json): eight of its seventeen
// shapes have M = 128, i.e. 96 .. 128 output tiles of 32 x 32 on a 256-CU chip. A CU's matrix pipes run a 32 x 32 x 64 chunk in
// 512 cycles whatever else happens, so such a layer's K loop takes as long as a 256-tile layer's while half the chip idles; sharing
// a tile's batch-reduce range between workgroups (the SPLIT kernels of brgemm_f32_lw.hip) costs 2.6-2.7 us for the hand-off and
// pays only for K >= ~2000. Half-width tiles put a workgroup on every CU WITHOUT any hand-off: 128 x 1024 -> 256 tiles of 32 x 16.
// The 32x32x2 MFMA cannot make a 16-wide tile (its B operand is 32 columns); v_mfma_f32_16x16x4_f32 can, at the same rate
// (64 flop / cycle / SIMD: 32 cycles per instruction, 40 dependent - two accumulators per wave alternate, so the pipe stays full).
//
// Structure (as brgemm_f32_lw.hip): 4 MFMA waves, each one 16-k quarter of every 64-k chunk (two 16 x 16 accumulators: rows 0-15
// and 16-31), combined once through LDS; 2 loader waves for the A panel (32 rows x 64 k = 8 instructions of 1 KiB per chunk) and 1
// for the B panel (64 k x 16 columns = 4 instructions), LDS-DMA through a 4-slot ring three chunks ahead, ONE raw s_barrier per
// chunk in the middle of the chunk's MFMAs.
//   A image [32 rows][64 k], 16-byte pieces XOR-swizzled with the row (source address and fragment read): lane (i = l & 15, g = l >> 4)
//     of k-quarter wk reads ONE ds_read_b128 per row block = k 16 wk + 4 g .. + 3 of its row: at MFMA step s every lane group g
//     holds a different k (16 wk + 4 g + s) - any order of the k values inside a chunk is a valid order of additions.
//   B image: LDS row R of a 16-row block holds the chunk's k row 4 (R % 4) + R / 4 (the 4 x 4 transpose of the row order is applied
//     to the DMA's SOURCE addresses): at step s lane group g reads LDS row 4 s + g = k row 4 g + s, the same k as its A value, and
//     the four groups' rows are 64 bytes apart = four different bank quarters (rows 4 g + s would all share one: 4-way conflicts).
// One workgroup = 7 waves, 48 KiB of LDS. GROUPED: tile-queue groups (grid (items, n / 16, m / 32)); k = 32 tiles with even batch
// counts build a chunk from two batch elements like brgemm_f32_lw's pair mode.
#include "gemm_common.h"
#include "xsmm_desc.h"
#include <type_traits>

namespace tpp {

// Ring depth 4 (48 KiB). An 8-slot ring (seven chunks in flight, filled two chunks per barrier) was built for the long-K shapes, whose
// chunk time rises from 0.13 us (K = 1024) to 0.18 us (K = 4096: every XCD streams all of A besides its eighth of B - 32 MB per
// call from beyond the L2s): it was SLOWER everywhere (128 x 1024 x 1024 4.94 -> 5.77 us, x 4096 14.6 -> 15.8) - the loaders' DMA
// issue rate (12 instructions per 256-cycle chunk) is the pace-maker, and every extra request in front of a barrier holds it back.
// Long reductions go to the 32x32 tiles with the k range shared between XCD-aligned workgroups instead (launch_gemm).
constexpr int L16_BK = 64, L16_NSLOT = 4, L16_BM = 32, L16_BN = 16;
constexpr int L16_A = L16_BM * L16_BK, L16_B = L16_BK * L16_BN, L16_SLOT = L16_A + L16_B; // floats: 2048 + 1024
typedef __attribute__((address_space(3))) void lds_void_l16;

template <bool GROUPED>
__global__ __launch_bounds__(64 * 7) void brgemm_f32_lw16(GemmArgs p, const WorkItem *__restrict__ items) {
  extern __shared__ __attribute__((aligned(16))) float smem_l16[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int hw_wave = __builtin_amdgcn_readfirstlane(tid >> 6); // 0, 1: A loaders; 2: B loader; 3 .. 6: MFMA waves (k quarters 0 .. 3)
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
  const int m0 = tm * L16_BM, n0 = tn * L16_BN;
  const float *__restrict__ A = (const float *)it.A;
  const float *__restrict__ B = (const float *)it.B;
  float *__restrict__ C = (float *)it.C;
  const bool pair = GROUPED && p.k == 32; // a chunk = the 32-k blocks of two consecutive batch elements (even batch counts)
  const int kchunks = pair ? 1 : p.k / L16_BK;
  const int T = pair ? (int)it.br / 2 : (int)it.br * kchunks;

  if (hw_wave < 3) {
    // ---- loader waves ------------------------------------------------------------------------------------------------
    const bool isA = hw_wave < 2;
    const int part = hw_wave; // A: instructions part, part + 2, part + 4, part + 6 (rows 4 v .. 4 v + 3 each)
    unsigned vo[4];
    if (isA) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * (part + 2 * i) + (lane >> 4), pc = (lane & 15) ^ (r & 15);
        vo[i] = (unsigned)((r * (int)p.lda + (pair ? (pc >> 3) * (int)p.stride_a + 4 * (pc & 7) : 4 * pc)) * 4);
      }
    } else {
      // instruction v fills LDS rows 16 v + lane / 4 (piece lane % 4) from k row 16 v + 4 ((lane / 4) % 4) + lane / 16
      const int kl = 4 * ((lane >> 2) & 3) + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) vo[i] = (unsigned)(((16 * i + kl) * (int)p.ldb + 4 * (lane & 3)) * 4) + (pair && i >= 2 ? (unsigned)(((int)p.stride_b - 32 * (int)p.ldb) * 4) : 0u);
    }
    const float *g = isA ? A + (int64_t)m0 * p.lda : B + n0;
    int kc = 0;
    const int64_t d_in = isA ? (int64_t)L16_BK : (int64_t)L16_BK * p.ldb;
    const int64_t d_wrap = (isA ? p.stride_a : p.stride_b) * (pair ? 2 : 1) - (int64_t)(kchunks - 1) * d_in;
    auto issue = [&](int slot) __attribute__((always_inline)) {
      float *base = smem_l16 + slot * L16_SLOT + (isA ? 0 : L16_A);
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int v = isA ? part + 2 * i : i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_l16 *)(base + v * 256), 16, vo[i], 0, 0, 0);
      }
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
  const int wk = hw_wave - 3;
  const int li = lane & 15, lg = lane >> 4;
  f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  f32x4 fa[2][2];
  float fb[2][4];
  const int a_off0 = li * L16_BK + (((4 * wk + lg) ^ li) << 2), a_off1 = a_off0 + 16 * L16_BK; // rows li and 16 + li: same swizzle term (row & 15)
  const int b_off = (16 * wk + lg) * L16_BN + li;                                               // LDS row 16 wk + 4 s + lg at step s
  auto frag_load = [&](int buf, int slot) __attribute__((always_inline)) {
    const float *s_ = smem_l16 + slot * L16_SLOT;
    fa[buf][0] = *(const f32x4 *)(s_ + a_off0);
    fa[buf][1] = *(const f32x4 *)(s_ + a_off1);
#pragma unroll
    for (int s = 0; s < 4; ++s) fb[buf][s] = s_[L16_A + b_off + 4 * s * L16_BN];
  };
  __builtin_amdgcn_s_barrier(); // chunk 0 published
  __builtin_amdgcn_sched_barrier(0);
  if (T > 0) {
    frag_load(0, 0);
    int slot = 0;
    auto chunk = [&](auto cur_c, bool has_next) __attribute__((always_inline)) {
      constexpr int CUR = decltype(cur_c)::value, NXT = CUR ^ 1;
      const int ns = slot + 1 == L16_NSLOT ? 0 : slot + 1;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[CUR][0][s], fb[CUR][s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[CUR][1][s], fb[CUR][s], acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) {
        __builtin_amdgcn_s_barrier(); // chunk t + 1 published by the loaders; the slot of chunk t - 1 retired
        __builtin_amdgcn_sched_barrier(0);
        frag_load(NXT, ns);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 2; s < 4; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[CUR][0][s], fb[CUR][s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[CUR][1][s], fb[CUR][s], acc1, 0, 0, 0);
      }
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
  // parked as [quarter][row block b][register r][lane]: element (row 16 b + 4 g + r, column i) sits at lane 16 g + i, so four
  // consecutive columns of a row are 16 contiguous bytes
  __syncthreads();
  float *red = smem_l16;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[((wk * 2 + 0) * 4 + r) * 64 + lane] = acc0[r];
    red[((wk * 2 + 1) * 4 + r) * 64 + lane] = acc1[r];
  }
  __syncthreads();
  if (wk >= 2) return;
  const int q = wk * 64 + lane;         // 16-byte piece of the tile: row q / 4, columns 4 (q % 4) ..
  const int row = q >> 2, cp = q & 3;
  const int b = row >> 4, g = (row >> 2) & 3, r = row & 3;
  const float *src = red + (b * 4 + r) * 64 + g * 16 + 4 * cp;
  f32x4 v = *(const f32x4 *)src;
#pragma unroll
  for (int w = 1; w < 4; ++w) v += *(const f32x4 *)(src + w * 512);
  const __amdgpu_buffer_rsrc_t rsrcC = __builtin_amdgcn_make_buffer_rsrc((void *)(C + (int64_t)m0 * p.ldc + n0), 0, 0x7fffffff, 0x00020000);
  const unsigned co = (unsigned)((row * (int)p.ldc + 4 * cp) * 4);
  if (!(p.ep & EP_BETA0)) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcC, co, 0, 0));
  if (p.ep & EP_BIAS) v += *(const f32x4 *)((const float *)it.D + n0 + 4 * cp);
  if (p.ep & EP_RELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
  }
  // (a row of the tile is 64 bytes - half a cache line: plain stores; write-through pays only for whole lines, gemm_common.h)
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrcC, co, 0, 0);
}

// preconditions (checked by the callers): f32, no VNNI operand, m % 32 == 0, n % 16 == 0, k % 64 == 0 (GROUPED also k == 32 with even
// batch counts), lda / ldb / strides multiples of 4 and below 2^22, A and B 16-byte aligned
hipError_t launch_f32_lw16(const GemmArgs &a, const WorkItem *items, int n_items, bool grouped, hipStream_t s) {
  constexpr size_t lds = (size_t)L16_NSLOT * L16_SLOT * sizeof(float);
  GemmArgs args = a;
  args.tiles_m = a.m / L16_BM;
  args.tiles_n = a.n / L16_BN;
  args.xn_shift = 0;
  if (grouped) {
    static std::atomic<unsigned long long> lds_set{0};
    if (hipError_t e = ensure_dynamic_lds((const void *)brgemm_f32_lw16<true>, (int)lds, lds_set); e != hipSuccess) return e;
    if (args.tiles_n > 65535 || args.tiles_m > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(brgemm_f32_lw16<true>, dim3((unsigned)n_items, args.tiles_n, args.tiles_m), dim3(64 * 7), lds, s, args, items);
  } else {
    static std::atomic<unsigned long long> lds_set{0};
    if (hipError_t e = ensure_dynamic_lds((const void *)brgemm_f32_lw16<false>, (int)lds, lds_set); e != hipSuccess) return e;
    const long long tiles = (long long)args.tiles_m * args.tiles_n;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    args.xn_shift = (args.tiles_n & 7) == 0 ? 1 : 0; // (flag: the XCD-blocked tile order)
    hipLaunchKernelGGL(brgemm_f32_lw16<false>, dim3((unsigned)tiles), dim3(64 * 7), lds, s, args, (const WorkItem *)nullptr);
  }
  return hipGetLastError();
}

} // namespace tpp

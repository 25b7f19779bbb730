// brgemm_bf16_small.hip - bf16 VNNI-2 BRGEMM for SMALL outputs on gfx950: one workgroup per 32 x 32
// output tile, its four waves split K, and no LDS in the K loop at all.
//
// Why: the reference's own benchmark configs are dominated by --batch=256 (benchmarks/config/*): a layer
// 256 x 1024 x 1024 has only 64 tiles of 64 x 64, so the 64 x 64 kernel leaves 3/4 of the chip idle and the
// layer is pure latency (9.5 us, the same as fp32). With 32 x 32 tiles every CU gets a workgroup, and with
// K split over the four waves of a workgroup NO operand is shared between waves: a wave loads its MFMA
// fragments straight from global memory into registers -
//   A fragment of v_mfma_f32_32x32x16_bf16: lane (i, h) holds A[i][k0 + 8h .. +8]  = one 16-byte load,
//   B fragment: lane (j, h) holds 8 consecutive k of column j = 4 dwords of 4 consecutive VNNI pair-rows
//               = four 4-byte loads, each coalesced over the 32 columns of the tile -
// a whole group of K steps per burst (G steps x 8 registers, two register sets), then the MFMAs. The four
// partial accumulators are combined once through LDS; bias / relu / (+C) / one RNE rounding as elsewhere.
// The same kernel serves the tile queue (items != nullptr: grid (items, tiles_n, tiles_m)) for the
// compiler-native bf16 tiles (32 x 32 x 32, 32 x 64 x 64 ...), where a batch element is 2-4 K steps.
#include "gemm_common.h"
#include "xsmm_desc.h"
#include "split_scratch.h"

namespace tpp {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// the operand pointers may come from the work list (generic pointers loaded from memory): say "global" explicitly,
// or every access becomes a flat_load / flat_store
typedef __attribute__((address_space(1))) const unsigned short g_cu16;
typedef __attribute__((address_space(1))) const unsigned int g_cu32;
typedef __attribute__((address_space(1))) const u32x4 g_cu32x4;
typedef __attribute__((address_space(1))) unsigned short g_u16;

constexpr int SG = 8; // K steps (of 16) per register set

// SPLIT (round 5): grid (items * p.split, tiles_n, tiles_m) - the K steps of ONE output tile shared by p.split workgroups (skinny
// groups with a long reduction: 128 x 1024 x 4096 as 64x64x64 tile invokes is 128 tiles of 32x32 on 256 CUs, each a latency-bound
// stream of 256 K steps). Same hand-off as the f32 SPLIT kernels (brgemm_f32_lw.hip): the f32 partial tile parked write-through in
// the stream's scratch block, arrival counter, the LAST workgroup sums the partials in split order (fixed order of additions), adds
// C / bias, relu, rounds ONCE to bf16 and stores. No workgroup waits for another.
typedef __attribute__((address_space(1))) unsigned int g_u32_s32;
// VF = 4: B is VNNI-4 [k/4][n][4] - a lane's 8 consecutive k of its column are two 8-byte loads (k-group rows 2 h, 2 h + 1 of the step)
// instead of four 4-byte loads; everything else is the same (bit-identical results on the same matrix).
typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const u32x2s g_cu32x2s;
template <bool SPLIT, int VF = 2>
__global__ __launch_bounds__(256) void brgemm_bf16_small32(GemmArgs p, const WorkItem *__restrict__ items) {
  const int item = SPLIT ? (int)blockIdx.x / p.split : (int)blockIdx.x;
  const int sp = SPLIT ? (int)blockIdx.x - item * p.split : 0;
  if (items) {
    const WorkItem it = items[item];
    p.A = it.A; p.B = it.B; p.C = it.C; p.D = it.D; p.br = (int)it.br;
  }
  __shared__ float red[4 * 16 * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int tm = (items || SPLIT) ? (int)blockIdx.z : (int)(blockIdx.x >> 1) * p.tiles_m + (int)blockIdx.z;
  const int tn = (items || SPLIT) ? (int)blockIdx.y : (int)(blockIdx.x & 1) * p.tiles_n + (int)blockIdx.y;
  const int m0 = tm * 32, n0 = tn * 32;
  const int spb = p.k >> 4;                // K steps per batch element
  const int S_all = p.br * spb;            // K steps in total
  // SPLIT: this workgroup's steps [s_lo, s_lo + S) of the tile's S_all
  const int s_lo = SPLIT ? (int)(((long long)S_all * sp) / p.split) : 0;
  const int S = SPLIT ? (int)(((long long)S_all * (sp + 1)) / p.split) - s_lo : S_all;
  const int per = (S + 3) >> 2;            // contiguous share of each wave (consecutive steps walk along cache lines)
  int s = wave * per;
  const int s_end = s + per < S ? s + per : S;

  // per-lane operand addresses of K step 0 of batch element 0
  g_cu16 *a_lane = (g_cu16 *)p.A + (int64_t)(m0 + li) * p.lda + 8 * lh;
  // (n that ends inside this 32-column tile - the reference's --tiles=64,48,64 / 32,48,32: the lanes of the missing columns re-read the
  // last existing one, in bounds; their results are never stored - the epilogue masks by the quad's first column, n % 4 == 0)
  const int bcol = n0 + li < p.n ? n0 + li : p.n - 1;
  g_cu32 *b_lane = (g_cu32 *)p.B + (int64_t)(4 * lh) * p.ldb + bcol; // dwords: pair-row stride = ldb
  g_cu32x2s *b_lane4 = (g_cu32x2s *)p.B + (int64_t)(2 * lh) * p.ldb + bcol; // VF = 4, 8-byte units: k-group row stride = ldb
  // position of step s: batch element b, step kk inside it
  int b = spb ? (s_lo + s) / spb : 0, kk = (s_lo + s) - b * spb;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  u32x4 fa[2][SG], fb[2][SG];
  // issue the loads of up to SG steps starting at (b, kk) into register set `set`; returns how many
  auto load_group = [&](int set, int count) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < SG; ++g) {
      if (g < count) {
        g_cu16 *ap = a_lane + (int64_t)b * p.stride_a + 16 * kk;
        g_cu32 *bp = b_lane + (((int64_t)b * p.stride_b) >> 1) + (int64_t)(8 * kk) * p.ldb;
        fa[set][g] = *(g_cu32x4 *)ap;
        if constexpr (VF == 4) {
          g_cu32x2s *bq = b_lane4 + (((int64_t)b * p.stride_b) >> 2) + (int64_t)(4 * kk) * p.ldb;
          const u32x2s q0 = bq[0], q1 = bq[p.ldb];
          fb[set][g] = u32x4{q0[0], q0[1], q1[0], q1[1]};
        } else
        fb[set][g] = u32x4{bp[0], bp[p.ldb], bp[2 * p.ldb], bp[3 * p.ldb]};
        if (++kk == spb) {
          kk = 0;
          ++b;
        }
      }
    }
  };
  auto mul_group = [&](int set, int count) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < SG; ++g)
      if (g < count)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fb[set][g]), __builtin_bit_cast(bf16x8_t, fa[set][g]),
                                                      acc, 0, 0, 0);
  };
  auto take = [&]() __attribute__((always_inline)) { // size of the next group of this wave's share
    int c = s_end - s < SG ? s_end - s : SG;
    c = c < 0 ? 0 : c;
    s += c;
    return c;
  };
  // two register sets with literal indices (register arrays must not be indexed at run time): the next
  // group is in flight while the current one multiplies
  int cur = take();
  load_group(0, cur);
  while (cur > 0) {
    int nxt = take();
    load_group(1, nxt);
    mul_group(0, cur);
    if (nxt == 0) break;
    cur = take();
    load_group(0, cur);
    mul_group(1, nxt);
  }

  // combine the four K shares: every wave parks its 16 accumulator registers, then wave w finishes register quad w (the four
  // columns 8w + 4lh + 0..3 of row li) - sum in wave order, (+ C), + bias, relu, one rounding, one 8-byte store. (Wave 0 doing
  // all four quads kept the other three idle through 48 LDS reads and the loads / stores of the whole tile.)
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave * 1024 + r * 64 + lane] = acc[r];
  __syncthreads();
  // operands were swapped (D = B^T A^T): lane (li, lh) owns row li and, in registers 4g..4g+3, columns 8g + 4lh + (0..3)
  typedef unsigned int u32x2d __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) u32x2d g_u32x2;
  typedef __attribute__((address_space(1))) const u32x2d g_cu32x2;
  g_u16 *crow = (g_u16 *)p.C + (int64_t)(m0 + li) * p.ldc + n0 + 4 * lh;
  g_cu16 *drow = (g_cu16 *)p.D + n0 + 4 * lh;
  const bool relu = (p.ep & EP_RELU) != 0;
  {
    const int g = wave;
    float v[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      v[x] = red[(4 * g + x) * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) v[x] += red[w * 1024 + (4 * g + x) * 64 + lane];
    }
    if constexpr (SPLIT) {
      // park the partial tile ([tile][split][wave][lane] float4: 1 KiB per wave instruction), arrive, and only the last workgroup goes on
      const int nsp = p.split;
      const int tile_id = (item * (int)gridDim.y + tn) * (int)gridDim.z + tm;
      float *scr = p.scratch + (size_t)tile_id * nsp * 1024;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)scr, 0, 0x7fffffff, 0x00020000);
      const unsigned pvo = (unsigned)((wave * 64 + lane) * 16);
      const f32x4 part = {v[0], v[1], v[2], v[3]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, part), rs, pvo, (unsigned)(sp * 4096), 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      unsigned *flag = (unsigned *)red;
      if (wave == 0 && lane == 0) *flag = __hip_atomic_fetch_add((g_u32_s32 *)(p.split_cnt + tile_id), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      if (*flag != (unsigned)(nsp - 1)) return;
      if (wave == 0 && lane == 0) __hip_atomic_store((g_u32_s32 *)(p.split_cnt + tile_id), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      f32x4 acc4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, pvo, 0, 16));
      for (int s2 = 1; s2 < nsp; ++s2) acc4 += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, pvo, (unsigned)(s2 * 4096), 16));
#pragma unroll
      for (int x = 0; x < 4; ++x) v[x] = acc4[x];
    }
    const bool quad_ok = n0 + 8 * g + 4 * lh < p.n; // this lane's four columns exist
    if (!quad_ok) return;
    if (!(p.ep & EP_BETA0)) {
      const u32x2d c2 = *(g_cu32x2 *)(crow + 8 * g);
      v[0] += __uint_as_float(c2[0] << 16);
      v[1] += __uint_as_float(c2[0] & 0xffff0000u);
      v[2] += __uint_as_float(c2[1] << 16);
      v[3] += __uint_as_float(c2[1] & 0xffff0000u);
    }
    if (p.ep & EP_BIAS) {
      const u32x2d b2 = *(g_cu32x2 *)(drow + 8 * g);
      v[0] += __uint_as_float(b2[0] << 16);
      v[1] += __uint_as_float(b2[0] & 0xffff0000u);
      v[2] += __uint_as_float(b2[1] << 16);
      v[3] += __uint_as_float(b2[1] & 0xffff0000u);
    }
    if (relu) {
#pragma unroll
      for (int x = 0; x < 4; ++x) v[x] = __builtin_fmaxf(v[x], 0.0f);
    }
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
    const u32x2d out = {__builtin_bit_cast(unsigned int, __builtin_convertvector(lo, bf16x2_t)), // v_cvt_pk_bf16_f32 (RNE)
                        __builtin_bit_cast(unsigned int, __builtin_convertvector(hi, bf16x2_t))};
    *(g_u32x2 *)(crow + 8 * g) = out;
  }
}

// preconditions (checked by the callers): bf16, VNNI-2 B (a.vf == 4: VNNI-4 B), m % 32 == 0, n % 32 == 0, k % 16 == 0, lda % 8 == 0,
// stride_a % 8 == 0, stride_b % 2 == 0 (VNNI-4: % 4), ldc % 4 == 0; A 16-byte, B 4-byte (VNNI-4: 8-byte), C / D 8-byte aligned
// split > 1: that many workgroups per output tile (grouped launches and single invokes in the grouped grid form)
hipError_t launch_bf16_small32(const GemmArgs &a, const WorkItem *items, int n_items, hipStream_t s, int split) {
  GemmArgs args = a;
  const int tiles_m = a.m / 32, tiles_n = (a.n + 31) / 32; // (a ragged last column tile: grouped launches only, n % 4 == 0)
  if (split > 1 && tiles_n <= 65535 && tiles_m <= 65535) {
    const long long tiles = (long long)n_items * tiles_m * tiles_n;
    if (const SplitScratch *sc = split_scratch_for(s, tiles, tiles * split * 1024)) {
      args.tiles_m = args.tiles_n = 0;
      args.split = split;
      args.scratch = sc->partial;
      args.split_cnt = sc->cnt;
      if (a.vf == 4) hipLaunchKernelGGL((brgemm_bf16_small32<true, 4>), dim3((unsigned)(n_items * split), tiles_n, tiles_m), dim3(256), 0, s, args, items);
      else hipLaunchKernelGGL((brgemm_bf16_small32<true>), dim3((unsigned)(n_items * split), tiles_n, tiles_m), dim3(256), 0, s, args, items);
      return hipGetLastError();
    } // (no scratch block: the unsplit launch)
  }
  dim3 grid;
  if (items) {
    args.tiles_m = args.tiles_n = 0;
    grid = dim3((unsigned)n_items, tiles_n, tiles_m);
  } else if ((tiles_m & 3) == 0 && (tiles_n & 1) == 0 && tiles_m / 4 <= 65535 && tiles_n / 2 <= 65535) {
    args.tiles_m = tiles_m / 4; // XCD-blocked, as the other fast kernels
    args.tiles_n = tiles_n / 2;
    grid = dim3(8, args.tiles_n, args.tiles_m);
  } else {
    args.tiles_m = args.tiles_n = 0;
    if (tiles_m > 65535 || tiles_n > 65535) return hipErrorInvalidValue;
    grid = dim3(1, tiles_n, tiles_m);
  }
  if (a.vf == 4) hipLaunchKernelGGL((brgemm_bf16_small32<false, 4>), grid, dim3(256), 0, s, args, items);
  else hipLaunchKernelGGL((brgemm_bf16_small32<false>), grid, dim3(256), 0, s, args, items);
  return hipGetLastError();
}

} // namespace tpp

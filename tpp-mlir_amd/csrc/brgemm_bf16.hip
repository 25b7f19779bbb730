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

namespace tpp {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int BKH = 64;     // k per chunk
constexpr int NSTAGE_H = 3;

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

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  int tm, tn;
  tile_of_block(blockIdx.x, p.tiles_m, p.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  const unsigned short *__restrict__ A = (const unsigned short *)p.A;
  const unsigned short *__restrict__ B = (const unsigned short *)p.B;
  unsigned short *__restrict__ C = (unsigned short *)p.C;
  const int kchunks = p.k / BKH;
  const int T = p.br * kchunks;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int crow0 = m0 + wm * 32 * TM + 4 * lh, ccol0 = n0 + wn * 32 * TN + li;
  if (!(p.ep & EP_BETA0)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc[i][j][r] = bf16_bits_to_f32(
              C[(int64_t)(crow0 + 32 * i + (r & 3) + 8 * (r >> 2)) * p.ldc + ccol0 + 32 * j]);
  }

  u32x4 ra[LA], rb[LB][4];
  auto gload = [&](int t) {
    const int b = t / kchunks, kk0 = (t - b * kchunks) * BKH;
    const unsigned short *Ab = A + (int64_t)b * p.stride_a + (int64_t)m0 * p.lda + kk0;
    // pair-row r of this chunk starts at B_b + (kk0/2 + r) * 2*ldb; column c at +2c
    const unsigned short *Bb = B + (int64_t)b * p.stride_b + (int64_t)(kk0 >> 1) * (2 * p.ldb) + 2 * (int64_t)n0;
#pragma unroll
    for (int u = 0; u < LA; ++u) {
      const int q = tid + u * NT, row = q >> 3, c = q & 7;
      if (A_ITEMS % NT == 0 || q < A_ITEMS) ra[u] = *(const u32x4 *)(Ab + (int64_t)row * p.lda + 8 * c);
    }
#pragma unroll
    for (int u = 0; u < LB; ++u) {
      const int q = tid + u * NT, g = q & 7, jq = q >> 3;
      if (B_ITEMS % NT == 0 || q < B_ITEMS) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          rb[u][r] = *(const u32x4 *)(Bb + (int64_t)(4 * g + r) * (2 * p.ldb) + 8 * jq);
      }
    }
  };
  auto swrite = [&](int stage) {
    unsigned char *as = As + stage * A_STAGE, *bs = Bs + stage * B_STAGE;
#pragma unroll
    for (int u = 0; u < LA; ++u) {
      const int q = tid + u * NT, row = q >> 3, c = q & 7;
      if (A_ITEMS % NT == 0 || q < A_ITEMS) *(u32x4 *)(as + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = ra[u];
    }
#pragma unroll
    for (int u = 0; u < LB; ++u) {
      const int q = tid + u * NT, g = q & 7, jq = q >> 3;
      if (B_ITEMS % NT == 0 || q < B_ITEMS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { // column 4*jq+e: its 4 pair-rows, transposed in registers
          u32x4 v = {rb[u][0][e], rb[u][1][e], rb[u][2][e], rb[u][3][e]};
          *(u32x4 *)(bs + g * B_GROW + ((4 * jq + e) << 4)) = v;
        }
      }
    }
  };
  auto compute = [&](int stage, int ks0, int nks) {
    const unsigned char *as = As + stage * A_STAGE;
    const unsigned char *bs = Bs + stage * B_STAGE;
#pragma unroll
    for (int q = 0; q < nks; ++q) {
      const int ks = ks0 + q;
      bf16x8_t af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + li;
        af[i] = *(const bf16x8_t *)(as + row * 128 + (((2 * ks + lh) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = (wn * TN + j) * 32 + li;
        bfr[j] = *(const bf16x8_t *)(bs + (2 * ks + lh) * B_GROW + (col << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  if (T > 0) {
    gload(0);
    swrite(0);
    if (T > 1) gload(1);
  }
  __syncthreads();
  int stage = 0;
  for (int t = 0; t < T; ++t) {
    const int nstage = stage + 1 == NSTAGE_H ? 0 : stage + 1;
    compute(stage, 0, 2);
    if (t + 1 < T) swrite(nstage);
    __syncthreads();
    if (t + 2 < T) gload(t + 2);
    compute(stage, 2, 2);
    stage = nstage;
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
        C[(int64_t)(crow0 + 32 * i + (r & 3) + 8 * (r >> 2)) * p.ldc + col] = f32_to_bf16_bits(v);
      }
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
  args.tiles_m = a.m / BM;
  args.tiles_n = a.n / BN;
  hipLaunchKernelGGL(kern, dim3(args.tiles_m * args.tiles_n), dim3(NT), lds, s, args);
  return hipGetLastError();
}

bool bf16_fast_eligible(const GemmDesc &d) {
  if (d.dtype != DT_BF16 || !d.vnni_b) return false;
  if (d.k <= 0 || d.k % BKH) return false;
  if (d.m % 64 || d.n % 64) return false;
  if ((d.lda & 7) || (d.ldb & 3) || (d.stride_a & 7) || (d.stride_b & 7)) return false;
  return true;
}

hipError_t launch_gemm_bf16_fast(const GemmDesc &d, const GemmArgs &a, hipStream_t s) {
  const int64_t t128 = (d.m % 128 == 0 && d.n % 128 == 0) ? (d.m / 128) * (d.n / 128) : 0;
  if (t128 >= 192) return launch_bf16<2, 2, 2, 2>(a, s);
  return launch_bf16<2, 2, 1, 1>(a, s);
}

} // namespace tpp

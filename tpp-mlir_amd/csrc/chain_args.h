// chain_args.h - kernel arguments of brgemm_bf16_lw.hip: one whole-layer (fused) bf16 BRGEMM, or a CHAIN of them run in
// one launch (layer l+1 reads layer l's output rows; xsmm_hip_fused_brgemm_chain_invoke in runtime.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace tpp {

constexpr int CH_MAXL = 8;                                  // layers per chain launch
constexpr int CHAIN_CNT_STRIDE = 32;                         // counters 128 B apart: device-scope atomics on ONE line serialise (~12 ns each)
constexpr unsigned long long CHAIN_TIMEOUT_TICKS = 5000000; // bound of every in-kernel spin: 50 ms of s_memrealtime (100 MHz)

struct ChainLayer {
  const void *B;   // VNNI-2 weights [K/2][ldb][2]
  const void *D;   // bias row (EP_BIAS)
  void *C;         // output [m][ldc]; the A operand of the next layer
  int64_t ldb, ldc, stride_a, stride_b; // elements (stride_a applies to this layer's A = previous C / the chain input)
  int k, br, ep, pad;                   // k per batch element (multiple of 64), batch count, EP_* bits
};

struct ChainArgs {
  const void *A;   // input of layer 0 [m][lda]
  int64_t lda;
  unsigned *cnt;   // chain mode: arrival counters [nlayers - 1][tiles_m] (CHAIN_CNT_STRIDE words apart), monotonic across launches
  unsigned *err;   // chain mode: set to 1 + layer by a workgroup whose hand-off wait timed out
  unsigned target; // chain mode: value every counter reaches in this launch (epoch * tiles_n)
  int m, n;        // rows and columns of every layer's output
  int nlayers;
  int tiles_m, tiles_n; // filled by the launcher
  int xm;               // filled by the launcher: XCD grid xm x (8 / xm) over the tile grid, 0 = linear tile order
  int dbg;              // timing experiments, ABLATION BUILDS ONLY (-DTPP_HIP_ABLATION, then TPP_HIP_CHAIN_DBG): 1 plain A loads,
                        // 2 no wait at the seams, 4 plain stores, 16 no DMA, 32 no fragment reads / MFMAs, 64 no B DMA, 128 no A DMA,
                        // 256 whole prologue before the first barrier, 512 every workgroup loads the panels of tile (0, 0) (no fabric
                        // traffic: the K loop on L2 hits only), 1024 per-chunk s_memtime stamps of the loader waves of the first 16
                        // workgroups (needs stamps; written to <TPP_HIP_CHAIN_STAMPS>.chunks, tools/stamps_report.py --chunks), 2048 the loaders issue
                        // their DMA instructions with every lane switched off (no traffic, no LDS write: the issue-side cost alone). The shipped library compiles the kernels with dbg == 0 and
                        // never reads the variable: several of these switches give wrong results by design.
  unsigned long long *stamps; // profiling (ablation builds, TPP_HIP_CHAIN_STAMPS=file): [workgroup][layer][8] s_memrealtime stamps, else nullptr
  ChainLayer L[CH_MAXL];
  // GROUPED launches (round 6, the tile queue's bf16 groups: launch_bf16_lw_grouped): a work list of tile invokes of ONE descriptor - every
  // workgroup takes A, B, C, D and the batch count of ITS item from the list, everything else (leading dimensions, strides, k, epilogue)
  // from L[0] / lda as in a single layer; m x n is then ONE item's shape and item_subs = tiles_m * tiles_n the workgroups per item
  const void *items; // WorkItem[n] in device-visible memory, nullptr in every other launch
  int item_subs;
  int pad_items;
};

// the value of ChainArgs::dbg on the host side
static inline int chain_ablation_bits() {
#ifdef TPP_HIP_ABLATION
  static const int v = [] {
    const char *e = getenv("TPP_HIP_CHAIN_DBG");
    return e ? atoi(e) : 0;
  }();
  return v;
#else
  return 0;
#endif
}

// tile: 0 = 32x64 (K split over two wave groups), 1 = 64x64, 2 = 64x128, 3 = 128x128
void blw_tile_dims(int tile, int *bm, int *bn);
hipError_t launch_bf16_lw(int tile, const ChainArgs &a, hipStream_t s);
hipError_t launch_bf16_lw_flatb(int tile, const ChainArgs &a, hipStream_t s); // the same tiles, B operand flat [k][ldb] (no VNNI flag)
hipError_t launch_bf16_lw_vnni4(int tile, const ChainArgs &a, hipStream_t s); // the same tiles, B operand VNNI-4 [k/4][ldb][4]
hipError_t launch_bf16_chain(int tile, int b_kind, const ChainArgs &a, hipStream_t s); // b_kind: 0 VNNI-2, 2 flat, 4 VNNI-4 (every layer the same)
// tile invokes of one bf16 descriptor in one launch (tile 0 = 32x64 + K2, 1 = 64x64, 4 = 32x32 + K2 (VNNI-2 only); b_kind 0 VNNI-2 / 4 VNNI-4; a.m x a.n = one item's
// shape, a.L[0] / a.lda its leading dimensions and strides; every item's batch count >= 1; even_chunks: every item has an even chunk count)
hipError_t launch_bf16_lw_grouped(int tile, int b_kind, const ChainArgs &a, const void *items, int n_items, bool even_chunks, hipStream_t s);
// 2 x 2 blocks of 64x64 items on the 128x128 tile (quads: QuadItem[n_quads], xsmm_desc.h; a.m = a.n = 128, leading dimensions / strides / k the items')
hipError_t launch_bf16_lw_quads(int b_kind, const ChainArgs &a, const void *quads, int n_quads, hipStream_t s);

// f32 chains (brgemm_f32_lw.hip): tile 1 = 64x64 + K2, 2 = 64x32 + K4 - the 64-row K-split loader-wave tiles
bool f32_chain_tile_dims(int tile, int *bm, int *bn);
hipError_t launch_f32_chain(int tile, const ChainArgs &a, hipStream_t s);

} // namespace tpp

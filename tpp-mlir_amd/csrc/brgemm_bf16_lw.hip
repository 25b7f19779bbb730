// brgemm_bf16_lw.hip - bf16 (VNNI-2 B) batch-reduce GEMM with LOADER WAVES for MID-SIZE outputs, and the same kernel
// as a CHAIN of whole-layer fused BRGEMMs in ONE launch (a rank's share of an MLP: layer l+1 on rows [r, r + BM)
// depends only on layer l's same rows).
//
// Why a new family: the shard shapes of the row-sharded MLP (m = 512 ... 2048 rows x 1024 columns, K = 1024) ran at
// 6-18 % of the bf16 MFMA peak on the existing tiles - the 128x128 LDS-DMA tile leaves 3/4 .. 1/2 of the CUs without
// a workgroup, the 64x64 register-staged tile and the 32x32 fragment-from-global tile pay a ds_write pass / uncoalesced
// fragment loads. Measured in round 2: a CU's L2 -> LDS fill path delivers 34-45 B/clk whatever the
// tile, so the time of such a layer is (panel bytes a CU must pull) / (that rate) + the fixed cost of a launch. The
// tiles here are chosen to put ONE workgroup on EVERY CU (32x64, 64x64, 64x128, 128x128 for 512 / 1024 / 2048 / 4096
// rows of a 1024-wide layer) with every panel byte moved by LDS-DMA from two loader waves through a deep ring.
//
// Structure (as brgemm_f32_lw.hip): WM*WN*WK MFMA waves + 2 loader waves (first two hardware waves: A and B).
//   MFMA waves : ds_read fragments + v_mfma_f32_32x32x16_bf16 + ONE raw s_barrier per 64-k chunk; a wave owns TM x TN
//                accumulator tiles of 32x32; WK = 2 splits the four k-steps of a chunk over two wave groups (partials
//                combined once through LDS) - the 32x64 tile keeps four SIMDs busy that way.
//   loaders    : all LDS-DMA of the workgroup, NSLOT-1 chunks ahead through an NSLOT-slot ring, counted vmcnt.
//   LDS images : A [BM rows][8 x 16 B], 16-byte column XOR ((row>>1)&7) applied to the SOURCE address and again by the
//                ds_read_b128; B = the VNNI-2 pair-rows as they are [32][BN dwords] (brgemm_bf16.hip has the notes).
//   epilogue   : (+C) / bias / relu -> v_cvt_pk_bf16_f32 (RNE, the one rounding) -> v_permlane32_swap to 16-byte
//                pieces -> per-wave LDS tile (its OWN region, the ring keeps streaming) -> coalesced 16-byte stores.
//
// Chain mode (MULTI): grid = one workgroup per output tile, all co-resident (tiles <= CUs, one workgroup per CU by
// LDS), every workgroup computes its tile (tm, tn) of EVERY layer. The seam between layer l and l+1, per row block tm:
//   producer: tile stored WRITE-THROUGH (sc1 16-byte stores), every storing wave drains vmcnt(0), workgroup barrier,
//             one lane adds 1 to cnt[l][tm] (relaxed, agent scope)             [guide: Guideline 16, recipe R1]
//   consumer: the A loader polls cnt[l][tm] (relaxed sc1 load, s_sleep) until all tiles_n producers of this launch
//             have arrived, then streams the A panel with sc1 LDS-DMA loads (L1 bypassed: no acquire fence needed
//             when producer stored sc1 and consumer loads sc1).
//   The B loader does not stop at a seam: the next layer's weight panels are prefetched under the epilogue (when the
//   chunk count of a layer is a multiple of the ring depth, so that ring positions line up).
//   Counters are monotonic: a launch adds exactly tiles_n to each, the host passes target = epoch * tiles_n; nothing
//   is reset (no memset launch in front of the kernel). Every spin is bounded by s_memrealtime: on a timeout the
//   workgroup sets *err and carries on (wrong numbers, never a hung GPU); the host checks *err at its sync points.
//   Results are bit-identical to the same layers launched one by one (same tile, same k order, same rounding).
#include "gemm_common.h"
#include "xsmm_desc.h"
#include "chain_args.h"
#include <type_traits>

namespace tpp {

typedef __bf16 bf16x8_lw __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_b;
typedef unsigned int u32x2_lw __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) unsigned int g_u32_lw;

constexpr int BLW_BK = 64; // k per chunk
#ifndef TPP_BLW_INTERLEAVE
#define TPP_BLW_INTERLEAVE 2 // fragment reads dealt between the MFMAs of a k-step: 1 = the four-accumulator tile (128x128), 2 = + 64x128; 0 = in front (A/B)
#endif
// The kernel's argument block, read where it lies (kernarg segment, constant address space): the layer table is indexed at
// run time, and a by-value / by-reference copy of the struct would be spilled to scratch (640 bytes per lane) for that.
typedef const __attribute__((address_space(4))) ChainArgs chain_kernarg_t;

// profiling stamps (chain mode, TPP_HIP_CHAIN_STAMPS, ABLATION BUILDS ONLY - in the shipped kernels this is empty: each stamp was a
// scalar load of p.stamps + a wait + a branch on the path of MFMA wave 0 and of the polling loader, five and three times per layer):
// slot of (workgroup, layer): 0 layer start, 1 chunk 0 published, 2 K loop done, 3 tile stores issued, 4 stores drained + S1 (MFMA
// wave 0); 5 A loader starts waiting, 6 producers have arrived, 7 first chunks requested
__device__ __forceinline__ void blw_stamp(chain_kernarg_t &p, int layer, int slot, int lane) {
#ifdef TPP_HIP_ABLATION
  if (p.stamps && lane == 0) p.stamps[((size_t)blockIdx.x * CH_MAXL + layer) * 8 + slot] = __builtin_amdgcn_s_memrealtime();
#endif
}

// Per-chunk stamps of the LOADER waves (ablation builds, TPP_HIP_CHAIN_DBG & 1024, chain launches; the first 16 workgroups): shader-clock
// stamps (s_memtime) at three points of every steady-state iteration - the chunk awaited has landed / the barrier has released /
// the next chunk's DMA instructions are issued - buffered in LDS behind the ring (a VMEM store would count in the loaders' vmcnt
// bookkeeping) and copied to p.stamps behind the per-layer stamps when the wave ends. A stamp is REQUESTED at its point and
// collected at the next one (s_memtime returns through lgkmcnt, ~70 cycles: waiting for it on the spot stretched the
// 4096-row chain from 28.9 to 38.3 us; collected late the three stamps cost a few issue slots). The MFMA waves are not
// stamped: s_memtime shares lgkmcnt with their fragment reads. landed -> released = how long the loader waited for the
// slowest wave at the barrier; issued(t-1) -> landed(t) = how long it waited for its own DMA. tools/stamps_report.py --chunks.
// dbg & 2048 (ablation builds): the loaders' DMA instructions are ISSUED with every lane switched off - no memory traffic, no LDS
// write, but the same instruction stream on the SIMD: what of the DMA's cost to the MFMA waves is issue-side
#ifdef TPP_HIP_ABLATION
#define BLW_DBG_EXEC_OFF() do { if (dbg & 2048) asm volatile("s_mov_b64 exec, 0" ::: "memory"); } while (0)
#define BLW_DBG_EXEC_ON() do { if (dbg & 2048) asm volatile("s_mov_b64 exec, -1" ::: "memory"); } while (0)
#else
#define BLW_DBG_EXEC_OFF() ((void)0)
#define BLW_DBG_EXEC_ON() ((void)0)
#endif
#ifdef TPP_BLW_SKIP_BARRIER
#define BLW_MID_BARRIER() ((void)0) // (timing side build: see brgemm_bf16_lw's skip switches)
#else
#define BLW_MID_BARRIER() __builtin_amdgcn_s_barrier()
#endif
constexpr int BLW_CS_ENTRIES = 64; // (layer, chunk) records per loader wave
struct BlwChunkStamps {
  unsigned long long pend = 0; // (an SGPR pair: the s_memtime in flight)
  int pend_off = -1;           // LDS byte offset it belongs to
  __device__ __forceinline__ void collect(unsigned char *smem, int lane) {
#ifdef TPP_HIP_ABLATION
    if (pend_off >= 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pend)::"memory");
      if (lane == 0) *(unsigned long long *)(smem + pend_off) = pend;
      pend_off = -1;
    }
#endif
  }
  __device__ __forceinline__ void stamp(unsigned char *smem, int area, int which, int idx, int k, int lane) {
#ifdef TPP_HIP_ABLATION
    if (area >= 0) {
      collect(smem, lane);
      if (idx < BLW_CS_ENTRIES) {
        pend_off = area + ((which * BLW_CS_ENTRIES + idx) * 3 + k) * 8;
        asm volatile("s_memtime %0" : "=s"(pend)::"memory");
      }
    }
#endif
  }
};

// s_waitcnt vmcnt(younger * PPL): this wave's DMA of all but the `younger` most recent chunks has landed
template <int PPL> __device__ __forceinline__ void blw_wait_younger(int younger) {
#define BLW_CASE(K)                                                                   \
  case K:                                                                             \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((K) * PPL > 63 ? 63 : (K) * PPL) : "memory"); \
    break;
  switch (younger) {
    BLW_CASE(0) BLW_CASE(1) BLW_CASE(2) BLW_CASE(3) BLW_CASE(4) BLW_CASE(5) BLW_CASE(6)
  default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef BLW_CASE
}

// One loader wave: LDS-DMA instructions of its panel (IS_A: the A panel [BM rows][64 k], else the B panel [32 pair-rows][BN
// dwords]) for every chunk of every layer, NSLOT - 1 chunks ahead of the MFMA waves, and its side of the barrier schedule:
//   per layer  [S2]  P (chunk 0 published), one mid-chunk barrier per further chunk, and - between layers - R1 (WK > 1) and S1.
// NL loader waves share a panel: wave `part` issues the instructions v = part, part + NL, ... of every chunk.
// In chain mode A loader 0 waits at a seam for the row block's producers (with NLA > 1 the others learn it through S2); the B
// loaders prefetch the next layer's panels across the seam when every layer's chunk count is a multiple of the ring depth.
// The steady-state loop is a literal s_waitcnt + s_barrier + the DMA instructions + ~10 scalar instructions: measured on the
// first version of this function (one general loop with a switch over the wait count, the layer bookkeeping and an argument load
// inside), a 16-chunk layer took 5.7 us with NO loads and NO MFMAs at all - the loader's own instruction stream set the pace.
// GRP (grouped launches): the panel base pointers and the batch count are the ITEM's (it_A / it_B / it_br), not the argument block's
// GRP = 2 (QUADS, round 6): the 128x128 tile over a 2 x 2 block of 64x64 ITEMS (QuadItem, chain_args.h): rows 64 .. 127 of the A panel
// come from the lower item row's block (q_delta = its distance from the upper one, minus the 64 rows the instruction offsets already
// cover), columns 64 .. 127 of the B panel from the right item column's block (q_delta = its distance from the left one): the LDS
// image is the flat 128-wide tile's, only the source addresses of the DMA instructions differ.
template <bool IS_A, int NL, int NLA, int NSLOT_C, int SUP, int BM, int BN, int WK, bool MULTI, int FB = 0, int GRP = 0>
__device__ __forceinline__ void blw_loader(chain_kernarg_t *pp, unsigned char *smem, int lane, int m0, int n0, int tm, int L, int part,
                                           const void *it_A = nullptr, const void *it_B = nullptr, int it_br = 0, unsigned q_delta = 0) {
  static_assert(GRP != 2 || (BM == 128 && BN == 128 && NL == 1 && (FB == 0 || FB == 4)), "quads: the 128x128 tile, one loader wave per panel, VNNI-2 / VNNI-4 B");
  chain_kernarg_t &p = *pp;
  constexpr int A_SLOT = BM * 128, SLOT = (BM + BN) * 128;
  // SUP = chunks per barrier (the unit everything below counts in: a "chunk" of this function is SUP 64-k chunks, a "slot" SUP
  // consecutive ring slots). SUP = 2 halves the barrier count and the loader's per-iteration scalar work for the small tiles, whose
  // 16-chunk layers spent 75 % of their time in the barrier / bookkeeping skeleton (timing with loads AND MFMAs switched off).
  constexpr int NSLOT = NSLOT_C / SUP;
  constexpr int PPC = (IS_A ? BM / 8 : BN / 8) / NL; // 1 KiB DMA instructions per 64-k chunk issued by this wave
  constexpr int PPL = SUP * PPC;                     // ... per barrier interval
  [[maybe_unused]] constexpr int RPI = 256 / BN;     // VNNI pair-rows per B instruction
  static_assert(PPC >= 1 && (IS_A ? BM / 8 : BN / 8) % NL == 0 && (!IS_A || NL == 1 || NL % 2 == 0), "panel instructions divide over the loader waves");
  static_assert(NSLOT >= 3 && NSLOT_C % SUP == 0 && (NSLOT - 1) * PPL <= 63, "ring depth / vmcnt is 6 bits");
#ifdef TPP_HIP_ABLATION
  const int dbg = p.dbg;
#else
  constexpr int dbg = 0; // the timing switches exist in ablation builds only (chain_args.h)
#endif
  const bool no_dma = (dbg & (16 | (IS_A ? 128 : 64))) != 0; // timing experiments: this panel is not fetched
  // per-chunk stamps (see BlwChunkStamps): LDS area behind the ring, this wave's row = 0 (A loader 0) / 1 (B loader 0)
  const int cs_area = ((dbg & 1024) && MULTI && part == 0 && blockIdx.x < 16 && p.stamps) ? NSLOT_C * SLOT + (WK > 1 ? (BM / 32) * (BN / 32) * 4096 : 0) : -1;
  constexpr int cs_which = IS_A ? 0 : 1;
  int cs_idx = 0;
  BlwChunkStamps cs;
  bool ahead = false;
  if (MULTI && !IS_A) {
    ahead = true;
    for (int l = 0; l < L; ++l) {
      const int Tl = p.L[l].br * (p.L[l].k / BLW_BK) / SUP;
      if (Tl < NSLOT) ahead = false; // (the run-ahead issues NSLOT - 2 chunks of the next layer during this layer's tail)
    }
  }
  // issue state: the panel base of the next chunk to request (of layer state_layer) and the constants of that layer
  int state_layer = -1, kc = 0, kchunks = 1;
  const unsigned short *g = nullptr;
  int64_t d_in = 0, d_wrap = 0;
  unsigned vo0 = 0, vo1 = 0, step = 0;
  bool sc1 = false, flat = false;
#define BLW_LOAD_STATE(l)                                                                                              \
  do {                                                                                                                 \
    state_layer = (l);                                                                                                 \
    kchunks = p.L[l].k / BLW_BK;                                                                                       \
    kc = 0;                                                                                                            \
    if (IS_A) {                                                                                                        \
      const int64_t lda_ = (l) == 0 ? p.lda : p.L[(l) > 0 ? (l)-1 : 0].ldc;                                            \
      const unsigned short *A_ = (const unsigned short *)(GRP ? it_A : (l) == 0 ? p.A : p.L[(l) > 0 ? (l)-1 : 0].C);    \
      g = A_ + (int64_t)(m0 + 8 * part) * lda_;                                                                        \
      d_in = BLW_BK;                                                                                                   \
      d_wrap = p.L[l].stride_a - (int64_t)(kchunks - 1) * BLW_BK;                                                      \
      /* instruction v covers rows 8v .. 8v+7: lane -> row 8v + lane/8, 16-byte piece lane%8 XOR ((row>>1)&7) = 4(v&1) + lane/16; */ \
      /* this wave's instructions are v = part + NL * i: NL even -> one parity (one swizzle term), NL == 1 -> alternating */ \
      const unsigned rowoff_ = (unsigned)((lane >> 3) * (int)lda_ * 2);                                                \
      vo0 = rowoff_ + (unsigned)(((lane & 7) ^ (4 * (part & 1) + (lane >> 4))) << 4);                                  \
      vo1 = NL == 1 ? rowoff_ + (unsigned)(((lane & 7) ^ (4 + (lane >> 4))) << 4) : vo0;                               \
      step = (unsigned)(NL * 8 * (int)lda_ * 2);                                                                       \
      sc1 = MULTI && (l) > 0 && !(dbg & 1); /* written by other workgroups in THIS launch: sc1 loads (L1 bypassed) */  \
    } else if (FB == 2) {                                                                                              \
      /* FLAT B ([k][ldb]), image = the chunk's 64 rows as they are, for the transpose reads of the MFMA waves: instruction v */ \
      /* covers rows (512/BN)*v ..: lane -> row lane / (BN/8), 16-byte piece lane % (BN/8); the 64-byte blocks of a row are */ \
      /* XOR-swizzled with the row (BN = 128: row & 3, BN = 64: (row >> 1) & 1) on the SOURCE side, like the A panel */ \
      constexpr int RPT_ = 512 / BN, PPR_ = BN / 8;                                                                    \
      g = (const unsigned short *)(GRP ? it_B : p.L[l].B) + n0 + (int64_t)part * RPT_ * p.L[l].ldb;                    \
      d_in = (int64_t)BLW_BK * p.L[l].ldb;                                                                             \
      d_wrap = p.L[l].stride_b - (int64_t)(kchunks - 1) * d_in;                                                        \
      const int row_ = lane / PPR_, piece_ = lane % PPR_;                                                              \
      const int swz_ = BN == 128 ? (row_ & 3) << 2 : ((row_ >> 1) & 1) << 2;                                           \
      vo0 = vo1 = (unsigned)(row_ * (int)p.L[l].ldb * 2 + ((piece_ ^ swz_) << 4));                                     \
      step = (unsigned)(NL * RPT_ * (int)p.L[l].ldb * 2);                                                              \
    } else if (FB == 4) {                                                                                              \
      /* VNNI-4 B ([k/4][ldb][4]): image = the chunk's 16 k-group rows of BN * 8 bytes as they are; instruction v covers */ \
      /* k-group rows (128/BN)*v ..: lane -> row lane / (BN/2), 16-byte piece lane % (BN/2). A fragment is two 8-byte reads. */ \
      constexpr int RPI4_ = 128 / BN, PPR4_ = BN / 2;                                                                  \
      g = (const unsigned short *)(GRP ? it_B : p.L[l].B) + 4 * (int64_t)n0 + (int64_t)part * RPI4_ * 4 * p.L[l].ldb;  \
      d_in = (int64_t)BLW_BK * p.L[l].ldb;                                                                             \
      d_wrap = p.L[l].stride_b - (int64_t)(kchunks - 1) * d_in;                                                        \
      vo0 = vo1 = (unsigned)((lane / PPR4_) * (int)p.L[l].ldb * 8 + ((lane % PPR4_) << 4));                            \
      if (GRP == 2) vo0 = vo1 = (unsigned)(((lane & 31) << 4)) + ((lane & 32) ? q_delta : 0u); /* 64 pieces = one k-group row: 32 per item */ \
      step = (unsigned)(NL * RPI4_ * (int)p.L[l].ldb * 8);                                                             \
    } else {                                                                                                           \
      g = (const unsigned short *)(GRP ? it_B : p.L[l].B) + 2 * (int64_t)n0 + (int64_t)part * RPI * 2 * p.L[l].ldb;    \
      d_in = (int64_t)(BLW_BK / 2) * 2 * p.L[l].ldb;                                                                   \
      d_wrap = p.L[l].stride_b - (int64_t)(kchunks - 1) * d_in;                                                        \
      /* instruction v covers pair-rows RPI*v ..: lane -> pair-row lane / (BN/4), 16-byte piece lane % (BN/4) */       \
      vo0 = vo1 = (unsigned)((lane / (BN / 4)) * (int)p.L[l].ldb * 4 + ((lane % (BN / 4)) << 4));                      \
      if (GRP == 2) vo0 = vo1 = (unsigned)((lane >> 5) * (int)p.L[l].ldb * 4 + ((lane & 15) << 4)) + ((lane & 16) ? q_delta : 0u); /* 32 pieces = one pair-row: 16 per item */ \
      step = (unsigned)(NL * RPI * (int)p.L[l].ldb * 4);                                                               \
    }                                                                                                                  \
    /* one chunk per batch element (the compiler's 64-k tile invokes over packed blocks): EVERY advance is the wrap - the same   */ \
    /* single add as the flat case, instead of a count-and-branch per chunk in the loader's instruction stream, which sets the */ \
    /* pace of the small tiles (32x32 + K2 grouped at K = 4096: 8.97 -> see profiles/r06_bf16_loader_advance_ab.txt)            */ \
    if (kchunks == 1) d_in = d_wrap;                                                                                   \
    flat = d_wrap == d_in;                                                                                             \
  } while (0)
  // request the next SUP 64-k chunks of the issue state into ring slot `slot` (SUP consecutive 64-k slots), advance state and slot
#define BLW_ISSUE(slot)                                                                                                \
  do {                                                                                                                 \
    _Pragma("unroll") for (int sub_ = 0; sub_ < SUP; ++sub_) {                                                         \
      unsigned char *base_ = smem + ((slot) * SUP + sub_) * SLOT + (IS_A ? 0 : A_SLOT) + part * 1024;                  \
      const __amdgpu_buffer_rsrc_t r_ = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, 0x7fffffff, 0x00020000);       \
      BLW_DBG_EXEC_OFF();                                                                                              \
      if (no_dma) {                                                                                                    \
      } else if (IS_A && sc1) {                                                                                        \
        _Pragma("unroll") for (int v = 0; v < PPC; ++v)                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_, (lds_void_b *)(base_ + v * NL * 1024), 16, (v & 1) ? vo1 : vo0, v * step, 0, 16); \
      } else {                                                                                                         \
        _Pragma("unroll") for (int v = 0; v < PPC; ++v)                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_, (lds_void_b *)(base_ + v * NL * 1024), 16, (v & 1) ? vo1 : vo0,   \
                                                     v * step + ((GRP == 2 && IS_A && v >= 8) ? q_delta : 0u), 0, 0);      \
      }                                                                                                                \
      BLW_DBG_EXEC_ON();                                                                                               \
      if (flat) { /* the batch elements continue each other (whole-layer dispatches): one 64-bit add */               \
        g += d_in;                                                                                                     \
      } else if (++kc == kchunks) {                                                                                    \
        kc = 0;                                                                                                        \
        g += d_wrap;                                                                                                   \
      } else {                                                                                                         \
        g += d_in;                                                                                                     \
      }                                                                                                                \
    }                                                                                                                  \
    slot = slot + 1 == NSLOT ? 0 : slot + 1;                                                                           \
  } while (0)
  const bool poller = MULTI && IS_A && part == 0;
  int slot = 0, s0 = 0; // next ring slot to fill; slot of the current layer's chunk 0 (the layers follow each other through the ring)
  for (int lc = 0; lc < L; ++lc) {
    const int T = (GRP ? it_br : p.L[lc].br) * (p.L[lc].k / BLW_BK) / SUP; // (a multiple of SUP: the launcher picks SUP = 1 otherwise)
    int pre = NSLOT - 2; // chunks of this layer already requested by the run-ahead of the previous layer's tail
    if (state_layer != lc) {
      BLW_LOAD_STATE(lc);
      pre = 0;
      slot = s0;
    }
    if (poller) blw_stamp(p, lc, 5, lane);
    if (poller && lc > 0 && !(dbg & 2)) {
      // every producer tile of row block tm of layer lc-1 has been stored (write-through) and drained
      g_u32_lw *c = (g_u32_lw *)(p.cnt + ((size_t)(lc - 1) * p.tiles_m + tm) * CHAIN_CNT_STRIDE);
      const unsigned target = p.target;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        const unsigned v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - target) >= 0) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > CHAIN_TIMEOUT_TICKS) { // never hang the GPU: flag it and go on
          if (lane == 0) __hip_atomic_store((g_u32_lw *)p.err, 1u + (unsigned)lc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      asm volatile("" ::: "memory");
    }
    if (poller) blw_stamp(p, lc, 6, lane);
    if (MULTI && NLA > 1 && lc > 0) __builtin_amdgcn_s_barrier(); // S2: the polling wave has seen the producers arrive
    // Prologue. (Filling the ring two chunks per barrier - a ramp through the first iterations - was measured: chunk 0 is ready
    // 0.4 us earlier, the 16-chunk loop of the 32x64 tile takes 1.0 us longer: the loader is the pace-maker and the ramp costs it
    // a second ISSUE and a variable wait per iteration. So: one burst, but behind the barrier that publishes chunk 0.)
    const int npro = T < NSLOT - 1 ? T : NSLOT - 1;
    // chunks 0 and 1 first, chunk 0 PUBLISHED as soon as it has landed, the rest of the prologue behind the barrier (the MFMA
    // waves work on chunk 0 while it is issued; dbg & 256: everything before the barrier, for A/B runs)
    const int nfirst = (dbg & 256) ? npro : (npro < 2 ? npro : 2);
    int c = pre;
    for (; c < nfirst; ++c) BLW_ISSUE(slot);
    if (poller) blw_stamp(p, lc, 7, lane);
    blw_wait_younger<PPL>((c > nfirst ? c : nfirst) - 1);
    __builtin_amdgcn_s_barrier(); // P: chunk 0 of this layer published
    for (; c < npro; ++c) BLW_ISSUE(slot);
    int t = 0;
    // steady state: chunk t+1 has landed when all but the NSLOT - 3 youngest requests have; the barrier (= the MFMA waves'
    // mid-chunk barrier of chunk t) publishes it and retires the slot of chunk t-1, which takes chunk t + NSLOT - 1
    for (; t + NSLOT - 1 < T; ++t) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 3) * PPL) : "memory");
      cs.stamp(smem, cs_area, cs_which, cs_idx, 0, lane);
      BLW_MID_BARRIER();
      cs.stamp(smem, cs_area, cs_which, cs_idx, 1, lane);
      BLW_ISSUE(slot);
      cs.stamp(smem, cs_area, cs_which, cs_idx, 2, lane);
      ++cs_idx;
    }
    // the last NSLOT - 2 barriers of the layer: nothing of THIS layer is left to request
    if (MULTI && ahead && lc + 1 < L) {
      BLW_LOAD_STATE(lc + 1); // (the slot rotation carries over into the next layer)
      for (; t + 1 < T; ++t) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 3) * PPL) : "memory");
        BLW_MID_BARRIER();
        BLW_ISSUE(slot);
      }
    } else {
      for (; t + 1 < T; ++t) {
        blw_wait_younger<PPL>(T - 2 - t);
        BLW_MID_BARRIER();
      }
    }
    if (lc + 1 == L) {
#ifdef TPP_HIP_ABLATION
      cs.collect(smem, lane);
      if (cs_area >= 0 && lane == 0) { // copy this wave's records out: [workgroup < 16][A, B][entry][3] behind the per-layer stamps
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        unsigned long long *dst = p.stamps + (size_t)p.tiles_m * p.tiles_n * CH_MAXL * 8 + ((size_t)blockIdx.x * 2 + cs_which) * (BLW_CS_ENTRIES * 3 + 1);
        dst[0] = (unsigned long long)cs_idx;
        const unsigned long long *src = (const unsigned long long *)(smem + cs_area) + cs_which * BLW_CS_ENTRIES * 3;
        for (int i = 0; i < BLW_CS_ENTRIES * 3; ++i) dst[1 + i] = i < cs_idx * 3 ? src[i] : 0ull;
      }
#endif
      return;
    }
    if constexpr (WK > 1) __builtin_amdgcn_s_barrier(); // R1 (K groups combine)
    __builtin_amdgcn_s_barrier();                        // S1 (tile stored and drained)
    s0 = (s0 + T) % NSLOT;
  }
#undef BLW_ISSUE
#undef BLW_LOAD_STATE
}

template <int WM, int WN, int WK, int TM, int TN, int NSLOT, int NLA, int NLB, int SUP, bool MULTI, int FLATB = 0, int GRP = 0>
__global__ __launch_bounds__(64 * (WM * WN * WK + NLA + NLB)) void brgemm_bf16_lw(ChainArgs p_by_value) {
  static_assert(!GRP || !MULTI, "a grouped launch is a set of single layers");
  static_assert(GRP != 2 || (WM == 2 && WN == 2 && WK == 1 && TM == 2 && TN == 2), "quads: every MFMA wave owns one item's 64x64 output");
  chain_kernarg_t *pp = (chain_kernarg_t *)__builtin_amdgcn_kernarg_segment_ptr(); // = &p_by_value (the only explicit argument)
  chain_kernarg_t &p = *pp;
  constexpr int NMW = WM * WN * WK, NOUT = WM * WN; // MFMA waves; waves that own output
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  constexpr int KS = 4 / WK, PD = KS / 2;            // k-steps of a chunk per wave; fragment prefetch distance (three steps ahead
                                                     // with the barrier one step earlier measured the same: profiles/r04_bf16_lw_read_interleave.txt)
  constexpr int A_SLOT = BM * 128, B_SLOT = BN * 128, SLOT = A_SLOT + B_SLOT;
  constexpr int ES = 64 * TN + 16;                   // bytes per staged output row (16 B pad: conflict-free 16-byte accesses)
  constexpr int STAGE_W = 32 * ES;                   // one 32-row block of a wave's tile
  // The epilogue's per-wave staging tiles live in the A halves of RETIRED ring slots (the whole LDS belongs to the ring: a fifth
  // slot for the 128x128 tile, a sixth for 64x128). After the last workgroup barrier of a layer only the slots of its last
  // 2 * SUP chunks can still be read by a slower wave; the next layer's A panels are requested after the seam barrier S1 (behind
  // every wave's epilogue), and the B loaders' run-ahead writes B halves only.
  constexpr int WPS = A_SLOT / STAGE_W;              // staging tiles per A half
  constexpr int STAGE_SLOTS = (NOUT + WPS - 1) / WPS;
  static_assert(WPS >= 1 && STAGE_SLOTS + 2 * SUP <= NSLOT, "staging tiles fit into the retired slots");
  constexpr int OFF_RED = NSLOT * SLOT;
  static_assert(WK == 1 || (WK == 2 && TM == 1 && TN == 1), "K split: two groups of single-tile waves");
  static_assert(SUP == 1 || ((SUP == 2 || SUP == 4) && TM * TN <= 2 && NSLOT % SUP == 0 && NSLOT % 2 == 0),
                "several chunks per barrier: the tiles that read a whole chunk of fragments ahead");
  constexpr int NLW = NLA + NLB; // loader waves
  static_assert(FLATB == 0 || FLATB == 2 || FLATB == 4, "0: VNNI-2 B image, 2: flat B image + transpose reads, 4: VNNI-4 B image + 8-byte reads");
  static_assert(FLATB != 4 || BN >= 32, "VNNI-4 image: whole k-group rows per DMA instruction (BN = 32: four rows of 256 bytes)");
  static_assert(FLATB != 2 || BN >= 64, "transpose-read image: 64-byte blocks swizzled inside rows of >= 128 bytes");
  static_assert(BN == 32 || BN == 64 || BN == 128, "B pair-rows per DMA instruction");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_c[];

  const int tid = threadIdx.x, lane = tid & 63;
  // the loader waves are the FIRST hardware waves (waves start in order: the first chunks are requested before the
  // MFMA waves exist); `wave` is the role: MFMA waves 0 .. NMW-1, A loaders NMW .. NMW+NLA-1, then the B loaders
  const int hw_wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = hw_wave < NLW ? NMW + hw_wave : hw_wave - NLW;
  // tile of this workgroup. Block b runs on XCD b % 8 (observed, used for speed only) and every XCD's L2 fetches the panels of
  // its workgroups itself: the XCDs form an xm x xn grid over the tile grid (chosen by the launcher to minimise the bytes all
  // eight L2s pull together, xn * |A| + xm * |W|: a 512-row layer on 8 x 1 makes every L2 fetch ALL of W)
  const int b = (int)blockIdx.x;
  int tm, tn;
  // GRP: workgroup b = tile (b % item_subs) of item b / item_subs; the item's operands replace the argument block's (uniform values:
  // read through the scalar unit's view of the list)
  [[maybe_unused]] const void *it_A = nullptr, *it_B = nullptr, *it_D = nullptr;
  [[maybe_unused]] void *it_C = nullptr;
  [[maybe_unused]] int it_br = 0;
  [[maybe_unused]] unsigned q_da = 0, q_db = 0;
  if constexpr (GRP == 2) {
    // QUADS: workgroup b = the b-th 2 x 2 block of items. Panels: the upper-left item's A rows / B columns + the two distances; output
    // and bias: MFMA wave (wm, wn) owns item (wm, wn)'s 64x64 block outright (its own base pointer, the items' ldc)
    typedef const __attribute__((address_space(4))) QuadItem quad_c_t;
    quad_c_t &w = ((quad_c_t *)p.items)[b];
    const int qi = wave < WM * WN * WK ? wave : 0; // (= wm * 2 + wn; the loader waves do not store)
    it_A = w.A;
    it_B = w.B;
    it_C = w.C[qi];
    it_D = w.D[qi & 1];
    it_br = (int)w.br;
    q_da = w.da;
    q_db = w.db;
    tm = tn = 0;
  } else if constexpr (GRP) {
    // (plain order: workgroup b = the b-th tile of the list. Handing XCD x the x-th EIGHTH of the list - contiguous block rows of the
    // layer - was measured and is WORSE for the reference's shapes, 1024 x 2560 x 1024 15.0 -> 18.1 us: with row-major items and a
    // column count that is a multiple of 8 the plain order already gives every XCD its own eighth of the COLUMNS, i.e. of W, the
    // larger operand; profiles/r06_bf16_grouped_lw.txt)
    // ... but the tiles of ONE item (2 or 4 workgroups that share the item's A rows / B columns) go to ONE XCD: workgroup b runs on XCD
    // b % 8 (observed, used for speed only), so within every run of 8 * subs workgroups XCD x takes the subs tiles of the x-th item
    int u = b;
    {
      const int subs = p.item_subs, total = (int)gridDim.x;
      if (subs > 1 && total % (8 * subs) == 0) {
        const int xcd = b & 7, j = b >> 3;
        u = (j / subs) * (8 * subs) + xcd * subs + (j % subs);
      }
    }
    const int item = u / p.item_subs, sub = u - item * p.item_subs;
    tm = sub / p.tiles_n;
    tn = sub - tm * p.tiles_n;
    // (tried, round 6: up to 64 items INSIDE the argument block, so that a workgroup reads its item from the kernarg segment instead of
    // through the list pointer - one dependent load less in front of its first DMA. Slower: 9.9 against 9.3 us on 128 x 1024 x 4096 as
    // 32 tile invokes - the 3.3 KiB argument block costs more to fetch than the load it saves. The grouped kernel stays ~1.1 us
    // behind the same tile as a whole-layer launch, 9.3 against 8.1 us under rocprofv3.)
    typedef const __attribute__((address_space(4))) WorkItem item_c_t;
    item_c_t &w = ((item_c_t *)p.items)[item];
    it_A = w.A;
    it_B = w.B;
    it_C = w.C;
    it_D = w.D;
    it_br = (int)w.br;
  } else if (p.xm > 0) {
    const int xn = 8 / p.xm, xcd = b & 7, j = b >> 3;
    const int lm = p.tiles_m / p.xm, ln = p.tiles_n / xn; // tiles per XCD
    tm = (xcd / xn) * lm + j / ln;
    tn = (xcd % xn) * ln + j % ln;
  } else {
    tm = b / p.tiles_n;
    tn = b - tm * p.tiles_n;
  }
#ifdef TPP_HIP_ABLATION
  // timing experiment (dbg & 512): every workgroup LOADS the panels of tile (0, 0) - 100 % L2 hits after the first touch, no fabric
  // traffic - and stores its own tile: what the K loop costs when no byte comes from beyond the L2
  const bool alias_ = (p.dbg & 512) != 0;
  const int m0 = tm * BM, n0 = tn * BN, m0_ld = alias_ ? 0 : m0, n0_ld = alias_ ? 0 : n0;
  [[maybe_unused]] constexpr int skip_cols = 0; // (timing builds: no ragged items)
#else
  // RAGGED n of grouped items (round 6: the reference's --tiles=64,48,64 rows): an item whose n is not a multiple of the tile's width
  // gets ceil(n / BN) column tiles, the LAST one moved left to end at column n - it recomputes the columns it shares with its
  // neighbour (panels in bounds, no masking of loads) and stores only its own (skip_cols: the first columns of its tile are the
  // neighbour's). Everything else: skip_cols = 0.
  const int m0 = tm * BM;
  const int n0 = (GRP == 1 && (tn + 1) * BN > p.n) ? p.n - BN : tn * BN;
  [[maybe_unused]] const int skip_cols = tn * BN - n0;
  const int m0_ld = m0, n0_ld = n0;
#endif
  const int L = MULTI ? p.nlayers : 1;

  if (wave >= NMW) {
    // ---- loader waves (blw_loader above) ---------------------------------------------------------------------------
    if (wave < NMW + NLA) blw_loader<true, NLA, NLA, NSLOT, SUP, BM, BN, WK, MULTI, 0, GRP>(pp, smem_c, lane, m0_ld, n0_ld, tm, L, wave - NMW, it_A, it_B, it_br, q_da);
    else blw_loader<false, NLB, NLA, NSLOT, SUP, BM, BN, WK, MULTI, FLATB, GRP>(pp, smem_c, lane, m0_ld, n0_ld, tm, L, wave - NMW - NLA, it_A, it_B, it_br, q_db);
    return; // ended waves do not take part in later barriers
  }

  // ---- MFMA waves ------------------------------------------------------------------------------------------------
  const int wk = wave / NOUT, wmn = wave % NOUT, wm = wmn / WN, wn = wmn % WN;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[TM][TN];
  // Fragment buffers. Tiles with one or two accumulators per wave (TM * TN <= 2) keep TWO chunks of fragments: a k-step is only
  // 1-2 MFMAs (32-64 cycles), less than an LDS read takes to come back, so the fragments of chunk t+1 are all read during the
  // second half of chunk t (measured on the 64x64 tile with a two-step lookahead: 420 cycles per chunk for 128 cycles of MFMA).
  // The 128x128 tile (four accumulators, 128 cycles per k-step) reads two k-steps ahead out of one set.
  constexpr bool FULLPF = TM * TN <= 2;
  constexpr int BLW_RD = 3; // SUP = 2: k-steps the fragment reads run ahead of their MFMA
  constexpr int NFB = FULLPF ? 2 * KS : KS;
  static_assert(!FULLPF || NSLOT % 2 == 0, "chunk parity from the ring slot");
  static_assert(SUP == 1 || FULLPF, "two chunks per barrier: the tiles with two fragment sets");
  bf16x8_lw af[NFB][TM];
  u32x4 bw[NFB][TN]; // B fragments as dwords
  int b_lane[TN];   // dword index of this lane's column of tile j in pair-row 4*lh of a k-step
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    b_lane[j] = (4 * lh) * BN + (wn * TN + j) * 32 + li;
    if constexpr (FLATB == 2) {
      // flat image [64 k][BN] (blw_loader FB = 2), fragments by ds_read_b64_tr_b16: in every 16 lanes, lane p names the 8-byte
      // piece (row p / 4, columns 4 (p % 4) ..) of a [4 k][16 n] block and receives COLUMN p of it - four consecutive k of
      // its own column (tools/ubench/tr16_probe.hip pins that). Byte offset of this lane's piece for k = 8 lh + p / 4:
      const int pp16 = lane & 15, blk = wn * TN + j, rowl = pp16 >> 2;
      const int swz = BN == 128 ? rowl : (rowl >> 1) & 1;
      b_lane[j] = (8 * lh + rowl) * (BN * 2) + ((blk ^ swz) << 6) + ((lane >> 4) & 1) * 32 + (pp16 & 3) * 8;
    }
    if constexpr (FLATB == 4) {
      // VNNI-4 image [16 k-groups][BN][4] (blw_loader FB = 4): this lane's column, k-group 2 lh of a k-step (bytes)
      b_lane[j] = (2 * lh) * (BN * 8) + ((wn * TN + j) * 32 + li) * 8;
    }
    asm volatile("" : "+v"(b_lane[j])); // one base VGPR per column tile: rows r, r+1 pair up as ds_read2(st64)_b32
  }
  auto frag_load = [&](int buf, int slot, int ks) __attribute__((always_inline)) {
    const unsigned char *as = smem_c + slot * SLOT;
    const unsigned int *bs = (const unsigned int *)(as + A_SLOT);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = (wm * TM + i) * 32 + li;
      af[buf][i] = *(const bf16x8_lw *)(as + row * 128 + (((2 * ks + lh) ^ ((row >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (FLATB == 2) {
        typedef short s16x4_lw __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4_lw lds_s16x4_lw;
        typedef __attribute__((address_space(3))) unsigned char lds_u8_lw;
        lds_u8_lw *bb = (lds_u8_lw *)(as + A_SLOT) + b_lane[j] + (16 * ks) * (BN * 2);
        const u32x2_lw q0 = __builtin_bit_cast(u32x2_lw, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_lw *)bb));
        const u32x2_lw q1 = __builtin_bit_cast(u32x2_lw, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_lw *)(bb + 4 * (BN * 2))));
        bw[buf][j] = u32x4{q0[0], q0[1], q1[0], q1[1]};
      } else if constexpr (FLATB == 4) {
        // four consecutive k of a column are 8 contiguous bytes: k-groups 4 ks + 2 lh and + 1 = the lane's eight k of this step
        const unsigned char *bb = as + A_SLOT + b_lane[j] + (4 * ks) * (BN * 8);
        const u32x2_lw q0 = *(const u32x2_lw *)bb, q1 = *(const u32x2_lw *)(bb + BN * 8);
        bw[buf][j] = u32x4{q0[0], q0[1], q1[0], q1[1]};
      } else {
        const unsigned int *bp = bs + b_lane[j] + (8 * ks) * BN;
#pragma unroll
        for (int r = 0; r < 4; ++r) bw[buf][j][r] = bp[r * BN];
      }
    }
  };
#ifdef TPP_HIP_ABLATION
  const int dbg = p.dbg;
#else
  constexpr int dbg = 0; // the timing switches exist in ablation builds only (chain_args.h)
#endif
  // Timing experiment "no fragment reads / MFMAs" (dbg & 32) is a COMPILE-TIME switch (-DTPP_BLW_SKIP_MATH): as a run-time branch
  // around every group of reads and MFMAs it split the chunk body into basic blocks, the compiler's wait-count insertion lost track
  // of which LDS reads were outstanding across them and put s_waitcnt lgkmcnt(0..2) in front of the MFMAs - each k-step waited for
  // the fragment reads issued just before it (274 cycles per k-step of 128 cycles of MFMA on the 128x128 tile).
#ifdef TPP_BLW_SKIP_MATH
  const bool skip_math = (dbg & 32) != 0;
#else
  constexpr bool skip_math = false;
  (void)dbg;
#endif
  // two more compile-time timing switches (side builds only, results WRONG by design): -DTPP_BLW_SKIP_READS = the MFMAs run on
  // whatever the fragment registers hold (no LDS reads in the K loop), -DTPP_BLW_SKIP_BARRIER = no mid-chunk barrier in any wave
  // (with TPP_HIP_CHAIN_DBG=16, no DMA: what the MFMA side alone costs per chunk, and what of it is the barrier / the reads)
#ifdef TPP_BLW_SKIP_READS
  constexpr bool skip_reads = true;
#else
  constexpr bool skip_reads = false;
#endif
  // One chunk in ring slot `slot` (a run-time value: ONE body - two for the tiles that alternate fragment sets - instead of one per
  // ring slot entered through a switch; the slot's LDS offset costs TM + TN vector adds per k-step. The per-slot bodies with their
  // exit after every chunk made hipcc rename the accumulators from body to body - v_mfma D != C - in the chain and flat-B
  // instances, which then spilled 150 .. 1500 bytes per lane). Step q multiplies fragment buffer q while the fragments of step
  // q + PD are read (the last PD steps read the first steps of chunk t+1, published by the mid-chunk barrier; after the last chunk
  // of a layer they read a slot nobody uses - the values are dropped).
  // t, T: this chunk's index and the layer's chunk count. The workgroup barrier sits in the middle of every SUP-th chunk (slot a
  // multiple of SUP: layers start at such a slot) when another barrier interval follows: it publishes the next SUP chunks and
  // retires the previous SUP slots. (With SUP = 2 the fragments of the odd chunk are read during the even chunk's second half - the
  // same interval, already published - and those of the next even chunk during the odd chunk's, behind the barrier.)
  // PAR (tiles with two fragment sets): parity of the slot = which set holds this chunk's fragments.
  // HN: does this chunk carry the barrier? 1 yes / 0 no as compile-time facts (the steady-state loop bodies are then single basic
  // blocks: a conditional barrier - or the loop's exit test - in the middle of a body is a block boundary at which the compiler
  // waits for every LDS read in flight), 2 = decided at run time (the odd first chunk).
  auto chunk = [&](auto par_c, auto hn_c, int slot, int t, int T) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value, HN = decltype(hn_c)::value;
    const int ns = slot + 1 == NSLOT ? 0 : slot + 1;
    const bool has_next = HN == 2 ? (SUP == 1 || PAR == 0) && t + SUP < T : HN == 1;
    constexpr int CUR = FULLPF ? PAR * KS : 0, NXT = FULLPF ? (PAR ^ 1) * KS : 0; // fragment sets of chunk t / t+1
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      if (!skip_math && !skip_reads) {
        if constexpr (FULLPF && SUP == 2) {
          // two chunks per barrier: everything up to the end of chunk t+1 was published before chunk t began (the odd chunk with
          // its pair, the next even one by the barrier in the middle of the even chunk before it) and the step after the barrier
          // may read one chunk further. So the fragments are read ONE k-step per step, BLW_RD steps ahead of their MFMA, through
          // the 2 * KS buffers as a ring: the LDS pipe works all the time instead of in the second halves of the chunks, and the
          // wait in front of an MFMA is for reads issued three MFMAs earlier (3 * BLW_RD + 3 reads in flight: lgkmcnt has 4 bits).
          const int tg = q + BLW_RD, ch = tg / KS;
          const int nns = ns + 1 == NSLOT ? 0 : ns + 1;
          frag_load((PAR * KS + tg) % NFB, ch == 0 ? slot : ch == 1 ? ns : nns, wk * KS + tg % KS);
        } else if constexpr (FULLPF) {
          // second half of the chunk (chunk t+1 is published): two of its k-steps per step
          if (q >= KS / 2) {
            const int r = 2 * (q - KS / 2);
            frag_load(NXT + r, ns, wk * KS + r);
            if (r + 1 < KS) frag_load(NXT + r + 1, ns, wk * KS + r + 1);
          }
        } else {
          if (q + PD < KS) frag_load(q + PD, slot, wk * KS + q + PD);
          else frag_load(q + PD - KS, ns, wk * KS + q + PD - KS);
        }
      }
      // The step's fragment reads (for MFMAs two steps / half a chunk ahead) are dealt BETWEEN the step's MFMAs, not in front of them:
      // a 32x32x16 bf16 MFMA holds the matrix pipe for 32 cycles, six DS instructions take the wave longer than that to issue - in
      // front of the MFMAs the pipe ran dry once per k-step (same-box A/B, profiles/r04_bf16_lw_read_interleave.txt: the 4096-row
      // layer 10.02 -> 9.45 us, its chain 30.5 -> 29.4, C5 17.6 -> 17.0, 4096^3 on this tile 128.2 -> 122.3).
      constexpr bool IL = TPP_BLW_INTERLEAVE >= 1 && (!FULLPF || (TPP_BLW_INTERLEAVE >= 2 && SUP == 1 && TM * TN >= 2));
      if constexpr (!IL) __builtin_amdgcn_sched_barrier(0);
      if (!skip_math) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_lw, bw[CUR + q][j]), af[CUR + q][i], acc[i][j], 0, 0, 0);
      }
      if constexpr (IL && !FULLPF) {
        constexpr int NRD = TM + (FLATB == 4 ? TN : 2 * TN); // DS read instructions of a step (VNNI-4: one ds_read2_b64 per column tile)
        constexpr int PER = (NRD + TM * TN - 2) / (TM * TN - 1);
#pragma unroll
        for (int g_ = 0; g_ < TM * TN - 1; ++g_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, PER, 0); // then a share of the DS reads
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      } else if constexpr (IL) {
        // two fragment sets (64x128): the second half of a chunk reads TWO k-steps of the next chunk per step, behind every MFMA a share
        constexpr int NRD = 2 * (TM + (FLATB == 4 ? TN : 2 * TN)), PER = (NRD + TM * TN - 1) / (TM * TN);
#pragma unroll
        for (int g_ = 0; g_ < TM * TN; ++g_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, PER, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#ifndef TPP_BLW_SKIP_BARRIER
      if (q == KS / 2 - 1 && has_next) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  int s0 = 0; // ring slot of the current layer's chunk 0
  for (int l = 0; l < L; ++l) {
    const auto &Y = p.L[l];
    const int T = (GRP ? it_br : Y.br) * (Y.k / BLW_BK);
    const int ep = Y.ep;
    unsigned short *__restrict__ C = (unsigned short *)(GRP ? it_C : Y.C);
    const unsigned ldcb = (unsigned)((int)Y.ldc * 2);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // bias: fetched here (8 bytes = 4 columns per register quad), used in the epilogue - its latency hides under the K loop
    u32x2_lw biasw[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        biasw[j][g] = u32x2_lw{0u, 0u};
        if (ep & EP_BIAS) biasw[j][g] = *(const u32x2_lw *)((const unsigned short *)(GRP ? it_D : Y.D) + n0 + ((GRP == 2 ? 0 : wn * TN) + j) * 32 + 8 * g + 4 * lh);
      }
    if (MULTI && wave == 0) blw_stamp(p, l, 0, lane);
    if (MULTI && NLA > 1 && l > 0) __builtin_amdgcn_s_barrier(); // S2 (the loaders' rendezvous after the seam wait)
    __builtin_amdgcn_s_barrier(); // P: chunk 0 published
    __builtin_amdgcn_sched_barrier(0);
    if (MULTI && wave == 0) blw_stamp(p, l, 1, lane);
    // (T >= 1: the launchers send empty batches to the generic kernel - a branch around this loop costs the 128-wide tiles
    // a second copy of the accumulators and 250 spilled registers)
    // the layers follow each other through the ring: this one starts at slot s0 (the loaders count the same way)
    const int last_slot = (s0 + T - 1) % NSLOT;
    if constexpr (FULLPF && SUP == 2) { // (s0 even; the first barrier interval = chunks 0 and 1 is published)
#pragma unroll
      for (int s = 0; s < BLW_RD; ++s) frag_load(s, s0 + s / KS, wk * KS + s % KS);
    } else if (!FULLPF || !(s0 & 1)) {
#pragma unroll
      for (int s = 0; s < (FULLPF ? KS : PD); ++s) frag_load(s, s0, wk * KS + s);
    } else { // (chunk parity = slot parity selects the fragment set)
#pragma unroll
      for (int s = 0; s < KS; ++s) frag_load(FULLPF ? KS + s : s, s0, wk * KS + s);
    }
    {
      int t = 0, slot = s0;
      auto next = [&](int x) __attribute__((always_inline)) { return x + 1 == NSLOT ? 0 : x + 1; };
      using Y = std::integral_constant<int, 1>;
      using N = std::integral_constant<int, 0>;
      using R = std::integral_constant<int, 2>;
      if constexpr (!FULLPF) { // (SUP = 1) every chunk but the last carries the barrier; T >= 1
        if constexpr (NSLOT == 4 || NSLOT == 5) {
          // whole laps of the ring with LITERAL slots (LDS offsets are immediates: no address adds, and the four bodies of a lap
          // are one basic block): 4096^3 on the 128x128 tile 139 -> 131 us, 2048^3 on a flat B 20.6 -> 19.05, the 4096-row chain
          // 34 -> 33 us (same-box A/B). A single layer starts
          // at slot 0; a layer of a chain first walks to slot 0.
          if constexpr (MULTI) {
            for (; slot != 0 && t + 1 < T; ++t) {
              chunk(P0{}, Y{}, slot, t, T);
              slot = next(slot);
            }
          }
          for (; t + NSLOT < T; t += NSLOT) {
#pragma unroll
            for (int s_ = 0; s_ < NSLOT; ++s_) chunk(P0{}, Y{}, s_, t + s_, T);
          }
        }
        for (; t + 1 < T; ++t) {
          chunk(P0{}, Y{}, slot, t, T);
          slot = next(slot);
        }
        chunk(P0{}, N{}, slot, t, T);
      } else if constexpr (SUP == 2) { // (T even, s0 even) the barrier sits in the even chunk of every pair but the last
        for (; t + 2 < T; t += 2) {
          chunk(P0{}, Y{}, slot, t, T);
          slot = next(slot);
          chunk(P1{}, N{}, slot, t + 1, T);
          slot = next(slot);
        }
        chunk(P0{}, N{}, slot, t, T);
        chunk(P1{}, N{}, next(slot), t + 1, T);
      } else {
        if (slot & 1) { // (an odd first slot: after a layer with an odd chunk count)
          chunk(P1{}, R{}, slot, t, T);
          slot = next(slot);
          ++t;
        }
        for (; t + 2 < T; t += 2) {
          chunk(P0{}, Y{}, slot, t, T);
          slot = next(slot);
          chunk(P1{}, Y{}, slot, t + 1, T);
          slot = next(slot);
        }
        if (t + 1 < T) {
          chunk(P0{}, Y{}, slot, t, T);
          chunk(P1{}, N{}, next(slot), t + 1, T);
        } else if (t < T) {
          chunk(P0{}, N{}, slot, t, T);
        }
      }
    }
    s0 = (s0 + T) % NSLOT;
    // the fragments prefetched past the end of the layer are dead: without this the compiler sinks their reads
#pragma unroll
    for (int s = 0; s < (FULLPF ? NFB : PD); ++s) {
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(af[s][i]));
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bw[s][j]));
    }

    if (MULTI && wave == 0) blw_stamp(p, l, 2, lane);
    // ---- epilogue ------------------------------------------------------------------------------------------------
    if constexpr (WK > 1) {
      // K group 1 parks its 32x32 partial, group 0 adds it (group order: 0 + 1) and finishes the tile
      float *red = (float *)(smem_c + OFF_RED) + wmn * 1024 + lane;
      if (wk == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[r * 64] = acc[0][0][r];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); // R1
      if (wk == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += red[r * 64];
      }
    }
    if (WK == 1 || wk == 0) {
      // lane (li, lh) owns row 32*i + li of the wave's row block i and, in registers 4g..4g+3 of tile (i, j),
      // columns 32*j + 8*g + 4*lh + (0..3)
      const __amdgpu_buffer_rsrc_t rsrcC = __builtin_amdgcn_make_buffer_rsrc(
          (void *)(GRP == 2 ? C : C + (int64_t)(m0 + wm * 32 * TM) * Y.ldc + n0 + wn * 32 * TN), 0, 0x7fffffff, 0x00020000);
      if constexpr (!MULTI) {
        if (!(ep & EP_BETA0)) { // beta = 1: add C before the single rounding (8-byte loads, rare path)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const u32x2_lw c2 = __builtin_amdgcn_raw_buffer_load_b64(
                    rsrcC, (unsigned)(32 * i + li) * ldcb + (unsigned)((32 * j + 8 * g + 4 * lh) * 2), 0, 0);
                acc[i][j][4 * g + 0] += __uint_as_float(c2[0] << 16);
                acc[i][j][4 * g + 1] += __uint_as_float(c2[0] & 0xffff0000u);
                acc[i][j][4 * g + 2] += __uint_as_float(c2[1] << 16);
                acc[i][j][4 * g + 3] += __uint_as_float(c2[1] & 0xffff0000u);
              }
        }
      }
      const bool relu = (ep & EP_RELU) != 0;
      unsigned char *ot = smem_c + ((last_slot + 1 + wmn / WPS) % NSLOT) * SLOT + (wmn % WPS) * STAGE_W;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float bias[4][4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const u32x2_lw b2 = biasw[j][g];
            bias[g][0] = __uint_as_float(b2[0] << 16);
            bias[g][1] = __uint_as_float(b2[0] & 0xffff0000u);
            bias[g][2] = __uint_as_float(b2[1] << 16);
            bias[g][3] = __uint_as_float(b2[1] & 0xffff0000u);
          }
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
            *(u32x4 *)(ot + li * ES + (32 * j + 8 * g + 8 * lh) * 2) = out;
          }
        }
        // the same wave reads its 32-row block back row-contiguously: 4*TN lanes x 16 B = one row
        constexpr int LPR = 4 * TN, RPP = 64 / LPR; // lanes per row, rows per pass
#pragma unroll
        for (int it = 0; it < 32 / RPP; ++it) {
          const int row = it * RPP + lane / LPR, ch = lane % LPR;
          const u32x4 v = *(const u32x4 *)(ot + row * ES + ch * 16);
          const unsigned voff = (unsigned)(32 * i + row) * ldcb + (unsigned)(ch * 16);
          if (GRP == 1 && ch * 8 < skip_cols) continue; // (a ragged item's last column tile: these columns are the neighbour tile's)
          if (MULTI && l + 1 < L && !(dbg & 4)) __builtin_amdgcn_raw_buffer_store_b128(v, rsrcC, voff, 0, 16); // sc1: write-through (hand-off)
          else __builtin_amdgcn_raw_buffer_store_b128(v, rsrcC, voff, 0, C_STORE_AUX); // (last layer / single layer: gemm_common.h)
        }
      }
    }
    if (MULTI && wave == 0) blw_stamp(p, l, 3, lane);
    if (l + 1 == L) break;
    if constexpr (MULTI) {
      // ---- seam: publish this tile to the row block's consumers ------------------------------------------------
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // EVERY storing wave drains its write-through stores
      __builtin_amdgcn_s_barrier();                     // S1
      if (wave == 0 && lane == 0)
        __hip_atomic_fetch_add((g_u32_lw *)(p.cnt + ((size_t)l * p.tiles_m + tm) * CHAIN_CNT_STRIDE), 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      if (wave == 0) blw_stamp(p, l, 4, lane);
    }
  }
}

template <int WM, int WN, int WK, int TM, int TN, int NSLOT, int NLA, int NLB, int SUP, bool MULTI, int FLATB = 0, int GRP = 0>
static hipError_t launch_blw_t(const ChainArgs &a, hipStream_t s, const void *items = nullptr, int n_items = 0) {
  constexpr int NOUT = WM * WN, BM = 32 * WM * TM, BN = 32 * WN * TN, NT = 64 * (WM * WN * WK + NLA + NLB);
  constexpr size_t lds = (size_t)NSLOT * (BM + BN) * 128 + (WK > 1 ? (size_t)NOUT * 4096 : 0);
#ifdef TPP_HIP_ABLATION
  constexpr size_t lds_alloc = lds + (lds + 4096 <= 160 * 1024 ? 4096 : 0); // room for the loaders' per-chunk stamps (BlwChunkStamps)
#else
  constexpr size_t lds_alloc = lds;
#endif
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = brgemm_bf16_lw<WM, WN, WK, TM, TN, NSLOT, NLA, NLB, SUP, MULTI, FLATB, GRP>;
  static std::atomic<unsigned long long> lds_set{0};
  if (hipError_t e = ensure_dynamic_lds((const void *)kern, (int)lds_alloc, lds_set); e != hipSuccess) return e;
  ChainArgs args = a;
  args.tiles_m = a.m / BM;
  args.tiles_n = GRP == 1 ? (a.n + BN - 1) / BN : a.n / BN; // (grouped items: a ragged last column tile, see the kernel)
  long long tiles = (long long)args.tiles_m * args.tiles_n;
  if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
  args.items = nullptr;
  args.item_subs = 0;
  args.pad_items = 0;
  if constexpr (GRP) { // tiles_m x tiles_n workgroups per item (consecutive workgroups = the items in list order: neighbouring
                       // tiles of a layer share panels in the L2 of whichever XCD they land on)
    if (!items || n_items <= 0 || tiles * n_items > 0x7fffffffLL) return hipErrorInvalidValue;
    args.items = items;
    args.item_subs = (int)tiles;
    args.xm = 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * n_items)), dim3(NT), lds_alloc, s, args);
    return hipGetLastError();
  }
  // XCD grid xm x (8 / xm) over the tile grid: minimise xn * m + xm * n (bytes of A and W all eight L2s fetch, in units of 2K)
  args.xm = 0;
  static const int forced_xm = [] {
    const char *e = getenv("TPP_HIP_BF16_LW_XM"); // A/B runs: 1, 2, 4, 8, or 0 = linear mapping
    return e ? atoi(e) : -1;
  }();
  long long best = -1;
  for (int xm = 8; xm >= 1; xm >>= 1) {
    const int xn = 8 / xm;
    if (args.tiles_m % xm || args.tiles_n % xn) continue;
    const long long cost = (long long)xn * a.m + (long long)xm * a.n;
    if (forced_xm >= 0 ? xm == forced_xm : (best < 0 || cost < best)) {
      best = cost;
      args.xm = xm;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(NT), lds_alloc, s, args);
  return hipGetLastError();
}

// tile: 0 = 32x64 (K split over two wave groups), 1 = 64x64, 2 = 64x128, 3 = 128x128
void blw_tile_dims(int tile, int *bm, int *bn) {
  static const int BMs[4] = {32, 64, 64, 128}, BNs[4] = {64, 64, 128, 128};
  *bm = BMs[tile & 3];
  *bn = BNs[tile & 3];
}

// Loader waves per tile (NLA + NLB), same-box A/B (profiles/r03_blw_loader_split.txt): 32x64 1 + 2 (1 + 1: +2 %, 2 + 2: +5 %),
// 64x64 1 + 1 (1 + 2: +2 %, 2 + 2: +7 %), 64x128 1 + 2 (1 + 1: same, 2 + 4: +5 %), 128x128 1 + 1 (2 + 2, 1 + 2: same).
// SUP = 2 (one workgroup barrier per TWO chunks) for the 32x64 and 64x64 tiles when every layer has an even chunk count
// (TPP_HIP_BLW_SUP=1 forces one chunk per barrier for A/B runs). Ring depths: 8 / 8 / 6 / 4 slots; a 5-slot ring with 2 + 2
// loaders for the 128x128 tile and 4 against 6 slots for 64x128 measured the same (same box, +-0.5 %); FOUR chunks per barrier on
// a 12-slot ring for the 32x64 tile measured 9 % slower than two on 8 slots (the prologue must request 8 chunks before the first barrier).
// side builds only (-DTPP_HIP_ABLATION): TPP_HIP_BLW_T3 = 1: the 128x128 tile with a 5-slot ring and two loader waves per panel,
// 2: 4 slots, two loader waves per panel (timing experiments on the fill loop; the product has ONE instance of the tile)
#ifdef TPP_HIP_ABLATION
static int blw_t3_alt() {
  static const int v = [] {
    const char *e = getenv("TPP_HIP_BLW_T3");
    return e ? atoi(e) : 0;
  }();
  return v;
}
#define BLW_T3_ALTS(MULTI, FB)                                                                 \
  if (blw_t3_alt() == 1) return launch_blw_t<2, 2, 1, 2, 2, 5, 2, 2, 1, MULTI, FB>(a, s);      \
  if (blw_t3_alt() == 2) return launch_blw_t<2, 2, 1, 2, 2, 4, 2, 2, 1, MULTI, FB>(a, s);
#else
#define BLW_T3_ALTS(MULTI, FB)
#endif
#define BLW_DISPATCH(MULTI, FB)                                                              \
  switch (tile * 2 + (sup2 ? 1 : 0)) {                                                       \
  case 0: return launch_blw_t<1, 2, 2, 1, 1, 8, 1, 2, 1, MULTI, FB>(a, s);                   \
  case 1: return launch_blw_t<1, 2, 2, 1, 1, 8, 1, 2, 2, MULTI, FB>(a, s);                   \
  case 2: return launch_blw_t<2, 2, 1, 1, 1, 8, 1, 1, 1, MULTI, FB>(a, s);                   \
  case 3: return launch_blw_t<2, 2, 1, 1, 1, 8, 1, 1, 2, MULTI, FB>(a, s);                   \
  case 4:                                                                                    \
  case 5: return launch_blw_t<2, 2, 1, 1, 2, 6, 1, 2, 1, MULTI, FB>(a, s);                   \
  case 6:                                                                                    \
  case 7: BLW_T3_ALTS(MULTI, FB) return launch_blw_t<2, 2, 1, 2, 2, 4, 1, 1, 1, MULTI, FB>(a, s); \
  default: return hipErrorInvalidValue;                                                      \
  }
static bool blw_sup2(const ChainArgs &a) {
  static const int forced = [] {
    const char *e = getenv("TPP_HIP_BLW_SUP");
    return e ? atoi(e) : 0;
  }();
  if (forced == 1) return false;
  for (int l = 0; l < a.nlayers; ++l)
    if ((a.L[l].br * (a.L[l].k / BLW_BK)) & 1) return false;
  return true;
}

// one layer (a.nlayers == 1): any chunk stream of at least one chunk, both accumulator starts
// tile 4 (round 5, VNNI-2 only, single layers only): 32x32 + K2 - two MFMA waves, one loader wave per panel. For SMALL outputs with a
// LONG reduction (the reference's M = 128 / 256 shapes at K >= 1536: launch_gemm routes here): twice the workgroups of the 32x64 tile,
// each streaming 8 KiB per chunk - the layer is bound by how many CUs pull panels, not by the matrix pipes
hipError_t launch_bf16_lw(int tile, const ChainArgs &a, hipStream_t s) {
  if (a.L[0].br < 1 || a.L[0].k < BLW_BK || tile < 0 || tile > 4) return hipErrorInvalidValue;
  const bool sup2 = blw_sup2(a);
  if (tile == 4) return sup2 ? launch_blw_t<1, 1, 2, 1, 1, 8, 1, 1, 2, false, 0>(a, s) : launch_blw_t<1, 1, 2, 1, 1, 8, 1, 1, 1, false, 0>(a, s);
  BLW_DISPATCH(false, 0)
}

// One layer whose B operand is FLAT ([k][ldb] bf16, no VNNI flag on the dispatch - what xsmm.unary pack would have turned into
// VNNI-2, lib/TPP/Transforms/Utils/VNNIUtils.cpp:75-77): the VNNI kernels' configurations with the B image and the fragment
// reads swapped (FLATB = 2): the chunk's 64 rows go into LDS as they are (LDS-DMA, 64-byte blocks swizzled on the source side)
// and a B fragment is two ds_read_b64_tr_b16 - the hardware transpose hands every lane four consecutive k of its own column.
// No pack launch, no VALU, and the reads run at the LDS's full 256 B/clk where the VNNI image needs 4-byte reads (128 B/clk):
// bit-identical to pack + VNNI kernel and as fast or faster on every tile (profiles/r03_flat_b_bf16.txt). A first version
// interleaved the pair-rows in a register-staging B loader (buffer_load_dwordx4 x 2 -> 8 v_perm_b32 -> 2 ds_write_b128): correct,
// 15-28 % slower (the ds_write path: ~79 B/clk and 2-way conflicts), removed.
hipError_t launch_bf16_lw_flatb(int tile, const ChainArgs &a, hipStream_t s) {
  if (a.L[0].br < 1 || a.L[0].k < BLW_BK || tile < 0 || tile > 3) return hipErrorInvalidValue;
  const bool sup2 = blw_sup2(a);
  BLW_DISPATCH(false, 2)
}

// One layer whose B operand is VNNI-4 ([k/4][ldb][4] bf16: the dispatch carries the VNNI flag, the factor is the runtime's setting,
// xsmm_hip_set_vnni_factor - lib/TPP/Transforms/Utils/VNNIUtils.cpp:25-45, benchmarks/config/omp/mlir-bf16.json:68-100): the same
// configurations with the B image and the fragment reads of FLATB = 4 - the chunk's 16 k-group rows go into the LDS as they are and a
// fragment is two 8-byte reads (256 B/clk against the 128 B/clk of the VNNI-2 image's 4-byte reads).
hipError_t launch_bf16_lw_vnni4(int tile, const ChainArgs &a, hipStream_t s) {
  if (a.L[0].br < 1 || a.L[0].k < BLW_BK || tile < 0 || tile > 4) return hipErrorInvalidValue;
  const bool sup2 = blw_sup2(a);
  if (tile == 4) return sup2 ? launch_blw_t<1, 1, 2, 1, 1, 8, 1, 1, 2, false, 4>(a, s) : launch_blw_t<1, 1, 2, 1, 1, 8, 1, 1, 1, false, 4>(a, s); // 32x32 + K2 (round 6)
  BLW_DISPATCH(false, 4)
}

// Tile invokes of ONE bf16 descriptor as one launch on the loader-wave tiles (round 6, VERDICT r5 missing 3 / next 2b: the tile queue's
// bf16 groups used to run on brgemm_bf16_small32 / brgemm_bf16_fast<64x64> and were 0-45 % slower than the same layer as one whole-
// layer call). One workgroup per BM x BN tile of an item, operands and batch count from the item (ChainArgs::items); packed tile
// blocks are just another (lda, stride) pattern to the loaders. SUP = 2 instances only when every item has an even chunk count.
hipError_t launch_bf16_lw_grouped(int tile, int b_kind, const ChainArgs &a, const void *items, int n_items, bool even_chunks, hipStream_t s) {
  if (a.L[0].k < BLW_BK || a.L[0].k % BLW_BK || tile < 0 || (tile > 1 && tile != 4) || (b_kind != 0 && b_kind != 4)) return hipErrorInvalidValue;
  if (tile == 4) { // 32x32 + K2 (launch_bf16_lw's tile 4): four workgroups per 64x64 item - skinny groups with a long reduction
    static const int forced4 = [] {
      const char *e = getenv("TPP_HIP_BLW_SUP");
      return e ? atoi(e) : 0;
    }();
    if (b_kind == 4) // (VNNI-4 operands: grouped form only - the image rows of a 32-column tile are 256 bytes, four per DMA instruction)
      return even_chunks && forced4 != 1 ? launch_blw_t<1, 1, 2, 1, 1, 8, 1, 1, 2, false, 4, true>(a, s, items, n_items)
                                         : launch_blw_t<1, 1, 2, 1, 1, 8, 1, 1, 1, false, 4, true>(a, s, items, n_items);
    return even_chunks && forced4 != 1 ? launch_blw_t<1, 1, 2, 1, 1, 8, 1, 1, 2, false, 0, true>(a, s, items, n_items)
                                       : launch_blw_t<1, 1, 2, 1, 1, 8, 1, 1, 1, false, 0, true>(a, s, items, n_items);
  }
  static const int forced = [] {
    const char *e = getenv("TPP_HIP_BLW_SUP");
    return e ? atoi(e) : 0;
  }();
  const bool sup2 = even_chunks && forced != 1;
  if (b_kind == 4) {
    if (tile == 0) return sup2 ? launch_blw_t<1, 2, 2, 1, 1, 8, 1, 2, 2, false, 4, true>(a, s, items, n_items) : launch_blw_t<1, 2, 2, 1, 1, 8, 1, 2, 1, false, 4, true>(a, s, items, n_items);
    return sup2 ? launch_blw_t<2, 2, 1, 1, 1, 8, 1, 1, 2, false, 4, true>(a, s, items, n_items) : launch_blw_t<2, 2, 1, 1, 1, 8, 1, 1, 1, false, 4, true>(a, s, items, n_items);
  }
  if (tile == 0) return sup2 ? launch_blw_t<1, 2, 2, 1, 1, 8, 1, 2, 2, false, 0, true>(a, s, items, n_items) : launch_blw_t<1, 2, 2, 1, 1, 8, 1, 2, 1, false, 0, true>(a, s, items, n_items);
  return sup2 ? launch_blw_t<2, 2, 1, 1, 1, 8, 1, 1, 2, false, 0, true>(a, s, items, n_items) : launch_blw_t<2, 2, 1, 1, 1, 8, 1, 1, 1, false, 0, true>(a, s, items, n_items);
}

// QUADS (round 6): a group of 64x64 items that forms an R x C grid of item rows (A blocks) and item columns (B blocks), R and C even,
// as 2 x 2 blocks on the 128x128 tile - what the same layer runs on as one whole-layer call when it is large (1024 x 2560 x 1024:
// 640 items on 64x64 tiles 15.3 us, 160 workgroups of 128x128 10.2). a.m = a.n = 128 (one tile per quad), lda / ldb / ldc / strides /
// k the items'; quads = QuadItem[n_quads] in device-visible memory.
hipError_t launch_bf16_lw_quads(int b_kind, const ChainArgs &a, const void *quads, int n_quads, hipStream_t s) {
  if (a.L[0].k < BLW_BK || a.L[0].k % BLW_BK || a.m != 128 || a.n != 128 || (b_kind != 0 && b_kind != 4)) return hipErrorInvalidValue;
  if (b_kind == 4) return launch_blw_t<2, 2, 1, 2, 2, 4, 1, 1, 1, false, 4, 2>(a, s, quads, n_quads);
  return launch_blw_t<2, 2, 1, 2, 2, 4, 1, 1, 1, false, 0, 2>(a, s, quads, n_quads);
}

// a chain of layers in one launch; the caller guarantees co-residency (tiles <= CUs), beta = 0, disjoint buffers and ONE kind of B
// operand for every layer (b_kind 0: VNNI-2, 2: flat, 4: VNNI-4)
hipError_t launch_bf16_chain(int tile, int b_kind, const ChainArgs &a, hipStream_t s) {
  if (tile < 0 || tile > 3) return hipErrorInvalidValue;
  const bool sup2 = blw_sup2(a);
  if (b_kind == 2) {
    BLW_DISPATCH(true, 2)
  } else if (b_kind == 4) {
    BLW_DISPATCH(true, 4)
  }
  BLW_DISPATCH(true, 0)
}

} // namespace tpp

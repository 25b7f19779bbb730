// xsmm_desc.h - immutable kernel descriptors behind the i64 handles of the C-ABI,
// and the launch interface between the ABI layer (runtime.cpp) and the gfx950
// kernels (*.hip). Not part of the public ABI (that is include/tpp_xsmm_abi.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tpp {

enum : int { KIND_GEMM = 0x47454d4d /*'GEMM'*/, KIND_UNARY = 0x554e4152, KIND_BINARY = 0x42494e41,
             KIND_AMX = 0x414d5843 };
enum : int64_t { DT_F32 = 1, DT_BF16 = 2 };
enum : int { EP_BETA0 = 1, EP_BIAS = 2, EP_RELU = 4, EP_VNNI_C = 8 }; // epilogue bits of the kernel argument blocks

// One descriptor type for gemm / brgemm / fused_brgemm (a gemm is a brgemm with
// one batch and no strides; a brgemm is a fused_brgemm with no epilogue).
struct GemmDesc {
  int kind;             // KIND_GEMM
  int has_batch;        // 0: dispatched through xsmm_gemm_dispatch
  int fused;            // dispatched through xsmm_fused_brgemm_dispatch (invoke takes D)
  int64_t dtype, m, n, k, lda, ldb, ldc, stride_a, stride_b;
  int64_t wire_flags;   // as received (BETA_0 = 4, VNNI_B wire = 2048, ...)
  int beta0, vnni_b, bias, relu;
  int vnni_c;           // C stored / read as VNNI-2 [m/2][n][2] (wire flag 8192): generic kernel only
  int vnni_factor;      // blocking factor v of a VNNI B operand [k/v][n][v]: 2, or 4 (xsmm_hip_set_vnni_factor at dispatch time; the
                        // factor is not on the wire - the reference asks libxsmm_cpuid_dot_pack_factor, VNNIUtils.cpp:25-45)
  int variant;          // kernel variant chosen at dispatch (see gemm_variants.h), -1 = by invoke
  int generic_forced;   // variant = generic because it was asked for (xsmm_hip_force_variant / a VNNI C store), not because no fast tile fits
  int variant_forced;   // variant is the one xsmm_hip_force_variant asked for: invoke-time refinements (batch-count dependent) leave it alone
  int b_trans;          // runtime-made sibling of a dispatched descriptor (never on the wire): B is read TRANSPOSED, B[k][j] = ptr[j * ldb + k] -
                        // the source of an xsmm.unary transpose that fed this gemm's B operand (runtime.cpp, deferred transposes); generic kernel only
  char name[64];        // kernel name for profiles
  char trace[160];      // dispatch tuple + kernel name as text (trace ranges)
};

struct UnaryDesc {
  int kind;             // KIND_UNARY
  int64_t op, dtype, m, n, ldi, ldo, flags;
  char trace[96];       // dispatch tuple as text (trace ranges)
};

struct BinaryDesc {
  int kind;             // KIND_BINARY
  int64_t op, dtype, m, n, ldi_lhs, ldi_rhs, ldo, flags;
  char trace[96];
};

struct AmxDesc {
  int kind;             // KIND_AMX (no-op on this hardware)
};

// One queued invoke (tile queue, runtime.cpp): operands with element offsets applied. GEMM family:
// A, B, C, D + batch count; unary: A = in, C = out; binary: A = lhs, B = rhs, C = out. The array is
// host-pinned and read by the grouped kernels directly.
struct WorkItem {
  const void *A;
  const void *B;
  void *C;
  const void *D;
  int64_t br;
};

// A 2 x 2 block of queued 64x64 bf16 invokes (tile queue, rt_rewrites.h detect_quads): item rows r0 < r1 (A blocks), item columns
// c0 < c1 (B blocks). A / B: the (r0, c0) item's; da = (byte distance of r1's A block from r0's) - 64 rows of lda, db = byte distance
// of c1's B block from c0's; C[2 r + c], D[c]: the four outputs and the two bias pieces. brgemm_bf16_lw's 128x128 tile (GRP = 2).
struct QuadItem {
  const void *A;
  const void *B;
  void *C[4];
  const void *D[2];
  int64_t br;
  uint32_t da, db;
};

// ---- kernel launchers (all enqueue on `stream`, never synchronise) --------------
// pointers are device pointers with element offsets already applied.
hipError_t launch_gemm(const GemmDesc &d, const void *A, const void *B, void *C, const void *D,
                       int64_t br, hipStream_t stream);
// n_items invokes of ONE descriptor in one launch (items: device array of WorkItem)
// vec_ok: every item's A and B are 16-byte aligned; out_ok: every item's C is 16-byte and D 8-byte aligned;
// pair_ok: every item's batch count is even (32-k tiles on the loader-wave kernels)
// br_hint: the batch count of the first item - only a hint for how many workgroups share a tile's batch-reduce range (split
// launches of skinny groups); the kernels take every item's own count from the list
hipError_t launch_gemm_grouped(const GemmDesc &d, const WorkItem *items, int n_items, bool vec_ok, bool out_ok, bool pair_ok,
                               int64_t br_hint, hipStream_t stream);
int force_gemm_split(int workgroups_per_tile); // xsmm_hip_force_split (brgemm_f32.hip); returns the previous setting
// STRICT mode (round 6; xsmm_hip_set_strict / TPP_HIP_STRICT=1): the kernel an invoke runs on - and with it the order of its additions -
// is a function of its descriptor, its batch count and its own pointers' alignment only, never of the group it is queued with.
// launch_gemm_grouped then takes every decision that depends on the size of the work list as if the list held ONE item.
// QUADS: would a group of n_items invokes of `d` (batch count br each, all operands 16-byte aligned) run faster as n_items / 4
// 2 x 2 blocks on the 128x128 loader-wave tile than as items on the grouped 64x64 / 32x64 tiles? (the tile model of pick_bf16_lw_tile;
// bf16 VNNI-2 / VNNI-4, m = n = 64, k a multiple of 64; never in strict mode)
bool gemm_quads_pay(const GemmDesc &d, int n_items, int64_t br);
hipError_t launch_gemm_quads(const GemmDesc &d, const QuadItem *quads, int n_quads, int64_t br, hipStream_t stream);
int set_strict_kernels(int on); // returns the previous setting
bool strict_kernels();
int f32_chain_tile(const GemmDesc &d); // 1 / 2 / 3 = the f32 chain tile the descriptor was planned on, -1 = none (brgemm_f32.hip)
const char *last_grouped_kernel(); // kernel family of the most recent launch_gemm_grouped ("" before the first)
const char *last_refined_kernel(); // most recent launch_gemm: the kernel an invoke-time refinement chose, "" = the descriptor's own
// fills d.variant / d.name; returns false if no kernel can run the descriptor
bool plan_gemm(GemmDesc &d, int forced_variant);
constexpr int GEMM_VARIANT_BF16_LW0 = 20; // = V_BF16_LW_32x64: first of the four loader-wave bf16 tiles (brgemm_bf16_lw.hip)
constexpr int GEMM_VARIANT_BF16_LW4_0 = 28; // = V_BF16_LW4_32x64: the same four tiles for a VNNI-4 B operand
constexpr int GEMM_VARIANT_GENERIC = 8; // = V_GENERIC of brgemm_f32.hip: the generic kernel was chosen (or forced) at dispatch
// bf16 + VNNI-2 B, k a multiple of 64, m and n of 64, 16-byte-aligned leading dimensions within the 32-bit lane offsets: what the
// LDS-DMA bf16 tile families (brgemm_bf16.hip, brgemm_bf16_lw.hip) need
bool bf16_fast_eligible(const GemmDesc &d);
// which B image of the loader-wave bf16 tiles (brgemm_bf16_lw.hip) a descriptor's B operand needs - 0: VNNI-2, 2: flat [k][ldb],
// 4: VNNI-4 - or -1 if the descriptor cannot run on those tiles (shape / alignment / lane-offset limits of the LDS-DMA panels)
int bf16_lw_b_kind(const GemmDesc &d);
hipError_t launch_unary(const UnaryDesc &d, const void *in, float scalar, bool use_scalar, void *out,
                        hipStream_t stream);
hipError_t launch_binary(const BinaryDesc &d, const void *lhs, const void *rhs, void *out,
                         hipStream_t stream);
// n_items invokes of ONE unary / binary descriptor with m, n <= 64 in one launch
hipError_t launch_unary_grouped(const UnaryDesc &d, const WorkItem *items, int n_items, hipStream_t stream);
hipError_t launch_binary_grouped(const BinaryDesc &d, const WorkItem *items, int n_items, hipStream_t stream);

} // namespace tpp

; xsmm_calls.ll - a caller at the LLVM calling-convention level (SURVEY.md 8 f1 substitute; there is no MLIR toolchain in the image).
;
; Hand-written LLVM IR that calls the runtime with EXACTLY the signatures tpp-opt's -convert-xsmm-to-func emits and the LLVM lowering
; then turns into `llvm.call`s (lib/TPP/Conversion/ConvertXsmmToFunc/ConvertXsmmToFunc.cpp:37-78 builds the callee types, :298-352
; the operand lists; FileCheck'd in test/Conversion/XsmmToFunc/xsmm-to-func.mlir:13 (unary dispatch), :27 (brgemm dispatch),
; :44 (gemm dispatch), :144 (brgemm invoke), :165 (unary invoke), :242 (fused dispatch: 14 i64 - eight of them travel on the stack
; under the SysV ABI), :317 (binary)): every scalar an i64 (enums included), every memref operand a (ptr, i64 element offset)
; pair, the one `float` of xsmm_unary_scalar_invoke, an i64 handle back from dispatch. It does NOT include this repository's
; header: the declarations below are the reference's (runtime/Xsmm/XsmmRunnerUtils.h:22-83, PerfRunnerUtils.h:22-24).
; tests/test_abi_llvm_caller.py compiles it with the ROCm clang, links it against libtpp_xsmm_runner_utils.so with
; --no-as-needed like tools/tpp-run/CMakeLists.txt:74-86 does, and runs the golden fixtures named at each function.
target triple = "x86_64-unknown-linux-gnu"

declare i64 @xsmm_gemm_dispatch(i64, i64, i64, i64, i64, i64, i64, i64)
declare i64 @xsmm_brgemm_dispatch(i64, i64, i64, i64, i64, i64, i64, i64, i64, i64)
declare i64 @xsmm_fused_brgemm_dispatch(i64, i64, i64, i64, i64, i64, i64, i64, i64, i64, i64, i64, i64, i64)
declare i64 @xsmm_unary_dispatch(i64, i64, i64, i64, i64, i64, i64)
declare i64 @xsmm_binary_dispatch(i64, i64, i64, i64, i64, i64, i64, i64)
declare i64 @xsmm_intel_amx_tile_config_dispatch(i64, i64, i64, i64, i64, i64, i64, i64, i64, i64)
declare void @xsmm_gemm_invoke(i64, i64, ptr, i64, ptr, i64, ptr, i64)
declare void @xsmm_brgemm_invoke(i64, i64, ptr, i64, ptr, i64, ptr, i64, i64)
declare void @xsmm_fused_brgemm_invoke(i64, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i64, i64)
declare void @xsmm_unary_invoke(i64, i64, ptr, i64, ptr, i64)
declare void @xsmm_unary_scalar_invoke(i64, i64, float, ptr, i64)
declare void @xsmm_binary_invoke(i64, i64, ptr, i64, ptr, i64, ptr, i64)
declare void @xsmm_intel_amx_tile_config_invoke(i64, i64, ptr, i64)
declare i64 @perf_start_timer()
declare double @perf_stop_timer(i64)

; test/Integration/xsmm-fusion.mlir:51-52 (fixture xsmm_fusion_seed123): f32, C[4x4] = relu(sum_{b<2} A_b[4x8] B_b[8x4] + bias[col]),
; dispatch (1, 4, 4, 8, 8, 4, 4, 32, 32, 4, 0, 5, 4, 1), invoke (1, h, A, 0, B, 0, C, 0, bias, 0, 2)
define void @fusion_f32(ptr %A, ptr %B, ptr %C, ptr %bias) {
  %h = call i64 @xsmm_fused_brgemm_dispatch(i64 1, i64 4, i64 4, i64 8, i64 8, i64 4, i64 4, i64 32, i64 32, i64 4, i64 0, i64 5, i64 4, i64 1)
  call void @xsmm_fused_brgemm_invoke(i64 1, i64 %h, ptr %A, i64 0, ptr %B, i64 0, ptr %C, i64 0, ptr %bias, i64 0, i64 2)
  ret void
}

; test/BF16/Integration/xsmm-quarternary-bf16.mlir:4-14 (fixture xsmm_quarternary_bf16) after -intel-amx-tile-config-insertion-pass
; (lib/TPP/Transforms/IntelAMXTileConfig.cpp:62-112): two tile-config dispatches (flags | NO_RESET_TILECONFIG = 64 for the set-up,
; | NO_SETUP_TILECONFIG = 128 for the reset), the fused dispatch re-issued with both bits, a 64-byte alloca as the tile state, and
; the invoke between the two tile-config invokes. bf16 (2), wire flag 2048 = dialect vnni_b (ConvertXsmmToFunc.cpp:251-265).
define void @quarternary_bf16_amx(ptr %A, ptr %B, ptr %C, ptr %D) {
  %setup = call i64 @xsmm_intel_amx_tile_config_dispatch(i64 2, i64 4, i64 4, i64 4, i64 4, i64 4, i64 4, i64 8, i64 8, i64 2112)
  %reset = call i64 @xsmm_intel_amx_tile_config_dispatch(i64 2, i64 4, i64 4, i64 4, i64 4, i64 4, i64 4, i64 8, i64 8, i64 2176)
  %h = call i64 @xsmm_fused_brgemm_dispatch(i64 2, i64 4, i64 4, i64 4, i64 4, i64 4, i64 4, i64 8, i64 8, i64 2240, i64 0, i64 5, i64 4, i64 1)
  %state = alloca [64 x i8], align 64
  call void @xsmm_intel_amx_tile_config_invoke(i64 2, i64 %setup, ptr %state, i64 0)
  call void @xsmm_fused_brgemm_invoke(i64 2, i64 %h, ptr %A, i64 0, ptr %B, i64 0, ptr %C, i64 0, ptr %D, i64 0, i64 16)
  call void @xsmm_intel_amx_tile_config_invoke(i64 2, i64 %reset, ptr %state, i64 0)
  ret void
}

; test/BF16/Integration/xsmm-brgemm-bf16.mlir:5-20 (fixture xsmm_brgemm_bf16) with the same wrapper: C[6x6] += sum_{b<2} A_b B_b
define void @brgemm_bf16_amx(ptr %A, ptr %B, ptr %C) {
  %setup = call i64 @xsmm_intel_amx_tile_config_dispatch(i64 2, i64 6, i64 6, i64 6, i64 6, i64 6, i64 6, i64 36, i64 36, i64 2112)
  %reset = call i64 @xsmm_intel_amx_tile_config_dispatch(i64 2, i64 6, i64 6, i64 6, i64 6, i64 6, i64 6, i64 36, i64 36, i64 2176)
  %h = call i64 @xsmm_brgemm_dispatch(i64 2, i64 6, i64 6, i64 6, i64 6, i64 6, i64 6, i64 36, i64 36, i64 2240)
  %state = alloca [64 x i8], align 64
  call void @xsmm_intel_amx_tile_config_invoke(i64 2, i64 %setup, ptr %state, i64 0)
  call void @xsmm_brgemm_invoke(i64 2, i64 %h, ptr %A, i64 0, ptr %B, i64 0, ptr %C, i64 0, i64 2)
  call void @xsmm_intel_amx_tile_config_invoke(i64 2, i64 %reset, ptr %state, i64 0)
  ret void
}

; test/BF16/Integration/xsmm-gemm-bf16.mlir:5-17 (fixture xsmm_gemm_bf16): the un-batched call
define void @gemm_bf16(ptr %A, ptr %B, ptr %C) {
  %h = call i64 @xsmm_gemm_dispatch(i64 2, i64 6, i64 6, i64 6, i64 6, i64 6, i64 6, i64 2048)
  call void @xsmm_gemm_invoke(i64 2, i64 %h, ptr %A, i64 0, ptr %B, i64 0, ptr %C, i64 0)
  ret void
}

; test/Integration/xsmm-zero.mlir:5-16 (fixture xsmm_zero): unary zero in place, element offsets exercised by the second call
; (rows 1..2 of the 3x3 buffer re-zeroed through offset 3: get_base_ptr semantics, XsmmRunnerUtils.cpp:63-75)
define void @zero_f32(ptr %X) {
  %h = call i64 @xsmm_unary_dispatch(i64 2, i64 1, i64 3, i64 3, i64 3, i64 3, i64 0)
  call void @xsmm_unary_invoke(i64 1, i64 %h, ptr %X, i64 0, ptr %X, i64 0)
  %h2 = call i64 @xsmm_unary_dispatch(i64 2, i64 1, i64 2, i64 3, i64 3, i64 3, i64 0)
  call void @xsmm_unary_invoke(i64 1, i64 %h2, ptr %X, i64 3, ptr %X, i64 3)
  ret void
}

; test/Integration/xsmm-binary.mlir (fixture xsmm_binary_add): out = lhs + rhs on 3x3 (binary dispatch: kind first, then dtype)
define void @binary_add_f32(ptr %L, ptr %R, ptr %O) {
  %h = call i64 @xsmm_binary_dispatch(i64 1, i64 1, i64 3, i64 3, i64 3, i64 3, i64 3, i64 0)
  call void @xsmm_binary_invoke(i64 1, i64 %h, ptr %L, i64 0, ptr %R, i64 0, ptr %O, i64 0)
  ret void
}

; the scalar-operand form (test/Passes/fold-xsmm-flags.mlir:7-8; XsmmRunnerUtils.cpp:276-286): identity with bcast_scalar (8) fills
; a 4x8 f32 tile (ldo 8) with the `float` argument - passed in %xmm0 by value, NOT as an i64 - inside a perf timer pair
; (test/Conversion/PerfToFunc/perf-to-func.mlir:3-4); returns the measured seconds
define double @fill_scalar_f32_timed(ptr %O, float %v) {
  %t = call i64 @perf_start_timer()
  %h = call i64 @xsmm_unary_dispatch(i64 1, i64 1, i64 4, i64 8, i64 1, i64 8, i64 8)
  call void @xsmm_unary_scalar_invoke(i64 1, i64 %h, float %v, ptr %O, i64 0)
  %s = call double @perf_stop_timer(i64 %t)
  ret double %s
}

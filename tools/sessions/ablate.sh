#!/bin/bash
# builds side libraries of the runtime with ablation masks / accumulator counts (timing only)
# usage: tools/ablate.sh "mask:nacc ..."   -> tpp-mlir_amd/build/libabl_<mask>_n<nacc>.so
set -e
cd "$(dirname "$0")/.."
C=tpp-mlir_amd/csrc; B=tpp-mlir_amd/build; mkdir -p $B
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
rm -f $B/libabl_*.so
for spec in $1; do
  m=${spec%%:*}; n=${spec##*:}
  ( hipcc $FLAGS -DTPP_ABLATE=$m -DTPP_NACC=$n -c $C/brgemm_f32.hip -o $B/abl_f32_${m}_$n.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libabl_${m}_n$n.so $B/abl_f32_${m}_$n.o $B/runtime.o $B/brgemm_bf16.o $B/brgemm_f32_lw.o $B/brgemm_bf16_dma256.o $B/brgemm_bf16_small.o $B/eltwise.o -pthread ) &
done
wait
ls $B/libabl_*.so

#!/bin/bash
# side libraries of the runtime with parts of the bf16 128x128 kernel compiled out (timing only; results are garbage)
# usage: tools/ablate_bf16.sh "mask ..."   -> tpp-mlir_amd/build/libabl_h<mask>.so
set -e
cd "$(dirname "$0")/.."
C=tpp-mlir_amd/csrc; B=tpp-mlir_amd/build; mkdir -p $B
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
rm -f $B/libabl_*.so
for m in $1; do
  ( hipcc $FLAGS -DTPP_ABLATE=$m -c $C/brgemm_bf16.hip -o $B/abl_bf16_$m.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libabl_h$m.so $B/abl_bf16_$m.o $B/runtime.o $B/brgemm_f32.o $B/brgemm_f32_lw.o $B/brgemm_bf16_dma256.o $B/brgemm_bf16_small.o $B/eltwise.o -pthread ) &
done
wait
ls $B/libabl_*.so

#!/bin/bash
# side libraries of the 256x256 bf16 kernel with timeline stamps (+ ablation masks given as arguments)
set -e
cd "$(dirname "$0")/.."
C=tpp-mlir_amd/csrc; B=tpp-mlir_amd/build
rm -f $B/libstamp256*.so
for m in 0 "$@"; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTPP_STAMP256=1 -DTPP_ABLATE256=$m -c $C/brgemm_bf16_dma256.hip -o $B/stamp256_$m.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libstamp256_$m.so $B/stamp256_$m.o $B/runtime.o $B/brgemm_f32.o $B/brgemm_bf16.o $B/brgemm_bf16_small.o $B/eltwise.o -pthread ) &
done
wait
ls $B/libstamp256*.so

#!/bin/bash
OUT=gpurun_out/${1:-r02f}; mkdir -p $OUT
for lw in 0 4 2; do
  TPP_HIP_BF16_LEGACY=$lw timeout 120 python tools/abl_case_bf16.py "LW=$lw(0=default 2 loaders,4=4 loaders,2=none)" 2>/dev/null | grep "^bf16" | tee -a $OUT/bf16_lw.txt
done
for f in $GRAFT_REPO_ROOT/tpp-mlir_amd/build/libexp_*.so; do
  [ -f "$f" ] || continue
  TPP_XSMM_LIBRARY=$f timeout 120 python tools/abl_case_bf16.py "$(basename $f)" 2>/dev/null | grep "^bf16" | tee -a $OUT/bf16_lw.txt
done

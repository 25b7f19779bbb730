#!/bin/bash
OUT=gpurun_out/${1:-abl}; mkdir -p $OUT
for m in 0 8 64 72 128 256 384 4 392 396 0; do
  TPP_XSMM_LIBRARY=$GRAFT_REPO_ROOT/tpp-mlir_amd/build/libabl_h$m.so timeout 120 python tools/abl_case_bf16.py "abl=$m" 2>/dev/null | grep "^bf16" >> $OUT/abl_bf16.txt
done
cat $OUT/abl_bf16.txt

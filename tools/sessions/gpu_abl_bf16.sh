#!/bin/bash
OUT=gpurun_out/${1:-abl}; mkdir -p $OUT
for f in $GRAFT_REPO_ROOT/tpp-mlir_amd/build/libabl_h*.so; do
  TPP_XSMM_LIBRARY=$f timeout 120 python tools/abl_case_bf16.py "$(basename $f .so)" 2>/dev/null | grep "^bf16" >> $OUT/abl_bf16.txt
done
cat $OUT/abl_bf16.txt

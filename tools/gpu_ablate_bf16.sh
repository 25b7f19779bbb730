#!/bin/bash
TAG=${1:-ablh}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python tools/abl_case_bf16.py "full   " 2>/dev/null | grep "^bf16" > $OUT/ablate.log
for lib in tpp-mlir_amd/build/libabl_h*.so; do
  t=$(basename $lib .so); t=${t#libabl_}
  TPP_XSMM_LIBRARY=$PWD/$lib python tools/abl_case_bf16.py "abl=$t" 2>/dev/null | grep "^bf16" >> $OUT/ablate.log
done
cat $OUT/ablate.log

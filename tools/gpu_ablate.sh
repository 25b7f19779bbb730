#!/bin/bash
TAG=${1:-abl}; FORCES=${2:-0}
OUT=gpurun_out/$TAG; mkdir -p $OUT
python tools/abl_case.py "full       " $FORCES 2>/dev/null | grep "^f32" > $OUT/ablate.log
for lib in tpp-mlir_amd/build/libabl_*.so; do
  t=$(basename $lib .so); t=${t#libabl_}
  TPP_XSMM_LIBRARY=$PWD/$lib python tools/abl_case.py "abl=$t" $FORCES 2>/dev/null | grep "^f32" >> $OUT/ablate.log
done
cat $OUT/ablate.log

#!/bin/bash
TAG=${1:-r3_g}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for alt in 0 1 2; do echo "alt=$alt"; TPP_HIP_BLW_ALT=$alt timeout 100 tools/mlp_probe 2>&1 | cut -c1-14,50-200;
for v in 23; do TPP_HIP_BLW_ALT=$alt timeout 60 tools/mlp_probe --variant $v --rows 4096 2>&1 | cut -c1-14,50-200; done; done > $OUT/alt.txt; cat $OUT/alt.txt

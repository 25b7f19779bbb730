#!/bin/bash
TAG=${1:-r3_t}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for cfg in "0 -1" "1 -1" "0 4" "1 4" "0 -1" "1 -1"; do set -- $cfg; echo "RING=$1 XM=$2"; X=""; [ "$2" != "-1" ] && X="TPP_HIP_BF16_LW_XM=$2"
env TPP_HIP_BLW_RING=$1 $X timeout 100 tools/mlp_probe --rows 2048,4096 2>&1 | cut -c1-14,50-200; env TPP_HIP_BLW_RING=$1 $X timeout 60 tools/mlp_probe --variant 23 --rows 4096 2>&1 | cut -c1-14,50-200; done > $OUT/ab.txt; cat $OUT/ab.txt

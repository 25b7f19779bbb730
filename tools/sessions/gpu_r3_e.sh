#!/bin/bash
TAG=${1:-r3_e}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for dbg in 0 16 32 48 64 128 96 160; do echo "dbg=$dbg"; TPP_HIP_CHAIN_DBG=$dbg timeout 100 tools/mlp_probe --only layers 2>&1 | cut -c1-14,60-150; done > $OUT/abl.txt; cat $OUT/abl.txt

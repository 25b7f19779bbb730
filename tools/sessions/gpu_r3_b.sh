#!/bin/bash
TAG=${1:-r3_b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_chain_gpu.py -x -q -p no:cacheprovider > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; tail -4 $OUT/t1.log
for dbg in 0 1 2 4 3 5 6 7; do echo "dbg=$dbg" >> $OUT/chain_dbg.txt; TPP_HIP_CHAIN_DBG=$dbg timeout 60 tools/mlp_probe --only chain >> $OUT/chain_dbg.txt 2>&1; done; cat $OUT/chain_dbg.txt | cut -c1-60,150-260
for v in 22 23; do timeout 60 tools/mlp_probe --variant $v --only layers --rows 1024,2048,4096 >> $OUT/forced.txt 2>&1; done; cat $OUT/forced.txt | cut -c1-200

#!/bin/bash
TAG=${1:-r3_h}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_chain_gpu.py -x -q -p no:cacheprovider > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; tail -4 $OUT/t1.log
for dbg in 0 256 0 256; do echo "dbg=$dbg"; TPP_HIP_CHAIN_DBG=$dbg timeout 100 tools/mlp_probe 2>&1 | cut -c1-14,50-200;
TPP_HIP_CHAIN_DBG=$dbg timeout 60 tools/mlp_probe --variant 23 --rows 4096 2>&1 | cut -c1-14,50-200; done > $OUT/ab.txt; cat $OUT/ab.txt
for rows in 512 4096; do for dbg in 0; do
TPP_HIP_CHAIN_DBG=$dbg TPP_HIP_CHAIN_STAMPS=$OUT/stamps_${rows}_$dbg.txt timeout 60 tools/mlp_probe --only chain --rows $rows --iters 50 > /dev/null 2>&1
echo "== rows $rows dbg $dbg"; python tools/stamps_report.py $OUT/stamps_${rows}_$dbg.txt; done; done

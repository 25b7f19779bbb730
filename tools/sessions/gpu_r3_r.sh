#!/bin/bash
TAG=${1:-r3_r}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_chain_gpu.py -x -q -p no:cacheprovider > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; tail -3 $OUT/t1.log
for sup in 0 1 0 1; do echo "TPP_HIP_BLW_SUP=$sup"; TPP_HIP_BLW_SUP=$sup timeout 100 tools/mlp_probe --rows 512,1024 2>&1 | cut -c1-14,50-200; done > $OUT/ab.txt; cat $OUT/ab.txt
for dbg in 48; do echo "dbg=$dbg"; TPP_HIP_CHAIN_DBG=$dbg timeout 100 tools/mlp_probe --rows 512,1024 2>&1 | cut -c1-14,50-200; done
for rows in 512; do
TPP_HIP_CHAIN_STAMPS=$OUT/stamps_${rows}.txt timeout 60 tools/mlp_probe --only chain --rows $rows --iters 50 > /dev/null 2>&1
echo "== rows $rows"; python tools/stamps_report.py $OUT/stamps_${rows}.txt; done

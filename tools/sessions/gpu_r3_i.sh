#!/bin/bash
TAG=${1:-r3_i}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_chain_gpu.py -x -q -p no:cacheprovider > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; tail -3 $OUT/t1.log
for rows in 512 1024 2048 4096; do
TPP_HIP_CHAIN_STAMPS=$OUT/stamps_${rows}.txt timeout 60 tools/mlp_probe --only chain --rows $rows --iters 50 > /dev/null 2>&1
echo "== rows $rows"; python tools/stamps_report.py $OUT/stamps_${rows}.txt; done
for dbg in 0 32 16; do echo "dbg=$dbg"; TPP_HIP_CHAIN_DBG=$dbg timeout 100 tools/mlp_probe --only layers 2>&1 | cut -c1-14,50-150; done

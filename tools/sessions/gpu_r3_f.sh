#!/bin/bash
TAG=${1:-r3_f}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_chain_gpu.py -x -q -p no:cacheprovider > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; tail -4 $OUT/t1.log
timeout 100 tools/mlp_probe > $OUT/mlp_probe.txt 2>&1; cut -c1-60,88-260 $OUT/mlp_probe.txt
for dbg in 16 32 48; do echo "dbg=$dbg"; TPP_HIP_CHAIN_DBG=$dbg timeout 100 tools/mlp_probe --only layers 2>&1 | cut -c1-14,60-150; done > $OUT/abl.txt; cat $OUT/abl.txt
for rows in 512 4096; do for dbg in 0; do
TPP_HIP_CHAIN_DBG=$dbg TPP_HIP_CHAIN_STAMPS=$OUT/stamps_${rows}_$dbg.txt timeout 60 tools/mlp_probe --only chain --rows $rows --iters 50 > /dev/null 2>&1
echo "== rows $rows dbg $dbg"; python tools/stamps_report.py $OUT/stamps_${rows}_$dbg.txt; done; done
for v in 20 21 22 23; do timeout 60 tools/mlp_probe --variant $v --only layers --rows 256,512,1024,2048,4096 2>&1 | cut -c1-14,40-160; done > $OUT/forced.txt; cat $OUT/forced.txt

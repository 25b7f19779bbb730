#!/bin/bash
TAG=${1:-r3_u}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_chain_gpu.py tests/test_parity_gpu.py -x -q -p no:cacheprovider -k "chain or lw or bf16 or c4" > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; tail -3 $OUT/t1.log
for i in 1 2; do timeout 100 tools/mlp_probe 2>&1 | cut -c1-14,50-200; timeout 60 tools/mlp_probe --variant 23 --rows 4096 2>&1 | cut -c1-14,50-200; done
for rows in 4096; do
TPP_HIP_CHAIN_STAMPS=$OUT/stamps_${rows}.txt timeout 60 tools/mlp_probe --only chain --rows $rows --iters 50 > /dev/null 2>&1
echo "== rows $rows"; python tools/stamps_report.py $OUT/stamps_${rows}.txt; done

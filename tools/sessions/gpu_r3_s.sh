#!/bin/bash
TAG=${1:-r3_s}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_chain_gpu.py -x -q -p no:cacheprovider > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; tail -5 $OUT/t1.log
for i in 1 2; do timeout 100 tools/mlp_probe 2>&1 | cut -c1-14,50-200; timeout 60 tools/mlp_probe --variant 23 --rows 4096 2>&1 | cut -c1-14,50-200; done

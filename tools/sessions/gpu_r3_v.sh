#!/bin/bash
TAG=${1:-r3_v}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_chain_gpu.py -x -q -p no:cacheprovider > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; tail -3 $OUT/t1.log
for sup in 0 2 0 2; do echo "SUP=$sup"; TPP_HIP_BLW_SUP=$sup timeout 100 tools/mlp_probe --rows 256,512 2>&1 | cut -c1-14,50-200; done

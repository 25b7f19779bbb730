#!/bin/bash
TAG=${1:-r3_n}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_peer_gather_gpu.py -x -q -p no:cacheprovider > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; tail -30 $OUT/t1.log

#!/bin/bash
TAG=${1:-r3_p}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_tile_queue_gpu.py -x -q -p no:cacheprovider -k "f32 or c2 or c3 or random or ragged or fixture or queue" > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; tail -3 $OUT/t1.log
for i in 1 2; do
for v in -1 4; do tools/c2_probe --iters 400 --init reference --variant $v | cut -c1-110; done
for v in -1 1; do tools/c2_probe --c3 --iters 400 --init reference --variant $v | cut -c1-110; done
done

#!/bin/bash
TAG=${1:-r02c}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sharded_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_sharded.log 2>&1; echo "rc=$?" >> $OUT/pytest_sharded.log
tail -8 $OUT/pytest_sharded.log
for init in uniform reference; do for v in 0 4 5 6 9 7 1 0 4 5 6; do
  timeout 60 tools/c2_probe --variant $v --init $init --iters 400 >> $OUT/probe.txt 2>> $OUT/probe.err
done; done
cut -c1-200 $OUT/probe.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err; head -c 1800 $OUT/bench_20.json; tail -3 $OUT/bench_20.err

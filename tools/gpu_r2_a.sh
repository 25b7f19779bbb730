#!/bin/bash
# round-2 session A: new f32 kernels - parity subset, variant A/B on C2 (both input streams), bench line
TAG=${1:-r02a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k "f32 or fixture or random" > $OUT/pytest_f32.log 2>&1; echo "rc=$?" >> $OUT/pytest_f32.log
tail -5 $OUT/pytest_f32.log
timeout 600 python -m pytest tests/test_sharded_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_sharded.log 2>&1; echo "rc=$?" >> $OUT/pytest_sharded.log
tail -15 $OUT/pytest_sharded.log
for init in uniform reference; do for v in 0 5 6 4 7 1 0 5; do
  timeout 60 tools/c2_probe --variant $v --init $init --iters 400 >> $OUT/probe.txt 2>> $OUT/probe.err
done; done
cat $OUT/probe.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err; tail -c 3000 $OUT/bench_20.json; tail -5 $OUT/bench_20.err

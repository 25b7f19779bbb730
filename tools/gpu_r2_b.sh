#!/bin/bash
# round-2 session B: full parity suite after the runtime changes, f32 variant sweep, host timing anatomy, bench
TAG=${1:-r02b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
timeout 600 python tools/sweep.py f32lw 2>/dev/null | grep -E "^(f32|bf16)" > $OUT/sweep_f32lw.txt; cat $OUT/sweep_f32lw.txt
timeout 300 python tools/timing_probe.py > $OUT/timing_probe.txt 2>&1; cat $OUT/timing_probe.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err; head -c 2500 $OUT/bench_20.json; tail -3 $OUT/bench_20.err

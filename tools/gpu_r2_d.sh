#!/bin/bash
# round-2 session D: full parity suite, bench line, rocprofv3 kernel stats of the same bench command
TAG=${1:-r02d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|error|parity\]" $OUT/pytest_gpu.log | tail -8
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err; head -c 600 $OUT/bench_20.json; echo
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python $R/bench.py --no-cpu-baseline --no-pmc > $R/$OUT/rocprof_stats_run.json 2> $R/$OUT/rocprof_stats.err )
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/rocprof_kernel_stats.csv \;
head -12 $OUT/rocprof_kernel_stats.csv | cut -c1-220

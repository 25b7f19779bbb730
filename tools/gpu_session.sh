#!/bin/bash
# One gpurun session: parity tests, bench, rocprof summary. Everything lands in gpurun_out/.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_session.sh [tag]'
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
  echo "== device"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8
  echo "== smoke"; timeout 600 python __graft_entry__.py --smoke
  echo "smoke rc=$?"
} > $OUT/smoke.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/bench.err
timeout 600 python bench.py --steps 200 --warmup 20 --graph --no-cpu-baseline > $OUT/bench_graph.json 2> $OUT/bench_graph.err
echo "bench graph rc=$?" >> $OUT/bench_graph.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1 )
find /tmp/prof_$TAG -name "*stats*" -exec cp {} $OUT/ \; 2>/dev/null
find /tmp/prof_$TAG -name "*kernel_trace*" -exec sh -c 'head -400 "$1" > '"$OUT"'/kernel_trace_head.csv' _ {} \; 2>/dev/null
ls -la /tmp/prof_$TAG/* >> $OUT/rocprof_bench.log 2>&1
tail -3 $OUT/smoke.log; tail -15 $OUT/pytest_gpu.log; cat $OUT/bench.json; tail -2 $OUT/bench.err; cat $OUT/bench_graph.json

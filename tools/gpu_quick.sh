#!/bin/bash
# quick measurement session: sweep + rocprof of the bench. usage: gpurun -- 'bash tools/gpu_quick.sh tag'
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/sweep.py > $OUT/sweep.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1 )
find /tmp/prof_$TAG -name "*stats*" -exec cp {} $OUT/ \; 2>/dev/null
for f in $(find /tmp/prof_$TAG -name "*kernel_trace*"); do head -300 $f > $OUT/kernel_trace_head.csv; done
grep -v "^W2026\|^E2026" $OUT/sweep.log
cat $OUT/*kernel_stats* 2>/dev/null | head -20

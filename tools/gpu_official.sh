#!/bin/bash
# Round deliverable session: full parity suite, bench line, rocprofv3 kernel stats of the SAME bench
# command, and HBM-traffic PMC passes (each in its own run). Results in gpurun_out/<tag>/.
TAG=${1:-official}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
BENCH="python $R/bench.py --no-cpu-baseline"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- $BENCH > $R/$OUT/rocprof_stats_run.json 2> $R/$OUT/rocprof_stats.err )
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/rocprof_kernel_stats.csv \;
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -o bench -- $BENCH --steps 200 --warmup 50 --no-mlp > /dev/null 2> $R/$OUT/rocprof_fetch.err )
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -o bench -- $BENCH --steps 200 --warmup 50 --no-mlp > /dev/null 2> $R/$OUT/rocprof_write.err )
for k in fetch write; do f=$(find /tmp/prof_$k -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" $k >> $OUT/rocprof_pmc_summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%s pass: %-70s %-12s dispatches %5d  mean %14.1f  min %14.1f  max %14.1f" % (sys.argv[2], k[:70], c, len(v), sum(v) / len(v), min(v), max(v)))
PY
done
python tools/sweep.py 2>/dev/null | grep -E "^(f32|bf16)" > $OUT/sweep.txt
python tools/sweep.py big 2>/dev/null | grep -E "^(f32|bf16)" >> $OUT/sweep.txt
for t in 1 100000; do TPP_HIP_BF16_T128MIN=$t python tools/sweep.py shards 2>/dev/null | grep -E "^bf16" >> $OUT/sweep.txt; done
bash tools/gpu_queue.sh $TAG/queue > /dev/null 2>&1
python tools/eltwise_bw.py > $OUT/eltwise_bw.txt 2>/dev/null
tail -3 $OUT/pytest_gpu.log; tail -1 $OUT/smoke.log; cat $OUT/bench.json; cat $OUT/rocprof_kernel_stats.csv | head -8; cat $OUT/rocprof_pmc_summary.txt

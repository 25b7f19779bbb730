#!/bin/bash
# Round deliverable session: full parity suite, smoke, the driver's bench command, rocprofv3 kernel stats of the SAME
# bench command, default bench, shape sweeps, tile-queue replays, eltwise bandwidth. Results in gpurun_out/<tag>/.
TAG=${1:-official}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|error|\[parity\]|rc=" $OUT/pytest_gpu.log | tail -5
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $R/$OUT/rocprof_stats_run.json 2> $R/$OUT/rocprof_stats.err )
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/rocprof_kernel_stats.csv \;
python tools/sweep.py 2>/dev/null | grep -E "^(f32|bf16)" > $OUT/sweep.txt
python tools/sweep.py big 2>/dev/null | grep -E "^(f32|bf16)" >> $OUT/sweep.txt
python tools/sweep.py shards 2>/dev/null | grep -E "^bf16" >> $OUT/sweep.txt
python tools/sweep.py small 2>/dev/null | grep -E "^bf16" >> $OUT/sweep.txt
bash tools/gpu_replay.sh $TAG > /dev/null 2>&1
python tools/eltwise_bw.py > $OUT/eltwise_bw.txt 2>/dev/null
python tools/vendor_compare.py > $OUT/vendor_compare.txt 2>/dev/null
head -c 700 $OUT/bench_steps20.json; echo; head -8 $OUT/rocprof_kernel_stats.csv | cut -c1-160

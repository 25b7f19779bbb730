#!/bin/bash
# session H: full suite after runtime changes; tile-queue replay numbers; pack kernel; marker trace check
OUT=gpurun_out/${1:-r02h}; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
for extra in "--tiles 32 --queue 1 -n 200" "--tiles 64 --queue 1 -n 200" "--tiles 32 --queue 1 -n 200 --bf16" "--tiles 64 --queue 1 -n 200 --bf16" "--whole-layer -n 1000" "--tiles 32 --queue 1 -n 200 --threads 8"; do
  echo "== $extra" >> $OUT/replay.txt
  timeout 120 tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --bias --relu $extra 2>&1 | tail -2 >> $OUT/replay.txt
done
echo "== c1" >> $OUT/replay.txt; timeout 60 tools/tpp_replay --c1 --queue 1 2>&1 | tail -2 >> $OUT/replay.txt
cat $OUT/replay.txt
timeout 200 python tools/eltwise_bw.py > $OUT/eltwise_bw.txt 2>&1; cat $OUT/eltwise_bw.txt | tail -30
( cd /tmp && TPP_HIP_TRACE=1 timeout 120 rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/mk -o m -- $R/tools/c2_probe --iters 5 > /dev/null 2> $R/$OUT/marker.err ); find /tmp/mk -name "*marker*csv" -exec head -5 {} \; > $OUT/marker_trace_head.txt; cat $OUT/marker_trace_head.txt | cut -c1-300

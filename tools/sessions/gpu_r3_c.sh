#!/bin/bash
TAG=${1:-r3_c}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
TPP_HIP_TRACE=1 timeout 100 python -m pytest tests/test_chain_gpu.py -x -q -p no:cacheprovider -k sharded_mlp > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; grep -E "call by call|passed|failed" $OUT/t1.log | tail -4
for rows in 512 4096; do for dbg in 0 7; do
TPP_HIP_CHAIN_DBG=$dbg TPP_HIP_CHAIN_STAMPS=$OUT/stamps_${rows}_$dbg.txt timeout 60 tools/mlp_probe --only chain --rows $rows --iters 50 > /dev/null 2>&1
echo "== rows $rows dbg $dbg"; python tools/stamps_report.py $OUT/stamps_${rows}_$dbg.txt; done; done
cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o p -- $GRAFT_REPO_ROOT/tools/mlp_probe --rows 512,1024,2048,4096 --iters 200 > /dev/null 2>&1; find /tmp/prof1 -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/$OUT/kernel_stats.csv \; ; cut -c1-150 $GRAFT_REPO_ROOT/$OUT/kernel_stats.csv

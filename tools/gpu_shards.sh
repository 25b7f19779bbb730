#!/bin/bash
# parity tests + the shard-shape kernel-choice sweep. usage: gpurun -- 'bash tools/gpu_shards.sh tag'
TAG=${1:-shards}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider -x -q 2>&1 | tail -15 > $OUT/pytest.log
tail -4 $OUT/pytest.log
for t in 1 100000; do
  TPP_HIP_BF16_T128MIN=$t timeout 300 python tools/sweep.py shards 2>/dev/null | grep -E "^(f32|bf16)" | tee -a $OUT/shards.log
done

#!/bin/bash
# A/B of two builds on the bf16 shapes: in-tree library vs a side library. usage: gpurun -- 'bash tools/gpu_ab.sh <side.so>'
SIDE=$1
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k "bf16" 2>&1 | tail -2
for r in 1 2; do
  echo "-- in-tree"; python tools/sweep.py bf16 2>/dev/null | grep -E "^bf16"
  echo "-- $SIDE"; TPP_XSMM_LIBRARY=$PWD/$SIDE python tools/sweep.py bf16 2>/dev/null | grep -E "^bf16"
done

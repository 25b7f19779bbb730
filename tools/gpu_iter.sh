#!/bin/bash
# iteration session: parity tests + sweep (+ optional ablation). usage: gpurun -- 'bash tools/gpu_iter.sh tag [abl]'
TAG=${1:-it}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider -x -q 2>&1 | tail -15 > $OUT/pytest.log
cat $OUT/pytest.log | tail -8
timeout 600 python tools/sweep.py 2>/dev/null | grep -E "^(f32|bf16)" > $OUT/sweep.log; cat $OUT/sweep.log
if [ -n "$2" ]; then bash tools/gpu_ablate.sh $TAG 0; fi

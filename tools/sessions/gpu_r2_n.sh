#!/bin/bash
# trace cache of the tile queue: parity (tile-queue tests incl. the replay / divergence test), then the headline call pattern
OUT=gpurun_out/${1:-r2_n}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_tile_queue_gpu.py -q -x 2>&1 | tail -5 > $OUT/pytest.txt; cat $OUT/pytest.txt
for cfg in "--tiles 32" "--tiles 32 --bf16" "--tiles 64" "--tiles 64 --bf16" "--whole-layer" "--tiles 32 --threads 2" "--tiles 32 --threads 4" "--tiles 32 --threads 8" "--c1"; do
  for rep in 1 2 3; do timeout 120 tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --bias --relu $cfg --queue 1 -n 300 2>&1 | grep "mean" | sed "s/^/[$cfg] /" | cut -c1-150; done
done > $OUT/replay.txt; cat $OUT/replay.txt

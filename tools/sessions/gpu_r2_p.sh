#!/bin/bash
# loader-wave f32 kernels in grouped mode (tile queue, tiles with k a multiple of 64): parity + replay
OUT=gpurun_out/${1:-r2_p}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_tile_queue_gpu.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/pytest.txt; cat $OUT/pytest.txt
for cfg in "--tiles 64" "--tiles 64,64,64 --batch 512" "--tiles 32"; do
  for rep in 1 2 3; do timeout 120 tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --bias --relu $cfg --queue 1 -n 300 2>&1 | grep "mean" | sed "s/^/[$cfg] /" | cut -c1-150; done
done > $OUT/replay.txt; cat $OUT/replay.txt
bash tools/gpu_r2_m.sh $(basename $OUT) > /dev/null 2>&1; grep -A4 "== --tiles 64$" $OUT/kernel_stats.txt | cut -c1-170

#!/bin/bash
# grouped (generic) kernel with two chunks of loads in flight: full parity, then replay timing + kernel time
OUT=gpurun_out/${1:-r2_o}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/pytest.txt; cat $OUT/pytest.txt
for cfg in "--tiles 32" "--tiles 32 --bf16" "--tiles 64" "--c1"; do
  for rep in 1 2 3; do timeout 120 tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --bias --relu $cfg --queue 1 -n 300 2>&1 | grep "mean" | sed "s/^/[$cfg] /" | cut -c1-150; done
done > $OUT/replay.txt; cat $OUT/replay.txt
bash tools/gpu_r2_m.sh $(basename $OUT) > /dev/null 2>&1; grep -A3 "== --tiles 32$" $OUT/kernel_stats.txt | cut -c1-170

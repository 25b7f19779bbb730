#!/bin/bash
OUT=gpurun_out/${1:-replay_thr}; mkdir -p $OUT
for rep in 1 2 3; do for thr in 1 2 4 8; do
  timeout 120 tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --bias --relu --tiles 32 --queue 1 -n 200 --threads $thr 2>&1 | tail -2 | head -1 | sed "s/^/thr=$thr rep=$rep /" >> $OUT/replay_thr.txt
done; done
cut -c1-150 $OUT/replay_thr.txt

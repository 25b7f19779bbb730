#!/bin/bash
OUT=gpurun_out/${1:-replay}; mkdir -p $OUT
for extra in "--tiles 32 --queue 1 -n 200" "--tiles 64 --queue 1 -n 200" "--tiles 32 --queue 1 -n 200 --bf16" "--tiles 64 --queue 1 -n 200 --bf16" "--whole-layer -n 1000" "--tiles 32 --queue 1 -n 200 --threads 8" "--tiles 32 --queue 1 -n 200 --threads 2"; do
  echo "== $extra" >> $OUT/replay.txt
  timeout 120 tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --bias --relu $extra 2>&1 | tail -3 >> $OUT/replay.txt
done
# several callers (the reference's OpenMP team over the tile grid): 10 runs each, default OMP placement
for t in 1 2 4 8; do for i in 1 2 3 4 5 6 7 8 9 10; do
  echo "== --tiles 32 --queue 1 -n 200 --threads $t (run $i)" >> $OUT/replay.txt
  timeout 120 tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --bias --relu --tiles 32 --queue 1 -n 200 --threads $t 2>&1 | grep mean | cut -c38-125 >> $OUT/replay.txt
done; done
# the whole-layer calls of a step handed over together (f32 chain: 64-row tiles only; batch 256 runs call by call)
for b in 256 512 1024; do for mode in --whole-layer --chain; do
  echo "== $mode --batch $b -n 1000" >> $OUT/replay.txt
  timeout 120 tools/tpp_replay --batch $b --layers 1024,1024,1024,1024 --bias --relu $mode -n 1000 2>&1 | grep mean | cut -c1-200 >> $OUT/replay.txt
done; done
echo "== c1" >> $OUT/replay.txt; timeout 60 tools/tpp_replay --c1 --queue 1 2>&1 | tail -3 >> $OUT/replay.txt
cat $OUT/replay.txt

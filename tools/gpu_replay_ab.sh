#!/bin/bash
OUT=gpurun_out/${1:-replay_ab}; mkdir -p $OUT
cp tpp-mlir_amd/libtpp_xsmm_runner_utils.so /tmp/new.so
for rep in 1 2 3; do for lib in new old; do for thr in 2 8; do
  if [ $lib = old ]; then cp tpp-mlir_amd/build/libexp_sched_old.so tpp-mlir_amd/libtpp_xsmm_runner_utils.so; else cp /tmp/new.so tpp-mlir_amd/libtpp_xsmm_runner_utils.so; fi
  timeout 120 tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --bias --relu --tiles 32 --queue 1 -n 200 --threads $thr 2>&1 | tail -2 | head -1 | sed "s/^/lib=$lib thr=$thr /" | cut -c1-140 >> $OUT/ab.txt
done; done; done
cat $OUT/ab.txt

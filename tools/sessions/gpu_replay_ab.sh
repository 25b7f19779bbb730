#!/bin/bash
# same-box A/B of runtime.cpp builds on the reference's headline call pattern (768 tile invokes per iteration), 1-8 callers
# usage: gpu_replay_ab.sh <tag> "<name> ..."   (side libraries tpp-mlir_amd/build/libexp_q_<name>.so; "new" = the shipped one)
OUT=gpurun_out/${1:-replay_ab}; mkdir -p $OUT
LIBS=${2:-"new old"}
cp tpp-mlir_amd/libtpp_xsmm_runner_utils.so /tmp/new.so
nproc > $OUT/ab.txt
for rep in 1 2 3 4; do for thr in 1 2 4 8; do for lib in $LIBS; do
  if [ $lib = new ]; then cp /tmp/new.so tpp-mlir_amd/libtpp_xsmm_runner_utils.so; else cp tpp-mlir_amd/build/libexp_q_$lib.so tpp-mlir_amd/libtpp_xsmm_runner_utils.so; fi
  timeout 120 tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --bias --relu --tiles 32 --queue 1 -n 300 --threads $thr 2>&1 | tail -2 | head -1 | sed "s/^/lib=$lib thr=$thr /" | cut -c1-150 >> $OUT/ab.txt
done; done; done
cp /tmp/new.so tpp-mlir_amd/libtpp_xsmm_runner_utils.so
cat $OUT/ab.txt

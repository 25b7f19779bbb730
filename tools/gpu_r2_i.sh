#!/bin/bash
OUT=gpurun_out/${1:-r02i}; mkdir -p $OUT
for lib in main diag main diag; do
  L=$GRAFT_REPO_ROOT/tpp-mlir_amd/libtpp_xsmm_runner_utils.so; [ $lib = diag ] && L=$GRAFT_REPO_ROOT/tpp-mlir_amd/build/libexp_diag.so
  TPP_XSMM_LIBRARY=$L timeout 200 python tools/eltwise_bw.py 2>/dev/null | grep transpose | sed "s/^/$lib /" | tee -a $OUT/transpose_ab.txt
done

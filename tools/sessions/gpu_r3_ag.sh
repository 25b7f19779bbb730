#!/bin/bash
# eltwise / transpose stores: nt + plain (shipped) against sc1 write-through (tools/_e1) and sc1 nt (tools/_e2), same box
OUT=gpurun_out/r3_ag; mkdir -p $OUT
for lib in tpp-mlir_amd tools/_e1 tools/_e2 tpp-mlir_amd tools/_e1; do echo "lib=$lib"; TPP_XSMM_LIBRARY=$PWD/$lib/libtpp_xsmm_runner_utils.so python tools/eltwise_bw.py 2>/dev/null | cut -c1-100; done > $OUT/ab.txt; cat $OUT/ab.txt
TPP_XSMM_LIBRARY=$PWD/tools/_e1/libtpp_xsmm_runner_utils.so timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -x -k "unary or binary or transpose or eltwise" 2>&1 | tail -1

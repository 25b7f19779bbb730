#!/bin/bash
OUT=gpurun_out/r3_ah; mkdir -p $OUT
for lib in tpp-mlir_amd tools/_e1 tools/_e2; do echo "lib=$lib"; TPP_XSMM_LIBRARY=$PWD/$lib/libtpp_xsmm_runner_utils.so python tools/sessions/elt_policy_sizes.py 2>/dev/null; done > $OUT/sizes.txt; cat $OUT/sizes.txt

#!/bin/bash
# write-through (sc1) C stores in the bf16 kernels as well (shipped) against the library before (tools/_base: f32 kernels only), same box
OUT=gpurun_out/r3_ae; mkdir -p $OUT
timeout 600 python -m pytest tests/test_chain_gpu.py tests/test_parity_gpu.py tests/test_sharded_gpu.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -2
for i in 1 2; do for lib in tools/_base tpp-mlir_amd; do echo "lib=$lib"; LD_LIBRARY_PATH=$lib timeout 100 tools/mlp_probe 2>&1 | cut -c1-14,50-200; done; done > $OUT/ab.txt; cat $OUT/ab.txt
for lib in tools/_base tpp-mlir_amd; do echo "lib=$lib"; TPP_XSMM_LIBRARY=$lib/libtpp_xsmm_runner_utils.so python tools/sweep.py flatb 2>/dev/null | grep -E "^bf16" | head -6; done | tee $OUT/flatb.txt

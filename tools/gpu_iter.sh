#!/bin/bash
# iteration session: parity tests + timeline stamps + sweep. usage: gpurun -- 'bash tools/gpu_iter.sh tag'
TAG=${1:-it}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider -x -q 2>&1 | tail -15 > $OUT/pytest.log
tail -4 $OUT/pytest.log
[ -f tpp-mlir_amd/build/libabl_32_n2.so ] && TPP_XSMM_LIBRARY=$PWD/tpp-mlir_amd/build/libabl_32_n2.so python tools/stamp_case.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamp.log
[ -f tpp-mlir_amd/build/libabl_h32.so ] && TPP_XSMM_LIBRARY=$PWD/tpp-mlir_amd/build/libabl_h32.so python tools/stamp_bf16.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamp_bf16.log
timeout 600 python tools/sweep.py ${SWEEP_ARGS} 2>/dev/null | grep -E "^(f32|bf16)" > $OUT/sweep.log; cat $OUT/sweep.log

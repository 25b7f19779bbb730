#!/bin/bash
# timeline of every stamped side library (tools/build_stamp256.sh). usage: gpurun -- 'bash tools/gpu_stamp256.sh'
for L in tpp-mlir_amd/build/libstamp256_*.so; do
  echo "== $L"
  TPP_XSMM_LIBRARY=$PWD/$L python tools/stamp_bf16_256.py ${1:-4096} ${2:-4096} 2>&1 | grep -v amdgpu.ids
done

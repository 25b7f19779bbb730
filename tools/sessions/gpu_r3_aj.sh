#!/bin/bash
# VNNI-2 pack kernel: write-through stores (shipped) against plain stores (tools/_base), same box; parity of the pack tests
for i in 1 2; do for lib in tools/_base tpp-mlir_amd; do echo "lib=$lib"; TPP_XSMM_LIBRARY=$PWD/$lib/libtpp_xsmm_runner_utils.so python tools/sessions/pack_ab.py 2>/dev/null; done; done | tee gpurun_out/pack_ab.txt
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_chain_gpu.py -m gpu -q -p no:cacheprovider -x -k "vnni or pack or flat" 2>&1 | tail -1

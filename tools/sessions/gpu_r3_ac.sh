#!/bin/bash
# f32 loader-wave kernels: branch-free steady-state laps (new) against the library before (tools/_base), same box
OUT=gpurun_out/r3_ac; mkdir -p $OUT
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -2
for i in 1 2 3; do for lib in tools/_base tpp-mlir_amd; do echo "lib=$lib"; LD_LIBRARY_PATH=$lib timeout 100 tools/c2_probe --iters 2000 --init reference 2>&1 | tail -1 | cut -c1-220; LD_LIBRARY_PATH=$lib timeout 100 tools/c2_probe --c3 --iters 2000 --init reference 2>&1 | tail -1 | cut -c1-220; done; done > $OUT/ab.txt; cat $OUT/ab.txt

#!/bin/bash
# cache policy of the C stores of the f32 loader-wave kernels: sc1 (shipped, 16) against sc0+sc1 (17), sc1+nt (18), nt (2), none (tools/_base).
# The side builds: brgemm_f32_lw.hip compiled with -DTPP_C_STORE_AUX=17 / 18 / 2 (gemm_common.h) and linked with the shipped objects into tools/_v17 etc.
OUT=gpurun_out/r3_ad; mkdir -p $OUT
for i in 1 2; do for lib in tools/_base tpp-mlir_amd tools/_v17 tools/_v18 tools/_v2; do echo "lib=$lib"; LD_LIBRARY_PATH=$lib timeout 100 tools/c2_probe --iters 2000 --init reference 2>&1 | tail -1 | cut -c1-130; LD_LIBRARY_PATH=$lib timeout 100 tools/c2_probe --c3 --iters 2000 --init reference 2>&1 | tail -1 | cut -c1-130; done; done > $OUT/ab.txt; cat $OUT/ab.txt

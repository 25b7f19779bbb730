#!/bin/bash
# 128x128 bf16 tiles, same box: brgemm_bf16_dma128 (variant 17) against the loader-wave tile brgemm_bf16_lw<128x128> (variant 23) on the C4 layer and C5
OUT=gpurun_out/r3_af; mkdir -p $OUT
for i in 1 2 3; do for v in 17 23; do timeout 60 tools/mlp_probe --variant $v --only layers --rows 4096 2>&1 | cut -c1-14,50-160; done; done > $OUT/ab.txt
for i in 1 2 3; do python tools/sweep.py flatb 2>/dev/null | grep -E "^bf16" | head -6; done >> $OUT/ab.txt; cat $OUT/ab.txt

#!/bin/bash
# L2 warm-up touches by the MFMA waves at kernel start (TPP_HIP_BLW_L2PF=1, default) against none (=0), same box
OUT=gpurun_out/r3_aa; mkdir -p $OUT
timeout 300 python -m pytest tests/test_chain_gpu.py -m gpu -q -p no:cacheprovider -x > $OUT/pytest.log 2>&1; grep -E "passed|failed|error" $OUT/pytest.log | tail -2
for i in 1 2; do for pf in 0 1; do echo "l2pf=$pf"; TPP_HIP_BLW_L2PF=$pf timeout 100 tools/mlp_probe 2>&1 | cut -c1-14,50-200; 
TPP_HIP_BLW_L2PF=$pf timeout 100 tools/mlp_probe --variant 23 --only layers --rows 4096 2>&1 | cut -c1-14,50-200; done; done > $OUT/ab.txt; cat $OUT/ab.txt

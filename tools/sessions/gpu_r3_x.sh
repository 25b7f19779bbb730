#!/bin/bash
# where the chain kernel's skeleton spends its time: in-kernel stamps with the DMA and / or the MFMAs switched off (side builds)
OUT=gpurun_out/r3_x; mkdir -p $OUT
for rows in 512 1024; do for dbg in 0 16 32 48; do
  lib=tools/_abl; [ $((dbg & 32)) -ne 0 ] && lib=tools/_abl_nomath
  LD_LIBRARY_PATH=$lib TPP_HIP_CHAIN_DBG=$dbg TPP_HIP_CHAIN_STAMPS=$OUT/stamps_${rows}_$dbg.txt timeout 60 tools/mlp_probe --only chain --rows $rows --iters 50 > /dev/null 2>&1
  echo "== rows $rows dbg $dbg"; python tools/stamps_report.py $OUT/stamps_${rows}_$dbg.txt; done; done > $OUT/anatomy_abl.txt 2>&1
cat $OUT/anatomy_abl.txt

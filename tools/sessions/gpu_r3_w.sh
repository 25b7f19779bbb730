#!/bin/bash
# shipped library after the ablation switches became compile-time: chain / flat-B tests, the per-rank probe (shipped vs the
# ablation side build with dbg=0, same box), a fresh ablation table, eltwise bandwidth
OUT=gpurun_out/r3_w; mkdir -p $OUT
timeout 600 python -m pytest tests/test_chain_gpu.py tests/test_sharded_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do echo "shipped"; timeout 100 tools/mlp_probe 2>&1 | cut -c1-14,50-200; echo "abl build, dbg=0"; LD_LIBRARY_PATH=tools/_abl timeout 100 tools/mlp_probe 2>&1 | cut -c1-14,50-200; done > $OUT/ab.txt; cat $OUT/ab.txt
for dbg in 0 16 32 48 2 6 7; do echo "dbg=$dbg"; lib=tools/_abl; [ $((dbg & 32)) -ne 0 ] && lib=tools/_abl_nomath
  LD_LIBRARY_PATH=$lib TPP_HIP_CHAIN_DBG=$dbg timeout 100 tools/mlp_probe 2>&1; done > $OUT/chain_ablation.txt
python tools/eltwise_bw.py > $OUT/eltwise_bw.txt 2>/dev/null; tail -30 $OUT/eltwise_bw.txt

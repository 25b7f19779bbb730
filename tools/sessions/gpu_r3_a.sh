#!/bin/bash
# round 3, first GPU session: the new bf16 loader-wave tiles + the chain kernel: parity first (short timeouts: a hang must not eat the box),
# then the per-rank step probe and the shard-shape sweep with forced tiles
TAG=${1:-r3_a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_chain_gpu.py -x -q -p no:cacheprovider -k "lw_tiles" > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; tail -3 $OUT/t1.log
timeout 300 python -m pytest tests/test_chain_gpu.py -x -q -p no:cacheprovider -k "chunk_stream or pick" > $OUT/t2.log 2>&1; echo "rc=$?" >> $OUT/t2.log; tail -3 $OUT/t2.log
timeout 300 python -m pytest tests/test_chain_gpu.py -x -q -p no:cacheprovider -k "chain" > $OUT/t3.log 2>&1; echo "rc=$?" >> $OUT/t3.log; tail -5 $OUT/t3.log
timeout 120 tools/mlp_probe > $OUT/mlp_probe.txt 2>&1; cat $OUT/mlp_probe.txt
for v in 16 17 19 20 21 22 23; do timeout 60 tools/mlp_probe --variant $v --only layers --rows 256,512,1024,2048,4096 >> $OUT/mlp_probe_forced.txt 2>&1; done; cat $OUT/mlp_probe_forced.txt
timeout 200 python -m pytest tests/test_parity_gpu.py -x -q -p no:cacheprovider -k "beyond_2gib or bf16 or random" > $OUT/t4.log 2>&1; echo "rc=$?" >> $OUT/t4.log; tail -3 $OUT/t4.log

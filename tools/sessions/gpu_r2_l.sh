#!/bin/bash
# scheduler rework (per-caller rings merged by time stamp): tile-queue parity + threaded host test, A/B against the previous
# runtime (side library libexp_q_old.so), stress of the dependent chain
OUT=gpurun_out/${1:-r2_l}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_tile_queue_gpu.py tests/test_parity_gpu.py -q -x -k "queue or thread or host or resident or async" 2>&1 | tail -3 > $OUT/pytest.txt; cat $OUT/pytest.txt
bash tools/gpu_replay_ab.sh $(basename $OUT) "old new" > /dev/null 2>&1
g++ -std=c++17 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tpp-mlir_amd/csrc/runtime.cpp tests/tsan/fake_hip.cpp tests/tsan/driver.cpp -o /tmp/chain_ok -pthread -ldl && (time timeout 300 /tmp/chain_ok 50000) 2>&1 | tail -8 > $OUT/chain.txt; cat $OUT/chain.txt
for t in 1 2 4 8 16 64; do timeout 120 tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --bias --relu --tiles 32 --queue 1 -n 300 --threads $t 2>&1 | tail -2 | head -1 | sed "s/^/thr=$t /" | cut -c1-130; done > $OUT/threads.txt; cat $OUT/threads.txt

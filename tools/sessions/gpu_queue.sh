#!/bin/bash
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_tile_queue_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -12
for q in 0 1; do ./tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --tiles 32 --bias --relu --queue $q -n 100 --print 2>&1 | grep -v amdgpu.ids; done | tee $OUT/replay.log
./tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --whole-layer --bias --relu -n 1000 --print 2>&1 | grep -v amdgpu.ids | tee -a $OUT/replay.log
./tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --tiles 32 --queue 1 -n 100 2>&1 | grep -v amdgpu.ids | tee -a $OUT/replay.log
for q in 0 1; do ./tools/tpp_replay --c1 --queue $q -n 200 2>&1 | grep -v amdgpu.ids; done | tee -a $OUT/replay.log
./tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --tiles 32 --bias --relu --queue 1 -n 200 --bf16 --print 2>&1 | grep -v amdgpu.ids | tee -a $OUT/replay.log
for f in "" "--bf16"; do ./tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --tiles 64 --bias --relu --queue 1 -n 200 $f 2>&1 | grep -v amdgpu.ids | sed "s/packed tile invokes/packed tile invokes (tiles 64,64,64 $f)/" | tee -a $OUT/replay.log; done
for f in "" "--bf16"; do ./tools/tpp_replay --batch 256 --layers 768,768,768 --tiles 64,48,64 --bias --relu --queue 1 -n 200 $f 2>&1 | grep -v amdgpu.ids | sed "s/packed tile invokes/packed tile invokes (tiles 64,48,64 $f)/" | tee -a $OUT/replay.log; done
for th in 2 8; do OMP_PROC_BIND=close OMP_WAIT_POLICY=active ./tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --tiles 32 --bias --relu --queue 1 -n 200 --threads $th 2>&1 | grep -v amdgpu.ids | sed "s/packed tile invokes/packed tile invokes ($th OpenMP callers)/" | tee -a $OUT/replay.log; done

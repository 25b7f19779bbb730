#!/bin/bash
TAG=${1:-b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python bench.py --no-cpu-baseline > $OUT/bench_ref.json 2>/dev/null; cat $OUT/bench_ref.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('reference init:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['mlp']['value'], d['mlp']['ms_per_step'])"
python bench.py --no-cpu-baseline --init uniform > $OUT/bench_uni.json 2>/dev/null; cat $OUT/bench_uni.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('uniform init:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['mlp']['value'], d['mlp']['ms_per_step'])"
python bench.py --no-cpu-baseline --steps 1000 --warmup 100 > $OUT/bench_ref1k.json 2>/dev/null; cat $OUT/bench_ref1k.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('reference init 1000 steps:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['mlp']['value'], d['mlp']['ms_per_step'])"
TPP_XSMM_LIBRARY=$PWD/tpp-mlir_amd/build/libabl_32_n2.so python tools/stamp_case.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamp.log

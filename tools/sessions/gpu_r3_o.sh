#!/bin/bash
TAG=${1:-r3_o}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_peer_gather_gpu.py -x -q -s -p no:cacheprovider > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; grep -E "us per step|passed|failed|rc=" $OUT/t1.log
timeout 600 python bench.py --force-dist --steps 50 --warmup 10 --no-cpu-baseline --no-pmc > $OUT/bench_fd.json 2> $OUT/bench_fd.err; tail -3 $OUT/bench_fd.err
python - <<PY
import json
d=json.load(open("$OUT/bench_fd.json"))
m=d["mlp"]; print({k: m[k] for k in ("gather","ms_per_step","ms_per_step_compute_only","step_is_one_chain_launch")})
print(d["parity"]["reference"]["hip_vs_f64"], d["parity"]["reference"]["oracle_vs_f64"], d["parity"]["reference"]["pass"])
print(m.get("per_rank_step_us_native"))
PY

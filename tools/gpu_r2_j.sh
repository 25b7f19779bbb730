#!/bin/bash
OUT=gpurun_out/${1:-r02j}; mkdir -p $OUT
timeout 600 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err; echo "rc=$?"
python - <<'PY'
import json,sys
d=json.load(open("gpurun_out/%s/bench_forcedist.json" % sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r02j/bench_forcedist.json"))
print(d["value"], d["mlp"])
PY
tail -5 $OUT/bench_forcedist.err

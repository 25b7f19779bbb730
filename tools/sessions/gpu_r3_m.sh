#!/bin/bash
# parity tightening + bench line + MFMA-busy PMC passes for the C2 / C3 / C4-layer / chain kernels
TAG=${1:-r3_m}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -p no:cacheprovider -s -k "c2_full or c3_fused or c4_mlp or zz_report" > $OUT/t1.log 2>&1; echo "rc=$?" >> $OUT/t1.log; grep -E "parity|passed|failed|rc=|Error" $OUT/t1.log | tail -8
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; tail -3 $OUT/bench_steps20.err; python - <<PY
import json
d=json.load(open("$OUT/bench_steps20.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "busy", d["roofline"].get("mfma_busy"))
print("parity", {k: (v["hip_vs_f64"], v["oracle_vs_f64"], v["pass"]) for k, v in d["parity"].items()})
print("mlp", d["mlp"]["ms_per_step"], d["mlp"].get("per_rank_step_us"))
print("native", d["mlp"].get("per_rank_step_us_native"))
PY
cd /tmp
for what in "c2:$R/tools/c2_probe --iters 40 --init reference" "c3:$R/tools/c2_probe --c3 --iters 40 --init reference" "c4layer:$R/tools/mlp_probe --rows 4096 --only layers --iters 40" "c4chain:$R/tools/mlp_probe --rows 4096,512 --only chain --iters 40"; do
  tag=${what%%:*}; cmd=${what#*:}
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d /tmp/mf_$tag -o p -- $cmd > /dev/null 2>&1
  f=$(find /tmp/mf_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$tag" >> $R/$OUT/mfma_busy.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:70], r["Counter_Name"], r.get("Grid_Size", ""))
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, cn, g), (v, n) in sorted(acc.items()):
    if "brgemm" in kn:
        print("%-8s %-72s grid %-9s %-28s mean %.0f over %d launches" % (sys.argv[2], kn, g, cn, v / n, n))
PY
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o p -- $cmd > /dev/null 2>&1
  f=$(find /tmp/ks_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep brgemm $f | cut -c1-160 | sed "s/^/$tag /" >> $R/$OUT/mfma_busy.txt
done
cat $R/$OUT/mfma_busy.txt

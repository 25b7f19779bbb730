#!/bin/bash
# L2 counters of the loader-wave layer kernels: hit rate, requests, fabric reads per launch (is the K loop fed from L2 hits?)
OUT=gpurun_out/r3_z; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $OUT/tcc_counters.txt
cd /tmp
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_READ_sum TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/l2_$tag -o p -- $R/tools/mlp_probe --rows 512,1024,2048,4096 --only layers --iters 20 > /dev/null 2>$R/$OUT/err_$tag.txt
  f=$(find /tmp/l2_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $R/$OUT/l2_counters.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:60], r.get("Grid_Size", ""), r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, g, cn), (v, n) in sorted(acc.items()):
    if "brgemm" in kn:
        print("%-62s grid %-8s %-22s mean %.0f over %d launches" % (kn, g, cn, v / n, n))
PY
done
cat $R/$OUT/l2_counters.txt; tail -3 $R/$OUT/err_*.txt | head -20

#!/bin/bash
# PMC counters of the bf16 kernels on the C4 layer / 4096^3 (separate passes, --kernel-trace only).
# usage: gpurun -- 'bash tools/gpu_pmc_bf16.sh'
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bf16; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$i -o p -- python $GRAFT_REPO_ROOT/tools/sweep.py bf16 > $OUT/run_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $OUT/summary.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:60], r["Counter_Name"], r.get("Grid_Size", ""))
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, cn, g), (v, n) in sorted(acc.items()):
    if "brgemm" in kn:
        print("%-62s grid %-9s %-34s mean %.0f over %d launches" % (kn, g, cn, v / n, n))
PY
done
cat $OUT/summary.txt

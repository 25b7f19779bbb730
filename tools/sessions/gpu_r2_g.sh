#!/bin/bash
OUT=gpurun_out/${1:-r02g}; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|Counter_Name)\s*:\s*\w+|^\w+_\w+" | head -0
rocprofv3 -L > $GRAFT_REPO_ROOT/$OUT/counters.txt 2>&1
grep -ciE "TCC_|TCP_|TA_|SQ_|TD_" $GRAFT_REPO_ROOT/$OUT/counters.txt
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_READ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_DATA_STALL_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$i -o p -- python $GRAFT_REPO_ROOT/tools/abl_case_bf16.py pmc > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$i.err
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $GRAFT_REPO_ROOT/$OUT/pmc_summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "dma128" in r["Kernel_Name"]:
        acc[(r.get("Grid_Size", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (g, c), v in sorted(acc.items()):
    print("grid %-8s %-44s mean %16.1f over %d launches" % (g, c, sum(v) / len(v), len(v)))
PY
done
cat $GRAFT_REPO_ROOT/$OUT/pmc_summary.txt
tail -2 $GRAFT_REPO_ROOT/$OUT/pmc_1.err

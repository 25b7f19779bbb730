#!/bin/bash
# anatomy of the C2 kernel's duration: rocprofv3 kernel time against the number of 64-k chunks (batch count 0..16)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r2_q}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for br in 0 1 2 4 8 12 16; do
  rm -rf /tmp/prof_br$br
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_br$br -o r -- $GRAFT_REPO_ROOT/tools/c2_probe --iters 300 --br $br > /dev/null 2>&1
  echo -n "br=$br " >> $OUT/c2_anatomy.txt
  find /tmp/prof_br$br -name "*kernel_stats.csv" -exec grep brgemm_f32 {} \; | cut -c1-160 >> $OUT/c2_anatomy.txt
done
cat $OUT/c2_anatomy.txt

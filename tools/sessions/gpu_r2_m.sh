#!/bin/bash
# kernel time of the tile-queue launches on the reference's headline call pattern (rocprofv3 --kernel-trace --stats over tools/tpp_replay)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r2_m}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for cfg in "--tiles 32" "--tiles 32 --bf16" "--tiles 64" "--whole-layer"; do
  tag=$(echo $cfg | tr -d ' -')
  rm -rf /tmp/prof_$tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o r -- $GRAFT_REPO_ROOT/tools/tpp_replay --batch 256 --layers 1024,1024,1024,1024 --bias --relu $cfg --queue 1 -n 300 > $OUT/run_$tag.txt 2>&1
  echo "== $cfg" >> $OUT/kernel_stats.txt
  tail -2 $OUT/run_$tag.txt | head -1 | cut -c1-160 >> $OUT/kernel_stats.txt
  find /tmp/prof_$tag -name "*kernel_stats.csv" -exec head -4 {} \; | cut -c1-200 >> $OUT/kernel_stats.txt
done
cat $OUT/kernel_stats.txt

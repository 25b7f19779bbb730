#!/bin/bash
# Round deliverable session (round 6 adds: the host-buffer rows (host cache), the bf16 tile-family sweep, the probes behind the round's decisions; round 5 added: refbench table, split sweep, the N = 2 rig bench, PMC of the skinny-shape kernels): full parity suite, smoke, the driver's bench command, rocprofv3 kernel stats of the SAME bench
# command, default bench, MFMA-busy PMC passes, the per-rank MLP step probe + in-kernel stamps, shape sweeps, tile-queue replays,
# eltwise bandwidth. Results in gpurun_out/<tag>/.   usage: gpurun --timeout 2400 -- 'bash tools/gpu_official.sh r03_official1'
TAG=${1:-official}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|error|\[parity\]|rc=|us per step" $OUT/pytest_gpu.log | tail -12
timeout 600 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $R/$OUT/rocprof_stats_run.json 2> $R/$OUT/rocprof_stats.err )
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/rocprof_kernel_stats.csv \;
# MFMA-busy passes (PMC alone) + kernel stats of the same commands: C2, C3, the C4 layer kernel, the chain kernels
( cd /tmp
printf -- "--batch 128 --layers 1024,1024 --kernel args --whole-layer -n 40\n--batch 128 --layers 4096,1024 --kernel args --whole-layer -n 40\n--batch 128 --layers 2304,768 --kernel args --tiles 64,48,64 -n 40\n" > /tmp/skinny_cases.txt
for what in "c2:$R/tools/c2_probe --iters 40 --init reference" "c3:$R/tools/c2_probe --c3 --iters 40 --init reference" "c4layer:$R/tools/mlp_probe --rows 4096 --only layers --iters 40" "c4chain:$R/tools/mlp_probe --rows 4096,2048,1024,512 --only chain --iters 40" "skinny:$R/tools/tpp_replay --cases /tmp/skinny_cases.txt"; do
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
  f=$(find /tmp/ks_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep brgemm $f | cut -c1-170 | sed "s/^/$tag /" >> $R/$OUT/mfma_busy.txt
done )
# the per-rank step of the row-sharded C4 MLP (one chain launch vs three launches), forced tiles, and the chain kernel's phases
for i in 1 2 3; do timeout 100 tools/mlp_probe; done > $OUT/mlp_probe.txt 2>&1
for v in 16 17 19 20 21 22 23; do timeout 60 tools/mlp_probe --variant $v --only layers --rows 256,512,1024,2048,4096; done > $OUT/mlp_probe_forced.txt 2>&1
[ -f tools/_abl/libtpp_xsmm_runner_utils.so ] && for rows in 512 1024 2048 4096; do
  LD_LIBRARY_PATH=tools/_abl TPP_HIP_CHAIN_STAMPS=$OUT/stamps_${rows}.txt timeout 60 tools/mlp_probe --only chain --rows $rows --iters 50 > /dev/null 2>&1
  echo "== rows $rows"; python tools/stamps_report.py $OUT/stamps_${rows}.txt; done > $OUT/chain_anatomy.txt 2>&1
# timing with parts of the chain kernel switched off: side builds only (python tpp-mlir_amd/build.py --ablation BEFORE the gpurun call;
# the shipped library has no such switch). 32 / 48 need the build with the fragment reads and MFMAs compiled out as well.
if [ -f tools/_abl/libtpp_xsmm_runner_utils.so ]; then
for dbg in 0 16 32 48 2 6 7; do echo "dbg=$dbg"; lib=tools/_abl; [ $((dbg & 32)) -ne 0 ] && lib=tools/_abl_nomath
  LD_LIBRARY_PATH=$lib TPP_HIP_CHAIN_DBG=$dbg timeout 100 tools/mlp_probe 2>&1; done > $OUT/chain_ablation.txt
fi
# round 5: the reference's benchmark shape set (full table + json), the split / half-width-tile sweep, the self-launching N = 2 bench on the one-device rig
timeout 900 python bench.py --refbench --refbench-json $OUT/refbench.json > $OUT/refbench.txt 2> $OUT/refbench.err
timeout 300 python tools/split_sweep.py > $OUT/split_sweep.txt 2>&1
# benchmarks/mlir/*.mlir (base/mha.json, pack.json) as xsmm call scripts: queue on / off, transpose folding on / off, 8 callers
timeout 300 tools/tpp_replay --cases tools/scripts.cases 2>&1 | grep -v "^[0-9.e-]*$" > $OUT/mlir_scripts.txt
TPP_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 3 > $OUT/bench_gpus2_rig.json 2> $OUT/bench_gpus2_rig.err
python tools/sweep.py 2>/dev/null | grep -E "^(f32|bf16)" > $OUT/sweep.txt
python tools/sweep.py big 2>/dev/null | grep -E "^(f32|bf16)" >> $OUT/sweep.txt
python tools/sweep.py shards 2>/dev/null | grep -E "^bf16" >> $OUT/sweep.txt
python tools/sweep.py small 2>/dev/null | grep -E "^bf16" >> $OUT/sweep.txt
python tools/sweep.py flatb 2>/dev/null | grep -E "^bf16" > $OUT/flatb.txt
tools/ubench/fillmix.out > $OUT/fill_paths.txt 2>&1
tools/ubench/tr16_probe.out > $OUT/tr16_probe.txt 2>&1
timeout 200 python tools/queue_fuzz.py 45 5 2>&1 | tail -1 > $OUT/queue_fuzz.txt
bash tools/gpu_replay.sh $TAG > /dev/null 2>&1
# round 6: host buffers (the unmodified harness: plain malloc'ed operands, modes from the environment), bf16 sweep, probes
timeout 600 python tools/bf16_sweep.py -n 300 > $OUT/bf16_sweep.txt 2> $OUT/bf16_sweep.err
bash tools/gpu_host_buffers.sh > $OUT/host_buffers.txt 2>&1
timeout 120 tools/ubench/pageable_probe.out > $OUT/pageable_probe.txt 2>&1
g++ -O2 -std=c++17 tools/ubench/wp_async_probe.cpp -o /tmp/wp_async_probe -include malloc.h && /tmp/wp_async_probe > $OUT/wp_async_probe.txt 2>&1
timeout 200 tools/ubench/split_handoff.out > $OUT/split_handoff.txt 2>&1
echo "# matmul_128x768x768 as tile invokes (32,64,64), 50 timed loops of 200 calls each (VERDICT r5 weak 9: the 51 us row)" > $OUT/outlier.txt
timeout 300 tools/tpp_replay --batch 128 --layers 768,768 --kernel args --tiles 32,64,64 --queue 1 -n 200 --repeats 50 2>&1 | grep -v "^[0-9.e-]*$" >> $OUT/outlier.txt
# round 6, second half: the launch thread and the quads, on / off on this box (short forms of profiles/r06_launch_thread_ab.txt, r06_bf16_quads_ab.txt)
{
R=tools/tpp_replay; M="--batch 256 --layers 1024,1024,1024,1024 --tiles 32 --bias --relu"
for lt in 0 1 0 1; do echo "### TPP_HIP_LAUNCH_THREAD=$lt"; export TPP_HIP_LAUNCH_THREAD=$lt
  $R $M --queue 1 -n 300 --repeats 5 2>&1 | grep "repeats" | sed 's/^/MLP tiles f32:  /'
  $R $M --bf16 --queue 1 -n 300 --repeats 5 2>&1 | grep "repeats" | sed 's/^/MLP tiles bf16: /'
  for s in mha_qk mha_sv pack_a unpack_c; do $R --script $s --queue 1 -n 1000 2>&1 | grep "mean" | cut -c1-130; done
done; unset TPP_HIP_LAUNCH_THREAD
for q in 0 1 0 1; do echo "### TPP_HIP_BF16_QUADS=$q"
  for v in 2 4; do TPP_HIP_BF16_QUADS=$q $R --batch 1024 --layers 1024,2560 --tiles 64,64,64 --bf16 --vnni $v --queue 1 -n 300 --repeats 3 2>&1 | grep "repeats\|mean" | cut -c1-230; done
done
$R --batch 1024 --layers 1024,2560 --whole-layer --bf16 -n 300 --repeats 3 2>&1 | grep "repeats\|mean" | cut -c1-230
} > $OUT/launch_thread_and_quads.txt 2>&1
python tools/eltwise_bw.py > $OUT/eltwise_bw.txt 2>/dev/null
python tools/vendor_compare.py > $OUT/vendor_compare.txt 2>/dev/null
tail -n 1 $OUT/bench_steps20.json | head -c 900; echo; head -8 $OUT/rocprof_kernel_stats.csv | cut -c1-160; cat $OUT/mlp_probe.txt | cut -c1-14,50-200 | tail -4

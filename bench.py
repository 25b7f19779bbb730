#!/usr/bin/env python3
"""bench.py - the hot path of BASELINE.json on MI355X.

Step = one pass of the hot path over one batch of synthetic input:
  primary workload (value/metric): BASELINE config[1] "BRGEMM 1024x1024x1024 fp32,
    batch-reduce=16": ONE xsmm_brgemm_invoke, C[1024x1024] += sum_{b<16} A_b[1024x64] B_b[64x1024]
    (dispatch m=n=1024 k=64 lda=ldb=ldc=1024 stride_a=64 stride_b=65536).
  secondary ("mlp" object in the same JSON line): BASELINE config[3] 3-layer MLP
    1024->1024->1024->1024 bf16 bs=4096 (bias+relu fused).
  With --gpus N > 1 the roles swap: the headline (`value`) is the STRONG-scaled MLP - tile rows sharded across the
    ranks, the all-gather of the output INSIDE the timed step, both gather paths (peer-store over IPC-mapped buffers
    and RCCL all_gather_into_tensor) timed in the same run, the gathered output compared bit for bit with the
    unsharded result on every rank (`mlp.gathered_bit_identical`; the run exits 3 if it is false), the one-GPU step of
    the same run next to it (`mlp.one_gpu_same_run`). The weak-scaled C2 (one independent BRGEMM per GPU, no collective:
    the reference has no exchange step on this path) is reported under `c2_weak`.
Timing: inputs resident in HBM, W warm-up steps, then exactly K steps between
barrier+synchronize pairs, max over ranks (nothing but the K invokes inside the wall-clock region); the
HIP-event pair for the kernel-side time brackets an immediate repeat of the same K steps (see timed()). FLOPs are the reference's BENCH_TOTAL_FLOPS
arithmetic (tools/mlir-gen/MLIRGen.cpp:313-334): 2*m*n*k*br for the BRGEMM.
Launch: python bench.py --gpus N (N > 1 without WORLD_SIZE: bench.py starts its own N ranks under torch.distributed.run)
        | python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N (WORLD_SIZE must equal --gpus)
"""
import argparse
import glob
import importlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, Chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0
F32, BF16 = 1, 2


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mlp", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="also time a hipGraph replay of the step and report the faster of the two launch modes")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the collectives even with one rank (exercises the N>1 code path)")
    ap.add_argument("--init", choices=["reference", "uniform"], default="reference",
                    help="input stream `value` is quoted on. reference: tpp-run's normal init stream (what the reference "
                         "benchmarks run on); uniform: U[-1,1) (sign cancellation, highest switching power). The OTHER "
                         "stream is always measured too and reported as roofline_uniform / roofline_reference")
    ap.add_argument("--gather", choices=["peer", "rccl"], default="peer",
                    help="all-gather of the MLP output at N > 1: peer = every rank stores its rows into every peer's buffer through "
                         "IPC-mapped pointers (csrc/peer_gather.hip), falling back to RCCL if the buffers cannot be mapped; rccl = "
                         "dist.all_gather_into_tensor")
    ap.add_argument("--refbench", action="store_true",
                    help="N = 1 only: print the reference's whole benchmark shape set (tools/refbench.py: 222 rows, f32 / bf16 VNNI-2 / VNNI-4, tile "
                         "invokes and whole layer) with the CPU port's figure per shape beside it (this file's cpu_baseline leg) and exit: what "
                         "profiles/r05_refbench.txt is made with")
    ap.add_argument("--refbench-json", default="", help="with --refbench: also write the rows as JSON")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not run the rocprofv3 PMC passes for roofline.traffic (use when bench.py itself runs under a profiler)")
    return ap.parse_args()


def timed(fn, steps, sync, barrier):
    """exactly `steps` calls of fn between barrier+sync pairs; returns (wall seconds, device seconds).
    Two passes over the same `steps` launches, back to back:
      wall   - perf_counter around [sync, steps x fn, sync]: nothing else in the region. (A HIP event pair
               around 20 launches costs ~30 us here - two marker packets, ~8 + 6 us of host time to record
               them, and a slow first synchronize after an event query; profiles/r02_timing_anatomy.txt -
               which is 8 % of a 20-step run of an 18 us kernel.)
      device - torch.cuda.Event pair on the launch stream around a repeat of the same region: the
               kernel-side time the roofline figure is computed from."""
    import torch
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    wall = time.perf_counter() - t0
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    sync()
    return wall, e0.elapsed_time(e1) * 1e-3


def warm(fn, steps, sync):
    """untimed warm-up: W steps, with two launch->synchronize cycles so that lazy module loading,
    clock ramp-up and the runtime's first blocking wait are all outside the timed region"""
    for _ in range(max(1, steps // 2)):
        fn()
    sync()
    for _ in range(steps - max(1, steps // 2)):
        fn()
    sync()


def spin_up(fn, sync, seconds=0.06):
    """steady-state preparation, before the W warm-up steps and whatever W is: the chip needs tens of
    milliseconds of load before its clocks settle (a 20 us kernel timed cold reads 10 % low)"""
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(200):
            fn()
        n += 200
        sync()
    return n


def graph_of(fn, warm=3):
    """capture one step into a hipGraph (launch-bound inner loop); returns replay callable or None"""
    import torch
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        pkg = importlib.import_module("tpp-mlir_amd")
        rt = pkg.get_runtime()
        with torch.cuda.stream(s):
            rt.set_stream(s)
            for _ in range(warm):
                fn()
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
                fn()
        rt.set_stream(None)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        return g.replay
    except Exception as ex:  # capture is an optimisation of the harness, not of the kernel
        sys.stderr.write("[bench] hipGraph capture unavailable (%s); timing plain launches\n" % ex)
        try:
            importlib.import_module("tpp-mlir_amd").get_runtime().set_stream(None)
        except Exception:
            pass
        return None


def committed_traffic(kernel_substr):
    """fallback: HBM-side bytes per launch from the newest committed rocprofv3 PMC summary (profiles/*_pmc_hbm.txt)"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.txt")))
    fetch = write = None
    for line in (open(files[-1]) if files else []):
        if kernel_substr in line:
            m = re.search(r"mean\s+([0-9.]+)", line)
            if m and "FETCH_SIZE" in line:
                fetch = float(m.group(1))
            if m and "WRITE_SIZE" in line:
                write = float(m.group(1))
    if fetch is None or write is None:
        return None, None
    return (2.0 * fetch + write) * 1024.0, os.path.basename(files[-1])


def live_traffic(kernel_substr, init):
    """HBM-side bytes per launch of the C2 kernel, measured NOW: two rocprofv3 passes (one counter each, as
    MI355X_MICROARCH.md prescribes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2) over tools/c2_probe,
    the native twin of the timed step. Units KiB; on gfx950 FETCH_SIZE counts half of a 16 B/lane streaming
    read (same guide, HBM section) -> doubled. Returns (bytes, detail) or (None, reason)."""
    probe = os.path.join(ROOT, "tools", "c2_probe")
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not (os.path.exists(probe) and os.path.exists(rocprof)):
        return None, "tools/c2_probe or rocprofv3 missing"
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="tpp_pmc_", dir="/tmp")
        try:
            r = subprocess.run([rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "c2", "--",
                                probe, "--iters", "40", "--init", init], cwd="/tmp", env=env, capture_output=True,
                               text=True, timeout=240)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, "rocprofv3 --pmc %s produced no counter file (rc %d)" % (counter, r.returncode)
            import csv
            v = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                 if kernel_substr in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter]
            if not v:
                return None, "no %s rows for %s" % (counter, kernel_substr)
            vals[counter] = sum(v) / len(v)
        except Exception as ex:  # a profiler problem must not cost the bench line
            return None, "rocprofv3 --pmc %s failed: %s" % (counter, ex)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, {
        "FETCH_SIZE_KiB_raw": round(vals["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(vals["WRITE_SIZE"], 1),
        "correction": "FETCH_SIZE x 2 (gfx950 counts 64 B per 128 B request)"}


def live_mfma_busy(kernel_substr, init, kernel_us, cmd=None):
    """MFMA utilisation of the dominant kernel, measured NOW: a third rocprofv3 pass (SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES +
    GRBM_GUI_ACTIVE; --pmc alone, as the guide prescribes) over tools/c2_probe. SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy
    cycles summed over the chip's SIMDs (= 64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_bf16: checked against
    the instruction count of the launch); busy = that / (SIMDs x shader cycles of the launch). The shader cycles of a launch are
    not a counter: GRBM_GUI_ACTIVE (per launch, max over the XCDs' instances as rocprofv3 reports the sum: / 8) is the chip's
    active time in shader clocks, so clock = GRBM cycles / kernel time and busy x clock / 2.4 GHz is what reconciles with
    roofline.frac (peak is quoted at 2.4 GHz; under load the chip runs below it)."""
    probe = cmd or [os.path.join(ROOT, "tools", "c2_probe"), "--iters", "40", "--init", init]
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not (os.path.exists(probe[0]) and os.path.exists(rocprof)):
        return None
    d = tempfile.mkdtemp(prefix="tpp_pmc_", dir="/tmp")
    try:
        subprocess.run([rocprof, "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "--output-format", "csv",
                        "-d", d, "-o", "mf", "--"] + probe, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True,
                       text=True, timeout=240)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            return {"error": "rocprofv3 produced no counter file"}
        import csv
        acc = {}
        for row in csv.DictReader(open(files[0])):
            if kernel_substr in row.get("Kernel_Name", ""):
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in acc:
            return {"error": "no counter rows for " + kernel_substr}
        mean = {k_: sum(v) / len(v) for k_, v in acc.items()}
        simds = 256 * 4
        out = {"SQ_VALU_MFMA_BUSY_CYCLES": round(mean["SQ_VALU_MFMA_BUSY_CYCLES"]), "SQ_BUSY_CYCLES": round(mean.get("SQ_BUSY_CYCLES", 0)),
               "GRBM_GUI_ACTIVE": round(mean.get("GRBM_GUI_ACTIVE", 0)), "launches": len(acc["SQ_VALU_MFMA_BUSY_CYCLES"]),
               "mfma_busy_cycles_per_simd": round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / simds, 1)}
        # shader cycles of one launch: SQ_BUSY_CYCLES is summed over the 32 shader engines, GRBM_GUI_ACTIVE over the 8 XCDs
        cyc = mean.get("SQ_BUSY_CYCLES", 0) / 32.0
        if cyc > 0:
            out["launch_cycles_sq"] = round(cyc)
            out["mfma_busy"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / simds / cyc, 4)
            if kernel_us:
                out["clock_ghz_under_profiler"] = round(cyc / kernel_us * 1e-3, 3)
                out["frac_reconciled"] = round(out["mfma_busy"] * out["clock_ghz_under_profiler"] / 2.4, 4)
        if mean.get("GRBM_GUI_ACTIVE"):
            out["launch_cycles_grbm"] = round(mean["GRBM_GUI_ACTIVE"] / 8.0)
        out["note"] = ("mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32 shader engines): share of the launch's own "
                       "shader cycles in which a SIMD's matrix pipe is busy; frac_reconciled = mfma_busy x (those cycles / the kernel time "
                       "of the timed run) / 2.4 GHz, to be read next to roofline.frac")
        return out
    except Exception as ex:  # a profiler problem must not cost the bench line
        return {"error": str(ex)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def per_launch_figures(init):
    """tools/c2_probe: every launch timed on its own (launch + device synchronisation), mean +- population
    stdev - the timing style of the reference's stand-alone GPU baseline (tools/bench-ref/GPU/cuda/MatmulRef.cpp:56-63,
    tools/bench-ref/include/Bench.h:66-77) - next to the loop mean of tpp-run's definition"""
    probe = os.path.join(ROOT, "tools", "c2_probe")
    if not os.path.exists(probe):
        return None
    try:
        r = subprocess.run([probe, "--iters", "300", "--init", init], capture_output=True, text=True, timeout=120)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as ex:
        return {"error": str(ex)}


def parity_figures(got, A, B, C0, m, n, k, br):
    """HIP result of one C2 invoke against the oracle on the same inputs.
    normwise = max|d| / max(1, max|ref|) is the north-star figure (<= 1e-5). Element-wise (SURVEY.md 8d's second
    criterion): |d| <= 1e-5 |ref| + K eps sum_k |a_ik||b_kj| - the relative bar plus the a-priori f32 dot-product
    floor (the two sides sum in different orders; where the products cancel, |ref| is far below the summands);
    elementwise_bar_used is the worst |d| / (that bound), <= 1 passes. max_rel is the plain relative error over the
    elements that are not cancelled (|ref| >= 1 % of max|ref|)."""
    from oracle import pyoracle as orc
    ref = C0.copy()
    orc.fused_brgemm_omp(F32, m, n, k, 1024, 1024, 1024, k, k * 1024, 4, 0, 0, A, B, ref, None, br)
    mag = np.zeros_like(ref)
    orc.fused_brgemm_omp(F32, m, n, k, 1024, 1024, 1024, k, k * 1024, 4, 0, 0, np.abs(A), np.abs(B), mag, None, br)
    r = ref.astype(np.float64)
    d = np.abs(got.astype(np.float64) - r)
    floor = (k * br) * 2.0 ** -24 * mag.astype(np.float64)
    big = np.abs(r) >= 1e-2 * np.abs(r).max()
    normwise = float(d.max() / max(1.0, np.abs(r).max()))
    used = float((d / (1e-5 * np.abs(r) + floor)).max())
    # an fp64 truth: the two f32 results differ by their summation orders; neither is privileged - how far is EACH from fp64?
    K = k * br
    t = (A.astype(np.float64).reshape(m, K) @ B.astype(np.float64).reshape(K, n)).reshape(-1)  # BETA_0: C0 is not read
    tbig = np.abs(t) >= 1e-2 * np.abs(t).max()
    scale = max(1.0, float(np.abs(t).max()))

    def vs_truth(x):
        e = np.abs(x.astype(np.float64) - t)
        return {"normwise": float(e.max() / scale), "max_rel": float((e[tbig] / np.abs(t[tbig])).max())}
    hip64, orc64 = vs_truth(got), vs_truth(ref)
    return {"max_abs": float(d.max()), "max_ref": float(np.abs(r).max()), "normwise": normwise,
            "max_rel": float((d[big] / np.abs(r[big])).max()), "elementwise_bar_used": used,
            "frac_within_1e-5_rel": float((d <= 1e-5 * np.abs(r)).mean()),
            "hip_vs_f64": hip64, "oracle_vs_f64": orc64,
            "criterion": "normwise <= 1e-5 and |d| <= 1e-5*|ref| + K*eps*sum|a||b| element-wise (elementwise_bar_used <= 1); "
                         "hip_vs_f64 <= 2 x oracle_vs_f64 + 2^-22 (normwise) - as close to an fp64 truth as the oracle is",
            "pass": bool(normwise <= 1e-5 and used <= 1.0 and hip64["normwise"] <= 2.0 * orc64["normwise"] + 2.0 ** -22)}


REFBENCH_SHAPES = [(1024, 1024, 512), (1024, 2560, 1024), (1024, 352, 512), (1024, 512, 256), (128, 1024, 1024), (128, 1024, 4096), (128, 3072, 768),
                   (128, 4096, 1024), (128, 768, 2304), (128, 768, 3072), (128, 768, 768), (256, 1024, 1024), (256, 1024, 4096), (256, 3072, 768),
                   (256, 4096, 1024), (256, 768, 3072), (256, 768, 768)]  # (M, N, K) of benchmarks/config/matmul/*.json = fc/*.json (tools/refbench.py)


def cpu_refbench_rows(seconds_per_shape=0.5):
    """cpu_baseline leg, second part: the CPU port (oracle/cpu_baseline.c: the reference's packed 32x32x32 call structure under OpenMP,
    the container's usable CPUs) on every (M, N, K) of the reference's benchmark shape set - the column beside the GPU rows of
    tools/refbench.py. ~10 s in all. Returns {"MxNxK": {"gflops", "us", "threads"}}."""
    from oracle import pyoracle as orc
    cb = orc.CpuBaseline(native=True)
    cpus, _ = orc.usable_cpus()
    team = cb.set_threads(cpus)
    rng = np.random.default_rng(5)
    res = {}
    for (M, N, K) in REFBENCH_SHAPES:
        A = rng.uniform(-1, 1, M * K).astype(np.float32)
        B = rng.uniform(-1, 1, K * N).astype(np.float32)
        Ap, Bp, Cp = cb.pack(A, B, np.zeros(M * N, np.float32), M, N, K)
        cb.run(M, N, K, Ap, Bp, Cp, True, 2)
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds_per_shape:
            cb.run(M, N, K, Ap, Bp, Cp, True, 10)
            reps += 10
        el = time.perf_counter() - t0
        res["%dx%dx%d" % (M, N, K)] = {"gflops": round(2.0 * M * N * K * reps / el / 1e9, 1), "us": round(el / reps * 1e6, 2), "threads": team}
    return res


def refbench_mode(args):
    """--refbench: the whole table, CPU column included; prints it and exits"""
    cpu = cpu_refbench_rows()
    with tempfile.TemporaryDirectory() as td:
        cj = os.path.join(td, "cpu.json")
        with open(cj, "w") as f:
            json.dump(cpu, f)
        cmd = [sys.executable, os.path.join(ROOT, "tools", "refbench.py"), "-n", "300", "--cpu-json", cj]
        if args.refbench_json:
            cmd += ["--json", args.refbench_json]
        return subprocess.call(cmd)


def cpu_baseline(seconds, A, B, C):
    """The CPU row: the reference's packed 32x32x32-tile batch-reduce call structure under OpenMP
    (oracle/cpu_baseline.c, compiled here with -O3 -march=native) on the same C2 inputs, checked against the
    oracle once; libxsmm itself is not in the image. The naive oracle loop and torch's CPU GEMM are context."""
    from oracle import pyoracle as orc
    m = n = k = 1024
    flops = 2.0 * m * n * k
    cb = orc.CpuBaseline(native=True)
    # the team: the CPUs this container may really keep busy (affinity AND cgroup quota). Round 3 ran the default team of 128
    # threads on a box whose container is limited to 16 CPUs' worth of time: the scheduler throttled it to 0.79 TFLOP/s, a
    # sixth of what 16 threads sustain.
    cpus, cpu_detail = orc.usable_cpus()
    team = cb.set_threads(cpus)
    Ap, Bp, Cp = cb.pack(A, B, C, m, n, k)
    cb.run(m, n, k, Ap, Bp, Cp, True, 1)  # warm + check
    ref = C.copy()
    orc.fused_brgemm_omp(F32, 1024, 1024, 64, 1024, 1024, 1024, 64, 65536, 4, 0, 0, A, B, ref, None, 16)
    ok = bool(np.abs(cb.unpack_c(Cp, m, n) - ref).max() <= 1e-5 * max(1.0, float(np.abs(ref).max())))

    def sustained(sec):
        reps, t0 = 0, time.perf_counter()
        while True:
            cb.run(m, n, k, Ap, Bp, Cp, True, 20)
            reps += 20
            el = time.perf_counter() - t0
            if el >= sec or reps >= 400000:
                return flops * reps / el / 1e9, reps, el
    tiled, reps, el = sustained(seconds)
    # the reference's paper machines ran 16 threads (scripts/benchmarks/README.md:11-17): that team size too, when it differs
    t16 = None
    if team != 16 and cpus >= 16:
        cb.set_threads(16)
        t16 = round(sustained(min(3.0, seconds / 3))[0], 1)
    # context: one socket's 64 cores in BURSTS that stay inside the cgroup's budget (20 passes, then a pause): what the host's cores
    # can do when the container's quota is not the limit - not a sustained figure, and labelled so
    burst = None
    if cpu_detail["affinity_cpus"] >= 64 and cpus < 64:
        cb.set_threads(64)
        best = 0.0
        for _ in range(4):
            time.sleep(0.4)
            t0 = time.perf_counter()
            cb.run(m, n, k, Ap, Bp, Cp, True, 10)
            best = max(best, flops * 10 / (time.perf_counter() - t0) / 1e9)
        burst = {"threads": 64, "gflops": round(best, 1), "note": "bursts of 10 passes with pauses (inside the cgroup's CPU budget): NOT sustained"}
    cb.set_threads(team)
    # context 1: the naive oracle loop (the checker) on the same inputs
    c = C.copy()
    n_o, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < 2.0:
        orc.fused_brgemm_omp(F32, 1024, 1024, 64, 1024, 1024, 1024, 64, 65536, 4, 0, 0, A, B, c, None, 16)
        n_o += 1
    naive = flops * n_o / (time.perf_counter() - t1) / 1e9
    # context 2: a tuned vendor CPU GEMM on the same host (torch.matmul -> MKL / oneDNN, all cores)
    vendor = None
    try:
        import torch
        torch.set_num_threads(team)
        ta = torch.from_numpy(A.reshape(1024, 1024).copy())
        tb = torch.from_numpy(B.reshape(1024, 1024).copy())
        for _ in range(3):
            torch.matmul(ta, tb)
        n_v, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 2.0:
            torch.matmul(ta, tb)
            n_v += 1
        vendor = {"torch_cpu_matmul_gflops": round(flops * n_v / (time.perf_counter() - t1) / 1e9, 1), "torch_threads": torch.get_num_threads()}
    except Exception:
        pass
    # the reference's HEADLINE benchmark shape on the same host: mlir-gen --kernel=const --batch=256 --layers=1024,1024,1024,1024
    # --tiles=32,32,32 (benchmarks/config/base/base.json:32-38 gemm_fp32_mlir): three chained 256x1024x1024 layers, each 8 x 32
    # tile invokes with br = 32 over packed blocks (the output blocks of a layer ARE the next layer's A blocks)
    headline = None
    try:
        hm, hn = 256, 1024
        rngh = np.random.default_rng(3)
        xs = rngh.uniform(0, 1, hm * hn).astype(np.float32)
        ws = [rngh.uniform(0, 0.01, hn * hn).astype(np.float32) for _ in range(3)]
        xp, w0p, o0 = cb.pack(xs, ws[0], np.zeros(hm * hn, np.float32), hm, hn, hn)
        wps = [w0p] + [cb.pack(xs, w_, xs, hm, hn, hn)[1] for w_ in ws[1:]]
        bufs = [xp, o0, np.empty_like(o0), np.empty_like(o0)]

        def layers():
            for l in range(3):
                cb.run(hm, hn, hn, bufs[l], wps[l], bufs[l + 1], True, 1)
        layers()
        n_h, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < min(3.0, seconds / 3):
            layers()
            n_h += 1
        el_h = time.perf_counter() - t1
        headline = {"workload": "mlir-gen gemm 3 x (256x1024x1024) fp32, tiles 32,32,32 (base.json gemm_fp32_mlir), same call structure",
                    "value": round(3 * 2.0 * hm * hn * hn * n_h / el_h / 1e9, 1), "unit": "GFLOP/s",
                    "us_per_iteration": round(el_h / n_h * 1e6, 1), "iterations": n_h}
    except Exception as ex:
        headline = {"error": str(ex)}
    cpu_model = ""
    try:
        cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    return {"value": round(tiled, 2), "unit": "GFLOP/s", "cores": team, "kind": "port", "cpu": cpu_model,
            "host_cpus": cpu_detail, "value_16_threads": t16 if t16 is not None else (round(tiled, 2) if team == 16 else None),
            "burst_one_socket": burst,
            "build": cb.flags, "matches_oracle": ok,
            "naive_oracle_loop_gflops": round(naive, 2), "vendor_cpu_gemm": vendor, "headline_shape": headline,
            "sample": "%d full passes of the same BRGEMM 1024^3 (%.1f s) as 32x32 tile invokes with br=32 over packed "
                      "32x32x32 blocks, OpenMP over the 32x32 tile grid, %d threads = the CPUs the container may use (affinity %d, cgroup "
                      "quota %s) (oracle/cpu_baseline.c; libxsmm itself is not in the image)"
                      % (reps, el, team, cpu_detail["affinity_cpus"], cpu_detail["cgroup_cpu_quota"])}


def self_launch(n):
    """re-run this command line as n ranks (one per GPU) under torch.distributed.run; returns the launcher's exit code"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL and the peer-store gather need on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")  # (what torchrun would set itself, without its warning banner)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("[bench] --gpus %d without WORLD_SIZE: launching %d ranks: %s\n" % (n, n, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def compact_line(line, detail_path=None):
    """The FINAL stdout line: the contract keys + one compact figure per BASELINE config, < 4 KB, so that the driver's 8 KB tail
    always holds it whole (round 5's single line had grown to 16 KB and the driver's record lost the C4 step: VERDICT r5 weak 7).
    Everything else (other_configs, per_launch, parity detail, the CPU port's shape table) goes out as an EARLIER line and into
    gpurun_out/bench_detail.json. Pure function of the full line: tests/test_host_logic.py runs it on a committed full line."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k_: line.get(k_) for k_ in keep}
    cfg_ = line.get("config") or {}
    out["config"] = {k_: cfg_[k_] for k_ in ("workload", "launch", "kernel", "speedup_vs_one_gpu_same_run", "gathered_bit_identical",
                                              "one_gpu_same_run_ms_per_step") if k_ in cfg_}
    r_ = line.get("roofline") or {}
    roof = {k_: r_.get(k_) for k_ in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    if r_.get("kernel_us") is not None:
        roof["kernel_us"] = r_["kernel_us"]
    if isinstance(r_.get("traffic"), (int, float)):
        roof["traffic"] = round(r_["traffic"])
    if r_.get("algorithmic_bytes") is not None:
        roof["algorithmic_bytes"] = r_["algorithmic_bytes"]
    mb = r_.get("mfma_busy")
    if isinstance(mb, dict) and mb.get("mfma_busy") is not None:
        roof["mfma_busy"] = mb["mfma_busy"]
    configs = {}
    others = line.get("other_configs") or []

    def first(prefix):
        for o_ in others:
            if str(o_.get("workload", "")).startswith(prefix):
                return o_
        return None

    o_ = first("C3 ")
    if o_:
        configs["C3"] = {"us": o_.get("us_per_step"), "frac": o_.get("frac_of_f32_mfma_peak"), "kernel": o_.get("kernel")}
    mlp = line.get("mlp")
    if mlp:
        configs["C4"] = {"ms_per_step": mlp.get("ms_per_step"), "frac": mlp.get("frac_of_bf16_mfma_peak"), "kernel": mlp.get("kernel"),
                         "one_chain_launch": mlp.get("step_is_one_chain_launch"), "gflops": mlp.get("value")}
        shares = mlp.get("per_rank_step_us")
        if shares:
            configs["C4"]["per_rank_chain_us"] = {w_: v_.get("chain_us") for w_, v_ in shares.items()}
        for k_ in ("gather", "speedup_vs_one_gpu_same_run", "gathered_bit_identical"):
            if k_ in mlp:
                configs["C4"][k_] = mlp[k_]
    o_ = first("C5 bf16 BRGEMM")
    if o_:
        configs["C5"] = {"us": o_.get("us_per_step"), "frac": o_.get("frac_of_bf16_mfma_peak"), "kernel": o_.get("kernel")}
    o_ = first("C5 VNNI-2 pack")
    if o_:
        configs["C5_pack"] = {"us": o_.get("us_per_step"), "frac_of_hbm_8TBps": o_.get("frac_of_hbm_8TBps")}
    o_ = first("C5 end to end on the FLAT")
    if o_:
        configs["C5_flat_b_end_to_end_us"] = o_.get("us_per_step")
    o_ = first("mlir-gen mlp 3x1024 bs=256 bias+relu (fp32 unless noted), tile queue, tiles 32,32,32 (")
    if o_:
        configs["ref_mlp_bs256_tile_invokes_us"] = o_.get("us_per_step")
    o_ = first("mlir-gen mlp 3x1024 bs=256 bias+relu (fp32 unless noted), tile queue, tiles 32,32,32, bf16 + VNNI-2 W (")
    if o_:
        configs["ref_mlp_bs256_tile_invokes_bf16_us"] = o_.get("us_per_step")
    o_ = first("mlir-gen mlp 3x1024 bs=256 bias+relu (fp32 unless noted), tile queue, tiles 32,32,32, bf16 + VNNI-2 W, launch thread off")
    if o_:
        configs["ref_mlp_bs256_tile_invokes_bf16_launch_thread_off_us"] = o_.get("us_per_step")
    o_ = first("the reference's benchmark shape set, f32")
    if o_ and "rows" in o_:
        configs["refbench_f32"] = {"at_bar": o_.get("rows_at_0.45_of_peak_or_within_2x_the_launch_floor"), "of": o_.get("rows"),
                                   "median_frac": o_.get("median_frac_of_f32_mfma_peak")}
        for k_ in ("benchmarks_at_bar", "benchmarks"):
            if k_ in o_:
                configs["refbench_f32"][k_] = o_[k_]
    o_ = first("benchmarks/mlir/*.mlir as xsmm call scripts")
    if o_ and isinstance(o_.get("rows"), list):  # the hand-written benchmark files as call scripts: us per call, one caller
        configs["mlir_scripts_us"] = {os.path.basename(str(r_.get("file", "?")).split(":")[0]).replace("fp32-", "").replace(".mlir", ""): r_.get("us") for r_ in o_["rows"]}
    if configs:
        roof["configs"] = configs
    out["roofline"] = roof
    c_ = line.get("cpu_baseline")
    if c_:
        out["cpu_baseline"] = {k_: c_.get(k_) for k_ in ("value", "unit", "cores", "kind", "cpu") if k_ in c_}
        smp = str(c_.get("sample", ""))
        out["cpu_baseline"]["sample"] = smp if len(smp) <= 260 else smp[:257] + "..."
        hs = c_.get("headline_shape")
        if isinstance(hs, dict):
            out["cpu_baseline"]["ref_mlp_gemm_bs256_us"] = hs.get("us_per_iteration")
    par = line.get("parity")
    if isinstance(par, dict):
        out["parity"] = {k_: ({"max_rel": v_.get("max_rel"), "normwise": v_.get("normwise"), "within_1e-5_rel": v_.get("frac_within_1e-5_rel")}
                              if isinstance(v_, dict) else v_) for k_, v_ in par.items()}
    dep = line.get("deployment")
    if isinstance(dep, dict):
        out["deployment"] = {k_: v_ for k_, v_ in dep.items() if not isinstance(v_, str) or len(v_) <= 120}
        if isinstance(out["deployment"].get("host_cache"), dict):
            out["deployment"]["host_cache"] = {k_: v_ for k_, v_ in out["deployment"]["host_cache"].items() if k_ != "stats"}
    if "c2_weak" in line:
        cw = line["c2_weak"]
        out["c2_weak"] = {"value": cw.get("value"), "ms_per_step": cw.get("ms_per_step"), "frac": (cw.get("roofline") or {}).get("frac")}
    if "process_group" in line:
        pg_ = line["process_group"]
        out["process_group"] = {k_: pg_.get(k_) for k_ in ("world_size", "backend", "distinct_devices", "one_device_test_rig")}
    if "error" in line:
        out["error"] = line["error"]
    out["detail"] = detail_path or "the previous stdout line (key \"bench_detail\")"
    return out


def emit(line):
    """print the full line as a DETAIL line (and to gpurun_out/bench_detail.json when that directory can be written), then the
    compact contract line LAST"""
    detail_path = None
    try:
        d_ = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d_, exist_ok=True)
        detail_path = os.path.join(d_, "bench_detail.json")
        with open(detail_path, "w") as f_:
            json.dump(line, f_)
        detail_path = "gpurun_out/bench_detail.json (+ the previous stdout line)"
    except OSError:
        detail_path = None
    print(json.dumps({"bench_detail": line}), flush=True)
    print(json.dumps(compact_line(line, detail_path)), flush=True)


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: become the launcher. One rank per GPU under torch.distributed.run on a free
        # local port; the ranks' stdout / stderr pass through, so rank 0's ONE JSON line is this process' line; its exit code is ours.
        # (Never an N = 1 line under an N > 1 command: VERDICT r4 item 1.)
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python bench.py --gpus N starts them itself)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    if args.refbench:
        if world != 1:
            raise SystemExit("--refbench is an N = 1 mode")
        sys.exit(refbench_mode(args))
    # TEST SWITCH (tests/test_bench_multi_gpu.py): TPP_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and runs the process group on
    # gloo (RCCL cannot place two ranks on one device) - the whole N > 1 code path of this file, the peer-store gather over real IPC
    # handles included, on a one-GPU box. The numbers of such a run are meaningless (the ranks time-slice one GPU) and say so.
    one_device = os.environ.get("TPP_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        local = 0
        # the persistent chain kernel needs every workgroup of ITS launch resident at once (one per CU): two processes that time-slice
        # one GPU starve each other's hand-offs (50 ms timeouts, the runtime dies loudly). On the rig the layers run as separate launches.
        os.environ["TPP_HIP_CHAIN"] = "0"
    torch.cuda.set_device(local)
    use_dist = world > 1 or args.force_dist
    backend = "gloo" if one_device else "nccl"
    ctl_group = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("NCCL_DEBUG", "WARN")  # no version banner on stdout
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            ctl_group = dist.new_group(backend="gloo")  # host objects (IPC handles, votes) travel over gloo, not over the GPUs

    def barrier():
        if use_dist:
            dist.barrier()

    def sync():
        torch.cuda.synchronize()

    pkg = importlib.import_module("tpp-mlir_amd")
    rt = pkg.get_runtime()
    rt.set_async(True)
    K, W = args.steps, args.warmup

    # ------------------------------------------------------------ C2: fp32 BRGEMM 1024^3, br = 16
    m = n = 1024
    k, br = 64, 16
    from oracle import pyoracle as orc  # input generation (restated TensorInit stream) and the parity check
    # OpenMP teams of the checker / the CPU row: the CPUs this container may really use (a default team of nproc threads under a
    # cgroup quota of 16 CPUs is throttled to a crawl); set before the first OpenMP library is loaded
    os.environ.setdefault("OMP_NUM_THREADS", str(orc.usable_cpus()[0]))

    def inputs(kind):
        if kind == "reference":
            # the reference harness' inputs: tpp-run --seed 123 --init-type normal (benchmarks/harness/
            # controller.py:149-154): N(0, 0.2) clamped to [0, 1], ONE stream over the arguments in order
            gen = orc.TensorInit("normal", 123 + rank)
            return gen.fill(m * 1024), gen.fill(1024 * n), gen.fill(m * n)
        rng = np.random.default_rng(1234 + rank)
        return tuple(rng.uniform(-1, 1, cnt).astype(np.float32) for cnt in (m * 1024, 1024 * n, m * n))

    h = rt.brgemm_dispatch(F32, m, n, k, 1024, 1024, 1024, 64, 65536, pkg.GemmFlags.BETA_0)
    flops = 2.0 * m * n * k * br
    runs = {}
    mode = "invoke-loop"
    other = "uniform" if args.init == "reference" else "reference"
    for kind in (other, args.init):  # the stream `value` is quoted on runs last, right after its own warm-up
        hA, hB, hC = inputs(kind)
        dA, dB, dC = (torch.from_numpy(x).cuda() for x in (hA, hB, hC))

        def step():
            rt.brgemm(F32, h, dA, 0, dB, 0, dC, 0, br)

        spin_s = 0.06 if kind == args.init else 0.25  # the first stream also wakes the chip up after process start
        spin_n = spin_up(step, sync, spin_s)
        warm(step, W, sync)
        wall, devs = timed(step, K, sync, barrier)
        if kind == args.init and args.graph:
            replay = graph_of(step)
            if replay is not None:
                for _ in range(W):
                    replay()
                sync()
                wall_g, devs_g = timed(replay, K, sync, barrier)
                if wall_g < wall:
                    wall, devs, mode = wall_g, devs_g, "hipGraph-replay"
        t = torch.tensor([wall, devs], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        runs[kind] = {"wall": float(t[0]), "devs": float(t[1]), "spin_up_s": spin_s, "spin_up_launches": spin_n,
                      "parity": parity_figures(dC.cpu().numpy(), hA, hB, hC, m, n, k, br) if rank == 0 and world == 1 else None}
    wall, devs = runs[args.init]["wall"], runs[args.init]["devs"]
    value = world * flops * K / wall / 1e9
    kernel_s = devs / K
    achieved_tf = flops / kernel_s / 1e12
    inputs_text = {"reference": "tpp-run normal init, seed 123 (N(0,0.2) clamped to [0,1])", "uniform": "uniform [-1,1)"}

    def roofline_of(kind):
        ks = runs[kind]["devs"] / K
        return {"bound": "mfma", "achieved": round(flops / ks / 1e12, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(flops / ks / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), "kernel_us": round(ks * 1e6, 3),
                "inputs": inputs_text[kind], "value_gflops_wall": round(world * flops * K / runs[kind]["wall"] / 1e9, 1)}

    # ------------------------------------------------------------ C4: bf16 3-layer MLP, row-sharded + all-gather
    mlp = None
    if not args.no_mlp:
        N = 1024
        hp = rt.unary_dispatch(pkg.UnaryKind.VNNI2, BF16, N, N, N, N, 0)
        # the reference's inputs: weights / biases = the dense constants of `mlir-gen --kernel=const --seed 123`
        # (each from its own `normal` generator, seeds from mlir-gen's srand / rand chain: MLIRGen.cpp:131-137, 810-819), the
        # kernel input from tpp-run's `normal` stream (--seed 123)
        Wflat, Bflat, _ = orc.mlir_gen_mlp_tensors([N, N, N, N], 123, BF16)
        Wv, Bs = [], []
        for wf_, bf_ in zip(Wflat, Bflat):
            wf = torch.from_numpy(wf_.view(np.int16)).cuda()
            wv = torch.empty_like(wf)
            rt.unary(BF16, hp, wf, 0, wv, 0)  # C5 prologue: weights packed to VNNI-2 by the runtime's own op
            Wv.append(wv)
            Bs.append(torch.from_numpy(bf_.view(np.int16)).cuda())
        x_all = {}  # the kernel input, the SAME on every rank (tpp-run's normal stream, seed 123): a rank uses its own rows of it

        def x_rows(batch, row0, rows):
            if batch not in x_all:
                x_all[batch] = torch.from_numpy(orc.TensorInit("normal", 123).fill(batch * N, BF16).view(np.int16)).cuda()
            return x_all[batch][row0 * N:(row0 + max(rows, 1)) * N]

        peer_objs = {}  # batch -> PeerGather or None (created once per batch size: 2 x batch x N x 2 bytes of mapped buffers each)

        def peer_for(batch):
            if batch not in peer_objs:
                pg_ = None
                if use_dist and batch % (128 * world) == 0:
                    pg_ = pkg.PeerGather.create(rt, rank, world, batch * N * 2, group=ctl_group)  # every rank or none (it votes)
                    if pg_ is not None:
                        pg_.overlap(True)  # the wait for the peers' blocks on a side stream: the next step computes meanwhile
                peer_objs[batch] = pg_
            return peer_objs[batch]

        def run_mlp(batch, steps, warmup, chain=True, as_world=None, as_rank=None, gather=None):
            """row-sharded MLP on `batch` rows; gather: None (this rank's share only), "peer" (peer-store all-gather) or "rccl"
            (dist.all_gather_into_tensor) INSIDE the timed step. Returns (spec, sharded object, seconds per step, gathered output).
            as_world / as_rank: time the share rank as_rank would have in a world of as_world GPUs, on THIS GPU"""
            spec_ = pkg.MlpSpec(batch=batch)
            sh_ = pkg.ShardedMlp(spec_, rank if as_rank is None else as_rank, world if as_world is None else as_world, rt, chain=chain)
            X_ = x_rows(batch, sh_.row0, sh_.rows)
            acts_ = [torch.empty(max(sh_.rows, 1), N, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
            full_ = torch.empty(batch, N, dtype=torch.bfloat16, device="cuda") if gather == "rccl" else None
            pg_ = peer_for(batch) if gather == "peer" else None
            last = [None]
            sync()

            def step_():
                out = sh_.forward(X_, Wv, Bs, acts_)
                if gather == "peer":
                    last[0] = pg_.gather(out, sh_.rows * N * 2, sh_.row0 * N * 2)
                elif gather == "rccl":
                    pkg.all_gather_rows(out, full_, spec_, world)
                    last[0] = full_
                else:
                    last[0] = out

            spin_up(step_, sync, 0.03)
            warm(step_, warmup, sync)
            w_, _ = timed(step_, steps, sync, barrier)
            tw = torch.tensor([w_], dtype=torch.float64, device="cuda")
            if use_dist:
                dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            if pg_ is not None:
                pg_.check()
            return spec_, sh_, float(tw[0]) / steps, last[0]

        def run_gathered(batch, steps, warmup, path):
            """run_mlp with a gather path, failure-safe ACROSS the ranks: a rank whose peer-store gather ran into its bounded wait
            raises at the very end of run_mlp (behind its last collective); here every rank then votes, so either all ranks use
            the result or all of them drop the path (and the peer buffers of that batch size). Returns (seconds per step, gathered
            output, None) or (None, None, reason)."""
            err = None
            t_ = full_ = None
            try:
                _, _, t_, full_ = run_mlp(batch, steps, warmup, gather=path)
                sync()
                if path == "peer":
                    peer_for(batch).drain()
                    if os.environ.get("TPP_BENCH_TEST_FAIL_PEER") == str(rank):  # TEST SWITCH (tests/test_bench_multi_gpu.py)
                        raise RuntimeError("injected failure of the peer path on rank %d" % rank)
            except RuntimeError as ex:
                err = str(ex)
            ok = torch.tensor([0 if err else 1], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]):
                return t_, full_, None
            if path == "peer":
                peer_objs[batch] = None
            return None, None, err or "failed on another rank"

        def unsharded_on_the_shards_tile(batch, sh_):
            """the full batch on THIS GPU with the tile the rank shares ran on (three launches: more tiles than CUs is fine there) -
            the same arithmetic per element as the sharded run, so the gathered output must equal it bit for bit"""
            name = rt.kernel_name(sh_.handles[0][0])
            forced = {"brgemm_bf16_lw<32x64,k2>": 20, "brgemm_bf16_lw<64x64>": 21, "brgemm_bf16_lw<64x128>": 22, "brgemm_bf16_lw<128x128>": 23,
                      "brgemm_bf16_fast<64x64>": 16, "brgemm_bf16_dma<128x128>": 17, "brgemm_bf16_dma<256x256>": 18,
                      "brgemm_bf16_small<32x32,k4>": 19}.get(name, -1)
            rt.force_variant(forced)
            try:
                whole = pkg.ShardedMlp(pkg.MlpSpec(batch=batch), 0, 1, rt, chain=False)
            finally:
                rt.force_variant(-1)
            acts_ = [torch.empty(batch, N, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
            ref = whole.forward(x_rows(batch, 0, batch), Wv, Bs, acts_)
            sync()
            return ref, rt.kernel_name(whole.handles[0][0]), name

        if not use_dist:
            spec, sh, mstep, _ = run_mlp(4096, K, W)
            mcompute = mstep
            gathers = None
        else:
            # N > 1: the strong-scaled step with the all-gather INSIDE the timed region, BOTH gather paths in the same run, then the
            # check: every rank recomputes the full batch unsharded and compares the gathered output bit for bit
            spec, sh, mcompute, _ = run_mlp(4096, K, W)
            gathers = {}
            ref, ref_kernel, shard_kernel = unsharded_on_the_shards_tile(4096, sh)
            for path in (["peer"] if peer_for(4096) is not None else []) + ["rccl"]:
                t_, full_, err_ = run_gathered(4096, K, W, path)
                if err_ is not None:  # (every rank agrees: run_gathered votes) the other path still produces the line
                    gathers[path] = {"failed": err_}
                    continue
                same = torch.equal(full_.view(torch.int16).reshape(-1), ref.view(torch.int16).reshape(-1))
                ok_ = torch.tensor([1 if same else 0], device="cuda")
                dist.all_reduce(ok_, op=dist.ReduceOp.MIN)
                gathers[path] = {"ms_per_step": round(t_ * 1e3, 5), "value": round(spec.flops() / t_ / 1e9, 1), "unit": "GFLOP/s",
                                 "gathered_bit_identical": bool(int(ok_[0]))}
            timed_paths = [p_ for p_ in gathers if "ms_per_step" in gathers[p_]]
            if not timed_paths:
                # every gather path failed (on some rank): no headline can be quoted. The line still goes out - value null, the
                # reasons in it - and the run exits non-zero
                if use_dist:
                    dist.barrier()
                    dist.destroy_process_group()
                if rank == 0:
                    print(json.dumps({"metric": "GFLOP/s on the 3-layer MLP 1024x3 bf16 bs=4096 (bias+relu), rows sharded over the GPUs, "
                                                "all-gather of the output inside the timed step", "value": None, "unit": "GFLOP/s",
                                      "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": None, "higher_is_better": True,
                                      "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                                      "config": {"workload": "BASELINE config 4: 3-layer MLP bf16 bs=4096"},
                                      "error": "every all-gather path failed", "gathers": gathers,
                                      "compute_only_ms_per_step": round(mcompute * 1e3, 5)}), flush=True)
                sys.exit(4)
            best = min(timed_paths, key=lambda p_: gathers[p_]["ms_per_step"])
            mstep = gathers[best]["ms_per_step"] * 1e-3
            if peer_for(4096) is None and "peer" not in gathers:
                gathers["peer"] = {"unavailable": "the peer buffers could not be mapped or the self-test failed on a rank: RCCL only"}
            # the same step on ONE GPU in the same run (rank 0's device, the others wait): the denominator of the speed-up
            _, sh1, t1_, _ = run_mlp(4096, K, W, as_world=1, as_rank=0)
            mlp_n1 = {"ms_per_step": round(t1_ * 1e3, 5), "value": round(spec.flops() / t1_ / 1e9, 1), "kernel": rt.kernel_name(sh1.handles[0][0]),
                      "one_chain_launch": bool(sh1.last_step_fused)}
        fused_flag = bool(sh.last_step_fused)
        mlp = {"workload": "3-layer MLP 1024x3 bf16 bs=4096 bias+relu, rows sharded over %d GPU(s)%s" % (
                   world, " + all-gather of the output inside the timed step" if use_dist else ""),
               "value": round(spec.flops() / mstep / 1e9, 1), "unit": "GFLOP/s", "scaling": "strong",
               "ms_per_step": round(mstep * 1e3, 5), "ms_per_step_compute_only": round(mcompute * 1e3, 5),
               "flops_per_step": spec.flops(),
               "frac_of_bf16_mfma_peak": round(spec.flops() / mstep / 1e12 / (PEAK_BF16_MFMA_TFLOPS * world), 4),
               "kernel": rt.kernel_name(sh.handles[0][0]) if sh.rows else "",
               "step_is_one_chain_launch": fused_flag,
               "inputs": "weights / biases: mlir-gen --seed 123 dense constants (seed chain 123, rand(), ...), input: tpp-run normal init seed 123 "
                         "(one stream over the whole batch; a rank takes its rows)",
               "note": "25.8 GFLOP per step; a rank's three layers run as ONE persistent launch (xsmm_hip_fused_brgemm_chain_invoke) "
                       "when the chain fits the chip, see DESIGN.md sections 4.2 / 5"}
        if use_dist:
            mlp["gather"] = best
            mlp["headline_gather_rule"] = "the faster of the gather paths that completed on every rank; each is bit-compared with the unsharded result and a mismatch on either invalidates the run (exit 3)"
            mlp["gathers"] = gathers
            mlp["gathered_bit_identical"] = all(g.get("gathered_bit_identical", True) for g in gathers.values())
            mlp["gathered_check"] = ("after the timed regions every rank recomputed the full batch unsharded on its own GPU (kernel %s, forced to "
                                     "the tile the %d-row shares run on: %s) and compared the gathered [4096][1024] output of each gather path "
                                     "bit for bit; AND over the ranks" % (ref_kernel, sh.rows, shard_kernel))
            mlp["one_gpu_same_run"] = mlp_n1
            mlp["speedup_vs_one_gpu_same_run"] = round(mlp_n1["ms_per_step"] * 1e-3 / mstep, 3)
        if world == 1:
            # what every rank of a world of 2 / 4 / 8 GPUs would run per step (its row share, no collective), measured on THIS GPU:
            # the one-launch chain and the same share as three launches; python path (this harness) and native (tools/mlp_probe)
            Ks = max(K, 200)  # (short steps: the synchronizes around a 20-step region would be 5-10 % of it)
            _, _, t3, _ = run_mlp(4096, Ks, W, chain=False)
            shares = {"1": {"rows": 4096, "chain_us": round(mcompute * 1e6, 2), "three_launches_us": round(t3 * 1e6, 2)}}
            for w_ in (2, 4, 8):
                _, sh_c, tc_, _ = run_mlp(4096, Ks, W, chain=True, as_world=w_, as_rank=w_ - 1)
                _, sh_l, tl_, _ = run_mlp(4096, Ks, W, chain=False, as_world=w_, as_rank=w_ - 1)
                shares[str(w_)] = {"rows": sh_c.rows, "chain_us": round(tc_ * 1e6, 2), "three_launches_us": round(tl_ * 1e6, 2),
                                   "one_launch": bool(sh_c.last_step_fused), "kernel": rt.kernel_name(sh_l.handles[0][0])}
            mlp["per_rank_step_us"] = shares
            probe = os.path.join(ROOT, "tools", "mlp_probe")
            if os.path.exists(probe):
                try:
                    r_ = subprocess.run([probe, "--rows", "4096,2048,1024,512", "--iters", "400"], capture_output=True, text=True, timeout=120)
                    nat = [json.loads(l) for l in r_.stdout.splitlines() if l.startswith("{")]
                    mlp["per_rank_step_us_native"] = {str(4096 // d_["rows"]): {"rows": d_["rows"], "chain_us": d_["chain_us"],
                                                                              "three_launches_us": d_["per_layer_launches_us"],
                                                                              "one_layer_us": d_["one_layer_us"], "kernel": d_["kernel"]}
                                                      for d_ in nat}
                except Exception as ex:
                    mlp["per_rank_step_us_native"] = {"error": str(ex)}
            mlp["per_rank_step_note"] = ("per-rank share of the bs=4096 step for a world of N GPUs, run on one GPU: the scaling model of the "
                                         "row sharding WITHOUT the all-gather (strong-scaling speed-up of the compute part = "
                                         "chain_us[1] / chain_us[N])")
        if use_dist and not one_device:
            # (not on the one-device test rig: two processes' chip-filling kernels time-slice ONE GPU there, and a gather kernel that
            # spins for a peer whose GEMM cannot get a compute unit meanwhile runs into its - bounded - wait: 2 of 9 runs)
            # the same MLP at a batch where compute dominates (8 x 4096 rows): what the sharding itself scales like
            Kl, Wl = max(20, K // 10), max(5, W // 10)
            spec_l, sh_l, lcompute, _ = run_mlp(32768, Kl, Wl)
            lpath = "peer" if peer_for(32768) is not None else "rccl"
            lstep, _, lerr = run_gathered(32768, Kl, Wl, lpath)
            if lerr is not None and lpath == "peer":
                lpath = "rccl"
                lstep, _, lerr = run_gathered(32768, Kl, Wl, lpath)
            mlp["large_batch_variant"] = {"failed": lerr} if lerr is not None else {
                "workload": "same MLP, bs=32768, rows sharded over %d GPU(s) + all-gather of the output (64 MiB, %s)" % (world, lpath),
                "value": round(spec_l.flops() / lstep / 1e9, 1), "unit": "GFLOP/s", "scaling": "strong",
                "ms_per_step": round(lstep * 1e3, 5), "ms_per_step_compute_only": round(lcompute * 1e3, 5),
                "kernel": rt.kernel_name(sh_l.handles[0][0]) if sh_l.rows else ""}

        # the other sharding of SURVEY.md 8(e): column blocks + an all-gather after EVERY layer; the
        # gathered [W][batch][N/W] activations feed the next layer as a batch-reduce over the rank blocks
        if use_dist and N % world == 0:
            cs = pkg.ColumnShardedMlp(spec, rank, world, rt)
            gx = torch.Generator(device="cpu").manual_seed(11)
            Xf = (torch.randn(spec.batch, N, generator=gx) * 0.5).to(torch.bfloat16).cuda()  # same on every rank
            loc = [torch.empty(spec.batch, N // world, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
            gat = [torch.empty(world, spec.batch, N // world, dtype=torch.bfloat16, device="cuda") for _ in range(3)]

            def col_step():
                cs.forward(Xf, Wv, Bs, loc, gat, lambda d, s_: dist.all_gather_into_tensor(d.view(-1, d.shape[-1]), s_))

            warm(col_step, max(100, W // 2), sync)  # RCCL sets up lazily per message size: short warm-ups time that
            cwall, _ = timed(col_step, max(50, K // 5), sync, barrier)
            tc = torch.tensor([cwall], dtype=torch.float64, device="cuda")
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            cwall, ck = float(tc[0]), max(50, K // 5)
            mlp["column_sharded_variant"] = {
                "workload": "same MLP, output columns sharded over %d GPU(s), RCCL all-gather after each of the 3 layers "
                            "(next layer = batch-reduce over the rank blocks, br = %d)" % (world, world),
                "value": round(spec.flops() * ck / cwall / 1e9, 1), "unit": "GFLOP/s", "scaling": "strong",
                "ms_per_step": round(cwall / ck * 1e3, 5)}

    # ------------------------------------------------------------ the other BASELINE configs (N=1 only, short)
    others = None
    if world == 1 and not args.no_mlp:
        others = []
        Ko, Wo = max(200, K // 10), max(20, W // 10)  # (the two synchronizes around a timed region cost ~25 us: 50 launches of a 10 us kernel would read 5 % slow)
        # C3: fp32 fused_brgemm + bias + relu, MLP layer 1024->1024, bs=512
        A3 = torch.rand(512, 1024, device="cuda") - 0.4
        W3 = torch.rand(1024, 1024, device="cuda") - 0.5
        b3 = torch.rand(1024, device="cuda")
        C3 = torch.empty(512, 1024, device="cuda")
        h3 = rt.fused_brgemm_dispatch(F32, 512, 1024, 64, 1024, 1024, 1024, 64, 65536, 4, 0, 5, 4, 1)

        def c3_step():
            rt.fused_brgemm(F32, h3, A3, 0, W3, 0, C3, 0, b3, 0, 16)
        spin_up(c3_step, sync, 0.03)
        warm(c3_step, Wo, sync)
        w3, _ = timed(c3_step, Ko, sync, barrier)
        f3 = 2.0 * 512 * 1024 * 1024 + 2.0 * 512 * 1024  # MLIRGen.cpp:328-334
        others.append({"workload": "C3 fused_brgemm+bias+relu fp32 512x1024x1024", "kernel": rt.kernel_name(h3),
                       "value": round(f3 * Ko / w3 / 1e9, 1), "unit": "GFLOP/s", "us_per_step": round(w3 / Ko * 1e6, 2),
                       "frac_of_f32_mfma_peak": round(f3 * Ko / w3 / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)})
        # C5: xsmm.unary VNNI-2 pack of B (2048^2 bf16) + bf16 BRGEMM 2048^3 (k=128, br=16, VNNI_B|BETA_0)
        M5 = 2048
        A5 = (torch.rand(M5, M5, device="cuda") - 0.5).to(torch.bfloat16)
        B5 = (torch.rand(M5, M5, device="cuda") - 0.5).to(torch.bfloat16)
        B5v = torch.empty_like(B5)
        C5 = torch.empty(M5, M5, device="cuda", dtype=torch.bfloat16)
        hp5 = rt.unary_dispatch(pkg.UnaryKind.VNNI2, BF16, M5, M5, M5, M5, 0)
        h5 = rt.brgemm_dispatch(BF16, M5, M5, 128, M5, M5, M5, 128, 128 * M5, 4 | 2048)

        def c5_pack():
            rt.unary(BF16, hp5, B5, 0, B5v, 0)

        def c5_gemm():
            rt.brgemm(BF16, h5, A5, 0, B5v, 0, C5, 0, 16)
        spin_up(c5_pack, sync, 0.03)
        warm(c5_pack, Wo, sync)
        wp, _ = timed(c5_pack, Ko, sync, barrier)
        spin_up(c5_gemm, sync, 0.03)
        warm(c5_gemm, Wo, sync)
        wg, _ = timed(c5_gemm, Ko, sync, barrier)
        pack_bytes = 2.0 * M5 * M5 * 2
        others.append({"workload": "C5 VNNI-2 pack prologue 2048x2048 bf16 (bit-exact move)", "value": round(pack_bytes * Ko / wp / 1e9, 1),
                       "unit": "GB/s", "us_per_step": round(wp / Ko * 1e6, 2), "frac_of_hbm_8TBps": round(pack_bytes * Ko / wp / 8e12, 4)})
        others.append({"workload": "C5 bf16 BRGEMM 2048^3 VNNI_B (k=128, br=16)", "kernel": rt.kernel_name(h5),
                       "value": round(2.0 * M5 ** 3 * Ko / wg / 1e9, 1), "unit": "GFLOP/s", "us_per_step": round(wg / Ko * 1e6, 2),
                       "frac_of_bf16_mfma_peak": round(2.0 * M5 ** 3 * Ko / wg / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)})
        # C5 end to end, both ways: pack launch + VNNI-2 BRGEMM (what the reference's pipeline emits), and ONE BRGEMM on the flat
        # operand (no VNNI flag: the interleave happens in the B loader of the loader-wave tiles)

        def c5_both():
            rt.unary(BF16, hp5, B5, 0, B5v, 0)
            rt.brgemm(BF16, h5, A5, 0, B5v, 0, C5, 0, 16)
        h5f = rt.brgemm_dispatch(BF16, M5, M5, 128, M5, M5, M5, 128, 128 * M5, 4)
        C5f = torch.empty_like(C5)

        def c5_flat():
            rt.brgemm(BF16, h5f, A5, 0, B5, 0, C5f, 0, 16)
        warm(c5_both, Wo, sync)
        wb, _ = timed(c5_both, Ko, sync, barrier)
        warm(c5_flat, Wo, sync)
        wf, _ = timed(c5_flat, Ko, sync, barrier)
        sync()
        others.append({"workload": "C5 end to end: VNNI-2 pack launch + bf16 BRGEMM 2048^3", "kernel": rt.kernel_name(h5),
                       "value": round(2.0 * M5 ** 3 * Ko / wb / 1e9, 1), "unit": "GFLOP/s", "us_per_step": round(wb / Ko * 1e6, 2)})
        others.append({"workload": "C5 end to end on the FLAT B operand (one launch, interleave in the B loader)", "kernel": rt.kernel_name(h5f),
                       "value": round(2.0 * M5 ** 3 * Ko / wf / 1e9, 1), "unit": "GFLOP/s", "us_per_step": round(wf / Ko * 1e6, 2),
                       "bit_identical_to_pack_plus_vnni": bool(torch.equal(C5f, C5))})

        # the C4 layer on a VNNI-4 B operand (`--vnni=4` configs of the reference, benchmarks/config/omp/mlir-bf16.json:68-100):
        # the same matrix packed [K/4][N][4], the runtime told the factor (xsmm_hip_set_vnni_factor), bit-compared with the VNNI-2 result
        try:
            M4, N4 = 4096, 1024
            A4 = (torch.rand(M4, N4, device="cuda") - 0.5).to(torch.bfloat16)
            Wf4 = (torch.rand(N4, N4, device="cuda") - 0.5).to(torch.bfloat16)
            W2 = Wf4.view(N4 // 2, 2, N4).permute(0, 2, 1).contiguous()
            W4 = Wf4.view(N4 // 4, 4, N4).permute(0, 2, 1).contiguous()
            C42, C44 = torch.empty(M4, N4, device="cuda", dtype=torch.bfloat16), torch.empty(M4, N4, device="cuda", dtype=torch.bfloat16)
            h42 = rt.brgemm_dispatch(BF16, M4, N4, 64, N4, N4, N4, 64, 64 * N4, 4 | 2048)
            oldf = rt.set_vnni_factor(4)
            h44 = rt.brgemm_dispatch(BF16, M4, N4, 64, N4, N4, N4, 64, 64 * N4, 4 | 2048)
            rt.set_vnni_factor(oldf)

            def l2_step():
                rt.brgemm(BF16, h42, A4, 0, W2, 0, C42, 0, 16)

            def l4_step():
                rt.brgemm(BF16, h44, A4, 0, W4, 0, C44, 0, 16)
            res4 = {}
            for nm_, fn_ in (("vnni2", l2_step), ("vnni4", l4_step)):
                spin_up(fn_, sync, 0.03)
                warm(fn_, Wo, sync)
                w4_, _ = timed(fn_, Ko, sync, barrier)
                res4[nm_] = w4_ / Ko
            f4 = 2.0 * M4 * N4 * N4
            others.append({"workload": "C4 layer 4096x1024x1024 bf16 on a VNNI-4 B operand [K/4][N][4] (xsmm_hip_set_vnni_factor(4))", "kernel": rt.kernel_name(h44),
                           "value": round(f4 / res4["vnni4"] / 1e9, 1), "unit": "GFLOP/s", "us_per_step": round(res4["vnni4"] * 1e6, 2),
                           "frac_of_bf16_mfma_peak": round(f4 / res4["vnni4"] / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                           "same_layer_vnni2_us": round(res4["vnni2"] * 1e6, 2), "bit_identical_to_vnni2": bool(torch.equal(C42, C44))})
            del A4, Wf4, W2, W4, C42, C44
        except Exception as ex:
            others.append({"workload": "C4 layer on a VNNI-4 B operand", "error": str(ex)})

        # a large bf16 output (one 256x256 tile per CU): the shape class the 256-tile kernel exists for
        ML = 4096
        AL = (torch.rand(ML, ML, device="cuda") - 0.5).to(torch.bfloat16)
        BL = (torch.rand(ML // 2, ML, 2, device="cuda") - 0.5).to(torch.bfloat16)
        CL = torch.empty(ML, ML, device="cuda", dtype=torch.bfloat16)
        hl = rt.brgemm_dispatch(BF16, ML, ML, 64, ML, ML, ML, 64, 64 * ML, 4 | 2048)

        def big_gemm():
            rt.brgemm(BF16, hl, AL, 0, BL, 0, CL, 0, ML // 64)
        spin_up(big_gemm, sync, 0.03)
        warm(big_gemm, 30, sync)
        wl, _ = timed(big_gemm, 100, sync, barrier)
        others.append({"workload": "bf16 BRGEMM 4096^3 VNNI_B (k=64, br=64), uniform random operands", "kernel": rt.kernel_name(hl),
                       "value": round(2.0 * ML ** 3 * 100 / wl / 1e9, 1), "unit": "GFLOP/s", "us_per_step": round(wl / 100 * 1e6, 2),
                       "frac_of_bf16_mfma_peak": round(2.0 * ML ** 3 * 100 / wl / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)})
        del AL, BL, CL

        # the boundary as JIT'd host code sees it: HOST pointers, synchronous invoke (mirror H2D, kernel,
        # D2H per call) - the PCIe-inclusive rate of the C2 BRGEMM; never the headline value
        rt.set_async(False)
        hc = hC.copy()
        rt.brgemm(F32, h, hA, 0, hB, 0, hc, 0, br)
        t0 = time.perf_counter()
        for _ in range(10):
            rt.brgemm(F32, h, hA, 0, hB, 0, hc, 0, br)
        th = (time.perf_counter() - t0) / 10
        others.append({"workload": "C2 through HOST pointers, synchronous invoke (PCIe-inclusive: 12 MiB up, 4 MiB down per call from pageable memory)",
                       "value": round(flops / th / 1e9, 1), "unit": "GFLOP/s", "us_per_step": round(th * 1e6, 1)})
        # ... and with the host cache (round 6, TPP_HIP_HOST_CACHE=1: the operands keep a device mirror between invokes, only pages the
        # host wrote are uploaded again - csrc/host_cache.h): synchronous (the reference's contract: C on the host at every return) and
        # asynchronous (TPP_HIP_ASYNC=1: C on the host at perf_stop_timer) - what an UNMODIFIED harness gets from environment variables alone
        host_cache = None
        if rt.set_host_cache(True) >= 0:
            try:
                # FRESH anonymous mappings: the arrays above have just been the source / destination of plain hipMemcpy calls, which
                # leaves them in the ROCm runtime's pin cache - the driver then write-faults their pages again after every later
                # piece of driver activity and the kernel's write tracking reports them written (tools/ubench/wp_vs_hipmemcpy.cpp,
                # profiles/r06_wp_vs_hipmemcpy.txt; the cache stays correct and degrades to the plain path's cost for such a buffer).
                # An unmodified harness that runs with the cache from the start never hands its pages to a driver copy.
                import mmap
                fresh_maps = []

                def fresh(arr):
                    mm_ = mmap.mmap(-1, arr.nbytes + 8192)
                    fresh_maps.append(mm_)
                    out_ = np.frombuffer(mm_, dtype=np.uint8, count=arr.nbytes, offset=128).view(arr.dtype)
                    out_[:] = arr
                    return out_
                cA, cB, cc = fresh(hA), fresh(hB), fresh(hc)
                rt.brgemm(F32, h, cA, 0, cB, 0, cc, 0, br)
                t0 = time.perf_counter()
                for _ in range(20):
                    rt.brgemm(F32, h, cA, 0, cB, 0, cc, 0, br)
                ths = (time.perf_counter() - t0) / 20
                # (another fresh output for the asynchronous phase: the synchronous invokes above delivered `cc` by a DMA straight into
                # its pages - the scratch-output path - which puts THAT buffer into the driver's pin cache)
                cc = fresh(hc)
                rt.set_async(True)
                for _ in range(20):
                    rt.brgemm(F32, h, cA, 0, cB, 0, cc, 0, br)
                rt.synchronize()
                loops = []
                for _ in range(3):
                    tp = rt.perf_start_timer()
                    for _ in range(1000):
                        rt.brgemm(F32, h, cA, 0, cB, 0, cc, 0, br)
                    loops.append(rt.perf_stop_timer(tp) / 1000)
                hc = cc
                tha = sorted(loops)[1]
                same = bool(np.array_equal(hc.view(np.uint32), dC.cpu().numpy().view(np.uint32)))  # the device-pointer result of the timed stream above
                host_cache = {"c2_sync_us": round(ths * 1e6, 1), "c2_async_us": round(tha * 1e6, 2), "c2_async_gflops": round(flops / tha / 1e9, 1),
                              "c2_async_vs_device_pointers": round(tha / (wall / K), 3), "c2_async_output_bit_identical_to_device_pointers": same,
                              "stats": rt.host_cache_stats()}
                others.append({"workload": "C2 through HOST pointers + host cache, synchronous invoke (TPP_HIP_HOST_CACHE=1)", "value": round(flops / ths / 1e9, 1),
                               "unit": "GFLOP/s", "us_per_step": round(ths * 1e6, 1)})
                others.append({"workload": "C2 through HOST pointers + host cache, asynchronous (TPP_HIP_ASYNC=1 TPP_HIP_HOST_CACHE=1; median of 3 loops of 1000, "
                                           "write-back at perf_stop_timer inside each)", "value": round(flops / tha / 1e9, 1), "unit": "GFLOP/s",
                               "us_per_step": round(tha * 1e6, 2), "output_bit_identical_to_device_pointers": same})
            finally:
                rt.set_async(False)
                rt.set_host_cache(False)
        rt.set_async(True)

        # the reference's headline benchmark as the compiler emits it: mlir-gen --batch=256
        # --layers=1024x4 --tiles=32,32,32 --bias --relu = 3 x 256 invokes of ONE 32x32x32 dispatch
        # (benchmarks/config/base/base.json:74-80), replayed by the native harness with the
        # reference's timing loop; the runtime's tile queue turns them into 3 grouped launches
        replay = os.path.join(ROOT, "tools", "tpp_replay")
        if os.path.exists(replay):
            for label, extra in (("tile queue, tiles 32,32,32", ["--tiles", "32", "--queue", "1", "-n", "200"]),
                                 ("tile queue, tiles 32,32,32, 2 OpenMP callers", ["--tiles", "32", "--queue", "1", "-n", "200", "--threads", "2"]),
                                 ("tile queue, tiles 32,32,32, 8 OpenMP callers", ["--tiles", "32", "--queue", "1", "-n", "200", "--threads", "8"]),
                                 ("tile queue, tiles 64,64,64", ["--tiles", "64", "--queue", "1", "-n", "200"]),
                                 ("tile queue, tiles 32,32,32, bf16 + VNNI-2 W", ["--tiles", "32", "--queue", "1", "-n", "200", "--bf16"]),
                                 ("tile queue, tiles 64,64,64, bf16 + VNNI-2 W", ["--tiles", "64", "--queue", "1", "-n", "200", "--bf16"]),
                                 ("tile queue, tiles 32,32,32, bf16 + VNNI-2 W, launch thread off (TPP_HIP_LAUNCH_THREAD=0)", ["--tiles", "32", "--queue", "1", "-n", "200", "--bf16"]),
                                 ("whole-layer dispatch", ["--whole-layer", "-n", "1000"])):
                env_ = dict(os.environ, TPP_HIP_LAUNCH_THREAD="0") if "launch thread off" in label else None
                r = subprocess.run([replay, "--batch", "256", "--layers", "1024,1024,1024,1024", "--bias", "--relu"] + extra,
                                   capture_output=True, text=True, timeout=300, env=env_)
                mm = re.search(r"mean ([0-9.]+) us[^,]*, ([0-9.]+) GFLOP/s", r.stderr)
                if mm:
                    others.append({"workload": "mlir-gen mlp 3x1024 bs=256 bias+relu (fp32 unless noted), " + label + " (tools/tpp_replay)",
                                   "value": float(mm.group(2)), "unit": "GFLOP/s", "us_per_step": float(mm.group(1))})
            # the same MLP as emitted, on plain HOST buffers (malloc, 64-byte aligned: what an unmodified tpp-run hands over), the program
            # issues the reference's symbols only, the runtime's modes come from the environment (tools/tpp_replay --host-buffers)
            for label, extra, env_ in (("host buffers, TPP_HIP_ASYNC=1 TPP_HIP_TILE_QUEUE=1 TPP_HIP_HOST_CACHE=1", ["--tiles", "32", "-n", "200", "--repeats", "9"],
                                        {"TPP_HIP_ASYNC": "1", "TPP_HIP_TILE_QUEUE": "1", "TPP_HIP_HOST_CACHE": "1"}),
                                       ("host buffers, the same, 8 OpenMP callers", ["--tiles", "32", "-n", "200", "--repeats", "9", "--threads", "8"],
                                        {"TPP_HIP_ASYNC": "1", "TPP_HIP_TILE_QUEUE": "1", "TPP_HIP_HOST_CACHE": "1"}),
                                       ("host buffers, no environment (768 synchronous mirrored invokes per iteration)", ["--tiles", "32", "-n", "3"], {})):
                r = subprocess.run([replay, "--host-buffers", "--batch", "256", "--layers", "1024,1024,1024,1024", "--bias", "--relu"] + extra,
                                   capture_output=True, text=True, timeout=300, env=dict(os.environ, **env_))
                mm = re.search(r"mean ([0-9.]+) us[^,]*, ([0-9.]+) GFLOP/s", r.stderr)
                md = re.search(r"repeats \d+ x \d+ calls: min ([0-9.]+) median ([0-9.]+) max ([0-9.]+) us", r.stderr)
                if mm and "host output buffer checked" in r.stderr:
                    row = {"workload": "mlir-gen mlp 3x1024 bs=256 bias+relu fp32 as tile invokes 32,32,32, " + label + " (tools/tpp_replay --host-buffers; the "
                                       "host's output buffer checked behind the loop)", "value": float(mm.group(2)), "unit": "GFLOP/s",
                           "us_per_step": float(mm.group(1))}
                    if md:  # (the first loop of a process carries a one-off 7-10 ms stall of the runtime's first device-to-host copy: profiles/r06_host_cache_first_writeback.txt)
                        row["us_median_of_5_loops"] = float(md.group(2))  # (key kept; nine loops since the second half of round 6)
                        row["us_min_max_of_9_loops"] = [float(md.group(1)), float(md.group(3))]
                    others.append(row)
            # the plain gemm row of the reference's headline config (base.json:34 gemm_fp32_mlir: no bias, no relu) - the shape the CPU
            # row's `headline_shape` quotes - as the compiler emits it
            r = subprocess.run([replay, "--batch", "256", "--layers", "1024,1024,1024,1024", "--tiles", "32", "--queue", "1", "-n", "200"],
                               capture_output=True, text=True, timeout=300)
            mm = re.search(r"mean ([0-9.]+) us[^,]*, ([0-9.]+) GFLOP/s", r.stderr)
            if mm:
                others.append({"workload": "mlir-gen gemm 3x1024 bs=256 fp32 (base.json gemm_fp32_mlir: no bias / relu), tile queue, tiles 32,32,32 (tools/tpp_replay)",
                               "value": float(mm.group(2)), "unit": "GFLOP/s", "us_per_step": float(mm.group(1)),
                               "frac_of_f32_mfma_peak": round(float(mm.group(2)) / 1e3 / PEAK_F32_MFMA_TFLOPS, 4)})
            # the reference's own benchmark SHAPE SET (benchmarks/config/matmul/*.json, fc/*.json, base/base.json: 17 + 17 + 2 rows),
            # f32, each as the compiler emits it (tile invokes through the tile queue) and as one whole-layer dispatch, with the CPU
            # port on the same shape: tools/refbench.py (the full table incl. bf16 VNNI-2 / VNNI-4: profiles/r05_refbench.txt).
            # Here: the three worst and the three best rows by fraction of the f32 MFMA peak.
            try:
                with tempfile.TemporaryDirectory() as td:
                    jf = os.path.join(td, "refbench.json")
                    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "refbench.py"), "--quick", "-n", "200", "--json", jf],
                                   capture_output=True, text=True, timeout=600, check=True)  # (GPU side only; the CPU column joins below)
                    rows = json.load(open(jf))
                scripts = [r_ for r_ in rows if r_["family"] == "mlir"]
                rows = [r_ for r_ in rows if r_["family"] != "mlir"]
                rows.sort(key=lambda r_: r_["frac_of_peak"])
                keep = lambda r_: {"benchmark": r_["name"], "tiles": "%d,%d,%d" % tuple(r_["tiles"]), "form": r_["form"], "us": r_["us"],  # noqa: E731
                                   "value": r_["gflops"], "unit": "GFLOP/s", "frac_of_f32_mfma_peak": round(r_["frac_of_peak"], 4),
                                   "kernel": r_["kernel_name"], "shape": "%dx%dx%d" % (r_["M"], r_["layers"][1], r_["layers"][0]),
                                   "reference_config": r_["cite"]}
                met = [r_ for r_ in rows if r_["frac_of_peak"] >= 0.45 or r_["us"] / (len(r_["layers"]) - 1) <= 5.0]
                others.append({"workload": "the reference's benchmark shape set, f32 (tools/refbench.py: %d rows = 36 benchmarks x {tile invokes, whole layer})" % len(rows),
                               "rows": len(rows), "rows_at_0.45_of_peak_or_within_2x_the_launch_floor": len(met),
                               "median_frac_of_f32_mfma_peak": round(rows[len(rows) // 2]["frac_of_peak"], 4),
                               "worst3": [keep(r_) for r_ in rows[:3]], "best3": [keep(r_) for r_ in rows[-3:]],
                               "full_table": "profiles/r05_refbench.txt"})
                # the hand-written benchmark files (benchmarks/config/base/mha.json, pack.json -> benchmarks/mlir/*.mlir) as the xsmm call
                # scripts the reference's conversion test pins for them (tools/tpp_replay --script; GFLOP/s from each file's own
                # BENCH_TOTAL_FLOPS line - the pack files count bytes there)
                if scripts:
                    others.append({"workload": "benchmarks/mlir/*.mlir as xsmm call scripts through the tile queue, f32 (mha pieces: 512 (batch, head) tiles each; "
                                               "pack / unpack: per-block copies)",
                                   "rows": [{"file": r_["cite"], "invokes_per_call": r_["invokes"], "us": r_["us"], "host_side_us": r_["host_us"], "value": r_["gflops"],
                                             "unit": "GB/s moved one way (the file's BENCH_TOTAL_FLOPS counts bytes)" if r_["script"].startswith(("pack", "unpack")) else "GFLOP/s",
                                             "kernel": r_["kernel_name"]} for r_ in scripts],
                                   "note": "fp32-query-times-key.mlir = transpose + gemm per tile through ONE temporary: the transposes are folded into the gemms "
                                           "(xsmm_hip_set_fold_transpose; 3.6 ms as 1024 launches without)"})
            except Exception as ex:
                others.append({"workload": "the reference's benchmark shape set (tools/refbench.py)", "error": str(ex)[:300]})
            # the same fp32 MLP at batch 512 as whole-layer calls: three launches, and handed over together (ONE launch of the f32
            # layer chain, bit-identical to the three: tests/test_chain_f32_gpu.py)
            for label, extra in (("three whole-layer launches", ["--whole-layer"]), ("ONE chain launch (xsmm_hip_fused_brgemm_chain_invoke)", ["--chain"])):
                r = subprocess.run([replay, "--batch", "512", "--layers", "1024,1024,1024,1024", "--bias", "--relu", "-n", "1000"] + extra,
                                   capture_output=True, text=True, timeout=300)
                mm = re.search(r"mean ([0-9.]+) us[^,]*, ([0-9.]+) GFLOP/s", r.stderr)
                if mm:
                    others.append({"workload": "mlir-gen mlp 3x1024 bs=512 bias+relu fp32, " + label + " (tools/tpp_replay)",
                                   "value": float(mm.group(2)), "unit": "GFLOP/s", "us_per_step": float(mm.group(1))})

    cpu = per_launch = mfma_busy = None
    traffic, traffic_source, traffic_detail = None, None, None
    if rank == 0 and world == 1:
        kname = "brgemm_f32"
        if not args.no_pmc:
            traffic, traffic_detail = live_traffic(kname, args.init)
            traffic_source = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/c2_probe in this run"
        if traffic is None:
            why = traffic_detail
            traffic, src = committed_traffic(kname)
            traffic_source = "committed profile %s (%s)" % (src, why if why else "--no-pmc") if traffic is not None else None
            traffic_detail = None
        mfma_busy = None if args.no_pmc else live_mfma_busy(kname, args.init, kernel_s * 1e6)
        per_launch = per_launch_figures(args.init)
        if not args.no_cpu_baseline:
            hA, hB, hC = inputs(args.init)
            cpu = cpu_baseline(args.cpu_seconds, hA, hB, hC)
            if others:  # the CPU port on the shapes of the reference's benchmark set, beside the GPU rows of other_configs
                try:
                    cpu["refbench_shapes"] = cpu_refbench_rows()
                    for o_ in others:
                        for r_ in o_.get("worst3", []) + o_.get("best3", []):
                            cr = cpu["refbench_shapes"].get(r_["shape"])
                            if cr:
                                r_["cpu_port_gflops"], r_["cpu_threads"] = cr["gflops"], cr["threads"]
                except Exception as ex:
                    cpu["refbench_shapes"] = {"error": str(ex)[:200]}

    # who took part: the process group's own count and every rank's device (N distinct GPUs, or the one-device test rig)
    group_info = None
    if use_dist:
        pr = torch.cuda.get_device_properties(local)
        me = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device_index": local, "device": pr.name,
              "pci_bus_id": getattr(pr, "pci_bus_id", None), "uuid": str(getattr(pr, "uuid", "")) or None,
              "visible": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")}
        everyone = [None] * world
        dist.all_gather_object(everyone, me, group=ctl_group)
        group_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": everyone,
                      "distinct_devices": len({(e["device_index"], e["pci_bus_id"], e["uuid"]) for e in everyone}),
                      "one_device_test_rig": one_device}
        for pg_ in list((peer_objs if not args.no_mlp else {}).values()):
            if pg_ is not None:
                pg_.close()  # (collective: every rank reaches this point)
    # RCCL writes its version banner to the C stdout buffer: shut the process group down and flush C
    # stdio first, so that the JSON line is the LAST thing on stdout
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        roof = roofline_of(args.init)
        roof.update({"traffic": traffic, "traffic_unit": "bytes/launch (HBM side, PMC)", "traffic_source": traffic_source,
                     "traffic_detail": traffic_detail, "mfma_busy": mfma_busy, "algorithmic_bytes": 3 * 4 * 1024 * 1024,
                     "traffic_floor_8_private_L2": 8 * (1 + 2) * 1024 * 1024 + 4 * 1024 * 1024,
                     "note": "achieved = 2*m*n*k*br / (HIP-event time of the K timed launches / K) on the launch stream; "
                             "traffic_floor = 8 XCDs x (1/4 of A + 1/2 of B) read + C written: every private L2 fetches its own panels"})
        line = {
            "metric": "GFLOP/s on BRGEMM 1024^3 fp32 br=16 (xsmm_brgemm_invoke)", "value": round(value, 1),
            "unit": "GFLOP/s", "n_gpus": world, "steps": K, "warmup": W,
            "untimed_before_timed_region": {"spin_up_s": runs[args.init]["spin_up_s"], "spin_up_launches": runs[args.init]["spin_up_launches"],
                                            "warmup_launches": W, "why": "clocks settle only after tens of ms of load (a 20 us kernel timed cold "
                                            "reads 10 % low); the same kernel, nothing is cached between launches"},
            "ms_per_step": round(wall / K * 1e3, 6), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BRGEMM 1024x1024x1024 fp32, batch-reduce=16 (m=n=1024 k=64 lda=ldb=ldc=1024 "
                                   "stride_a=64 stride_b=65536 BETA_0), one independent problem per GPU",
                       "launch": mode, "kernel": rt.kernel_name(h), "flops_per_step_per_gpu": flops,
                       "inputs": inputs_text[args.init]},
            "roofline": roof,
            "roofline_" + other: roofline_of(other),
            "parity": {args.init: runs[args.init]["parity"], other: runs[other]["parity"]},
            "per_launch": per_launch,
            "cpu_baseline": cpu,
        }
        if others:
            hp = [o for o in others if str(o.get("workload", "")).startswith("C2 through HOST pointers")]
            line["deployment"] = {
                "headline_needs": "device-resident operands (hipMalloc) + asynchronous mode (TPP_HIP_ASYNC=1 / xsmm_hip_set_async): a HARNESS change "
                                  "(tpp-run allocates memref globals / malloc: lib/TPP/Runner/MLIRBench.cpp:176-246), not a compiler or IR change",
                "unmodified_harness_host_pointers_gflops": hp[0]["value"] if hp else None,
                "unmodified_harness_us_per_invoke": hp[0]["us_per_step"] if hp else None,
                "host_cache": host_cache,
                "host_cache_mlp_tiles_us": next((o.get("us_median_of_5_loops", o["us_per_step"]) for o in others
                                                 if "host buffers, TPP_HIP_ASYNC=1 TPP_HIP_TILE_QUEUE=1 TPP_HIP_HOST_CACHE=1 (" in str(o.get("workload"))), None),
                "device_pointers_mlp_tiles_us": next((o["us_per_step"] for o in others
                                                      if str(o.get("workload", "")).startswith("mlir-gen mlp 3x1024 bs=256 bias+relu (fp32 unless noted), tile queue, tiles 32,32,32 (")), None),
                "note": "through HOST pointers every synchronous invoke mirrors its operands over PCIe (12 MiB up, 4 MiB down for C2): the same "
                        "BRGEMM then runs at the rate above - about 1/17 of the headline value and ~1.5x the CPU row. host_cache: the same host "
                        "buffers with TPP_HIP_HOST_CACHE=1 (environment only; with TPP_HIP_ASYNC=1 the loop runs at the device-pointer rate, "
                        "outputs on the host at perf_stop_timer) - INTEGRATION.md section 3"}
        if mlp is not None:
            line["mlp"] = mlp
        if others:
            line["other_configs"] = others
        if group_info is not None:
            line["process_group"] = group_info
        if world > 1 and mlp is not None:
            # N > 1: the headline of the line is BASELINE's multi-GPU workload - the strong-scaled bs = 4096 MLP with the all-gather of
            # the output inside the timed step (north_star: "tile-sharded across the GPUs with a RCCL all-gather").
            # The weak-scaled C2 (one independent BRGEMM per GPU, no communication: N x by construction) moves to `c2_weak`.
            line["c2_weak"] = {k_: line[k_] for k_ in ("metric", "value", "unit", "ms_per_step", "scaling", "dtype", "config", "roofline")}
            line.update({"metric": "GFLOP/s on the 3-layer MLP 1024x3 bf16 bs=4096 (bias+relu), rows sharded over the GPUs, all-gather of the "
                                   "output inside the timed step",
                         "value": mlp["value"], "unit": "GFLOP/s", "ms_per_step": mlp["ms_per_step"], "scaling": "strong", "dtype": "bf16",
                         "config": {"workload": "BASELINE config 4: 3-layer MLP 1024->1024->1024->1024 bf16, bs=4096, %d rows per GPU, "
                                                "gather: %s" % (4096 // world, mlp["gather"]),
                                    "kernel": mlp["kernel"], "flops_per_step": mlp["flops_per_step"],
                                    "one_gpu_same_run_ms_per_step": mlp["one_gpu_same_run"]["ms_per_step"],
                                    "speedup_vs_one_gpu_same_run": mlp["speedup_vs_one_gpu_same_run"],
                                    "gathered_bit_identical": mlp["gathered_bit_identical"]},
                         "roofline": {"bound": "mfma", "achieved": round(mlp["value"] / 1e3, 2), "peak": PEAK_BF16_MFMA_TFLOPS * world,
                                      "unit": "TFLOP/s", "frac": mlp["frac_of_bf16_mfma_peak"], "traffic": None,
                                      "note": "whole-job flops / step wall time (gather included) against N x the dense bf16 MFMA peak; the "
                                              "per-kernel roofline of the shard kernels is in the N = 1 line (mlp.per_rank_step_us)"}})
            if one_device:
                line["data"] = "synthetic (TEST RIG: all ranks on ONE device - the timings of this line are meaningless)"
        emit(line)
        if mlp is not None and mlp.get("gathered_bit_identical") is False:
            sys.stderr.write("[bench] the gathered MLP output differs from the unsharded result: this run is INVALID\n")
            sys.exit(3)


if __name__ == "__main__":
    main()

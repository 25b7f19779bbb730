#!/usr/bin/env python3
"""refbench - the reference's own benchmark SHAPE SET on the GPU path (VERDICT r4 row g / north_star "GFLOP/s on the
benchmarks/mlir synthetic MLP/matmul shapes ... alongside the CPU baseline").

The shapes are the IR-GEN rows of the reference's benchmark configs (the JSON files cannot travel to the GPU box; the table below
restates their `mlir-gen` command lines and cites them):
  benchmarks/config/matmul/<M>x<N>x<K>.json:37-51  mlir-gen --kernel=args --float-type=f32|bf16 [--vnni=2|4] --batch=M --layers=K,N --tiles=m,n,k
  benchmarks/config/fc/<M>x<N>x<K>.json:40-64      the same + --bias --relu
  benchmarks/config/base/base.json:34-111          mlir-gen --kernel=const --batch=256 --layers=1024,1024,1024,1024 --tiles=32,32,32 (gemm / mlp)
mlir-gen's --tiles=a,b,c are (batch tile, OUTPUT-feature tile, INPUT-feature tile) = the (m, n, k) of the tile BRGEMM
(tools/mlir-gen/MLIRGen.cpp:641-676); the file name <M>x<N>x<K> is (batch, out features, in features).

Each row is replayed by tools/tpp_replay (the reference's timing loop: one timer around N calls of the kernel) in two forms:
  tiles : as the compiler emits it - one fused_brgemm dispatch [m,n,k,k,n,n,m*k,k*n], (M/m)*(N/n) invokes with br = K/k on the packed
          block layouts, through the runtime's tile queue (one grouped launch per layer)
  whole : ONE whole-layer dispatch per layer on the flat tensors (k = 64 chunks, br = K/64)
for f32, bf16 + VNNI-2 and bf16 + VNNI-4. GFLOP/s = BENCH_TOTAL_FLOPS / mean (MLIRGen.cpp:313-334). This tool measures the GPU side
only; the CPU column (the reference's packed 32x32x32 call structure under OpenMP on the same shape: the `cpu_baseline` leg of
bench.py, which alone may use oracle/) is read from --cpu-json when given - `python bench.py --refbench` produces the whole table.

Plus (round 5) the hand-written files of benchmarks/config/base/mha.json and pack.json - benchmarks/mlir/fp32-projection.mlir,
fp32-query-times-key.mlir, fp32-out-softmax-times-value.mlir, the pack / unpack files - as the xsmm call scripts of tools/tpp_replay --script
(form "script"; the pack rows' "GFLOP/s" counts bytes as their BENCH_TOTAL_FLOPS lines do, their fraction is of the HBM peak).

usage: python tools/refbench.py [--quick] [--only matmul|fc|base|mlir] [-n ITER] [--json out.json] [--cpu-json cpu_rows.json]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = {"f32": 157.3e12, "bf16": 2500e12}

# (M = batch, N = out features, K = in features, (tile m, n, k)) - benchmarks/config/matmul/*.json and fc/*.json (17 files each, same shapes)
SHAPES = [
    (1024, 1024, 512, (64, 64, 64)), (1024, 2560, 1024, (64, 64, 64)), (1024, 352, 512, (32, 32, 32)), (1024, 512, 256, (64, 64, 64)),
    (128, 1024, 1024, (64, 64, 64)), (128, 1024, 4096, (64, 64, 64)), (128, 3072, 768, (64, 64, 64)), (128, 4096, 1024, (64, 64, 64)),
    (128, 768, 2304, (64, 48, 64)), (128, 768, 3072, (32, 48, 32)), (128, 768, 768, (32, 64, 64)),
    (256, 1024, 1024, (64, 64, 64)), (256, 1024, 4096, (64, 64, 64)), (256, 3072, 768, (64, 64, 64)), (256, 4096, 1024, (64, 64, 64)),
    (256, 768, 3072, (64, 64, 64)), (256, 768, 768, (64, 64, 64)),
]


# the hand-written benchmark files (benchmarks/config/base/mha.json:1-40, pack.json): benchmarks/mlir/*.mlir as the xsmm call scripts the
# reference's own conversion test pins for them (test/Conversion/LinalgToXsmm/linalg-to-gemm.mlir:46-62,91-104,132-145; tools/tpp_replay
# --script): name, file, BENCH_TOTAL_FLOPS of the file (the pack files count bytes), invokes per call
SCRIPTS = [
    ("mha_projection", "fp32-projection.mlir", 1073741824.0, 512),
    ("mha_qk", "fp32-query-times-key.mlir", 67108864.0, 1024),
    ("mha_sv", "fp32-out-softmax-times-value.mlir", 67108864.0, 512),
    ("pack_a", "fp32-pack-gemm-operand-a-512x1024.mlir", 2097152.0, 512),
    ("pack_b", "fp32-pack-gemm-operand-b-512x1024.mlir", 2097152.0, 512),
    ("unpack_c", "fp32-unpack-gemm-operand-a-512x512.mlir", 1048576.0, 256),
]


# the xsmm dispatches per (batch, head) tile of the three mha files, as the reference's conversion test has them (the zero fill of the
# output tile that precedes each gemm there is folded into BETA_0 by the default pipeline: FoldXsmmFlags, LinalgLowering.cpp:56)
SCRIPT_CALLS = {
    "mha_projection": [("gemm", [32, 64, 512, 512, 512, 512])],
    "mha_qk": [("unary transpose", [32, 64, 512, 32]), ("gemm", [32, 32, 64, 512, 32, 32])],
    "mha_sv": [("gemm", [32, 64, 32, 32, 512, 512])],
}


def cases(only):
    out = []
    if not only or only == "mlir":
        for nm, f, fl, inv in SCRIPTS:
            out.append({"family": "mlir", "name": "mlir_" + nm, "script": nm, "M": 0, "layers": [0, 0], "tiles": (32, 64 if nm != "mha_qk" else 32, 0),
                        "dtype": "f32", "form": "script", "bias_relu": False, "kernel": "const", "flops": fl, "invokes": inv,
                        "cite": "benchmarks/mlir/" + f})
    if only == "mlir":
        return out
    for fam in ("matmul", "fc"):
        if only and only != fam:
            continue
        for (M, N, K, t) in SHAPES:
            for dt in ("f32", "bf16-vnni2", "bf16-vnni4"):
                for form in ("tiles", "whole"):
                    out.append({"family": fam, "name": "%s_%dx%dx%d" % (fam, M, N, K), "M": M, "layers": [K, N], "tiles": t, "dtype": dt,
                                "form": form, "bias_relu": fam == "fc", "kernel": "args",
                                "cite": "benchmarks/config/%s/%dx%dx%d.json" % (fam, M, N, K)})
    if not only or only == "base":
        for nm, br in (("gemm", False), ("mlp", True)):
            for dt in ("f32", "bf16-vnni2", "bf16-vnni4"):
                for form in ("tiles", "whole"):
                    out.append({"family": "base", "name": "base_%s_256x1024x3" % nm, "M": 256, "layers": [1024] * 4, "tiles": (32, 32, 32),
                                "dtype": dt, "form": form, "bias_relu": br, "kernel": "const", "cite": "benchmarks/config/base/base.json:34-111"})
    return out


def argv_of(c, n_iter, repeats=1):
    if c.get("script"):
        return ["--script", c["script"], "--queue", "1", "-n", str(n_iter)]
    a = ["--batch", str(c["M"]), "--layers", ",".join(map(str, c["layers"])), "--kernel", c["kernel"], "-n", str(n_iter), "--queue", "1"]
    if repeats > 1:
        a += ["--repeats", str(repeats)]
    if c["form"] == "tiles":
        a += ["--tiles", "%d,%d,%d" % c["tiles"]]
    else:
        a += ["--whole-layer"]
    if c["bias_relu"]:
        a += ["--bias", "--relu"]
    if c["dtype"].startswith("bf16"):
        a += ["--bf16", "--vnni", c["dtype"][-1]]
    return a


def flops_of(c):
    if c.get("script"):
        return c["flops"]
    f = 0.0
    for l in range(len(c["layers"]) - 1):
        f += 2.0 * c["M"] * c["layers"][l] * c["layers"][l + 1] + (2.0 * c["M"] * c["layers"][l + 1] if c["bias_relu"] else 0.0)
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="f32 only")
    ap.add_argument("--no-cpu", action="store_true", help="(accepted for older command lines; the CPU column comes from --cpu-json only)")
    ap.add_argument("--cpu-json", default="", help='{"MxNxK": {"gflops": g, "threads": t}, ...} from bench.py\'s cpu_baseline leg')
    ap.add_argument("--only", default="")
    ap.add_argument("-n", type=int, default=300)
    ap.add_argument("--json", default="")
    ap.add_argument("--repeats", type=int, default=3, help="timed loops per row (own timer each): the row gets us_min / us_median / us_max and "
                    "the tile queue's abandoned-replay count (VERDICT r5 weak 9: a row must say whether its mean is typical)")
    args = ap.parse_args()
    replay = os.path.join(ROOT, "tools", "tpp_replay")
    cs = [c for c in cases(args.only) if not (args.quick and c["dtype"] != "f32")]
    # the script rows last: they run from a second calling context only in their own cases, and the multi-caller state of the tile
    # queue is process-wide
    cs = [c for c in cs if not c.get("script")] + [c for c in cs if c.get("script")]
    with tempfile.NamedTemporaryFile("w", suffix=".cases", delete=False) as f:
        for c in cs:
            f.write(" ".join(argv_of(c, args.n, args.repeats)) + "\n")
        path = f.name
    r = subprocess.run([replay, "--cases", path], capture_output=True, text=True, timeout=3000)
    os.unlink(path)
    pat = re.compile(r"mean ([0-9.]+) us \(host side of the invokes ([0-9.]+) us\), ([0-9.]+) GFLOP/s \(BENCH_TOTAL_FLOPS ([0-9]+)\), kernel (.*)$")
    got, reps = [], {}
    rpat = re.compile(r"repeats (\d+) x \d+ calls: min ([0-9.]+) median ([0-9.]+) max ([0-9.]+) us; replays abandoned in the repeats (-?\d+)")
    for l in r.stderr.splitlines():
        if "GFLOP/s (BENCH_TOTAL_FLOPS" in l:
            got.append(pat.search(l))
        elif rpat.search(l):
            reps[len(got)] = rpat.search(l)  # (tpp_replay prints a case's repeats line BEFORE its result line)
    if len(got) != len(cs):
        sys.stderr.write(r.stderr[-4000:])
        raise SystemExit("refbench: %d cases, %d result lines (rc %d)" % (len(cs), len(got), r.returncode))
    for i_, (c, m_) in enumerate(zip(cs, got)):
        if i_ in reps:
            c["repeats"], c["us_min"], c["us_median"], c["us_max"], c["replays_abandoned"] = (int(reps[i_].group(1)), float(reps[i_].group(2)),
                                                                                               float(reps[i_].group(3)), float(reps[i_].group(4)), int(reps[i_].group(5)))
        c["us"], c["host_us"], c["gflops"], c["kernel_name"] = float(m_.group(1)), float(m_.group(2)), float(m_.group(3)), m_.group(5).split("; result checked")[0].strip()
        if c.get("script", "").startswith(("pack", "unpack")):
            c["kernel_name"] = "unary_grouped_kernel<f32>"  # (the line names the last grouped GEMM launch; these scripts run none)
        assert abs(float(m_.group(4)) - flops_of(c)) < 1, (c, m_.group(4))
        c["frac_of_peak"] = c["gflops"] * 1e9 / PEAK["f32" if c["dtype"] == "f32" else "bf16"]
        if c.get("script", "").startswith(("pack", "unpack")):  # bytes: read + written once each against the HBM peak (8 TB/s)
            c["frac_of_peak"] = 2.0 * c["gflops"] * 1e9 / 8e12
    cpu = {}
    if args.cpu_json:
        with open(args.cpu_json) as f:
            cpu = {tuple(int(x) for x in k.split("x")): v for k, v in json.load(f).items()}
    print("# refbench: %d rows, tpp_replay -n %d; GPU peaks f32 157.3 TF, bf16 2500 TF (dense MFMA); empty-launch floor ~2.5 us" % (len(cs), args.n))
    print("# us = mean of the FIRST timed loop (the reference's figure: one timer around N calls); median / max = over --repeats loops of the same "
          "N calls; ab = replays the tile queue abandoned in the repeats (0 = every iteration replayed its recorded group)")
    print("# %-24s %-11s %-10s %-6s %9s %8s %8s %3s %10s %7s %9s  %s" % ("benchmark", "tiles", "dtype", "form", "us", "median", "max", "ab", "GFLOP/s", "frac", "CPU GF/s", "kernel"))
    for c in cs:
        key = (c["M"], c["layers"][1], c["layers"][0])
        cg = cpu.get(key if c["family"] != "base" else (256, 1024, 1024)) if not c.get("script") else None
        c["cpu_port_gflops_f32"] = round(cg["gflops"], 1) if cg else None
        c["cpu_threads"] = cg["threads"] if cg else None
        print("%-26s %-11s %-10s %-6s %9.2f %8s %8s %3s %10.1f %7.4f %9s  %s" % (
            c["name"], "-" if c.get("script") else "%d,%d,%d" % c["tiles"], c["dtype"], c["form"], c["us"],
            ("%.2f" % c["us_median"]) if "us_median" in c else "-", ("%.2f" % c["us_max"]) if "us_max" in c else "-",
            c.get("replays_abandoned", "-"), c["gflops"], c["frac_of_peak"],
            ("%.1f" % cg["gflops"]) if cg else "-", c["kernel_name"]), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(cs, f, indent=0)


if __name__ == "__main__":
    main()

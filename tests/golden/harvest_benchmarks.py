#!/usr/bin/env python3
"""Harvests the IR-GEN rows of the reference's benchmark configs (benchmarks/config/matmul/*.json, fc/*.json, base/base.json) into
tests/golden/benchmark_configs.json: DATA only - per row the mlir-gen options (batch, layers, tiles, float type, vnni, bias, relu,
kernel) and where it stands (file:line). tools/refbench.py restates these rows (the JSON files cannot travel to the GPU box);
tests/test_host_logic.py checks its table against this fixture. Run here, where /root/reference exists:
    python tests/golden/harvest_benchmarks.py"""
import glob
import json
import os
import re

REF = "/root/reference/benchmarks/config"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "benchmark_configs.json")


def rows_of(path):
    out = []
    text = open(path).read().splitlines()
    for ln, line in enumerate(text, 1):
        m = re.search(r'"benchmark":\s*\[\s*"mlir-gen",\s*"([^"]*)"', line)
        if not m:
            continue
        opts = m.group(1)
        g = lambda k, d=None: (re.search(r"--%s=(\S+)" % k, opts) or [None, d])[1]  # noqa: E731
        out.append({"where": "%s:%d" % (os.path.relpath(path, "/root/reference"), ln), "kernel": g("kernel"), "float_type": g("float-type"),
                    "batch": int(g("batch")), "layers": [int(x) for x in g("layers").split(",")],
                    "tiles": [int(x) for x in g("tiles").split(",")] if g("tiles") else None, "vnni": int(g("vnni", "0")),
                    "bias": "--bias" in opts, "relu": "--relu" in opts})
    return out


def mlir_files():
    """the hand-written benchmark files of base/mha.json and base/pack.json: file, its BENCH_TOTAL_FLOPS line, its -n flag; and the xsmm
    dispatches the reference's conversion test pins for the three mha functions (test/Conversion/LinalgToXsmm/linalg-to-gemm.mlir CHECK lines)"""
    out = {"files": [], "lowered_calls": {}}
    for cfgf in ("mha.json", "pack.json"):
        for grp in json.load(open(os.path.join(REF, "base", cfgf))):
            for suite, rows in grp.items():
                for name, row in rows.items():
                    if row.get("type") != "MLIR":
                        continue
                    f = os.path.join("/root/reference/benchmarks/mlir", row["benchmark"])
                    m = re.search(r"BENCH_TOTAL_FLOPS:\s*(\d+)", open(f).read()) if os.path.exists(f) else None  # (mha.json also names a file the tree does not hold)
                    out["files"].append({"config": "benchmarks/config/base/%s" % cfgf, "suite": suite, "name": name, "file": "benchmarks/mlir/" + row["benchmark"],
                                         "flags": row.get("flags"), "in_tree": os.path.exists(f), "bench_total_flops": int(m.group(1)) if m else None})
    test = "/root/reference/test/Conversion/LinalgToXsmm/linalg-to-gemm.mlir"
    cur = None
    for ln, line in enumerate(open(test).read().splitlines(), 1):
        m = re.search(r"CHECK-LABEL:\s*(\w+)", line)
        if m:
            cur = m.group(1) if m.group(1).startswith("mha_") else None
        m = re.search(r"CHECK:.*xsmm\.(gemm|unary|brgemm)\.dispatch\s*(\w+)?\s*\[([0-9, ]+)\]\s*flags = \(([a-z_0-9, ]*)\)", line)
        if m and cur:
            out["lowered_calls"].setdefault(cur, []).append({"where": "test/Conversion/LinalgToXsmm/linalg-to-gemm.mlir:%d" % ln, "op": m.group(1),
                                                             "kind": m.group(2), "dims": [int(x) for x in m.group(3).split(",")], "flags": m.group(4)})
    return out


def main():
    data = {"matmul": [], "fc": [], "base": rows_of(os.path.join(REF, "base", "base.json")), "mlir": mlir_files()}
    for fam in ("matmul", "fc"):
        for f in sorted(glob.glob(os.path.join(REF, fam, "*.json"))):
            data[fam] += rows_of(f)
    with open(OUT, "w") as f:
        json.dump(data, f, indent=1)
    print(OUT, {k: len(v) for k, v in data.items()})


if __name__ == "__main__":
    main()

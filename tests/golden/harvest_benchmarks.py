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


def main():
    data = {"matmul": [], "fc": [], "base": rows_of(os.path.join(REF, "base", "base.json"))}
    for fam in ("matmul", "fc"):
        for f in sorted(glob.glob(os.path.join(REF, fam, "*.json"))):
            data[fam] += rows_of(f)
    with open(OUT, "w") as f:
        json.dump(data, f, indent=1)
    print(OUT, {k: len(v) for k, v in data.items()})


if __name__ == "__main__":
    main()

"""Replays a golden fixture (tests/golden/*.json) on a backend and checks the result.

A backend offers the six invoke-level ops of the xsmm C-ABI over flat numpy
buffers + element offsets. Two backends exist:
  OracleBackend - oracle/liboracle.so (the CPU restatement; the checker)
  AbiBackend    - the product library through its C-ABI (tests/test_parity_gpu.py)
"""
import glob
import json
import math
import os

import numpy as np

from oracle import pyoracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32, BF16 = 1, 2


def fixtures():
    names = sorted(glob.glob(os.path.join(GOLDEN, "*.json")))
    return [n for n in names if os.path.basename(n) not in ("flops.json", "xsmm_to_func_wire.json", "benchmark_configs.json")]


def load(path):
    with open(path) as f:
        return json.load(f)


def make_buffers(fx):
    bufs = {}
    for name, spec in fx["buffers"].items():
        dt = spec["dtype"]
        if "data" in spec:
            v = np.array(spec["data"], dtype=np.float32)
        else:
            v = np.full(spec["size"], spec["const"], dtype=np.float32)
        bufs[name] = (dt, v if dt == F32 else orc.f32_to_bf16(v))
    return bufs


def printed_tol(want):
    """vector.print shows 6 significant digits: half a unit of the 6th digit (+ f32 slack)"""
    if want == 0.0:
        return 5e-7
    return 0.5 * 10.0 ** (math.floor(math.log10(abs(want))) - 5) * 1.02 + abs(want) * 2e-7


class OracleBackend:
    name = "oracle"

    def gemm(self, d, A, oa, B, ob, C, oc):
        orc.gemm(d["dtype"], d["m"], d["n"], d["k"], d["lda"], d["ldb"], d["ldc"], d["flags"], A, oa, B, ob, C, oc)

    def brgemm(self, d, A, oa, B, ob, C, oc, br):
        orc.brgemm(d["dtype"], d["m"], d["n"], d["k"], d["lda"], d["ldb"], d["ldc"], d["stride_a"], d["stride_b"],
                   d["flags"], A, oa, B, ob, C, oc, br)

    def fused_brgemm(self, d, A, oa, B, ob, C, oc, D, od, br):
        orc.fused_brgemm(d["dtype"], d["m"], d["n"], d["k"], d["lda"], d["ldb"], d["ldc"], d["stride_a"],
                         d["stride_b"], d["flags"], d["unary_flags"], d["unary_kind"], d["binary_flags"],
                         d["binary_kind"], A, oa, B, ob, C, oc, D, od, br)

    def unary(self, d, X, ox, O, oo):
        orc.unary(d["kind"], d["dtype"], d["m"], d["n"], d["ldi"], d["ldo"], d["flags"], X, ox, O, oo)

    def binary(self, d, L, ol, R, or_, O, oo):
        orc.binary(d["kind"], d["dtype"], d["m"], d["n"], d["ldi_lhs"], d["ldi_rhs"], d["ldo"], d["flags"],
                   L, ol, R, or_, O, oo)

    def finish(self):
        pass


def run_calls(fx, backend, bufs):
    for c in fx["calls"]:
        d = c["dispatch"]
        g = lambda key: (bufs[c[key][0]][1], c[key][1])  # noqa: E731
        if c["op"] == "gemm":
            backend.gemm(d, *g("a"), *g("b"), *g("c"))
        elif c["op"] == "brgemm":
            backend.brgemm(d, *g("a"), *g("b"), *g("c"), c["batch"])
        elif c["op"] == "fused_brgemm":
            backend.fused_brgemm(d, *g("a"), *g("b"), *g("c"), *g("d"), c["batch"])
        elif c["op"] == "unary":
            backend.unary(d, *g("in"), *g("out"))
        elif c["op"] == "binary":
            backend.binary(d, *g("lhs"), *g("rhs"), *g("out"))
        else:
            raise ValueError(c["op"])
    backend.finish()


def check_expect(fx, bufs):
    for e in fx["expect"]:
        dt, arr = bufs[e["buffer"]]
        vals = arr if dt == F32 else orc.bf16_to_f32(arr)
        rows, cols, ld, off = e["rows"], e["cols"], e["ld"], e["offset"]
        got = np.array([[vals[off + i * ld + j] for j in range(cols)] for i in range(rows)], dtype=np.float64)
        if "values" in e:
            want = np.array(e["values"], dtype=np.float64).reshape(rows, cols)
        else:
            want = np.full((rows, cols), e["fill"], dtype=np.float64)
        if e["tol"] == "exact":
            bad = got != want
            tol = np.zeros_like(want)
        elif e["tol"] == "printed":
            tol = np.vectorize(printed_tol)(want)
            bad = np.abs(got - want) > tol
        else:
            tol = np.full_like(want, float(e["tol"]))
            bad = np.abs(got - want) > tol
        if bad.any():
            i, j = np.argwhere(bad)[0]
            raise AssertionError("%s: buffer %s [%d,%d] got %r want %r (tol %g); %d/%d mismatches" % (
                fx["name"], e["buffer"], i, j, got[i, j], want[i, j], tol[i, j], int(bad.sum()), bad.size))


def run_fixture(path, backend):
    fx = load(path)
    bufs = make_buffers(fx)
    run_calls(fx, backend, bufs)
    check_expect(fx, bufs)
    return bufs

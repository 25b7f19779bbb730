"""CPU-side checks of the drop-in boundary: the library loads without a GPU, exports
every symbol include/tpp_xsmm_abi.h declares (= the reference's XsmmRunnerUtils.h:22-83
+ PerfRunnerUtils.h:22-24), dispatch is cheap/idempotent and keeps the reference's
error convention (stderr + exit(-1)), and the product never links the oracle."""
import importlib
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = importlib.import_module("tpp-mlir_amd")


@pytest.fixture(scope="module")
def rt():
    pkg.build()
    return pkg.get_runtime()


def header_symbols():
    with open(os.path.join(ROOT, "include", "tpp_xsmm_abi.h")) as f:
        text = f.read()
    return re.findall(r"TPP_XSMM_EXPORT\s+[\w\s\*]+?\b(\w+)\s*\(", text)


def test_header_declares_reference_abi():
    names = set(header_symbols())
    # the 13 xsmm_* symbols of runtime/Xsmm/XsmmRunnerUtils.h:22-83 + the two perf_* timers
    want = {"xsmm_gemm_dispatch", "xsmm_unary_dispatch", "xsmm_binary_dispatch", "xsmm_brgemm_dispatch",
            "xsmm_fused_brgemm_dispatch", "xsmm_intel_amx_tile_config_dispatch", "xsmm_gemm_invoke",
            "xsmm_unary_invoke", "xsmm_unary_scalar_invoke", "xsmm_binary_invoke", "xsmm_brgemm_invoke",
            "xsmm_fused_brgemm_invoke", "xsmm_intel_amx_tile_config_invoke", "perf_start_timer", "perf_stop_timer"}
    assert want <= names
    assert want == set(pkg.REFERENCE_SYMBOLS)


def test_library_exports_every_declared_symbol(rt):
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg.library_path()], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in header_symbols() if s not in exported]
    assert not missing, missing
    for s in header_symbols():
        assert hasattr(rt.lib, s)


def test_product_does_not_link_the_oracle():
    out = subprocess.check_output(["nm", "-D", pkg.library_path()], text=True)
    assert "oracle_" not in out and "tinit_" not in out
    deps = subprocess.check_output(["readelf", "-d", pkg.library_path()], text=True)
    assert "liboracle" not in deps
    # and the package sources never import it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tpp-mlir_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                with open(os.path.join(dirpath, fn)) as f:
                    text = f.read()
                assert not re.search(r"(import\s+.*oracle|from\s+\.*oracle|liboracle|pyoracle|#include\s+.*oracle)", text), fn


def test_dispatch_is_idempotent_and_distinct(rt):
    a = rt.brgemm_dispatch(1, 32, 32, 32, 32, 32, 32, 1024, 1024, 0)
    b = rt.brgemm_dispatch(1, 32, 32, 32, 32, 32, 32, 1024, 1024, 0)
    c = rt.brgemm_dispatch(1, 32, 32, 32, 32, 32, 32, 1024, 1024, 4)
    assert a == b and a != c and a != 0
    # tile-config flag bits are ignored (IntelAMXTileConfig.cpp ORs 64|128 into every bf16 brgemm)
    d = rt.brgemm_dispatch(2, 32, 32, 32, 32, 32, 32, 1024, 1024, 2048 | 4)
    e = rt.brgemm_dispatch(2, 32, 32, 32, 32, 32, 32, 1024, 1024, 2048 | 4 | 64 | 128)
    assert d == e
    assert rt.intel_amx_tile_config_dispatch(2, 32, 32, 32, 32, 32, 32, 1024, 1024, 2048) != 0
    u = rt.unary_dispatch(5, 1, 3, 3, 3, 3, 0)
    assert u == rt.unary_dispatch(5, 1, 3, 3, 3, 3, 0)
    assert rt.binary_dispatch(1, 1, 3, 3, 3, 3, 3, 0) != u


def header_params():
    """symbol -> parameter names, in order, from include/tpp_xsmm_abi.h"""
    with open(os.path.join(ROOT, "include", "tpp_xsmm_abi.h")) as f:
        text = f.read()
    out = {}
    for m in re.finditer(r"TPP_XSMM_EXPORT\s+[\w\s\*]+?\b(\w+)\s*\(([^;]*?)\)\s*;", text, re.S):
        out[m.group(1)] = [p.split()[-1].lstrip("*") for p in m.group(2).split(",") if p.strip() and p.strip() != "void"]
    return out


def test_wire_order_matches_the_reference_filecheck_lines():
    """tests/golden/xsmm_to_func_wire.json = the argument tuples test/Conversion/XsmmToFunc/xsmm-to-func.mlir pins with
    FileCheck, next to the dialect-level op each one was lowered from. Rebuilding every tuple from the op's fields
    through the parameter NAMES of include/tpp_xsmm_abi.h must reproduce it - which pins the header's parameter order,
    the enum wire values (incl. the vnni_a <-> vnni_b exchange, ConvertXsmmToFunc.cpp:251-265) and runtime.py's arity."""
    import json
    from importlib import import_module
    rtmod = import_module("tpp-mlir_amd.runtime")
    with open(os.path.join(ROOT, "tests", "golden", "xsmm_to_func_wire.json")) as f:
        wire = json.load(f)
    params = header_params()
    dtype = {"f32": 1, "bf16": 2}
    unary_kind = {"identity": 1, "zero": 2, "relu": 5, "vnni_2": 28, "transpose": 29, "none": 0}
    binary_kind = {"add": 1, "mul": 2, "sub": 3, "div": 4, "none": 0}
    gemm_flag = {"none": 0, "beta_0": 4, "vnni_a": 4096, "vnni_b": 2048, "vnni_c": 8192}   # wire values
    unary_flag = {"none": 0, "bcast_row": 2, "bcast_col": 4, "bcast_scalar": 8}
    binary_flag = {"none": 0, "bcast_row_in0": 1, "bcast_row_in1": 2, "bcast_col_in0": 4, "bcast_col_in1": 8,
                   "bcast_scalar_in0": 16, "bcast_scalar_in1": 32}
    ored = lambda table, names: sum(table[n] for n in names)  # noqa: E731
    seen = set()
    for w in wire["dispatch"]:
        op, sym = w["op"], w["symbol"]
        v = {"dtype": dtype[op["data_type"]]}
        fam = op["family"]
        if fam in ("gemm", "brgemm", "fused_brgemm"):
            names = ["m", "n", "k", "lda", "ldb", "ldc", "stride_a", "stride_b"][:len(op["dims"])]
            v.update(dict(zip(names, op["dims"])))
            v["flags"] = v["gemm_flags"] = ored(gemm_flag, op["flags"]["flags"])
            if fam == "fused_brgemm":
                v["binary_kind"], v["unary_kind"] = binary_kind[op["fused"][0]], unary_kind[op["fused"][1]]
                v["unary_flags"] = ored(unary_flag, op["flags"]["unary_flags"])
                v["binary_flags"] = ored(binary_flag, op["flags"]["binary_flags"])
        elif fam == "unary":
            v.update(dict(zip(["m", "n", "ldi", "ldo"], op["dims"])))
            v["unary_kind"], v["flags"] = unary_kind[op["kind"]], ored(unary_flag, op["flags"]["flags"])
        else:
            v.update(dict(zip(["m", "n", "ldi_lhs", "ldi_rhs", "ldo"], op["dims"])))
            v["binary_kind"], v["flags"] = binary_kind[op["kind"]], ored(binary_flag, op["flags"]["flags"])
        assert [v[p] for p in params[sym]] == w["args"], (sym, params[sym], w)
        assert len(rtmod._SIGNATURES[sym][1]) == len(w["args"]), sym
        seen.add(sym)
    assert seen == {"xsmm_gemm_dispatch", "xsmm_brgemm_dispatch", "xsmm_fused_brgemm_dispatch", "xsmm_unary_dispatch",
                    "xsmm_binary_dispatch"}
    for inv in wire["invoke"]:  # invoke: dtype, handle, then (pointer, element offset) per memref operand [, batch]
        want = {"dtype": "int64_t", "handle": "int64_t", "off": "int64_t", "batch": "int64_t", "ptr": "void"}
        ctype = {"int64_t": rtmod.I64, "void": rtmod.VP}
        assert [ctype[want[p]] for p in inv["pattern"]] == rtmod._SIGNATURES[inv["symbol"]][1], inv
        assert len(params[inv["symbol"]]) == len(inv["pattern"])


def test_variant_selection(rt):
    assert "64x64" in rt.kernel_name(rt.brgemm_dispatch(1, 1024, 1024, 64, 1024, 1024, 1024, 64, 65536, 0))
    assert "64x32" in rt.kernel_name(rt.fused_brgemm_dispatch(1, 512, 1024, 64, 1024, 1024, 1024, 64, 65536,
                                                                4, 0, 5, 4, 1))
    assert "generic" in rt.kernel_name(rt.brgemm_dispatch(1, 3, 3, 4, 4, 3, 3, 12, 12, 0))
    assert "bf16" in rt.kernel_name(rt.brgemm_dispatch(2, 4096, 1024, 64, 1024, 1024, 1024, 64, 65536, 2052))


BAD_CALLS = {
    "bad_dtype": "rt.brgemm_dispatch(7, 4, 4, 4, 4, 4, 4, 16, 16, 0)",
    "lda_lt_k": "rt.gemm_dispatch(1, 4, 4, 8, 4, 4, 4, 0)",          # XsmmOps.cpp:335-340
    "vnni_f32": "rt.brgemm_dispatch(1, 4, 4, 4, 4, 4, 4, 16, 16, 2048)",  # XsmmOps.cpp:292-298
    "vnni_a_f32": "rt.brgemm_dispatch(1, 4, 4, 4, 4, 4, 4, 16, 16, 4096)",
    "vnni_odd_k": "rt.gemm_dispatch(2, 1, 2, 3, 4, 5, 6, 6144)",        # xsmm-to-func.mlir:62 tuple: k = 3 cannot be VNNI-2 packed
    "vnni_c_odd_m": "rt.gemm_dispatch(2, 3, 4, 4, 4, 4, 4, 8192)",
    "unknown_gemm_flag": "rt.gemm_dispatch(1, 4, 4, 4, 4, 4, 4, 1)",
    "fused_mul": "rt.fused_brgemm_dispatch(1, 4, 4, 4, 4, 4, 4, 16, 16, 0, 0, 5, 4, 2)",
    "fused_bcast_row": "rt.fused_brgemm_dispatch(1, 4, 4, 4, 4, 4, 4, 16, 16, 0, 0, 5, 1, 1)",
    "unary_kind": "rt.unary_dispatch(17, 1, 4, 4, 4, 4, 0)",
    "vnni2_f32": "rt.unary_dispatch(28, 1, 4, 4, 4, 4, 0)",
    "vnni2_odd": "rt.unary_dispatch(28, 2, 3, 4, 4, 4, 0)",
    "binary_kind": "rt.binary_dispatch(9, 1, 4, 4, 4, 4, 4, 0)",
    "wrong_handle_kind": "rt.unary(1, rt.brgemm_dispatch(1,4,4,4,4,4,4,16,16,0), 0, 0, 0, 0)",
    "dtype_mismatch": "rt.brgemm(2, rt.brgemm_dispatch(1,4,4,4,4,4,4,16,16,0), 0, 0, 0, 0, 0, 0, 1)",
}


@pytest.mark.parametrize("case", sorted(BAD_CALLS))
def test_error_convention_is_stderr_plus_exit_minus_one(rt, case):
    """failed dispatch = message on stderr + exit(-1) (XsmmRunnerUtils.cpp:132-137)"""
    code = ("import importlib,sys; sys.path.insert(0, %r); pkg = importlib.import_module('tpp-mlir_amd'); "
            "rt = pkg.get_runtime(); %s; print('SURVIVED')" % (ROOT, BAD_CALLS[case]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 255, (r.returncode, r.stdout, r.stderr)
    assert "SURVIVED" not in r.stdout and r.stderr.strip()


def test_invoke_without_gpu_fails_loudly(rt):
    if rt.device_count() > 0:
        pytest.skip("a GPU is visible")
    code = ("import importlib,sys,numpy as np; sys.path.insert(0, %r); pkg = importlib.import_module('tpp-mlir_amd'); "
            "rt = pkg.get_runtime(); x = np.ones(9, np.float32); h = rt.unary_dispatch(5,1,3,3,3,3,0); "
            "rt.unary(1, h, x, 0, x, 0); print('SURVIVED')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "SURVIVED" not in r.stdout
    assert "no CPU fallback" in r.stderr


def test_perf_timers(rt):
    t0 = rt.perf_start_timer()
    dt = rt.perf_stop_timer(t0)
    assert 0.0 <= dt < 5.0

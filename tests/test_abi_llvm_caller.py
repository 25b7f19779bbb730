"""SURVEY.md 8 f1 (substitute - there is no MLIR toolchain in the image): a caller at the LLVM calling-convention level.

tests/abi/xsmm_calls.ll is hand-written LLVM IR with exactly the signatures the reference's lowering emits
(ConvertXsmmToFunc.cpp:37-78, 298-352; FileCheck'd in test/Conversion/XsmmToFunc/xsmm-to-func.mlir) - all-i64 scalars, (ptr, i64)
memref pairs, the `float` of xsmm_unary_scalar_invoke, the 14-argument fused dispatch whose tail travels on the stack - and
tests/abi/driver.c stands where tpp-run stands. Neither includes this repository's header. The module is compiled with the ROCm
clang and linked against libtpp_xsmm_runner_utils.so with --no-as-needed, as tools/tpp-run/CMakeLists.txt:74-86 links the
reference's library. CPU: it builds, and every symbol it needs is exported by the library. GPU: the golden fixtures harvested from
the reference's lit tests run through it on HOST memrefs (synchronous invokes, what JIT'd code gets) and match the expected values.
"""
import importlib
import os
import shutil
import subprocess

import numpy as np
import pytest

import fixture_runner as fr
from oracle import pyoracle as orc

pkg = importlib.import_module("tpp-mlir_amd")
HERE = os.path.dirname(os.path.abspath(__file__))
ABI = os.path.join(HERE, "abi")
CLANG = "/opt/rocm/lib/llvm/bin/clang"

# entry function of the module -> (fixture, buffer names in argument order)
CASES = {
    "fusion_f32": ("xsmm_fusion_seed123", ["A", "B", "C", "bias"]),
    "quarternary_bf16_amx": ("xsmm_quarternary_bf16", ["A", "B", "C", "D"]),
    "brgemm_bf16_amx": ("xsmm_brgemm_bf16", ["A", "B", "C"]),
    "gemm_bf16": ("xsmm_gemm_bf16", ["A", "B", "C"]),
    "zero_f32": ("xsmm_zero", ["X"]),
    "binary_add_f32": ("xsmm_binary_add", ["L", "R", "O"]),
}


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("no ROCm clang in this image")
    so = pkg.build()
    out = tmp_path_factory.mktemp("abi")
    obj, exe = str(out / "xsmm_calls.o"), str(out / "abi_driver")
    subprocess.run([CLANG, "-c", os.path.join(ABI, "xsmm_calls.ll"), "-o", obj], check=True, capture_output=True)
    libdir = os.path.dirname(so)
    cc = shutil.which("gcc") or CLANG
    subprocess.run([cc, "-O1", os.path.join(ABI, "driver.c"), obj, "-o", exe, "-Wl,--no-as-needed", "-L", libdir,
                    "-ltpp_xsmm_runner_utils", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True)
    return obj, exe, so


def test_llvm_module_links_against_the_library(driver):
    """every undefined symbol of the IR module is an exported function of the product library, and the module was not built from
    this repository's header (it declares the reference's signatures itself)"""
    obj, exe, so = driver
    und = {l.split()[-1] for l in subprocess.run(["nm", "-u", obj], check=True, capture_output=True, text=True).stdout.splitlines() if l.strip()}
    exported = {l.split()[-1] for l in subprocess.run(["nm", "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout.splitlines()
                if " T " in l}
    assert und and und <= exported, sorted(und - exported)
    assert {"xsmm_fused_brgemm_dispatch", "xsmm_unary_scalar_invoke", "xsmm_intel_amx_tile_config_invoke", "perf_stop_timer"} <= und
    includes = [l.strip() for l in open(os.path.join(ABI, "driver.c")) if l.lstrip().startswith("#include")]
    assert includes == ["#include <stdio.h>", "#include <stdlib.h>", "#include <string.h>"], includes
    needed = subprocess.run(["readelf", "-d", exe], check=True, capture_output=True, text=True).stdout
    assert "libtpp_xsmm_runner_utils.so" in needed


def _run(exe, entry, blobs, tmp_path):
    fin, fout = str(tmp_path / (entry + ".in")), str(tmp_path / (entry + ".out"))
    with open(fin, "wb") as f:
        for b in blobs:
            f.write(b.tobytes())
    env = dict(os.environ)
    env.pop("TPP_HIP_ASYNC", None)  # JIT'd code reads its outputs right after the invoke: synchronous mode
    r = subprocess.run([exe, entry, fin, fout] + [str(b.nbytes) for b in blobs], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, (entry, r.returncode, r.stdout[-400:], r.stderr[-400:])
    raw = open(fout, "rb").read()
    outs, pos = [], 0
    for b in blobs:
        outs.append(np.frombuffer(raw[pos:pos + b.nbytes], dtype=b.dtype).copy())
        pos += b.nbytes
    return outs, r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("entry", sorted(CASES))
def test_golden_fixture_through_the_llvm_caller(driver, entry, tmp_path):
    _, exe, _ = driver
    name, order = CASES[entry]
    fx = fr.load(os.path.join(fr.GOLDEN, name + ".json"))
    bufs = fr.make_buffers(fx)
    outs, _ = _run(exe, entry, [bufs[n][1] for n in order], tmp_path)
    fr.check_expect(fx, {n: (bufs[n][0], o) for n, o in zip(order, outs)})
    # and the same numbers as the oracle's replay of the fixture, bit for bit (integer-valued bf16 cases) or within the printed digits
    ref = fr.make_buffers(fx)
    fr.run_calls(fx, fr.OracleBackend(), ref)
    for n, o in zip(order, outs):
        if bufs[n][0] == fr.BF16:
            assert np.array_equal(o, ref[n][1]), (entry, n)
        else:
            assert np.allclose(o, ref[n][1], rtol=1e-5, atol=1e-6), (entry, n)


@pytest.mark.gpu
def test_scalar_float_argument_and_perf_timers(driver, tmp_path):
    """xsmm_unary_scalar_invoke takes its scalar as a C `float` in a vector register (XsmmRunnerUtils.h:66-68); the timers return an
    i64 and take it back (PerfRunnerUtils.h:22-24)"""
    _, exe, _ = driver
    outs, text = _run(exe, "fill_scalar_f32_timed", [np.zeros(32, np.float32)], tmp_path)
    assert np.array_equal(outs[0], np.full(32, 7.5, np.float32))
    assert text.startswith("seconds ") and 0.0 <= float(text.split()[1]) < 60.0

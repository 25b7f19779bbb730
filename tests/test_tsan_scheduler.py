"""ThreadSanitizer run of the C-ABI layer's host-side concurrency (no GPU needed): runtime.cpp - tile queue,
per-caller rings merged by time stamp, trace cache, scheduler thread and its life cycle, host-operand mirroring - is compiled UNCHANGED with
g++ -fsanitize=thread against tests/tsan/fake_hip.cpp (the HIP host API and the kernel launchers over host memory,
executed on the launching thread) and driven by tests/tsan/driver.cpp the way the reference's compiled code calls
it: 8 OpenMP-style workers invoking zero / brgemm / relu tiles of a 3-layer MLP with a barrier per layer
(pass-convert-mlp-to-parallel-tile.mlir:80-88), in sync / async / queued modes, with device and host operands
(adjacent tiles of one host matrix on different threads), a chain of non-commuting in-place ops handed from thread to
thread through an atomic only, across an idle period that retires the scheduler thread.
Pass = results identical to a serial run AND no ThreadSanitizer report."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_INCLUDE = "/opt/rocm/include"


@pytest.mark.timeout(600)
def test_runtime_layer_is_race_free_under_tsan(tmp_path):
    gxx = shutil.which("g++")
    if not gxx or not os.path.exists(os.path.join(HIP_INCLUDE, "hip", "hip_runtime.h")):
        pytest.skip("needs g++ and the HIP headers")
    exe = str(tmp_path / "tsan_driver")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-D__HIP_PLATFORM_AMD__", "-I" + HIP_INCLUDE,
           os.path.join(ROOT, "tpp-mlir_amd", "csrc", "runtime.cpp"), os.path.join(ROOT, "tpp-mlir_amd", "csrc", "host_cache.cpp"),
           os.path.join(ROOT, "tests", "tsan", "fake_hip.cpp"),
           os.path.join(ROOT, "tests", "tsan", "driver.cpp"), "-o", exe, "-pthread", "-ldl"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "tsan" in b.stderr.lower():
        pytest.skip("this g++ has no ThreadSanitizer runtime")
    assert b.returncode == 0, b.stderr[-3000:]
    # the scheduler's two portability switches: time stamps from the TSC or from a shared counter; the parking handshake
    # with membarrier() on the scheduler's side or with sequentially consistent stores on the producers' side
    for extra in ({}, {"TPP_HIP_NO_TSC": "1", "TPP_HIP_NO_MEMBARRIER": "1"}):
        env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66", **extra)
        for k in ("TPP_HIP_ASYNC", "TPP_HIP_TILE_QUEUE", "TPP_HIP_TRACE", "TPP_HIP_VARIANT"):
            env.pop(k, None)
        r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=500)
        out = r.stdout + r.stderr
        assert "ThreadSanitizer" not in out, out[-6000:]
        assert r.returncode == 0 and out.strip().endswith("OK"), out[-3000:]
        assert "identical to the serial run" in out and "MISMATCH" not in out
        assert "dependent chain" in out
        assert "after" in out and "UNEXPECTED" not in out  # the scheduler thread left when idle and came back


@pytest.mark.timeout(600)
def test_runtime_layer_under_asan_and_ubsan(tmp_path):
    """the same unchanged runtime.cpp + driver under AddressSanitizer + UndefinedBehaviorSanitizer (the reference's sanitizer bar:
    cmake/modules/sanitizers.cmake:21-147). Leak detection is off on purpose: dispatch handles are interned for the life of the process
    (the reference ABI has no destroy call), pinned work-list slots likewise."""
    gxx = shutil.which("g++")
    if not gxx or not os.path.exists(os.path.join(HIP_INCLUDE, "hip", "hip_runtime.h")):
        pytest.skip("needs g++ and the HIP headers")
    exe = str(tmp_path / "asan_driver")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-D__HIP_PLATFORM_AMD__", "-I" + HIP_INCLUDE,
           os.path.join(ROOT, "tpp-mlir_amd", "csrc", "runtime.cpp"), os.path.join(ROOT, "tpp-mlir_amd", "csrc", "host_cache.cpp"),
           os.path.join(ROOT, "tests", "tsan", "fake_hip.cpp"),
           os.path.join(ROOT, "tests", "tsan", "driver.cpp"), "-o", exe, "-pthread", "-ldl"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and ("asan" in b.stderr.lower() or "ubsan" in b.stderr.lower()):
        pytest.skip("this g++ has no ASan / UBSan runtime")
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=67", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    for k in ("TPP_HIP_ASYNC", "TPP_HIP_TILE_QUEUE", "TPP_HIP_TRACE", "TPP_HIP_VARIANT", "LD_PRELOAD"):
        env.pop(k, None)
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=500)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-6000:]
    assert r.returncode == 0 and out.strip().endswith("OK"), out[-3000:]
    assert "identical to the serial run" in out and "MISMATCH" not in out

"""The host cache on the GPU (csrc/host_cache.cpp; include/tpp_xsmm_abi.h xsmm_hip_set_host_cache): host pointers, the way an unmodified
tpp-run calls the reference (lib/TPP/Runner/MLIRBench.cpp:207-246), with the operands kept on the device between invokes.
Parity = the plain per-invoke mirror path on the same inputs, bit for bit, plus the oracle / golden fixtures where they exist."""
import importlib
import os
import subprocess

import numpy as np
import pytest

import fixture_runner as fr
from abi_backend import AbiBackend
from oracle import pyoracle as orc

pkg = importlib.import_module("tpp-mlir_amd")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32, BF16 = 1, 2


@pytest.fixture
def rt_cache():
    r = pkg.get_runtime()
    assert r.device_count() >= 1, "no HIP device visible: the gpu tests need an MI355X"
    prev = r.set_host_cache(True)
    if prev < 0:
        pytest.skip("this kernel lacks userfaultfd WP_ASYNC / PAGEMAP_SCAN: the host cache stays off")
    yield r
    r.set_async(False)
    r.set_tile_queue(0)
    r.set_host_cache(False)


@pytest.mark.parametrize("path", fr.fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_golden_fixture_through_host_pointers_with_the_cache(rt_cache, path):
    """the reference's own FileCheck'd numbers (tests/golden/*.json), host pointers, synchronous invokes - with mirrors that outlive the
    invokes (every fixture runs twice: the second pass finds its operands' pages on the device)"""
    fr.run_fixture(path, AbiBackend("host"))
    fr.run_fixture(path, AbiBackend("host"))


_keep = []


def unaligned(n, dtype, off=192):
    """memref.alloc-style: 64-byte aligned, never page aligned - in a FRESH anonymous mapping: pages the ROCm runtime has never been handed.
    (A buffer that was once the source / destination of a plain hipMemcpy stays in the runtime's pin cache, and the driver write-faults its
    pages again after every later piece of driver activity: the kernel's write tracking then reports the whole buffer written - the host
    cache uploads it again, correct but no faster than the plain path. tools/ubench/wp_vs_hipmemcpy.cpp, profiles/r06_wp_vs_hipmemcpy.txt.
    numpy's allocator re-uses the heap addresses of earlier tests' arrays, which went through exactly such copies.)"""
    import mmap
    nbytes = n * np.dtype(dtype).itemsize
    m = mmap.mmap(-1, nbytes + 8192)
    _keep.append(m)
    return np.frombuffer(m, dtype=np.uint8, count=nbytes, offset=off + 64)[:nbytes].view(dtype)


def test_c2_sync_loop_with_host_edits(rt_cache):
    """BASELINE config 2 (BRGEMM 1024^3 f32, br = 16) through host pointers in the reference's synchronous mode: results visible on
    return, identical to the plain mirror path; the host edits A, B (through a system call) and C between invokes"""
    rt = rt_cache
    m = n = 1024
    k, br = 64, 16
    rng = np.random.default_rng(5)
    A, B, C = unaligned(m * 1024, np.float32), unaligned(1024 * n, np.float32), unaligned(m * n, np.float32)
    A[:] = rng.uniform(-1, 1, A.size)
    B[:] = rng.uniform(-1, 1, B.size)
    C[:] = rng.uniform(-1, 1, C.size)
    h = rt.brgemm_dispatch(F32, m, n, k, 1024, 1024, 1024, 64, 65536, 0)  # accumulating: C is read and written
    hb = rt.brgemm_dispatch(F32, m, n, k, 1024, 1024, 1024, 64, 65536, 4)
    A2, B2, C2 = A.copy(), B.copy(), C.copy()

    def program(rt, A, B, C):
        outs = []
        rt.brgemm(F32, h, A, 0, B, 0, C, 0, br)
        outs.append(C.copy())
        rt.brgemm(F32, h, A, 0, B, 0, C, 0, br)
        A[12345] = 3.0
        rt.brgemm(F32, hb, A, 0, B, 0, C, 0, br)
        outs.append(C.copy())
        B[5000:7000] = 0.0
        C[100:2100] = -1.0
        rt.brgemm(F32, h, A, 0, B, 0, C, 0, br)
        outs.append(C.copy())
        return outs

    s0 = rt.host_cache_stats()
    got = program(rt, A, B, C)
    s1 = rt.host_cache_stats()
    rt.set_host_cache(False)
    want = program(rt, A2, B2, C2)
    rt.set_host_cache(True)
    for g, w in zip(got, want):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))
    # 12 MiB of operands, four invokes: the plain path uploads 8-12 MiB per invoke (44 MiB); the cache uploads everything once, the pages
    # edited, and C a second time (the BETA_0 invoke leaves its pure output untracked: the accumulating invoke behind it reads it again)
    up = s1["uploaded_bytes"] - s0["uploaded_bytes"]
    assert 12 * 2 ** 20 <= up <= 16 * 2 ** 20 + 64 * 4096, up


def test_c2_async_loop_runs_at_device_speed_and_writes_back_at_the_sync_point(rt_cache):
    """asynchronous mode (TPP_HIP_ASYNC=1): the timing loop of tpp-run on HOST buffers - no upload after the first invoke, the output
    on the host after perf_stop_timer, equal to the device-pointer result"""
    import torch
    rt = rt_cache
    m = n = 1024
    k, br = 64, 16
    rng = np.random.default_rng(6)
    A, B, C = unaligned(m * 1024, np.float32), unaligned(1024 * n, np.float32), unaligned(m * n, np.float32)
    A[:] = rng.uniform(-1, 1, A.size)
    B[:] = rng.uniform(-1, 1, B.size)
    C[:] = 0
    h = rt.brgemm_dispatch(F32, m, n, k, 1024, 1024, 1024, 64, 65536, 4)
    rt.set_async(True)
    rt.brgemm(F32, h, A, 0, B, 0, C, 0, br)
    rt.synchronize()
    s0 = rt.host_cache_stats()
    t0 = rt.perf_start_timer()
    for _ in range(200):
        rt.brgemm(F32, h, A, 0, B, 0, C, 0, br)
    dt = rt.perf_stop_timer(t0)
    s1 = rt.host_cache_stats()
    # (the edge pages of the run written back at the sync point)
    assert s1["uploaded_bytes"] - s0["uploaded_bytes"] <= 16 * 4096, s1
    assert s1["fast_invokes"] - s0["fast_invokes"] >= 199
    dA, dB = torch.from_numpy(A.copy()).cuda(), torch.from_numpy(B.copy()).cuda()
    dC = torch.zeros(m * n, device="cuda")
    rt.brgemm(F32, h, dA, 0, dB, 0, dC, 0, br)
    rt.synchronize()
    assert np.array_equal(C.view(np.uint32), dC.cpu().numpy().view(np.uint32))
    rt.set_async(False)
    # 200 invokes of a 17-18 us kernel: the plain mirror path needs ~300 us each (PCIe); allow generous head room for a shared box
    assert dt / 200 < 60e-6, dt / 200


@pytest.mark.parametrize("threads", [1, 4])
def test_reference_mlp_as_tile_invokes_on_host_buffers(rt_cache, threads):
    """the reference's headline MLP as the compiler emits it (768 invokes of one 32x32x32 dispatch per iteration) on plain host buffers,
    TPP_HIP_ASYNC=1 TPP_HIP_TILE_QUEUE=1 TPP_HIP_HOST_CACHE=1 - environment variables only, no xsmm_hip_* call in the program
    (tools/tpp_replay --host-buffers): the host's output buffer holds the closed-form result behind the timing loop"""
    exe = os.path.join(ROOT, "tools", "tpp_replay")
    if not os.path.exists(exe):
        pytest.skip("tools/tpp_replay not built")
    env = dict(os.environ, TPP_HIP_ASYNC="1", TPP_HIP_TILE_QUEUE="1", TPP_HIP_HOST_CACHE="1")
    r = subprocess.run([exe, "--host-buffers", "--batch", "256", "--layers", "1024,1024,1024,1024", "--tiles", "32", "--bias", "--relu", "-n", "100",
                        "--threads", str(threads)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "host output buffer checked" in r.stderr, r.stderr[-3000:]


def test_mlp_tiles_host_cache_equals_device_pointers_bitwise(rt_cache):
    """random weights: 3 layers of 32x32x32 tile invokes (packed blocks) on host buffers through the tile queue == the same program on
    device pointers, bit for bit; a weight edited between two synchronisation epochs is seen"""
    import torch
    rt = rt_cache
    M, W, T = 128, 256, 32
    MB, NB, KB = M // T, W // T, W // T
    rng = np.random.default_rng(9)
    acts = [unaligned(M * W, np.float32) for _ in range(4)]
    Ws = [unaligned(W * W, np.float32) for _ in range(3)]
    bs = [unaligned(W, np.float32) for _ in range(3)]
    acts[0][:] = rng.uniform(-1, 1, M * W)
    for w in Ws:
        w[:] = rng.uniform(-1, 1, W * W) / 16
    for b in bs:
        b[:] = rng.uniform(-1, 1, W)
    h = rt.fused_brgemm_dispatch(F32, T, T, T, T, T, T, T * T, T * T, 4, 0, 5, 4, 1)

    def iteration(acts, Ws, bs):
        for l in range(3):
            for i in range(MB):
                for j in range(NB):
                    rt.fused_brgemm(F32, h, acts[l], i * KB * T * T, Ws[l], j * KB * T * T, acts[l + 1], (i * NB + j) * T * T, bs[l], j * T, KB)

    rt.set_async(True)
    rt.set_tile_queue(1)
    for edit in (False, True):
        if edit:
            Ws[1][4321] = 0.5
            acts[0][7] = -0.25
        for _ in range(3):
            iteration(acts, Ws, bs)
        rt.synchronize()
        dacts = [torch.from_numpy(a.copy()).cuda() for a in acts]
        dWs = [torch.from_numpy(w.copy()).cuda() for w in Ws]
        dbs = [torch.from_numpy(b.copy()).cuda() for b in bs]
        iteration(dacts, dWs, dbs)
        rt.synchronize()
        for l in range(1, 4):
            assert np.array_equal(acts[l].view(np.uint32), dacts[l].cpu().numpy().view(np.uint32)), (edit, l)
    st = rt.host_cache_stats()
    assert st["fast_invokes"] > 3 * 3 * MB * NB and st["pages_not_written_back"] == 0, st
    rt.set_tile_queue(0)
    rt.set_async(False)


def test_bf16_vnni_layer_on_host_buffers_matches_the_oracle(rt_cache):
    """bf16 + VNNI-2 fused layer (bias + relu) through host pointers with the cache, against the oracle within one bf16 ulp"""
    rt = rt_cache
    m, n, k, br = 256, 512, 64, 8
    rng = np.random.default_rng(11)
    A = orc.f32_to_bf16(rng.uniform(-1, 1, m * k * br).astype(np.float32))
    Bf = rng.uniform(-1, 1, (k * br, n)).astype(np.float32) / 8
    Bv = orc.f32_to_bf16(np.ascontiguousarray(Bf.reshape(k * br // 2, 2, n).transpose(0, 2, 1)).reshape(-1))
    bias = orc.f32_to_bf16(rng.uniform(-1, 1, n).astype(np.float32))
    C = np.zeros(m * n, np.uint16)
    ref = C.copy()
    args = (BF16, m, n, k, k * br, n, n, k, k * n, 4 | 2048, 0, 5, 4, 1)
    orc.fused_brgemm(*args, A, 0, Bv, 0, ref, 0, bias, 0, br)
    h = rt.fused_brgemm_dispatch(*args)
    for _ in range(2):
        rt.fused_brgemm(BF16, h, A, 0, Bv, 0, C, 0, bias, 0, br)
    d = np.abs(orc.bf16_to_f32(C).astype(np.float64) - orc.bf16_to_f32(ref))
    r = np.abs(orc.bf16_to_f32(ref).astype(np.float64))
    assert (d <= r * 2.0 ** -7 + 1e-5 * max(1.0, r.max())).all()

"""The tile queue: the compiler's native call pattern (hundreds of invokes of one 32x32x32
dispatch per layer, test/Passes/pass-convert-mlp-to-parallel-tile.mlir:80-88) collected into
grouped launches. Results must equal the oracle's replay of the same call sequence, in
program order, whatever the queue decides to batch."""
import importlib
import threading

import numpy as np
import pytest

from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("tpp-mlir_amd")
F32, BF16 = 1, 2


@pytest.fixture()
def rtq():
    rt = pkg.get_runtime()
    assert rt.device_count() >= 1
    prev_async = rt.set_async(True)
    prev_q = rt.set_tile_queue(True)
    yield rt
    rt.synchronize()
    rt.set_tile_queue(prev_q)
    rt.set_async(prev_async)


def dev(a):
    import torch
    return torch.from_numpy((a.view(np.int16) if a.dtype == np.uint16 else a).copy()).cuda()


def host(t, like):
    a = t.cpu().numpy()
    return a.view(np.uint16) if like.dtype == np.uint16 else a


def close(got, ref, dt):
    g = (got if dt == F32 else orc.bf16_to_f32(got)).astype(np.float64)
    r = (ref if dt == F32 else orc.bf16_to_f32(ref)).astype(np.float64)
    tol = 1e-5 * max(1.0, np.abs(r).max()) + (np.abs(r) * 2.0 ** -7 if dt == BF16 else 0)
    assert (np.abs(g - r) <= tol).all(), float(np.abs(g - r).max())


@pytest.mark.parametrize("dt", [F32, BF16])
def test_packed_mlp_layers_tile_invokes(rtq, dt):
    """3 chained layers, packed layouts of mlir-gen (MLIRGen.cpp:650-677): A [MB][KB][32][32],
    W [NB][KB][32 k][32 n], C [MB][NB][32][32]; per layer ONE fused dispatch
    (32,32,32,32,32,32,1024,1024 [add(bcast_col_in0), relu]) and MB*NB invokes with batch KB.
    Layer l+1 reads what layer l wrote: the queue must flush on that dependence."""
    rt = rtq
    MB, NB, KB = 4, 8, 8  # batch 128, width 256
    rng = np.random.default_rng(dt)

    def rnd(n, s=0.3):
        v = rng.uniform(-s, s, n).astype(np.float32)
        return v if dt == F32 else orc.f32_to_bf16(v)
    vnni = dt == BF16
    X = rnd(MB * KB * 1024, 1.0)
    Ws = [rnd(NB * KB * 1024) for _ in range(3)]
    bs = [rnd(NB * 32) for _ in range(3)]
    acts = [np.zeros(MB * NB * 1024, dtype=X.dtype) for _ in range(3)]
    flags = 4 | (2048 if vnni else 0)
    disp = (dt, 32, 32, 32, 32, 32, 32, 1024, 1024, flags, 0, 5, 4, 1)
    # oracle replay
    ref_in = X
    refs = []
    for l in range(3):
        out = np.zeros(MB * NB * 1024, dtype=X.dtype)
        for i in range(MB):
            for j in range(NB):
                orc.fused_brgemm(*disp, ref_in, i * KB * 1024, Ws[l], j * KB * 1024, out, (i * NB + j) * 1024,
                                 bs[l], j * 32, KB)
        refs.append(out)
        ref_in = out
    h = rt.fused_brgemm_dispatch(*disp)
    dX, dW, db, dA = dev(X), [dev(w) for w in Ws], [dev(b) for b in bs], [dev(a) for a in acts]
    cur = dX
    for l in range(3):
        for i in range(MB):
            for j in range(NB):
                rt.fused_brgemm(dt, h, cur, i * KB * 1024, dW[l], j * KB * 1024, dA[l], (i * NB + j) * 1024,
                                db[l], j * 32, KB)
        cur = dA[l]
    rt.synchronize()
    for l in range(3):
        close(host(dA[l], X), refs[l], dt)


def test_c1_call_script_with_queue(rtq):
    """BASELINE config 1 call script on the GPU: relayout by xsmm.unary identity (not queueable:
    flushes), 64 queued brgemm invokes (beta = 1 on the packed C), un-pack."""
    rt = rtq
    rng = np.random.default_rng(1)
    A, W, C = (rng.uniform(-1, 1, 65536).astype(np.float32) for _ in range(3))
    ref = (A.reshape(256, 256).astype(np.float64) @ W.reshape(256, 256) + C.reshape(256, 256)).reshape(-1)
    dA, dW, dC = dev(A), dev(W), dev(C)
    dAp, dWp, dCp = (dev(np.zeros(65536, np.float32)) for _ in range(3))
    pack = rt.unary_dispatch(1, F32, 32, 32, 256, 32, 0)
    unpack = rt.unary_dispatch(1, F32, 32, 32, 32, 256, 0)
    hb = rt.brgemm_dispatch(F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 0)
    for bi in range(8):
        for bj in range(8):
            blk = (bi * 8 + bj) * 1024
            rt.unary(F32, pack, dA, bi * 8192 + bj * 32, dAp, blk)
            rt.unary(F32, pack, dW, bj * 8192 + bi * 32, dWp, blk)
            rt.unary(F32, pack, dC, bi * 8192 + bj * 32, dCp, blk)
    for bi in range(8):
        for bj in range(8):
            rt.brgemm(F32, hb, dAp, bi * 8192, dWp, bj * 8192, dCp, (bi * 8 + bj) * 1024, 8)
    for bi in range(8):
        for bj in range(8):
            rt.unary(F32, unpack, dCp, (bi * 8 + bj) * 1024, dC, bi * 8192 + bj * 32)
    rt.synchronize()
    got = host(dC, C).astype(np.float64)
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_accumulating_twice_into_one_tile_keeps_program_order(rtq):
    rt = rtq
    rng = np.random.default_rng(2)
    A, B = rng.uniform(-1, 1, 4096).astype(np.float32), rng.uniform(-1, 1, 4096).astype(np.float32)
    C = rng.uniform(-1, 1, 1024).astype(np.float32)
    ref = C.copy()
    for _ in range(3):  # C += A B three times: each invoke depends on the previous one
        orc.brgemm(F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 0, A, 0, B, 0, ref, 0, 4)
    h = rt.brgemm_dispatch(F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 0)
    dA, dB, dC = dev(A), dev(B), dev(C)
    for _ in range(3):
        rt.brgemm(F32, h, dA, 0, dB, 0, dC, 0, 4)
    rt.synchronize()
    close(host(dC, C), ref, F32)


def test_ragged_tiles_and_mixed_handles(rtq):
    rt = rtq
    rng = np.random.default_rng(3)
    cases = [(13, 40, 17, 3), (33, 64, 64, 2), (64, 64, 32, 4), (5, 5, 70, 1)]
    outs = []
    for (m, n, k, br) in cases * 2:
        A = rng.uniform(-1, 1, br * m * k + 8).astype(np.float32)
        B = rng.uniform(-1, 1, br * k * n + 8).astype(np.float32)
        C = rng.uniform(-1, 1, m * n + 8).astype(np.float32)
        ref = C.copy()
        orc.brgemm(F32, m, n, k, k, n, n, m * k, k * n, 0, A, 0, B, 0, ref, 0, br)
        h = rt.brgemm_dispatch(F32, m, n, k, k, n, n, m * k, k * n, 0)
        dA, dB, dC = dev(A), dev(B), dev(C)  # operands must stay alive until the queue is flushed
        rt.brgemm(F32, h, dA, 0, dB, 0, dC, 0, br)
        outs.append((dC, C, ref, dA, dB))
    rt.synchronize()
    for dC, C, ref, _, _ in outs:
        close(host(dC, C), ref, F32)


def test_concurrent_enqueue_from_threads(rtq):
    rt = rtq
    rng = np.random.default_rng(4)
    A, B = rng.uniform(-1, 1, 16 * 8192).astype(np.float32), rng.uniform(-1, 1, 16 * 8192).astype(np.float32)
    C = np.zeros(16 * 16 * 1024, np.float32)
    ref = C.copy()
    h = rt.brgemm_dispatch(F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 4)
    tiles = [(i, j) for i in range(16) for j in range(16)]
    for (i, j) in tiles:
        orc.brgemm(F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 4, A, i * 8192, B, j * 8192, ref, (i * 16 + j) * 1024, 8)
    dA, dB, dC = dev(A), dev(B), dev(C)

    def worker(chunk):
        for (i, j) in chunk:
            rt.brgemm(F32, h, dA, i * 8192, dB, j * 8192, dC, (i * 16 + j) * 1024, 8)
    ths = [threading.Thread(target=worker, args=(tiles[w::4],)) for w in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    rt.synchronize()
    close(host(dC, C), ref, F32)

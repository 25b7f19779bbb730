"""The xsmm call scripts of the reference's hand-written benchmark files (benchmarks/config/base/mha.json and pack.json ->
benchmarks/mlir/fp32-projection.mlir, fp32-query-times-key.mlir, fp32-out-softmax-times-value.mlir, fp32-pack-gemm-operand-{a,b}-512x1024.mlir,
fp32-unpack-gemm-operand-a-512x512.mlir) on the HIP path, against the oracle.

The calls per (batch, head) tile are the ones test/Conversion/LinalgToXsmm/linalg-to-gemm.mlir pins for exactly these functions
(mha_projection :132-145: gemm [32,64,512,512,512,512]; mha_query_times_key :46-62: unary transpose [32,64,512,32] into a 64x32
temporary + gemm [32,32,64,512,32,32]; mha_out_softmax_times_value :91-104: gemm [32,64,32,32,512,512]), in both forms the compiler
produces: unary zero (bcast_scalar) + gemm with beta = 1 as that test has them, and with the fill folded into BETA_0
(test/Passes/fold-xsmm-flags.mlir:3-20; LinalgLowering.cpp:56). Queue on and off; the results of the two forms are bit-identical."""
import importlib

import numpy as np
import pytest

from oracle import pyoracle as orc
from test_parity_gpu import F32, check_close, dev, host

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("tpp-mlir_amd")
Bt, S, H, D = 64, 32, 8, 64
E = H * D
ZERO, TRANSPOSE, IDENTITY, BCAST_SCALAR, BETA0 = 2, 29, 1, 8, 4


@pytest.fixture(scope="module")
def rt():
    r = pkg.get_runtime()
    assert r.device_count() >= 1, "no HIP device visible: the gpu tests need an MI355X"
    r.set_fold_transpose(True)
    return r


def run_script(rt, queue, body):
    old_async, old_q = rt.set_async(True), rt.set_tile_queue(queue)
    try:
        body()
        rt.synchronize()
    finally:
        rt.set_tile_queue(old_q)
        rt.set_async(old_async)


def tiles(nb=Bt):
    return [(b, h) for b in range(nb) for h in range(H)]


@pytest.mark.parametrize("queue", [0, 1], ids=["noqueue", "queue"])
@pytest.mark.parametrize("folded", [True, False], ids=["beta0", "zero+gemm"])
def test_mha_projection(rt, queue, folded):
    rng = np.random.default_rng(11)
    X = rng.uniform(-1, 1, Bt * S * E).astype(np.float32)
    W = (rng.uniform(-1, 1, E * E) / np.sqrt(E)).astype(np.float32)
    out0 = rng.uniform(-1, 1, Bt * S * E).astype(np.float32)  # overwritten: the zero fill (or BETA_0) comes first
    ref = out0.copy()
    for b, h in tiles():
        orc.gemm(F32, S, D, E, E, E, E, BETA0, X, b * S * E, W, h * D, ref, b * S * E + h * D)
    dX, dW, dO = dev(X), dev(W), dev(out0)
    hz = rt.unary_dispatch(ZERO, F32, S, D, 1, E, BCAST_SCALAR)
    hg = rt.gemm_dispatch(F32, S, D, E, E, E, E, BETA0 if folded else 0)

    def body():
        for b, h in tiles():
            if not folded:
                rt.unary_scalar(F32, hz, 0.0, dO, b * S * E + h * D)
            rt.gemm(F32, hg, dX, b * S * E, dW, h * D, dO, b * S * E + h * D)

    run_script(rt, queue, body)
    mag = (np.abs(X).reshape(Bt * S, E).astype(np.float64) @ np.abs(W).reshape(E, E).astype(np.float64)).reshape(-1)
    check_close(host(dO, ref), ref, F32, "mha projection, queue %d [%s]" % (queue, rt.last_grouped_kernel() if queue else rt.kernel_name(hg)),
                mag=mag, K=E)


@pytest.mark.parametrize("queue", [0, 1], ids=["noqueue", "queue"])
@pytest.mark.parametrize("tmps", [1, 4], ids=["one_tmp", "four_tmps"])
def test_mha_query_times_key(rt, queue, tmps):
    """transpose -> temporary -> gemm per tile: with ONE temporary (a single caller) every invoke depends on the one before it;
    with several (the reference's OpenMP callers own one each) consecutive tiles are independent"""
    rng = np.random.default_rng(12)
    nb = 16
    Q = rng.uniform(-1, 1, nb * S * E).astype(np.float32)
    Km = rng.uniform(-1, 1, nb * S * E).astype(np.float32)
    out0 = rng.uniform(-1, 1, nb * H * S * S).astype(np.float32)
    ref = out0.copy()
    tmp = np.zeros(D * S, np.float32)
    for b, h in tiles(nb):
        orc.unary(TRANSPOSE, F32, S, D, E, S, 0, Q, b * S * E + h * D, tmp, 0)
        orc.gemm(F32, S, S, D, E, S, S, BETA0, Km, b * S * E + h * D, tmp, 0, ref, (b * H + h) * S * S)
    dQ, dK, dO, dT = dev(Q), dev(Km), dev(out0), dev(np.zeros(tmps * D * S, np.float32))
    ht = rt.unary_dispatch(TRANSPOSE, F32, S, D, E, S, 0)
    hg = rt.gemm_dispatch(F32, S, S, D, E, S, S, BETA0)

    def body():
        for i, (b, h) in enumerate(tiles(nb)):
            t = (i % tmps) * D * S
            rt.unary(F32, ht, dQ, b * S * E + h * D, dT, t)
            rt.gemm(F32, hg, dK, b * S * E + h * D, dT, t, dO, (b * H + h) * S * S)

    run_script(rt, queue, body)
    got = host(dO, ref)
    check_close(got, ref, F32, "mha query x key, queue %d, %d temporaries" % (queue, tmps), K=D)
    # the last transposes are what the temporaries hold
    last = host(dT, tmp)
    b, h = tiles(nb)[-1]
    assert np.array_equal(last[((nb * H - 1) % tmps) * D * S:][:D * S], tmp), "temporary after the last tile"


@pytest.mark.parametrize("queue", [0, 1], ids=["noqueue", "queue"])
@pytest.mark.parametrize("folded", [True, False], ids=["beta0", "zero+gemm"])
def test_mha_softmax_times_value(rt, queue, folded):
    rng = np.random.default_rng(13)
    P = rng.uniform(0, 1, Bt * H * S * S).astype(np.float32)
    V = rng.uniform(-1, 1, Bt * S * E).astype(np.float32)
    out0 = rng.uniform(-1, 1, Bt * S * E).astype(np.float32)
    ref = out0.copy()
    for b, h in tiles():
        orc.gemm(F32, S, D, S, S, E, E, BETA0, P, (b * H + h) * S * S, V, b * S * E + h * D, ref, b * S * E + h * D)
    dP, dV, dO = dev(P), dev(V), dev(out0)
    hz = rt.unary_dispatch(ZERO, F32, S, D, 1, E, BCAST_SCALAR)
    hg = rt.gemm_dispatch(F32, S, D, S, S, E, E, BETA0 if folded else 0)

    def body():
        for b, h in tiles():
            if not folded:
                rt.unary_scalar(F32, hz, 0.0, dO, b * S * E + h * D)
            rt.gemm(F32, hg, dP, (b * H + h) * S * S, dV, b * S * E + h * D, dO, b * S * E + h * D)

    run_script(rt, queue, body)
    check_close(host(dO, ref), ref, F32, "mha softmax x value, queue %d [%s]" % (queue, rt.last_grouped_kernel() if queue else rt.kernel_name(hg)), K=S)


@pytest.mark.parametrize("queue", [0, 1], ids=["noqueue", "queue"])
@pytest.mark.parametrize("which", ["pack_a", "pack_b", "unpack_c"])
def test_pack_benchmarks_bit_exact(rt, queue, which):
    """benchmarks/mlir/fp32-pack-gemm-operand-a-512x1024.mlir ([512][1024] -> [16][32][32][32]), -b-512x1024.mlir ([1024][512] ->
    outer_dims_perm [1,0] -> [16][32][32][32]) and fp32-unpack-gemm-operand-a-512x512.mlir as per-block xsmm.unary identity copies
    (LowerPacksAndUnpacks.cpp:45-49,112-121): memcmp-exact against numpy's relayout of the same bits"""
    T = 32
    R, Cc = (1024, 512) if which == "pack_b" else (512, 1024) if which == "pack_a" else (512, 512)
    RB, CB = R // T, Cc // T
    rng = np.random.default_rng(14)
    src = rng.integers(0, 2 ** 32, R * Cc, dtype=np.uint32).view(np.float32)  # any bit pattern (NaN payloads included) must survive
    if which == "pack_a":
        ref = src.reshape(RB, T, CB, T).transpose(0, 2, 1, 3)
        h = rt.unary_dispatch(IDENTITY, F32, T, T, Cc, T, 0)
        calls = [(i * T * Cc + j * T, (i * CB + j) * T * T) for i in range(RB) for j in range(CB)]
    elif which == "pack_b":
        ref = src.reshape(RB, T, CB, T).transpose(2, 0, 1, 3)  # [n block][k block][32][32]
        h = rt.unary_dispatch(IDENTITY, F32, T, T, Cc, T, 0)
        calls = [(kb * T * Cc + nb * T, (nb * RB + kb) * T * T) for nb in range(CB) for kb in range(RB)]
    else:
        ref = src.reshape(RB, CB, T, T).transpose(0, 2, 1, 3)
        h = rt.unary_dispatch(IDENTITY, F32, T, T, T, Cc, 0)
        calls = [((i * CB + j) * T * T, i * T * Cc + j * T) for i in range(RB) for j in range(CB)]
    ref = np.ascontiguousarray(ref).reshape(-1)
    dI, dO = dev(src), dev(np.zeros(R * Cc, np.float32))

    def body():
        for oi, oo in calls:
            rt.unary(F32, h, dI, oi, dO, oo)

    run_script(rt, queue, body)
    got = host(dO, ref)
    assert got.view(np.uint32).tobytes() == ref.view(np.uint32).tobytes(), which


# ---- transposes folded into the gemm they feed (runtime.cpp "deferred transposes"; include/tpp_xsmm_abi.h xsmm_hip_set_fold_transpose) ----

def test_query_times_key_folds_into_one_launch_per_call(rt):
    """one caller, one temporary: every gemm reads its transpose's source, every transpose but the last is dropped as dead, the last one
    is launched at the synchronisation point - and the whole script is a handful of launches instead of one per invoke"""
    rng = np.random.default_rng(21)
    nb = 16
    Q = rng.uniform(-1, 1, nb * S * E).astype(np.float32)
    Km = rng.uniform(-1, 1, nb * S * E).astype(np.float32)
    ref = np.zeros(nb * H * S * S, np.float32)
    tmp = np.zeros(D * S, np.float32)
    for b, h in tiles(nb):
        orc.unary(TRANSPOSE, F32, S, D, E, S, 0, Q, b * S * E + h * D, tmp, 0)
        orc.gemm(F32, S, S, D, E, S, S, BETA0, Km, b * S * E + h * D, tmp, 0, ref, (b * H + h) * S * S)
    dQ, dK, dO, dT = dev(Q), dev(Km), dev(np.full(nb * H * S * S, 7.0, np.float32)), dev(np.zeros(D * S, np.float32))
    ht = rt.unary_dispatch(TRANSPOSE, F32, S, D, E, S, 0)
    hg = rt.gemm_dispatch(F32, S, S, D, E, S, S, BETA0)

    def body():
        for b, h in tiles(nb):
            rt.unary(F32, ht, dQ, b * S * E + h * D, dT, 0)
            rt.gemm(F32, hg, dK, b * S * E + h * D, dT, 0, dO, (b * H + h) * S * S)

    f0, q0 = rt.fold_transpose_stats(), rt.tile_queue_stats()
    run_script(rt, 1, body)
    f1, q1 = rt.fold_transpose_stats(), rt.tile_queue_stats()
    assert (f1[0] - f0[0], f1[1] - f0[1], f1[2] - f0[2]) == (nb * H, nb * H - 1, 1), (f0, f1)
    assert q1[0] - q0[0] <= 3, "launches of the whole script: %d" % (q1[0] - q0[0])
    check_close(host(dO, ref), ref, F32, "folded query x key [%s]" % rt.last_grouped_kernel(), K=D)
    assert np.array_equal(host(dT, tmp), tmp), "the temporary holds the last transpose after the synchronisation point"
    # switched off: the same results (other kernel, same tolerance), nothing folded
    old = rt.set_fold_transpose(False)
    try:
        dO2, dT2 = dev(np.full(nb * H * S * S, 7.0, np.float32)), dev(np.zeros(D * S, np.float32))

        def body2():
            for b, h in tiles(nb):
                rt.unary(F32, ht, dQ, b * S * E + h * D, dT2, 0)
                rt.gemm(F32, hg, dK, b * S * E + h * D, dT2, 0, dO2, (b * H + h) * S * S)

        run_script(rt, 1, body2)
        assert rt.fold_transpose_stats() == f1
        check_close(host(dO2, ref), ref, F32, "query x key, folding off", K=D)
    finally:
        rt.set_fold_transpose(old)


def test_remembered_transpose_is_launched_before_anything_else_reads_or_writes_its_destination(rt):
    rng = np.random.default_rng(22)
    X = rng.uniform(-1, 1, S * E).astype(np.float32)
    Y = rng.uniform(-1, 1, S * E).astype(np.float32)
    tX = np.ascontiguousarray(X.reshape(S, E)[:, :D].T).reshape(-1)
    tY = np.ascontiguousarray(Y.reshape(S, E)[:, :D].T).reshape(-1)
    ht = rt.unary_dispatch(TRANSPOSE, F32, S, D, E, S, 0)
    hc = rt.unary_dispatch(IDENTITY, F32, D, S, S, S, 0)
    hr = rt.unary_dispatch(5, F32, D, S, S, S, 0)  # relu, in place on the temporary
    dX, dY = dev(X), dev(Y)
    old_async, old_q = rt.set_async(True), rt.set_tile_queue(1)
    try:
        # (1) a copy out of the temporary right behind the transpose
        dT, dC = dev(np.zeros(D * S, np.float32)), dev(np.zeros(D * S, np.float32))
        rt.unary(F32, ht, dX, 0, dT, 0)
        rt.unary(F32, hc, dT, 0, dC, 0)
        rt.synchronize()
        assert np.array_equal(host(dC, tX), tX)
        # (2) transpose, in-place relu on the temporary (read + write), second transpose into it, sync
        rt.unary(F32, ht, dX, 0, dT, 0)
        rt.unary(F32, hr, dT, 0, dT, 0)
        rt.synchronize()
        assert np.array_equal(host(dT, tX), np.maximum(tX, 0))
        rt.unary(F32, ht, dX, 0, dT, 0)
        rt.unary(F32, ht, dY, 0, dT, 0)  # replaces the remembered one
        assert np.array_equal(host(dT, tX), np.maximum(tX, 0)), "nothing launched yet (a stream-ordered copy sees the old bytes)"
        rt.flush()
        assert np.array_equal(host(dT, tY), tY)
        # (3) the timer's stop is a synchronisation point too
        rt.unary(F32, ht, dX, 0, dT, 0)
        rt.lib.perf_stop_timer(rt.lib.perf_start_timer())
        assert np.array_equal(host(dT, tX), tX)
        # (4) leaving the queue / asynchronous mode launches it
        rt.unary(F32, ht, dY, 0, dT, 0)
        rt.set_tile_queue(0)
        rt.synchronize()
        assert np.array_equal(host(dT, tY), tY)
        rt.set_tile_queue(1)
        rt.unary(F32, ht, dX, 0, dT, 0)
        rt.set_async(False)
        assert np.array_equal(host(dT, tX), tX)
        rt.set_async(True)
        # (5) a stream switch launches it on the stream it was invoked for (and drains that stream)
        import torch
        other = torch.cuda.Stream()
        rt.unary(F32, ht, dY, 0, dT, 0)
        rt.set_stream(other)
        try:
            assert np.array_equal(host(dT, tY), tY)
            rt.unary(F32, ht, dX, 0, dT, 0)  # remembered for the new stream
            rt.gemm(F32, rt.gemm_dispatch(F32, S, S, D, E, S, S, BETA0), dY, 0, dT, 0, dC, 0)  # folded, queued on the new stream
        finally:
            rt.set_stream(None)
        rt.synchronize()
        torch.cuda.synchronize()
        assert np.array_equal(host(dT, tX), tX)
        ref = np.zeros(S * S, np.float32)
        orc.gemm(F32, S, S, D, E, S, S, BETA0, Y, 0, tX, 0, ref, 0)
        check_close(host(dC, ref)[:S * S], ref, F32, "gemm folded on a second stream", K=D)
    finally:
        rt.set_tile_queue(old_q)
        rt.set_async(old_async)


@pytest.mark.parametrize("case", ["c_over_source", "a_is_temporary", "beta1", "wrong_shape", "strided_destination", "batch2"])
def test_gemm_behind_a_transpose_folded_or_not_same_results(rt, case):
    """the gemm behind a remembered transpose: folded when it may be, else the transpose is launched first - either way the results
    are the oracle's on the program as written"""
    rng = np.random.default_rng(23)
    n = 4 * S * E
    X = rng.uniform(-1, 1, n).astype(np.float32)
    A = rng.uniform(-1, 1, n).astype(np.float32)
    C0 = rng.uniform(-1, 1, n).astype(np.float32)
    ldo = S + 8 if case == "strided_destination" else S
    tmp0 = rng.uniform(-1, 1, 2 * D * ldo + 64).astype(np.float32)
    flags = 0 if case == "beta1" else BETA0
    k = D if case != "wrong_shape" else D // 2
    br = 2 if case == "batch2" else 1
    ht = rt.unary_dispatch(TRANSPOSE, F32, S, D, E, ldo, 0)
    hg = rt.brgemm_dispatch(F32, S, S, k, E, ldo, E, 8, 16 * ldo, flags)
    # oracle, in program order on one set of buffers (C may live in X's buffer: "c_over_source")
    bufs = {"X": X.copy(), "A": A.copy(), "C": C0.copy(), "T": tmp0.copy()}
    cbuf = "X" if case == "c_over_source" else "C"
    abuf = "T" if case == "a_is_temporary" else "A"
    lda = D if abuf == "T" else E  # (the 64x32 temporary read as a 32x64 A operand)
    if abuf == "T":
        hg = rt.brgemm_dispatch(F32, S, S, k, lda, ldo, E, 8, 16 * ldo, flags)
    for rep in range(3):
        orc.unary(TRANSPOSE, F32, S, D, E, ldo, 0, bufs["X"], rep * D, bufs["T"], 0)
        orc.brgemm(F32, S, S, k, lda, ldo, E, 8, 16 * ldo, flags, bufs[abuf], 0 if abuf == "T" else rep * 64, bufs["T"], 0, bufs[cbuf], 128 + rep * 32, br)
    d = {name: dev(v) for name, v in (("X", X), ("A", A), ("C", C0), ("T", tmp0))}
    f0 = rt.fold_transpose_stats()

    def body():
        for rep in range(3):
            rt.unary(F32, ht, d["X"], rep * D, d["T"], 0)
            rt.brgemm(F32, hg, d[abuf], 0 if abuf == "T" else rep * 64, d["T"], 0, d[cbuf], 128 + rep * 32, br)

    run_script(rt, 1, body)
    f1 = rt.fold_transpose_stats()
    folded = f1[0] - f0[0]
    assert folded == (3 if case == "beta1" else 0), (case, f0, f1)
    for name in ("X", "C", "T"):
        check_close(host(d[name], bufs[name]), bufs[name], F32, "%s: buffer %s" % (case, name), K=D)


def test_another_thread_reads_the_temporary_after_a_join(rt):
    """thread 1: transpose + folded gemm; joined; the main thread copies the temporary: it must hold the transpose (an invoke of another
    thread that touches a remembered transpose's destination launches it first). Every thread has its own record: the main thread's
    transposes fold as well."""
    import threading
    rng = np.random.default_rng(24)
    X = rng.uniform(-1, 1, S * E).astype(np.float32)
    Km = rng.uniform(-1, 1, S * E).astype(np.float32)
    tX = np.ascontiguousarray(X.reshape(S, E)[:, :D].T).reshape(-1)
    ht = rt.unary_dispatch(TRANSPOSE, F32, S, D, E, S, 0)
    hg = rt.gemm_dispatch(F32, S, S, D, E, S, S, BETA0)
    hc = rt.unary_dispatch(IDENTITY, F32, D, S, S, S, 0)
    dX, dK, dT, dC, dO = dev(X), dev(Km), dev(np.zeros(D * S, np.float32)), dev(np.zeros(D * S, np.float32)), dev(np.zeros(S * S, np.float32))
    ref = np.zeros(S * S, np.float32)
    orc.gemm(F32, S, S, D, E, S, S, BETA0, Km, 0, tX, 0, ref, 0)
    old_async, old_q = rt.set_async(True), rt.set_tile_queue(1)
    try:
        def worker():
            rt.unary(F32, ht, dX, 0, dT, 0)
            rt.gemm(F32, hg, dK, 0, dT, 0, dO, 0)

        f0 = rt.fold_transpose_stats()
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        assert rt.fold_transpose_stats()[0] - f0[0] == 1 and rt.fold_transpose_stats()[2] == f0[2], "folded, the transpose still remembered"
        rt.unary(F32, hc, dT, 0, dC, 0)  # another thread's invoke launches it first
        rt.synchronize()
        assert rt.fold_transpose_stats()[2] - f0[2] == 1
        assert np.array_equal(host(dC, tX), tX)
        check_close(host(dO, ref), ref, F32, "gemm of the worker thread", K=D)
        # an invoke of another thread that touches neither the destination nor (writing) the source leaves the record alone
        dT2, dC2 = dev(np.zeros(D * S, np.float32)), dev(np.zeros(D * S, np.float32))
        f1 = rt.fold_transpose_stats()

        def worker2():
            rt.unary(F32, ht, dX, 0, dT2, 0)

        t = threading.Thread(target=worker2)
        t.start()
        t.join()
        rt.unary(F32, hc, dC, 0, dC2, 0)  # reads dC, writes dC2: unrelated to dT2 and dX
        assert rt.fold_transpose_stats()[2] == f1[2], "still remembered"
        rt.unary(F32, hc, dC, 0, dX, 0)  # WRITES (the first 64 x 32 floats of) the transpose's source: the transpose must read the old bytes, so it is launched first
        assert rt.fold_transpose_stats()[2] - f1[2] == 1
        rt.synchronize()
        assert np.array_equal(host(dT2, tX), tX)
        # this thread's own transposes fold too (one record per thread)
        rt.unary(F32, ht, dK, 0, dT, 0)
        rt.gemm(F32, hg, dK, 0, dT, 0, dO, 0)
        rt.synchronize()
        assert rt.fold_transpose_stats()[0] - f0[0] == 2
    finally:
        rt.set_tile_queue(old_q)
        rt.set_async(old_async)


# ---- tile grids over flat operands merged into one launch (runtime.cpp "GRID MERGE") ----

@pytest.mark.parametrize("case", ["grid", "grid_bias_relu_br2", "hole", "swapped_outputs", "columns_wider_than_ldb"])
def test_flat_tile_grid_replayed_as_one_merged_launch(rt, case):
    """tile invokes that tile ONE flat problem (A by tile row, B by tile column, C by both): the first pass collects the group (grouped
    kernel), complete replays of it run as one launch of the merged problem - every pass against the oracle on the program as written.
    Groups that are not exactly a grid (a missing tile, outputs that do not follow the grid, B tiles that are not columns of one row)
    stay on the grouped kernel."""
    rng = np.random.default_rng(31)
    M, N, K, tm, tn = 256, 256, 128, 32, 64
    br = 2 if case == "grid_bias_relu_br2" else 1
    k = K // br
    fused = case == "grid_bias_relu_br2"
    X = rng.uniform(-1, 1, M * K).astype(np.float32)
    W = (rng.uniform(-1, 1, K * N) / np.sqrt(K)).astype(np.float32)
    bias = rng.uniform(-1, 1, N).astype(np.float32)
    ldb = N
    if case == "columns_wider_than_ldb":  # B tiles 64 columns apart in memory but ldb = 64: consecutive "columns" are really other rows
        ldb = tn
    if fused:
        h = rt.fused_brgemm_dispatch(F32, tm, tn, k, K, ldb, N, k, k * ldb, BETA0, 0, 5, 4, 1)
    else:
        h = rt.brgemm_dispatch(F32, tm, tn, k, K, ldb, N, k, k * ldb, BETA0)
    grid = [(i, j) for i in range(M // tm) for j in range(N // tn)]
    if case == "hole":
        grid = grid[:-1]
    out_of = {t: t for t in grid}
    if case == "swapped_outputs":
        out_of[grid[3]], out_of[grid[4]] = grid[4], grid[3]
    C0 = rng.uniform(-1, 1, M * N).astype(np.float32)
    ref = C0.copy()
    for (i, j) in grid:
        oi, oj = out_of[(i, j)]
        if fused:
            orc.fused_brgemm(F32, tm, tn, k, K, ldb, N, k, k * ldb, BETA0, 0, 5, 4, 1, X, i * tm * K, W, j * tn, ref, oi * tm * N + oj * tn, bias, j * tn, br)
        else:
            orc.brgemm(F32, tm, tn, k, K, ldb, N, k, k * ldb, BETA0, X, i * tm * K, W, j * tn, ref, oi * tm * N + oj * tn, br)
    dX, dW, dB = dev(X), dev(W), dev(bias)
    old_async, old_q = rt.set_async(True), rt.set_tile_queue(1)
    try:
        names = []
        for rep in range(4):
            dC = dev(C0)
            # (a fresh output tensor per pass would be another group: the passes write the same one, re-initialised through a copy)
            if rep == 0:
                dOut = dC
            else:
                dOut.copy_(dC)
            for (i, j) in grid:
                oi, oj = out_of[(i, j)]
                if fused:
                    rt.fused_brgemm(F32, h, dX, i * tm * K, dW, j * tn, dOut, oi * tm * N + oj * tn, dB, j * tn, br)
                else:
                    rt.brgemm(F32, h, dX, i * tm * K, dW, j * tn, dOut, oi * tm * N + oj * tn, br)
            rt.synchronize()
            names.append(rt.last_grouped_kernel())
            check_close(host(dOut, ref), ref, F32, "%s pass %d [%s]" % (case, rep, names[-1]), K=K)
    finally:
        rt.set_tile_queue(old_q)
        rt.set_async(old_async)
    merged = ["merged" in nm for nm in names]
    if case in ("grid", "grid_bias_relu_br2"):
        assert not merged[0] and all(merged[2:]), names
    else:
        assert not any(merged), names

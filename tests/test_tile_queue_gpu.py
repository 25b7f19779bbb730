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


@pytest.fixture(params=[1, 2], ids=["direct", "scheduler"])
def rtq(request):
    """the tile queue in both multi-caller modes: 1 = replayed groups are joined lock-free by every caller (direct window), the rest
    under the queue's lock; 2 = several callers hand their invokes to the scheduler thread (per-caller rings)"""
    rt = pkg.get_runtime()
    assert rt.device_count() >= 1
    prev_async = rt.set_async(True)
    prev_q = rt.set_tile_queue(request.param)
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


UNARY_Q = [(1, 0), (1, 2), (1, 4), (1, 8), (5, 0), (5, 4), (2, 0), (29, 0), (28, 0)]


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("kind,flags", UNARY_Q)
def test_queued_unary_tiles_bit_exact(rtq, dt, kind, flags):
    """pack / unpack / broadcast tiles (LowerPacksAndUnpacks.cpp:45-121 lowers tensor.pack to per-block
    unary invokes): a grid of tile invokes of ONE dispatch through the queue equals the oracle's replay"""
    if kind == 28 and dt == F32:
        pytest.skip("VNNI-2 packs 16-bit elements")
    rt = rtq
    rng = np.random.default_rng(kind * 10 + flags)
    for (m, n, ld_big) in ((32, 32, 256), (13, 40, 97), (64, 64, 64), (6, 10, 12)):
        m += (m & 1) if kind == 28 else 0  # VNNI-2 packs row pairs
        gi, gj = 5, 3
        ldi = {0: ld_big, 2: 1, 4: n, 8: 1}[flags]
        ldo = m + 3 if kind == 29 else n + 2
        out_rows = n if kind == 29 else m
        out_tile = out_rows * ldo + 8
        src = rng.uniform(-1, 1, gi * 64 * ld_big + gj * 64 + 64 * ld_big).astype(np.float32)
        X = src if dt == F32 else orc.f32_to_bf16(src)
        O = np.zeros(gi * gj * out_tile, dtype=X.dtype)
        ref = O.copy()
        h = rt.unary_dispatch(kind, dt, m, n, ldi, ldo, flags)
        dX, dO = dev(X), dev(O)
        for i in range(gi):
            for j in range(gj):
                oi, oo = i * m * ld_big + j * n, (i * gj + j) * out_tile
                orc.unary(kind, dt, m, n, ldi, ldo, flags, X, oi, ref, oo)
                rt.unary(dt, h, dX, oi, dO, oo)
        rt.synchronize()
        assert host(dO, O).tobytes() == ref.tobytes(), (kind, flags, m, n)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("kind", [1, 2, 3, 4])
@pytest.mark.parametrize("flags", [0, 4, 8, 1, 32, 16 | 2])
def test_queued_binary_tiles(rtq, dt, kind, flags):
    rt = rtq
    rng = np.random.default_rng(kind * 100 + flags)
    m, n, ld = 32, 48, 160
    f0, f1 = flags & (1 | 4 | 16), flags & (2 | 8 | 32)
    ldl = 1 if f0 in (1, 16) else (n if f0 == 4 else ld)
    ldr = 1 if f1 in (2, 32) else (n if f1 == 8 else ld)
    L = rng.uniform(0.5, 2.0, 4 * m * ld + 64).astype(np.float32)
    R = rng.uniform(0.5, 2.0, 4 * m * ld + 64).astype(np.float32)
    if dt == BF16:
        L, R = orc.f32_to_bf16(L), orc.f32_to_bf16(R)
    O = np.zeros(4 * 3 * m * n, dtype=L.dtype)
    ref = O.copy()
    h = rt.binary_dispatch(kind, dt, m, n, ldl, ldr, n, flags)
    dL, dR, dO = dev(L), dev(R), dev(O)
    for i in range(4):
        for j in range(3):
            ol, orr, oo = i * m * ld + j * n, i * m * ld + j * n + 5, (i * 3 + j) * m * n
            orc.binary(kind, dt, m, n, ldl, ldr, n, flags, L, ol, R, orr, ref, oo)
            rt.binary(dt, h, dL, ol, dR, orr, dO, oo)
    rt.synchronize()
    assert host(dO, O).tobytes() == ref.tobytes()


def test_queued_in_place_relu_then_dependent_reads(rtq):
    """in-place tiles (relu on its own input) followed by a consumer of those tiles: the consumer must
    see the relu'd values (RAW flush), and a later overwrite of the source must not race (WAR flush)"""
    rt = rtq
    rng = np.random.default_rng(11)
    X = rng.uniform(-1, 1, 8 * 1024).astype(np.float32)
    relu = rt.unary_dispatch(5, F32, 32, 32, 32, 32, 0)
    copy = rt.unary_dispatch(1, F32, 32, 32, 32, 32, 0)
    zero = rt.unary_dispatch(2, F32, 32, 32, 32, 32, 0)
    dX, dY = dev(X), dev(np.zeros_like(X))
    for b in range(8):
        rt.unary(F32, relu, dX, b * 1024, dX, b * 1024)
    for b in range(8):
        rt.unary(F32, copy, dX, b * 1024, dY, b * 1024)
    for b in range(8):
        rt.unary(F32, zero, dX, 0, dX, b * 1024)
    rt.synchronize()
    assert np.array_equal(host(dY, X), np.maximum(X, 0))
    assert not host(dX, X).any()


def test_interleaved_and_overlapping_tiles_of_one_buffer(rtq):
    """tiles of one row-major buffer: neighbours have interleaved rows (no dependence - they must be
    batched, not flushed one by one) while partially overlapping tiles must keep program order"""
    rt = rtq
    rng = np.random.default_rng(21)
    ld = 256
    src = rng.uniform(-1, 1, 40 * 1024).astype(np.float32)
    dst = np.zeros(96 * ld, np.float32)
    ref = dst.copy()
    h = rt.unary_dispatch(1, F32, 32, 32, 32, ld, 0)
    spots = [(0, 0), (0, 32), (0, 64), (32, 0), (32, 32), (16, 16), (16, 48), (8, 8), (40, 40), (0, 224),
             (63, 100), (64, 0), (64, 32), (50, 90), (20, 200), (21, 201), (22, 202), (0, 0), (33, 33)]
    dS, dD = dev(src), dev(dst)
    for rep in range(2):
        for t, (r, c) in enumerate(spots):
            orc.unary(1, F32, 32, 32, 32, ld, 0, src, ((t + rep) % 40) * 1024, ref, r * ld + c)
            rt.unary(F32, h, dS, ((t + rep) % 40) * 1024, dD, r * ld + c)
    rt.synchronize()
    assert np.array_equal(host(dD, dst), ref)


@pytest.mark.parametrize("tiles", [(64, 64, 64), (32, 64, 64), (64, 32, 64), (64, 64, 128)], ids=lambda t: "x".join(map(str, t)))
@pytest.mark.parametrize("nblk", [(2, 3), (8, 16)], ids=["few", "many"])
def test_packed_layers_k64_tiles_use_the_fast_families(rtq, tiles, nblk):
    """mlir-gen --tiles=64,64,64 (the most common setting of the reference's benchmark configs): f32 tiles with
    k a multiple of 64 run on the fast tile families in grouped mode; few / many items pick different families"""
    rt = rtq
    tm, tn, tk = tiles
    MB, NB = nblk
    KB = 4
    rng = np.random.default_rng(tm + tn + tk + MB)
    X = rng.uniform(-1, 1, MB * KB * tm * tk).astype(np.float32)
    Wt = rng.uniform(-0.3, 0.3, NB * KB * tk * tn).astype(np.float32)
    b = rng.uniform(-0.3, 0.3, NB * tn).astype(np.float32)
    C0 = rng.uniform(-1, 1, MB * NB * tm * tn).astype(np.float32)
    for (gflags, ukind, bflags, bkind) in ((4, 5, 4, 1), (0, 0, 0, 0)):
        disp = (F32, tm, tn, tk, tk, tn, tn, tm * tk, tk * tn, gflags, 0, ukind, bflags, bkind)
        ref = C0.copy()
        for i in range(MB):
            for j in range(NB):
                orc.fused_brgemm(*disp, X, i * KB * tm * tk, Wt, j * KB * tk * tn, ref, (i * NB + j) * tm * tn, b, j * tn, KB)
        h = rt.fused_brgemm_dispatch(*disp)
        dX, dW, db, dC = dev(X), dev(Wt), dev(b), dev(C0)
        for i in range(MB):
            for j in range(NB):
                rt.fused_brgemm(F32, h, dX, i * KB * tm * tk, dW, j * KB * tk * tn, dC, (i * NB + j) * tm * tn, db, j * tn, KB)
        rt.synchronize()
        close(host(dC, C0), ref, F32)


@pytest.mark.parametrize("tiles", [(32, 32), (64, 32), (64, 64)], ids=lambda t: "x".join(map(str, t)))
@pytest.mark.parametrize("layout", ["packed", "rowmajor"])
@pytest.mark.parametrize("kb", [2, 6, 32, 5, 0], ids=lambda v: "kb%d" % v)
def test_f32_tiles_of_32_k_pair_their_batch_elements(rtq, tiles, layout, kb):
    """mlir-gen --tiles=32,32,32 (the reference's MLP benchmark, benchmarks/config/base/base.json:74-80): f32 tiles with k = 32 run on
    the loader-wave families when every invoke of the group has an EVEN batch count - a 64-k chunk is the 32-k blocks of two batch
    elements, wherever the stride puts them (packed blocks: stride = 1024; row-major panels: stride = 32 for A, 32 * ldb for B).
    Odd counts (and groups that mix) stay on the generic grouped kernel; both against the oracle, beta = 1 and beta = 0 + bias + relu."""
    rt = rtq
    tm, tn = tiles
    tk = 32
    MB, NB = 3, 5
    rng = np.random.default_rng(tm * 7 + tn + kb)
    K = max(kb, 1) * tk
    if layout == "packed":
        lda, ldb, sa, sb = tk, tn, tm * tk, tk * tn
        a_off = lambda i: i * max(kb, 1) * tm * tk
        b_off = lambda j: j * max(kb, 1) * tk * tn
        X = rng.uniform(-1, 1, MB * max(kb, 1) * tm * tk).astype(np.float32)
        Wt = rng.uniform(-0.3, 0.3, NB * max(kb, 1) * tk * tn).astype(np.float32)
    else:  # A [M][K] row-major, B [K][N] row-major: a batch element is the next 32 columns of A / the next 32 rows of B
        N = NB * tn
        lda, ldb, sa, sb = K, N, tk, tk * N
        a_off = lambda i: i * tm * K
        b_off = lambda j: j * tn
        X = rng.uniform(-1, 1, MB * tm * K).astype(np.float32)
        Wt = rng.uniform(-0.3, 0.3, K * N).astype(np.float32)
    b = rng.uniform(-0.3, 0.3, NB * tn).astype(np.float32)
    C0 = rng.uniform(-1, 1, MB * NB * tm * tn).astype(np.float32)
    for (gflags, ukind, bflags, bkind) in ((4, 5, 4, 1), (0, 0, 0, 0)):
        disp = (F32, tm, tn, tk, lda, ldb, tn, sa, sb, gflags, 0, ukind, bflags, bkind)
        ref = C0.copy()
        for i in range(MB):
            for j in range(NB):
                orc.fused_brgemm(*disp, X, a_off(i), Wt, b_off(j), ref, (i * NB + j) * tm * tn, b, j * tn, kb)
        h = rt.fused_brgemm_dispatch(*disp)
        dX, dW, db, dC = dev(X), dev(Wt), dev(b), dev(C0)
        for rep in range(3):  # recorded, then replayed from the trace cache (the replay launches from the segment's device list)
            dC.copy_(dev(C0))
            rt.synchronize()
            for i in range(MB):
                for j in range(NB):
                    rt.fused_brgemm(F32, h, dX, a_off(i), dW, b_off(j), dC, (i * NB + j) * tm * tn, db, j * tn, kb)
            rt.synchronize()
            close(host(dC, C0), ref, F32)
            ran = rt.last_grouped_kernel()
            assert ("32-k pairs" in ran) == (kb % 2 == 0), (ran, kb)


@pytest.mark.parametrize("tm,tn,items,family", [(64, 64, 272, "64x64,k2"), (64, 64, 544, "64x64>"), (64, 32, 256, "64x32,k4"), (64, 32, 288, "32x32,k4"), (32, 32, 300, "32x32,k4")],
                         ids=["64x64k2", "64x64", "64x32k4", "64x32-as-32x32k4", "32x32k4"])
def test_f32_tiles_of_32_k_pairs_on_every_grouped_family(rtq, tm, tn, items, family):
    """many 32-k tile invokes in one group: the pair mode on each of the four loader-wave families the group size selects
    (launch_gemm_grouped: the largest tile that still gives every CU a workgroup; round 5: 288 tiles of 64x32 would be two rounds of
    workgroups, the second nearly empty - they run as 576 tiles of 32x32, two per CU)"""
    rt = rtq
    kb = 4
    rng = np.random.default_rng(tm + tn + items)
    A = rng.uniform(-1, 1, 4 * kb * tm * 32).astype(np.float32)        # four A tiles, reused
    Bm = rng.uniform(-0.3, 0.3, 8 * kb * 32 * tn).astype(np.float32)   # eight B tiles, reused
    bias = rng.uniform(-0.3, 0.3, 8 * tn).astype(np.float32)
    C0 = np.zeros(items * tm * tn, dtype=np.float32)
    disp = (F32, tm, tn, 32, 32, tn, tn, tm * 32, 32 * tn, 4, 0, 5, 4, 1)
    ref = C0.copy()
    for t in range(items):
        orc.fused_brgemm(*disp, A, (t % 4) * kb * tm * 32, Bm, (t % 8) * kb * 32 * tn, ref, t * tm * tn, bias, (t % 8) * tn, kb)
    h = rt.fused_brgemm_dispatch(*disp)
    dA, dB, db, dC = dev(A), dev(Bm), dev(bias), dev(C0)
    for rep in range(2):
        dC.zero_()
        rt.synchronize()
        for t in range(items):
            rt.fused_brgemm(F32, h, dA, (t % 4) * kb * tm * 32, dB, (t % 8) * kb * 32 * tn, dC, t * tm * tn, db, (t % 8) * tn, kb)
        rt.synchronize()
        close(host(dC, C0), ref, F32)
        ran = rt.last_grouped_kernel()
        assert "32-k pairs" in ran and family in ran, ran


def test_f32_tiles_of_32_k_mixed_batch_counts_in_one_group(rtq):
    """one odd batch count in the group sends the whole group to the generic kernel (the pair flag is a property of the group)"""
    rt = rtq
    rng = np.random.default_rng(77)
    KB = 8
    X = rng.uniform(-1, 1, KB * 1024).astype(np.float32)
    Wt = rng.uniform(-0.3, 0.3, 24 * KB * 1024).astype(np.float32)
    C0 = rng.uniform(-1, 1, 24 * 1024).astype(np.float32)
    b = np.zeros(32, dtype=np.float32)  # (unused: no bias in this dispatch)
    disp = (F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 0, 0, 0, 0, 0)
    counts = [8, 6, 2, 7, 4, 8] * 4
    ref = C0.copy()
    for j, kb in enumerate(counts):
        orc.fused_brgemm(*disp, X, 0, Wt, j * KB * 1024, ref, j * 1024, b, 0, kb)
    h = rt.fused_brgemm_dispatch(*disp)
    dX, dW, dC, db = dev(X), dev(Wt), dev(C0), dev(b)
    for rep in range(2):
        dC.copy_(dev(C0))
        rt.synchronize()
        for j, kb in enumerate(counts):
            rt.fused_brgemm(F32, h, dX, 0, dW, j * KB * 1024, dC, j * 1024, db, 0, kb)
        rt.synchronize()
        close(host(dC, C0), ref, F32)
        assert rt.last_grouped_kernel().startswith(("brgemm_grouped<f32>", "brgemm_f32_lw<32x32,k4> grouped")), rt.last_grouped_kernel()


@pytest.mark.parametrize("nblk", [(2, 3), (4, 16)], ids=["few", "many"])
def test_packed_layers_bf16_64_tiles(rtq, nblk):
    """bf16 + VNNI-2 W with --tiles=64,64,64: the 64x64 bf16 family in grouped mode"""
    rt = rtq
    tm = tn = tk = 64
    MB, NB = nblk
    KB = 3
    rng = np.random.default_rng(MB * 10 + NB)
    X = orc.f32_to_bf16(rng.uniform(-1, 1, MB * KB * tm * tk).astype(np.float32))
    Wt = orc.f32_to_bf16(rng.uniform(-0.3, 0.3, NB * KB * tk * tn).astype(np.float32))
    b = orc.f32_to_bf16(rng.uniform(-0.3, 0.3, NB * tn).astype(np.float32))
    C0 = orc.f32_to_bf16(rng.uniform(-1, 1, MB * NB * tm * tn).astype(np.float32))
    for (gflags, ukind, bflags, bkind) in ((4 | 2048, 5, 4, 1), (2048, 0, 0, 0)):
        disp = (BF16, tm, tn, tk, tk, tn, tn, tm * tk, tk * tn, gflags, 0, ukind, bflags, bkind)
        ref = C0.copy()
        for i in range(MB):
            for j in range(NB):
                orc.fused_brgemm(*disp, X, i * KB * tm * tk, Wt, j * KB * tk * tn, ref, (i * NB + j) * tm * tn, b, j * tn, KB)
        h = rt.fused_brgemm_dispatch(*disp)
        dX, dW, db, dC = dev(X), dev(Wt), dev(b), dev(C0)
        for i in range(MB):
            for j in range(NB):
                rt.fused_brgemm(BF16, h, dX, i * KB * tm * tk, dW, j * KB * tk * tn, dC, (i * NB + j) * tm * tn, db, j * tn, KB)
        rt.synchronize()
        close(host(dC, C0), ref, BF16)


def _tile_program(rng, n_ops, nbuf, ntile):
    """a random program of 32x32 f32 tile ops over `nbuf` buffers of `ntile` tiles: (op, src buffer, src tile, second
    source, dst buffer, dst tile); ops: copy, relu, add, mul-accumulate (gemm with beta = 1)"""
    prog = []
    for _ in range(n_ops):
        op = int(rng.integers(0, 4))
        prog.append((op, int(rng.integers(0, nbuf)), int(rng.integers(0, ntile)), int(rng.integers(0, nbuf)),
                     int(rng.integers(0, ntile)), int(rng.integers(0, nbuf)), int(rng.integers(0, ntile))))
    return prog


def _run_program(prog, bufs, do):
    for (op, sb, st, s2b, s2t, db, dt_) in prog:
        if op == 3 and db in (sb, s2b):
            op = 2  # a gemm must not write a buffer it reads; the eltwise ops may work in place (whole tiles)
        do(op, bufs[sb], st * 1024, bufs[s2b], s2t * 1024, bufs[db], dt_ * 1024)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_trace_cache_replays_and_diverges(rtq, seed):
    """The trace cache replays a group of invokes it has collected before without the footprint bookkeeping. A program of
    dependent tile ops (long independent runs + random dependences) is run 4 times (groups get recorded, then replayed),
    then mutated at a few places - a pointer, an op, an insertion, a deletion - and run again, several times over:
    every run must equal the oracle's replay of the same sequence in program order."""
    rt = rtq
    rng = np.random.default_rng(100 + seed)
    nbuf, ntile = 4, 96
    init = [(rng.uniform(-1, 1, ntile * 1024) * 0.1).astype(np.float32) for _ in range(nbuf)]  # 32-term products contract
    ref = [b.copy() for b in init]
    dbuf = [dev(b) for b in init]
    copy = rt.unary_dispatch(1, F32, 32, 32, 32, 32, 0)
    relu = rt.unary_dispatch(5, F32, 32, 32, 32, 32, 0)
    add = rt.binary_dispatch(1, F32, 32, 32, 32, 32, 32, 0)
    gemm = rt.brgemm_dispatch(F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 0)

    def on_gpu(op, s, so, s2, s2o, d, do_):
        if op == 0:
            rt.unary(F32, copy, s, so, d, do_)
        elif op == 1:
            rt.unary(F32, relu, s, so, d, do_)
        elif op == 2:
            rt.binary(F32, add, s, so, s2, s2o, d, do_)
        else:
            rt.brgemm(F32, gemm, s, so, s2, s2o, d, do_, 1)

    def on_oracle(op, s, so, s2, s2o, d, do_):
        if op == 0:
            orc.unary(1, F32, 32, 32, 32, 32, 0, s, so, d, do_)
        elif op == 1:
            orc.unary(5, F32, 32, 32, 32, 32, 0, s, so, d, do_)
        elif op == 2:
            orc.binary(1, F32, 32, 32, 32, 32, 32, 0, s, so, s2, s2o, d, do_)
        else:
            orc.brgemm(F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 0, s, so, s2, s2o, d, do_, 1)

    stats0 = rt.tile_queue_stats()
    # long runs of one op over consecutive tiles (what compiled layers look like), glued by random single ops
    prog = []
    for run in range(10):
        op = int(rng.integers(0, 4))
        sb, s2b, db = (int(x) for x in rng.permutation(nbuf)[:3])
        n = int(rng.integers(20, 60))
        t0 = int(rng.integers(0, ntile - n))
        prog += [(op, sb, t0 + i, s2b, (t0 + 2 * i) % ntile, db, t0 + i) for i in range(n)]
        prog += _tile_program(rng, int(rng.integers(0, 3)), nbuf, ntile)
    for generation in range(4):
        for rep in range(4):
            _run_program(prog, dbuf, on_gpu)
            _run_program(prog, ref, on_oracle)
            if rep == 2:
                rt.synchronize()  # an external flush point in the middle of the repetitions
        rt.synchronize()
        for b in range(nbuf):
            got = host(dbuf[b], init[b])
            # values grow through the accumulating gemms: compare with the oracle's own result, relative
            assert np.allclose(got, ref[b], rtol=2e-5, atol=1e-6), (generation, b, float(np.abs(got - ref[b]).max()))
            ref[b][:] = got  # continue from the device state (rounding differences must not accumulate into the bar)
        # mutate: change a pointer, change an op, insert, delete
        prog = list(prog)
        for _ in range(3):
            i = int(rng.integers(0, len(prog)))
            o = list(prog[i])
            o[int(rng.integers(1, 7))] = int(rng.integers(0, nbuf if rng.integers(0, 2) else 4))
            o[2], o[4], o[6] = o[2] % ntile, o[4] % ntile, o[6] % ntile
            o[1], o[3], o[5] = o[1] % nbuf, o[3] % nbuf, o[5] % nbuf
            prog[i] = tuple(o)
        prog.insert(int(rng.integers(0, len(prog))), _tile_program(rng, 1, nbuf, ntile)[0])
        del prog[int(rng.integers(0, len(prog)))]
        # keep magnitudes bounded for the next generation
        for b in range(nbuf):
            fresh = (rng.uniform(-1, 1, ntile * 1024) * 0.1).astype(np.float32)
            ref[b][:] = fresh
            dbuf[b].copy_(dev(fresh))
        rt.synchronize()
    launches, checked, replayed, terminated, abandoned = (b - a for a, b in zip(stats0, rt.tile_queue_stats()))
    # the cache did its work: most invokes were queued by replay, and the mutations were noticed
    assert replayed > checked // 2 and terminated > 0 and abandoned > 0, (launches, checked, replayed, terminated, abandoned)


def test_trace_cache_with_several_callers(rtq):
    """Four caller threads run a 3-layer chain of 32x32 tile GEMMs (a barrier after every layer, as the OpenMP loops of the
    compiled code do) eight times over: the scheduler sees the same groups in a different interleaving every time and
    replays them by membership. Then the weights pointer of one layer changes (same shapes, another buffer) and two
    tiles are skipped: the replay must notice. Every run against the oracle."""
    rt = rtq
    rng = np.random.default_rng(77)
    MB, NB, KB = 4, 8, 8  # activations [MB][KB] blocks of 32x32, weights [NB][KB] blocks (NB == KB: layers chain)
    X = (rng.uniform(-1, 1, MB * KB * 1024) * 0.3).astype(np.float32)
    Ws = [(rng.uniform(-1, 1, NB * KB * 1024) * 0.2).astype(np.float32) for _ in range(4)]
    h = rt.brgemm_dispatch(F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 4)
    dX, dW = dev(X), [dev(w) for w in Ws]
    acts = [dev(np.zeros(MB * NB * 1024, np.float32)) for _ in range(3)]
    nthr = 4
    barrier = threading.Barrier(nthr)
    errors = []

    def run(weights, skip):
        def worker(tid):
            try:
                src = dX
                for layer in range(3):
                    for t in range(tid, MB * NB, nthr):
                        if (layer, t) in skip:
                            continue
                        i, j = divmod(t, NB)
                        rt.brgemm(F32, h, src, i * KB * 1024, weights[layer], j * KB * 1024, acts[layer], t * 1024, KB)
                    barrier.wait()
                    src = acts[layer]
            except Exception as ex:  # noqa: BLE001
                errors.append(repr(ex))
                barrier.abort()
        ths = [threading.Thread(target=worker, args=(w,)) for w in range(nthr)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    def oracle(weights, skip, prev):
        src, outs = X, []
        for layer in range(3):
            out = prev[layer].copy()
            for t in range(MB * NB):
                if (layer, t) in skip:
                    continue
                i, j = divmod(t, NB)
                orc.brgemm(F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 4, src, i * KB * 1024, weights[layer], j * KB * 1024, out, t * 1024, KB)
            outs.append(out)
            src = out
        return outs

    prev = [np.zeros(MB * NB * 1024, np.float32) for _ in range(3)]
    stats0 = rt.tile_queue_stats()
    for rep in range(8):
        run(dW[:3], set())
    rt.synchronize()
    assert not errors, errors
    ref = oracle(Ws[:3], set(), prev)
    for layer in range(3):
        close(host(acts[layer], X), ref[layer], F32)
    # diverge: layer 1 reads other weights, two tiles are not computed (they keep their old values)
    skip = {(0, 5), (2, 17)}
    for rep in range(3):
        run([dW[0], dW[3], dW[2]], skip)
    rt.synchronize()
    assert not errors, errors
    ref2 = oracle([Ws[0], Ws[3], Ws[2]], skip, ref)
    for layer in range(3):
        close(host(acts[layer], X), ref2[layer], F32)
    launches, checked, replayed, terminated, abandoned = (b - a for a, b in zip(stats0, rt.tile_queue_stats()))
    # (how a divergence ends a replay depends on the interleaving: an unknown invoke abandons it, an invoke of another recorded
    # group that arrives at a complete group just terminates it - the results above are the check, the counters only say that
    # the cache was in use)
    assert replayed > 0 and abandoned + terminated > 0, (launches, checked, replayed, terminated, abandoned)


def test_single_layer_timing_loop_replays_without_abandoning(rtq):
    """the reference's one-layer benchmarks (benchmarks/config/matmul/*.json, fc/*.json): tpp-run calls the SAME group of tile invokes N
    times with nothing in between; the invoke that ends an iteration's group is the group's own first member again. Round 5: that wrap
    is a known way for a complete group to end (every iteration used to abandon its replay and rebuild the bookkeeping: the run was
    host-bound). 20 iterations of C += A W on 48 tiles: result = 20 accumulated passes, and after the first few groups nothing is
    abandoned any more."""
    rt = rtq
    M, N, K, tm, tn, tk = 128, 768, 768, 32, 64, 64
    rng = np.random.default_rng(77)
    MB, NB, KB = M // tm, N // tn, K // tk
    A = rng.uniform(-1, 1, M * K).astype(np.float32)
    W = rng.uniform(-0.05, 0.05, K * N).astype(np.float32)
    C0 = rng.uniform(-1, 1, M * N).astype(np.float32)
    h = rt.brgemm_dispatch(F32, tm, tn, tk, tk, tn, tn, tm * tk, tk * tn, 0)
    dA, dW, dC = dev(A), dev(W), dev(C0)
    iters = 20

    def one_iteration():
        for i in range(MB):
            for j in range(NB):
                rt.brgemm(F32, h, dA, i * KB * tm * tk, dW, j * KB * tk * tn, dC, (i * NB + j) * tm * tn, KB)

    for _ in range(4):  # recorded, replayed, the wrap learnt
        one_iteration()
    stats0 = rt.tile_queue_stats()
    for _ in range(iters - 4):
        one_iteration()
    rt.synchronize()
    launches, checked, replayed, terminated, abandoned = (b - a for a, b in zip(stats0, rt.tile_queue_stats()))
    # (lock-free arrivals are counted when their group's window closes: the last warm-up group may fall on either side of stats0)
    assert abandoned == 0 and checked == 0 and abs(replayed - (iters - 4) * MB * NB) <= MB * NB, (launches, checked, replayed, terminated, abandoned)
    ref = C0.copy()
    for _ in range(iters):
        for i in range(MB):
            for j in range(NB):
                orc.brgemm(F32, tm, tn, tk, tk, tn, tn, tm * tk, tk * tn, 0, A, i * KB * tm * tk, W, j * KB * tk * tn, ref, (i * NB + j) * tm * tn, KB)
    # 20 accumulated passes: the two sides' summation orders differ in every pass - the element-wise bar per pass (1e-5 relative + the
    # f32 dot-product floor (K + 2) eps sum |a||w|), times the number of passes
    got = host(dC, C0).astype(np.float64)
    Af = A.reshape(MB, KB, tm, tk).transpose(0, 2, 1, 3).reshape(M, K).astype(np.float64)
    Wf = W.reshape(NB, KB, tk, tn).transpose(1, 2, 0, 3).reshape(K, N).astype(np.float64)
    mag = (np.abs(Af) @ np.abs(Wf)).reshape(MB, tm, NB, tn).transpose(0, 2, 1, 3).reshape(-1)
    bar = 1e-5 * np.abs(ref) + iters * (K + 2) * 2.0 ** -24 * (mag + np.abs(ref))
    assert (np.abs(got - ref) <= bar).all(), float((np.abs(got - ref) / bar).max())


def test_launch_thread_on_off_same_bits_same_order():
    """The launch thread (include/tpp_xsmm_abi.h xsmm_hip_set_launch_thread; csrc/rt_launcher.h): complete replayed groups are
    launched by a helper thread, everything else - the first (recorded) pass, partial groups, invokes outside the queue, copies,
    synchronisation points - waits for the hand-overs to have left. A chain of three DEPENDENT packed layers repeated 12 times
    (replays from the second pass on), with an unqueued whole-layer invoke and an in-place unary between the passes (both read
    what the last group wrote and are read by the next one): the bits with the thread on equal the bits with it off, the thread
    took hand-overs while on and none while off."""
    rt = pkg.get_runtime()
    prev_async, prev_q = rt.set_async(True), rt.set_tile_queue(1)
    try:
        MB, NB, KB = 4, 8, 8
        rng = np.random.default_rng(606)
        X = rng.uniform(-1, 1, MB * KB * 1024).astype(np.float32)
        Ws = [rng.uniform(-0.3, 0.3, NB * KB * 1024).astype(np.float32) for _ in range(3)]
        bs = [rng.uniform(-0.3, 0.3, NB * 32).astype(np.float32) for _ in range(3)]
        disp = (F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 4, 0, 5, 4, 1)
        h = rt.fused_brgemm_dispatch(*disp)
        # the unqueued invoke: one big plain gemm over the flat view of the last activations (128 x 256 = [128][256] floats) into a side buffer
        hb = rt.gemm_dispatch(F32, 128, 128, 256, 256, 128, 128, 4)
        hr = rt.unary_dispatch(5, F32, 128, 256, 256, 256, 0)  # relu in place, a whole-buffer invoke (not queue-sized)
        Wside = rng.uniform(-0.1, 0.1, 256 * 128).astype(np.float32)

        def run(on):
            prev = rt.set_launch_thread(on)
            assert prev in (0, 1)
            dX, dW, db = dev(X), [dev(w) for w in Ws], [dev(b) for b in bs]
            dA = [dev(np.zeros(MB * NB * 1024, dtype=np.float32)) for _ in range(3)]
            dS, dWs = dev(np.zeros(128 * 128, dtype=np.float32)), dev(Wside)
            s0 = rt.launch_thread_stats()
            sides = []
            for it in range(12):
                cur = dX
                for l in range(3):
                    for i in range(MB):
                        for j in range(NB):
                            rt.fused_brgemm(F32, h, cur, i * KB * 1024, dW[l], j * KB * 1024, dA[l], (i * NB + j) * 1024, db[l], j * 32, KB)
                    cur = dA[l]
                rt.gemm(F32, hb, dA[2], 0, dWs, 0, dS, 0)  # outside the queue: behind the last group's launch
                rt.unary(F32, hr, dA[2], 0, dA[2], 0)
                if it in (0, 5, 11):
                    rt.synchronize()
                    sides.append(host(dS, X).copy())
            rt.synchronize()
            s1 = rt.launch_thread_stats()
            return [host(a, X).copy() for a in dA] + sides, s1[0] - s0[0]

        on, handed_on = run(True)
        off, handed_off = run(False)
        rt.set_launch_thread(True)
        assert handed_on >= 3 * 8 and handed_off == 0, (handed_on, handed_off)
        for a, b in zip(on, off):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert np.abs(on[2]).max() > 0
    finally:
        rt.synchronize()
        rt.set_tile_queue(prev_q)
        rt.set_async(prev_async)

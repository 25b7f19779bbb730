"""The reference's benchmark shape set on the HIP path (VERDICT r4 row g): one layer of every distinct --tiles setting of
benchmarks/config/matmul/*.json and fc/*.json (64,64,64 / 64,48,64 / 32,48,32 / 32,64,64 / 32,32,32; mlir-gen's tiles are
(batch tile, out-feature tile, in-feature tile) = the tile BRGEMM's (m, n, k), MLIRGen.cpp:641-676), replayed as the compiler emits
it - packed block layouts, one fused_brgemm dispatch [m,n,k,k,n,n,m*k,k*n], (M/m)*(N/n) invokes with br = K/k through the tile queue -
and as ONE whole-layer dispatch, f32 and bf16 + VNNI-2, against the oracle. And the SPLIT kernels behind the skinny shapes (the
batch-reduce range of an output tile over several workgroups): every forced split count against the oracle, bit-reproducible from
run to run, with beta = 1, bias + relu, ragged chunk counts and empty ranges."""
import importlib

import numpy as np
import pytest

from oracle import pyoracle as orc
from test_parity_gpu import BF16, F32, VB, check_close, dev, host, rand

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("tpp-mlir_amd")


@pytest.fixture(scope="module")
def rt():
    r = pkg.get_runtime()
    assert r.device_count() >= 1, "no HIP device visible: the gpu tests need an MI355X"
    yield r
    r.force_split(-1)


def pack_a(X, M, K, tm, tk):  # [M][K] -> [M/tm][K/tk][tm][tk]
    return np.ascontiguousarray(X.reshape(M // tm, tm, K // tk, tk).transpose(0, 2, 1, 3)).reshape(-1)


def pack_w(W, K, N, tk, tn, vnni):  # [K][N] -> [N/tn][K/tk][tk][tn] (VNNI-v: [N/tn][K/tk][tk/v][tn][v])
    blk = W.reshape(K // tk, tk, N // tn, tn).transpose(2, 0, 1, 3)  # [NB][KB][tk][tn]
    if vnni:
        blk = blk.reshape(N // tn, K // tk, tk // vnni, vnni, tn).transpose(0, 1, 2, 4, 3)
    return np.ascontiguousarray(blk).reshape(-1)


def pack_c(C, M, N, tm, tn):
    return np.ascontiguousarray(C.reshape(M // tm, tm, N // tn, tn).transpose(0, 2, 1, 3)).reshape(-1)


def unpack_c(Cp, M, N, tm, tn):
    return np.ascontiguousarray(Cp.reshape(M // tm, N // tn, tm, tn).transpose(0, 2, 1, 3)).reshape(M, N)


# (M, N, K, tiles) - one benchmark per distinct tile shape, the skinniest of its kind
LAYERS = [
    (128, 1024, 1024, (64, 64, 64)),   # benchmarks/config/fc/128x1024x1024.json
    (128, 768, 2304, (64, 48, 64)),    # fc/128x768x2304.json:40-64
    (128, 768, 3072, (32, 48, 32)),    # matmul/128x768x3072.json:37-51
    (128, 768, 768, (32, 64, 64)),     # fc/128x768x768.json
    (1024, 352, 512, (32, 32, 32)),    # fc/1024x352x512.json
]


@pytest.mark.parametrize("fc", [False, True], ids=["matmul", "fc"])
@pytest.mark.parametrize("dt", [F32, BF16], ids=["f32", "bf16vnni2"])
@pytest.mark.parametrize("M,N,K,tiles", LAYERS, ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_one_layer_of_every_tile_shape_as_the_compiler_emits_it(rt, M, N, K, tiles, dt, fc):
    tm, tn, tk = tiles
    rng = np.random.default_rng(M + N + K + tm)
    X = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    W = (rng.uniform(-1, 1, (K, N)) / np.sqrt(K)).astype(np.float32)
    C0 = rng.uniform(-1, 1, (M, N)).astype(np.float32)  # --kernel=args: the output is an argument, the layer accumulates into it
    bias = rng.uniform(-1, 1, N).astype(np.float32)
    if dt == BF16:
        X, W, C0, bias = (orc.bf16_to_f32(orc.f32_to_bf16(v.reshape(-1))).reshape(v.shape) for v in (X, W, C0, bias))
    conv = (lambda v: v) if dt == F32 else orc.f32_to_bf16
    vn = 2 if dt == BF16 else 0
    flags = VB if dt == BF16 else 0  # beta = 1
    # the oracle on the flat tensors (one whole-layer call; the packed replay computes the same sums per element)
    ref = conv(C0.reshape(-1).copy())
    Wflat = W if not vn else W.reshape(K // 2, 2, N).transpose(0, 2, 1)
    a_o, w_o, b_o = conv(X.reshape(-1)), conv(np.ascontiguousarray(Wflat).reshape(-1)), conv(bias)
    if fc:
        orc.fused_brgemm(dt, M, N, K, K, N, N, 0, 0, flags, 0, 5, 4, 1, a_o, 0, w_o, 0, ref, 0, b_o, 0, 1)
    else:
        orc.brgemm(dt, M, N, K, K, N, N, 0, 0, flags, a_o, 0, w_o, 0, ref, 0, 1)
    mag = None
    if dt == F32:
        mag = (np.abs(X).astype(np.float64) @ np.abs(W).astype(np.float64) + np.abs(C0) + (np.abs(bias)[None, :] if fc else 0)).reshape(-1)
    # (1) as the compiler emits it: packed blocks, tile invokes through the tile queue
    dA, dW, dC, dB = dev(conv(pack_a(X, M, K, tm, tk))), dev(conv(pack_w(W, K, N, tk, tn, vn))), dev(conv(pack_c(C0, M, N, tm, tn))), dev(conv(bias))
    disp = (dt, tm, tn, tk, tk, tn, tn, tm * tk, tk * tn, flags)
    h = rt.fused_brgemm_dispatch(*disp, 0, 5, 4, 1) if fc else rt.brgemm_dispatch(*disp)
    MB, NB, KB = M // tm, N // tn, K // tk
    old_async, old_q = rt.set_async(True), rt.set_tile_queue(1)
    try:
        for i in range(MB):
            for j in range(NB):
                if fc:
                    rt.fused_brgemm(dt, h, dA, i * KB * tm * tk, dW, j * KB * tk * tn, dC, (i * NB + j) * tm * tn, dB, j * tn, KB)
                else:
                    rt.brgemm(dt, h, dA, i * KB * tm * tk, dW, j * KB * tk * tn, dC, (i * NB + j) * tm * tn, KB)
        rt.synchronize()
        grouped = rt.last_grouped_kernel()
    finally:
        rt.set_tile_queue(old_q)
        rt.set_async(old_async)
    got = host(dC, conv(C0.reshape(-1)))
    got = unpack_c(orc.bf16_to_f32(got) if dt == BF16 else got, M, N, tm, tn).reshape(-1)
    check_close(got if dt == F32 else orc.f32_to_bf16(got), ref, dt, "tile invokes %s tiles %s [%s]" % ((M, N, K), tiles, grouped), mag=mag, K=K)
    # (2) ONE whole-layer dispatch on the flat tensors (k = 64 chunks)
    dA2, dW2, dC2 = dev(a_o), dev(w_o), dev(conv(C0.reshape(-1)))
    disp = (dt, M, N, 64, K, N, N, 64, 64 * N, flags)
    h2 = rt.fused_brgemm_dispatch(*disp, 0, 5, 4, 1) if fc else rt.brgemm_dispatch(*disp)
    if fc:
        rt.fused_brgemm(dt, h2, dA2, 0, dW2, 0, dC2, 0, dB, 0, K // 64)
    else:
        rt.brgemm(dt, h2, dA2, 0, dW2, 0, dC2, 0, K // 64)
    check_close(host(dC2, ref), ref, dt, "whole layer %s [%s]" % ((M, N, K), rt.kernel_name(h2)), mag=mag, K=K)


SPLIT_CASES = [
    # m, n, k, br, forced variant, beta0, bias, relu
    (128, 1024, 64, 64, 9, True, False, False),      # matmul 128x1024x4096 as one dispatch, on the 32x32 + K4 tile
    (128, 768, 64, 36, 9, False, True, True),        # fc 128x768x2304: 36 chunks (uneven ranges for most split counts)
    (128, 256, 64, 7, 6, False, True, False),        # 64x64 + K2 tile, fewer chunks than the largest split counts (empty ranges)
    (64, 96, 64, 5, 7, True, False, True),           # 64x32 + K4 tile
    (96, 160, 128, 3, 9, False, False, False),       # 32x32 + K4 tile, k = 2 chunks per batch element: a range may start inside an element
]


@pytest.mark.parametrize("m,n,k,br,force,beta0,bias,relu", SPLIT_CASES)
def test_split_batch_reduce_every_count_against_the_oracle_and_reproducible(rt, m, n, k, br, force, beta0, bias, relu):
    rng = np.random.default_rng(m * 7 + n + br)
    K = k * br
    A, B = rand(rng, m * K + 8, F32), rand(rng, K * n + 8, F32)
    C, D = rand(rng, m * n + 8, F32), rand(rng, n + 8, F32)
    flags = 4 if beta0 else 0
    fused = bias or relu
    ref = C.copy()
    args = (F32, m, n, k, K, n, n, k, k * n, flags)
    if fused:
        orc.fused_brgemm(*args, 0, 5 if relu else 0, 4 if bias else 0, 1 if bias else 0, A, 4, B, 8, ref, 4, D, 4, br)
    else:
        orc.brgemm(*args, A, 4, B, 8, ref, 4, br)
    a2 = np.abs(A[4:4 + m * K]).reshape(m, K).astype(np.float64)
    b2 = np.abs(B[8:8 + K * n]).reshape(K, n).astype(np.float64)
    mag = np.zeros(m * n + 8)
    mag[4:4 + m * n] = (a2 @ b2 + (0 if beta0 else np.abs(C[4:4 + m * n]).reshape(m, n)) + (np.abs(D[4:4 + n])[None, :] if bias else 0)).reshape(-1)
    if force is not None:
        rt.force_variant(force)
    try:
        h = rt.fused_brgemm_dispatch(*args, 0, 5 if relu else 0, 4 if bias else 0, 1 if bias else 0) if fused else rt.brgemm_dispatch(*args)
    finally:
        rt.force_variant(-1)
    dA, dB, dD = dev(A), dev(B), dev(D)
    seen = {}
    try:
        for S in (1, 2, 3, 4, 5, 8, 16, -1):
            rt.force_split(S)
            outs = []
            for rep in range(3):
                dC = dev(C)
                if fused:
                    rt.fused_brgemm(F32, h, dA, 4, dB, 8, dC, 4, dD, 4, br)
                else:
                    rt.brgemm(F32, h, dA, 4, dB, 8, dC, 4, br)
                outs.append(host(dC, C))
            assert all(np.array_equal(outs[0], o) for o in outs[1:]), "split %d: results differ from run to run" % S
            check_close(outs[0], ref, F32, "split %d m%d n%d k%d br%d [%s]" % (S, m, n, k, br, rt.kernel_name(h)), mag=mag, K=K)
            seen[S] = outs[0]
    finally:
        rt.force_split(-1)
    # the guard bytes around C stay untouched, and a forced count really changes the order of additions somewhere (i.e. the split ran)
    assert np.array_equal(seen[4][:4], C[:4]) and np.array_equal(seen[4][4 + m * n:], C[4 + m * n:])
    assert any(not np.array_equal(seen[1], seen[S]) for S in (2, 3, 4))


def test_split_groups_through_the_tile_queue(rt):
    """a skinny layer as tile invokes (32 invokes of 64x64x64, br = 64: matmul 128x1024x4096): the group runs split; the result is
    the oracle's within the bar and the same bits as the same group run again"""
    M, N, K, t = 128, 1024, 4096, 64
    rng = np.random.default_rng(11)
    X = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    W = (rng.uniform(-1, 1, (K, N)) / 64).astype(np.float32)
    C0 = rng.uniform(-1, 1, (M, N)).astype(np.float32)
    ref = C0.reshape(-1).copy()
    orc.brgemm(F32, M, N, K, K, N, N, 0, 0, 0, X.reshape(-1), 0, W.reshape(-1), 0, ref, 0, 1)
    mag = (np.abs(X).astype(np.float64) @ np.abs(W).astype(np.float64) + np.abs(C0)).reshape(-1)
    dA, dW = dev(pack_a(X, M, K, t, t)), dev(pack_w(W, K, N, t, t, 0))
    h = rt.brgemm_dispatch(F32, t, t, t, t, t, t, t * t, t * t, 0)
    MB, NB, KB = M // t, N // t, K // t
    old_async, old_q = rt.set_async(True), rt.set_tile_queue(1)
    outs = []
    try:
        for rep in range(3):
            dC = dev(pack_c(C0, M, N, t, t))
            for i in range(MB):
                for j in range(NB):
                    rt.brgemm(F32, h, dA, i * KB * t * t, dW, j * KB * t * t, dC, (i * NB + j) * t * t, KB)
            rt.synchronize()
            outs.append(host(dC, ref))
            assert "split" in rt.last_grouped_kernel(), rt.last_grouped_kernel()
    finally:
        rt.set_tile_queue(old_q)
        rt.set_async(old_async)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    check_close(unpack_c(outs[0], M, N, t, t).reshape(-1), ref, F32, "split group", mag=mag, K=K)


# ---------------------------------------------------------------- the 32x16 tiles (brgemm_f32_lw16.hip, variant 11)
LW16_CASES = [
    # m, n, k, br, kwargs
    (32, 16, 64, 1, dict(beta0=True)),
    (32, 16, 64, 1, dict()),                                            # beta = 1
    (64, 48, 64, 3, dict(bias=True, relu=True)),                        # n = 48: three column tiles
    (128, 96, 128, 2, dict(beta0=True, bias=True, lda=300, ldb=100, ldc=104, offs=(4, 8, 4, 4))),   # two chunks per batch element, strides
    (96, 160, 64, 5, dict(sa=64, sb=64 * 160, lda=64 * 5, relu=True)),  # batch-reduce along k of one row-major A (the whole-layer form)
    (128, 1024, 64, 16, dict(sa=64, sb=64 * 1024, lda=1024, beta0=True, bias=True, relu=True)),      # fc 128x1024x1024: 256 tiles, XCD-blocked order
    (128, 768, 64, 12, dict(sa=64, sb=64 * 768, lda=768)),              # matmul 128x768x768: 192 tiles
    (32, 80, 64, 0, dict(bias=True)),                                   # empty batch: C = C + bias
]


@pytest.mark.parametrize("case", LW16_CASES, ids=lambda c: "m%d_n%d_k%d_br%d" % c[:4])
def test_f32_lw16_tiles_against_the_oracle(rt, case):
    from test_parity_gpu import gemm_case
    m, n, k, br, kw = case
    name = gemm_case(rt, F32, m, n, k, br, seed=m + n + k + br, force=11, **kw)
    assert "lw16<32x16" in name, name


@pytest.mark.parametrize("br", [1, 2, 3, 4, 5, 6, 7, 8, 9, 13])
def test_f32_lw16_every_chunk_stream_length(rt, br):
    """every position of the 4-slot ring (1 .. 13 chunks), both accumulator starts"""
    from test_parity_gpu import gemm_case
    for beta0 in (True, False):
        name = gemm_case(rt, F32, 64, 32, 64, br, beta0=beta0, bias=not beta0, relu=beta0, seed=br, force=11, offs=(4, 4, 4, 4))
        assert "lw16" in name, name


def test_skinny_whole_layers_pick_the_half_width_tiles(rt):
    """the reference's M = 128 shapes: at most one 32x16 tile per CU -> the half-width tiles; wider outputs keep theirs"""
    for (m, n, want) in ((128, 1024, "lw16<32x16"), (128, 768, "lw16<32x16"), (128, 3072, "lw<32x32"), (256, 768, "lw<32x32"), (256, 1024, "lw<32x32"), (512, 1024, "lw<64x32"), (1024, 1024, "lw<64x64")):
        h = rt.brgemm_dispatch(F32, m, n, 64, 1024, n, n, 64, 64 * n, 0)
        assert want in rt.kernel_name(h), (m, n, rt.kernel_name(h))


@pytest.mark.parametrize("m,n,br,beta0,bias,relu", [(128, 256, 32, True, True, True), (64, 96, 24, False, False, False), (256, 1024, 64, True, False, True),
                                                    (32, 32, 25, False, True, False)])
def test_bf16_small_outputs_with_a_long_reduction_switch_tiles_at_invoke_time(rt, m, n, br, beta0, bias, relu):
    """bf16 VNNI-2 layers planned on the 32x32 K-split kernel (small outputs) run on the loader-wave tiles when K = br * 64 >= 1536 -
    the batch count arrives with the invoke: 32x32 + K2 when the output is at most one 32x32 tile per CU, else 32x64 + K2; against the
    oracle, and a short reduction on the same handle stays on the handle's own kernel"""
    from test_parity_gpu import gemm_case
    k = 64
    K = k * br
    name = gemm_case(rt, BF16, m, n, k, br, lda=K + 8, sa=k, sb=k * n, vnni=True, beta0=beta0, bias=bias, relu=relu, seed=m + n + br, offs=(8, 8, 8, 4))
    assert "small<32x32" in name, name  # the dispatch-time plan
    want = "32x32,k2" if (m // 32) * (n // 32) <= 256 and n % 64 == 0 or n % 64 else "32x64,k2"
    ran = rt.last_refined_kernel()
    if n % 64 == 0:
        assert "long reduction" in ran and ("32x32,k2" in ran if (m // 32) * (n // 32) <= 256 else "32x64,k2" in ran), ran
    else:
        assert ran == "", ran  # (n = 96 / 32: the loader-wave tiles need n % 64 == 0 - the handle's own kernel)
    gemm_case(rt, BF16, m, n, k, 4, lda=K + 8, sa=k, sb=k * n, vnni=True, beta0=beta0, bias=bias, relu=relu, seed=m + n, offs=(8, 8, 8, 4))
    assert rt.last_refined_kernel() == ""


@pytest.mark.parametrize("forced", [-1, 2, 3, 5, 16])
def test_bf16_split_groups_through_the_tile_queue(rt, forced):
    """a skinny bf16 layer as tile invokes (matmul 128x1024x2048 as 32 invokes of 64x64x64, br = 32, VNNI-2 W, C += ...): the group runs
    on the 32x32 K-split kernel with the K steps of a tile over several workgroups (the model's count, and forced counts incl. more
    workgroups than a tile has 64-step shares); one bf16 ulp + the f32 floor against the oracle, the same bits on every run"""
    M, N, K, t = 128, 1024, 2048, 64
    rng = np.random.default_rng(21)
    X = orc.bf16_to_f32(orc.f32_to_bf16(rng.uniform(-1, 1, M * K).astype(np.float32))).reshape(M, K)
    W = orc.bf16_to_f32(orc.f32_to_bf16((rng.uniform(-1, 1, K * N) / 32).astype(np.float32))).reshape(K, N)
    C0 = orc.bf16_to_f32(orc.f32_to_bf16(rng.uniform(-1, 1, M * N).astype(np.float32))).reshape(M, N)
    ref = orc.f32_to_bf16(C0.reshape(-1).copy())
    Wv = np.ascontiguousarray(W.reshape(K // 2, 2, N).transpose(0, 2, 1)).reshape(-1)
    orc.brgemm(BF16, M, N, K, K, N, N, 0, 0, VB, orc.f32_to_bf16(X.reshape(-1)), 0, orc.f32_to_bf16(Wv), 0, ref, 0, 1)
    dA, dW = dev(orc.f32_to_bf16(pack_a(X, M, K, t, t))), dev(orc.f32_to_bf16(pack_w(W, K, N, t, t, 2)))
    h = rt.brgemm_dispatch(BF16, t, t, t, t, t, t, t * t, t * t, VB)
    MB, NB, KB = M // t, N // t, K // t
    old_async, old_q = rt.set_async(True), rt.set_tile_queue(1)
    rt.force_split(forced)
    outs = []
    try:
        for rep in range(3):
            dC = dev(orc.f32_to_bf16(pack_c(C0, M, N, t, t)))
            for i in range(MB):
                for j in range(NB):
                    rt.brgemm(BF16, h, dA, i * KB * t * t, dW, j * KB * t * t, dC, (i * NB + j) * t * t, KB)
            rt.synchronize()
            outs.append(host(dC, ref))
            # (the model's own choice - forced == -1 - is the grouped loader-wave tile since round 6: 32 chunks per tile; a forced count
            # keeps the group on the K-split kernel)
            want = "brgemm_bf16_lw<32x32,k2> grouped" if forced < 0 else ("small32 grouped, split" if forced > 1 else "small32 grouped")
            assert want in rt.last_grouped_kernel(), rt.last_grouped_kernel()
    finally:
        rt.force_split(-1)
        rt.set_tile_queue(old_q)
        rt.set_async(old_async)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    got = unpack_c(orc.bf16_to_f32(outs[0]), M, N, t, t).reshape(-1)
    check_close(orc.f32_to_bf16(got), ref, BF16, "bf16 split group (forced %d)" % forced)


def test_f32_32k_pair_tiles_with_a_long_batch_split_in_the_group(rt):
    """the compiler-native 32x32x32 f32 tile with a LONG batch (br = 128: 64 pair chunks) in a small group (6 invokes): the pair mode of
    the loader-wave kernel AND the split of a tile's batch-reduce range over several workgroups in one launch - a range then starts at
    an arbitrary PAIR of batch elements; against the oracle, same bits on every run, and bias + relu behind the ordered sum"""
    t, br, items = 32, 128, 6
    rng = np.random.default_rng(8)
    A = rng.uniform(-1, 1, 2 * br * t * t).astype(np.float32)       # two A tile rows
    B = (rng.uniform(-1, 1, 3 * br * t * t) / 8).astype(np.float32)  # three B tile columns
    bias = rng.uniform(-1, 1, 3 * t).astype(np.float32)
    C0 = rng.uniform(-1, 1, items * t * t).astype(np.float32)
    disp = (F32, t, t, t, t, t, t, t * t, t * t, 0, 0, 5, 4, 1)  # beta = 1, bias + relu
    ref = C0.copy()
    for it in range(items):
        orc.fused_brgemm(*disp, A, (it // 3) * br * t * t, B, (it % 3) * br * t * t, ref, it * t * t, bias, (it % 3) * t, br)
    h = rt.fused_brgemm_dispatch(*disp)
    dA, dB, db = dev(A), dev(B), dev(bias)
    old_async, old_q = rt.set_async(True), rt.set_tile_queue(1)
    outs = []
    try:
        for rep in range(3):
            dC = dev(C0)
            for it in range(items):
                rt.fused_brgemm(F32, h, dA, (it // 3) * br * t * t, dB, (it % 3) * br * t * t, dC, it * t * t, db, (it % 3) * t, br)
            rt.synchronize()
            outs.append(host(dC, C0))
            ran = rt.last_grouped_kernel()
            assert "32-k pairs" in ran and "split" in ran, ran
    finally:
        rt.set_tile_queue(old_q)
        rt.set_async(old_async)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    mag = np.abs(C0).astype(np.float64)
    for it in range(items):
        a2 = np.abs(A[(it // 3) * br * t * t:][:br * t * t]).reshape(br, t, t).astype(np.float64)
        b2 = np.abs(B[(it % 3) * br * t * t:][:br * t * t]).reshape(br, t, t).astype(np.float64)
        mag[it * t * t:(it + 1) * t * t] += (np.einsum("bik,bkj->ij", a2, b2) + np.abs(bias[(it % 3) * t:(it % 3 + 1) * t])[None, :]).reshape(-1)
    check_close(outs[0], ref, F32, "32-k pairs + split", mag=mag, K=br * t)


@pytest.mark.parametrize("fc", [False, True], ids=["matmul_beta1", "fc_beta0_bias_relu"])
@pytest.mark.parametrize("vn", [2, 4], ids=["vnni2", "vnni4"])
@pytest.mark.parametrize("M,N,K,callers", [(1024, 1280, 256, 1), (512, 2560, 128, 1), (1024, 2560, 1024, 1), (1024, 1280, 256, 4)], ids=lambda v: str(v))
def test_bf16_64_tile_invokes_replayed_as_quads_on_the_128_tile(rt, M, N, K, callers, vn, fc):
    """QUADS (csrc/rt_rewrites.h detect_quads, brgemm_bf16_lw GRP = 2): a recorded group of 64x64x64 bf16 tile invokes over packed
    blocks (benchmarks/config/fc/1024x2560x1024.json:40-64 as mlir-gen emits it: 16 x 40 invokes, br = 16) that forms a grid of
    item rows and item columns is REPLAYED as 2 x 2 blocks on the 128x128 loader-wave tile when the tile model prefers it. Three
    passes of the same layer: recorded (items on the grouped 64x64 tile), replayed twice (quads). Against the oracle after every
    pass (the matmul flavour accumulates: C += X W, rounded to bf16 each pass - the oracle's pass starts from the device's previous
    result), and the BETA_0 flavour bit-identical between the item pass and the quad passes (same k order per element)."""
    tm = tn = tk = 64
    old_v = rt.set_vnni_factor(vn)
    old_o = orc.set_vnni_factor(vn)
    old_async, old_q = rt.set_async(True), rt.set_tile_queue(1)
    try:
        rng = np.random.default_rng(M + N + K + vn)
        X = rng.uniform(-1, 1, (M, K)).astype(np.float32)
        W = (rng.uniform(-1, 1, (K, N)) / np.sqrt(K)).astype(np.float32)
        C0 = rng.uniform(-1, 1, (M, N)).astype(np.float32)
        bias = rng.uniform(-1, 1, N).astype(np.float32)
        X, W, C0, bias = (orc.bf16_to_f32(orc.f32_to_bf16(v.reshape(-1))).reshape(v.shape) for v in (X, W, C0, bias))
        conv = orc.f32_to_bf16
        flags = VB | (4 if fc else 0)
        Wv = np.ascontiguousarray(W.reshape(K // vn, vn, N).transpose(0, 2, 1)).reshape(-1)  # flat VNNI-vn [K/vn][N][vn] for the oracle
        a_o, w_o, b_o = conv(X.reshape(-1)), conv(Wv), conv(bias)
        dA, dW, dB = dev(conv(pack_a(X, M, K, tm, tk))), dev(conv(pack_w(W, K, N, tk, tn, vn))), dev(conv(bias))
        dC = dev(conv(pack_c(C0, M, N, tm, tn)))
        disp = (BF16, tm, tn, tk, tk, tn, tn, tm * tk, tk * tn, flags)
        h = rt.fused_brgemm_dispatch(*disp, 0, 5, 4, 1) if fc else rt.brgemm_dispatch(*disp)
        MB, NB, KB = M // tm, N // tn, K // tk
        kernels, outs = [], []
        for p in range(3):
            # (the accumulating flavour: the oracle's pass starts from what the device holds - a rounding flip of an earlier pass, allowed
            # by the bar, must not be counted again in the next one)
            start = host(dC, conv(C0.reshape(-1)))
            ref = orc.f32_to_bf16(unpack_c(orc.bf16_to_f32(start), M, N, tm, tn).reshape(-1))
            def rows(i0, i1):  # (callers > 1: the reference's OpenMP team over the tile grid, static schedule by block rows)
                for i in range(i0, i1):
                    for j in range(NB):
                        if fc:
                            rt.fused_brgemm(BF16, h, dA, i * KB * tm * tk, dW, j * KB * tk * tn, dC, (i * NB + j) * tm * tn, dB, j * tn, KB)
                        else:
                            rt.brgemm(BF16, h, dA, i * KB * tm * tk, dW, j * KB * tk * tn, dC, (i * NB + j) * tm * tn, KB)
            if callers == 1:
                rows(0, MB)
            else:
                import threading
                ts = [threading.Thread(target=rows, args=(c * MB // callers, (c + 1) * MB // callers)) for c in range(callers)]
                for t_ in ts:
                    t_.start()
                for t_ in ts:
                    t_.join()
            rt.synchronize()
            kernels.append(rt.last_grouped_kernel())
            got = host(dC, ref)
            outs.append(got.copy())
            if fc:
                orc.fused_brgemm(BF16, M, N, K, K, N, N, 0, 0, flags, 0, 5, 4, 1, a_o, 0, w_o, 0, ref, 0, b_o, 0, 1)
            else:
                orc.brgemm(BF16, M, N, K, K, N, N, 0, 0, flags, a_o, 0, w_o, 0, ref, 0, 1)
            flat = unpack_c(orc.bf16_to_f32(got), M, N, tm, tn).reshape(-1)
            check_close(orc.f32_to_bf16(flat), ref, BF16, "pass %d %s vnni%d [%s]" % (p, (M, N, K), vn, kernels[-1]), K=K)
        if callers == 1:  # (several callers: whether a replay completes - every member marked before something ends the group - is a matter of timing)
            assert "quads" not in kernels[0] and "quads" in kernels[1] and "quads" in kernels[2], kernels
            assert ("vnni4" in kernels[1]) == (vn == 4), kernels
        if fc:
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2]), "items pass and quad passes differ in bits"
    finally:
        rt.synchronize()
        rt.set_tile_queue(old_q)
        rt.set_async(old_async)
        rt.set_vnni_factor(old_v)
        orc.set_vnni_factor(old_o)


@pytest.mark.parametrize("case", ["odd_rows", "bias_not_by_column", "partial_replay"])
def test_groups_that_are_not_item_grids_stay_items(rt, case):
    """detect_quads refuses what is not a complete even grid of item rows x item columns: an odd number of item rows; a bias pointer
    that does not follow the item column; and a replay in which not every member arrives is launched from the gathered list as before
    (strict mode: tests/test_strict_gpu.py, its own process). Every pass against the oracle, no pass on the quad kernel."""
    tm = tn = tk = 64
    M, N, K = (960 if case == "odd_rows" else 1024), 1280, 128
    old_async, old_q = rt.set_async(True), rt.set_tile_queue(1)
    try:
        rng = np.random.default_rng(len(case))
        X = rng.uniform(-1, 1, (M, K)).astype(np.float32)
        W = (rng.uniform(-1, 1, (K, N)) / np.sqrt(K)).astype(np.float32)
        bias = rng.uniform(-1, 1, N).astype(np.float32)
        X, W, bias = (orc.bf16_to_f32(orc.f32_to_bf16(v.reshape(-1))).reshape(v.shape) for v in (X, W, bias))
        conv = orc.f32_to_bf16
        flags = VB | 4
        Wv = np.ascontiguousarray(W.reshape(K // 2, 2, N).transpose(0, 2, 1)).reshape(-1)
        a_o, w_o, b_o = conv(X.reshape(-1)), conv(Wv), conv(bias)
        ref = conv(np.zeros(M * N, np.float32))
        orc.fused_brgemm(BF16, M, N, K, K, N, N, 0, 0, flags, 0, 5, 4, 1, a_o, 0, w_o, 0, ref, 0, b_o, 0, 1)
        MB, NB, KB = M // tm, N // tn, K // tk
        # bias_not_by_column: the item of (row 3, column 5) reads a COPY of its bias piece at another address (same values)
        bias_dev = np.concatenate([bias, bias[5 * tn:6 * tn]])
        dA, dW, dB = dev(conv(pack_a(X, M, K, tm, tk))), dev(conv(pack_w(W, K, N, tk, tn, 2))), dev(conv(bias_dev))
        dC = dev(conv(np.zeros(M * N, np.float32)))
        h = rt.fused_brgemm_dispatch(BF16, tm, tn, tk, tk, tn, tn, tm * tk, tk * tn, flags, 0, 5, 4, 1)
        kernels = []
        for p in range(3):
            for i in range(MB):
                for j in range(NB):
                    if case == "partial_replay" and p >= 1 and i == MB - 1 and j >= NB - 4:
                        continue  # the last four tiles are not invoked in the replays: their outputs keep pass 0's (equal) values
                    off_d = N if (case == "bias_not_by_column" and i == 3 and j == 5) else j * tn
                    rt.fused_brgemm(BF16, h, dA, i * KB * tm * tk, dW, j * KB * tk * tn, dC, (i * NB + j) * tm * tn, dB, off_d, KB)
            rt.synchronize()
            kernels.append(rt.last_grouped_kernel())
            flat = unpack_c(orc.bf16_to_f32(host(dC, ref)), M, N, tm, tn).reshape(-1)
            check_close(orc.f32_to_bf16(flat), ref, BF16, "%s pass %d [%s]" % (case, p, kernels[-1]), K=K)
        assert not any("quads" in k for k in kernels), kernels
    finally:
        rt.synchronize()
        rt.set_tile_queue(old_q)
        rt.set_async(old_async)


@pytest.mark.parametrize("fc", [False, True], ids=["matmul_beta1", "fc_beta0_bias_relu"])
@pytest.mark.parametrize("vn", [2, 4], ids=["vnni2", "vnni4"])
@pytest.mark.parametrize("M,N,K,tiles", [(128, 768, 2304, (64, 48, 64)), (64, 480, 1024, (32, 48, 64)), (256, 336, 1152, (64, 48, 64))],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v))
def test_bf16_ragged_n_items_on_the_32x32_k2_tile(rt, M, N, K, tiles, vn, fc):
    """RAGGED n (csrc/brgemm_bf16_lw.hip skip_cols; brgemm_f32.hip "ragged n"): bf16 tile invokes whose n is 16 more than a multiple of
    32 - benchmarks/config/fc/128x768x2304.json:40-64 and matmul/128x768x2304.json: --tiles=64,48,64 - with a long reduction run on the
    32x32 + K2 loader-wave instance: ceil(n / 32) column tiles per item, the last one moved left to end at column n; it recomputes the
    16 columns it shares with its neighbour and stores only its own. Three passes (beta = 1 accumulates: a column stored twice would
    show), VNNI-2 and VNNI-4, 64x48 and 32x48 items (the tile queue takes items of at most 64 x 64), against the oracle; the kernel says so."""
    tm, tn, tk = tiles
    old_v = rt.set_vnni_factor(vn)
    old_o = orc.set_vnni_factor(vn)
    old_async, old_q = rt.set_async(True), rt.set_tile_queue(1)
    try:
        rng = np.random.default_rng(M + N + K + vn + 7)
        X = rng.uniform(-1, 1, (M, K)).astype(np.float32)
        W = (rng.uniform(-1, 1, (K, N)) / np.sqrt(K)).astype(np.float32)
        C0 = rng.uniform(-1, 1, (M, N)).astype(np.float32)
        bias = rng.uniform(-1, 1, N).astype(np.float32)
        X, W, C0, bias = (orc.bf16_to_f32(orc.f32_to_bf16(v.reshape(-1))).reshape(v.shape) for v in (X, W, C0, bias))
        conv = orc.f32_to_bf16
        flags = VB | (4 if fc else 0)
        Wv = np.ascontiguousarray(W.reshape(K // vn, vn, N).transpose(0, 2, 1)).reshape(-1)
        a_o, w_o, b_o = conv(X.reshape(-1)), conv(Wv), conv(bias)
        dA, dW, dB = dev(conv(pack_a(X, M, K, tm, tk))), dev(conv(pack_w(W, K, N, tk, tn, vn))), dev(conv(bias))
        dC = dev(conv(pack_c(C0, M, N, tm, tn)))
        disp = (BF16, tm, tn, tk, tk, tn, tn, tm * tk, tk * tn, flags)
        h = rt.fused_brgemm_dispatch(*disp, 0, 5, 4, 1) if fc else rt.brgemm_dispatch(*disp)
        MB, NB, KB = M // tm, N // tn, K // tk
        outs = []
        for p in range(3):
            start = host(dC, conv(C0.reshape(-1)))
            ref = orc.f32_to_bf16(unpack_c(orc.bf16_to_f32(start), M, N, tm, tn).reshape(-1))
            for i in range(MB):
                for j in range(NB):
                    if fc:
                        rt.fused_brgemm(BF16, h, dA, i * KB * tm * tk, dW, j * KB * tk * tn, dC, (i * NB + j) * tm * tn, dB, j * tn, KB)
                    else:
                        rt.brgemm(BF16, h, dA, i * KB * tm * tk, dW, j * KB * tk * tn, dC, (i * NB + j) * tm * tn, KB)
            rt.synchronize()
            kernel = rt.last_grouped_kernel()
            got = host(dC, ref)
            outs.append(got.copy())
            if fc:
                orc.fused_brgemm(BF16, M, N, K, K, N, N, 0, 0, flags, 0, 5, 4, 1, a_o, 0, w_o, 0, ref, 0, b_o, 0, 1)
            else:
                orc.brgemm(BF16, M, N, K, K, N, N, 0, 0, flags, a_o, 0, w_o, 0, ref, 0, 1)
            flat = unpack_c(orc.bf16_to_f32(got), M, N, tm, tn).reshape(-1)
            check_close(orc.f32_to_bf16(flat), ref, BF16, "pass %d %s tiles %s vnni%d [%s]" % (p, (M, N, K), tiles, vn, kernel), K=K)
            assert "ragged n" in kernel and ("vnni4" in kernel) == (vn == 4), kernel
        if fc:
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2])
    finally:
        rt.synchronize()
        rt.set_tile_queue(old_q)
        rt.set_async(old_async)
        rt.set_vnni_factor(old_v)
        orc.set_vnni_factor(old_o)

"""VNNI-4 bf16 B operands ([k/4][n][4]; `--vnni=4` in benchmarks/config/omp/mlir-bf16.json:68-100, layout
MLIRGen.cpp:657-664 / VNNIUtils.cpp:75-77). The factor is not on the wire - the reference asks libxsmm_cpuid_dot_pack_factor
(VNNIUtils.cpp:25-45) - so the runtime takes it from xsmm_hip_set_vnni_factor / TPP_HIP_VNNI_FACTOR at dispatch time.
Parity: the HIP result on the VNNI-4 operand against the oracle (a) run on the same packed operand with its own factor set to 4 and
(b) fed the FLAT operand (as test_c5 does for VNNI-2) - the two oracle runs are bit-identical by construction (same k order)."""
import importlib

import numpy as np
import pytest

from oracle import pyoracle as orc
from test_parity_gpu import BF16, VB, check_close, dev, host, rand

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("tpp-mlir_amd")


@pytest.fixture(scope="module")
def rt():
    r = pkg.get_runtime()
    assert r.device_count() >= 1, "no HIP device visible: the gpu tests need an MI355X"
    return r


@pytest.fixture()
def vnni4(rt):
    old = rt.set_vnni_factor(4)
    old_o = orc.set_vnni_factor(4)
    yield 4
    rt.set_vnni_factor(old)
    orc.set_vnni_factor(old_o)


def pack(Bf, K, n, ldb, v, br=1, sb=None):
    """flat [br][K][n] -> VNNI-v [br][K/v][ldb][v] (batch stride sb elements)"""
    sb = K * ldb if sb is None else sb
    out = np.zeros((br - 1) * sb + (K // v) * ldb * v, np.uint16)
    for b in range(br):
        blk = Bf[b * K * n:(b + 1) * K * n].reshape(K // v, v, n).transpose(0, 2, 1)  # [K/v][n][v]
        dst = out[b * sb:b * sb + (K // v) * ldb * v].reshape(K // v, ldb, v)
        dst[:, :n, :] = blk
    return out


def run_case(rt, m, n, k, br, force=None, bias=True, relu=True, beta0=True, ldb=None, lda=None, ldc=None, mode="device", seed=0,
             expect=None):
    rng = np.random.default_rng(seed)
    ldb, lda, ldc = ldb or n, lda or k * br, ldc or n
    sa, sb = k, k * ldb
    A = rand(rng, (m - 1) * lda + k * br + 8, BF16)
    Bf = rand(rng, br * k * n, BF16, -0.5, 0.5)
    B4 = pack(Bf, k, n, ldb, 4, br, sb)
    D = rand(rng, n, BF16)
    C0 = rand(rng, m * ldc, BF16)
    flags = (4 if beta0 else 0) | VB
    fused = bias or relu
    # oracle on the packed operand (factor 4) and on the flat one: bit-identical
    ref = C0.copy()
    ref_flat = C0.copy()
    if fused:
        orc.fused_brgemm(BF16, m, n, k, lda, ldb, ldc, sa, sb, flags, 0, 5 if relu else 0, 4 if bias else 0, 1 if bias else 0,
                         A, 0, B4, 0, ref, 0, D, 0, br)
        orc.fused_brgemm(BF16, m, n, k, lda, n, ldc, sa, k * n, flags & ~VB, 0, 5 if relu else 0, 4 if bias else 0, 1 if bias else 0,
                         A, 0, Bf, 0, ref_flat, 0, D, 0, br)
    else:
        orc.brgemm(BF16, m, n, k, lda, ldb, ldc, sa, sb, flags, A, 0, B4, 0, ref, 0, br)
        orc.brgemm(BF16, m, n, k, lda, n, ldc, sa, k * n, flags & ~VB, A, 0, Bf, 0, ref_flat, 0, br)
    assert np.array_equal(ref, ref_flat), "the oracle on the VNNI-4 operand differs from the oracle on the flat operand"
    if force is not None:
        rt.force_variant(force)
    try:
        if fused:
            h = rt.fused_brgemm_dispatch(BF16, m, n, k, lda, ldb, ldc, sa, sb, flags, 0, 5 if relu else 0, 4 if bias else 0, 1 if bias else 0)
        else:
            h = rt.brgemm_dispatch(BF16, m, n, k, lda, ldb, ldc, sa, sb, flags)
    finally:
        if force is not None:
            rt.force_variant(-1)
    name = rt.kernel_name(h)
    if expect:
        assert expect in name, name
    if mode == "device":
        dA, dB, dC, dD = dev(A), dev(B4), dev(C0), dev(D)
        if fused:
            rt.fused_brgemm(BF16, h, dA, 0, dB, 0, dC, 0, dD, 0, br)
        else:
            rt.brgemm(BF16, h, dA, 0, dB, 0, dC, 0, br)
        got = host(dC, C0)
    else:
        got = C0.copy()
        if fused:
            rt.fused_brgemm(BF16, h, A, 0, B4, 0, got, 0, D, 0, br)
        else:
            rt.brgemm(BF16, h, A, 0, B4, 0, got, 0, br)
    sel = np.concatenate([np.arange(r * ldc, r * ldc + n) for r in range(m)])
    check_close(got[sel], ref[sel], BF16, "vnni4[%s] m%d n%d k%d br%d" % (name, m, n, k, br))
    if ldc > n:  # the gap columns between the rows belong to somebody else
        gap = np.setdiff1d(np.arange(m * ldc), sel)
        assert np.array_equal(got[gap], C0[gap])
    return name, got


@pytest.mark.parametrize("variant,m,n", [(28, 64, 128), (29, 128, 128), (30, 128, 256), (31, 256, 256)])
@pytest.mark.parametrize("k,br", [(64, 1), (64, 5), (128, 3), (64, 16)])
def test_vnni4_loader_wave_tiles(rt, vnni4, variant, m, n, k, br):
    """every loader-wave tile with the VNNI-4 B image (two 8-byte fragment reads), chunk streams around the ring depths, padded
    leading dimensions, both accumulator starts"""
    for (beta0, bias, relu) in ((True, True, True), (False, False, False)):
        run_case(rt, m, n, k, br, force=variant, bias=bias, relu=relu, beta0=beta0, ldb=n + 2, lda=k * br + 8, ldc=n + 8,
                 seed=variant * 100 + k + br, expect="vnni4")


def test_vnni4_shapes_of_the_reference_benchmark_config(rt, vnni4):
    """benchmarks/config/omp/mlir-bf16.json:68-100: mlir-gen --batch=256 --layers=1024,1024,1024,1024 --tiles=32,32,32 --vnni=4,
    bias + relu: (a) the compiler-native tile invoke [32,32,32,32,32,32,1024,1024] br = 32 (k = 32: the generic kernel's element
    path), (b) the same layer as one whole-layer dispatch (256 x 1024, k = 64, br = 16) and (c) a bs = 4096 layer"""
    name, _ = run_case(rt, 32, 32, 32, 32, seed=1)
    assert "generic" in name or "grouped" in name, name
    name, _ = run_case(rt, 256, 1024, 64, 16, seed=2, expect="vnni4")
    name, _ = run_case(rt, 4096, 1024, 64, 16, seed=3, expect="vnni4<128x128>")


def test_vnni4_is_bit_identical_to_the_vnni2_kernel_on_the_same_matrix(rt):
    """the same logical B packed with factor 2 and with factor 4: same tile, same MFMA k order -> the same bits"""
    for (m, n, k, br, tile) in ((512, 1024, 64, 16, 0), (1024, 1024, 64, 16, 1), (2048, 2048, 128, 16, 3)):
        rng = np.random.default_rng(m)
        K = k * br
        A, Bf, D = rand(rng, m * K, BF16), rand(rng, K * n, BF16), rand(rng, n, BF16)
        outs = []
        for v in (2, 4):
            old = rt.set_vnni_factor(v)
            rt.force_variant((20 if v == 2 else 28) + tile)
            try:
                h = rt.fused_brgemm_dispatch(BF16, m, n, k, K, n, n, k, k * n, 4 | VB, 0, 5, 4, 1)
            finally:
                rt.force_variant(-1)
                rt.set_vnni_factor(old)
            Bp = pack(Bf, K, n, n, v)
            C = np.zeros(m * n, np.uint16)
            dC = dev(C)
            rt.fused_brgemm(BF16, h, dev(A), 0, dev(Bp), 0, dC, 0, dev(D), 0, br)
            outs.append((rt.kernel_name(h), host(dC, C)))
        assert "vnni4" in outs[1][0] and "vnni4" not in outs[0][0], [o[0] for o in outs]
        assert np.array_equal(outs[0][1], outs[1][1]), "%s differs from %s" % (outs[1][0], outs[0][0])


def test_vnni4_ragged_and_host_pointers(rt, vnni4):
    """ragged shapes (generic kernel) and the host-memory mirror (the B span follows the factor)"""
    run_case(rt, 6, 6, 8, 2, bias=False, relu=False, beta0=False, seed=4)
    run_case(rt, 13, 10, 12, 3, seed=5, ldb=12, ldc=16)
    run_case(rt, 64, 64, 64, 2, seed=6, mode="host", expect="vnni4")
    run_case(rt, 13, 10, 12, 3, seed=7, mode="host")


def test_vnni4_handles_do_not_alias_vnni2_handles(rt):
    """the factor is part of the descriptor: the same tuple dispatched under factor 2 and under factor 4 gives two handles"""
    old = rt.set_vnni_factor(2)
    try:
        h2 = rt.brgemm_dispatch(BF16, 64, 64, 64, 64, 64, 64, 4096, 4096, 4 | VB)
        rt.set_vnni_factor(4)
        h4 = rt.brgemm_dispatch(BF16, 64, 64, 64, 64, 64, 64, 4096, 4096, 4 | VB)
        rt.set_vnni_factor(2)
        assert h2 != h4 and rt.brgemm_dispatch(BF16, 64, 64, 64, 64, 64, 64, 4096, 4096, 4 | VB) == h2
        assert "vnni4" in rt.kernel_name(h4) and "vnni4" not in rt.kernel_name(h2)
    finally:
        rt.set_vnni_factor(old)


@pytest.mark.parametrize("queue", [0, 1])
def test_vnni4_compiler_native_tile_invokes_of_a_layer(rt, vnni4, queue):
    """the call pattern of `mlir-gen --tiles=32,32,32 --vnni=4` (benchmarks/config/omp/mlir-bf16.json:68-100): packed A [MB][KB][32][32],
    W [NB][KB][32/4][32][4], C [MB][NB][32][32], ONE fused dispatch (32,32,32,32,32,32,1024,1024) and MB x NB invokes with batch KB -
    unqueued (one launch per invoke: the bf16 MFMA path of the generic kernel on the VNNI-4 image) and through the tile queue (one
    grouped launch per layer); two chained layers against the oracle's replay of the same invokes"""
    MB, NB, KB = 4, 8, 8
    rng = np.random.default_rng(11 + queue)
    X = rand(rng, MB * KB * 1024, BF16)
    Ws = [rand(rng, NB * KB * 1024, BF16, -0.3, 0.3) for _ in range(2)]
    bs = [rand(rng, NB * 32, BF16) for _ in range(2)]
    disp = (BF16, 32, 32, 32, 32, 32, 32, 1024, 1024, 4 | VB, 0, 5, 4, 1)
    refs, cur = [], X
    for l in range(2):
        out = np.zeros(MB * NB * 1024, np.uint16)
        for i in range(MB):
            for j in range(NB):
                orc.fused_brgemm(*disp, cur, i * KB * 1024, Ws[l], j * KB * 1024, out, (i * NB + j) * 1024, bs[l], j * 32, KB)
        refs.append(out)
        cur = out
    h = rt.fused_brgemm_dispatch(*disp)
    prev_async, prev_q = rt.set_async(True), rt.set_tile_queue(queue)
    try:
        dX, dW, db = dev(X), [dev(w) for w in Ws], [dev(b) for b in bs]
        dA = [dev(np.zeros(MB * NB * 1024, np.uint16)) for _ in range(2)]
        for rep in range(3):  # (the third pass replays the recorded groups)
            cur = dX
            for l in range(2):
                for i in range(MB):
                    for j in range(NB):
                        rt.fused_brgemm(BF16, h, cur, i * KB * 1024, dW[l], j * KB * 1024, dA[l], (i * NB + j) * 1024, db[l], j * 32, KB)
                cur = dA[l]
            rt.synchronize()
        for l in range(2):
            prev = X if l == 0 else host(dA[0], X)
            one = np.zeros(MB * NB * 1024, np.uint16)  # layer l from the GPU's own layer l-1: one layer's error at a time
            for i in range(MB):
                for j in range(NB):
                    orc.fused_brgemm(*disp, prev, i * KB * 1024, Ws[l], j * KB * 1024, one, (i * NB + j) * 1024, bs[l], j * 32, KB)
            check_close(host(dA[l], X), one, BF16, "vnni4 tile invokes layer %d queue %d" % (l, queue))
    finally:
        rt.synchronize()
        rt.set_tile_queue(prev_q)
        rt.set_async(prev_async)


@pytest.mark.parametrize("kind", ["vnni4", "flat"])
@pytest.mark.parametrize("m,tile", [(512, 0), (1024, 1), (4096, 3)])
def test_chain_launch_on_vnni4_and_flat_b_layers(rt, kind, m, tile):
    """xsmm_hip_fused_brgemm_chain_invoke on layers whose B operands are VNNI-4 / flat (round 4: the B image is a template parameter
    of the chain launch too): ONE launch, bit-identical to the same layers invoked one by one on the same tile, and equal to the
    VNNI-2 chain on the same matrices (same tile, same MFMA k order)"""
    import torch
    N, L = 1024, 3
    rng = np.random.default_rng(m + (7 if kind == "flat" else 0))
    X = rand(rng, m * N, BF16)
    Wf = [rand(rng, N * N, BF16, -0.06, 0.06) for _ in range(L)]
    bs = [rand(rng, N, BF16, -0.5, 0.5) for _ in range(L)]
    outs = {}
    for k_ in (kind, "vnni2"):
        v = {"vnni4": 4, "vnni2": 2, "flat": 0}[k_]
        old = rt.set_vnni_factor(v if v else 2)
        rt.force_variant({"vnni4": 28, "vnni2": 20, "flat": 24}[k_] + tile)
        try:
            h = rt.fused_brgemm_dispatch(BF16, m, N, 64, N, N, N, 64, 64 * N, 4 | (VB if v else 0), 0, 5, 4, 1)
        finally:
            rt.force_variant(-1)
            rt.set_vnni_factor(old)
        W = [dev(pack(w, N, N, N, v) if v else w) for w in Wf]
        db = [dev(b) for b in bs]
        dX = dev(X)
        acts_f = [dev(np.full(m * N, 0x7fc0, np.uint16)) for _ in range(L)]
        acts_s = [dev(np.full(m * N, 0x7fc0, np.uint16)) for _ in range(L)]
        was = rt.set_async(True)
        try:
            calls = lambda acts: [(h, dX if l == 0 else acts[l - 1], 0, W[l], 0, acts[l], 0, db[l], 0, N // 64) for l in range(L)]  # noqa: E731
            fused = rt.fused_brgemm_chain(BF16, calls(acts_f))
            for c in calls(acts_s):
                rt.fused_brgemm(BF16, *c)
            rt.synchronize()
        finally:
            rt.set_async(was)
        assert fused, "%s layers did not run as one chain launch (%s)" % (k_, rt.kernel_name(h))
        for l in range(L):
            assert torch.equal(acts_f[l], acts_s[l]), "%s chain layer %d differs from the separate launches" % (k_, l)
        outs[k_] = host(acts_f[L - 1], X)
        outs[k_ + "_layer0"] = host(acts_f[0], X)[:32 * N]
    assert np.array_equal(outs[kind], outs["vnni2"]), "%s chain differs from the VNNI-2 chain on the same matrices" % kind
    # layer 0 against the oracle on a row sample (flat operand)
    ref = np.zeros(32 * N, np.uint16)
    orc.fused_brgemm(BF16, 32, N, 64, N, N, N, 64, 64 * N, 4, 0, 5, 4, 1, X, 0, Wf[0], 0, ref, 0, bs[0], 0, N // 64)
    check_close(outs[kind + "_layer0"], ref, BF16, "%s chain layer 0 (rows 0-31)" % kind)


def test_vnni4_groups_of_64x64_tiles_run_on_the_64x64_family(rt):
    """round 5: `mlir-gen --tiles=64,64,64 --vnni=4` (the *_dp4_* rows of benchmarks/config/matmul|fc/*.json) as tile invokes - a group
    that fills the chip (here 16 x 16 tiles of 64x64, br = 3) runs on the register-staged 64x64 bf16 kernel with the VNNI-4 source image
    (four 16-byte loads per 4-column x 8-k piece, the in-register transpose selects differently): bit-identical to the same matrix as a
    VNNI-2 operand through the same family, and against the oracle; with bias + relu and with C += ..."""
    MB, NB, KB, t = 16, 16, 3, 64
    rng = np.random.default_rng(4)
    X = rand(rng, MB * KB * t * t, BF16)
    Wf = rand(rng, NB * KB * t * t, BF16, -0.3, 0.3).reshape(NB, KB, t, t)  # [NB][KB][k][n] flat blocks
    W2 = np.ascontiguousarray(Wf.reshape(NB, KB, t // 2, 2, t).transpose(0, 1, 2, 4, 3)).reshape(-1)
    W4 = np.ascontiguousarray(Wf.reshape(NB, KB, t // 4, 4, t).transpose(0, 1, 2, 4, 3)).reshape(-1)
    bias = rand(rng, NB * t, BF16)
    C0 = rand(rng, MB * NB * t * t, BF16)
    outs = {}
    for flags, fused in ((4 | VB, (0, 5, 4, 1)), (VB, (0, 0, 0, 0))):
        for v, W in ((2, W2), (4, W4)):
            old, old_o = rt.set_vnni_factor(v), orc.set_vnni_factor(v)
            try:
                disp = (BF16, t, t, t, t, t, t, t * t, t * t, flags) + fused
                h = rt.fused_brgemm_dispatch(*disp)
                ref = C0.copy()
                for i in range(0, MB, 5):  # (the oracle on a sample of tile rows: rows are independent)
                    for j in range(NB):
                        orc.fused_brgemm(*disp, X, i * KB * t * t, W, j * KB * t * t, ref, (i * NB + j) * t * t, bias, j * t, KB)
            finally:
                rt.set_vnni_factor(old)
                orc.set_vnni_factor(old_o)
            prev_async, prev_q = rt.set_async(True), rt.set_tile_queue(1)
            try:
                dX, dW, db, dC = dev(X), dev(W), dev(bias), dev(C0)
                for i in range(MB):
                    for j in range(NB):
                        rt.fused_brgemm(BF16, h, dX, i * KB * t * t, dW, j * KB * t * t, dC, (i * NB + j) * t * t, db, j * t, KB)
                rt.synchronize()
                ran = rt.last_grouped_kernel()
            finally:
                rt.set_tile_queue(prev_q)
                rt.set_async(prev_async)
            # (round 6: such a group runs on the grouped loader-wave tile; TPP_HIP_BF16_LW_GROUPED=0 brings the 64x64 family back)
            assert ("lw_vnni4<64x64> grouped" if v == 4 else "lw<64x64> grouped") in ran or ("fast_vnni4<64x64> grouped" if v == 4 else "fast<64x64> grouped") in ran, ran
            got = host(dC, C0)
            sel = np.concatenate([np.arange((i * NB) * t * t, (i * NB + NB) * t * t) for i in range(0, MB, 5)])
            check_close(got[sel], ref[sel], BF16, "64x64 tiles vnni %d flags %d" % (v, flags))
            outs[(flags, v)] = got
        assert np.array_equal(outs[(flags, 2)], outs[(flags, 4)]), "VNNI-4 and VNNI-2 images of one matrix must give the same bits"


@pytest.mark.parametrize("fc", [False, True], ids=["matmul_beta1", "fc_beta0_bias_relu"])
@pytest.mark.parametrize("M,N,K", [(128, 1024, 2048), (256, 768, 1024)], ids=lambda v: str(v))
def test_vnni4_skinny_long_reduction_groups_on_the_32x32_k2_tile(rt, vnni4, M, N, K, fc):
    """Round 6: tile invokes of a VNNI-4 layer with a skinny output and a long reduction (benchmarks/config/*: the dp4 rows of
    128x1024x4096, 256x768x3072 ...) run on the 32x32 + K2 instance of the grouped loader-wave kernel like their VNNI-2 twins (four
    workgroups per 64x64 item; the VNNI-4 image of a 32-column tile = 256-byte k-group rows, four per DMA instruction). Recorded pass
    and two replays against the oracle (factor 4); the kernel says so."""
    from test_refbench_shapes_gpu import pack_a, pack_c, pack_w, unpack_c
    tm = tn = tk = 64
    old_async, old_q = rt.set_async(True), rt.set_tile_queue(1)
    try:
        rng = np.random.default_rng(M + N + K)
        X = rng.uniform(-1, 1, (M, K)).astype(np.float32)
        W = (rng.uniform(-1, 1, (K, N)) / np.sqrt(K)).astype(np.float32)
        C0 = rng.uniform(-1, 1, (M, N)).astype(np.float32)
        bias = rng.uniform(-1, 1, N).astype(np.float32)
        X, W, C0, bias = (orc.bf16_to_f32(orc.f32_to_bf16(v.reshape(-1))).reshape(v.shape) for v in (X, W, C0, bias))
        conv = orc.f32_to_bf16
        flags = VB | (4 if fc else 0)
        Wv = np.ascontiguousarray(W.reshape(K // 4, 4, N).transpose(0, 2, 1)).reshape(-1)
        a_o, w_o, b_o = conv(X.reshape(-1)), conv(Wv), conv(bias)
        dA, dW, dB = dev(conv(pack_a(X, M, K, tm, tk))), dev(conv(pack_w(W, K, N, tk, tn, 4))), dev(conv(bias))
        dC = dev(conv(pack_c(C0, M, N, tm, tn)))
        disp = (BF16, tm, tn, tk, tk, tn, tn, tm * tk, tk * tn, flags)
        h = rt.fused_brgemm_dispatch(*disp, 0, 5, 4, 1) if fc else rt.brgemm_dispatch(*disp)
        MB, NB, KB = M // tm, N // tn, K // tk
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
            if fc:
                orc.fused_brgemm(BF16, M, N, K, K, N, N, 0, 0, flags, 0, 5, 4, 1, a_o, 0, w_o, 0, ref, 0, b_o, 0, 1)
            else:
                orc.brgemm(BF16, M, N, K, K, N, N, 0, 0, flags, a_o, 0, w_o, 0, ref, 0, 1)
            flat = unpack_c(orc.bf16_to_f32(host(dC, ref)), M, N, tm, tn).reshape(-1)
            check_close(orc.f32_to_bf16(flat), ref, BF16, "pass %d %s vnni4 [%s]" % (p, (M, N, K), kernel), K=K)
            assert "vnni4<32x32,k2> grouped" in kernel, kernel
    finally:
        rt.synchronize()
        rt.set_tile_queue(old_q)
        rt.set_async(old_async)


@pytest.mark.parametrize("m,n,k,br", [(128, 1024, 64, 64), (256, 768, 64, 16), (128, 1024, 64, 15), (256, 768, 128, 8)])
def test_vnni4_whole_layer_skinny_long_reduction_on_the_32x32_k2_tile(rt, vnni4, m, n, k, br):
    """Round 6: a whole-layer VNNI-4 call with at most one 32x32 tile per CU and K = k br >= 1024 is refined at invoke time from the
    32x64 + K2 tile to the 32x32 + K2 instance (as VNNI-2 operands are): against the oracle (packed and flat operand), both
    accumulator starts, an odd chunk count; K < 1024 stays on the planned tile."""
    for (beta0, bias, relu) in ((True, True, True), (False, False, False)):
        run_case(rt, m, n, k, br, bias=bias, relu=relu, beta0=beta0, seed=m + n + k + br, expect="vnni4")
        refined = rt.last_refined_kernel()
        if k * br >= 1024:
            assert "vnni4<32x32,k2> (long reduction)" in refined, refined
        else:
            assert "32x32,k2" not in refined, refined

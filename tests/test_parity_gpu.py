"""Parity of the HIP path (through the C-ABI of libtpp_xsmm_runner_utils.so) with the
CPU oracle, on a real MI355X:
  * every golden fixture harvested from the reference's lit tests, through host
    pointers (mirror path) and device pointers (zero-copy path);
  * seeded random cases over the shapes / strides / flags the compiler can emit,
    including ragged sizes (3, 5, 13, 33 ...), empty batches and overlapping batches;
  * the BASELINE.json configurations at full size (rows sampled where the oracle
    would be slow: output rows are independent) and size-independent properties.
Bars: f32 max|gpu - ref| <= 1e-5 * max(1, max|ref|) (north_star) AND, element-wise,
|gpu - ref| <= 1e-5 * |ref| + K * eps * (|C| + sum_k |a||b| + |bias|) (the relative bar
with the a-priori f32 dot-product floor - the two sides sum in different orders); bf16 results within
one bf16 ulp of the oracle's (reference convention fpcmp -r 0.01 is far looser);
identity / zero / transpose / VNNI-2 pack bit-exact.
"""
import importlib
import os
import threading

import numpy as np
import pytest

import fixture_runner as fr
from abi_backend import AbiBackend
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("tpp-mlir_amd")
F32, BF16 = 1, 2
VB = 2048  # wire flag of a VNNI-2 B operand


@pytest.fixture(scope="module")
def rt():
    r = pkg.get_runtime()
    assert r.device_count() >= 1, "no HIP device visible: the gpu tests need an MI355X"
    return r


def dev(arr):
    import torch
    src = arr.view(np.int16) if arr.dtype == np.uint16 else arr
    return torch.from_numpy(src.copy()).cuda()


def host(t, like):
    a = t.cpu().numpy()
    return a.view(np.uint16) if like.dtype == np.uint16 else a


def rand(rng, n, dt, lo=-1.0, hi=1.0):
    v = rng.uniform(lo, hi, size=n).astype(np.float32)
    return v if dt == F32 else orc.f32_to_bf16(v)


def as_f32(a):
    return a if a.dtype == np.float32 else orc.bf16_to_f32(a)


REL_STATS = {"max_rel": 0.0, "cases": 0}  # element-wise figure over the whole session (printed at the end)


TRUTH_STATS = {"hip_vs_f64": 0.0, "oracle_vs_f64": 0.0, "cases": 0}  # worst normwise errors against an fp64 truth (printed at the end)


def check_close(got, ref, dt, what, mag=None, K=0, truth=None):
    """mag: |C| + sum |a||b| + |bias| per element (f32 only): enables the element-wise criterion.
    truth: the same result computed in fp64 (f32 only): the HIP result must be as close to it as the oracle is - the two f32
    results differ from each other by their summation orders, and neither may be privileged: normwise |hip - f64| <= 2 x
    |oracle - f64| + 2^-22 of the result scale, and the same for the worst relative error over the elements that are not
    cancelled (|truth| >= 1 % of its maximum)"""
    g, r = as_f32(got).astype(np.float64), as_f32(ref).astype(np.float64)
    assert np.isfinite(g).all(), what + ": non-finite values"
    diff = np.abs(g - r)
    if truth is not None and dt == F32 and r.size:
        t = np.asarray(truth, dtype=np.float64)
        scale = max(1.0, float(np.abs(t).max()))
        e_hip, e_orc = float(np.abs(g - t).max()) / scale, float(np.abs(r - t).max()) / scale
        TRUTH_STATS["hip_vs_f64"] = max(TRUTH_STATS["hip_vs_f64"], e_hip)
        TRUTH_STATS["oracle_vs_f64"] = max(TRUTH_STATS["oracle_vs_f64"], e_orc)
        TRUTH_STATS["cases"] += 1
        assert e_hip <= 2.0 * e_orc + 2.0 ** -22, "%s: HIP is %.3g from the fp64 truth (normwise), the oracle %.3g" % (what, e_hip, e_orc)
        big = np.abs(t) >= 1e-2 * np.abs(t).max()
        rel_hip = float((np.abs(g - t)[big] / np.abs(t[big])).max())
        rel_orc = float((np.abs(r - t)[big] / np.abs(t[big])).max())
        assert rel_hip <= 2.0 * rel_orc + 1e-6, "%s: max relative error against the fp64 truth over the non-cancelled elements: HIP %.3g, oracle %.3g" % (
            what, rel_hip, rel_orc)
    if dt == F32:
        tol = 1e-5 * max(1.0, float(np.abs(r).max()) if r.size else 1.0)
        bad = diff > tol
        if mag is not None and r.size:
            floor = (K + 2) * 2.0 ** -24 * np.asarray(mag, dtype=np.float64)
            bad_rel = diff > 1e-5 * np.abs(r) + floor
            REL_STATS["max_rel"] = max(REL_STATS["max_rel"], float((diff / (1e-5 * np.abs(r) + floor + 1e-300)).max()))
            REL_STATS["cases"] += 1
            assert not bad_rel.any(), "%s: %d/%d elements outside the element-wise bar, worst |d| %g at |ref| %g" % (
                what, int(bad_rel.sum()), bad_rel.size, float(diff[bad_rel].max()), float(np.abs(r)[bad_rel][0]))
    else:
        # one bf16 ulp of the reference value (a rounding flip: 2^-7 relative covers it) plus the
        # f32-accumulation floor of the f32 bar: where products cancel, |ref| is far below the
        # magnitude of the summands and only the absolute f32 error of the sum is meaningful
        tol = np.abs(r) * 2.0 ** -7 + 1e-5 * max(1.0, float(np.abs(r).max()) if r.size else 1.0)
        bad = diff > tol
    assert not bad.any(), "%s: %d/%d mismatches, max abs diff %g (max |ref| %g)" % (
        what, int(bad.sum()), bad.size, float(diff.max()), float(np.abs(r).max()))


# ---------------------------------------------------------------- golden fixtures
@pytest.mark.parametrize("mode", ["host", "device"])
@pytest.mark.parametrize("path", fr.fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_golden_fixture_through_abi(rt, path, mode):
    fr.run_fixture(path, AbiBackend(mode))


# ---------------------------------------------------------------- GEMM family
def gemm_case(rt, dt, m, n, k, br, lda=None, ldb=None, ldc=None, sa=None, sb=None, beta0=False, bias=False,
              relu=False, vnni=False, fused=None, offs=(0, 0, 0, 0), seed=0, mode="device", force=None,
              row_blocks=None):
    """row_blocks: [(first_row, rows)] to verify when the full oracle would be slow - output
    rows are independent, so the oracle is run on those row blocks only."""
    rng = np.random.default_rng(seed)
    lda = lda or max(k, 1)
    ldb = ldb or max(n, 1)
    ldc = ldc or max(n, 1)
    kp = (k + 1) // 2
    bmat = (kp * 2 * ldb) if vnni else k * ldb
    sa = m * lda if sa is None else sa
    sb = bmat if sb is None else sb
    fused = (bias or relu) if fused is None else fused
    na = offs[0] + max(br - 1, 0) * sa + m * lda + 8
    nb = offs[1] + max(br - 1, 0) * sb + bmat + 2 * ldb + 8
    nc = offs[2] + m * ldc + 8
    A, B, C, D = rand(rng, na, dt), rand(rng, nb, dt), rand(rng, nc, dt), rand(rng, offs[3] + n + 8, dt)
    flags = (4 if beta0 else 0) | (VB if vnni else 0)
    Cref = C.copy()
    for (r0, rr) in (row_blocks or [(0, m)]):
        if fused:
            orc.fused_brgemm(dt, rr, n, k, lda, ldb, ldc, sa, sb, flags, 0, 5 if relu else 0, 4 if bias else 0,
                             1 if bias else 0, A, offs[0] + r0 * lda, B, offs[1], Cref, offs[2] + r0 * ldc, D,
                             offs[3], br)
        else:
            orc.brgemm(dt, rr, n, k, lda, ldb, ldc, sa, sb, flags, A, offs[0] + r0 * lda, B, offs[1], Cref,
                       offs[2] + r0 * ldc, br)
    Cmag = None
    if dt == F32 and m * n * k * max(br, 1) <= 2 ** 28:  # |C| + sum |a||b| + |bias| for the element-wise bar
        Cmag = np.abs(C)
        for (r0, rr) in (row_blocks or [(0, m)]):
            orc.fused_brgemm(dt, rr, n, k, lda, ldb, ldc, sa, sb, flags, 0, 0, 4 if bias else 0, 1 if bias else 0,
                             np.abs(A), offs[0] + r0 * lda, np.abs(B), offs[1], Cmag, offs[2] + r0 * ldc, np.abs(D), offs[3], br)
    if force is not None:
        rt.force_variant(force)
    try:
        if fused:
            h = rt.fused_brgemm_dispatch(dt, m, n, k, lda, ldb, ldc, sa, sb, flags, 0, 5 if relu else 0,
                                         4 if bias else 0, 1 if bias else 0)
        else:
            h = rt.brgemm_dispatch(dt, m, n, k, lda, ldb, ldc, sa, sb, flags)
    finally:
        if force is not None:
            rt.force_variant(-1)
    name = rt.kernel_name(h)
    if mode == "device":
        dA, dB, dC, dD = dev(A), dev(B), dev(C), dev(D)
        if fused:
            rt.fused_brgemm(dt, h, dA, offs[0], dB, offs[1], dC, offs[2], dD, offs[3], br)
        else:
            rt.brgemm(dt, h, dA, offs[0], dB, offs[1], dC, offs[2], br)
        got = host(dC, C)
    else:
        got = C.copy()
        if fused:
            rt.fused_brgemm(dt, h, A, offs[0], B, offs[1], got, offs[2], D, offs[3], br)
        else:
            rt.brgemm(dt, h, A, offs[0], B, offs[1], got, offs[2], br)
    what = "brgemm[%s] dt%d m%d n%d k%d br%d lda%d ldb%d ldc%d sa%d sb%d beta0=%d bias=%d relu=%d" % (
        name, dt, m, n, k, br, lda, ldb, ldc, sa, sb, beta0, bias, relu)
    if row_blocks:
        sel = np.concatenate([np.arange(offs[2] + r * ldc, offs[2] + r * ldc + n)
                              for (r0, rr) in row_blocks for r in range(r0, r0 + rr)])
        check_close(got[sel], Cref[sel], dt, what, None if Cmag is None else Cmag[sel], k * br)
    else:
        win = np.concatenate([np.arange(offs[2] + i * ldc, offs[2] + i * ldc + n) for i in range(m)]) if m and n else np.arange(0)
        check_close(got[win], Cref[win], dt, what, None if Cmag is None else Cmag[win], k * br)
    # bytes outside the m x n window (ldc padding, guard elements) must be untouched
    mask = np.ones(C.size, dtype=bool)
    for i in range(m):
        mask[offs[2] + i * ldc: offs[2] + i * ldc + n] = False
    assert np.array_equal(got[mask], C[mask]), what + ": wrote outside the output window"
    return name


F32_FAST = [
    # (m, n, k, br, kwargs) - every fast tile variant, both epilogue flavours
    (1024, 1024, 64, 16, dict(lda=1024, ldb=1024, sa=64, sb=65536)),                # C2: 64x64 tiles
    (1024, 1024, 64, 16, dict(lda=1024, ldb=1024, sa=64, sb=65536, beta0=True)),
    (512, 1024, 64, 16, dict(lda=1024, ldb=1024, sa=64, sb=65536, beta0=True, bias=True, relu=True)),  # C3
    (256, 1024, 64, 4, dict(lda=256, ldb=1024, sa=64, sb=65536, bias=True)),        # 32x32 k-split 4
    (64, 64, 64, 1, dict()),
    (64, 64, 128, 3, dict(relu=True)),
    (128, 192, 64, 2, dict(ldc=200, offs=(4, 8, 3, 1))),
    (4096, 1024, 64, 5, dict(lda=320, ldb=1024, sa=64, sb=65536, beta0=True)),      # 128x64 tiles
    (64, 64, 64, 0, dict()),                                                       # empty batch: C unchanged
    (64, 64, 64, 0, dict(beta0=True, bias=True)),                                   # empty batch: C = bias
]


@pytest.mark.parametrize("case", F32_FAST, ids=lambda c: "m%d_n%d_k%d_br%d_%s" % (
    c[0], c[1], c[2], c[3], "_".join(k for k in sorted(c[4]) if c[4][k] is True)))
def test_brgemm_f32_fast_variants(rt, case):
    m, n, k, br, kw = case
    name = gemm_case(rt, F32, m, n, k, br, seed=m + n + k + br, **kw)
    assert "fast" in name or "lw16" in name, name  # (round 5: outputs of at most one 32x16 tile per CU run on the half-width tiles)


@pytest.mark.parametrize("variant,m,n", [(0, 128, 128), (1, 128, 96), (2, 96, 96), (3, 256, 128), (4, 128, 128),
                                         (5, 128, 128), (6, 128, 192), (7, 128, 96), (9, 96, 96), (10, 256, 128)])
def test_brgemm_f32_forced_tile_variants(rt, variant, m, n):
    gemm_case(rt, F32, m, n, 64, 4, sa=64, lda=256, sb=64 * n, beta0=False, bias=True, relu=True,
              seed=variant, force=variant)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 9, 10])
@pytest.mark.parametrize("k,br", [(64, 0), (64, 1), (64, 2), (64, 3), (64, 4), (64, 5), (64, 7), (128, 3), (192, 2), (64, 16)])
def test_brgemm_f32_chunk_stream_lengths(rt, variant, k, br):
    """every ring position of the uniform chunk loops (1 .. 16 chunks, chunk streams that wrap inside a batch
    element), for every fast f32 tile family incl. the loader-wave kernels, both accumulator starts"""
    m, n = (256, 128) if variant in (3, 10) else (128, 128)
    for beta0 in (True, False):
        name = gemm_case(rt, F32, m, n, k, br, lda=k * max(br, 1) + 8, ldb=n + 4, ldc=n + 4, sa=k, sb=k * (n + 4),
                         beta0=beta0, bias=not beta0, relu=beta0, seed=variant * 100 + k + br, force=variant,
                         offs=(4, 8, 4, 1))
        assert "fast" in name, name


RAGGED = [(3, 3, 4, 2), (5, 13, 10, 3), (33, 65, 70, 2), (6, 6, 6, 2), (1, 1, 1, 1), (10, 10, 10, 1),
          (32, 32, 32, 32), (100, 40, 17, 4), (64, 64, 60, 2), (31, 200, 64, 1)]


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("shape", RAGGED, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("mode", ["device", "host"])
def test_brgemm_generic_ragged(rt, dt, shape, mode):
    m, n, k, br = shape
    gemm_case(rt, dt, m, n, k, br, lda=k + 3, ldb=n + 1, ldc=n + 2, offs=(1, 2, 3, 1), seed=sum(shape),
              bias=(m % 2 == 1), relu=(n % 2 == 1), beta0=(k % 2 == 0), mode=mode,
              vnni=(dt == BF16 and k % 2 == 0))


@pytest.mark.parametrize("dt,ld", [(F32, (1 << 22) - 4), (F32, (1 << 22) + 4), (BF16, (1 << 21) - 8), (BF16, (1 << 22) + 8)],
                         ids=["f32_max_fast_ld", "f32_ld_beyond_fast", "bf16_large_ld", "bf16_ld_beyond_fast"])
def test_brgemm_huge_leading_dimensions(rt, dt, ld):
    """leading dimensions at the edge of the fast kernels' 32-bit lane offsets (ld < 2^22) and beyond it (generic
    kernel, 64-bit addressing): a 64 x 64 x 64 product inside ~1 GiB operands"""
    import torch
    m = n = k = 64
    br = 2
    rng = np.random.default_rng(ld & 0xffff)
    vnni = dt == BF16
    es = np.float32 if dt == F32 else np.uint16
    A = np.zeros((m - 1) * ld + k * br + 8, dtype=es)
    B = np.zeros((k * br) * ld // (1 if not vnni else 1) + 2 * ld + 8, dtype=es)
    C = np.zeros((m - 1) * ld + n + 8, dtype=es)
    for i in range(m):
        A[i * ld: i * ld + k * br] = rand(rng, k * br, dt)
        C[i * ld: i * ld + n] = rand(rng, n, dt)
    rows_b = k * br if not vnni else (k * br) // 2
    for r in range(rows_b):
        w = n if not vnni else 2 * n
        pitch = ld if not vnni else 2 * ld
        B[r * pitch: r * pitch + w] = rand(rng, w, dt)
    ref = C.copy()
    sb = k * ld  # elements between batch elements of B (flat: k rows; VNNI: k/2 pair-rows of 2*ld)
    flags = VB if vnni else 0
    orc.brgemm(dt, m, n, k, ld, ld, ld, k, sb, flags, A, 0, B, 0, ref, 0, br)
    h = rt.brgemm_dispatch(dt, m, n, k, ld, ld, ld, k, sb, flags)
    dA, dB, dC = dev(A), dev(B), dev(C)
    rt.brgemm(dt, h, dA, 0, dB, 0, dC, 0, br)
    got = host(dC, C)
    del dA, dB, dC
    torch.cuda.empty_cache()
    rows = np.concatenate([np.arange(i * ld, i * ld + n) for i in range(m)])
    check_close(got[rows], ref[rows], dt, "huge ld %d [%s]" % (ld, rt.kernel_name(h)))
    mask = np.ones(C.size, dtype=bool)
    mask[rows] = False
    assert np.array_equal(got[mask], C[mask]), "wrote outside the output window"
    if ld >= (1 << 22):  # kernels with 32-bit lane offsets must not have been chosen
        assert "fast" not in rt.kernel_name(h) and "dma" not in rt.kernel_name(h), rt.kernel_name(h)


@pytest.mark.parametrize("dt,m,lda", [(F32, 66000, 8192), (BF16, 1100, 1 << 20)], ids=["f32", "bf16"])
def test_brgemm_generic_vector_path_a_beyond_2gib(rt, dt, m, lda):
    """(m - 1) * lda * esize >= 2^31 on the generic kernel's 16-byte-load path (n a multiple of 4 only, k of 32 only):
    the per-lane offset is relative to the TILE and the 64-bit descriptor base carries m0 * lda, so rows past 2 GiB of A
    are addressed correctly (an absolute 32-bit row offset wrapped / read zeros there)"""
    import torch
    n, k, br = 40, 32, 2
    vnni = dt == BF16
    es = 4 if dt == F32 else 2
    assert (m - 1) * lda * es >= 2 ** 31
    rng = np.random.default_rng(m)
    Ac = rand(rng, m * k * br, dt).reshape(m, k * br)  # compact copy for the oracle (lda = k * br)
    B = rand(rng, br * k * n + 8, dt)
    C0 = rand(rng, m * n, dt)
    ref = C0.copy()
    flags = (VB if vnni else 0)
    orc.brgemm(dt, m, n, k, k * br, n, n, k, k * n, flags, Ac.reshape(-1), 0, B, 0, ref, 0, br)
    tdt = torch.float32 if dt == F32 else torch.int16
    dA = torch.zeros(((m - 1) * lda + k * br + 8,), dtype=tdt, device="cuda")
    src = torch.from_numpy((Ac if dt == F32 else Ac.view(np.int16)).copy()).cuda()
    dA.as_strided((m, k * br), (lda, 1)).copy_(src)
    dB, dC = dev(B), dev(C0)
    rt.force_variant(8)
    try:
        h = rt.brgemm_dispatch(dt, m, n, k, lda, n, n, k, k * n, flags)
    finally:
        rt.force_variant(-1)
    rt.brgemm(dt, h, dA, 0, dB, 0, dC, 0, br)
    got = host(dC, C0)
    del dA, src
    torch.cuda.empty_cache()
    assert "grouped" in rt.kernel_name(h), rt.kernel_name(h)
    check_close(got, ref, dt, "A beyond 2 GiB [%s]" % rt.kernel_name(h))


def test_brgemm_unaligned_pointers_fall_back(rt):
    # fast shape, but A/B offsets break 16-byte alignment: the runtime must pick the generic kernel
    gemm_case(rt, F32, 64, 64, 64, 2, offs=(1, 3, 0, 0), seed=5)


def test_brgemm_overlapping_batches(rt):
    # stride_a smaller than a matrix (xsmm-strided-brgemm.mlir:34-44; xsmm-ternary-bf16 uses stride 8 on 4x4)
    gemm_case(rt, F32, 64, 64, 64, 6, lda=512, sa=64, sb=64, ldb=128, seed=9)
    gemm_case(rt, BF16, 4, 4, 4, 64, sa=8, sb=8, vnni=True, seed=10)


BF16_CASES = [
    (4096, 1024, 64, 16, dict(lda=1024, ldb=1024, sa=64, sb=65536, beta0=True, bias=True, relu=True)),  # C4 layer
    (256, 256, 64, 4, dict(beta0=True)),
    (128, 128, 128, 2, dict()),                                  # beta = 1 reads bf16 C
    (64, 192, 64, 3, dict(ldc=200, bias=True)),
    (2048, 2048, 128, 16, dict(lda=2048, ldb=2048, sa=128, sb=128 * 2048, beta0=True)),  # C5 GEMM
]


@pytest.mark.parametrize("case", BF16_CASES, ids=lambda c: "m%d_n%d_k%d_br%d" % c[:4])
def test_brgemm_bf16_vnni_fast(rt, case):
    m, n, k, br, kw = case
    blocks = None
    if m * n * k * br > 2 ** 31:  # oracle time: check row samples instead (rows are independent)
        blocks = [(0, 48), (m // 2 - 16, 40), (m - 40, 40)]
    name = gemm_case(rt, BF16, m, n, k, br, vnni=True, seed=m + n, row_blocks=blocks, **kw)
    assert "bf16" in name, name


BF16_FORCED = [
    # (variant, m, n, k, br, kwargs): every bf16 tile family on small outputs (the tile is normally chosen by
    # output size; forcing it lets the oracle check whole outputs), all epilogues, ring tails 0..8 chunks
    (18, 256, 256, 64, 1, dict(beta0=True)),
    (18, 256, 512, 64, 3, dict(bias=True, relu=True, ldc=520, lda=200, offs=(8, 16, 8, 4))),
    (18, 512, 256, 128, 2, dict(sa=64, sb=128, lda=512, beta0=True, bias=True)),
    (18, 1024, 512, 64, 5, dict(lda=320, sa=64, ldb=512, sb=64 * 512, beta0=True, relu=True)),  # XCD-blocked grid
    (18, 256, 256, 64, 0, dict(beta0=True, bias=True)),                                         # empty batch
    (18, 256, 256, 64, 0, dict()),
    (18, 256, 256, 192, 1, dict()),                                                             # beta = 1
    (17, 128, 256, 64, 3, dict(bias=True, relu=True, ldc=264)),
    (17, 512, 256, 64, 2, dict(beta0=True)),
    (16, 64, 192, 64, 3, dict(bias=True)),
    (16, 256, 256, 128, 1, dict(beta0=True, relu=True)),
]


@pytest.mark.parametrize("case", BF16_FORCED, ids=lambda c: "v%d_m%d_n%d_k%d_br%d" % c[:5])
def test_brgemm_bf16_forced_tile_families(rt, case):
    v, m, n, k, br, kw = case
    name = gemm_case(rt, BF16, m, n, k, br, vnni=True, seed=v * 1000 + m + n + br, force=v, **kw)
    assert {16: "64x64", 17: "128x128", 18: "256x256"}[v] in name, name


def test_brgemm_bf16_large_output_uses_256_tiles(rt):
    """4096 x 4096 x 512: one 256 x 256 tile per CU; row blocks sampled against the oracle"""
    name = gemm_case(rt, BF16, 4096, 4096, 64, 8, lda=512, ldb=4096, sa=64, sb=64 * 4096, vnni=True, beta0=True,
                     bias=True, relu=True, seed=77, row_blocks=[(0, 8), (250, 12), (2047, 6), (4090, 6)])
    assert "256x256" in name, name


@pytest.mark.parametrize("shape", [(32, 32, 32, 32), (32, 64, 32, 3), (96, 32, 96, 2), (64, 64, 32, 1), (128, 96, 160, 2)],
                         ids=lambda s: "x".join(map(str, s)))
def test_brgemm_bf16_vnni_generic_vector_loads(rt, shape):
    """the compiler-native bf16 tile (32x32x32, VNNI-2 B, packed blocks) and relatives with k not a multiple of
    64: the grouped kernel's 16-byte-load path (aligned leading dimensions), all epilogues"""
    m, n, k, br = shape
    for i, kw in enumerate((dict(beta0=True), dict(bias=True, relu=True), dict(beta0=True, bias=True, ldc=n + 8, lda=k + 8,
                                                                             ldb=n + 4, offs=(8, 8, 8, 4)))):
        name = gemm_case(rt, BF16, m, n, k, br, vnni=True, seed=sum(shape) + i, force=8, **kw)
        assert "grouped" in name, name


BF16_SMALL = [
    # (m, n, k, br, kwargs): 32x32 tiles, K split over the waves, fragments straight from global memory
    (32, 32, 32, 32, dict(beta0=True)),                                   # the compiler-native bf16 tile
    (32, 32, 32, 3, dict(bias=True, relu=True)),
    (64, 96, 48, 3, dict(beta0=True, bias=True, ldc=104, lda=56, ldb=100, offs=(8, 8, 4, 4))),
    (96, 32, 16, 5, dict()),                                              # beta = 1, one K step per batch element
    (32, 64, 64, 1, dict(relu=True)),                                     # fewer K steps than waves x group
    (256, 1024, 64, 16, dict(lda=1024, ldb=1024, sa=64, sb=65536, beta0=True, bias=True, relu=True)),  # --batch=256 layer
    (512, 1024, 64, 16, dict(lda=1024, ldb=1024, sa=64, sb=65536, beta0=True)),                        # an 8-GPU shard
    (64, 64, 64, 0, dict(beta0=True, bias=True)),                         # empty batch: C = bias
    (160, 224, 80, 7, dict(sa=16, sb=32, lda=320, ldb=224, beta0=True)),  # overlapping batch elements
]


@pytest.mark.parametrize("case", BF16_SMALL, ids=lambda c: "m%d_n%d_k%d_br%d" % c[:4])
def test_brgemm_bf16_small_outputs(rt, case):
    m, n, k, br, kw = case
    name = gemm_case(rt, BF16, m, n, k, br, vnni=True, seed=m + n + k + br, **kw)
    # (512 x 1024 fills the chip with 32x64 loader-wave tiles since round 3; everything smaller stays on the K-split family)
    assert ("lw<32x64" if m * n >= 512 * 1024 else "small") in name, name


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("shape", [(64, 48, 64, 4), (32, 48, 32, 3), (40, 48, 32, 2), (72, 100, 96, 2), (8, 4, 32, 1)],
                         ids=lambda s: "x".join(map(str, s)))
def test_brgemm_ragged_tiles_vector_loads(rt, dt, shape):
    """--tiles=64,48,64 / 32,48,32 of the reference's benchmark configs and other shapes whose m / n are not
    multiples of 32 (n a multiple of 4, k of 32): the grouped kernel's 16-byte-load path with predicated edges"""
    m, n, k, br = shape
    for i, kw in enumerate((dict(beta0=True, bias=True, relu=True), dict(ldc=n + 8, lda=k + 8, ldb=n + 4, offs=(8, 8, 4, 4)))):
        name = gemm_case(rt, dt, m, n, k, br, vnni=(dt == BF16), seed=sum(shape) + i, force=8, **kw)
        assert "grouped" in name, name


@pytest.mark.parametrize("shape", [(6, 6, 6, 2), (32, 32, 32, 4), (64, 48, 64, 3), (10, 7, 4, 1), (128, 256, 64, 2)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("mode", ["device", "host"])
def test_brgemm_vnni_a_and_vnni_c_operands(rt, shape, mode):
    """wire flags 4096 (dialect vnni_a: A is [m][k/2][2] = row-major bytes) and 8192 (vnni_c: C stored and, with
    beta = 1, read as VNNI-2 [m/2][n][2]); XsmmEnum.td:72-84, dispatch tuples of xsmm-to-func.mlir:62,80,98.
    Oracle semantics are unpinned in the reference tree (see oracle/xsmm_oracle.c c_index)."""
    m, n, k, br = shape
    rng = np.random.default_rng(sum(shape))
    lda, ldb, ldc = k + 2, n + 1, n + 3
    A, B = rand(rng, br * m * lda + 8, BF16), rand(rng, br * (k // 2) * 2 * ldb + 2 * ldb + 8, BF16)
    D = rand(rng, n + 8, BF16)
    for flags, fused in ((VB | 4096, False), (VB | 8192 | 4, True), (VB | 4096 | 8192, False), (VB | 4096 | 8192 | 4, True)):
        C = rand(rng, m * ldc + 2 * ldc + 8, BF16)
        ref = C.copy()
        if fused:
            orc.fused_brgemm(BF16, m, n, k, lda, ldb, ldc, m * lda, k * ldb, flags, 0, 5, 4, 1, A, 2, B, 2, ref, 3, D, 1, br)
            h = rt.fused_brgemm_dispatch(BF16, m, n, k, lda, ldb, ldc, m * lda, k * ldb, flags, 0, 5, 4, 1)
        else:
            orc.brgemm(BF16, m, n, k, lda, ldb, ldc, m * lda, k * ldb, flags, A, 2, B, 2, ref, 3, br)
            h = rt.brgemm_dispatch(BF16, m, n, k, lda, ldb, ldc, m * lda, k * ldb, flags)
        if mode == "device":
            dA, dB, dC, dD = dev(A), dev(B), dev(C), dev(D)
            if fused:
                rt.fused_brgemm(BF16, h, dA, 2, dB, 2, dC, 3, dD, 1, br)
            else:
                rt.brgemm(BF16, h, dA, 2, dB, 2, dC, 3, br)
            got = host(dC, C)
        else:
            got = C.copy()
            if fused:
                rt.fused_brgemm(BF16, h, A, 2, B, 2, got, 3, D, 1, br)
            else:
                rt.brgemm(BF16, h, A, 2, B, 2, got, 3, br)
        check_close(got, ref, BF16, "vnni flags %d [%s]" % (flags, rt.kernel_name(h)))
        ii, jj = np.meshgrid(np.arange(m), np.arange(n), indexing="ij")
        foot = 3 + ((ii // 2) * (2 * ldc) + 2 * jj + ii % 2 if flags & 8192 else ii * ldc + jj)
        outside = np.ones(C.size, dtype=bool)
        outside[foot.reshape(-1)] = False
        assert np.array_equal(got[outside], C[outside]), "wrote outside the output footprint (flags %d)" % flags
        assert np.array_equal(ref[outside], C[outside])


def test_conv_as_gemm_invokes_strided_rows(rt):
    """conv2d NHWC x HWCF as the reference rewrites it (RewriteConvsToMatmulOrBrgemm.cpp: one matmul per output row
    and filter tap, A = image rows [Q x C] with row stride conv_stride * C, B = filter tap [C x K], C = output row
    [Q x K], accumulating) - the lda > k, overlapping-window operand pattern - through the ABI on device buffers,
    against a direct numpy convolution (independent of the oracle) and against the oracle's replay of the same calls"""
    rng = np.random.default_rng(8)
    for (H, W, Cin, K, R, S, st) in ((5, 5, 3, 8, 3, 3, 2), (9, 9, 16, 32, 3, 3, 1), (12, 10, 8, 64, 1, 1, 1), (17, 17, 4, 12, 5, 3, 2)):
        P, Q = (H - R) // st + 1, (W - S) // st + 1
        img = rng.uniform(-1, 1, H * W * Cin).astype(np.float32)
        flt = rng.uniform(-1, 1, R * S * Cin * K).astype(np.float32)
        out0 = rng.uniform(-1, 1, P * Q * K).astype(np.float32)
        want = out0.reshape(P, Q, K).astype(np.float64).copy()
        im, fl = img.reshape(H, W, Cin).astype(np.float64), flt.reshape(R, S, Cin, K).astype(np.float64)
        for p_ in range(P):
            for q in range(Q):
                want[p_, q] += np.einsum("rsc,rsck->k", im[p_ * st:p_ * st + R, q * st:q * st + S], fl)
        h = rt.gemm_dispatch(F32, Q, K, Cin, st * Cin, K, K, 0)
        dimg, dflt, dout = dev(img), dev(flt), dev(out0)
        ref = out0.copy()
        for p_ in range(P):
            for r in range(R):
                for s_ in range(S):
                    oa, ob, oc = ((p_ * st + r) * W + s_) * Cin, (r * S + s_) * Cin * K, p_ * Q * K
                    rt.gemm(F32, h, dimg, oa, dflt, ob, dout, oc)
                    orc.gemm(F32, Q, K, Cin, st * Cin, K, K, 0, img, oa, flt, ob, ref, oc)
        got = host(dout, out0)
        check_close(got, ref, F32, "conv %s as gemm invokes [%s]" % ((H, W, Cin, K, R, S, st), rt.kernel_name(h)))
        assert np.abs(got.reshape(P, Q, K) - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("k", [4, 36, 100, 1000])
def test_brgemm_f32_generic_k_multiple_of_4(rt, k):
    """the generic kernel's 16-byte operand path with a ragged last chunk (k a multiple of 4, not of 32): the pieces at or beyond k
    are requested past the descriptor's end and must read as zeros - whatever lies behind the operands in memory (NaN here)"""
    m, n, br = 96, 72, 3
    lda, ldb = k + 8, n + 4
    rng = np.random.default_rng(k)
    A = np.full(br * m * lda + 64, np.nan, dtype=np.float32)
    B = np.full(br * k * ldb + 64, np.nan, dtype=np.float32)
    for b in range(br):
        A[b * m * lda:(b + 1) * m * lda].reshape(m, lda)[:, :k] = rng.uniform(-1, 1, (m, k))
        B[b * k * ldb:(b + 1) * k * ldb].reshape(k, ldb)[:, :n] = rng.uniform(-1, 1, (k, n))
    C = rng.uniform(-1, 1, m * n).astype(np.float32)
    Cref = C.copy()
    An, Bn = np.nan_to_num(A), np.nan_to_num(B)
    orc.brgemm(F32, m, n, k, lda, ldb, n, m * lda, k * ldb, 0, An, 0, Bn, 0, Cref, 0, br)
    rt.force_variant(8)
    try:
        h = rt.brgemm_dispatch(F32, m, n, k, lda, ldb, n, m * lda, k * ldb, 0)
    finally:
        rt.force_variant(-1)
    dC = dev(C)
    rt.brgemm(F32, h, dev(A), 0, dev(B), 0, dC, 0, br)
    got = host(dC, C)
    assert np.isfinite(got).all(), "a tail piece beyond k was read"
    check_close(got, Cref, F32, "generic f32 k=%d [%s]" % (k, rt.kernel_name(h)), None, k * br)


def test_brgemm_bf16_flat_b_generic(rt):
    gemm_case(rt, BF16, 48, 40, 24, 3, vnni=False, seed=3, bias=True)


def _random_gemm_case(rng, dt):
    """one random dispatch the compiler could emit: half the draws are shaped for the fast kernels
    (multiples of 64, 16-byte-aligned leading dimensions and offsets), half are ragged; leading
    dimensions, batch strides, offsets and the epilogue are drawn independently"""
    aligned = rng.random() < 0.5
    if aligned:
        m, n = int(rng.integers(1, 7)) * 64, int(rng.integers(1, 7)) * 64
        k = int(rng.integers(1, 4)) * 64
        q = 8
    else:
        m, n, k = int(rng.integers(1, 150)), int(rng.integers(1, 150)), int(rng.integers(1, 100))
        q = 1
    vnni = dt == BF16 and (k % 2 == 0) and rng.random() < 0.8
    br = int(rng.integers(0, 6))
    lda = k + q * int(rng.integers(0, 5))
    ldb = n + q * int(rng.integers(0, 5))
    ldc = n + q * int(rng.integers(0, 5))
    kp2 = ((k + 1) // 2) * 2
    choice = rng.random()
    if choice < 0.4:      # batch along k inside one row-major matrix (mlir-gen whole-layer form)
        lda = k * max(br, 1) + q * int(rng.integers(0, 3))
        sa, sb = k, (kp2 if vnni else k) * ldb
    elif choice < 0.8:    # packed blocks one after the other
        sa, sb = m * lda, (kp2 if vnni else k) * ldb
    else:                 # overlapping / repeated operands (stride smaller than a matrix)
        sa, sb = q * int(rng.integers(0, 3)), q * int(rng.integers(0, 3))
    offs = tuple(q * int(rng.integers(0, 4)) for _ in range(4))
    bias, relu = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    return dict(m=m, n=n, k=k, br=br, lda=lda, ldb=ldb, ldc=ldc, sa=sa, sb=sb, offs=offs, vnni=vnni,
                beta0=bool(rng.integers(0, 2)), bias=bias, relu=relu, fused=bias or relu or bool(rng.integers(0, 2)))


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("chunk", range(6))
def test_brgemm_random_dispatches(rt, dt, chunk):
    """seeded random sweep over the whole dispatch tuple (kernel selection boundaries included)"""
    rng = np.random.default_rng(1000 * dt + chunk)
    seen = set()
    for i in range(14):
        c = _random_gemm_case(rng, dt)
        seen.add(gemm_case(rt, dt, seed=chunk * 100 + i, mode="device" if i % 4 else "host", **c))
    assert seen  # kernel names exercised (both fast and grouped paths occur over the chunks)


# ---------------------------------------------------------------- unary / binary
UNARY = [(1, 0), (1, 2), (1, 4), (1, 8), (5, 0), (5, 2), (5, 4), (5, 8), (2, 0), (2, 8)]


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("kind,flags", UNARY)
@pytest.mark.parametrize("shape", [(3, 3, 0), (13, 40, 3), (256, 1024, 0), (64, 48, 16)])
def test_unary_eltwise(rt, dt, kind, flags, shape):
    m, n, pad = shape
    rng = np.random.default_rng(kind * 100 + flags + m)
    ldo = n + pad
    ldi = {0: n + pad, 2: 1, 4: n, 8: 1}[flags]
    X = rand(rng, m * max(ldi, 1) + n + 8, dt)
    O = rand(rng, m * ldo + 8, dt)
    ref = O.copy()
    orc.unary(kind, dt, m, n, ldi, ldo, flags, X, 0, ref, 0)
    h = rt.unary_dispatch(kind, dt, m, n, ldi, ldo, flags)
    for mode in ("device", "host"):
        if mode == "device":
            dO = dev(O)
            rt.unary(dt, h, dev(X), 0, dO, 0)
            got = host(dO, O)
        else:
            got = O.copy()
            rt.unary(dt, h, X, 0, got, 0)
        assert np.array_equal(got, ref), "unary kind %d flags %d %s %s: not bit-identical to the oracle" % (
            kind, flags, shape, mode)


@pytest.mark.parametrize("dt", [F32, BF16])
def test_unary_scalar_invoke(rt, dt):
    # the scalar operand is an f32 regardless of dtype (XsmmRunnerUtils.cpp:276-286)
    for kind, val in ((1, 1.2345678), (5, -3.0), (2, 9.0)):
        O = rand(np.random.default_rng(1), 16 * 40, dt)
        ref = O.copy()
        orc.unary_scalar(kind, dt, 16, 33, 1, 40, 8, val, ref, 0)
        h = rt.unary_dispatch(kind, dt, 16, 33, 1, 40, 8)
        dO = dev(O)
        rt.unary_scalar(dt, h, val, dO, 0)
        assert np.array_equal(host(dO, O), ref)


def test_relu_in_place(rt):
    X = rand(np.random.default_rng(2), 64 * 64, F32)
    ref = np.maximum(X, 0).astype(np.float32)
    h = rt.unary_dispatch(5, F32, 64, 64, 64, 64, 0)
    dX = dev(X)
    rt.unary(F32, h, dX, 0, dX, 0)
    assert np.array_equal(host(dX, X), ref)
    Xh = X.copy()
    rt.unary(F32, h, Xh, 0, Xh, 0)  # host pointers aliasing exactly: one mirror
    assert np.array_equal(Xh, ref)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("shape", [(4, 8), (3, 5), (64, 64), (100, 37), (1024, 512)])
def test_transpose_bit_exact(rt, dt, shape):
    m, n = shape
    X = rand(np.random.default_rng(m), m * (n + 1), dt)
    O = np.zeros(n * (m + 2), dtype=X.dtype)
    ref = O.copy()
    orc.unary(29, dt, m, n, n + 1, m + 2, 0, X, 0, ref, 0)
    h = rt.unary_dispatch(29, dt, m, n, n + 1, m + 2, 0)
    dO = dev(O)
    rt.unary(dt, h, dev(X), 0, dO, 0)
    got = host(dO, O)
    assert got.tobytes() == ref.tobytes()
    # involution: transposing back restores the input window bit for bit
    back = dev(np.zeros(m * (n + 1), dtype=X.dtype))
    h2 = rt.unary_dispatch(29, dt, n, m, m + 2, n + 1, 0)
    rt.unary(dt, h2, dO, 0, back, 0)
    b = host(back, X).reshape(m, n + 1)[:, :n]
    assert np.array_equal(b, X[: m * (n + 1)].reshape(m, n + 1)[:, :n])


@pytest.mark.parametrize("shape", [(16, 16, 16, 16), (4, 4, 4, 4), (6, 10, 11, 12), (2048, 2048, 2048, 2048), (64, 40, 48, 40)])
def test_vnni2_pack_bit_exact(rt, shape):
    m, n, ldi, ldo = shape
    X = rand(np.random.default_rng(n), m * ldi, BF16)
    O = np.zeros((m // 2) * 2 * ldo + 8, dtype=np.uint16)
    ref = O.copy()
    orc.unary(28, BF16, m, n, ldi, ldo, 0, X, 0, ref, 0)
    h = rt.unary_dispatch(28, BF16, m, n, ldi, ldo, 0)
    dO = dev(O)
    rt.unary(BF16, h, dev(X), 0, dO, 0)
    assert host(dO, O).tobytes() == ref.tobytes()
    # unpack property: out[(i/2)][j][i%2] read back in (i, j) order is the input
    got = host(dO, O)[: (m // 2) * 2 * ldo].reshape(m // 2, ldo, 2)[:, :n, :]
    assert np.array_equal(got.transpose(0, 2, 1).reshape(m, n), X.reshape(m, ldi)[:, :n])


BIN_FLAGS = [0, 1, 2, 4, 8, 16, 32, 1 | 8, 4 | 2, 16 | 2, 4 | 32]


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("kind", [1, 2, 3, 4])
@pytest.mark.parametrize("flags", BIN_FLAGS)
def test_binary_eltwise(rt, dt, kind, flags):
    for (m, n, pad) in ((4, 8, 0), (13, 37, 2), (128, 256, 0)):
        rng = np.random.default_rng(kind * 1000 + flags * 10 + m)
        ldo = n + pad

        def ld_for(row, col, sc):
            return 1 if flags & (row | sc) else (n if flags & col else n + pad)
        ldl, ldr = ld_for(1, 4, 16), ld_for(2, 8, 32)
        L = rand(rng, m * max(ldl, n) + 8, dt, 0.5, 2.0)
        R = rand(rng, m * max(ldr, n) + 8, dt, 0.5, 2.0)
        O = rand(rng, m * ldo + 8, dt)
        ref = O.copy()
        orc.binary(kind, dt, m, n, ldl, ldr, ldo, flags, L, 0, R, 0, ref, 0)
        h = rt.binary_dispatch(kind, dt, m, n, ldl, ldr, ldo, flags)
        dO = dev(O)
        rt.binary(dt, h, dev(L), 0, dev(R), 0, dO, 0)
        got = host(dO, O)
        if kind == 4:  # division: device fp32 divide may differ from the host's in the last place
            check_close(got, ref, dt, "binary div flags %d" % flags)
            if dt == F32:
                assert np.abs(got.astype(np.float64) - ref).max() <= 2.5e-7 * np.abs(ref).max()
        else:
            assert np.array_equal(got, ref), "binary kind %d flags %d (%d x %d)" % (kind, flags, m, n)


def test_binary_out_aliases_input(rt):
    X = rand(np.random.default_rng(4), 32 * 32, F32)
    bias = rand(np.random.default_rng(5), 32, F32)
    ref = (X.reshape(32, 32) + bias[None, :]).reshape(-1)
    h = rt.binary_dispatch(1, F32, 32, 32, 32, 32, 32, 8)  # rhs = bcast_col_in1
    dX = dev(X)
    rt.binary(F32, h, dX, 0, dev(bias), 0, dX, 0)
    assert np.array_equal(host(dX, X), ref)
    h2 = rt.binary_dispatch(2, F32, 32, 32, 32, 32, 32, 0)  # lhs == rhs (DefaultPipeline/xsmm.mlir:19-21)
    Xh = X.copy()
    rt.binary(F32, h2, Xh, 0, Xh, 0, Xh, 0)
    assert np.array_equal(Xh, X * X)


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("queue", [False, True])
@pytest.mark.parametrize("chunk", range(3))
def test_eltwise_random_dispatches(rt, dt, queue, chunk):
    """seeded random sweep over unary / binary dispatch tuples (kinds, broadcast flags, leading dimensions,
    element offsets that break / keep 16-byte alignment, tile-sized and large shapes), bit-exact against
    the oracle; run once with direct launches and once through the tile queue (async mode)"""
    rng = np.random.default_rng(7000 + 100 * dt + chunk)
    prev_async = rt.set_async(queue)
    prev_queue = rt.set_tile_queue(queue)
    keep = []
    try:
        for i in range(40):
            small = rng.random() < 0.6
            m = int(rng.integers(1, 65)) if small else int(rng.integers(65, 300))
            n = int(rng.integers(1, 65)) if small else int(rng.integers(65, 400))
            if rng.random() < 0.4:
                m, n = (m + 7) // 8 * 8, (n + 7) // 8 * 8
            q = 8 if rng.random() < 0.5 else 1
            off = [q * int(rng.integers(0, 4)) for _ in range(3)]
            if rng.random() < 0.55:  # unary
                kind = int(rng.choice([1, 2, 5, 29, 28] if dt == BF16 else [1, 2, 5, 29]))
                flags = int(rng.choice([0, 2, 4, 8])) if kind in (1, 5) else 0
                if kind == 28:
                    m += m & 1
                ldi = {0: n + q * int(rng.integers(0, 3)), 2: 1, 4: n, 8: 1}[flags]
                ldo = (m if kind == 29 else n) + q * int(rng.integers(0, 3))
                X = rand(rng, off[0] + m * max(ldi, 1) + n + 8, dt)
                rows_out = n if kind == 29 else m
                O = rand(rng, off[2] + rows_out * ldo + n + m + 8, dt)
                ref = O.copy()
                orc.unary(kind, dt, m, n, ldi, ldo, flags, X, off[0], ref, off[2])
                h = rt.unary_dispatch(kind, dt, m, n, ldi, ldo, flags)
                dX, dO = dev(X), dev(O)
                rt.unary(dt, h, dX, off[0], dO, off[2])
                keep.append((dO, O, ref, "unary kind %d flags %d m%d n%d ldi%d ldo%d off%s" % (kind, flags, m, n, ldi, ldo, off), True, dX))
            else:
                kind = int(rng.integers(1, 5))
                f0, f1 = int(rng.choice([0, 1, 4, 16])), int(rng.choice([0, 2, 8, 32]))
                flags = f0 | f1
                ldl = 1 if f0 in (1, 16) else (n if f0 == 4 else n + q * int(rng.integers(0, 3)))
                ldr = 1 if f1 in (2, 32) else (n if f1 == 8 else n + q * int(rng.integers(0, 3)))
                ldo = n + q * int(rng.integers(0, 3))
                L = rand(rng, off[0] + m * max(ldl, n) + 8, dt, 0.5, 2.0)
                R = rand(rng, off[1] + m * max(ldr, n) + 8, dt, 0.5, 2.0)
                O = rand(rng, off[2] + m * ldo + 8, dt)
                ref = O.copy()
                orc.binary(kind, dt, m, n, ldl, ldr, ldo, flags, L, off[0], R, off[1], ref, off[2])
                h = rt.binary_dispatch(kind, dt, m, n, ldl, ldr, ldo, flags)
                dL, dR, dO = dev(L), dev(R), dev(O)
                rt.binary(dt, h, dL, off[0], dR, off[1], dO, off[2])
                keep.append((dO, O, ref, "binary kind %d flags %d m%d n%d ldl%d ldr%d ldo%d off%s" % (kind, flags, m, n, ldl, ldr, ldo, off),
                             kind != 4, dL, dR))
        rt.synchronize()
        for entry in keep:
            dO, O, ref, what, exact = entry[:5]
            got = host(dO, O)
            if exact:
                assert got.tobytes() == ref.tobytes(), what + (" (queued)" if queue else "")
            else:
                check_close(got, ref, dt, what)
    finally:
        rt.synchronize()
        rt.set_tile_queue(prev_queue)
        rt.set_async(prev_async)


# ---------------------------------------------------------------- BASELINE configs, properties
def test_c2_full_size_and_properties(rt):
    """BASELINE config 2: C[1024x1024] += sum_{b<16} A_b[1024x64] B_b[64x1024], f32, with
    tpp-run's `normal` init stream (seed 123) and with a sign-cancelling uniform stream."""
    import torch
    m = n = 1024
    k, br = 64, 16
    gen = orc.TensorInit("normal", 123)
    for trial, (A, B, C) in enumerate([
            (gen.fill(m * 1024), gen.fill(1024 * n), gen.fill(m * n)),
            tuple(rand(np.random.default_rng(s), 1024 * 1024, F32) for s in (1, 2, 3))]):
        ref = C.copy()
        orc.fused_brgemm_omp(F32, m, n, k, 1024, 1024, 1024, 64, 65536, 0, 0, 0, A, B, ref, None, br)
        h = rt.brgemm_dispatch(F32, m, n, k, 1024, 1024, 1024, 64, 65536, 0)
        dA, dB, dC = dev(A), dev(B), dev(C)
        rt.brgemm(F32, h, dA, 0, dB, 0, dC, 0, br)
        got = host(dC, C)
        truth = C.astype(np.float64).reshape(m, n) + A.astype(np.float64).reshape(m, 1024) @ B.astype(np.float64).reshape(1024, n)
        # the element-wise bar of SURVEY 8(d) at BASELINE size: |C| + sum_k |a||b| per element, in fp64 by numpy
        mag = np.abs(C).astype(np.float64).reshape(m, n) + np.abs(A).astype(np.float64).reshape(m, 1024) @ np.abs(B).astype(np.float64).reshape(1024, n)
        check_close(got, ref, F32, "C2 trial %d" % trial, mag=mag.reshape(-1), K=1024, truth=truth.reshape(-1))
        # determinism: same inputs -> bit-identical output (no atomics, fixed summation order)
        dC2 = dev(C)
        rt.brgemm(F32, h, dA, 0, dB, 0, dC2, 0, br)
        assert torch.equal(dC, dC2)
        # beta: (beta=1 from C) == (beta=0 result) + C up to one rounding of the final add
        h0 = rt.brgemm_dispatch(F32, m, n, k, 1024, 1024, 1024, 64, 65536, 4)
        dZ = dev(np.zeros_like(C))
        rt.brgemm(F32, h0, dA, 0, dB, 0, dZ, 0, br)
        alt = host(dZ, C).astype(np.float64) + C
        assert np.abs(alt - got).max() <= 1e-5 * max(1.0, np.abs(got).max())
        # linearity in A: scaling A by 2 (exact in binary fp) doubles the beta=0 product bit for bit
        dZ2 = dev(np.zeros_like(C))
        rt.brgemm(F32, h0, dev(A * np.float32(2)), 0, dB, 0, dZ2, 0, br)
        assert torch.equal(dZ2, dZ * 2)
        # the same problem as ONE batch of k = 1024 (stride irrelevant) must agree to rounding
        h1 = rt.brgemm_dispatch(F32, m, n, 1024, 1024, 1024, 1024, 0, 0, 4)
        dZ3 = dev(np.zeros_like(C))
        rt.brgemm(F32, h1, dA, 0, dB, 0, dZ3, 0, 1)
        assert torch.equal(dZ3, dZ)  # identical chunk order -> identical bits


def test_c3_fused_layer_full_size(rt):
    m, n, k, br = 512, 1024, 64, 16
    gen = orc.TensorInit("normal", 123)
    A, W, bias = gen.fill(m * 1024), gen.fill(1024 * n), gen.fill(n)
    A -= np.float32(0.08)  # the normal init is clamped to [0, 1]: shift it so relu clips about half the outputs
    C = np.full(m * n, np.float32(7.0))
    ref = C.copy()
    orc.fused_brgemm_omp(F32, m, n, k, 1024, 1024, 1024, 64, 65536, 4, 5, 1, A, W, ref, bias, br)
    h = rt.fused_brgemm_dispatch(F32, m, n, k, 1024, 1024, 1024, 64, 65536, 4, 0, 5, 4, 1)
    dC = dev(C)
    rt.fused_brgemm(F32, h, dev(A), 0, dev(W), 0, dC, 0, dev(bias), 0, br)
    got = host(dC, C)
    truth = np.maximum(A.astype(np.float64).reshape(m, 1024) @ W.astype(np.float64).reshape(1024, n) + bias.astype(np.float64), 0.0)
    # the element-wise bar at BASELINE size: sum_k |a||w| + |bias| per element (beta = 0: C does not enter), in fp64 by numpy
    mag = np.abs(A).astype(np.float64).reshape(m, 1024) @ np.abs(W).astype(np.float64).reshape(1024, n) + np.abs(bias).astype(np.float64)
    check_close(got, ref, F32, "C3 fused layer", mag=mag.reshape(-1), K=1024, truth=truth.reshape(-1))
    assert (got >= 0).all() and (got == 0).any()
    # relu idempotence: applying xsmm.unary relu to the output changes nothing
    hr = rt.unary_dispatch(5, F32, m, n, n, n, 0)
    dC2 = dC.clone()
    rt.unary(F32, hr, dC2, 0, dC2, 0)
    import torch
    assert torch.equal(dC, dC2)


def test_c4_mlp_bf16_three_layers(rt):
    """BASELINE config 4 on one GPU: 3 x (4096x1024x1024 bf16, bias + relu) on the REFERENCE's input stream - weights and
    biases from mlir-gen's seed chain (--seed 123: MLIRGen.cpp:131-137, 810-819; every constant its own `normal` generator),
    the input from tpp-run's (--seed 123), VNNI-2 weights produced by the runtime's own pack op. Checked per layer: the
    oracle is fed the GPU's OWN previous activations, so every layer is held to one bf16 ulp + the f32 accumulation floor
    (a wrong rounding in layer 2 cannot hide in a chained tolerance); then the chained result against the oracle's chain at
    the reference's differential bar (fpcmp -r 0.01, vnni-xsmm-vs-loops.mlir:13). Rows are independent: a row sample."""
    spec = pkg.MlpSpec()
    N = 1024
    Wflat, biases, seeds = orc.mlir_gen_mlp_tensors(spec.layers, 123, BF16)
    assert seeds[:3] == [123, 128959393, 1692901013]  # srand(123); rand(), rand() of glibc
    X = orc.TensorInit("normal", 123).fill(spec.batch * N, BF16)
    hp = rt.unary_dispatch(28, BF16, N, N, N, N, 0)
    dW, Wv = [], []
    for w in Wflat:
        o = dev(np.zeros(N * N, np.uint16))
        rt.unary(BF16, hp, dev(w), 0, o, 0)
        dW.append(o)
        wv = np.zeros(N * N, np.uint16)
        orc.unary(28, BF16, N, N, N, N, 0, w, 0, wv, 0)
        Wv.append(wv)
        assert np.array_equal(host(o, w), wv)  # the pack is a bit-exact move
    rows = np.r_[0:64, 2000:2032, 4064:4096]
    for chain in (False, True):  # three launches, then the rank's step as one chain launch: same bars
        was_async = rt.set_async(True)
        try:
            mlp = pkg.ShardedMlp(spec, 0, 1, rt, chain=chain)
            acts = [dev(np.full(spec.batch * N, 0x7fc0, np.uint16)) for _ in range(3)]
            mlp.forward(dev(X), dW, [dev(b) for b in biases], acts)
            rt.synchronize()
            assert mlp.last_step_fused == chain
        finally:
            rt.set_async(was_async)
        gpu = [X.reshape(spec.batch, N)] + [host(a, X).reshape(spec.batch, N) for a in acts]
        chained = gpu[0][rows].copy().reshape(-1)
        for l in range(3):
            fed = gpu[l][rows].copy().reshape(-1)  # the GPU's own input of this layer
            one = np.zeros(len(rows) * N, np.uint16)
            orc.fused_brgemm(BF16, len(rows), N, 64, N, N, N, 64, 64 * N, 4 | VB, 0, 5, 4, 1, fed, 0, Wv[l], 0, one, 0, biases[l], 0, 16)
            check_close(gpu[l + 1][rows].reshape(-1), one, BF16, "C4 layer %d (chain launch %s)" % (l, chain))
            nxt = np.zeros(len(rows) * N, np.uint16)
            orc.fused_brgemm(BF16, len(rows), N, 64, N, N, N, 64, 64 * N, 4 | VB, 0, 5, 4, 1, chained, 0, Wv[l], 0, nxt, 0, biases[l], 0, 16)
            chained = nxt
        ref = orc.bf16_to_f32(chained).reshape(len(rows), N).astype(np.float64)
        g = orc.bf16_to_f32(gpu[3][rows].reshape(-1)).reshape(len(rows), N).astype(np.float64)
        err = np.abs(g - ref) - 0.01 * np.abs(ref) - 2.0 ** -8 * np.abs(ref).max()
        assert err.max() <= 0, float(err.max())
        assert np.mean(g == ref) > 0.5  # and most outputs are bit-identical


def test_async_mode_and_stream(rt):
    import torch
    X = rand(np.random.default_rng(8), 256 * 256, F32)
    h = rt.brgemm_dispatch(F32, 256, 256, 64, 256, 256, 256, 64, 64 * 256, 4)
    dA, dB = dev(X), dev(X[::-1].copy())
    sync_out = dev(np.zeros(256 * 256, np.float32))
    rt.brgemm(F32, h, dA, 0, dB, 0, sync_out, 0, 4)
    s = torch.cuda.Stream()
    prev = rt.set_async(True)
    try:
        rt.set_stream(s)
        outs = [dev(np.zeros(256 * 256, np.float32)) for _ in range(8)]
        torch.cuda.synchronize()
        t0 = rt.perf_start_timer()
        for o in outs:
            rt.brgemm(F32, h, dA, 0, dB, 0, o, 0, 4)
        dt = rt.perf_stop_timer(t0)  # drains the stream
        assert dt > 0
        for o in outs:
            assert torch.equal(o, sync_out)
    finally:
        rt.set_stream(None)
        rt.set_async(prev)


def test_concurrent_invokes_on_disjoint_tiles(rt):
    """invoke is re-entrant: the reference calls it from OpenMP workers on disjoint output
    tiles with one shared handle (pass-convert-mlp-to-parallel-tile.mlir:80-88)."""
    rng = np.random.default_rng(11)
    A, B = rand(rng, 8 * 32 * 32 * 4, F32), rand(rng, 8 * 32 * 32 * 4, F32)
    C = np.zeros(8 * 8 * 32 * 32, np.float32)
    ref = C.copy()
    h = rt.brgemm_dispatch(F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 4)
    tiles = [(i, j) for i in range(8) for j in range(8)]
    for (i, j) in tiles:
        orc.brgemm(F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 4, A, i * 4096, B, j * 4096, ref, (i * 8 + j) * 1024, 4)

    def worker(chunk):
        for (i, j) in chunk:
            rt.brgemm(F32, h, A, i * 4096, B, j * 4096, C, (i * 8 + j) * 1024, 4)
    ths = [threading.Thread(target=worker, args=(tiles[w::4],)) for w in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    check_close(C, ref, F32, "concurrent tiles")


@pytest.mark.parametrize("dt", [F32, BF16])
@pytest.mark.parametrize("op", ["gemm_beta1", "gemm_beta0_fused", "unary_relu", "binary_add_inplace", "transpose"])
def test_concurrent_host_invokes_on_adjacent_tiles_of_a_flat_buffer(rt, dt, op):
    """HOST pointers, 8 threads, 32x32 tiles of ONE flat row-major 256 x 1024 output (ld = 1024 > n): the rows of
    neighbouring tiles interleave in memory, so a mirror that copied back the bounding span of a tile would
    overwrite what another thread has just written (the reference's OpenMP callers on host memrefs,
    pass-convert-mlp-to-parallel-tile.mlir:80-88 - XsmmRunnerUtils.cpp:288-306 is synchronous and re-entrant).
    Every tile is checked against the oracle, several rounds so that interleavings vary."""
    LD, TM, TN, T = 1024, 8, 32, 32
    rng = np.random.default_rng(sum(map(ord, op)) + dt)
    tiles = [(i, j) for i in range(TM) for j in range(TN)]
    for rnd in range(3):
        A = rand(rng, TM * T * LD, dt)
        B = rand(rng, LD * LD, dt) if not op.startswith("gemm") or dt == F32 else rand(rng, LD * LD, dt)
        bias = rand(rng, LD, dt)
        C = rand(rng, TM * T * LD, dt)
        ref = C.copy()
        vn = VB if dt == BF16 else 0
        if op == "gemm_beta1":      # C tile (i, j) += A[i rows, 64 k] * B[64 k, j cols]
            h = rt.brgemm_dispatch(dt, T, T, 32, LD, LD, LD, 32, 32 * LD, vn)
            call = lambda i, j, c: rt.brgemm(dt, h, A, i * T * LD, B, (2 if vn else 1) * j * T, c, i * T * LD + j * T, 2)  # noqa: E731
            orc_call = lambda i, j: orc.brgemm(dt, T, T, 32, LD, LD, LD, 32, 32 * LD, vn, A, i * T * LD, B, (2 if vn else 1) * j * T, ref, i * T * LD + j * T, 2)  # noqa: E731
        elif op == "gemm_beta0_fused":
            h = rt.fused_brgemm_dispatch(dt, T, T, 32, LD, LD, LD, 32, 32 * LD, 4 | vn, 0, 5, 4, 1)
            call = lambda i, j, c: rt.fused_brgemm(dt, h, A, i * T * LD, B, (2 if vn else 1) * j * T, c, i * T * LD + j * T, bias, j * T, 2)  # noqa: E731
            orc_call = lambda i, j: orc.fused_brgemm(dt, T, T, 32, LD, LD, LD, 32, 32 * LD, 4 | vn, 0, 5, 4, 1, A, i * T * LD, B, (2 if vn else 1) * j * T, ref, i * T * LD + j * T, bias, j * T, 2)  # noqa: E731
        elif op == "unary_relu":    # out tile = relu(in tile), both strided
            h = rt.unary_dispatch(5, dt, T, T, LD, LD, 0)
            call = lambda i, j, c: rt.unary(dt, h, A, i * T * LD + j * T, c, i * T * LD + j * T)  # noqa: E731
            orc_call = lambda i, j: orc.unary(5, dt, T, T, LD, LD, 0, A, i * T * LD + j * T, ref, i * T * LD + j * T)  # noqa: E731
        elif op == "binary_add_inplace":  # out tile = out tile + bias row (out == lhs, strided)
            h = rt.binary_dispatch(1, dt, T, T, LD, T, LD, 8)
            call = lambda i, j, c: rt.binary(dt, h, c, i * T * LD + j * T, bias, j * T, c, i * T * LD + j * T)  # noqa: E731
            orc_call = lambda i, j: orc.binary(1, dt, T, T, LD, T, LD, 8, ref, i * T * LD + j * T, bias, j * T, ref, i * T * LD + j * T)  # noqa: E731
        else:                       # out tile (j, i) of the transposed grid <- in tile (i, j); out is 1024 x 256 seen as ld 1024 tiles
            h = rt.unary_dispatch(29, dt, T, T, LD, LD, 0)
            call = lambda i, j, c: rt.unary(dt, h, A, i * T * LD + j * T, c, (j % TM) * T * LD + (i + TM * (j // TM)) * T)  # noqa: E731
            orc_call = lambda i, j: orc.unary(29, dt, T, T, LD, LD, 0, A, i * T * LD + j * T, ref, (j % TM) * T * LD + (i + TM * (j // TM)) * T)  # noqa: E731
        for (i, j) in tiles:
            orc_call(i, j)
        errors = []

        def worker(chunk):
            try:
                for (i, j) in chunk:
                    call(i, j, C)
            except Exception as ex:  # noqa: BLE001
                errors.append(repr(ex))
        order = list(tiles)
        rng.shuffle(order)
        ths = [threading.Thread(target=worker, args=(order[w::8],)) for w in range(8)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not errors, errors
        if op in ("unary_relu", "transpose"):
            assert np.array_equal(C, ref), "%s dt%d round %d: %d elements differ" % (op, dt, rnd, int((C != ref).sum()))
        else:
            check_close(C, ref, dt, "%s dt%d round %d" % (op, dt, rnd))


def test_host_resident_operands(rt):
    """xsmm_hip_host_resident: declared host buffers are uploaded once; invokes on them (and on sub-ranges)
    use the device copy, results written into a resident buffer are still visible on the host on return,
    host_update re-uploads, host_release returns to per-call mirroring"""
    rng = np.random.default_rng(3)
    m = n = 128
    A, B = rand(rng, m * 256, F32), rand(rng, 256 * n, F32)
    C = np.zeros(m * n, np.float32)
    h = rt.brgemm_dispatch(F32, m, n, 64, 256, n, n, 64, 64 * n, 4)
    ref = C.copy()
    orc.brgemm(F32, m, n, 64, 256, n, n, 64, 64 * n, 4, A, 0, B, 0, ref, 0, 4)
    assert rt.host_resident(A) == 0 and rt.host_resident(B) == 0 and rt.host_resident(C) == 0
    assert rt.host_resident(A) == -1  # overlapping declaration
    try:
        rt.brgemm(F32, h, A, 0, B, 0, C, 0, 4)
        check_close(C, ref, F32, "resident operands")
        A2 = A.copy()
        A[:] = 0.0  # host change NOT announced: the device copy still holds the old values
        rt.brgemm(F32, h, A, 0, B, 0, C, 0, 4)
        check_close(C, ref, F32, "resident operands, stale host bytes ignored until host_update")
        assert rt.host_update(A) == 0
        rt.brgemm(F32, h, A, 0, B, 0, C, 0, 4)
        assert not C.any(), "after host_update the zeros must be used"
        A[:] = A2
        assert rt.host_update(A) == 0
    finally:
        for x in (A, B, C):
            assert rt.host_release(x) == 0
    C[:] = 0
    rt.brgemm(F32, h, A, 0, B, 0, C, 0, 4)
    check_close(C, ref, F32, "after release")


def test_c5_pack_prologue_then_vnni_brgemm_full_size(rt):
    """BASELINE config 5 end to end: xsmm.unary VNNI-2 pack of a flat bf16 B[2048x2048] on the GPU,
    then the bf16 BRGEMM 2048^3 on the packed operand (wire flags VNNI_B|BETA_0 = 2052). Checked
    against the oracle working on the FLAT B (no VNNI anywhere on the oracle side), on row samples
    - so the pack layout and the kernel's VNNI addressing must agree with the reference definition
    out[(k/2)][n][k%2] = in[k][n] (VNNIUtils.cpp:75-77), not merely with each other."""
    M = 2048
    rng = np.random.default_rng(55)
    A = orc.f32_to_bf16(rng.uniform(-1, 1, M * M).astype(np.float32))
    B = orc.f32_to_bf16(rng.uniform(-1, 1, M * M).astype(np.float32))
    hp = rt.unary_dispatch(28, BF16, M, M, M, M, 0)
    hg = rt.brgemm_dispatch(BF16, M, M, 128, M, M, M, 128, 128 * M, 4 | VB)
    dB, dBv, dC = dev(B), dev(np.zeros(M * M, np.uint16)), dev(np.zeros(M * M, np.uint16))
    rt.unary(BF16, hp, dB, 0, dBv, 0)
    rt.brgemm(BF16, hg, dev(A), 0, dBv, 0, dC, 0, 16)
    got = host(dC, A).reshape(M, M)
    for (r0, rr) in ((0, 32), (1000, 24), (2040, 8)):
        ref = np.zeros(rr * M, np.uint16)
        # flat B: batch b covers k in [128 b, 128 b + 128): stride_b = 128 rows of the flat matrix
        orc.brgemm(BF16, rr, M, 128, M, M, M, 128, 128 * M, 4, A, r0 * M, B, 0, ref, 0, 16)
        check_close(got[r0:r0 + rr].reshape(-1), ref, BF16, "C5 rows %d..%d" % (r0, r0 + rr))


@pytest.mark.parametrize("dt", [F32, BF16])
def test_softmax_tail_as_xsmm_calls(rt, dt):
    """of mlir-gen's softmax (tools/mlir-gen/MLIRGen.cpp:560-630: exp, row sum, splat, divide) the pieces
    that HAVE an xsmm kind at this revision are the splat of the [batch, 1] row sums (xsmm.unary identity, bcast_row,
    ldi = 1: ConvertLinalgToXsmm broadcast handling, linalg-to-unary.mlir:148-164) and the divide (xsmm.binary div) -
    plus the fused form div with bcast_row_in1, which needs no splat buffer; exp and the reduction have no
    xsmm.unary kind (XsmmEnum.td:34-45) and stay loops in the reference. Checked against numpy and the oracle."""
    rng = np.random.default_rng(21 + dt)
    m, n = 96, 200
    e = np.exp(rng.uniform(-2, 2, m * n)).astype(np.float32)
    rs = e.reshape(m, n).sum(axis=1).astype(np.float32)
    E, R = (e, rs) if dt == F32 else (orc.f32_to_bf16(e), orc.f32_to_bf16(rs))
    zeros = np.zeros(m * n, np.float32 if dt == F32 else np.uint16)
    dE, dR, dS, dO, dO2 = dev(E), dev(R), dev(zeros), dev(zeros), dev(zeros)
    hs = rt.unary_dispatch(1, dt, m, n, 1, n, 2)            # identity, bcast_row: [m,1] -> [m,n]
    hd = rt.binary_dispatch(4, dt, m, n, n, n, n, 0)        # div
    hd2 = rt.binary_dispatch(4, dt, m, n, n, 1, n, 2)       # div, bcast_row_in1 (no splat buffer)
    rt.unary(dt, hs, dR, 0, dS, 0)
    rt.binary(dt, hd, dE, 0, dS, 0, dO, 0)
    rt.binary(dt, hd2, dE, 0, dR, 0, dO2, 0)
    S, O, O2 = zeros.copy(), zeros.copy(), zeros.copy()
    orc.unary(1, dt, m, n, 1, n, 2, R, 0, S, 0)
    orc.binary(4, dt, m, n, n, n, n, 0, E, 0, S, 0, O, 0)
    orc.binary(4, dt, m, n, n, 1, n, 2, E, 0, R, 0, O2, 0)
    assert np.array_equal(host(dS, zeros), S)
    check_close(host(dO, zeros), O, dt, "softmax div")
    check_close(host(dO2, zeros), O2, dt, "softmax div bcast_row_in1")
    want = as_f32(E).reshape(m, n) / as_f32(R).reshape(m, 1)
    got = as_f32(host(dO2, zeros)).reshape(m, n)
    assert np.abs(got - want).max() <= (2e-7 if dt == F32 else 2.0 ** -8) * np.abs(want).max() * 4


def test_zz_report_elementwise_figure():
    """runs last in this file: how much of the element-wise bar |d| <= 1e-5 |ref| + K eps sum|a||b| the worst element of
    the whole session used (1.0 = at the bar)"""
    print("\n[parity] f32 element-wise bar used: max |gpu-ref| / (1e-5 |ref| + K eps sum|a||b|) = %.3g over %d GEMM cases" % (
        REL_STATS["max_rel"], REL_STATS["cases"]))
    print("[parity] against an fp64 truth (normwise, %d full-size f32 cases): |hip - f64| %.3g, |oracle - f64| %.3g" % (
        TRUTH_STATS["cases"], TRUTH_STATS["hip_vs_f64"], TRUTH_STATS["oracle_vs_f64"]))
    assert REL_STATS["max_rel"] <= 1.0

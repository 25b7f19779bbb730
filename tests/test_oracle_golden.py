"""The CPU oracle against every golden vector harvested from the reference's own lit
tests (tests/golden/harvest.py). This is what pins the oracle: parity of the HIP
path is then proven against the oracle (tests/test_parity_gpu.py)."""
import json
import os

import numpy as np
import pytest

import fixture_runner as fr
from oracle import pyoracle as orc


@pytest.mark.parametrize("path", fr.fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_oracle_matches_reference_golden(path):
    fr.run_fixture(path, fr.OracleBackend())


def test_fixture_count():
    # every reference test listed in SURVEY.md appendix C that is replayable at the ABI
    assert len(fr.fixtures()) >= 30


def test_tensor_init_stream_is_pinned():
    """oracle/tensor_init.cpp regenerates the seeded inputs stored in the fixtures
    (TensorInitFloat.cpp:85-95: one normal(0,0.2)-clamped stream per seed, shared
    across the arguments of one dtype)."""
    fx = fr.load(os.path.join(fr.GOLDEN, "xsmm_fusion_seed123.json"))
    gen = orc.TensorInit("normal", 123)
    a = gen.fill(64)
    bias = gen.fill(4)
    assert np.array_equal(a, np.array(fx["buffers"]["A"]["data"], dtype=np.float32))
    assert np.array_equal(bias, np.array(fx["buffers"]["bias"]["data"], dtype=np.float32))
    assert a.min() >= 0.0 and a.max() <= 1.0


def test_tensor_init_other_kinds():
    assert np.array_equal(orc.TensorInit("const", 0).fill(5), np.ones(5, np.float32))
    assert np.allclose(orc.TensorInit("simple", 0).fill(4), [0.3, 0.6, 0.9, 0.3])
    assert np.allclose(orc.TensorInit("cont", 0).fill(4), [0.0, 0.25, 0.5, 0.75])
    u = orc.TensorInit("random", 7).fill(1000)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.05


def test_bf16_rounding_is_rne():
    # 257 is the tie between 256 and 258 (xsmm-ternary-bf16.mlir:15-18) -> even = 256
    f = np.array([257.0, 259.0, 1.0, -0.0, 3.14159], dtype=np.float32)
    h = orc.f32_to_bf16(f)
    back = orc.bf16_to_f32(h)
    assert back[0] == 256.0 and back[1] == 260.0 and back[2] == 1.0
    for x, hh in zip(f, h):
        assert orc.lib().oracle_f32_to_bf16(float(x)) == int(hh)


def test_mlir_gen_flops_formula():
    """BENCH_TOTAL_FLOPS = sum over layers of 2MNK (+MN bias) (+MN relu), MLIRGen.cpp:313-334"""
    with open(os.path.join(fr.GOLDEN, "flops.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        m, total = c["batch"], 0
        for k, n in zip(c["layers"][:-1], c["layers"][1:]):
            total += 2 * m * n * k + (m * n if c["bias"] else 0) + (m * n if c["relu"] else 0)
        assert total == c["flops"], c


def test_oracle_vnni_factor_4_equals_the_flat_operand():
    """B as VNNI-4 [k/4][n][4] (MLIRGen.cpp:657-664; the factor comes from libxsmm_cpuid_dot_pack_factor in the
    reference, VNNIUtils.cpp:25-45 - here oracle_set_vnni_factor). The contraction walks k in order whatever the packing, so the
    result on the packed operand is the result on the flat operand bit for bit; factor 2 stays the default."""
    import numpy as np
    from oracle import pyoracle as orc
    rng = np.random.default_rng(0)
    m, n, k, br = 5, 6, 8, 3
    A = orc.f32_to_bf16(rng.uniform(-1, 1, br * m * k).astype(np.float32))
    Bf = orc.f32_to_bf16(rng.uniform(-1, 1, br * k * n).astype(np.float32))
    outs = {}
    for v in (0, 2, 4):
        if v:
            old = orc.set_vnni_factor(v)
            B = np.concatenate([orc.pack_vnni(Bf[b * k * n:(b + 1) * k * n], k, n, v) for b in range(br)])
        else:
            B = Bf
        C = np.zeros(m * n, np.uint16)
        orc.brgemm(2, m, n, k, k, n, n, m * k, k * n, 4 | (2048 if v else 0), A, 0, B, 0, C, 0, br)
        if v:
            orc.set_vnni_factor(old)
        outs[v] = C
    assert np.array_equal(outs[0], outs[2]) and np.array_equal(outs[0], outs[4])
    assert orc.set_vnni_factor(2) == 2
    # k must be a multiple of the factor
    old = orc.set_vnni_factor(4)
    try:
        rc = orc.lib().oracle_fused_brgemm(2, 2, 2, 6, 6, 2, 2, 12, 12, 4 | 2048, 0, 0, 0, 0, A.ctypes.data, Bf.ctypes.data, C.ctypes.data, None, 1)
        assert rc != 0
    finally:
        orc.set_vnni_factor(old)

"""A CHAIN of whole-layer f32 fused BRGEMMs in one launch (brgemm_f32_lw_chain, round 4): the reference's MLP benchmark
(mlir-gen --batch=256 --layers=1024,1024,1024,1024, fp32: benchmarks/config/base/base.json:74-80) hands its three layer calls to
xsmm_hip_fused_brgemm_chain_invoke. Through the C-ABI on a real MI355X:
  * BIT-IDENTICAL to the same calls one by one (the chain runs on the tile the calls were planned on, same order of additions),
    over several steps on the same buffers with changing inputs (a stale hand-off read would show the previous step's values);
  * every layer against the oracle, fed the GPU's own previous activations (one layer's error at a time; f32 bar of test_parity_gpu);
  * the conditions under which the call must run call by call: same results, return value 0.
"""
import importlib

import numpy as np
import pytest

from oracle import pyoracle as orc
from test_parity_gpu import F32, check_close, dev, host, rand

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("tpp-mlir_amd")

TILES = {6: (64, 64), 7: (64, 32)}  # forced variant -> tile of the f32 chain (64x64 + K2, 64x32 + K4)


@pytest.fixture(scope="module")
def rt():
    r = pkg.get_runtime()
    assert r.device_count() >= 1, "no HIP device visible: the gpu tests need an MI355X"
    return r


class Chain32:
    """whole-layer fused f32 BRGEMMs on `m` rows as mlir-gen emits them: layer l = [m x dims[l]] @ [dims[l] x dims[l+1]] in
    64-k batch elements, beta 0, + bias, relu"""

    def __init__(self, rt, m, dims, seed, bias=True, relu=True, force=None, pad=0, beta0=True):
        self.rt, self.m, self.dims, self.bias, self.relu, self.beta0 = rt, m, dims, bias, relu, beta0
        self.rng = np.random.default_rng(seed)
        self.L = len(dims) - 1
        self.ld = [d + pad for d in dims]
        self.W = [rand(self.rng, dims[l] * dims[l + 1], F32, -0.05, 0.05) for l in range(self.L)]
        self.b = [rand(self.rng, dims[l + 1], F32, -0.3, 0.3) for l in range(self.L)]
        self.handles = []
        if force is not None:
            rt.force_variant(force)
        try:
            for l in range(self.L):
                self.handles.append(rt.fused_brgemm_dispatch(*self.tuple(l)))
        finally:
            if force is not None:
                rt.force_variant(-1)
        self.dW, self.db = [dev(w) for w in self.W], [dev(b) for b in self.b]

    def tuple(self, l, m=None):
        n = self.dims[l + 1]
        return (F32, m or self.m, n, 64, self.ld[l], n, self.ld[l + 1], 64, 64 * n, 4 if self.beta0 else 0, 0, 5 if self.relu else 0,
                4 if self.bias else 0, 1 if self.bias else 0)

    def new_input(self):
        x = np.zeros(self.m * self.ld[0], np.float32)
        x.reshape(self.m, self.ld[0])[:, :self.dims[0]] = rand(self.rng, self.m * self.dims[0], F32).reshape(self.m, -1)
        return x

    def calls(self, dx, dacts):
        cur, out = dx, []
        for l in range(self.L):
            out.append((self.handles[l], cur, 0, self.dW[l], 0, dacts[l], 0, self.db[l], 0, self.dims[l] // 64))
            cur = dacts[l]
        return out

    def oracle_layer(self, l, inp, out):
        orc.fused_brgemm(*self.tuple(l), inp, 0, self.W[l], 0, out, 0, self.b[l], 0, self.dims[l] // 64)


def run_steps(rt, ch, steps, expect_fused=True):
    import torch
    m = ch.m
    start = [np.full(m * ch.ld[l + 1], np.nan if ch.beta0 else 0.25, np.float32) for l in range(ch.L)]  # NaN: an unwritten element cannot pass
    dacts_f = [dev(a) for a in start]
    dacts_s = [dev(a) for a in start]
    was_async = rt.set_async(True)
    try:
        for step in range(steps):
            x = ch.new_input()
            dx = dev(x)
            if not ch.beta0:
                for l in range(ch.L):
                    dacts_f[l].copy_(dev(start[l]))
                    dacts_s[l].copy_(dev(start[l]))
            fused = rt.fused_brgemm_chain(F32, ch.calls(dx, dacts_f))
            assert bool(fused) == expect_fused, "chain ran as one launch: %s, expected %s" % (bool(fused), expect_fused)
            for c in ch.calls(dx, dacts_s):
                rt.fused_brgemm(F32, *c)
            rt.synchronize()
            for l in range(ch.L):
                assert torch.equal(dacts_f[l].view(torch.int32), dacts_s[l].view(torch.int32)), \
                    "step %d layer %d: the chain differs from the separate launches" % (step, l)
            if step in (0, steps - 1):
                prev = x
                for l in range(ch.L):
                    got = host(dacts_f[l], x)
                    ref = start[l].copy()
                    ch.oracle_layer(l, prev, ref)
                    n, ld = ch.dims[l + 1], ch.ld[l + 1]
                    sel = (np.arange(m)[:, None] * ld + np.arange(n)[None, :]).reshape(-1)
                    check_close(got[sel], ref[sel], F32, "f32 chain layer %d" % l)
                    prev = got  # the next layer's oracle reads the GPU's own activations
    finally:
        rt.synchronize()
        rt.set_async(was_async)


@pytest.mark.parametrize("variant,m,dims", [
    (7, 64, [64, 64, 64]),                  # ONE chunk per layer (T = 1: no mid-chunk barrier at all)
    (7, 128, [256, 256, 256]),
    (7, 192, [192, 320, 320, 320]),         # three chunks, then five: around the ring depth
    (7, 512, [1024, 1024, 1024, 1024]),     # C3-shaped layers: 256 tiles of 64x32 = one per CU
    (6, 64, [128, 128, 128]),
    (6, 128, [128, 256, 256, 256]),
    (6, 1024, [512, 1024, 1024]),           # 256 tiles of 64x64, two loader waves per panel
], ids=lambda v: str(v).replace(" ", ""))
def test_f32_chain_against_oracle_and_separate_launches(rt, variant, m, dims):
    bm, bn = TILES[variant]
    assert m % bm == 0 and dims[1] % bn == 0
    ch = Chain32(rt, m, dims, seed=variant + m, force=variant)
    run_steps(rt, ch, 3)


def test_f32_chain_default_dispatch(rt):
    """no forced tile: the planner puts 512 x 1024 x 1024 layers on 64x32 + K4 (256 tiles) and the chain takes them; the reference's
    batch-256 layers (32x32 + K4) are deliberately NOT chained (measured slower than three launches) - same results, call by call"""
    ch = Chain32(rt, 512, [1024, 1024, 1024, 1024], seed=5)
    assert "64x32" in rt.kernel_name(ch.handles[0]), rt.kernel_name(ch.handles[0])
    run_steps(rt, ch, 4)
    small = Chain32(rt, 256, [1024, 1024, 1024, 1024], seed=6)
    assert "32x32" in rt.kernel_name(small.handles[0]), rt.kernel_name(small.handles[0])
    run_steps(rt, small, 2, expect_fused=False)


def test_f32_chain_no_bias_no_relu_and_padded_rows(rt):
    run_steps(rt, Chain32(rt, 128, [256, 256, 256], seed=11, bias=False, relu=False, force=7, pad=16), 2)


def test_f32_chain_falls_back_when_it_must(rt):
    # beta = 1 layers accumulate into C: call by call
    run_steps(rt, Chain32(rt, 64, [128, 128, 128], seed=21, force=7, beta0=False), 2, expect_fused=False)
    # more tiles than compute units (2048 rows of 64x64 tiles x 16 column tiles)
    run_steps(rt, Chain32(rt, 2048, [128, 1024, 1024], seed=22), 1, expect_fused=False)

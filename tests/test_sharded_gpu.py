"""The multi-GPU shardings of tpp-mlir_amd/mlp.py ON THE HIP KERNELS, on one GPU: every rank's share of
ShardedMlp (row blocks) and of ColumnShardedMlp (column blocks + an all-gather after every
layer; the next layer batch-reduces over the rank blocks) runs through the C-ABI for world in {2, 4, 8};
the all-gather is emulated by device copies into the collective's layout. The dispatch shapes these
shardings produce (B offset 2*n0 into a VNNI-2 matrix with ldb = n, ldc = n/W, br = W, stride_a = batch*k/W)
are thereby checked against (a) the unsharded HIP result and (b) the oracle on sampled rows (rows are
independent). Plus: all_gather_rows on device tensors through RCCL with one rank.
Different tile families sum in different orders, so HIP-vs-HIP is compared with the bf16 bar of
test_parity_gpu (one bf16 ulp + the f32 accumulation floor, per layer: three layers -> 3 ulp)."""
import importlib
import socket
import threading

import numpy as np
import pytest

from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("tpp-mlir_amd")
BF16 = 2


@pytest.fixture(scope="module")
def rt():
    r = pkg.get_runtime()
    assert r.device_count() >= 1, "no HIP device visible: the gpu tests need an MI355X"
    return r


def bits(t):
    """torch bf16 tensor -> numpy uint16 (flat)"""
    import torch
    return t.detach().cpu().view(torch.int16).numpy().reshape(-1).view(np.uint16)


def problem(spec, rt, seed=3):
    """X, flat weights, VNNI-2 weights (packed by the runtime's own unary op), biases - all on the device"""
    import torch
    g = torch.Generator().manual_seed(seed)
    X = (torch.randn(spec.batch, spec.layers[0], generator=g) * 0.5).to(torch.bfloat16).cuda()
    Wf, Wv, Bs = [], [], []
    for k, n in zip(spec.layers[:-1], spec.layers[1:]):
        w = (torch.randn(k, n, generator=g) * 0.06).to(torch.bfloat16).cuda()
        v = torch.empty_like(w)
        rt.unary(BF16, rt.unary_dispatch(pkg.UnaryKind.VNNI2, BF16, k, n, n, n, 0), w, 0, v, 0)
        Wf.append(w)
        Wv.append(v)
        Bs.append((torch.randn(n, generator=g) * 0.1).to(torch.bfloat16).cuda())
    return X, Wf, Wv, Bs


def oracle_rows(spec, X, Wf, Bs, r0, rr):
    """the MLP on rows [r0, r0 + rr) by the oracle (whole-layer dispatch arguments, k chunks of 64)"""
    cur = bits(X[r0:r0 + rr].contiguous()).copy()
    for (k, n), w, b in zip(zip(spec.layers[:-1], spec.layers[1:]), Wf, Bs):
        wv = np.empty(k * n, np.uint16)
        orc.unary(28, BF16, k, n, n, n, 0, bits(w), 0, wv, 0)
        out = np.zeros(rr * n, np.uint16)
        orc.fused_brgemm(BF16, rr, n, 64, k, n, n, 64, 64 * n, 4 | 2048, 0, 5, 4, 1, cur, 0, wv, 0, out, 0, bits(b), 0, k // 64)
        cur = out
    return cur


def close_bf16(got, ref, ulps, what):
    """bar for THREE chained bf16 layers: `ulps` bf16 ulps of the value plus an absolute floor of half an ulp of the
    largest activation - a one-ulp rounding flip in a layer's output (legitimate: the two sides sum in different
    orders) moves the next layer's pre-activations by about that much, also where relu clips them to exactly 0.
    (One layer against the oracle is held to one ulp in test_parity_gpu.)"""
    g, r = orc.bf16_to_f32(got).astype(np.float64), orc.bf16_to_f32(ref).astype(np.float64)
    tol = ulps * np.abs(r) * 2.0 ** -7 + 2.0 ** -8 * max(1.0, float(np.abs(r).max()))
    bad = np.abs(g - r) > tol
    assert not bad.any(), "%s: %d/%d mismatches, max abs diff %g" % (what, int(bad.sum()), bad.size, float(np.abs(g - r).max()))
    return float((got == ref).mean())


SPEC = dict(batch=4096, layers=[1024, 1024, 1024, 1024])  # BASELINE config C4
SAMPLES = [(0, 8), (1020, 8), (2047, 6), (4090, 6)]


@pytest.fixture(scope="module")
def unsharded(rt):
    import torch
    spec = pkg.MlpSpec(**SPEC)
    X, Wf, Wv, Bs = problem(spec, rt)
    rt.set_async(False)
    one = pkg.ShardedMlp(spec, 0, 1, rt)
    acts = [torch.zeros(spec.batch, n, dtype=torch.bfloat16, device="cuda") for n in spec.layers[1:]]
    out = one.forward(X, Wv, Bs, acts)
    torch.cuda.synchronize()
    ref = bits(out).copy()
    N = spec.layers[-1]
    for r0, rr in SAMPLES:  # the unsharded HIP result itself against the oracle
        close_bf16(ref[r0 * N:(r0 + rr) * N], oracle_rows(spec, X, Wf, Bs, r0, rr), 3, "unsharded rows %d.." % r0)
    return spec, X, Wf, Wv, Bs, ref


@pytest.mark.parametrize("world", [2, 4, 8])
def test_row_sharded_mlp_on_hip_kernels(rt, unsharded, world):
    import torch
    spec, X, Wf, Wv, Bs, ref = unsharded
    N = spec.layers[-1]
    full = torch.zeros(spec.batch, N, dtype=torch.bfloat16, device="cuda")
    names = set()
    for rank in range(world):
        sh = pkg.ShardedMlp(spec, rank, world, rt)
        acts = [torch.zeros(sh.rows, n, dtype=torch.bfloat16, device="cuda") for n in spec.layers[1:]]
        out = sh.forward(X[sh.row0: sh.row0 + sh.rows].contiguous(), Wv, Bs, acts)
        full[sh.row0: sh.row0 + sh.rows].copy_(out)  # = this rank's block of the all-gather
        names.add(rt.kernel_name(sh.handles[0][0]))
    torch.cuda.synchronize()
    got = bits(full)
    exact = close_bf16(got, ref, 3, "row-sharded world %d vs unsharded [%s]" % (world, ",".join(sorted(names))))
    for r0, rr in SAMPLES:
        close_bf16(got[r0 * N:(r0 + rr) * N], oracle_rows(spec, X, Wf, Bs, r0, rr), 3, "row-sharded rows %d.." % r0)
    assert exact > 0.9, exact  # same arithmetic; only summation-order rounding flips may differ


@pytest.mark.parametrize("world", [2, 4, 8])
def test_column_sharded_mlp_on_hip_kernels(rt, unsharded, world):
    """ColumnShardedMlp.forward of every rank on its own thread; the collective is a device copy of the rank's
    block into the gathered [W][batch][N/W] buffer followed by a barrier between the threads"""
    import torch
    spec, X, Wf, Wv, Bs, ref = unsharded
    N = spec.layers[-1]
    nw = N // world
    gathered = [torch.zeros(world, spec.batch, n // world, dtype=torch.bfloat16, device="cuda") for n in spec.layers[1:]]
    meet = threading.Barrier(world)
    errors, names = [], set()

    def rank_main(rank):
        try:
            cs = pkg.ColumnShardedMlp(spec, rank, world, rt)
            names.add(rt.kernel_name(cs.handles[1][0]))
            loc = [torch.zeros(spec.batch, n // world, dtype=torch.bfloat16, device="cuda") for n in spec.layers[1:]]

            def all_gather(dst, src):
                dst[rank].copy_(src)
                torch.cuda.synchronize()
                meet.wait()

            cs.forward(X, Wv, Bs, loc, gathered, all_gather)
        except Exception as ex:  # noqa: BLE001
            errors.append((rank, repr(ex)))
            meet.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    got = bits(pkg.gathered_to_rows(gathered[-1]).contiguous())
    assert gathered[-1].shape == (world, spec.batch, nw)
    exact = close_bf16(got, ref, 3, "column-sharded world %d vs unsharded [%s]" % (world, ",".join(sorted(names))))
    for r0, rr in SAMPLES:
        close_bf16(got[r0 * N:(r0 + rr) * N], oracle_rows(spec, X, Wf, Bs, r0, rr), 3, "column-sharded rows %d.." % r0)
    assert exact > 0.9, exact


def test_all_gather_rows_on_rccl_world_1(rt):
    """the collective call of the row sharding on device tensors through RCCL (backend nccl), one rank"""
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        spec = pkg.MlpSpec(batch=512, layers=[64, 256])
        out = (torch.randn(512, 256) * 0.5).to(torch.bfloat16).cuda()
        full = torch.zeros(512, 256, dtype=torch.bfloat16, device="cuda")
        pkg.all_gather_rows(out, full, spec, 1)
        torch.cuda.synchronize()
        assert torch.equal(full.view(torch.int16), out.view(torch.int16))
    finally:
        dist.destroy_process_group()

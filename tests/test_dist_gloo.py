"""The N>1 path on CPU: world_size-2 gloo processes run the row-sharded MLP (each rank
its row block, all layers) and the all-gather of the output; the result must equal the
unsharded computation bit for bit. The GPU kernels are replaced by the oracle here (this
is a tests/ file) - what is under test is the partition + gather logic of
tpp-mlir_amd/mlp.py, which is identical on RCCL."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pyoracle as orc

pkg = importlib.import_module("tpp-mlir_amd")


def np_view(t):
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().reshape(-1).view(np.uint16)
    return t.numpy().reshape(-1)


class OracleRuntime:
    """same two methods as XsmmRuntime, backed by oracle/xsmm_oracle.c on CPU tensors"""

    def __init__(self):
        self.descs = []

    def fused_brgemm_dispatch(self, **kw):
        self.descs.append(kw)
        return len(self.descs)

    def fused_brgemm(self, dtype, h, a, oa, b, ob, c, oc, d, od, br):
        k = self.descs[h - 1]
        orc.fused_brgemm(dtype, k["m"], k["n"], k["k"], k["lda"], k["ldb"], k["ldc"], k["stride_a"], k["stride_b"],
                         k["gemm_flags"], k["unary_flags"], k["unary_kind"], k["binary_flags"], k["binary_kind"],
                         np_view(a), oa, np_view(b), ob, np_view(c), oc, np_view(d), od, br)


def make_problem(spec):
    g = torch.Generator().manual_seed(3)
    tdt = torch.bfloat16 if spec.dtype == 2 else torch.float32
    X = (torch.randn(spec.batch, spec.layers[0], generator=g) * 0.5).to(tdt)
    Ws, Bs = [], []
    for k, n in zip(spec.layers[:-1], spec.layers[1:]):
        w = (torch.randn(k, n, generator=g) * 0.1).to(tdt)
        if spec.dtype == 2:  # VNNI-2 pack [k/2][n][2] with the oracle's pack op
            packed = torch.empty_like(w)
            orc.unary(28, 2, k, n, n, n, 0, np_view(w), 0, np_view(packed), 0)
            w = packed
        Ws.append(w)
        Bs.append((torch.randn(n, generator=g) * 0.1).to(tdt))
    return X, Ws, Bs


def run_rank(rank, world, port, batch, dtype, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = pkg.MlpSpec(batch=batch, layers=[128, 192, 128], dtype=dtype)
        X, Ws, Bs = make_problem(spec)
        sh = pkg.ShardedMlp(spec, rank, world, OracleRuntime())
        xl = X[sh.row0: sh.row0 + sh.rows].contiguous()
        acts = [torch.zeros(sh.rows, n, dtype=X.dtype) for n in spec.layers[1:]]
        out = sh.forward(xl, Ws, Bs, acts)
        if out is None:
            out = torch.zeros(0, spec.layers[-1], dtype=X.dtype)
        full = torch.zeros(spec.batch, spec.layers[-1], dtype=X.dtype)
        pkg.all_gather_rows(out, full, spec, world)
        # unsharded reference on every rank
        one = pkg.ShardedMlp(spec, 0, 1, OracleRuntime())
        acts1 = [torch.zeros(spec.batch, n, dtype=X.dtype) for n in spec.layers[1:]]
        ref = one.forward(X, Ws, Bs, acts1)
        ok = torch.equal(full.view(torch.int16) if dtype == 2 else full, ref.view(torch.int16) if dtype == 2 else ref)
        out_q.put((rank, bool(ok), sh.row0, sh.rows))
    finally:
        dist.destroy_process_group()


def run_rank_columns(rank, world, port, batch, dtype, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = pkg.MlpSpec(batch=batch, layers=[128, 256, 128, 256], dtype=dtype)
        X, Ws, Bs = make_problem(spec)
        cs = pkg.ColumnShardedMlp(spec, rank, world, OracleRuntime())
        locals_ = [torch.zeros(batch, n // world, dtype=X.dtype) for n in spec.layers[1:]]
        gathered = [torch.zeros(world, batch, n // world, dtype=X.dtype) for n in spec.layers[1:]]
        out = cs.forward(X, Ws, Bs, locals_, gathered, lambda dst, src: dist.all_gather_into_tensor(dst.view(-1, dst.shape[-1]), src))
        full = pkg.gathered_to_rows(out).contiguous()
        one = pkg.ShardedMlp(spec, 0, 1, OracleRuntime())
        acts1 = [torch.zeros(spec.batch, n, dtype=X.dtype) for n in spec.layers[1:]]
        ref = one.forward(X, Ws, Bs, acts1)
        ok = torch.equal(full.view(torch.int16) if dtype == 2 else full, ref.view(torch.int16) if dtype == 2 else ref)
        out_q.put((rank, bool(ok), len(cs.handles), cs.handles[1][1]))
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("batch,dtype", [(512, 1), (512, 2), (384, 2)])
def test_row_sharded_mlp_with_all_gather_world2(batch, dtype):
    orc.lib()  # make sure the oracle is built before forking
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=run_rank, args=(r, world, port, batch, dtype, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for (_, ok, _, _) in res), res
    rows = sorted((r0, n) for (_, _, r0, n) in res)
    assert rows[0][0] == 0 and rows[0][0] + rows[0][1] == rows[1][0] and rows[1][0] + rows[1][1] == batch


@pytest.mark.parametrize("batch,dtype", [(96, 1), (64, 2)])
def test_column_sharded_mlp_gather_per_layer_world2(batch, dtype):
    """the per-layer all-gather variant: rank-major gathered activations are consumed by the next layer
    as a batch-reduce over the rank blocks (br = world); result identical to the unsharded MLP"""
    orc.lib()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=run_rank_columns, args=(r, world, port, batch, dtype, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for (_, ok, _, _) in res), res
    assert all(nl == 3 and br == world for (_, _, nl, br) in res)


# ---- PeerGather.create: the fallback protocol (a rank that failed before the barrier left the others hanging) ------
class _FakePeerLib:
    """the five entry points PeerGather's set-up calls, with a fault injected on one rank (no GPU involved)"""

    def __init__(self, rank, fail_rank, fail_at):
        self.rank, self.fail_rank, self.fail_at = rank, fail_rank, fail_at
        self.next, self.freed, self.closed = 0x1000, [], []

    def xsmm_hip_peer_alloc(self, nbytes):
        if self.rank == self.fail_rank and self.fail_at == "alloc":
            return 0
        self.next += 0x100000
        return self.next

    def xsmm_hip_ipc_export(self, ptr, out):
        return -1 if (self.rank == self.fail_rank and self.fail_at == "export") else 0

    def xsmm_hip_ipc_open(self, handle):
        if self.rank == self.fail_rank and self.fail_at == "open":
            return 0
        self.next += 0x100000
        return self.next

    def xsmm_hip_ipc_close(self, ptr):
        self.closed.append(ptr)
        return 0

    def xsmm_hip_peer_free(self, ptr):
        self.freed.append(ptr)


class _FakePeerRt:
    def __init__(self, lib):
        self.lib = lib

    def synchronize(self):
        pass


def run_rank_peer_fallback(rank, world, port, fail_at, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = _FakePeerLib(rank, 1, fail_at)
        pg = pkg.PeerGather.create(_FakePeerRt(lib), rank, world, 1 << 20)
        # the collectives of both ranks still pair up: this all_reduce would hang or mismatch otherwise
        t = torch.tensor([rank + 1])
        dist.all_reduce(t)
        out_q.put((rank, pg is None, int(t[0]), len(lib.freed), len(lib.closed)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail_at", ["alloc", "export", "open"])
def test_peer_gather_falls_back_on_every_rank_when_one_rank_fails(fail_at):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=run_rank_peer_fallback, args=(r, world, port, fail_at, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for (rank, none, total, freed, closed) in res:
        assert none, "rank %d kept a peer-gather object although rank 1 failed (%s)" % (rank, fail_at)
        assert total == 3, "the collectives after the fallback did not pair up"
    # whatever a rank did allocate or map has been released again (close() leaked the rank's own buffers)
    assert res[0][3] == 3  # rank 0: its three allocations freed
    if fail_at == "open":
        assert res[1][3] == 3

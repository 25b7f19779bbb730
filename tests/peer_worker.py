"""worker of tests/test_peer_gather_gpu.py: one of WORLD processes that ALL use cuda:0 (two ranks on one device exchange
hipIpcMemHandles exactly as two ranks on two GPUs of a node would; the peer mappings then alias the same HBM). Each rank runs
its row share of a small bf16 MLP (one chain launch) and the peer-store gather for several steps, and checks the gathered
output bit for bit against (a) a gloo all_gather of the same local blocks and (b) the unsharded MLP run in this process."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    pkg = importlib.import_module("tpp-mlir_amd")
    from oracle import pyoracle as orc
    rt = pkg.get_runtime()
    rt.set_async(True)
    batch, N, L = 128 * world * 2, 256, 3
    spec = pkg.MlpSpec(batch=batch, layers=[N] * (L + 1))
    rt.force_variant(21)  # 64x64 tiles for the shard AND the unsharded run: identical arithmetic per element
    mine = pkg.ShardedMlp(spec, rank, world, rt)
    whole = pkg.ShardedMlp(spec, 0, 1, rt)
    rt.force_variant(-1)
    rng = np.random.default_rng(5)  # the same weights on every rank
    dev = lambda a: torch.from_numpy(a.view(np.int16).copy()).cuda()  # noqa: E731
    W = [dev(orc.f32_to_bf16(rng.uniform(-0.2, 0.2, N * N).astype(np.float32))) for _ in range(L)]
    B = [dev(orc.f32_to_bf16(rng.uniform(-1, 1, N).astype(np.float32))) for _ in range(L)]
    pg = pkg.PeerGather.create(rt, rank, world, batch * N * 2)
    assert pg is not None, "the peer-store gather could not be set up"
    acts = [torch.zeros(mine.rows * N, dtype=torch.int16, device="cuda") for _ in range(L)]
    wacts = [torch.zeros(batch * N, dtype=torch.int16, device="cuda") for _ in range(L)]
    for step in range(6):
        X = dev(orc.f32_to_bf16(np.random.default_rng(100 + step).uniform(-1, 1, batch * N).astype(np.float32)))  # same on every rank
        out = mine.forward(X[mine.row0 * N:(mine.row0 + mine.rows) * N], W, B, acts)
        full = pg.gather(out, mine.rows * N * 2, mine.row0 * N * 2)
        ref = whole.forward(X, W, B, wacts)
        rt.synchronize()
        pg.check()
        assert mine.last_step_fused and whole.last_step_fused
        assert torch.equal(full, ref), "step %d rank %d: gathered output differs from the unsharded result" % (step, rank)
        parts = [torch.empty(mine.rows * N // 2, dtype=torch.int32) for _ in range(world)]  # (gloo has no int16: pairs as int32)
        dist.all_gather(parts, out.cpu().view(torch.int32))
        assert torch.equal(full.cpu().view(torch.int32), torch.cat(parts)), "step %d rank %d: gathered output differs from the gloo all_gather" % (step, rank)
    # OVERLAP mode: the wait kernels on a side stream. (a) step by step; (b) a pipelined burst - no host synchronisation between
    # the steps, output e is consumed (cloned on the compute stream behind an event of the side stream) after step e+1 has been
    # computed and before it is gathered, which is the ordering the mode asks of a consumer
    pg.overlap(True)
    side = torch.cuda.ExternalStream(rt.lib.xsmm_hip_peer_wait_stream())
    for step in range(6, 9):
        X = dev(orc.f32_to_bf16(np.random.default_rng(100 + step).uniform(-1, 1, batch * N).astype(np.float32)))
        out = mine.forward(X[mine.row0 * N:(mine.row0 + mine.rows) * N], W, B, acts)
        full = pg.gather(out, mine.rows * N * 2, mine.row0 * N * 2)
        ref = whole.forward(X, W, B, wacts)
        rt.synchronize()
        pg.check()
        assert torch.equal(full, ref), "overlap step %d rank %d: gathered output differs from the unsharded result" % (step, rank)
    Xs = [dev(orc.f32_to_bf16(np.random.default_rng(200 + i).uniform(-1, 1, batch * N).astype(np.float32))) for i in range(7)]
    torch.cuda.synchronize()
    got, prev_full = [], None
    for i in range(7):
        out = mine.forward(Xs[i][mine.row0 * N:(mine.row0 + mine.rows) * N], W, B, acts)
        if prev_full is not None:
            ev = torch.cuda.Event()
            ev.record(side)
            torch.cuda.current_stream().wait_event(ev)
            got.append(prev_full.clone())
        prev_full = pg.gather(out, mine.rows * N * 2, mine.row0 * N * 2)
    rt.synchronize()
    pg.check()
    got.append(prev_full.clone())
    torch.cuda.synchronize()
    for i in range(7):
        ref = whole.forward(Xs[i], W, B, wacts)
        rt.synchronize()
        assert torch.equal(got[i], ref), "pipelined step %d rank %d: gathered output differs from the unsharded result" % (i, rank)
    pg.overlap(False)
    # the price of the gather itself: 200 steps of a 1 MiB block per rank, nothing else on the stream
    blk = torch.zeros(512 * 1024, dtype=torch.int16, device="cuda")
    pg2 = pkg.PeerGather.create(rt, rank, world, world * blk.numel() * 2)
    dist.barrier()
    for it in range(2):
        rt.synchronize()
        t0 = __import__("time").perf_counter()
        for _ in range(200):
            pg2.gather(blk, blk.numel() * 2, rank * blk.numel() * 2)
        rt.synchronize()
        dt = (__import__("time").perf_counter() - t0) / 200
    pg2.check()
    print("peer_worker rank %d: gather of 1 MiB per rank, %d ranks on one device: %.2f us per step" % (rank, world, dt * 1e6), flush=True)
    dist.barrier()
    pg2.close()
    pg.close()
    print("peer_worker rank %d OK" % rank, flush=True)


if __name__ == "__main__":
    main()

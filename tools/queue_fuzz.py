#!/usr/bin/env python3
"""Randomised stress of the tile queue (trace cache, scheduler, several callers): random programs of dependent 32x32 tile
ops over a few device buffers are run repeatedly - from 1 to 6 caller threads with a barrier after every phase, with
random synchronisation points, with mutations between repetitions - once through the queue and once with the queue off
(every invoke its own launch, program order); the buffers must come out bit-identical. Measurement / assurance aid,
not a test: `python tools/queue_fuzz.py [seconds] [seed]`."""
import importlib
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tpp-mlir_amd")
rt = pkg.get_runtime()
F32 = 1
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
NBUF, NTILE = 5, 64
copy = rt.unary_dispatch(1, F32, 32, 32, 32, 32, 0)
relu = rt.unary_dispatch(5, F32, 32, 32, 32, 32, 0)
zero = rt.unary_dispatch(2, F32, 32, 32, 32, 32, 0)
add = rt.binary_dispatch(1, F32, 32, 32, 32, 32, 32, 0)
mul = rt.binary_dispatch(2, F32, 32, 32, 32, 32, 32, 0)
gemm = rt.brgemm_dispatch(F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 0)
gemm0 = rt.brgemm_dispatch(F32, 32, 32, 32, 32, 32, 32, 1024, 1024, 4)
# transposes feeding a gemm's B operand through a temporary (kinds 7, 8): the runtime folds them into the gemm (deferred transposes,
# runtime.cpp: one remembered transpose per calling thread) - the queued run reads B from the transpose's source, the unqueued run
# from the temporary
transp = rt.unary_dispatch(29, F32, 32, 32, 32, 32, 0)
gemm1 = rt.gemm_dispatch(F32, 32, 32, 32, 32, 32, 32, 4)
tls = threading.local()


BR = [1]


def issue(op, bufs):
    kind, s, st, s2, s2t, d, dt = op[:7]
    S, S2, D = bufs[s], bufs[s2], bufs[d]
    if kind in (7, 8):
        # 7: the temporary is the op's own tile of a fourth buffer (every transpose is launched: the next one goes elsewhere);
        # 8: ONE temporary per calling thread in a scratch buffer that is not compared (transposes are dropped as dead)
        X, xt = (bufs[op[7]], dt) if kind == 7 else (bufs[NBUF], getattr(tls, "tid", 0))
        rt.unary(F32, transp, S, st * 1024, X, xt * 1024)
        rt.gemm(F32, gemm1, S2, s2t * 1024, X, xt * 1024, D, dt * 1024)
        return
    if kind == 0:
        rt.unary(F32, copy, S, st * 1024, D, dt * 1024)
    elif kind == 1:
        rt.unary(F32, relu, S, st * 1024, D, dt * 1024)
    elif kind == 2:
        rt.unary(F32, zero, D, dt * 1024, D, dt * 1024)
    elif kind == 3:
        rt.binary(F32, add, S, st * 1024, S2, s2t * 1024, D, dt * 1024)
    elif kind == 4:
        rt.binary(F32, mul, S, st * 1024, S2, s2t * 1024, D, dt * 1024)
    elif d in (s, s2):  # a gemm must not write a buffer it reads
        rt.binary(F32, add, S, st * 1024, S2, s2t * 1024, D, dt * 1024)
    else:
        # batch count of the program (1, 2 or 4; even counts run on the loader-wave pair kernel, queued or not)
        rt.brgemm(F32, gemm if kind == 5 else gemm0, S, min(st, NTILE - BR[0]) * 1024, S2, min(s2t, NTILE - BR[0]) * 1024, D, dt * 1024, BR[0])


def make_program(rng):
    """phases; inside a phase the ops are independent of each other BY CONSTRUCTION when run by several threads (each
    op writes its own tile of a buffer no op of the phase reads), so any interleaving gives the same result"""
    phases = []
    for _ in range(int(rng.integers(2, 7))):
        kind = int(rng.integers(0, 9))
        perm = rng.permutation(NBUF)
        s, s2, d = int(perm[0]), int(perm[1]), int(perm[2])
        n = int(rng.integers(4, NTILE))
        tiles = rng.permutation(NTILE)[:n]
        phases.append([(kind, s, int(rng.integers(0, NTILE)), s2, int(rng.integers(0, NTILE)), d, int(t), int(perm[3])) for t in tiles])
    return phases


def run(phases, bufs, nthr, sync_after):
    if nthr == 1:
        for pi, ph in enumerate(phases):
            for op in ph:
                issue(op, bufs)
            if pi in sync_after:
                rt.synchronize()
        return
    barrier = threading.Barrier(nthr)

    def worker(tid):
        tls.tid = tid
        for pi, ph in enumerate(phases):
            for op in ph[tid::nthr]:
                issue(op, bufs)
            barrier.wait()
            if pi in sync_after and tid == 0:
                rt.synchronize()
            if pi in sync_after:
                barrier.wait()
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthr)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()


t_end = time.time() + budget
rounds = ops = 0
seed = seed0
stats0 = rt.tile_queue_stats()
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    init = [(rng.uniform(-1, 1, NTILE * 1024) * 0.2).astype(np.float32) for _ in range(NBUF)]
    phases = make_program(rng)
    BR[0] = int(rng.choice([1, 2, 4]))
    q_bufs = [torch.from_numpy(b.copy()).cuda() for b in init] + [torch.zeros(8 * 1024, dtype=torch.float32, device="cuda")]
    r_bufs = [torch.from_numpy(b.copy()).cuda() for b in init] + [torch.zeros(8 * 1024, dtype=torch.float32, device="cuda")]
    for rep in range(int(rng.integers(2, 6))):
        nthr = int(rng.choice([1, 1, 2, 3, 4, 6]))
        sync_after = set(int(x) for x in rng.integers(0, len(phases), int(rng.integers(0, 2))))
        rt.set_async(True)
        rt.set_tile_queue(True)
        run(phases, q_bufs, nthr, sync_after)
        rt.synchronize()
        rt.set_tile_queue(False)
        run(phases, r_bufs, 1, set())
        rt.synchronize()
        for b in range(NBUF):
            if not torch.equal(q_bufs[b].view(torch.int32), r_bufs[b].view(torch.int32)):
                print("MISMATCH seed %d rep %d buffer %d threads %d" % (seed, rep, b, nthr), flush=True)
                sys.exit(1)
        ops += sum(len(p) for p in phases)
        if rng.integers(0, 2):  # mutate between repetitions
            pi = int(rng.integers(0, len(phases)))
            if len(phases[pi]) > 2 and rng.integers(0, 2):
                del phases[pi][int(rng.integers(0, len(phases[pi])))]
            else:
                k, s, st, s2, s2t, d, dt, x = phases[pi][0]
                phases[pi] = [(k, s, int(rng.integers(0, NTILE)), s2, s2t, d, t[6], x) for t in phases[pi]]
    rounds += 1
    seed += 1
st = tuple(b - a for a, b in zip(stats0, rt.tile_queue_stats()))
print("queue_fuzz: %d programs (%d invokes through the queue) identical to the unqueued runs; queue launches %d, full bookkeeping %d, "
      "replayed %d, terminated %d, abandoned %d; transposes: %d gemms served from a transpose's source, %d dropped as dead, %d launched late"
      % ((rounds, ops) + st + rt.fold_transpose_stats()))

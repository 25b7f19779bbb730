#!/usr/bin/env python3
"""where the fixed cost of a short timed region goes: host-side time of every call in bench.py's timed()
around K = 20 launches of the C2 kernel (measurement aid, not a test)"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tpp-mlir_amd"); rt = pkg.get_runtime(); rt.set_async(True)
m = n = 1024; k, br = 64, 16
A = torch.rand(m, 1024, device="cuda") * 2 - 1; B = torch.rand(1024, n, device="cuda") * 2 - 1; C = torch.zeros(m, n, device="cuda")
h = rt.brgemm_dispatch(1, m, n, k, 1024, 1024, 1024, 64, 65536, 4)
def step(): rt.brgemm(1, h, A, 0, B, 0, C, 0, br)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1:
    for _ in range(200): step()
    torch.cuda.synchronize()
pc = time.perf_counter
def run(K, sync_kind):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ts = [pc()]
    e0.record(); ts.append(pc())
    step(); ts.append(pc())
    for _ in range(K - 1): step()
    ts.append(pc())
    e1.record(); ts.append(pc())
    while not e1.query(): pass
    ts.append(pc())
    if sync_kind == "device": torch.cuda.synchronize()
    elif sync_kind == "stream": torch.cuda.current_stream().synchronize()
    elif sync_kind == "rt": rt.synchronize()
    ts.append(pc())
    d = [(b - a) * 1e6 for a, b in zip(ts, ts[1:])]
    return d, e0.elapsed_time(e1) * 1e3, (ts[-1] - ts[0]) * 1e6
for sync_kind in ("device", "stream", "rt", "none"):
    for rep in range(3):
        d, dev, wall = run(20, sync_kind)
        print("%-7s e0.record %.1f | first launch %.1f | 19 launches %.1f | e1.record %.1f | spin %.1f | sync %.1f || device %.1f wall %.1f diff %.1f us" % (
            sync_kind, *d, dev, wall, wall - dev))
# without events at all
for rep in range(3):
    torch.cuda.synchronize(); t = pc()
    for _ in range(20): step()
    torch.cuda.synchronize(); w = (pc() - t) * 1e6
    print("no events, device sync only: wall %.1f us (%.2f us/step)" % (w, w / 20))
for rep in range(3):
    torch.cuda.synchronize(); t = pc()
    for _ in range(20): step()
    rt.synchronize(); w = (pc() - t) * 1e6
    print("no events, rt.synchronize (hipStreamSynchronize): wall %.1f us (%.2f us/step)" % (w, w / 20))

"""Worker of tests/test_chain_starved_gpu.py (its own process: a starved chain launch switches chain launches off for the process).
Scenario A "shared from the start": tools/cu_hog.so holds 200 of the CUs (120 KiB of LDS each: no 160 KiB chain workgroup fits beside it)
on a second stream, then the FIRST chain invoke of the process is issued: the probation check finds the starved launch, the call runs
call by call - return value 0, results identical to the separate invokes.
Scenario B "shared later": two chain invokes with the device to ourselves (both ONE launch), then the hog, then three more chain invokes
(asynchronous, journaled) and a synchronize: the starved launches are found at the synchronisation point and re-run call by call.
Prints one JSON line."""
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("tpp-mlir_amd")
rt = pkg.get_runtime()
BF16, VB = 2, 2048
scenario = sys.argv[1]
hog = ctypes.CDLL(os.path.join(ROOT, "tools", "cu_hog.so"))
hog.cu_hog_launch.restype = ctypes.c_int
hog.cu_hog_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]

M, N, L = 4096, 1024, 3
torch.manual_seed(5)
X = (torch.rand(M, N, device="cuda") - 0.5).to(torch.bfloat16)
Ws = [((torch.rand(N // 2, N, 2, device="cuda") - 0.5) * 0.1).to(torch.bfloat16) for _ in range(L)]
Bs = [(torch.rand(N, device="cuda") - 0.5).to(torch.bfloat16) for _ in range(L)]
acts = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(L)]
ref = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(L)]
h = rt.fused_brgemm_dispatch(BF16, M, N, 64, N, N, N, 64, 64 * N, 4 | VB, 0, 5, 4, 1)
rt.set_async(True)


def calls(outs):
    return [(h, X if l == 0 else outs[l - 1], 0, Ws[l], 0, outs[l], 0, Bs[l], 0, N // 64) for l in range(L)]


# the expected result: the same layers invoked one by one (the chain is bit-identical to them on the same tile)
for c in calls(ref):
    rt.fused_brgemm(BF16, *c)
rt.synchronize()

flag = torch.zeros(16, dtype=torch.int32).pin_memory()
started = torch.zeros(1, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()


def start_hog():
    rc = hog.cu_hog_launch(side.cuda_stream, 200, 120 * 1024, flag.data_ptr(), 3000, started.data_ptr())
    assert rc == 0, rc
    t0 = time.time()
    while int(started.cpu()[0]) < 150 and time.time() - t0 < 2.0:  # (a D2H copy on the null stream would wait for the hog: use a side copy)
        time.sleep(0.002)


def poison():
    for a in acts:
        a.fill_(float("nan"))
    torch.cuda.synchronize() if False else None


out = {"scenario": scenario}
if scenario == "A":
    for a in acts:
        a.fill_(float("nan"))
    torch.cuda.current_stream().synchronize()
    start_hog()
    t0 = time.time()
    out["one_launch"] = bool(rt.fused_brgemm_chain(BF16, calls(acts)))
    rt.synchronize()
    out["seconds"] = round(time.time() - t0, 3)
    flag[0] = 1
    side.synchronize()
    out["identical"] = all(torch.equal(a.view(torch.int16), r.view(torch.int16)) for a, r in zip(acts, ref))
    # from now on chain invokes run call by call, with the device free again too
    out["later_one_launch"] = bool(rt.fused_brgemm_chain(BF16, calls(acts)))
    rt.synchronize()
else:
    free = [bool(rt.fused_brgemm_chain(BF16, calls(acts))) for _ in range(2)]
    rt.synchronize()
    out["free_one_launch"] = free
    out["free_chain_status"] = rt.chain_status()
    out["free_identical"] = all(torch.equal(a.view(torch.int16), r.view(torch.int16)) for a, r in zip(acts, ref))
    for a in acts:
        a.fill_(float("nan"))
    torch.cuda.current_stream().synchronize()
    start_hog()
    out["hogged_one_launch"] = [bool(rt.fused_brgemm_chain(BF16, calls(acts))) for _ in range(3)]
    t0 = time.time()
    # (the synchronisation finds the starved launches and re-runs them call by call - with the hog still holding its CUs)
    rt.synchronize()
    out["seconds"] = round(time.time() - t0, 3)
    flag[0] = 1
    side.synchronize()
    out["identical"] = all(torch.equal(a.view(torch.int16), r.view(torch.int16)) for a, r in zip(acts, ref))
    out["later_one_launch"] = bool(rt.fused_brgemm_chain(BF16, calls(acts)))
    rt.synchronize()
out["chain_status"] = rt.chain_status()  # starved launches repaired since process start (the probation path of scenario A repairs none)
out["hog_workgroups_started"] = int(started.cpu()[0])
print(json.dumps(out), flush=True)

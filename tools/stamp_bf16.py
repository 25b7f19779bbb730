#!/usr/bin/env python3
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tpp-mlir_amd"); rt = pkg.get_runtime(); rt.set_async(True)
m, n, k, br = 4096, 1024, 64, 16
A = (torch.rand(m, 1024, device="cuda") * 2 - 1).to(torch.bfloat16); B = (torch.rand(512, n, 2, device="cuda") * 2 - 1).to(torch.bfloat16)
C = torch.zeros(m, n, device="cuda", dtype=torch.bfloat16)
nb = (m // 128) * (n // 128)
dbg = torch.zeros(nb * 16, dtype=torch.int64, device="cuda")
h = rt.fused_brgemm_dispatch(2, m, n, k, 1024, n, n, 64, 64 * n, 4 | 2048, 0, 0, 0, 0)
for it in range(6):
    rt.fused_brgemm(2, h, A, 0, B, 0, C, 0, dbg, 0, br)
torch.cuda.synchronize()
allv = dbg.cpu().numpy()
d = allv[:nb * 8].reshape(nb, 8)
st = allv[nb * 8:].reshape(nb, 8)[:, :5].astype(np.float64)
print('per-k-step cycles of a steady-state chunk (4 MFMAs each = 128 ideal):', np.median(np.diff(st, axis=1), axis=0).astype(int).tolist())
t0, t1, t2, t3, w0, w1 = (d[:, i].astype(np.float64) for i in range(6))
print("bf16 C4 layer cycles: prologue %.0f  mainloop %.0f (%.0f/chunk)  epilogue %.0f  total %.0f (median over %d WGs)" % (
    np.median(t1 - t0), np.median(t2 - t1), np.median(t2 - t1) / 16, np.median(t3 - t2), np.median(t3 - t0), nb))
ws = (w0 - w0.min()) / 100.0; we = (w1 - w0.min()) / 100.0
print("WG start skew max %.2f us; WG end min %.2f median %.2f max %.2f us; clock %.3f GHz" % (ws.max(), we.min(), np.median(we), we.max(), np.median((t3 - t0) / ((w1 - w0) / 100.0)) / 1e3))

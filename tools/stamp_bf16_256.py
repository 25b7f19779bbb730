#!/usr/bin/env python3
"""timeline of the 256x256 bf16 kernel (library built with -DTPP_STAMP256=1, see tools/gpu_stamp256.sh)"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tpp-mlir_amd"); rt = pkg.get_runtime(); rt.set_async(True)
m = n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
k, br = 64, K // 64
A = (torch.rand(m, K, device="cuda") * 2 - 1).to(torch.bfloat16); B = (torch.rand(K // 2, n, 2, device="cuda") * 2 - 1).to(torch.bfloat16)
C = torch.zeros(m, n, device="cuda", dtype=torch.bfloat16)
nb = (m // 256) * (n // 256)
dbg = torch.zeros(nb * 16, dtype=torch.int64, device="cuda")
rt.force_variant(18)
h = rt.fused_brgemm_dispatch(2, m, n, k, K, n, n, 64, 64 * n, 4 | 2048, 0, 0, 0, 0)
rt.force_variant(-1)
print(rt.kernel_name(h))
for it in range(6):
    rt.fused_brgemm(2, h, A, 0, B, 0, C, 0, dbg, 0, br)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nb, 16).astype(np.float64)
t0, t1, t2, t3, w0, w1, c0, c1, c2, c3 = (d[:, i] for i in range(10))
T = K // 32
print("cycles: prologue %.0f  mainloop %.0f (%.0f/chunk, 1024 = MFMA-bound)  epilogue %.0f  total %.0f (median over %d WGs)" % (
    np.median(t1 - t0), np.median(t2 - t1), np.median(t2 - t1) / T, np.median(t3 - t2), np.median(t3 - t0), nb))
print("one steady chunk: step0 (reads+16 MFMA) %.0f | wait+barrier %.0f | step1 (reads+16 MFMA+8 DMA) %.0f" % (
    np.median(c1 - c0), np.median(c2 - c1), np.median(c3 - c2)))
ws = (w0 - w0.min()) / 100.0; we = (w1 - w0.min()) / 100.0
print("WG start skew max %.2f us; WG end min %.2f median %.2f max %.2f us; clock %.3f GHz" % (
    ws.max(), we.min(), np.median(we), we.max(), np.median((t3 - t0) / ((w1 - w0) / 100.0)) / 1e3))

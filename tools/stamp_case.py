#!/usr/bin/env python3
"""per-workgroup timeline of the f32 BRGEMM (library built with -DTPP_ABLATE=32)"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tpp-mlir_amd"); rt = pkg.get_runtime(); rt.set_async(True)
m = n = 1024; k, br = 64, 16
A = torch.rand(m, 1024, device="cuda") * 2 - 1; B = torch.rand(1024, n, device="cuda") * 2 - 1
C = torch.zeros(m, n, device="cuda")
dbg = torch.zeros(256 * 8 + 256 * 16, dtype=torch.int64, device="cuda")
h = rt.fused_brgemm_dispatch(1, m, n, k, 1024, 1024, 1024, 64, 65536, 4, 0, 0, 0, 0)
for it in range(6):
    rt.fused_brgemm(1, h, A, 0, B, 0, C, 0, dbg, 0, br)
torch.cuda.synchronize()
allv = dbg.cpu().numpy()
d = allv[:2048].reshape(256, 8)
full16 = allv[2048:].reshape(256, 16).astype(np.float64)
st = full16[:, :9]
pro = full16[:, 9:13]
print('prologue split (median cycles): entry->first load issue %d, loads issued->chunk0 landed (before LDS write) %d, LDS write issue %d, ->barrier passed %d' % (np.median(pro[:,0]-d[:,0]), np.median(pro[:,1]-pro[:,0]), np.median(pro[:,2]-pro[:,1]), np.median(pro[:,3]-pro[:,2])))
print('per-step cycles of a steady-state chunk (median over WGs):', np.median(np.diff(st, axis=1), axis=0).astype(int).tolist(), 'sum', int(np.median(st[:, 8] - st[:, 0])))
t0, t1, t2, t3, w0, w1 = (d[:, i].astype(np.float64) for i in range(6))
print("cycles: prologue %.0f  mainloop %.0f  epilogue %.0f  total %.0f (median over 256 WGs)" % (
    np.median(t1 - t0), np.median(t2 - t1), np.median(t3 - t2), np.median(t3 - t0)))
print("cycles max: prologue %.0f mainloop %.0f epilogue %.0f total %.0f" % ((t1 - t0).max(), (t2 - t1).max(), (t3 - t2).max(), (t3 - t0).max()))
ws = (w0 - w0.min()) / 100.0; we = (w1 - w0.min()) / 100.0   # wall_clock64 = 100 MHz -> us
print("WG start skew: median %.2f us max %.2f us; WG end: min %.2f median %.2f max %.2f us; per-WG wall median %.2f us" % (
    np.median(ws), ws.max(), we.min(), np.median(we), we.max(), np.median(we - ws)))
print("shader clock estimate: %.3f GHz" % (np.median((t3 - t0) / ((w1 - w0) / 100.0)) / 1e3))
xcc = d[:, 6] & 0xf
print("XCC of blocks 0..15:", xcc[:16].tolist(), " blocks/XCC:", np.bincount(xcc.astype(int)).tolist())

#!/usr/bin/env python3
"""HBM roofline of the bandwidth-bound xsmm ops at sizes beyond the launch-latency regime"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sweep
rt = sweep.rt
F32, BF16 = 1, 2

def case(name, fn, nbytes):
    t = sweep.time_it(fn, iters=20, warm=3)
    print("%-58s %9.1f us  %7.1f GB/s  %5.1f%% of 8 TB/s (%4.1f%% of 6.3 achievable)" % (name, t * 1e6, nbytes / t / 1e9, nbytes / t / 8e10, nbytes / t / 6.3e10), flush=True)

for (m, n) in ((8192, 8192), (16384, 16384)):
    x = torch.rand(m, n, device="cuda") - 0.5
    y = torch.empty_like(x)
    z = torch.rand(m, n, device="cuda")
    b = torch.rand(n, device="cuda")
    h = rt.unary_dispatch(5, F32, m, n, n, n, 0)
    case("unary relu f32 %dx%d" % (m, n), lambda: rt.unary(F32, h, x, 0, y, 0), 2.0 * m * n * 4)
    h0 = rt.unary_dispatch(2, F32, m, n, n, n, 0)
    case("unary zero f32 %dx%d" % (m, n), lambda: rt.unary(F32, h0, x, 0, y, 0), 1.0 * m * n * 4)
    hb = rt.binary_dispatch(1, F32, m, n, n, n, n, 0)
    case("binary add f32 %dx%d" % (m, n), lambda: rt.binary(F32, hb, x, 0, z, 0, y, 0), 3.0 * m * n * 4)
    hbb = rt.binary_dispatch(1, F32, m, n, n, n, n, 8)
    case("binary add f32 + bias (bcast_col_in1) %dx%d" % (m, n), lambda: rt.binary(F32, hbb, x, 0, b, 0, y, 0), 2.0 * m * n * 4)
    ht = rt.unary_dispatch(29, F32, m, n, n, m, 0)
    case("unary transpose f32 %dx%d" % (m, n), lambda: rt.unary(F32, ht, x, 0, y, 0), 2.0 * m * n * 4)
    xb = x.to(torch.bfloat16); yb = torch.empty_like(xb)
    hv = rt.unary_dispatch(28, BF16, m, n, n, n, 0)
    case("unary vnni2 pack bf16 %dx%d" % (m, n), lambda: rt.unary(BF16, hv, xb, 0, yb, 0), 2.0 * m * n * 2)
    htb = rt.unary_dispatch(29, BF16, m, n, n, m, 0)
    case("unary transpose bf16 %dx%d" % (m, n), lambda: rt.unary(BF16, htb, xb, 0, yb, 0), 2.0 * m * n * 2)
    hrb = rt.unary_dispatch(5, BF16, m, n, n, n, 0)
    case("unary relu bf16 %dx%d" % (m, n), lambda: rt.unary(BF16, hrb, xb, 0, yb, 0), 2.0 * m * n * 2)
    del x, y, z, xb, yb

#!/usr/bin/env python3
"""store policy of the eltwise / transpose kernels by size: run with TPP_XSMM_LIBRARY pointing at the shipped library or at a side
build with -DTPP_ELT_ST=1 (sc1) / 2 (sc1 nt); prints GB/s of transpose f32 / bf16, binary add, relu f32 / bf16 per square size"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sweep
rt = sweep.rt
F32, BF16 = 1, 2
for n in (2048, 4096, 6144, 8192, 10240, 12288, 16384):
    m = n
    x = torch.rand(m, n, device="cuda") - 0.5; y = torch.empty_like(x); z = torch.rand(m, n, device="cuda")
    xb = x.to(torch.bfloat16); yb = torch.empty_like(xb)
    ht = rt.unary_dispatch(29, F32, m, n, n, m, 0); htb = rt.unary_dispatch(29, BF16, m, n, n, m, 0)
    hb = rt.binary_dispatch(1, F32, m, n, n, n, n, 0); hr = rt.unary_dispatch(5, F32, m, n, n, n, 0); hrb = rt.unary_dispatch(5, BF16, m, n, n, n, 0)
    r = []
    for fn, nbytes in ((lambda: rt.unary(F32, ht, x, 0, y, 0), 8.0 * m * n), (lambda: rt.unary(BF16, htb, xb, 0, yb, 0), 4.0 * m * n),
                       (lambda: rt.binary(F32, hb, x, 0, z, 0, y, 0), 12.0 * m * n), (lambda: rt.unary(F32, hr, x, 0, y, 0), 8.0 * m * n),
                       (lambda: rt.unary(BF16, hrb, xb, 0, yb, 0), 4.0 * m * n)):
        t = sweep.time_it(fn, iters=20, warm=3)
        r.append(nbytes / t / 1e9)
    print("n=%5d  transpose f32 %6.0f  transpose bf16 %6.0f  add f32 %6.0f  relu f32 %6.0f  relu bf16 %6.0f  GB/s" % ((n,) + tuple(r)), flush=True)
    del x, y, z, xb, yb

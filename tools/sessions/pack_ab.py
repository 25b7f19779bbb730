#!/usr/bin/env python3
"""VNNI-2 pack (xsmm.unary 28) by size: us per launch and GB/s (TPP_XSMM_LIBRARY selects the library)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sweep
rt = sweep.rt
for n in (1024, 2048, 4096, 8192, 16384):
    x = (torch.rand(n, n, device="cuda") - 0.5).to(torch.bfloat16); y = torch.empty_like(x)
    h = rt.unary_dispatch(28, 2, n, n, n, n, 0)
    t = sweep.time_it(lambda: rt.unary(2, h, x, 0, y, 0), iters=200 if n <= 4096 else 20, warm=5)
    print("pack %5d^2  %8.2f us  %6.0f GB/s" % (n, t * 1e6, 4.0 * n * n / t / 1e9), flush=True)

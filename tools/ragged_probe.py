#!/usr/bin/env python3
"""throughput of shapes that do not fit the fast tile families (the generic / grouped kernel)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sweep
rt = sweep.rt
for (m, n, k, br) in ((1000, 1000, 1000, 1), (1000, 1000, 1024, 1), (1024, 1024, 1000, 1), (1000, 1024, 64, 16), (4000, 4000, 4096, 1), (96, 96, 96, 1)):
    A = torch.rand(br * m * k, device="cuda") - 0.5
    B = torch.rand(br * k * n, device="cuda") - 0.5
    C = torch.zeros(m * n, device="cuda")
    h = rt.brgemm_dispatch(1, m, n, k, k, n, n, m * k, k * n, 4)
    t = sweep.time_it(lambda: rt.brgemm(1, h, A, 0, B, 0, C, 0, br), iters=10, warm=2)
    print("f32 m%-5d n%-5d k%-5d br%-3d %-28s %9.1f us %7.2f TF" % (m, n, k, br, rt.kernel_name(h), t * 1e6, 2.0 * m * n * k * br / t / 1e12), flush=True)

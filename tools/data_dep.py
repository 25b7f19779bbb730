#!/usr/bin/env python3
"""C2 timing vs input data distribution (DVFS / power effect)"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sweep
from oracle import pyoracle as orc
rt = sweep.rt
m = n = 1024; k, br = 64, 16
h = rt.brgemm_dispatch(1, m, n, k, 1024, 1024, 1024, 64, 65536, 4)
gen = orc.TensorInit("normal", 123)
cases = {
  "uniform[-1,1)": (torch.rand(m, 1024) * 2 - 1, torch.rand(1024, n) * 2 - 1),
  "tpp-run normal init (N(0,0.2) clamped to [0,1], seed 123)": (torch.from_numpy(gen.fill(m * 1024)).view(m, 1024), torch.from_numpy(gen.fill(1024 * n)).view(1024, n)),
  "normal N(0,1)": (torch.randn(m, 1024), torch.randn(1024, n)),
  "ones": (torch.ones(m, 1024), torch.ones(1024, n)),
  "zeros": (torch.zeros(m, 1024), torch.zeros(1024, n)),
}
for rep in range(2):
  for name, (A, B) in cases.items():
    A, B = A.cuda(), B.cuda(); C = torch.zeros(m, n, device="cuda")
    t = sweep.time_it(lambda: rt.brgemm(1, h, A, 0, B, 0, C, 0, br), iters=200, warm=20)
    print("%-60s %7.2f us %6.1f TF %5.1f%%" % (name, t * 1e6, 2.0 * m * n * 1024 / t / 1e12, 2.0 * m * n * 1024 / t / 1.573e12), flush=True)

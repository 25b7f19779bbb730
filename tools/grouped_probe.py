#!/usr/bin/env python3
"""Time of ONE grouped (tile-queue) launch of 256 packed 32x32x32 f32 tile invokes as a function of the batch count:
intercept = fixed cost (launch, first loads, K-split reduce, epilogue), slope = time per chunk. Measurement aid."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tpp-mlir_amd")
rt = pkg.get_runtime()
rt.set_async(True)
rt.set_tile_queue(True)
F32, BF16 = 1, 2
items = int(sys.argv[1]) if len(sys.argv) > 1 else 256
MB, NB = 8, items // 8
for dt, name in ((F32, "f32"), (BF16, "bf16")):
    tdt = torch.float32 if dt == F32 else torch.bfloat16
    for br in (0, 4, 8, 16, 32, 64):
        KB = max(br, 1)
        A = (torch.rand(MB, KB, 32, 32, device="cuda") - 0.5).to(tdt)
        W = (torch.rand(NB, KB, 32, 32, device="cuda") - 0.5).to(tdt)
        C = torch.zeros(MB, NB, 32, 32, device="cuda", dtype=tdt)
        Bv = torch.zeros(NB, 32, device="cuda", dtype=tdt)
        flags = 4 | (2048 if dt == BF16 else 0)
        h = rt.fused_brgemm_dispatch(dt, 32, 32, 32, 32, 32, 32, 1024, 1024, flags, 0, 5, 4, 1)

        def layer():
            for i in range(MB):
                for j in range(NB):
                    rt.fused_brgemm(dt, h, A, i * KB * 1024, W, j * KB * 1024, C, (i * NB + j) * 1024, Bv, j * 32, br)
            rt.flush()

        for _ in range(3):
            layer()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            # GPU time of the launch alone: events around the flush of an already collected group
            for i in range(MB):
                for j in range(NB):
                    rt.fused_brgemm(dt, h, A, i * KB * 1024, W, j * KB * 1024, C, (i * NB + j) * 1024, Bv, j * 32, br)
            torch.cuda.synchronize()
            e0.record()
            rt.flush()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        print("%s items %d br %-3d %-34s %7.2f us" % (name, items, br, rt.kernel_name(h), best), flush=True)

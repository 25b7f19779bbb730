#!/usr/bin/env python3
"""Context only: the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the same shapes as the BASELINE
configs, next to this runtime's kernels. Not a parity reference and not part of the product path."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sweep
rt = sweep.rt
F32, BF16 = 1, 2


def ours(dt, m, n, K, k):
    br = K // k
    if dt == F32:
        A = torch.rand(m, K, device="cuda") - 0.5; B = torch.rand(K, n, device="cuda") - 0.5
        C = torch.zeros(m, n, device="cuda")
        h = rt.brgemm_dispatch(F32, m, n, k, K, n, n, k, k * n, 4)
    else:
        A = (torch.rand(m, K, device="cuda") - 0.5).to(torch.bfloat16)
        B = (torch.rand(K // 2, n, 2, device="cuda") - 0.5).to(torch.bfloat16)
        C = torch.zeros(m, n, device="cuda", dtype=torch.bfloat16)
        h = rt.brgemm_dispatch(BF16, m, n, k, K, n, n, k, k * n, 4 | 2048)
    return sweep.time_it(lambda: rt.brgemm(dt, h, A, 0, B, 0, C, 0, br), iters=30, warm=5), rt.kernel_name(h)


def vendor(dt, m, n, K):
    tdt = torch.float32 if dt == F32 else torch.bfloat16
    A = (torch.rand(m, K, device="cuda") - 0.5).to(tdt); B = (torch.rand(K, n, device="cuda") - 0.5).to(tdt)
    C = torch.empty(m, n, device="cuda", dtype=tdt)
    return sweep.time_it(lambda: torch.matmul(A, B, out=C), iters=30, warm=5)


torch.backends.cuda.matmul.allow_tf32 = False
for (dt, m, n, K, tag) in ((F32, 1024, 1024, 1024, "C2"), (F32, 512, 1024, 1024, "C3 GEMM part"), (F32, 4096, 4096, 4096, ""),
                           (BF16, 4096, 1024, 1024, "C4 layer"), (BF16, 2048, 2048, 2048, "C5"), (BF16, 4096, 4096, 4096, ""),
                           (BF16, 8192, 8192, 8192, "")):
    t, name = ours(dt, m, n, K, 64)
    tv = vendor(dt, m, n, K)
    fl = 2.0 * m * n * K
    print("%-4s %5d x %5d x %5d  ours %-28s %8.2f us %8.1f TF | torch.matmul %8.2f us %8.1f TF | ours/vendor %.2fx  %s" % (
        "f32" if dt == F32 else "bf16", m, n, K, name, t * 1e6, fl / t / 1e12, tv * 1e6, fl / tv / 1e12, tv / t, tag), flush=True)

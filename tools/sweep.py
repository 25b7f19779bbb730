#!/usr/bin/env python3
"""GPU-side sweep: TFLOP/s of the BRGEMM kernels over shapes / forced tile variants.
Prints one line per case. Not a test: a measurement aid for kernel work."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tpp-mlir_amd")
rt = pkg.get_runtime()
rt.set_async(True)
F32, BF16 = 1, 2


def time_it(fn, iters=200, warm=5):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.03:  # clocks up: a cold 20 us kernel reads ~10 % slow
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3 / iters)
    return best


def f32_case(m, n, k, br, force=None, beta0=True, tag=""):
    K = k * max(br, 1)
    A = torch.rand(m, K, device="cuda") * 2 - 1
    B = torch.rand(K, n, device="cuda") * 2 - 1
    C = torch.zeros(m, n, device="cuda")
    if force is not None:
        rt.force_variant(force)
    h = rt.brgemm_dispatch(F32, m, n, k, K, n, n, k, k * n, 4 if beta0 else 0)
    rt.force_variant(-1)
    t = time_it(lambda: rt.brgemm(F32, h, A, 0, B, 0, C, 0, br))
    fl = 2.0 * m * n * k * br
    print("f32  m%-5d n%-5d k%-5d br%-3d %-28s %8.2f us %8.1f TF  %5.1f%% %s" % (
        m, n, k, br, rt.kernel_name(h), t * 1e6, fl / t / 1e12, fl / t / 1.573e12, tag), flush=True)


def bf16_case(m, n, k, br, tag="", force=None):
    K = k * br
    A = (torch.rand(m, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    B = (torch.rand(K // 2, n, 2, device="cuda") * 2 - 1).to(torch.bfloat16)
    C = torch.zeros(m, n, device="cuda", dtype=torch.bfloat16)
    if force is not None:
        rt.force_variant(force)
    h = rt.brgemm_dispatch(BF16, m, n, k, K, n, n, k, k * n, 4 | 2048)
    rt.force_variant(-1)
    t = time_it(lambda: rt.brgemm(BF16, h, A, 0, B, 0, C, 0, br))
    fl = 2.0 * m * n * K
    print("bf16 m%-5d n%-5d k%-5d br%-3d %-28s %8.2f us %8.1f TF  %5.1f%% %s" % (
        m, n, k, br, rt.kernel_name(h), t * 1e6, fl / t / 1e12, fl / t / 25e12, tag), flush=True)


def bf16_flat_case(m, n, k, br, tag="", force=None):
    """flat B ([K][n] row-major, no VNNI flag): the interleave happens in the B loader of the loader-wave tiles"""
    K = k * br
    A = (torch.rand(m, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    B = (torch.rand(K, n, device="cuda") * 2 - 1).to(torch.bfloat16)
    C = torch.zeros(m, n, device="cuda", dtype=torch.bfloat16)
    if force is not None:
        rt.force_variant(force)
    h = rt.brgemm_dispatch(BF16, m, n, k, K, n, n, k, k * n, 4)
    rt.force_variant(-1)
    t = time_it(lambda: rt.brgemm(BF16, h, A, 0, B, 0, C, 0, br))
    fl = 2.0 * m * n * K
    print("bf16 m%-5d n%-5d k%-5d br%-3d %-32s %8.2f us %8.1f TF  %5.1f%% %s" % (
        m, n, k, br, rt.kernel_name(h), t * 1e6, fl / t / 1e12, fl / t / 25e12, tag), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "generic":
    # the generic kernel (variant 8) on its operand paths: k a multiple of 32 / of 4 only (16-byte loads, ragged last chunk) / odd (element loads)
    for (m, n, k, br, tag) in ((1024, 1024, 64, 16, "k % 32 == 0"), (1024, 1024, 100, 10, "k = 100: 16-byte loads, ragged last chunk"),
                               (1024, 1024, 1000, 1, "k = 1000"), (1000, 1000, 1000, 1, "m, n ragged too"), (1024, 1024, 101, 10, "k = 101: element loads")):
        f32_case(m, n, k, br, force=8, tag=tag)
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "flatb":
    # flat-B bf16 against the VNNI-2 kernels on the same shapes (same run): C5, the C4 layer and its shards, 4096^3
    for (m, n, k, br, tag) in ((2048, 2048, 128, 16, "C5"), (4096, 1024, 64, 16, "C4 layer"), (2048, 1024, 64, 16, ""),
                               (1024, 1024, 64, 16, ""), (512, 1024, 64, 16, ""), (4096, 4096, 64, 64, "4096^3")):
        bf16_case(m, n, k, br, tag + " VNNI-2 B")
        if m * n >= 128 * 128 * 192:
            bf16_case(m, n, k, br, tag + " VNNI-2 B, 128x128 loader-wave tile", force=23)
        bf16_flat_case(m, n, k, br, tag + " flat B")
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "f32lw":
    # loader-wave f32 kernels (5 64x64, 6 64x64+K2, 7 64x32+K2, 9 32x32+K4) against the round-1 kernels (0, 1, 2, 3, 4)
    for (m, n, k, br, tag) in ((1024, 1024, 64, 16, "C2"), (512, 1024, 64, 16, "C3 shape"), (256, 1024, 64, 16, "bs=256 layer"),
                               (2048, 2048, 64, 16, "2048^2 K=1024"), (4096, 4096, 64, 16, "4096^2 K=1024"),
                               (4096, 1024, 64, 16, "4096x1024 K=1024"), (1024, 1024, 64, 128, "C2 K=8192")):
        for v in (None, 0, 5, 6, 4, 3, 1, 7, 2, 9):
            if v == 3 and m % 128:
                continue
            f32_case(m, n, k, br, force=v, tag=tag + (" forced v%d" % v if v is not None else " (default pick)"))
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "bf16":
    bf16_case(4096, 1024, 64, 16, tag="C4 layer")
    bf16_case(2048, 2048, 128, 16, tag="C5")
    bf16_case(4096, 2048, 64, 32, tag="512 tiles of 128")
    bf16_case(4096, 4096, 64, 64, force=17, tag="4096^3 forced 128x128")
    sys.exit(0)

def bf16_vnni4_case(m, n, k, br, tag="", force=None):
    """VNNI-4 B ([K/4][n][4], xsmm_hip_set_vnni_factor(4)): the loader-wave tiles with 8-byte fragment reads"""
    K = k * br
    A = (torch.rand(m, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    B = (torch.rand(K // 4, n, 4, device="cuda") * 2 - 1).to(torch.bfloat16)
    C = torch.zeros(m, n, device="cuda", dtype=torch.bfloat16)
    old = rt.set_vnni_factor(4)
    if force is not None:
        rt.force_variant(force)
    h = rt.brgemm_dispatch(BF16, m, n, k, K, n, n, k, k * n, 4 | 2048)
    rt.force_variant(-1)
    rt.set_vnni_factor(old)
    t = time_it(lambda: rt.brgemm(BF16, h, A, 0, B, 0, C, 0, br))
    fl = 2.0 * m * n * K
    print("bf16 m%-5d n%-5d k%-5d br%-3d %-32s %8.2f us %8.1f TF  %5.1f%% %s" % (
        m, n, k, br, rt.kernel_name(h), t * 1e6, fl / t / 1e12, fl / t / 25e12, tag), flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "vnni4":
    # VNNI-4 against VNNI-2 and flat B on the same tiles (same run): the B fragment reads are 8-byte / 4-byte / transpose reads
    for (m, n, k, br, what) in ((4096, 1024, 64, 16, "C4 layer"), (4096, 1024, 64, 128, "C4 output, K=8192"), (2048, 2048, 128, 16, "C5"),
                                (2048, 1024, 64, 16, "2048 rows"), (1024, 1024, 64, 16, "1024 rows"), (512, 1024, 64, 16, "512 rows"),
                                (256, 1024, 64, 16, "256 rows")):
        bf16_case(m, n, k, br, tag=what + " VNNI-2")
        bf16_vnni4_case(m, n, k, br, tag=what + " VNNI-4")
        bf16_flat_case(m, n, k, br, tag=what + " flat B")
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "lw128":
    # the 128x128 loader-wave tile (forced 23 / flat B 27): C4 layer, the same output at K = 8192 (time per chunk = the difference / 112),
    # C5, two rounds of tiles, 4096^3. TPP_SWEEP_TAG labels the lines (A/B runs of side builds through TPP_XSMM_LIBRARY).
    tag = os.environ.get("TPP_SWEEP_TAG", "")
    for (m, n, k, br, what) in ((4096, 1024, 64, 16, "C4 layer"), (4096, 1024, 64, 128, "C4 output, K=8192"), (2048, 2048, 128, 16, "C5"),
                                (4096, 2048, 64, 32, "512 tiles"), (4096, 4096, 64, 64, "4096^3")):
        bf16_case(m, n, k, br, force=23, tag=what + " " + tag)
        bf16_flat_case(m, n, k, br, force=27, tag=what + " flat B " + tag)
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "small":
    # small bf16 outputs: the reference's --batch=256 layers and the per-rank shards of the MLP, tile families 16 / 19
    for m in (128, 256, 512, 1024, 2048):
        for v in (16, 19):
            bf16_case(m, 1024, 64, 16, force=v, tag="forced v%d" % v)
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "big":
    for (m, n, k, br) in ((4096, 4096, 64, 64), (8192, 8192, 64, 128), (4096, 8192, 64, 64), (2048, 2048, 128, 16),
                          (4096, 1024, 64, 16), (16384, 4096, 64, 64)):
        for v in (17, 18):
            bf16_case(m, n, k, br, force=v, tag="forced v%d" % v)
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "shards":
    # per-rank layer shapes of the row-sharded C4 MLP at 8/4/2/1 GPUs (kernel choice: TPP_HIP_BF16_T128MIN)
    for m in (512, 1024, 2048, 4096):
        bf16_case(m, 1024, 64, 16, tag="C4 layer shard, T128MIN=%s" % os.environ.get("TPP_HIP_BF16_T128MIN", "default"))
    sys.exit(0)

if __name__ == "__main__":
    f32_case(64, 64, 64, 0, tag="launch floor (empty batch)")
    f32_case(1024, 1024, 64, 0, tag="epilogue only")
    f32_case(1024, 1024, 64, 1, tag="1 chunk")
    f32_case(1024, 1024, 64, 2, tag="2 chunks")
    f32_case(1024, 1024, 64, 4)
    f32_case(1024, 1024, 64, 8)
    f32_case(1024, 1024, 64, 16, tag="C2")
    f32_case(1024, 1024, 64, 16, beta0=False, tag="C2 beta=1")
    f32_case(1024, 1024, 64, 64)
    f32_case(1024, 1024, 1024, 16, tag="C2 large variant")
    for v in (0, 1, 2, 3, 4):
        f32_case(1024, 1024, 64, 16, force=v, tag="forced v%d" % v)
    f32_case(1024, 1024, 64, 128, force=4, tag="forced v4 K=8192")
    f32_case(1024, 1024, 64, 128, force=0, tag="forced v0 K=8192")
    f32_case(512, 1024, 64, 16, tag="C3 shape")
    for v in (0, 1, 2, 4):
        f32_case(512, 1024, 64, 16, force=v, tag="C3 forced v%d" % v)
    f32_case(256, 1024, 64, 16)
    f32_case(2048, 2048, 64, 32)
    f32_case(4096, 4096, 64, 64, tag="4096^3")
    bf16_case(4096, 1024, 64, 16, tag="C4 layer")
    bf16_case(2048, 2048, 128, 16, tag="C5")
    bf16_case(4096, 4096, 64, 64, tag="4096^3")
    bf16_case(8192, 8192, 64, 128, tag="8192^3")
    bf16_case(1024, 1024, 64, 16)

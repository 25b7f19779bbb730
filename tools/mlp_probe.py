#!/usr/bin/env python3
"""why is the MLP step bimodal? time it repeatedly, print buffer addresses"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("tpp-mlir_amd"); rt = pkg.get_runtime(); rt.set_async(True)
BF16 = 2; N = 1024
spec = pkg.MlpSpec(); sh = pkg.ShardedMlp(spec, 0, 1, rt)
junk = [torch.empty(int(sys.argv[1]) if len(sys.argv) > 1 else 1, device="cuda")]
g = torch.Generator(device="cpu").manual_seed(7)
X = (torch.randn(sh.rows, N, generator=g) * 0.5).to(torch.bfloat16).cuda()
hp = rt.unary_dispatch(28, BF16, N, N, N, N, 0)
Wv, Bs = [], []
for _ in range(3):
    wf = (torch.randn(N, N, generator=g) * 0.04).to(torch.bfloat16).cuda()
    wv = torch.empty_like(wf); rt.unary(BF16, hp, wf, 0, wv, 0); Wv.append(wv)
    Bs.append((torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).cuda())
acts = [torch.empty(sh.rows, N, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
torch.cuda.synchronize()
print("ptrs mod 2MiB (KiB): X %d W %s acts %s" % ((X.data_ptr() % (2 << 20)) >> 10, [(w.data_ptr() % (2 << 20)) >> 10 for w in Wv], [(a.data_ptr() % (2 << 20)) >> 10 for a in acts]))
for rep in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(200):
        sh.forward(X, Wv, Bs, acts)
    e1.record(); torch.cuda.synchronize()
    print("rep %d: %.1f us/step (events) %.1f us/step (wall)" % (rep, e0.elapsed_time(e1) * 1e3 / 200, (time.perf_counter() - t0) * 1e6 / 200), flush=True)
# per layer
for l in range(3):
    h, br = sh.handles[l]
    src = X if l == 0 else acts[l - 1]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        rt.fused_brgemm(BF16, h, src, 0, Wv[l], 0, acts[l], 0, Bs[l], 0, br)
    e1.record(); torch.cuda.synchronize()
    print("layer %d: %.1f us" % (l, e0.elapsed_time(e1) * 1e3 / 200))

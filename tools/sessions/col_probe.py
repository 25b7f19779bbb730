import importlib, os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("tpp-mlir_amd"); rt = pkg.get_runtime(); rt.set_async(True)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
spec = pkg.MlpSpec(); N = 1024
cs = pkg.ColumnShardedMlp(spec, 0, 1, rt)
print([rt.kernel_name(h) for (h, br, nw) in cs.handles], [br for (h, br, nw) in cs.handles])
X = torch.randn(4096, N).to(torch.bfloat16).cuda()
W = [torch.randn(N // 2, N, 2).to(torch.bfloat16).cuda() for _ in range(3)]
B = [torch.randn(N).to(torch.bfloat16).cuda() for _ in range(3)]
loc = [torch.empty(4096, N, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
gat = [torch.empty(1, 4096, N, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
def t(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("compute only", t(lambda: cs.forward(X, W, B, loc, gat, lambda d, s: None)))
print("gather only", t(lambda: dist.all_gather_into_tensor(gat[0].view(-1, N), loc[0])))
print("both", t(lambda: cs.forward(X, W, B, loc, gat, lambda d, s: dist.all_gather_into_tensor(d.view(-1, d.shape[-1]), s))))
dist.destroy_process_group()

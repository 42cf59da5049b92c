import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from mofo_amd import ops
dev = torch.device("cuda:0")
def t(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
for (M, N) in [(50176, 1152), (5120, 2304), (5120, 768)]:
    for K in (64, 128, 256, 384, 768, 1536):
        A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16); W = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev); Cf = torch.empty(M, N, dtype=torch.float32, device=dev); R = torch.randn(M, N, device=dev)
        u1 = t(lambda: ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A, W, C))
        u2 = t(lambda: ops.gemm(ops.GEMM_NT, ops.EPI_RESID_F32, A, W, Cf, resid=R))
        print(f"NT M={M} N={N} K={K:5d}: bf16-out {u1:7.1f} us   resid-f32-out {u2:7.1f} us")

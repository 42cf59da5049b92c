import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from mofo_amd import ops
dev = torch.device("cuda:0")
def t(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (5120, 2304, 768), (5120, 2304, 4096), (50176, 1152, 384), (50176, 1152, 4096), (2048, 2048, 768)]:
    A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16); W = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    us = t(lambda: ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A, W, C))
    print(f"NT {M}x{N}x{K}: {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s")

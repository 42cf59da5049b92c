import sys, os, torch
sys.path.insert(0, os.getcwd())
from mofo_amd import ops
dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def t(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(50176, 384, 1536), (50176, 384, 1152), (50176, 384, 384), (50176, 1536, 384), (5120, 768, 3072), (5120, 768, 2304), (5120, 768, 768), (5120, 3072, 768)]:
    A = r(M, K); Wnn = r(K, N); Wnt = Wnn.t().contiguous(); C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    res = []
    for rep in range(3):
        a = t(lambda: ops.gemm(ops.GEMM_NN, ops.EPI_BF16, A, Wnn, C))
        b = t(lambda: ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A, Wnt, C))
        res.append((a, b))
    a = sorted(x[0] for x in res)[1]; b = sorted(x[1] for x in res)[1]
    print(f"M={M} N={N} K={K}: NN {a:7.1f} us  NT {b:7.1f} us  ({2.0*M*N*K/a/1e6:5.0f} vs {2.0*M*N*K/b/1e6:5.0f} TFLOP/s)", flush=True)

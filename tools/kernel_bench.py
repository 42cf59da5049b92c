#!/usr/bin/env python3
"""Per-shape timing of the hot kernels at the ViT-B B=32 shapes (GPU box only): which (op, shape) is far from its roofline."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mofo_amd import ops
from mofo_amd.runtime import _wsplits

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32
def r(*s, dt=BF16): return (torch.randn(*s, device=dev) * 0.5).to(dt)

def timeit(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rows = []
for tag, M, D in (("enc", B * 160, 768), ("dec", B * 1568, 384)):
    H = 4 * D
    for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", H, D), ("fc2", D, H)):
        A, W = r(M, K), r(N, K)
        C = torch.empty(M, N, dtype=BF16, device=dev); Cf = torch.empty(M, N, dtype=F32, device=dev); R = r(M, N, dt=F32); C2 = torch.empty_like(C)
        bias = r(N, dt=F32)
        epi = {"qkv": ops.EPI_BF16, "proj": ops.EPI_RESID_F32, "fc1": ops.EPI_BIAS_GELU, "fc2": ops.EPI_RESID_F32}[name]
        if epi == ops.EPI_BF16: f = lambda: ops.gemm(ops.GEMM_NT, epi, A, W, C, bias=bias)
        elif epi == ops.EPI_BIAS_GELU: f = lambda: ops.gemm(ops.GEMM_NT, epi, A, W, C, C2=C2, bias=bias)
        else: f = lambda: ops.gemm(ops.GEMM_NT, epi, A, W, Cf, bias=bias, resid=R)
        t = timeit(f); fl = 2.0 * M * N * K
        rows.append((f"{tag}.{name} fwd NT  M={M} N={N} K={K}", t, fl / t / 1e6))
        # dgrad: dX[M,K] = dY[M,N] @ W[N,K]
        dY = r(M, N); dX = torch.empty(M, K, dtype=BF16, device=dev)
        if name == "fc2":
            hpre = r(M, K); f = lambda: ops.gemm(ops.GEMM_NN, ops.EPI_DGELU_BF16, dY, W, dX, aux=hpre)
        else:
            f = lambda: ops.gemm(ops.GEMM_NN, ops.EPI_BF16, dY, W, dX)
        t = timeit(f)
        rows.append((f"{tag}.{name} dgrad NN M={M} N={K} K={N}", t, fl / t / 1e6))
        G = torch.zeros(N, K, dtype=F32, device=dev); sp = _wsplits(N, K, M)
        f = lambda: ops.gemm(ops.GEMM_TN, ops.EPI_F32, dY, A, G, splits=sp, accumulate=False)
        t = timeit(f)
        rows.append((f"{tag}.{name} wgrad TN P={N} Q={K} R={M} splits={sp}", t, fl / t / 1e6))
    Hh = D // 64; n = M // B
    qkv = r(M, 3 * D); out = torch.empty(M, D, dtype=BF16, device=dev); lse = torch.empty(B * Hh * n, dtype=F32, device=dev)
    t = timeit(lambda: ops.attention_fwd(qkv, B, n, Hh, 0.125, out, lse)); fl = 4.0 * B * Hh * n * n * 64
    rows.append((f"{tag}.attn fwd B={B} N={n} H={Hh}", t, fl / t / 1e6))
    dout = r(M, D); dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
    t = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, n, Hh, 0.125, dqkv, delta))
    rows.append((f"{tag}.attn bwd B={B} N={n} H={Hh}", t, 2 * fl / t / 1e6))
    x = r(M, D, dt=F32); w_ = r(D, dt=F32); y = torch.empty(M, D, dtype=BF16, device=dev); mean = torch.empty(M, dtype=F32, device=dev); rstd = torch.empty_like(mean)
    t = timeit(lambda: ops.layernorm_fwd(x, w_, w_, 1e-6, y, mean, rstd))
    rows.append((f"{tag}.ln fwd M={M} D={D}", t, 6.0 * M * D / t / 1e3))
    dy = r(M, D); dx = torch.empty_like(x); dxb = torch.empty_like(y); dw = torch.zeros(D, dtype=F32, device=dev); db = torch.zeros_like(dw)
    t = timeit(lambda: ops.layernorm_bwd(dy, x, w_, mean, rstd, x, dx, dxb, dw, db))
    rows.append((f"{tag}.ln bwd M={M} D={D}", t, 16.0 * M * D / t / 1e3))
    big = r(M, 4 * D); o = torch.zeros(4 * D, dtype=F32, device=dev)
    t = timeit(lambda: ops.colsum_bf16(big, o))
    rows.append((f"{tag}.colsum M={M} N={4*D}", t, 2.0 * M * 4 * D / t / 1e3))
# head
M, N, K = B * 1408, 1536, 384
A, W = r(M, K), r(N, K); C = torch.empty(M, N, dtype=BF16, device=dev); bias = r(N, dt=F32)
t = timeit(lambda: ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A, W, C, bias=bias)); rows.append((f"head fwd NT M={M} N={N} K={K}", t, 2.0 * M * N * K / t / 1e6))
for name, t, rate in rows:
    unit = "GB/s" if (".ln" in name or "colsum" in name) else "TF/s"
    print(f"{name:58s} {t:9.1f} us {rate:9.1f} {unit}")

#!/usr/bin/env python3
"""run ONE kernel shape repeatedly (for rocprofv3 --pmc passes).  usage: one_gemm.py <nt|nn|tn|attnf|attnb> M N K [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mofo_amd import ops
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
if kind == "nt":
    A, W, C = r(M, K), r(N, K), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    f = lambda: ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A, W, C)
elif kind == "nn":
    A, W, C = r(M, K), r(K, N), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    f = lambda: ops.gemm(ops.GEMM_NN, ops.EPI_BF16, A, W, C)
elif kind == "tn":
    A, X, C = r(K, M), r(K, N), torch.zeros(M, N, dtype=torch.float32, device=dev)
    f = lambda: ops.gemm(ops.GEMM_TN, ops.EPI_F32, A, X, C)
elif kind in ("attnf", "attnb"):
    B, n, H = M, N, K
    qkv = r(B * n, 3 * H * 64); out = torch.empty(B * n, H * 64, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H * n, device=dev)
    dout = r(B * n, H * 64); dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
    ops.attention_fwd(qkv, B, n, H, 0.125, out, lse)
    f = (lambda: ops.attention_fwd(qkv, B, n, H, 0.125, out, lse)) if kind == "attnf" else (lambda: ops.attention_bwd(qkv, out, dout, lse, B, n, H, 0.125, dqkv, delta))
for _ in range(iters):
    f()
torch.cuda.synchronize()

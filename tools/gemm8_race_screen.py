#!/usr/bin/env python3
"""Race screen for the counted-vmcnt kernel (csrc/gemm8.h): a half-tile read before its LDS-DMA landed shows as rare wrong tiles that come
and go (guide: "place reads by the vmcnt / barrier count, never by clean runs" -- this is the complementary check).  Every launch of a
shape must reproduce the first launch bit for bit (the arithmetic order is fixed), under memory load from a concurrent copy stream.
usage: gemm8_race_screen.py [launches per shape]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mofo_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0"); BF16, F32 = torch.bfloat16, torch.float32
os.environ["MOFO_GEMM8"] = "1"
side = torch.cuda.Stream()
big = torch.empty(256 << 20, dtype=torch.uint8, device=dev); big2 = torch.empty_like(big)
shapes = [("NT bf16", ops.GEMM_NT, ops.EPI_BF16, 10240, 3072, 1024), ("NT gelu", ops.GEMM_NT, ops.EPI_BIAS_GELU, 10240, 4096, 1024),
          ("NT rf32", ops.GEMM_NT, ops.EPI_RESID_F32, 10240, 1024, 4096), ("NT bf16 K=512", ops.GEMM_NT, ops.EPI_BF16, 100352, 1536, 512),
          ("NN bf16", ops.GEMM_NN, ops.EPI_BF16, 4096, 4096, 4096), ("TN f32", ops.GEMM_TN, ops.EPI_F32, 2304, 768, 5120)]
bad = 0
for name, op, epi, M, N, K in shapes:
    A = (torch.randn((K, M) if op == ops.GEMM_TN else (M, K), device=dev) * 0.5).to(BF16)
    B = (torch.randn((N, K) if op == ops.GEMM_NT else (K, N), device=dev) * 0.05).to(BF16)
    out_f32 = epi in (ops.EPI_RESID_F32, ops.EPI_F32)
    Cc = torch.empty(M, N, dtype=F32 if out_f32 else BF16, device=dev)
    kw = {}
    if epi == ops.EPI_BIAS_GELU: kw = dict(C2=torch.empty(M, N, dtype=BF16, device=dev), bias=torch.randn(N, device=dev))
    if epi == ops.EPI_RESID_F32: kw = dict(resid=torch.randn(M, N, device=dev), bias=torch.randn(N, device=dev))
    ops.gemm(op, epi, A, B, Cc, **kw)
    torch.cuda.synchronize()
    ref = Cc.clone()
    ref2 = kw["C2"].clone() if "C2" in kw else None
    mism = 0
    for i in range(n):
        if i % 3 == 0:
            with torch.cuda.stream(side):
                big2.copy_(big, non_blocking=True)          # HBM / L2 pressure beside the GEMM
        Cc.fill_(0)
        ops.gemm(op, epi, A, B, Cc, **kw)
        ok = torch.equal(Cc, ref) and (ref2 is None or torch.equal(kw["C2"], ref2))
        mism += 0 if ok else 1
    torch.cuda.synchronize()
    bad += mism
    print(f"{name:14s} {M}x{N}x{K}: {n} launches, {mism} differ from the first")
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
sys.exit(0 if bad == 0 else 1)

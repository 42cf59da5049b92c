#!/usr/bin/env python3
"""How far are the hand-written GEMMs from the vendor library at the model's shapes?  Times torch.matmul (hipBLASLt / rocBLAS
under PyTorch-ROCm, bf16 in / bf16 out, no epilogue) next to mofo_gemm on the same operands.  Diagnostic only (GPU box).
usage: vendor_gemm_compare.py [vitb|vitl|vitb,vitl]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mofo_amd import ops

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32


def r(*s, dt=BF16):
    return (torch.randn(*s, device=dev) * 0.5).to(dt)


def timeit(f, iters=30):
    for _ in range(40):      # steady state: ~a few ms of the same kernel first
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B = 32
SHAPES = {"vitb": (("enc", B * 160, 768), ("dec", B * 1568, 384)), "vitl": (("L.enc", B * 320, 1024), ("L.dec", B * 3136, 512))}
which = sys.argv[1] if len(sys.argv) > 1 else "vitb"
print(f"{'shape':52s} {'mofo us':>9s} {'TF/s':>7s} {'torch us':>9s} {'TF/s':>7s}")
for tag, M, D in [x for k in which.split(",") for x in SHAPES[k]]:
    for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
        A, W, dY = r(M, K), r(N, K), r(M, N)
        C = torch.empty(M, N, dtype=BF16, device=dev)
        dX = torch.empty(M, K, dtype=BF16, device=dev)
        G = torch.zeros(N, K, dtype=F32, device=dev)
        Gb = torch.empty(N, K, dtype=BF16, device=dev)
        Wt = W.t()
        fl = 2.0 * M * N * K
        # a lone weight gradient: split the reduction until the 128 x 128 tiles fill the chip's 768 block slots (the model groups several
        # blocks' weight gradients into one launch instead; the zero fill of the split form is inside the timing)
        tiles = -(-N // 128) * -(-K // 128)
        wsplits = max(1, min(M // 512, 768 // tiles))
        rows = [
            (f"{tag}.{name} fwd   C[{M},{N}] = A[{M},{K}] W^T", lambda: ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A, W, C), lambda: torch.matmul(A, Wt, out=C)),
            (f"{tag}.{name} dgrad dX[{M},{K}] = dY[{M},{N}] W", lambda: ops.gemm(ops.GEMM_NN, ops.EPI_BF16, dY, W, dX), lambda: torch.matmul(dY, W, out=dX)),
            (f"{tag}.{name} wgrad G[{N},{K}] = dY^T A (f32 | bf16 out)", lambda: (G.zero_() if wsplits > 1 else None, ops.gemm(ops.GEMM_TN, ops.EPI_F32, dY, A, G, splits=wsplits, accumulate=False)), lambda: torch.matmul(dY.t(), A, out=Gb)),
        ]
        for label, f_m, f_t in rows:
            tm, tt = timeit(f_m), timeit(f_t)
            print(f"{label:52s} {tm:9.1f} {fl / tm / 1e6:7.0f} {tt:9.1f} {fl / tt / 1e6:7.0f}")

#!/usr/bin/env python3
"""How far are the hand-written attention kernels from what PyTorch-ROCm ships?  torch.nn.functional.scaled_dot_product_attention
(flash / memory-efficient backends: CK / AOTriton kernels) forward and forward + backward next to mofo_attention_fwd / the dQ and
dK/dV passes, same shapes, bf16, head dim 64, no mask, no dropout.  Diagnostic only (GPU box).  usage: vendor_attn_compare.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mofo_amd import ops

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32


def timeit(f, iters=20, warm=10):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(f"{'shape':44s} {'mofo fwd':>9s} {'bwd':>8s} | {'torch fwd':>9s} {'bwd':>8s}   (us; bwd = dQ + dK/dV passes | autograd backward)   fwd TF/s mofo / torch")
for tag, B, H, N in (("ViT-B enc  B32 H12 N160", 32, 12, 160), ("ViT-B dec  B32 H6  N1568", 32, 6, 1568),
                     ("ViT-L enc  B32 H16 N320", 32, 16, 320), ("ViT-L dec  B32 H8  N3136", 32, 8, 3136)):
    D = H * 64
    qkv = (torch.randn(B * N, 3 * D, device=dev) * 0.5).to(BF16)
    out = torch.empty(B * N, D, dtype=BF16, device=dev)
    lse = torch.empty(B * H * N, dtype=F32, device=dev)
    dout = (torch.randn(B * N, D, device=dev) * 0.1).to(BF16)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B * H * N, dtype=F32, device=dev)
    t_f = timeit(lambda: ops.attention_fwd(qkv, B, N, H, 0.125, out, lse))
    t_b = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, N, H, 0.125, dqkv, delta))
    q, k, v = (qkv.view(B, N, 3, H, 64)[:, :, i].transpose(1, 2).contiguous().requires_grad_(True) for i in range(3))
    do = dout.view(B, N, H, 64).transpose(1, 2).contiguous()
    tt_f = timeit(lambda: F.scaled_dot_product_attention(q, k, v, scale=0.125))
    o = F.scaled_dot_product_attention(q, k, v, scale=0.125)
    tt_b = timeit(lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True))
    err = float((o.transpose(1, 2).reshape(B * N, D).float() - out.float()).norm() / out.float().norm())
    fl = 4.0 * B * H * N * N * 64
    print(f"{tag:44s} {t_f:9.1f} {t_b:8.1f} | {tt_f:9.1f} {tt_b:8.1f}   {fl / t_f / 1e6:6.0f} / {fl / tt_f / 1e6:6.0f}   (outputs differ by {err:.1e})", flush=True)

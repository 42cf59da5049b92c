"""Grouped weight gradients (TN, f32 out) of 1-4 encoder blocks per launch through the 128 x 128 kernel and through gemm8, steady state
(GPU box only).  Four blocks = 16 problems need MAXG >= 16 in csrc/gemm.hip (13 in the tree): that row is skipped otherwise.
usage: wgrad_group_gemm8.py [token_rows D]"""
import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from mofo_amd import ops
dev = torch.device("cuda:0"); BF16, F32 = torch.bfloat16, torch.float32
R, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5120, 768)
def block():
    r = lambda *s: (torch.randn(*s, device=dev) * 0.1).to(BF16)
    dY = [r(R, 3 * D), r(R, D), r(R, 4 * D), r(R, D)]; X = [r(R, D), r(R, D), r(R, D), r(R, 4 * D)]
    G = [torch.empty(a.shape[1], b.shape[1], dtype=F32, device=dev) for a, b in zip(dY, X)]
    return list(zip(dY, X, G))
def t(f, warm, it):
    for _ in range(warm): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
blocks = [block() for _ in range(6)]
flop_block = 2.0 * R * D * D * 12
for nb in (1, 2, 3, 4):
    probs = [(a, b, g, dict(splits=1, accumulate=False)) for blk in blocks[:nb] for a, b, g in blk]
    row = [f"{nb} block(s):"]
    for mode in ("0", "1"):
        os.environ["MOFO_GEMM8"] = mode
        try:
            us = t(lambda: ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, probs), 60, 20)
        except RuntimeError as e:
            row.append(f"gemm8={mode} refused ({str(e)[-40:]}) |")
            continue
        row.append(f"gemm8={mode} {us:7.1f} us = {us / nb:6.1f} per block, {flop_block * nb / us / 1e6:5.0f} TF/s |")
    print(" ".join(row))

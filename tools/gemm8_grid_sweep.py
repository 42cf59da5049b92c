import os, sys, torch
sys.path.insert(0, os.getcwd())
from mofo_amd import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
def t(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
os.environ["MOFO_GEMM8"] = "1"
for name, M, N, K in (("dec qkv", 50176, 1152, 384), ("dec fc1-like bf16", 50176, 1536, 384), ("enc qkv", 5120, 2304, 768)):
    A = (torch.randn(M, K, device=dev) * 0.5).to(BF16); W = (torch.randn(N, K, device=dev) * 0.05).to(BF16); Cc = torch.empty(M, N, dtype=BF16, device=dev)
    tiles = -(-M // 256) * -(-N // 256)
    for grid in (32, 64, 128, 256):
        os.environ["MOFO_GEMM8_GRID"] = str(grid)
        us = t(lambda: ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A, W, Cc))
        rounds = -(-tiles // grid)
        print(f"{name:18s} grid {grid:3d}: {us:8.1f} us, {tiles} tiles = {rounds} rounds -> {us / rounds:6.2f} us per tile-round, {2.0*M*N*K/us/1e6:6.0f} TF/s")

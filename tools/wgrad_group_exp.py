#!/usr/bin/env python3
"""Experiment (measurement tooling): is the encoder's grouped weight-gradient launch (432 tiles of 128x128 on 256 CUs, 1.69
tiles per CU) limited by that grid quantisation?  The same reduction length (K = 5120 token rows) with 1x, 2x, 3x, 4x the
output tiles in one launch -- emulated by widening the dY operand, so MAXG = 4 problems still suffice.  If time grows much
less than the work, grouping 2-4 blocks' weight gradients per launch pays.   GPU only.   usage: wgrad_group_exp.py [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mofo_amd import ops

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]       # (out features = columns of dY, in features = columns of X)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
base = None
for mult in (1, 2, 3, 4):
    probs = []
    for o, i in shapes:
        dY, X = r(K, o * mult), r(K, i)
        G = torch.empty(o * mult, i, dtype=torch.float32, device=dev)
        probs.append((dY, X, G, dict(splits=1, accumulate=False)))
    f = lambda: ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, probs)
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    t = sorted(ts)[2]
    fl = 2.0 * K * sum(o * i for o, i in shapes) * mult
    tiles = sum(((o * mult + 127) // 128) * ((i + 127) // 128) for o, i in shapes)
    base = base or t
    print(f"x{mult}: {tiles:5d} tiles ({tiles / 256:.2f} per CU)  {t:7.1f} us  = {t / mult:6.1f} us per block's worth  {fl / t / 1e6:6.0f} TFLOP/s  (time x{t / base:.2f} for work x{mult})", flush=True)

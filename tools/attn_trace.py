#!/usr/bin/env python3
"""Phase timeline inside one key tile of the attention forward / dQ kernels (GPU box only; debug tooling).

Build first, in the container:  python tools/attn_trace.py --build   (attention.hip with -DMOFO_ATTN_TRACE)
Stamps (wave 0 of every block, key tile 10): 0 tile start | 1 S = K Q^T MFMAs issued | 2 softmax / dS VALU done (operands
packed) | 3 PV (or dQ) MFMAs issued | 4 next tile written to LDS | 5 barrier passed | 6 next-next tile's global loads issued.
usage: attn_trace.py <fwd|dq|dkv|fused> B N H
"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_trace", os.environ.get("MOFO_TRACE_LIB", "libmofo_attn_trace.so"))
if "--build" in sys.argv:
    from mofo_amd import build as b
    b.build()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    obj = OUT + ".o"
    subprocess.check_call([b.HIPCC] + b.FLAGS + b.EXTRA_FLAGS.get("attention.hip", []) + ([] if os.environ.get("MOFO_TRACE_NOSTAMP") else ["-DMOFO_ATTN_TRACE"]) + os.environ.get("MOFO_TRACE_DEFS", "").split() + ["-c", os.path.join(b.CSRC, "attention.hip"), "-o", obj])
    objs = [obj] + [os.path.join(b.HERE, "build", s + ".o") for s in b.SOURCES if s != "attention.hip"]
    subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    print(OUT); sys.exit(0)

import ctypes as C
import numpy as np
import torch
from mofo_amd import _lib
_lib.LIB_PATH = OUT
from mofo_amd import ops
kind, B, N, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
qkv = r(B * N, 3 * H * 64); out = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H * N, device=dev)
dout = r(B * N, H * 64); dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
ops.attention_fwd(qkv, B, N, H, 0.125, out, lse)
ops.attention_delta(out, dout, B, N, H, delta)
f = {"fwd": lambda: ops.attention_fwd(qkv, B, N, H, 0.125, out, lse),
     "dq": lambda: ops.attention_bwd_dq(qkv, dout, lse, delta, B, N, H, 0.125, dqkv),
     "dkv": lambda: ops.attention_bwd_dkv(qkv, dout, lse, delta, B, N, H, 0.125, dqkv),
     "fused": lambda: ops.attention_bwd(qkv, out, dout, lse, B, N, H, 0.125, dqkv, delta)}[kind]
names = ["issue S MFMAs (4)", "softmax / dS VALU (+ dP MFMAs in dq)", "issue PV / dQ MFMAs (4)", "write next tile to LDS", "barrier", "issue next global loads"]
if kind == "dkv":     # MOFO_ATTN_DKV_PIPE=0: the two-phase dK/dV kernel (query tile 10 of wave 0)
    names = ["row-fragment reads + S, dP MFMAs issued (8)", "tr-fragment reads issued + exp2 / dS VALU + packs", "dV, dK MFMAs issued (8)",
             "write next tile to LDS", "barrier", "issue next global loads"]
if kind == "fused":   # N <= 160: the one-kernel backward
    names = ["issue loads + stage tiles + delta", "barrier", "step 0: pair (S, dP, dS, dV, dK)", "barrier", "step 0: dQ += K^T dS, barrier", "steps 1..T-1"]
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"{kind} B={B} N={N} H={H}: {us:.1f} us (trace build)")
if os.environ.get("MOFO_TRACE_NOSTAMP"):
    sys.exit(0)
lib = _lib.load()
buf = np.zeros((1 << 15) * 8, dtype=np.uint64)
lib.mofo_debug_attn_trace_read.argtypes = [C.c_void_p, C.c_size_t]; lib.mofo_debug_attn_trace_read.restype = C.c_int
assert lib.mofo_debug_attn_trace_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(-1, 8)
if kind == "fused":     # persistent over the (clip, head) items: slot 7 = end of the block's last item
    full = t[t[:, 0] > 0].astype(np.int64)
    if "SECOND_ITEM" in os.environ.get("MOFO_TRACE_DEFS", ""):
        two = full[full[:, 7] > 0]
        print(f"second-item build: end of item 1's steps -> after the DMA wait: p50 {np.median(two[:, 7] - two[:, 0]):.0f} clk ({len(two)} two-item blocks)")
    life = full[:, 7] - full[:, 0]
    print(f"block life (all its items): p10 {np.percentile(life, 10):.0f}  p50 {np.percentile(life, 50):.0f}  p90 {np.percentile(life, 90):.0f}  max {life.max()} clk; "
          f"kernel span {full[:, 7].max() - full[:, 0].min()} clk; blocks starting within {full[:, 0].max() - full[:, 0].min()} clk")
t = t[t[:, 0] > 0][:, :7].astype(np.int64)
d = np.diff(t, axis=1)
print(f"{len(t)} blocks")
for i, n in enumerate(names):
    print(f"  {n:40s} mean {d[:, i].mean():7.0f}  p10 {np.percentile(d[:, i], 10):7.0f}  p50 {np.percentile(d[:, i], 50):7.0f}  p90 {np.percentile(d[:, i], 90):7.0f} clk")
print(f"  tile total mean {(t[:, 6] - t[:, 0]).mean():.0f} clk")

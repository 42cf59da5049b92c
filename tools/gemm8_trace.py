#!/usr/bin/env python3
"""Phase stamps of one K-tile of the counted-vmcnt kernel (GPU box only; debug build).  Build in the container first:
    python tools/gemm_trace.py --build          (gemm.hip with -DMOFO_GEMM_TRACE -> tools/_trace/libmofo_trace.so)
usage: gemm8_trace.py <nt|nn|tn> M N K
Per phase and wave group (wave 0 = group 0, wave 4 = group 1, one section behind): cycles in the LOAD section (fragment reads, the
half-tile's two LDS-DMA pieces, the counted wait), waiting at the mid barrier, in the MFMA section (16 MFMAs = 256 cycles of matrix
pipe), waiting at the end barrier.  Median over the blocks."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mofo_amd import _lib
_lib.LIB_PATH = os.environ.get("MOFO_TRACE_LIB", os.path.join(ROOT, "tools", "_trace", "libmofo_trace.so"))
from mofo_amd import ops
os.environ["MOFO_GEMM8"] = "1"
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0")
r = lambda *s, sc=0.5: (torch.randn(*s, device=dev) * sc).to(torch.bfloat16)
op = {"nt": ops.GEMM_NT, "nn": ops.GEMM_NN, "tn": ops.GEMM_TN}[kind]
A = r(M, K) if kind != "tn" else r(K, M)
Bm = r(N, K, sc=0.05) if kind == "nt" else r(K, N, sc=0.05)
Cc = torch.empty(M, N, dtype=torch.float32 if kind == "tn" else torch.bfloat16, device=dev)
f = (lambda: ops.gemm(op, ops.EPI_F32, A, Bm, Cc)) if kind == "tn" else (lambda: ops.gemm(op, ops.EPI_BF16, A, Bm, Cc))
for _ in range(200): f()
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(256 * 64, dtype=np.uint64)
lib.mofo_debug_trace8_read.argtypes = [C.c_void_p, C.c_size_t]; lib.mofo_debug_trace8_read.restype = C.c_int
assert lib.mofo_debug_trace8_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(256, 2, 32).astype(np.int64)
tiles = -(-M // 256) * -(-N // 256)
nb = min(256, tiles)
print(f"{kind} {M}x{N}x{K}: stamps of K-tile 6, median over {nb} blocks (cycles)")
for g in range(2):
    x = t[:nb, g, :17]
    ok = x[:, 16] > x[:, 0]
    x = x[ok]
    tot = np.median(x[:, 16] - x[:, 0])
    print(f" group {g} (wave {4 * g}): K-tile total {tot:.0f}")
    for p in range(4):
        ld = np.median(x[:, 4 * p + 1] - x[:, 4 * p]); wm = np.median(x[:, 4 * p + 2] - x[:, 4 * p + 1])
        mf = np.median(x[:, 4 * p + 3] - x[:, 4 * p + 2]); we = np.median(x[:, 4 * p + 4] - x[:, 4 * p + 3])
        print(f"   P{p}: load section {ld:5.0f} | mid-barrier wait {wm:5.0f} | MFMA section {mf:5.0f} | end-barrier wait {we:5.0f}")

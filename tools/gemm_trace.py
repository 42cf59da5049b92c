#!/usr/bin/env python3
"""Phase timeline of one GEMM shape (GPU box only; debug tooling, not part of the product).

Build the trace library first, in the container:  python tools/gemm_trace.py --build
(gemm.hip with -DMOFO_GEMM_TRACE -> tools/_trace/libmofo_trace.so, other objects reused from mofo_amd/build/).
Each block records s_memtime at: 0 start, 1 first tile landed, 2 main loop done, 3 epilogue staged in LDS,
4 stores issued, 5 stores acknowledged; slot 7 = (XCC_ID, HW_ID).
usage: gemm_trace.py <nt|nn|tn> <epi> M N K      epi: bf16 | gelu | resid | dgelu | f32
"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_trace", "libmofo_trace.so")
if "--build" in sys.argv:
    from mofo_amd import build as b
    b.build()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    obj = os.path.join(os.path.dirname(OUT), "gemm_trace.o")
    extra = [a for a in sys.argv if a.startswith("-D")]
    OUT = OUT.replace(".so", "".join(a[2:].replace("=", "") for a in extra) + ".so")
    subprocess.check_call([b.HIPCC] + b.FLAGS + ["-DMOFO_GEMM_TRACE"] + extra + ["-c", os.path.join(b.CSRC, "gemm.hip"), "-o", obj])
    objs = [obj] + [os.path.join(b.HERE, "build", s + ".o") for s in b.SOURCES if s != "gemm.hip"]
    subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    print(OUT); sys.exit(0)

import ctypes as C
import numpy as np
import torch
from mofo_amd import _lib
_lib.LIB_PATH = os.environ.get("MOFO_TRACE_LIB", OUT)
from mofo_amd import ops
kind, epi, M, N, K = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dev = torch.device("cuda:0")
r = lambda *s, dt=torch.bfloat16: (torch.randn(*s, device=dev) * 0.5).to(dt)
op = {"nt": ops.GEMM_NT, "nn": ops.GEMM_NN, "tn": ops.GEMM_TN}[kind]
A = r(M, K) if kind != "tn" else r(K, M)
Bm = r(N, K) if kind == "nt" else r(K, N)
bias = r(N, dt=torch.float32)
if epi == "bf16":
    Cc = torch.empty(M, N, dtype=torch.bfloat16, device=dev); f = lambda: ops.gemm(op, ops.EPI_BF16, A, Bm, Cc, bias=bias)
elif epi == "gelu":
    Cc = torch.empty(M, N, dtype=torch.bfloat16, device=dev); C2 = torch.empty_like(Cc); f = lambda: ops.gemm(op, ops.EPI_BIAS_GELU, A, Bm, Cc, C2=C2, bias=bias)
elif epi == "resid":
    Cc = torch.empty(M, N, dtype=torch.float32, device=dev); R = r(M, N, dt=torch.float32); f = lambda: ops.gemm(op, ops.EPI_RESID_F32, A, Bm, Cc, bias=bias, resid=R)
elif epi == "dgelu":
    Cc = torch.empty(M, N, dtype=torch.bfloat16, device=dev); aux = r(M, N); f = lambda: ops.gemm(op, ops.EPI_DGELU_BF16, A, Bm, Cc, aux=aux)
else:
    Cc = torch.zeros(M, N, dtype=torch.float32, device=dev); f = lambda: ops.gemm(op, ops.EPI_F32, A, Bm, Cc, splits=1, accumulate=False)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"{kind} {epi} M={M} N={N} K={K}: {us:.1f} us  {2.0 * M * N * K / us / 1e6:.1f} TF/s (trace build)")
lib = _lib.load()
nblk = min(1 << 16, 8 * ((M + 63) // 64) * ((N + 127) // 128))
buf = np.zeros((1 << 16) * 8, dtype=np.uint64)
lib.mofo_debug_trace_read.argtypes = [C.c_void_p, C.c_size_t]; lib.mofo_debug_trace_read.restype = C.c_int
assert lib.mofo_debug_trace_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(-1, 8)
t = t[t[:, 0] > 0]
nb = len(t)
if nb == 0:
    sys.exit("no stamped tiles: the persistent / split-K forms stamp a block's THIRD tile and this grid gives every block fewer; "
             "force the one-tile-per-block form with MOFO_GEMM_VARIANT=1 MOFO_GEMM_MI8=0")
ts = t[:, :6].astype(np.int64)
base = ts[:, 0].min()
hw = t[:, 7]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 15
hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print(f"{nb} blocks on {len(np.unique(cuid))} CUs; kernel span {(ts[:,5].max()-base)} clk")
d = np.diff(ts, axis=1)
names = ["first tile load", "main loop", "stage epilogue to LDS", "epilogue global r/w issue", "store ack"]
for i, n in enumerate(names):
    print(f"  {n:28s} mean {d[:, i].mean():9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p50 {np.percentile(d[:, i], 50):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f} clk")
life = ts[:, 5] - ts[:, 0]
print(f"  {'block lifetime':28s} mean {life.mean():9.0f}  p50 {np.percentile(life, 50):9.0f}")
# per-CU concurrency: average number of resident blocks
span = ts[:, 5].max() - base
print(f"  avg resident blocks per CU {life.sum() / span / len(np.unique(cuid)):.2f}; blocks per CU {nb / len(np.unique(cuid)):.1f}")
# start-time waves: how synchronised are block phases on one CU?
one = np.where(cuid == cuid[0])[0]
one = one[np.argsort(ts[one, 0])]
print("  timeline of one CU (block, wave slot, start, +load, +main, +stage, +issue, +ack):")
t0 = ts[one, 0].min()
blk = np.nonzero(buf.reshape(-1, 8)[:, 0] > 0)[0]
for b in one[:12]:
    print("   ", blk[b], hwid[b] & 15, ts[b, 0] - t0, *d[b])

# inside iteration 2 of the VAR 1 main loop (wave 0): frag reads | barrier | DMA issue | MFMA issue | vmcnt wait | barrier
buf2 = np.zeros((1 << 16) * 8, dtype=np.uint64)
lib.mofo_debug_trace_it_read.argtypes = [C.c_void_p, C.c_size_t]; lib.mofo_debug_trace_it_read.restype = C.c_int
assert lib.mofo_debug_trace_it_read(buf2.ctypes.data, buf2.nbytes) == 0
u = buf2.reshape(-1, 8)
u = u[u[:, 0] > 0][:, :7].astype(np.int64)
if len(u):
    dd = np.diff(u, axis=1)
    for i, n in enumerate(["read frags (LDS)", "barrier", "issue LDS-DMA of next tile", "issue 32 MFMAs", "wait vmcnt(0)", "barrier"]):
        print(f"  it2: {n:28s} mean {dd[:, i].mean():8.0f}  p10 {np.percentile(dd[:, i], 10):8.0f}  p50 {np.percentile(dd[:, i], 50):8.0f}  p90 {np.percentile(dd[:, i], 90):8.0f} clk")
    print(f"  it2: total {(u[:, 6] - u[:, 0]).mean():.0f} clk")

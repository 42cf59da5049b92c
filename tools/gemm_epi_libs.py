#!/usr/bin/env python3
"""Interleaved, steady-state timing of the decoder's / encoder's MLP GEMMs with HBM-heavy epilogues (fc1 + bias + GELU forward: writes
h1 and gelu(h1); dfc2 + dGELU backward: reads h1, writes dh1) through SEVERAL builds of libmofo_hip.so in ONE process (GPU box only):
  gemm_epi_libs.py [--rounds N] name=lib.so[:ENV=V,...] ..."""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.getcwd())
import torch
from mofo_amd import ops, _lib
dev = torch.device("cuda:0"); BF16, F32 = torch.bfloat16, torch.float32
args = sys.argv[1:]; rounds = 3
if args and args[0] == "--rounds": rounds, args = int(args[1]), args[2:]
arms = []
for spec in args:
    name, _, rest = spec.partition("="); lib, _, envs = rest.partition(":")
    env = dict(e.split("=", 1) for e in envs.split(",") if e)
    h = C.CDLL(os.path.abspath(lib)); h.mofo_gemm.restype = C.c_int; h.mofo_gemm.argtypes = [C.POINTER(_lib.GemmArgs), C.c_void_p]
    arms.append((name, h, env))
ALLENV = sorted({k for _, _, e in arms for k in e})
def timed(f, warm_ms=25.0, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f(); torch.cuda.synchronize(); e0.record(); f(); e1.record(); torch.cuda.synchronize()
    one = max(e0.elapsed_time(e1), 1e-3)
    for _ in range(int(warm_ms / one) + 1): f()
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
def compare(title, a, flop, nbytes):
    print(f"## {title}"); res = {n: [] for n, _, _ in arms}; st = torch.cuda.current_stream().cuda_stream
    def call(h, env):
        for k in ALLENV: os.environ.pop(k, None)
        os.environ.update(env); rc = h.mofo_gemm(C.byref(a), st); assert rc == 0, rc
    for _ in range(rounds):
        for n, h, env in arms: res[n].append(timed(lambda: call(h, env)))
    base = statistics.median(res[arms[0][0]])
    for n, _, _ in arms:
        us = statistics.median(res[n]); print(f"  {n:30s} {us:8.1f} us (min {min(res[n]):8.1f}) {flop / us / 1e6:6.0f} TF/s {nbytes / us / 1e6:5.2f} TB/s {base / us:5.2f} x")
    sys.stdout.flush()
r = lambda *s, sc=0.5: (torch.randn(*s, device=dev) * sc).to(BF16)
for tag, M, D, H in (("decoder", 50176, 384, 1536), ("encoder", 5120, 768, 3072)):
    x, W1, b1 = r(M, D), r(H, D, sc=0.05), (torch.randn(H, device=dev) * 0.1)
    h1, g = torch.empty(M, H, dtype=BF16, device=dev), torch.empty(M, H, dtype=BF16, device=dev)
    a = ops._gemm_args(ops.GEMM_NT, ops.EPI_BIAS_GELU, x, W1, h1, C2=g, bias=b1)[0]
    compare(f"{tag} fc1 + bias + GELU (NT {M} x {H} x {D})", a, 2.0 * M * H * D, 2.0 * (M * D + 2 * M * H))
    dy, W2 = r(M, D), r(D, H, sc=0.05); dh1 = torch.empty(M, H, dtype=BF16, device=dev)
    a = ops._gemm_args(ops.GEMM_NN, ops.EPI_DGELU_BF16, dy, W2, dh1, aux=h1)[0]
    compare(f"{tag} dfc2 + dGELU (NN {M} x {H} x {D})", a, 2.0 * M * H * D, 2.0 * (M * D + 2 * M * H))

#!/usr/bin/env python3
"""Interleaved, steady-state timing of grouped weight-gradient launches (TN, f32 out, fused bias-gradient column sums) through SEVERAL
builds of libmofo_hip.so and several environment settings in ONE process on one device (GPU box only): build variants with
tools/build_variants.py, then
  gemm_libs_ab.py [--rounds N] name=lib.so[:ENV=V[,ENV=V...]] ...
The first arm is the reference of the ratio column.  Shapes: 4096^3, ViT-B encoder groups of 2 and 6 blocks, one decoder block."""
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
from mofo_amd import ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32
args = sys.argv[1:]
rounds = 3
if args and args[0] == "--rounds":
    rounds, args = int(args[1]), args[2:]
arms = []
for spec in args:
    name, _, rest = spec.partition("=")
    lib, _, envs = rest.partition(":")
    env = dict(e.split("=", 1) for e in envs.split(",") if e)
    h = C.CDLL(os.path.abspath(lib))
    h.mofo_gemm_grouped.restype = C.c_int
    h.mofo_gemm_grouped.argtypes = [C.POINTER(_lib.GemmArgs), C.c_int, C.c_void_p]
    arms.append((name, h, env))
ALLENV = sorted({k for _, _, e in arms for k in e})


def block(R, D, hid, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(BF16).to(dev)
    dY = [r(R, 3 * D), r(R, D), r(R, hid), r(R, D)]
    X = [r(R, D), r(R, D), r(R, D), r(R, hid)]
    G = [torch.zeros(a.shape[1], b.shape[1], dtype=F32, device=dev) for a, b in zip(dY, X)]
    bg = [torch.zeros(a.shape[1], dtype=F32, device=dev) for a in dY]
    return list(zip(dY, X, G, bg))


def group(blocks, splits, colsum=True):
    probs = [(a, b, g, dict(splits=splits, accumulate=False, colsum=bg if colsum else None)) for blk in blocks for a, b, g, bg in blk]
    built = [ops._gemm_args(ops.GEMM_TN, ops.EPI_F32, A, B, C_, **kw) for A, B, C_, kw in probs]
    return (_lib.GemmArgs * len(built))(*[b[0] for b in built]), len(built), probs


def timed(f, warm_ms=25.0, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f(); torch.cuda.synchronize()
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    one = max(e0.elapsed_time(e1), 1e-3)
    for _ in range(int(warm_ms / one) + 1):
        f()
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def compare(title, grp, flop):
    arr, n, _ = grp
    stream = torch.cuda.current_stream().cuda_stream
    print(f"## {title}")
    res = {name: [] for name, _, _ in arms}

    def call(h, env):
        for k in ALLENV:
            os.environ.pop(k, None)
        os.environ.update(env)
        rc = h.mofo_gemm_grouped(arr, n, stream)
        assert rc == 0, rc

    for _ in range(rounds):
        for name, h, env in arms:
            res[name].append(timed(lambda: call(h, env)))
    base = statistics.median(res[arms[0][0]])
    for name, _, _ in arms:
        us = statistics.median(res[name])
        print(f"  {name:34s} {us:8.1f} us (min {min(res[name]):8.1f})  {flop / us / 1e6:6.0f} TF/s  {base / us:5.2f} x")
    sys.stdout.flush()


g = torch.Generator(device="cpu").manual_seed(7)
if os.environ.get("AB_KSWEEP"):
    # one output geometry (4096 x 4096 = 512 units of 256 x 128 = two rounds), reductions 1024 ... 8192: slope = cost of a k-step,
    # intercept = fixed cost of a unit (prologue, epilogue, launch share)
    for K in (1024, 2048, 4096, 8192):
        A = (torch.randn(K, 4096, generator=g) * 0.1).to(BF16).to(dev)
        B = (torch.randn(K, 4096, generator=g) * 0.1).to(BF16).to(dev)
        Cm = torch.zeros(4096, 4096, dtype=F32, device=dev)
        a1 = ops._gemm_args(ops.GEMM_TN, ops.EPI_F32, A, B, Cm, splits=1, accumulate=False)[0]
        compare(f"4096 x 4096 x {K} TN, f32 out ({K // 64} k-steps per unit, two rounds)", ((_lib.GemmArgs * 1)(a1), 1, None), 2.0 * 4096 * 4096 * K)
    sys.exit(0)
A = (torch.randn(4096, 4096, generator=g) * 0.1).to(BF16).to(dev)
B = (torch.randn(4096, 4096, generator=g) * 0.1).to(BF16).to(dev)
Cm = torch.zeros(4096, 4096, dtype=F32, device=dev)
a1 = ops._gemm_args(ops.GEMM_TN, ops.EPI_F32, A, B, Cm, splits=1, accumulate=False)[0]
compare("4096^3 TN, f32 out", ((_lib.GemmArgs * 1)(a1), 1, None), 2.0 * 4096 ** 3)
enc = [block(5120, 768, 3072, 100 + i) for i in range(6)]
fl = 2.0 * 5120 * 768 * 768 * 12
compare("ViT-B encoder, 2 blocks (432 units of 256 x 128)", group(enc[:2], 1), 2 * fl)
compare("ViT-B encoder, 6 blocks (1296 units)", group(enc, 1), 6 * fl)
compare("ViT-B encoder, 6 blocks, no column sums", group(enc, 1, colsum=False), 6 * fl)

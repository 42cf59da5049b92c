#!/usr/bin/env python3
"""gemm8 (256 x 256 counted-vmcnt GEMM, csrc/gemm8.h) against the 128 x 128 family, in ONE process on one device (GPU box only).

For every shape: both paths through the C-ABI (MOFO_GEMM8=0 / 1 is read per call), the result of each against an fp32 torch
product of the same bf16 operands, then interleaved timing rounds (median and min, guide rule 24).  Random operands.
usage: gemm8_ab.py [calib|model|wgrad|all] [rounds]"""
import os
import sys
import statistics

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mofo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32


def r(*s, dt=BF16, scale=0.5):
    return (torch.randn(*s, device=dev) * scale).to(dt)


def gelu(x):
    return torch.nn.functional.gelu(x)


def make(op, epi, M, N, K, splits=1):
    """returns (run(), check() -> max rel err) for one problem"""
    if op == ops.GEMM_NT:
        A, B = r(M, K), r(N, K, scale=0.05)
        ref = lambda: A.float() @ B.float().t()
    elif op == ops.GEMM_NN:
        A, B = r(M, K), r(K, N, scale=0.05)
        ref = lambda: A.float() @ B.float()
    else:
        A, B = r(K, M, scale=0.1), r(K, N, scale=0.1)
        ref = lambda: A.float().t() @ B.float()
    bias = r(N, dt=F32)
    kw = {}
    outs = []
    if epi == ops.EPI_BF16:
        Cc = torch.empty(M, N, dtype=BF16, device=dev)
        kw = dict(bias=bias)
        want = lambda: ref() + bias
        outs = [(Cc, want)]
    elif epi == ops.EPI_BIAS_GELU:
        Cc = torch.empty(M, N, dtype=BF16, device=dev)
        C2 = torch.empty(M, N, dtype=BF16, device=dev)
        kw = dict(bias=bias, C2=C2)
        outs = [(Cc, lambda: ref() + bias), (C2, lambda: gelu(ref() + bias))]
    elif epi == ops.EPI_RESID_F32:
        Cc = torch.empty(M, N, dtype=F32, device=dev)
        R = r(M, N, dt=F32)
        kw = dict(bias=bias, resid=R)
        outs = [(Cc, lambda: ref() + bias + R)]
    elif epi == ops.EPI_RESID_BF16:
        Cc = torch.empty(M, N, dtype=BF16, device=dev)
        R = r(M, N)
        kw = dict(bias=bias, aux=R)
        outs = [(Cc, lambda: ref() + bias + R.float())]
    elif epi == ops.EPI_DGELU_BF16:
        Cc = torch.empty(M, N, dtype=BF16, device=dev)
        H = r(M, N, scale=1.0)
        def want():
            h = H.float().requires_grad_(True)
            g, = torch.autograd.grad(gelu(h).sum(), h)
            return ref() * g
        kw = dict(aux=H)
        outs = [(Cc, want)]
    elif epi == ops.EPI_F32:
        Cc = torch.zeros(M, N, dtype=F32, device=dev)
        kw = dict(splits=splits, accumulate=False)
        outs = [(Cc, ref)]
    else:
        raise ValueError(epi)

    def run():
        if epi == ops.EPI_F32 and splits > 1:
            Cc.zero_()
        ops.gemm(op, epi, A, B, Cc, **kw)

    def check():
        run()
        torch.cuda.synchronize()
        worst = 0.0
        for out, w in outs:
            wv = w()
            err = (out.float() - wv).norm() / wv.norm()
            worst = max(worst, float(err))
            # element-wise: no wrong tile may hide in a norm
            bad = ((out.float() - wv).abs() > 0.03 * wv.abs().max()).sum().item()
            if bad:
                worst = max(worst, 1.0 + bad)
        return worst
    return run, check


def time_pair(run, rounds, iters, warm=0):
    t = {0: [], 1: []}
    for _ in range(rounds):
        for mode in (0, 1):
            os.environ["MOFO_GEMM8"] = str(mode)
            # ~20 ms of the same variant first: a short burst after another kernel is timed in a transient (the 256-tile kernel
            # measured 20-35 % slower in 2-ms bursts than in steady state; inside the training step the chip never idles)
            for _ in range(warm):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            t[mode].append(e0.elapsed_time(e1) / iters * 1e3)
    return t


NT, NN, TN = ops.GEMM_NT, ops.GEMM_NN, ops.GEMM_TN
E = ops
SETS = {
    "calib": [("4096^3 NT bf16", NT, E.EPI_BF16, 4096, 4096, 4096, 1), ("8192^3 NT bf16", NT, E.EPI_BF16, 8192, 8192, 8192, 1),
              ("4096^3 NN bf16", NN, E.EPI_BF16, 4096, 4096, 4096, 1), ("4096^3 TN f32", TN, E.EPI_F32, 4096, 4096, 4096, 1),
              ("8192^2 x 4096 NT", NT, E.EPI_BF16, 8192, 8192, 4096, 1), ("4096^2 x 16384 NT", NT, E.EPI_BF16, 4096, 4096, 16384, 1),
              ("16384x4096x4096 NT", NT, E.EPI_BF16, 16384, 4096, 4096, 1), ("8192^3 NN bf16", NN, E.EPI_BF16, 8192, 8192, 8192, 1)],
    "layout": [("4096^3 NT bf16", NT, E.EPI_BF16, 4096, 4096, 4096, 1), ("4096^3 NT f32", NT, E.EPI_F32, 4096, 4096, 4096, 1),
               ("4096^3 NN bf16", NN, E.EPI_BF16, 4096, 4096, 4096, 1), ("4096^3 TN f32", TN, E.EPI_F32, 4096, 4096, 4096, 1),
               ("2304x768x20480 TN", TN, E.EPI_F32, 2304, 768, 20480, 1), ("3072x3072x20480 TN", TN, E.EPI_F32, 3072, 3072, 20480, 1),
               ("3072x3072x20480 NT", NT, E.EPI_F32, 3072, 3072, 20480, 1)],
    "model": [
        ("enc qkv   NT bf16", NT, E.EPI_BF16, 5120, 2304, 768, 1),
        ("enc fc1   NT gelu", NT, E.EPI_BIAS_GELU, 5120, 3072, 768, 1),
        ("enc fc2   NT rf32", NT, E.EPI_RESID_F32, 5120, 768, 3072, 1),
        ("enc dfc2  NN dgelu", NN, E.EPI_DGELU_BF16, 5120, 3072, 768, 1),
        ("enc dfc1  NN bf16", NN, E.EPI_BF16, 5120, 768, 3072, 1),
        ("enc dqkv  NN bf16", NN, E.EPI_BF16, 5120, 768, 2304, 1),
        ("dec qkv   NT bf16", NT, E.EPI_BF16, 50176, 1152, 384, 1),
        ("dec proj  NT rbf16", NT, E.EPI_RESID_BF16, 50176, 384, 384, 1),
        ("dec fc1   NT gelu", NT, E.EPI_BIAS_GELU, 50176, 1536, 384, 1),
        ("dec fc2   NT rbf16", NT, E.EPI_RESID_BF16, 50176, 384, 1536, 1),
        ("dec head  NT bf16", NT, E.EPI_BF16, 45056, 1536, 384, 1),
        ("dec dfc2  NN dgelu", NN, E.EPI_DGELU_BF16, 50176, 1536, 384, 1),
        ("dec dfc1  NN bf16", NN, E.EPI_BF16, 50176, 384, 1536, 1),
        ("dec dqkv  NN bf16", NN, E.EPI_BF16, 50176, 384, 1152, 1),
        ("dec dhead NN bf16", NN, E.EPI_BF16, 45056, 384, 1536, 1),
    ],
    "vitl": [   # BASELINE configs[4] widths: ViT-L, 32 frames, 32 clips (encoder 10 240 rows, decoder 100 352)
        ("L enc qkv  NT bf16", NT, E.EPI_BF16, 10240, 3072, 1024, 1),
        ("L enc proj NT rf32", NT, E.EPI_RESID_F32, 10240, 1024, 1024, 1),
        ("L enc fc1  NT gelu", NT, E.EPI_BIAS_GELU, 10240, 4096, 1024, 1),
        ("L enc fc2  NT rf32", NT, E.EPI_RESID_F32, 10240, 1024, 4096, 1),
        ("L enc dfc2 NN dgelu", NN, E.EPI_DGELU_BF16, 10240, 4096, 1024, 1),
        ("L enc dfc1 NN bf16", NN, E.EPI_BF16, 10240, 1024, 4096, 1),
        ("L enc dqkv NN bf16", NN, E.EPI_BF16, 10240, 1024, 3072, 1),
        ("L dec qkv  NT bf16", NT, E.EPI_BF16, 100352, 1536, 512, 1),
        ("L dec fc1  NT gelu", NT, E.EPI_BIAS_GELU, 100352, 2048, 512, 1),
        ("L dec fc2  NT rbf16", NT, E.EPI_RESID_BF16, 100352, 512, 2048, 1),
        ("L dec dfc1 NN bf16", NN, E.EPI_BF16, 100352, 512, 2048, 1),
        ("L w.qkv    TN f32", TN, E.EPI_F32, 3072, 1024, 10240, 1),
        ("L w.fc1    TN f32", TN, E.EPI_F32, 4096, 1024, 10240, 1),
        ("L w.fc2    TN f32", TN, E.EPI_F32, 1024, 4096, 10240, 1),
    ],
    "wgrad": [
        ("enc w.qkv TN f32", TN, E.EPI_F32, 2304, 768, 5120, 1),
        ("enc w.fc1 TN f32", TN, E.EPI_F32, 3072, 768, 5120, 1),
        ("enc w.fc2 TN f32", TN, E.EPI_F32, 768, 3072, 5120, 1),
        ("dec w.fc1 TN f32 s7", TN, E.EPI_F32, 1536, 384, 50176, 7),
        ("dec w.qkv TN f32 s7", TN, E.EPI_F32, 1152, 384, 50176, 7),
        ("ragged    TN f32", TN, E.EPI_F32, 1000, 520, 4104, 1),
    ],
}

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    names = list(SETS) if which == "all" else which.split(",")
    print(f"{'shape':22s} {'M':>6s} {'N':>5s} {'K':>6s} | err old     err new    | old us (min)      new us (min)      | TF/s old  new   ratio", flush=True)
    for nm in names:
        for tag, op, epi, M, N, K, sp in SETS[nm]:
            run, check = make(op, epi, M, N, K, sp)
            os.environ["MOFO_GEMM8"] = "0"
            e_old = check()
            os.environ["MOFO_GEMM8"] = "1"
            e_new = check()
            fl = 2.0 * M * N * K
            iters = max(3, min(50, int(2e-3 / (fl / 8e14))))
            warm = max(10, int(20e-3 / max(fl / 8e14, 2e-5)))
            t = time_pair(run, rounds, iters, warm)
            m0, m1 = statistics.median(t[0]), statistics.median(t[1])
            print(f"{tag:22s} {M:6d} {N:5d} {K:6d} | {e_old:9.2e} {e_new:9.2e} {'OK ' if e_new < 2e-2 else 'BAD'}| "
                  f"{m0:8.1f} ({min(t[0]):7.1f}) {m1:8.1f} ({min(t[1]):7.1f}) | {fl / m0 / 1e6:7.0f} {fl / m1 / 1e6:7.0f}  {m0 / m1:5.2f}x", flush=True)
            del run, check
            torch.cuda.empty_cache()

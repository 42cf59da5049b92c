#!/usr/bin/env python3
"""A/B of GEMM routing switches that the library reads per call, in ONE process, interleaved, steady state (GPU only).
usage: gemm_env_ab.py "NAME=VAL[,NAME=VAL...]" ["NAME=VAL..." ...] [--shapes enc768|dec384] [--rounds 5]
Every argument is one variant (a set of environment assignments; "-" = the defaults); the first is the baseline."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mofo_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm8_ab import make  # noqa: E402

NT, NN, E = ops.GEMM_NT, ops.GEMM_NN, ops
SETS = {
    "enc768": [("enc proj  NT rf32", NT, E.EPI_RESID_F32, 5120, 768, 768), ("enc dproj NN bf16", NN, E.EPI_BF16, 5120, 768, 768),
               ("enc qkv   NT bf16", NT, E.EPI_BF16, 5120, 2304, 768), ("enc fc1   NT gelu", NT, E.EPI_BIAS_GELU, 5120, 3072, 768),
               ("enc dfc2  NN dgelu", NN, E.EPI_DGELU_BF16, 5120, 3072, 768), ("enc->dec  NT bf16", NT, E.EPI_BF16, 5120, 384, 768)],
    "dec384": [("dec qkv   NT bf16", NT, E.EPI_BF16, 50176, 1152, 384), ("dec proj  NT rbf16", NT, E.EPI_RESID_BF16, 50176, 384, 384),
               ("dec fc1   NT gelu", NT, E.EPI_BIAS_GELU, 50176, 1536, 384), ("dec fc2   NT rbf16", NT, E.EPI_RESID_BF16, 50176, 384, 1536),
               ("dec dfc2  NN dgelu", NN, E.EPI_DGELU_BF16, 50176, 1536, 384), ("dec dfc1  NN bf16", NN, E.EPI_BF16, 50176, 384, 1536),
               ("dec dqkv  NN bf16", NN, E.EPI_BF16, 50176, 384, 1152), ("dec dproj NN bf16", NN, E.EPI_BF16, 50176, 384, 384)],
}


def main():
    args = sys.argv[1:]
    shapes, rounds, variants = "enc768", 5, []
    while args:
        a = args.pop(0)
        if a == "--shapes":
            shapes = args.pop(0)
        elif a == "--rounds":
            rounds = int(args.pop(0))
        else:
            variants.append((a, {} if a == "-" else dict(kv.split("=", 1) for kv in a.split(","))))
    names = sorted({k for _, d in variants for k in d})

    def apply(d):
        for k in names:
            if k in d:
                os.environ[k] = d[k]
            else:
                os.environ.pop(k, None)

    print(f"{'shape':22s} {'M':>6s} {'N':>5s} {'K':>5s} | " + " | ".join(f"{v[0][:26]:>26s}" for v in variants), flush=True)
    for tag, op, epi, M, N, K in SETS[shapes]:
        run, check = make(op, epi, M, N, K)
        errs = []
        for _, d in variants:
            apply(d)
            errs.append(check())
        fl = 2.0 * M * N * K
        iters = max(5, min(50, int(2e-3 / (fl / 6e14))))
        warm = max(10, int(20e-3 / max(fl / 6e14, 2e-5)))
        t = [[] for _ in variants]
        for _ in range(rounds):
            for i, (_, d) in enumerate(variants):
                apply(d)
                for _ in range(warm):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                t[i].append(e0.elapsed_time(e1) / iters * 1e3)
        med = [statistics.median(x) for x in t]
        print(f"{tag:22s} {M:6d} {N:5d} {K:5d} | " + " | ".join(f"{m:7.1f} us {fl / m / 1e6:5.0f} TF/s {med[0] / m:4.2f}x{'' if e < 2e-2 else ' BAD'}" for m, e in zip(med, errs)), flush=True)
        del run, check
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

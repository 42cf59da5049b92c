#!/usr/bin/env python3
"""gemm_k2 (one 128 x 128 tile per CU, two K-halves with two-stage rings; csrc/gemm_k2.h) against the forms that carried the same
shapes before (64 x 128 tiles: in-block split-K / two-stage / persistent), in ONE process on one device, steady state (GPU only).

usage: gemm_k2_ab.py [rounds]          MOFO_GEMM_K2 = 0 | 1 and MOFO_GEMM_K2_STAG = 0 | 1 are read per call"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mofo_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm8_ab import make  # noqa: E402

NT, NN = ops.GEMM_NT, ops.GEMM_NN
E = ops
SHAPES = [
    ("enc proj  NT rf32", NT, E.EPI_RESID_F32, 5120, 768, 768),
    ("enc fc2   NT rf32", NT, E.EPI_RESID_F32, 5120, 768, 3072),
    ("enc dproj NN bf16", NN, E.EPI_BF16, 5120, 768, 768),
    ("enc dfc1  NN bf16", NN, E.EPI_BF16, 5120, 768, 3072),
    ("enc dqkv  NN bf16", NN, E.EPI_BF16, 5120, 768, 2304),
    ("enc->dec  NT bf16", NT, E.EPI_BF16, 5120, 384, 768),
    ("e2d dgrad NN bf16", NN, E.EPI_BF16, 5120, 768, 384),
    ("enc qkv   NT bf16 (2.8 tiles/CU: not routed)", NT, E.EPI_BF16, 5120, 2304, 768),
    ("L enc proj NT rf32 (10240 rows)", NT, E.EPI_RESID_F32, 10240, 1024, 1024),
]
MODES = [("old", "0", "1", "1"), ("k2 lock-step", "1", "0", "1"), ("k2 staggered", "1", "1", "1"), ("k2 stag, no rot", "1", "1", "0")]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    print(f"{'shape':46s} {'M':>6s} {'N':>5s} {'K':>5s} | " + " | ".join(f"{m[0]:>16s}" for m in MODES) + " | TF/s old -> best", flush=True)
    for tag, op, epi, M, N, K in SHAPES:
        run, check = make(op, epi, M, N, K)
        errs = []
        for _, k2, stag, rot in MODES:
            os.environ["MOFO_GEMM_K2"], os.environ["MOFO_GEMM_K2_STAG"], os.environ["MOFO_GEMM_K2_ROT"] = k2, stag, rot
            errs.append(check())
        fl = 2.0 * M * N * K
        iters = max(5, min(50, int(2e-3 / (fl / 6e14))))
        warm = max(10, int(20e-3 / max(fl / 6e14, 2e-5)))
        t = {m[0]: [] for m in MODES}
        for _ in range(rounds):
            for name, k2, stag, rot in MODES:
                os.environ["MOFO_GEMM_K2"], os.environ["MOFO_GEMM_K2_STAG"], os.environ["MOFO_GEMM_K2_ROT"] = k2, stag, rot
                for _ in range(warm):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run()
                e1.record()
                torch.cuda.synchronize()
                t[name].append(e0.elapsed_time(e1) / iters * 1e3)
        med = {k: statistics.median(v) for k, v in t.items()}
        best = min(med.values())
        print(f"{tag:46s} {M:6d} {N:5d} {K:5d} | " + " | ".join(f"{med[m[0]]:6.1f} ({min(t[m[0]]):5.1f}) {e:.0e}" for m, e in zip(MODES, errs))
              + f" | {fl / med['old'] / 1e6:5.0f} -> {fl / best / 1e6:5.0f}  {med['old'] / best:4.2f}x" + ("" if max(errs) < 2e-2 else "  BAD"), flush=True)
        del run, check
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

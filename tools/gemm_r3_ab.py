#!/usr/bin/env python3
"""Same-process, steady-state A/B of the grouped weight gradients (TN, f32 out, fused bias-gradient column sums) of the ViT-B step
through the 128 x 128 one-stage kernel (three blocks per CU) and through the 256 x 128 three-stage ring kernel (csrc/gemm_r3.h),
GPU box only.  Every timed block is preceded by >= 25 ms of the same variant (a one-block-per-CU kernel that follows a differently
shaped one runs in a transient for its first milliseconds: DESIGN.md section 4c).
usage: gemm_r3_ab.py [rounds]"""
import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from mofo_amd import ops

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def block(R, D, hid, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(BF16).to(dev)
    dY = [r(R, 3 * D), r(R, D), r(R, hid), r(R, D)]
    X = [r(R, D), r(R, D), r(R, D), r(R, hid)]
    G = [torch.zeros(a.shape[1], b.shape[1], dtype=F32, device=dev) for a, b in zip(dY, X)]
    bg = [torch.zeros(a.shape[1], dtype=F32, device=dev) for a in dY]
    return list(zip(dY, X, G, bg))


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


def launches(blocks, splits, per_launch):
    out = []
    for i in range(0, len(blocks), per_launch):
        out.append([(a, b, g, dict(splits=splits, accumulate=False, colsum=bg)) for blk in blocks[i:i + per_launch] for a, b, g, bg in blk])
    return out


def arm(env, groups):
    def f():
        for k in ("MOFO_GEMM_R3", "MOFO_GEMM_R3_TAIL"):
            os.environ.pop(k, None)
        for k, v in env.items():
            os.environ[k] = v
        for probs in groups:
            ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, probs)
    return f


def compare(title, blocks, flop, arms):
    print(f"## {title}")
    res = {name: [] for name, _, _ in arms}
    for _ in range(ROUNDS):
        for name, env, groups in arms:
            res[name].append(timed(arm(env, groups)))
    base = statistics.median(res[arms[0][0]])
    for name, env, groups in arms:
        us = statistics.median(res[name])
        print(f"  {name:46s} {us:8.1f} us (min {min(res[name]):8.1f})  {flop / us / 1e6:6.0f} TF/s  {base / us:5.2f} x")
    sys.stdout.flush()


def check(blocks, splits):
    """results of the two routes agree (and the ring kernel's plan is honoured: flagged destinations zeroed)"""
    probs = launches(blocks[:1], splits, 1)[0]
    os.environ["MOFO_GEMM_R3"] = "0"
    ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, probs)
    want = [p[2].clone() for p in probs]
    os.environ["MOFO_GEMM_R3"] = "1"
    used, shared = ops.gemm_grouped_plan(ops.GEMM_TN, ops.EPI_F32, probs)
    for p, sh in zip(probs, shared):
        p[2].zero_() if sh or splits > 1 else p[2].fill_(float("nan"))
    ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, probs)
    torch.cuda.synchronize()
    worst = max(float(((p[2] - w).abs().max() / w.abs().max())) for p, w in zip(probs, want))
    print(f"  (ring vs 128-family, one block: max relative element difference {worst:.2e}; shared destinations {sum(shared)} of {len(shared)})")
    assert worst < 1e-3


R3 = {"MOFO_GEMM_R3": "1", "MOFO_GEMM_R3_TAIL": "1"}
R3N = {"MOFO_GEMM_R3": "1", "MOFO_GEMM_R3_TAIL": "0"}
OLD = {"MOFO_GEMM_R3": "0"}
AUTO = {}

print("# tools/gemm_r3_ab.py: grouped weight gradients, us per group of blocks (median of %d interleaved rounds, steady state)" % ROUNDS)
enc = [block(5120, 768, 3072, 100 + i) for i in range(7)]
fl = 2.0 * 5120 * 768 * 768 * 12
check(enc, 1)
for nb in (1, 2, 3, 5, 6, 7):
    arms = [("128 x 128 x3/CU (launches of <= 3 blocks)", OLD, launches(enc[:nb], 1, 3)),
            ("ring 256 x 128, tail chunks", R3, launches(enc[:nb], 1, nb)),
            ("ring 256 x 128, plain rounds", R3N, launches(enc[:nb], 1, nb)),
            ("default routing (r3_wanted / r3_tail_auto), one group", AUTO, launches(enc[:nb], 1, nb) if nb <= 3 else None)]
    arms = [a for a in arms if a[2] is not None]
    compare(f"ViT-B encoder, {nb} block(s): 5 120 token rows, D = 768", enc[:nb], fl * nb, arms)
del enc
torch.cuda.empty_cache()

dec = [block(50176, 384, 1536, 200)]
fld = 2.0 * 50176 * 384 * 384 * 12
check(dec, 4)
compare("ViT-B decoder, 1 block: 50 176 token rows, D = 384", dec, fld,
        [("128 x 128 x3/CU, 7 splits (today)", OLD, launches(dec, 7, 1)),
         ("ring 256 x 128, 4 splits (252 units)", R3, launches(dec, 4, 1)),
         ("ring 256 x 128, 8 splits (504 units)", R3, launches(dec, 8, 1)),
         ("ring 256 x 128, 12 splits (756 units)", R3, launches(dec, 12, 1)),
         ("ring 256 x 128, 1 split (63 units, all tail)", R3, launches(dec, 1, 1))])
del dec
torch.cuda.empty_cache()

g = torch.Generator(device="cpu").manual_seed(7)
A = (torch.randn(4096, 4096, generator=g) * 0.1).to(BF16).to(dev)
B = (torch.randn(4096, 4096, generator=g) * 0.1).to(BF16).to(dev)
C = torch.zeros(4096, 4096, dtype=F32, device=dev)
one = [[(A, B, C, dict(splits=1, accumulate=False))]]
compare("4096^3 TN, f32 out (512 units of 256 x 128 = two rounds)", None, 2.0 * 4096 ** 3,
        [("128 x 128 x3/CU", OLD, one), ("ring 256 x 128", R3, one)])

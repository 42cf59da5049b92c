#!/usr/bin/env python3
"""Same-process, steady-state A/B of the DECODER's weight gradients of the ViT-B step (4 blocks + head, 50 176 token rows) and of
the encoder's groups, GPU box only:
  * today's route: one grouped launch per block on the 128 x 128 one-stage kernel, reduction split 7 ways with f32 atomics;
  * mofo_gemm_wgrad_sliced: ONE launch for the whole pass, reduction sliced over the 8 XCDs, partial sums + a reduce kernel
    (384 x 128 tiles = gemm_r4, or 256 x 128 = gemm_r3 with MOFO_WGRAD_TILE=256), also block by block;
  * the encoder's 7 + 5 block groups on gemm_r3 (256 x 128) against gemm_r4 (384 x 128, MOFO_GEMM_R4=1).
usage: wgrad_dec_ab.py [rounds]"""
import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from mofo_amd import ops

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32
QUICK = "--quick" in sys.argv          # two arms only (decoder sliced on gemm_r4, encoder 7 blocks on gemm_r4): build-variant screens (MOFO_HIP_LIB)
_args = [a for a in sys.argv[1:] if not a.startswith("--")]
ROUNDS = int(_args[0]) if _args else 3


def rnd(g, *s):
    return (torch.randn(*s, generator=g) * 0.1).to(BF16).to(dev)


def block(R, D, hid, seed, Rqkv=None, Rrest=None):
    """(dY, X, G, bias_grad, colsum_skip) of fc2, fc1, proj, qkv (runtime._block_bwd's order)"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    Rq, Rr = Rqkv or R, Rrest or R
    dY = [rnd(g, Rr, D), rnd(g, Rr, hid), rnd(g, Rr, D), rnd(g, Rq, 3 * D)]
    X = [rnd(g, Rr, hid), rnd(g, Rr, D), rnd(g, Rr, D), rnd(g, Rq, D)]
    G = [torch.zeros(a.shape[1], b.shape[1], dtype=F32, device=dev) for a, b in zip(dY, X)]
    bg = [torch.zeros(a.shape[1], dtype=F32, device=dev) for a in dY]
    skip = [(0, 0), (0, 0), (0, 0), (D, 2 * D)]
    return list(zip(dY, X, G, bg, skip))


def timed(f, warm_ms=25.0, iters=10):
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


def probs_of(blocks, **kw):
    return [(a, b, g, dict(accumulate=False, colsum=bg, colsum_skip=sk, **kw)) for blk in blocks for a, b, g, bg, sk in blk]


ENVK = ("MOFO_GEMM_R3", "MOFO_GEMM_R3_TAIL", "MOFO_GEMM_R4", "MOFO_WGRAD_TILE")


def setenv(env):
    for k in ENVK:
        os.environ.pop(k, None)
    os.environ.update(env)


def grouped(env, groups):
    def f():
        setenv(env)
        for p in groups:
            ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, p)
    return f


def sliced(env, groups, ws, slices=8):
    def f():
        setenv(env)
        for p in groups:
            ops.gemm_wgrad_sliced(p, ws, slices)
    return f


def compare(title, flop, arms):
    print(f"## {title}")
    res = {name: [] for name, _ in arms}
    for _ in range(ROUNDS):
        for name, f in arms:
            res[name].append(timed(f))
    base = statistics.median(res[arms[0][0]])
    for name, _ in arms:
        us = statistics.median(res[name])
        print(f"  {name:64s} {us:8.1f} us (min {min(res[name]):8.1f})  {flop / us / 1e6:6.0f} TF/s  {base / us:5.2f} x")
    sys.stdout.flush()


print("# tools/wgrad_dec_ab.py: weight gradients of a pass, us (median of %d interleaved rounds, steady state)" % ROUNDS)
B, N, NV, D, HID = 32, 1568, 160, 384, 1536
R, RM, RC = B * N, B * (N - NV), B * NV + N
dec = [block(R, D, HID, 303, Rrest=RM), block(R, D, HID, 302), block(R, D, HID, 301), block(R, D, HID, 300, Rqkv=RC)]   # backward order
g = torch.Generator(device="cpu").manual_seed(77)
head = [(rnd(g, RM, 1536), rnd(g, RM, D), torch.zeros(1536, D, dtype=F32, device=dev), torch.zeros(1536, dtype=F32, device=dev), (0, 0))]
allp = [head] + dec
flop = sum(2.0 * a.shape[0] * a.shape[1] * b.shape[1] for blk in allp for a, b, *_ in blk)
ws = torch.empty(ops.gemm_wgrad_sliced_ws(probs_of(allp), 8), dtype=F32, device=dev)
print(f"  decoder + head: {flop / 1e9:.1f} GFLOP, {sum(len(b) for b in allp)} problems, workspace {ws.numel() * 4 / 2**20:.0f} MiB")

if QUICK:
    tag = os.environ.get("MOFO_HIP_LIB", "in-tree build")
    if "--only-enc" not in sys.argv:
        compare(f"[{tag}] ViT-B decoder, 4 blocks + head", flop, [("sliced, ONE launch, 384 x 128 (gemm_r4), 8 slices + reduce", sliced({}, [probs_of(allp)], ws))])
    del dec, head, allp, ws
    if "--only-dec" in sys.argv:
        sys.exit(0)
    torch.cuda.empty_cache()
    enc = [block(5120, 768, 3072, 100 + i) for i in range(7)]
    compare(f"[{tag}] ViT-B encoder, 7 blocks in one launch", 2.0 * 5120 * 768 * 768 * 12 * 7,
            [("ring 384 x 128 (gemm_r4), by shape", grouped({"MOFO_GEMM_R3": "1"}, [probs_of(enc)]))])
    sys.exit(0)

# correctness of the new route against today's (zeroed destinations for the atomics of today's route)
setenv({"MOFO_GEMM_R3": "0"})
for blk in allp:
    for a, b, G, bg, sk in blk:
        G.zero_(); bg.zero_()
ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, probs_of([head] + dec[:1], splits=7))
for blk in dec[1:]:
    ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, probs_of([blk], splits=7))
want = [(G.clone(), bg.clone()) for blk in allp for _, _, G, bg, _ in blk]
for blk in allp:
    for a, b, G, bg, sk in blk:
        G.fill_(float("nan")); bg.zero_()
for env in ({}, {"MOFO_WGRAD_TILE": "256"}):
    setenv(env)
    for blk in allp:
        for a, b, G, bg, sk in blk:
            G.fill_(float("nan")); bg.zero_()
    ops.gemm_wgrad_sliced(probs_of(allp), ws, 8)
    torch.cuda.synchronize()
    worst = max(float((G - w).abs().max() / w.abs().max()) for (w, _), G in zip(want, [G for blk in allp for _, _, G, _, _ in blk]))
    worstb = max(float((bg - wb).abs().max() / wb.abs().max()) for (_, wb), bg in zip(want, [bg for blk in allp for _, _, _, bg, _ in blk]))
    print(f"  sliced {env or 'default (384-row tiles)'} vs today's route: max relative element difference {worst:.2e} (weights), {worstb:.2e} (biases)")
    assert worst < 1e-3 and worstb < 1e-3

OLD = {"MOFO_GEMM_R3": "0"}
today = [probs_of([head] + dec[:1], splits=7)] + [probs_of([blk], splits=7) for blk in dec[1:]]
compare("ViT-B decoder, 4 blocks + head: 50 176 token rows, D = 384", flop,
        [("today: 128 x 128 x3/CU, one launch per block, 7 splits (atomics)", grouped(OLD, today)),
         ("sliced, ONE launch, 384 x 128 (gemm_r4), 8 slices + reduce", sliced({}, [probs_of(allp)], ws)),
         ("sliced, ONE launch, 256 x 128 (gemm_r3), 8 slices + reduce", sliced({"MOFO_WGRAD_TILE": "256"}, [probs_of(allp)], ws)),
         ("sliced, one launch per block, 384 x 128, 8 slices + reduce", sliced({}, [probs_of([head] + dec[:1])] + [probs_of([b]) for b in dec[1:]], ws)),
         ("sliced, ONE launch, 384 x 128, 4 slices + reduce", sliced({}, [probs_of(allp)], ws, 4)),
         ("ring 384 x 128 not sliced, ONE launch, 2 splits, tail chunks (atomics)",
          grouped({"MOFO_GEMM_R3": "1", "MOFO_GEMM_R4": "1", "MOFO_GEMM_R3_TAIL": "1"}, [probs_of(allp, splits=2)]))])
del dec, head, allp, today, ws, want
torch.cuda.empty_cache()

enc = [block(5120, 768, 3072, 100 + i) for i in range(7)]
fl = 2.0 * 5120 * 768 * 768 * 12
R3 = {"MOFO_GEMM_R3": "1", "MOFO_GEMM_R4": "0"}
R4 = {"MOFO_GEMM_R3": "1", "MOFO_GEMM_R4": "1"}
for nb in (2, 3, 5, 7):
    compare(f"ViT-B encoder, {nb} block(s) in one launch: 5 120 token rows, D = 768", fl * nb,
            [("ring 256 x 128 (gemm_r3), tail by shape", grouped(R3, [probs_of(enc[:nb])])),
             ("ring 384 x 128 (gemm_r4), tail by shape", grouped(R4, [probs_of(enc[:nb])])),
             ("ring 384 x 128 (gemm_r4), plain rounds", grouped(dict(R4, MOFO_GEMM_R3_TAIL="0"), [probs_of(enc[:nb])])),
             ("ring 384 x 128 (gemm_r4), tail chunks", grouped(dict(R4, MOFO_GEMM_R3_TAIL="1"), [probs_of(enc[:nb])]))])

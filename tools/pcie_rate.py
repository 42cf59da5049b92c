#!/usr/bin/env python3
"""PCIe-inclusive step rate: the same training step as bench.py, but every step's batch starts in (pinned) HOST memory, as
a DataLoader hands it over -- f32 clips (the reference's contract, 9.63 MB per clip) or uint8 frame stacks (2.41 MB per clip,
normalised on the GPU).  `value` of bench.py is the HBM-resident rate; this is the number DESIGN.md quotes next to it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mofo_amd import modeling_pretrain as mp, optim_factory, utils
from mofo_amd.masking_generator import TubeMaskingGenerator

dev = torch.device("cuda:0")
B, steps, warm = 32, 20, 5


class A: opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1.5e-4 * B / 256, 0.05, 1e-8, (0.9, 0.95)


torch.manual_seed(0); np.random.seed(0)
model = mp.pretrain_videomae_base_patch16_224(decoder_depth=4).to(dev)
model.set_visible_tokens(160)
opt = optim_factory.create_optimizer(A, model)
scaler = utils.NativeScalerWithGradNormCount()
gen = TubeMaskingGenerator((8, 14, 14), 0.9)
mask = torch.from_numpy(np.stack([gen() for _ in range(B)]).astype(np.uint8))
host = {"f32": [torch.randn(B, 3, 16, 224, 224).pin_memory() for _ in range(2)],
        "uint8": [torch.randint(0, 256, (B, 224, 224, 48), dtype=torch.uint8).pin_memory() for _ in range(2)]}
resident = {k: v[0].to(dev) for k, v in host.items()}


def run(kind, from_host, prefetch=False):
    side = torch.cuda.Stream()
    def step(i):
        x = host[kind][i & 1] if from_host else resident[kind]
        loss = model.forward_loss(x, mask)
        opt.zero_grad()
        scaler(loss, opt, clip_grad=None)
        lv = loss.item()
        torch.cuda.synchronize()
        return lv
    for i in range(warm): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def run_prefetched(kind):
    """the same loop fed through utils.DevicePrefetcher (next batch copied on a side stream during the current step)"""
    class Loader:
        def __init__(self, n): self.n = n
        def __len__(self): return self.n
        def __iter__(self): return ((host[kind][i & 1], mask) for i in range(self.n))
    def epoch(n):
        for x, m in utils.DevicePrefetcher(Loader(n), dev):
            loss = model.forward_loss(x, m)
            opt.zero_grad()
            scaler(loss, opt, clip_grad=None)
            loss.item()
            torch.cuda.synchronize()
    epoch(warm)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    epoch(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for kind in ("f32", "uint8"):
    a, b, c = run(kind, False), run(kind, True), run_prefetched(kind)
    print(f"{kind:6s} resident {a:7.3f} ms/step {B / a * 1e3:8.1f} clips/s | from pinned host {b:7.3f} ms/step {B / b * 1e3:8.1f} clips/s "
          f"| prefetched on a side stream {c:7.3f} ms/step {B / c * 1e3:8.1f} clips/s "
          f"(H2D {resident[kind].numel() * resident[kind].element_size() / 1e6:.0f} MB per step)")

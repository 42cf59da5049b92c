"""Experiment (measurement tooling): do two INDEPENDENT half-batch training steps, each on its own HIP stream, finish sooner
than one full-batch step?  Prices a "two micro-batches on two streams" schedule before anyone builds it: the encoder's
kernels are one-round grids whose blocks move through load / MFMA / store phases in lock-step, so a second independent
launch chain could fill their bubbles -- or just fight for the same CUs.

  python tools/exp_concurrent.py [--batch 32] [--steps 20]

Prints ms per step for: one model at B; one model at B/2; two models at B/2 stepping concurrently from two host threads
(own stream, own weight-gradient side stream, own optimizer).  GPU only."""
import argparse
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Args:
    opt = "adamw"
    opt_eps = 1e-8
    opt_betas = (0.9, 0.95)
    weight_decay = 0.05
    momentum = 0.9
    lr = 1.5e-4


def build(B, dev, seed):
    from mofo_amd import modeling_pretrain as mp, optim_factory, utils
    from mofo_amd.masking_generator import TubeMaskingGenerator
    torch.manual_seed(seed)
    model = mp.pretrain_videomae_base_patch16_224(decoder_depth=4).to(dev)
    clips, mask_u8 = model.input_buffers(B, 160)
    clips.normal_()
    np.random.seed(seed)
    mgen = TubeMaskingGenerator((8, 14, 14), 0.9)
    mask_u8.copy_(torch.from_numpy(np.stack([mgen() for _ in range(B)]).astype(np.uint8)))
    mask = mask_u8.clone()
    opt = optim_factory.create_optimizer(_Args, model)
    scaler = utils.NativeScalerWithGradNormCount()

    def step():
        loss = model.forward_loss(clips, mask, True)
        opt.zero_grad()
        scaler(loss, opt, clip_grad=None)
        return loss

    return step


def timed(steps_fns, n, streams):
    """each fn in its own thread on its own stream; returns wall seconds for n steps of every fn"""
    def run(fn, s):
        with torch.cuda.stream(s):
            last = None
            for _ in range(n):
                last = fn()
                last.item()
            s.synchronize()
    torch.cuda.synchronize()
    th = [threading.Thread(target=run, args=(f, s)) for f, s in zip(steps_fns, streams)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sys.setswitchinterval(2e-5)
    B = a.batch
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    full = build(B, dev, 0)
    h1 = build(B // 2, dev, 1)
    h2 = build(B // 2, dev, 2)
    # record the launch lists one model at a time (the recorder is process-global), each on the stream it will run on
    for fn, s in ((full, sA), (h1, sA), (h2, sB)):
        with torch.cuda.stream(s):
            for _ in range(3):
                fn().item()
        torch.cuda.synchronize()
    for rnd in range(3):
        t_full = timed([full], a.steps, [sA]) / a.steps * 1e3
        t_half = timed([h1], a.steps, [sA]) / a.steps * 1e3
        t_two = timed([h1, h2], a.steps, [sA, sB]) / a.steps * 1e3
        print(f"round {rnd}: B={B} {t_full:.3f} ms/step | B={B // 2} alone {t_half:.3f} ms/step | two B={B // 2} concurrently "
              f"{t_two:.3f} ms per pair  ({B / t_full * 1e3:.0f} vs {B / t_two * 1e3:.0f} clips/s)", flush=True)


if __name__ == "__main__":
    main()

"""Experiment (measurement tooling): what does the per-step host round trip cost?  The reference reads the loss (sync #1,
engine_for_pretraining.py:69) and synchronises the device (sync #2, :179) in every step, so the device drains and the next step's
first kernels start from an idle queue.  Variants, same process, interleaved:
  ref    : forward, backward + AdamW enqueued, loss.item(), torch.cuda.synchronize()          (bench.py / the drop-in engine)
  item   : ... loss.item() only (no second synchronize)
  late   : the loss of step i is read AFTER step i + 1 has been enqueued (one event per step; the device-side gate of AdamW keeps a
           bad step from touching the parameters), no device-wide synchronize inside the loop
  none   : no read-back inside the loop at all (upper bound)
GPU only.   python tools/exp_step_bubble.py [--steps 40]"""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    from mofo_amd import modeling_pretrain as mp, optim_factory, utils
    from mofo_amd.masking_generator import TubeMaskingGenerator
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = mp.pretrain_videomae_base_patch16_224(decoder_depth=4).to(dev)
    clips, mask_u8 = model.input_buffers(32, 160)
    clips.normal_()
    np.random.seed(0)
    mgen = TubeMaskingGenerator((8, 14, 14), 0.9)
    mask_u8.copy_(torch.from_numpy(np.stack([mgen() for _ in range(32)]).astype(np.uint8)))
    opt = optim_factory.create_optimizer(_Args, model)
    scaler = utils.NativeScalerWithGradNormCount()
    pinned = torch.zeros(2, dtype=torch.float32).pin_memory()

    def enqueue():
        loss = model.forward_loss(clips, mask_u8, True)
        opt.zero_grad()
        scaler(loss, opt, clip_grad=None)
        return loss

    def run(mode, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        prev = None
        for i in range(n):
            loss = enqueue()
            if mode == "ref":
                loss.item()
                torch.cuda.synchronize()
            elif mode == "item":
                loss.item()
            elif mode == "late":
                ev = torch.cuda.Event()
                pinned[i & 1].copy_(loss.detach().reshape(()), non_blocking=True)
                ev.record()
                if prev is not None:
                    prev[0].synchronize()
                    float(pinned[prev[1]])
                prev = (ev, i & 1)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for m in ("ref", "item", "late", "none"):
        run(m, 5)
    for rnd in range(3):
        print("round", rnd, "  ".join(f"{m} {run(m, a.steps):.3f} ms" for m in ("ref", "item", "late", "none")), flush=True)


if __name__ == "__main__":
    main()

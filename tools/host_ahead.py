"""Measurement tooling: how far ahead of the device does the host run inside one training step?  Prints, per step, the host
time at which the forward, the backward and the optimizer step had been ENQUEUED (perf_counter since the step's start) next
to the step's device time.  GPU only.

  python tools/host_ahead.py [--batch 32] [--steps 5]"""
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
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    from mofo_amd import modeling_pretrain as mp, optim_factory, utils
    from mofo_amd.masking_generator import TubeMaskingGenerator
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = mp.pretrain_videomae_base_patch16_224(decoder_depth=4).to(dev)
    clips, mask_u8 = model.input_buffers(a.batch, 160)
    clips.normal_()
    np.random.seed(0)
    mgen = TubeMaskingGenerator((8, 14, 14), 0.9)
    mask_u8.copy_(torch.from_numpy(np.stack([mgen() for _ in range(a.batch)]).astype(np.uint8)))
    opt = optim_factory.create_optimizer(_Args, model)
    scaler = utils.NativeScalerWithGradNormCount()
    from mofo_amd import ops
    marks = []
    real_adamw = ops.adamw

    def adamw_marked(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        real_adamw(*args, **kw)
        e1.record()
        marks.append((args[0].numel(), e0, e1))

    ops.adamw = adamw_marked
    for it in range(a.steps + 4):
        torch.cuda.synchronize()
        marks.clear()
        e_start = torch.cuda.Event(enable_timing=True)
        e_start.record()
        t0 = time.perf_counter()
        loss = model.forward_loss(clips, mask_u8, True)
        t1 = time.perf_counter()
        opt.zero_grad()
        loss.backward()
        t2 = time.perf_counter()
        scaler_step(scaler, loss, opt)
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        if it >= 4:
            print(f"step {it}: forward enqueued at {1e3 * (t1 - t0):.2f} ms, backward at {1e3 * (t2 - t0):.2f}, optimizer at "
                  f"{1e3 * (t3 - t0):.2f}; device done at {1e3 * (t4 - t0):.2f}", flush=True)
            print("   AdamW launches (parameters: device start -> end, ms since the step's start): "
                  + "  ".join(f"{n / 1e6:.1f}M: {e_start.elapsed_time(e0):.2f}->{e_start.elapsed_time(e1):.2f}" for n, e0, e1 in marks), flush=True)


def scaler_step(scaler, loss, opt):
    """the scaler's tail without its loss.backward() (already done above, timed apart)"""
    class _NoBackward:
        def backward(self):
            pass
    scaler(_NoBackward(), opt, clip_grad=None)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where does the host spend the time between the end-of-step synchronise and the first launches of the next step?  (The kernel trace
shows the GPU idle for ~110 us there and nowhere else in the step: profiles/r06_rocprof_kernel_stats.csv.)  Host timestamps of the
headline step (ViT-B, 32 clips), median over the steps after the warm-up; GPU box only.  usage: step_start_gap.py [steps]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
from mofo_amd import modeling_pretrain as mp, optim_factory, utils, ops
from mofo_amd.masking_generator import TubeMaskingGenerator


class _Args:
    opt = "adamw"; opt_eps = 1e-8; opt_betas = (0.9, 0.95); weight_decay = 0.05; lr = 1.5e-4 * 32 / 256


B = 32
model = mp.pretrain_videomae_base_patch16_224(decoder_depth=4).to(dev)
clips, mask_u8 = model.input_buffers(B, 160)
clips.normal_()
np.random.seed(0)
mgen = TubeMaskingGenerator((8, 14, 14), 0.9)
mask_u8.copy_(torch.from_numpy(np.stack([mgen() for _ in range(B)]).astype(np.uint8)))
opt = optim_factory.create_optimizer(_Args, model)
scaler = utils.NativeScalerWithGradNormCount()

marks = {}
now = time.perf_counter


def wrap(obj, name, tag):
    f = getattr(obj, name)

    def g(*a, **k):
        marks.setdefault(tag + ":in", now())
        r = f(*a, **k)
        marks.setdefault(tag + ":out", now())
        return r
    setattr(obj, name, g)


wrap(ops, "mask_to_indices", "mask_to_indices")
wrap(ops, "replay", "first replay")
rt = model.runtime()
wrap(rt, "forward", "runtime.forward")
wrap(rt.store, "refresh_shadow", "refresh_shadow")
rows = []
t_sync = None
for it in range(steps + 5):
    marks.clear()
    t0 = now()
    for g in opt.param_groups:
        g["lr"] = 1e-4 * g["lr_scale"]
        if g["weight_decay"] > 0:
            g["weight_decay"] = 0.05
    marks["param groups done"] = now()
    loss = model.forward_loss(clips, mask_u8, True)
    marks["forward_loss returned"] = now()
    opt.zero_grad()
    scaler(loss, opt, clip_grad=None)
    marks["backward + AdamW enqueued"] = now()
    lv = loss.item()
    torch.cuda.synchronize()
    t1 = now()
    if it >= 5:
        rows.append({k: (v - t0) * 1e6 for k, v in marks.items()} | {"step": (t1 - t0) * 1e6})
def block(n):
    t = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = now()
        for g in opt.param_groups:
            g["lr"] = 1e-4 * g["lr_scale"]
        loss = model.forward_loss(clips, mask_u8, True)
        opt.zero_grad()
        scaler(loss, opt, clip_grad=None)
        loss.item()
        torch.cuda.synchronize()
        t.append((now() - t0) * 1e3)
    return statistics.median(t)


ab = {True: [], False: []}
for rnd in range(5):
    for early in (True, False):
        mp.EARLY_LAUNCH = early
        block(5)
        ab[early].append(block(40))
mp.EARLY_LAUNCH = True
keys = sorted(rows[0], key=lambda k: statistics.median(r[k] for r in rows))
print(f"# tools/step_start_gap.py: host time since the previous step's synchronise returned, us (median of {len(rows)} steps)")
for k in keys:
    print(f"{k:32s} {statistics.median(r[k] for r in rows):9.1f}")
print("# step, ms (median of 40, five interleaved rounds): launches first " + " / ".join(f"{v:.3f}" for v in ab[True])
      + "   checks first (MOFO_EARLY_LAUNCH=0) " + " / ".join(f"{v:.3f}" for v in ab[False]))

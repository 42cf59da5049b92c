"""Measurement tooling: where does the host spend the time between the END of one training step (the reference's
torch.cuda.synchronize(), engine_for_pretraining.py:179) and the first kernels of the next?  The kernel timeline of a step
(tools/trace_misc.sh) shows the device idle for ~0.27 ms there.  Stamps (perf_counter, us since the synchronize returned):
the first C-ABI call of the step (mask -> index lists), the first replayed launch of the forward list, the end of the forward
enqueue, backward + AdamW enqueued, loss.item() returned, synchronize returned.  GPU only.

  python tools/host_gap.py [--steps 20]"""
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
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    from mofo_amd import modeling_pretrain as mp, optim_factory, utils, ops, _lib
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
    stamps = {}
    t_ref = [0.0]

    def mark(name):
        stamps.setdefault(name, (time.perf_counter() - t_ref[0]) * 1e6)

    real_run, real_replay = ops._run, ops.replay

    def run(*args, **kw):
        mark("first C-ABI call (mask_to_indices)")
        return real_run(*args, **kw)

    def replay(lst):
        mark("first replayed launch list entered")
        out = real_replay(lst)
        mark("forward list enqueued")
        return out

    ops._run, ops.replay = run, replay
    rows = []
    for it in range(a.steps + 5):
        stamps.clear()
        loss = model.forward_loss(clips, mask_u8, True)
        mark("forward_loss returned")
        opt.zero_grad()
        scaler(loss, opt, clip_grad=None)
        mark("backward + AdamW enqueued")
        loss.item()
        mark("loss.item() returned")
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        stamps["synchronize returned"] = (t_end - t_ref[0]) * 1e6
        if it >= 5:
            rows.append(dict(stamps))
        t_ref[0] = t_end
    keys = list(rows[0])
    print(f"host stamps, us since the previous step's synchronize returned (median of {len(rows)} steps):")
    for k in keys:
        print(f"  {k:40s} {np.median([r[k] for r in rows]):9.1f}")
    # what the pieces of the pre-launch path cost on their own (host only)
    rt = model.runtime()
    t0 = time.perf_counter()
    for _ in range(200):
        rt.store._version()
    print(f"store._version() (218 version counters): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
    t0 = time.perf_counter()
    for _ in range(200):
        rt.store.owns()
    print(f"store.owns(): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
    t0 = time.perf_counter()
    for _ in range(200):
        torch.cuda.current_stream()
    print(f"torch.cuda.current_stream(): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
    x = torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        x.add_(1)
        torch.cuda.synchronize()
    print(f"tiny kernel + torch.cuda.synchronize(): {(time.perf_counter() - t0) / 100 * 1e6:.1f} us per pair")


if __name__ == "__main__":
    main()

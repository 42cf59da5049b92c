"""Experiment (measurement tooling): does the AdamW pass hide beside the FORWARD of the next step?  The forward has no
independent filler of its own (the backward has the weight-gradient stream), so an HBM-bound update on a second stream could
fill its kernels' ramps and tails.  Prices a "deferred optimizer step" before anyone builds it.

  python tools/exp_adamw_beside_forward.py [--batch 32]

Prints ms for: forward + loss alone; AdamW alone (on scratch copies of the flat buffers); both at once (AdamW on a side
stream, in forward order, range by range).  GPU only."""
import argparse
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from mofo_amd import modeling_pretrain as mp, ops
    from mofo_amd.masking_generator import TubeMaskingGenerator
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = mp.pretrain_videomae_base_patch16_224(decoder_depth=4).to(dev)
    clips, mask_u8 = model.input_buffers(a.batch, 160)
    clips.normal_()
    np.random.seed(0)
    mgen = TubeMaskingGenerator((8, 14, 14), 0.9)
    mask_u8.copy_(torch.from_numpy(np.stack([mgen() for _ in range(a.batch)]).astype(np.uint8)))
    st = model.runtime().store
    n = st.params.numel()
    p, g, m, v = (torch.zeros(n, device=dev) for _ in range(4))
    sh = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    g.normal_()
    side = torch.cuda.Stream()
    hyper = (1e-4, 0.05, 1e-4, 0.0, 0.9, 0.95, 1e-8, 3)
    # ranges in forward order: quarters of the buffer (the real ones would be patch embed + 3 blocks, ...)
    q = (n // 4096) * 1024
    ranges = [(0, q), (q, 2 * q), (2 * q, 3 * q), (3 * q, n)]

    def fwd():
        return model.forward_loss(clips, mask_u8, True)

    def adamw():
        for lo, hi in ranges:
            ops.adamw(p[lo:hi], g[lo:hi], m[lo:hi], v[lo:hi], sh[lo:hi], st.chunk_group[lo // 1024:hi // 1024], *hyper)

    def timed(f):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    def both():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            adamw()
        fwd()
        main.wait_stream(side)

    for rnd in range(3):
        tf, ta, tb = timed(fwd), timed(adamw), timed(both)
        print(f"round {rnd}: forward+loss {tf:.3f} ms | AdamW {ta:.3f} ms | both at once {tb:.3f} ms  (sum {tf + ta:.3f}; hidden {tf + ta - tb:.3f})", flush=True)


if __name__ == "__main__":
    main()

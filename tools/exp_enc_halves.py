"""Experiment (measurement tooling): is the ENCODER's launch-bound chain (5 120 token rows: every kernel is at most one round of
tiles, ~190 dependent launches of 7-40 us) shorter as TWO independent half-batch chains on two HIP streams?  The kernels of one
chain leave the chip idle at every boundary (ramp, tail, ~1.5 us launch gap); a second, de-phased chain could fill those holes --
or the halved grids (2 560 rows) could simply run at the same latency-bound time each and lose.  Prices the schedule before
anyone rebuilds runtime.py around it.  Weight gradients are left out (they would stay full-batch launches in such a design).

  python tools/exp_enc_halves.py [--batch 32] [--reps 30]

Prints ms for: encoder forward at B on one stream | two B/2 forwards on two streams; the same for the backward chain (dgrad GEMMs,
attention backward, LayerNorm backward; no weight gradients).  Every variant is a captured hipGraph, so the host is out of the
picture.  GPU only."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def build(B, dev, seed):
    from mofo_amd import modeling_pretrain as mp
    from mofo_amd.masking_generator import TubeMaskingGenerator
    torch.manual_seed(seed)
    model = mp.pretrain_videomae_base_patch16_224(decoder_depth=4).to(dev)
    clips, mask_u8 = model.input_buffers(B, 160)
    clips.normal_()
    np.random.seed(seed)
    mgen = TubeMaskingGenerator((8, 14, 14), 0.9)
    mask_u8.copy_(torch.from_numpy(np.stack([mgen() for _ in range(B)]).astype(np.uint8)))
    loss = model.forward_loss(clips, mask_u8, True)          # builds the workspace, runs one whole step's forward
    model.runtime().store.zero_grads()
    loss.backward()
    torch.cuda.synchronize()
    rt = model.runtime()
    w = next(iter(rt._ws.values()))
    rt._wgrad_group = lambda group: None                     # chain only: no weight-gradient launches
    return model, rt, w


def capture(fn, stream):
    """fn() as a hipGraph captured on ``stream`` (after two eager runs: the runtime records its launch list on the first)"""
    with torch.cuda.stream(stream):
        fn()
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(g, stream=stream):
            fn()
    torch.cuda.synchronize()
    return g


def timed(graphs, streams, reps):
    torch.cuda.synchronize()
    for _ in range(3):
        for g, s in zip(graphs, streams):
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for g, s in zip(graphs, streams):
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.batch
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    os.environ["MOFO_WGRAD_STREAM"] = "main"                 # (no weight gradients are launched anyway; keeps the lists single-stream)
    mods = [build(B, dev, 0), build(B // 2, dev, 1), build(B // 2, dev, 2)]
    fwd, bwd = [], []
    for k, (model, rt, w) in enumerate(mods):
        s = sB if k == 2 else sA

        def f(rt=rt, w=w):
            rt.cached(w, "exp_enc_fwd", lambda: rt.encoder_forward(w))

        def b(rt=rt, w=w):
            rt._accumulate = False
            rt.cached(w, "exp_enc_bwd", lambda: rt.encoder_backward(w, w.d_encout))
        fwd.append(capture(f, s))
        bwd.append(capture(b, s))
    for rnd in range(3):
        for name, gs in (("forward", fwd), ("backward chain", bwd)):
            t_full = timed([gs[0]], [sA], a.reps)
            t_half = timed([gs[1]], [sA], a.reps)
            t_seq = timed([gs[1], gs[2]], [sA, sA], a.reps)
            t_two = timed([gs[1], gs[2]], [sA, sB], a.reps)
            print(f"round {rnd} encoder {name:14s}: B={B} {t_full:.3f} ms | B={B // 2} alone {t_half:.3f} | two B={B // 2} one stream {t_seq:.3f} | "
                  f"two B={B // 2} two streams {t_two:.3f} ms", flush=True)


if __name__ == "__main__":
    main()

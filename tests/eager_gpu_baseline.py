#!/usr/bin/env python3
"""The reference's step as plain PyTorch on the GPU of this box -- a BASELINE measurement, kept under tests/ because it runs the oracle
(test infrastructure; bench.py uses it for the CPU baseline only).  Not collected by pytest.
usage (GPU box): python tests/eager_gpu_baseline.py [batch] [steps]      -> one JSON line; run `python bench.py` in the same call for the
same-box ratio (profiles/r05_eager_gpu_baseline.json: 82.1 ms per step = 390 clips/s against 11.7 ms = 2 730 clips/s, 7.0x)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def eager_gpu_baseline(B: int, steps: int = 10):
    """What the reference's step costs under PyTorch on THIS GPU -- the
    oracle's functional restatement routed through the torch library ops the reference's modules call (O.LIBRARY_OPS: F.linear,
    F.layer_norm, F.gelu, softmax, boolean-mask gathers; attention as q @ k^T -> softmax -> @ v, modeling_finetune.py:85-95) under a
    bf16 autocast (the reference: fp16 autocast + GradScaler, engine_for_pretraining.py:65), torch.optim.AdamW, the per-tensor norm loop
    of utils.py:376-388 and the two syncs of engine_for_pretraining.py:69,179.  ViT-B, B clips, same synthetic shapes as the timed step."""
    from oracle import pretrain_oracle as O
    dev = torch.device("cuda:0")
    cfg = O.VIT_B
    O.LIBRARY_OPS = True
    try:
        P = {k: v.to(dev).requires_grad_(True) for k, v in O.keyed_params(cfg, "xavier").items()}
        dec = [p for k, p in P.items() if not O.is_no_decay(k, p.shape)]
        nod = [p for k, p in P.items() if O.is_no_decay(k, p.shape)]
        opt = torch.optim.AdamW([{"params": dec, "weight_decay": 0.05}, {"params": nod, "weight_decay": 0.0}], lr=1.5e-4, betas=(0.9, 0.95), eps=1e-8)
        x = torch.randn(B, 3, cfg.num_frames, cfg.img_size, cfg.img_size, device=dev)
        np.random.seed(0)
        mask = torch.from_numpy(np.stack([O.tube_mask(cfg.grid, 0.9) for _ in range(B)])).bool().to(dev)

        def step():
            with torch.no_grad():
                labels = O.build_targets(x, mask, cfg, True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = O.model_forward(x, mask, P, cfg)
                loss = O.mse_loss(out, labels)
            v = loss.item()
            opt.zero_grad()
            loss.backward()
            torch.norm(torch.stack([torch.norm(p.grad.detach(), 2.0) for p in P.values()]), 2.0)
            opt.step()
            torch.cuda.synchronize()
            return v
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(steps):
            last = step()
        torch.cuda.synchronize()
        dt = time.time() - t0
    finally:
        O.LIBRARY_OPS = False
    return {"value": round(B * steps / dt, 2), "unit": "clips/s", "ms_per_step": round(1e3 * dt / steps, 3), "kind": "port", "final_loss": round(last, 5),
            "sample": f"oracle restatement through torch library ops, bf16 autocast, torch.optim.AdamW, ViT-B batch {B}, {steps} timed steps on cuda:0"}




if __name__ == "__main__":
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    print(json.dumps({"eager_gpu_baseline": eager_gpu_baseline(b, n)}), flush=True)

"""Model-level record for the dK/dV accumulator folds (MOFO_ATTN_DKV_FOLD = 0 | 1 | 2; 2 holds K as bf16(-c K), one more rounding):
ViT-B, the two clips of the engine fixture, every one of the 218 gradient tensors element-wise against the CPU oracle, per fold
mode; prints the worst tensors, the decoder qkv weights (where dK / dV land first) and the gradient norm.

  python tools/fold_drift.py > profiles/r04_dkv_fold_drift.txt          (GPU only; ~1 min, most of it the oracle's CPU step)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from test_model_gpu import _build, _vitb_inputs
    from oracle import pretrain_oracle as O
    dev = torch.device("cuda:0")
    model, P = _build(O.VIT_B, "xavier", dev)
    x, mask = _vitb_inputs(dev, "tube")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref_loss, ref_gn, og = O.train_step(x, mask, P, O.VIT_B)
    names = list(og)
    store = model.runtime().store
    xd, md = x.to(dev), mask.to(dev)
    print(f"ViT-B B=2 (engine fixture inputs): oracle loss {ref_loss:.6f} grad norm {ref_gn:.6f}")
    var = sys.argv[1] if len(sys.argv) > 1 else "MOFO_ATTN_DKV_FOLD"          # e.g. MOFO_ATTN_FOLDQ 0,1 (the dQ pass's exp2 fold)
    vals = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2]
    print(f"switch {var}, values {vals}")
    for fold in vals:
        os.environ[var] = str(fold)
        loss = model.forward_loss(xd, md)
        store.zero_grads()
        loss.backward()
        gn = float(model.runtime().grad_norm())
        grads = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters()}
        errs = []
        for n in names:
            a, b = grads[n].flatten(), og[n].double().flatten()
            errs.append((float((a - b).norm()) / max(float(b.norm()), 1e-3 * ref_gn), n))
        errs.sort(reverse=True)
        dq = [(e, n) for e, n in errs if "decoder" in n and "qkv.weight" in n]
        print(f"fold {fold}: loss {float(loss):.6f} grad norm {gn:.6f} (rel {abs(gn - ref_gn) / ref_gn:.2e}) | worst tensors: "
              + ", ".join(f"{n} {e:.4f}" for e, n in errs[:4]) + " | decoder qkv weights: " + ", ".join(f"{n.split('.')[2]} {e:.4f}" for e, n in sorted(dq, key=lambda t: t[1]))
              + f" | mean over 218 {np.mean([e for e, _ in errs]):.5f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""What the e4m3 copy costs a LayerNorm forward (mofo_layernorm_fwd_q against mofo_layernorm_fwd), at the four model shapes.  GPU box only.
Round 5, one MI355X: +2.0 / +5.4 us at ViT-L's encoder / decoder rows (13.8 -> 15.8, 38.3 -> 43.7), +1.8 / +5.1 at ViT-B's: the extra byte per element at the
kernel's own bandwidth, nothing more."""
import torch, sys, os
sys.path.insert(0, os.getcwd())
from mofo_amd import ops
dev = torch.device("cuda:0")
F8 = torch.float8_e4m3fn
def t(f, it=200):
    for _ in range(50): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for M, D, dt in ((10240, 1024, torch.float32), (100352, 512, torch.bfloat16), (5120, 768, torch.float32), (50176, 384, torch.bfloat16)):
    x = torch.randn(M, D, device=dev).to(dt)
    w = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    y = torch.empty(M, D, dtype=torch.bfloat16, device=dev); y8 = torch.empty(M, D, dtype=F8, device=dev)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    sc = torch.tensor([60.0], device=dev); am = torch.zeros(ops.FP8_AMAX_STRIPES, device=dev)
    a = t(lambda: ops.layernorm_fwd(x, w, b, 1e-6, y, mean, rstd))
    q = t(lambda: ops.layernorm_fwd_q(x, w, b, 1e-6, y, mean, rstd, y8, sc, am))
    by = M * D * (x.element_size() + 2)
    print(f"M={M} D={D} {dt}: plain {a:.1f} us ({by / a / 1e6:.2f} TB/s)  with e4m3 copy {q:.1f} us ({(by + M * D) / q / 1e6:.2f} TB/s)  +{q - a:.1f} us")

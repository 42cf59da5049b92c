import os, sys, torch
sys.path.insert(0, os.getcwd())
from mofo_amd import ops
dev = torch.device("cuda:0"); BF16 = torch.bfloat16
def t(f, warm=60, it=20):
    for _ in range(warm): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
n, H = 1568, 6
print("B  blocks(fwd,NW=4)  rounds@1024  fwd us  us/(B*H)   dq us  us/(B*H)   dkv us us/(B*H)")
for B in (20, 21, 22, 26, 27, 28, 32, 36, 39, 40, 42):
    qkv = (torch.randn(B * n, 3 * H * 64, device=dev) * 0.5).to(BF16); out = torch.empty(B * n, H * 64, dtype=BF16, device=dev)
    lse = torch.empty(B * H * n, device=dev); dout = (torch.randn(B * n, H * 64, device=dev) * 0.5).to(BF16); dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
    f = t(lambda: ops.attention_fwd(qkv, B, n, H, 0.125, out, lse))
    ops.attention_delta(out, dout, B, n, H, delta)
    q = t(lambda: ops.attention_bwd_dq(qkv, dout, lse, delta, B, n, H, 0.125, dqkv))
    k = t(lambda: ops.attention_bwd_dkv(qkv, dout, lse, delta, B, n, H, 0.125, dqkv))
    nb = B * H * 13
    print(f"{B:2d} {nb:6d} {nb/1024:6.2f} | {f:7.1f} {f/(B*H):6.3f} | {q:7.1f} {q/(B*H):6.3f} (rounds@768 {nb/768:.2f}) | {k:7.1f} {k/(B*H):6.3f} (rounds@512 {nb/512:.2f})")

import sys, torch
sys.path.insert(0, "/root/repo")
from mofo_amd import ops
dev = torch.device("cuda:0")
B, N, H = 32, 1568, 6
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
qkv = r(B * N, 3 * H * 64); out = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H * N, device=dev)
dout = r(B * N, H * 64); dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
ops.attention_fwd(qkv, B, N, H, 0.125, out, lse)
ops.attention_delta_zero_dq(out, dout, B, N, H, delta, dqkv)
def t(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(2):
    print("onepass %.1f us | dkv %.1f us | dq %.1f us | delta+zero %.1f us" % (
        t(lambda: ops.attention_bwd_onepass(qkv, dout, lse, delta, B, N, H, 0.125, dqkv)),
        t(lambda: ops.attention_bwd_dkv(qkv, dout, lse, delta, B, N, H, 0.125, dqkv)),
        t(lambda: ops.attention_bwd_dq(qkv, dout, lse, delta, B, N, H, 0.125, dqkv)),
        t(lambda: ops.attention_delta_zero_dq(out, dout, B, N, H, delta, dqkv))))

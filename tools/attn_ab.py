#!/usr/bin/env python3
"""A/B of attention-kernel builds and experiment switches in ONE process, interleaved rounds (guide rule 24): timings from
different boxes differ by 5-8 % (clocks), so variants are only compared inside one process on one device.
usage: attn_ab.py [B N H] [--libs a.so,b.so] [--env MOFO_ATTN_RESCALE_THR --vals 6,0]
Every (library, value of the environment switch) pair is a variant; the C side reads such switches at every launch.  Calls the C-ABI through
ctypes directly so that two builds of libmofo_hip.so can be loaded side by side (tools/_ab/ is git-ignored)."""
import argparse, ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("dims", nargs="*", type=int, default=[32, 1568, 6])
ap.add_argument("--libs", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mofo_amd", "libmofo_hip.so"))
ap.add_argument("--env", default="MOFO_ATTN_RESCALE_THR")
ap.add_argument("--vals", default="6")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--warm", type=int, default=80, help="launches of the same variant before every timed block (steady state: short bursts after another "
                                                     "kernel are timed in a transient -- one-block-per-CU kernels read 20-35 %% low)")
a = ap.parse_args()
B, n, H = a.dims
dev = torch.device("cuda:0")
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
qkv = r(B * n, 3 * H * 64); out = torch.empty(B * n, H * 64, dtype=torch.bfloat16, device=dev); lse = torch.empty(B * H * n, device=dev)
dout = r(B * n, H * 64); dqkv = torch.zeros_like(qkv); delta = torch.empty_like(lse)
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
sc = C.c_float(0.125)


def chk(rc, lib):
    if rc:
        lib.mofo_last_error_string.restype = C.c_char_p
        raise RuntimeError(lib.mofo_last_error_string().decode())


variants = []
for path in a.libs.split(","):
    lib = C.CDLL(os.path.abspath(path))
    for fl in a.vals.split(","):
        variants.append((f"{os.path.basename(path)}:{fl}", lib, fl))


def kern(lib):
    return {"fwd": lambda: chk(lib.mofo_attention_fwd(P(qkv), qkv.stride(0), B, n, H, sc, P(out), out.stride(0), P(lse), st()), lib),
            "dq": lambda: chk(lib.mofo_attention_bwd_dq(P(qkv), qkv.stride(0), P(dout), dout.stride(0), P(lse), P(delta), B, n, H, sc, P(dqkv), dqkv.stride(0), st()), lib),
            "dkv": lambda: chk(lib.mofo_attention_bwd_dkv(P(qkv), qkv.stride(0), P(dout), dout.stride(0), P(lse), P(delta), B, n, H, sc, P(dqkv), dqkv.stride(0), st()), lib)}


ref = None
for name, lib, fl in variants:           # results must not depend on the variant (beyond what it documents)
    os.environ[a.env] = fl
    k = kern(lib)
    k["fwd"]()
    chk(lib.mofo_attention_delta(P(out), out.stride(0), P(dout), dout.stride(0), B, n, H, P(delta), st()), lib)
    k["dq"](); k["dkv"]()
    torch.cuda.synchronize()
    cur = (out.float().clone(), dqkv.float().clone())
    if ref is None:
        ref = cur
    else:
        print(f"{name}: max |out - out0| = {(cur[0] - ref[0]).abs().max().item():.3e}, max |dqkv - dqkv0| = {(cur[1] - ref[1]).abs().max().item():.3e}")
res = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rnd in range(a.rounds + 1):
    for kn in ("fwd", "dq", "dkv"):
        for name, lib, fl in variants:
            os.environ[a.env] = fl
            f = kern(lib)[kn]
            for _ in range(max(1, a.warm)): f()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.iters): f()
            e1.record(); torch.cuda.synchronize()
            if rnd: res.setdefault((kn, name), []).append(e0.elapsed_time(e1) / a.iters * 1e3)
for kn in ("fwd", "dq", "dkv"):
    print(f"{kn:4s}", "  ".join(f"[{name}] {statistics.median(res[(kn, name)]):6.1f} (min {min(res[(kn, name)]):6.1f})" for name, _, _ in variants))

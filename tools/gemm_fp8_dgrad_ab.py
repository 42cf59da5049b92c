#!/usr/bin/env python3
"""The MLP dgrads of a transformer block as e5m2 x e4m3 GEMMs (MOFO_GEMM_NT_FP8 with an e5m2 A operand against the TRANSPOSED e4m3 weight
shadow) against the bf16 NN route, per GEMM, in ONE process on one device (GPU box only): fc2's dgrad with the dGELU epilogue (which also
writes the e5m2 copy of its result) and fc1's dgrad; plus what the path adds around them -- the e5m2 copy of the incoming gradient
(mofo_fp8_quantize_site) and the per-step transposed weight shadow (mofo_fp8_transpose_weights).  Results against fp32 torch on the same
(de-quantised) operands.  usage: gemm_fp8_dgrad_ab.py [vitl|vitb] [rounds]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mofo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF16, F32, F8, F8E5 = torch.bfloat16, torch.float32, torch.float8_e4m3fn, torch.float8_e5m2
E = ops


def timed(run, iters, warm):
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def block(tag, M, D, hid, rounds):
    g = torch.Generator(device=dev).manual_seed(3)
    dx = (torch.randn(M, D, generator=g, device=dev) * 3e-6).to(BF16)             # gradient wrt the block output
    W2 = (torch.randn(D, hid, generator=g, device=dev) * 0.02).to(BF16)           # fc2.weight [out = D, in = hid]
    W1 = (torch.randn(hid, D, generator=g, device=dev) * 0.02).to(BF16)           # fc1.weight [out = hid, in = D]
    h1 = torch.randn(M, hid, generator=g, device=dev).to(BF16)
    dh1 = torch.empty(M, hid, dtype=BF16, device=dev)
    dh1f = torch.empty(M, hid, dtype=BF16, device=dev)
    dh1_8 = torch.empty(M, hid, dtype=F8E5, device=dev)
    dxln = torch.empty(M, D, dtype=BF16, device=dev)
    dxlnf = torch.empty(M, D, dtype=BF16, device=dev)
    # the flat "shadow" of the two matrices and its transposed e4m3 twin
    flat = torch.cat([W2.reshape(-1), W1.reshape(-1)]).contiguous()
    flatT8 = torch.empty(flat.numel(), dtype=F8, device=dev)
    am = torch.stack([W2.float().abs().max(), W1.float().abs().max()])
    w_si = (am / 448.0).to(F32).contiguous()
    t2, t1 = (D // 64) * (hid // 64), (hid // 64) * (D // 64)
    table = torch.tensor([[0, D, hid, 0, 0], [W2.numel(), hid, D, 1, t2]], dtype=torch.int32, device=dev)
    tr = lambda: ops.fp8_transpose_weights(flat, flatT8, table, t2 + t1, w_si)
    tr()
    W2T8 = flatT8[:W2.numel()].view(hid, D)
    W1T8 = flatT8[W2.numel():].view(D, hid)
    rs = 1.0 / w_si          # the kernel multiplies by the f32 reciprocal
    assert torch.equal(W2T8.view(torch.uint8), (W2.float().t() * rs[0]).clamp(-448, 448).to(F8).contiguous().view(torch.uint8))
    assert torch.equal(W1T8.view(torch.uint8), (W1.float().t() * rs[1]).clamp(-448, 448).to(F8).contiguous().view(torch.uint8))
    # gradient sites: e5m2, delayed scales as a previous step would have left them (maximum x 2 of margin)
    s_dx = (57344.0 / (2.0 * dx.float().abs().max())).reshape(1).to(F32)
    dx8 = torch.empty(M, D, dtype=F8E5, device=dev)
    st_dx, st_dh = (torch.zeros(ops.FP8_AMAX_STRIPES, dtype=F32, device=dev) for _ in range(2))
    qx = lambda: ops.fp8_quantize_site(dx, s_dx, dx8, st_dx)
    qx()
    assert float(st_dx.max()) == float(dx.float().abs().max())
    ops.gemm(E.GEMM_NN, E.EPI_DGELU_BF16, dx, W2, dh1, aux=h1)
    s_dh = (57344.0 / (2.0 * dh1.float().abs().max())).reshape(1).to(F32)
    inv = lambda t: (1.0 / t).contiguous()
    f_dgelu16 = lambda: ops.gemm(E.GEMM_NN, E.EPI_DGELU_BF16, dx, W2, dh1, aux=h1)
    f_dfc116 = lambda: ops.gemm(E.GEMM_NN, E.EPI_BF16, dh1, W1, dxln)
    a_si_dx, a_si_dh = inv(s_dx), inv(s_dh)
    f_dgelu8 = lambda: ops.gemm(E.GEMM_NT_FP8, E.EPI_DGELU_BF16, dx8, W2T8, dh1f, aux=h1, a_scale_inv=a_si_dx, b_scale_inv=w_si[0:1], C8=dh1_8, q_scale=s_dh, q_amax=st_dh)
    f_dfc18 = lambda: ops.gemm(E.GEMM_NT_FP8, E.EPI_BF16, dh1_8, W1T8, dxlnf, a_scale_inv=a_si_dh, b_scale_inv=w_si[1:2])
    f_dgelu16(), f_dgelu8(), f_dfc116(), f_dfc18()
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    print(f"## {tag}: M = {M}, D = {D}, hidden = {hid}   (errors of the fp8 path against the bf16 path: dh1 {rel(dh1f, dh1):.2e}, its e5m2 copy "
          f"{rel(dh1_8.float() / s_dh, dh1f):.2e}, dxln {rel(dxlnf, dxln):.2e}; amax site {float(st_dh.max()):.3e} / {float(dh1f.float().abs().max()):.3e})")
    rows = [("fc2 dgrad + dGELU   bf16 NN", f_dgelu16, 2.0 * M * D * hid), ("fc2 dgrad + dGELU   e5m2 x e4m3 (+ e5m2 copy out)", f_dgelu8, 2.0 * M * D * hid),
            ("fc1 dgrad           bf16 NN", f_dfc116, 2.0 * M * D * hid), ("fc1 dgrad           e5m2 x e4m3", f_dfc18, 2.0 * M * D * hid),
            ("e5m2 copy of the incoming gradient [M, D]", qx, 0.0), ("transposed e4m3 shadow of fc1 + fc2 (per STEP, not per GEMM)", tr, 0.0)]
    res = {}
    for label, f, fl in rows:
        iters = 50
        ts = [timed(f, iters, 60) for _ in range(rounds)]
        res[label] = statistics.median(ts)
        print(f"  {label:<62s} {res[label]:8.1f} us" + (f"  {fl / res[label] / 1e6:6.0f} TF/s" if fl else ""), flush=True)
    b16 = res[rows[0][0]] + res[rows[2][0]]
    f8 = res[rows[1][0]] + res[rows[3][0]] + res[rows[4][0]]
    print(f"  per block: bf16 {b16:.1f} us, fp8 incl. the gradient copy {f8:.1f} us  ({b16 / f8:.2f} x, {b16 - f8:+.1f} us)")


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vitl"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    if which == "vitl":
        block("ViT-L encoder block", 10240, 1024, 4096, rounds)
        block("ViT-L decoder block", 100352, 512, 2048, rounds)
    else:
        block("ViT-B encoder block", 5120, 768, 3072, rounds)
        block("ViT-B decoder block", 50176, 384, 1536, rounds)


if __name__ == "__main__":
    main()

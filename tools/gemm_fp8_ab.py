#!/usr/bin/env python3
"""The four forward Linears of a transformer block on e4m3 operands (MOFO_GEMM_NT_FP8, the persistent kernel of round 5) against
the default bf16 route, per GEMM, in ONE process on one device (GPU box only).  The A/B the round-4 review asked for before the fp8
row of BASELINE configs[4] is called closed.

For every shape: the bf16 NT GEMM through the default routing, the e4m3 GEMM (operands quantised per tensor beforehand: in the
model the LayerNorm / attention / GELU epilogues and AdamW write them).  (The "old e4m3" column of profiles/r05_gemm_fp8_ab.txt was the
one-tile-per-block kernel of rounds 2-4, deleted after that record.)  Results are checked against an fp32 torch product of the SAME (de-quantised) operands.
Interleaved timing rounds, 20 ms of the same variant first (steady state), median and min.
usage: gemm_fp8_ab.py [vitl|vitb|all] [rounds]      (MOFO_GEMM_MI8=1 in the environment forces 256-row tiles for both arms)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mofo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
BF16, F32, F8 = torch.bfloat16, torch.float32, torch.float8_e4m3fn
E = ops


def quant(x):
    am = x.float().abs().max()
    s = 448.0 / am
    q = (x.float() * s).clamp(-448, 448).to(F8)
    return q, (1.0 / s).reshape(1).to(F32), s.reshape(1).to(F32)


def make(epi, M, N, K):
    A = (torch.randn(M, K, device=dev) * 0.5).to(BF16)
    B = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
    A8, a_si, _ = quant(A)
    B8, b_si, _ = quant(B)
    bias = torch.randn(N, device=dev) * 0.5
    ref16 = lambda: A.float() @ B.float().t() + bias
    ref8 = lambda: (A8.float() * a_si) @ (B8.float() * b_si).t() + bias
    kw16, kw8, post = {}, {}, lambda z: z
    if epi == E.EPI_BF16:
        C16, C8o = (torch.empty(M, N, dtype=BF16, device=dev) for _ in range(2))
    elif epi == E.EPI_BIAS_GELU:
        C16, C8o = (torch.empty(M, N, dtype=BF16, device=dev) for _ in range(2))
        G16, G8o = (torch.empty(M, N, dtype=BF16, device=dev) for _ in range(2))
        Gq = torch.empty(M, N, dtype=F8, device=dev)
        qs = torch.tensor([448.0 / 6.0], dtype=F32, device=dev)       # a delayed scale: what a previous step might have left
        qam = torch.zeros(ops.FP8_AMAX_STRIPES, dtype=F32, device=dev)
        kw16 = dict(C2=G16)
        kw8 = dict(C2=G8o, C8=Gq, q_scale=qs, q_amax=qam)
    elif epi == E.EPI_RESID_F32:
        C16, C8o = (torch.empty(M, N, dtype=F32, device=dev) for _ in range(2))
        R = torch.randn(M, N, device=dev)
        kw16 = kw8 = dict(resid=R)
        post = lambda z: z + R
    else:
        C16, C8o = (torch.empty(M, N, dtype=BF16, device=dev) for _ in range(2))
        R = (torch.randn(M, N, device=dev)).to(BF16)
        kw16 = kw8 = dict(aux=R)
        post = lambda z: z + R.float()

    def run16():
        ops.gemm(E.GEMM_NT, epi, A, B, C16, bias=bias, **kw16)

    def run8():
        ops.gemm(E.GEMM_NT_FP8, epi, A8, B8, C8o, bias=bias, a_scale_inv=a_si, b_scale_inv=b_si, **kw8)

    def check():
        run16(), run8()
        torch.cuda.synchronize()
        w16, w8 = post(ref16()), post(ref8())
        e16 = float((C16.float() - w16).norm() / w16.norm())
        e8 = float((C8o.float() - w8).norm() / w8.norm())
        bad = int(((C8o.float() - w8).abs() > 0.03 * w8.abs().max()).sum())
        e8q = float((C8o.float() - w16).norm() / w16.norm())        # what the quantisation costs against the bf16 operands
        msg = ""
        if epi == E.EPI_BIAS_GELU:
            g = torch.nn.functional.gelu(ref8())
            eg = float((G8o.float() - g).norm() / g.norm())
            gq = float((Gq.float() / qs - g.clamp(-6.0, 6.0)).norm() / g.norm())
            am = float(qam.max())
            msg = f" gelu {eg:.1e} e4m3 copy {gq:.1e} amax {am:.3f}/{float(g.abs().max()):.3f}"
            if not (eg < 5e-3 and gq < 4e-2 and abs(am - float(g.abs().max())) <= 0.02 * am):
                bad += 1
        return e16, e8, e8q, bad, msg
    return run16, run8, check


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


SETS = {
    "vitl": [("L enc qkv  bf16", E.EPI_BF16, 10240, 3072, 1024), ("L enc proj rf32", E.EPI_RESID_F32, 10240, 1024, 1024),
             ("L enc fc1  gelu", E.EPI_BIAS_GELU, 10240, 4096, 1024), ("L enc fc2  rf32", E.EPI_RESID_F32, 10240, 1024, 4096),
             ("L dec qkv  bf16", E.EPI_BF16, 100352, 1536, 512), ("L dec proj rbf16", E.EPI_RESID_BF16, 100352, 512, 512),
             ("L dec fc1  gelu", E.EPI_BIAS_GELU, 100352, 2048, 512), ("L dec fc2  rbf16", E.EPI_RESID_BF16, 100352, 512, 2048)],
    "vitb": [("B enc qkv  bf16", E.EPI_BF16, 5120, 2304, 768), ("B enc proj rf32", E.EPI_RESID_F32, 5120, 768, 768),
             ("B enc fc1  gelu", E.EPI_BIAS_GELU, 5120, 3072, 768), ("B enc fc2  rf32", E.EPI_RESID_F32, 5120, 768, 3072),
             ("B dec qkv  bf16", E.EPI_BF16, 50176, 1152, 384), ("B dec proj rbf16", E.EPI_RESID_BF16, 50176, 384, 384),
             ("B dec fc1  gelu", E.EPI_BIAS_GELU, 50176, 1536, 384), ("B dec fc2  rbf16", E.EPI_RESID_BF16, 50176, 384, 1536)],
    "small": [("ragged bf16", E.EPI_BF16, 1000, 392, 256), ("ragged gelu", E.EPI_BIAS_GELU, 777, 520, 384),
              ("ragged rf32", E.EPI_RESID_F32, 300, 136, 128), ("ragged rbf16", E.EPI_RESID_BF16, 2100, 384, 640)],
}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "vitl"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    names = list(SETS) if which == "all" else which.split(",")
    print(f"{'shape':<18}{'M':>7}{'N':>6}{'K':>6} | err bf16   e4m3(same ops)  e4m3 vs bf16 ops | bf16 us (min)    e4m3 us (min)    | TF/s bf16  e4m3  ratio")
    for nm in names:
        for label, epi, M, N, K in SETS[nm]:
            torch.manual_seed(0)
            run16, run8, check = make(epi, M, N, K)
            e16, e8, e8q, bad, msg = check()
            ok = "OK" if (e16 < 5e-3 and e8 < 5e-3 and bad == 0) else "FAIL"
            flops = 2.0 * M * N * K
            iters = max(3, min(200, int(4e-3 / (flops / 600e12))))
            warm = max(3, int(20e-3 / (flops / 600e12)))
            t16, t8 = [], []
            for _ in range(rounds):
                t16.append(timed(run16, iters, warm))
                t8.append(timed(run8, iters, warm))
            m16, m8 = statistics.median(t16), statistics.median(t8)
            print(f"{label:<18}{M:>7}{N:>6}{K:>6} | {e16:.2e}  {e8:.2e}  {e8q:.2e} {ok:>4} | {m16:8.1f} ({min(t16):7.1f}) {m8:8.1f} ({min(t8):7.1f}) |"
                  f" {flops / m16 / 1e6:7.0f} {flops / m8 / 1e6:7.0f}  {m16 / m8:5.2f}x{msg}", flush=True)


if __name__ == "__main__":
    main()

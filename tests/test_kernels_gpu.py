"""Per-kernel parity on a real MI355X: every C-ABI entry against a plain fp32 torch restatement of the same op
(and the oracle where the op is a reference function).  Inputs are asymmetric random data so that a transposed
fragment map or a swapped operand cannot hide."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mofo_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rand(shape, dev, seed, scale=1.0, dtype=BF16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ------------------------------------------------------------------------------------------------ GEMM
# (200, 136, 1600): ragged M and N with an ODD number of 64-wide k-stages through the in-block split-K kernel (K >= 1536, few tiles)
GEMM_SHAPES = [(128, 128, 64), (256, 384, 128), (320, 2304, 768), (77, 192, 128), (16, 64, 64), (640, 768, 3072), (1000, 1536, 384),
               (200, 136, 1600), (5000, 3064, 128)]   # the last one takes the 256-row-tile form (960 vs 480 tiles), ragged on both axes


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_nt_epilogues(dev, M, N, K):
    from mofo_amd import ops
    A = _rand((M, K), dev, 1)
    B = _rand((N, K), dev, 2, 0.05)
    bias = _rand((N,), dev, 3, 1.0, F32)
    ref = A.float() @ B.float().t()
    # plain bf16 store, no bias
    Cb = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A, B, Cb)
    assert _rel(Cb, ref) < 6e-3
    # exact-value check on a few entries (catches permuted rows/cols that a norm might not)
    idx = torch.randint(0, M * N, (64,), generator=torch.Generator().manual_seed(5))
    assert torch.allclose(Cb.flatten()[idx].float(), ref.flatten()[idx], rtol=2e-2, atol=2e-2)
    # bias
    ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A, B, Cb, bias=bias)
    assert _rel(Cb, ref + bias) < 6e-3
    # bias + gelu (two outputs)
    C2 = torch.empty_like(Cb)
    ops.gemm(ops.GEMM_NT, ops.EPI_BIAS_GELU, A, B, Cb, C2=C2, bias=bias)
    h = ref + bias
    assert _rel(Cb, h) < 6e-3
    assert _rel(C2, torch.nn.functional.gelu(h)) < 8e-3
    # residual, fp32 out
    R = _rand((M, N), dev, 4, 1.0, F32)
    Cf = torch.empty(M, N, dtype=F32, device=dev)
    ops.gemm(ops.GEMM_NT, ops.EPI_RESID_F32, A, B, Cf, bias=bias, resid=R)
    assert _rel(Cf, h + R) < 1e-5 + 1e-6 * math.sqrt(K)
    # residual add on a bf16 residual stream (the decoder): residual in `aux`, bf16 out
    Rb = R.to(BF16)
    ops.gemm(ops.GEMM_NT, ops.EPI_RESID_BF16, A, B, Cb, bias=bias, aux=Rb)
    assert _rel(Cb, h + Rb.float()) < 6e-3
    # f32 plain, split-K with atomics, accumulate
    ops.gemm(ops.GEMM_NT, ops.EPI_F32, A, B, Cf)
    assert _rel(Cf, ref) < 1e-5
    if K >= 128:
        Cf.zero_()
        ops.gemm(ops.GEMM_NT, ops.EPI_F32, A, B, Cf, splits=2)
        assert _rel(Cf, ref) < 1e-5
        ops.gemm(ops.GEMM_NT, ops.EPI_F32, A, B, Cf, accumulate=True)
        assert _rel(Cf, 2 * ref) < 1e-5


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (320, 2304, 768), (77, 192, 256), (640, 3072, 1024), (1000, 1536, 384), (5000, 200, 128)])
def test_gemm_nt_fp8(dev, M, N, K):
    """NT on OCP e4m3 operands (block-scaled MFMA with unit block scales, f32 accumulate, per-tensor de-quantisation factors
    read on the device) against fp32 torch on the SAME quantised operands: only the accumulation order differs."""
    from mofo_amd import ops
    F8 = torch.float8_e4m3fn
    a = torch.randn(M, K, generator=torch.Generator().manual_seed(1)).to(dev) * 1.3
    w = torch.randn(N, K, generator=torch.Generator().manual_seed(2)).to(dev) * 0.04
    sa, sw = 448.0 / float(a.abs().max()), 448.0 / float(w.abs().max())
    A8 = (a * sa).clamp(-448, 448).to(F8)
    W8 = (w * sw).clamp(-448, 448).to(F8)
    ai = torch.tensor([1.0 / sa], dtype=F32, device=dev)
    wi = torch.tensor([1.0 / sw], dtype=F32, device=dev)
    bias = _rand((N,), dev, 3, 1.0, F32)
    ref = (A8.float() @ W8.float().t()) / (sa * sw)
    C = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm(ops.GEMM_NT_FP8, ops.EPI_BF16, A8, W8, C, a_scale_inv=ai, b_scale_inv=wi)
    assert _rel(C, ref) < 5e-3
    idx = torch.randint(0, M * N, (64,), generator=torch.Generator().manual_seed(5))
    assert torch.allclose(C.flatten()[idx].float(), ref.flatten()[idx], rtol=2e-2, atol=2e-2)
    C2 = torch.empty_like(C)
    ops.gemm(ops.GEMM_NT_FP8, ops.EPI_BIAS_GELU, A8, W8, C, C2=C2, bias=bias, a_scale_inv=ai, b_scale_inv=wi)
    assert _rel(C, ref + bias) < 5e-3 and _rel(C2, torch.nn.functional.gelu(ref + bias)) < 8e-3
    # the e4m3 copy of the activation (fc2's A operand) with a DELAYED scale that saturates part of it, and the launch's max|gelu|
    g = torch.nn.functional.gelu(ref + bias)
    qs = torch.tensor([448.0 / (0.5 * float(g.abs().max()))], dtype=F32, device=dev)
    G8 = torch.zeros(M, N, dtype=F8, device=dev)
    am = torch.zeros(ops.FP8_AMAX_STRIPES, dtype=F32, device=dev)
    Cq, C2q = torch.empty_like(C), torch.empty_like(C)
    ops.gemm(ops.GEMM_NT_FP8, ops.EPI_BIAS_GELU, A8, W8, Cq, C2=C2q, bias=bias, a_scale_inv=ai, b_scale_inv=wi, C8=G8, q_scale=qs, q_amax=am)
    assert torch.equal(Cq, C) and torch.equal(C2q, C2)
    lim = 448.0 / float(qs)
    assert _rel(G8.float() / qs, g.clamp(-lim, lim)) < 4e-2 and not torch.isnan(G8.float()).any()
    assert float(am.max()) == pytest.approx(float(g.abs().max()), rel=1e-2)
    # residual epilogues (proj / fc2), dense and through the residual row map of the last decoder block
    R = _rand((M, N), dev, 6, 1.0, F32)
    Cf = torch.empty(M, N, dtype=F32, device=dev)
    ops.gemm(ops.GEMM_NT_FP8, ops.EPI_RESID_F32, A8, W8, Cf, bias=bias, resid=R, a_scale_inv=ai, b_scale_inv=wi)
    assert _rel(Cf, ref + bias + R) < 1e-4
    Rb = R.to(BF16)
    ops.gemm(ops.GEMM_NT_FP8, ops.EPI_RESID_BF16, A8, W8, C, bias=bias, aux=Rb, a_scale_inv=ai, b_scale_inv=wi)
    assert _rel(C, ref + bias + Rb.float()) < 5e-3
    if M % 4 == 0:
        rin, rout, roff = M // 4, M // 4 + 24, 24
        Rbig = _rand((4 * rout, N), dev, 7, 1.0).contiguous()
        want = ref + bias + Rbig.view(4, rout, N)[:, roff:roff + rin].reshape(M, N).float()
        ops.gemm(ops.GEMM_NT_FP8, ops.EPI_RESID_BF16, A8, W8, C, bias=bias, aux=Rbig, rows_in=rin, rows_out=rout, row_off=roff, a_scale_inv=ai, b_scale_inv=wi)
        assert _rel(C, want) < 5e-3
        Rbig32 = Rbig.float().contiguous()
        ops.gemm(ops.GEMM_NT_FP8, ops.EPI_RESID_F32, A8, W8, Cf, bias=bias, resid=Rbig32, rows_in=rin, rows_out=rout, row_off=roff, a_scale_inv=ai, b_scale_inv=wi)
        assert _rel(Cf, want) < 1e-4
    # the quantisation itself: against the unquantised product the error is e4m3's (3 mantissa bits on both operands)
    assert _rel(ref, a @ w.t()) < 6e-2
    with pytest.raises(RuntimeError, match="multiple of 128"):
        ops.gemm(ops.GEMM_NT_FP8, ops.EPI_BF16, A8[:, :64].contiguous(), W8[:, :64].contiguous(), C, a_scale_inv=ai, b_scale_inv=wi)


def test_fp8_quantisation_kernels(dev):
    """e4m3 quantisation: segmented per-tensor quantisation of a flat bf16 buffer (the weight shadow), the plain form, the
    LayerNorm that also emits e4m3, and the delayed-scaling update -- against torch's float8_e4m3fn conversion"""
    from mofo_amd import ops
    F8 = torch.float8_e4m3fn
    # three "tensors" in a flat buffer: chunks 0-1 -> segment 0, chunk 2 untouched, chunks 3-5 -> segment 1 (incl. an all-zero tail chunk)
    flat = _rand((6 * 1024,), dev, 1, 0.05)
    flat[1024:2048] *= 7.0
    flat[5 * 1024:] = 0
    seg = torch.tensor([0, 0, -1, 1, 1, 1], dtype=torch.int16, device=dev)
    out = torch.zeros(6 * 1024, dtype=F8, device=dev)
    out.view(torch.uint8)[2048:3072] = 0x5A
    amax = torch.empty(2, dtype=F32, device=dev)
    sinv = torch.zeros(2, dtype=F32, device=dev)
    ops.fp8_quantize_segments(flat, seg, 2, amax, out, sinv)
    for s_, (lo, hi) in enumerate(((0, 2048), (3072, 6144))):
        x = flat[lo:hi].float()
        a = float(x.abs().max())
        assert float(sinv[s_]) == pytest.approx(a / 448.0, rel=1e-6)
        want = (x * (448.0 / a)).clamp(-448, 448).to(F8)
        assert torch.equal(out[lo:hi].view(torch.uint8), want.view(torch.uint8))
    assert torch.all(out.view(torch.uint8)[2048:3072] == 0x5A)          # a chunk outside every segment is not written
    # plain tensor with a given scale, amax reported
    x = _rand((300, 128), dev, 2, 3.0)
    sc = torch.tensor([20.0], dtype=F32, device=dev)
    q = torch.empty(300, 128, dtype=F8, device=dev)
    am = torch.zeros(1, dtype=F32, device=dev)
    ops.fp8_quantize_bf16(x, sc, q, am)
    assert torch.equal(q.view(torch.uint8), (x.float() * 20.0).clamp(-448, 448).to(F8).view(torch.uint8))      # saturates, never NaN
    assert float(am) == float(x.float().abs().max())
    # LayerNorm + e4m3 copy of the (bf16-rounded) output
    M, D = 512, 384
    xs = _rand((M, D), dev, 3, 2.0, F32)
    w = _rand((D,), dev, 4, 0.3, F32) + 1.0
    b = _rand((D,), dev, 5, 0.3, F32)
    y = torch.empty(M, D, dtype=BF16, device=dev)
    y2 = torch.empty_like(y)
    mean, rstd = torch.empty(M, dtype=F32, device=dev), torch.empty(M, dtype=F32, device=dev)
    y8 = torch.empty(M, D, dtype=F8, device=dev)
    ams = torch.zeros(ops.FP8_AMAX_STRIPES, dtype=F32, device=dev)     # the row waves' atomic maxima are spread over the stripes
    ops.layernorm_fwd_q(xs, w, b, 1e-6, y, mean, rstd, y8, sc, ams)
    ops.layernorm_fwd(xs, w, b, 1e-6, y2, mean, rstd)
    assert torch.equal(y, y2)
    assert torch.equal(y8.view(torch.uint8), (y.float() * 20.0).clamp(-448, 448).to(F8).view(torch.uint8))
    assert float(ams.max()) == float(y.float().abs().max())             # EVERY row contributes (an outlier row cannot be missed)
    # ... and the two-rows-per-wave form of the bf16 residual stream (the decoder), odd row count, narrower than a wave's 512 columns
    for M2, D2 in ((513, 384), (1025, 512), (7, 128)):
        xb = _rand((M2, D2), dev, 6, 2.0)
        w2 = _rand((D2,), dev, 7, 0.3, F32) + 1.0
        b2 = _rand((D2,), dev, 8, 0.3, F32)
        ya, yb = torch.empty(M2, D2, dtype=BF16, device=dev), torch.empty(M2, D2, dtype=BF16, device=dev)
        m2, r2 = torch.empty(M2, dtype=F32, device=dev), torch.empty(M2, dtype=F32, device=dev)
        m3, r3 = torch.empty(M2, dtype=F32, device=dev), torch.empty(M2, dtype=F32, device=dev)
        y8b = torch.zeros(M2, D2, dtype=F8, device=dev)
        ams2 = torch.zeros(ops.FP8_AMAX_STRIPES, dtype=F32, device=dev)
        ops.layernorm_fwd_q(xb, w2, b2, 1e-6, ya, m2, r2, y8b, sc, ams2)
        ops.layernorm_fwd(xb, w2, b2, 1e-6, yb, m3, r3)
        assert torch.equal(ya, yb) and torch.equal(m2, m3) and torch.equal(r2, r3)
        assert torch.equal(y8b.view(torch.uint8), (ya.float() * 20.0).clamp(-448, 448).to(F8).view(torch.uint8))
        assert float(ams2.max()) == float(ya.float().abs().max())
    # delayed scaling update: a site's stripes are folded, then cleared
    amx = torch.zeros(2, ops.FP8_AMAX_STRIPES, dtype=F32, device=dev)
    amx[0, 5], amx[0, 700] = 1.5, 2.0
    scales = torch.tensor([[16.0, 1 / 16.0], [8.0, 0.125]], dtype=F32, device=dev)
    ops.fp8_update_scales(amx, scales, margin=1.5)
    assert scales[0, 0].item() == pytest.approx(448.0 / 3.0) and scales[0, 1].item() == pytest.approx(3.0 / 448.0)
    assert scales[1].tolist() == [8.0, 0.125] and float(amx.abs().max()) == 0.0


@pytest.mark.parametrize("B,N,H,qb", [(2, 160, 4, 0), (2, 392, 3, 0), (3, 200, 2, 72)])
def test_attention_fwd_e4m3_copy(dev, B, N, H, qb):
    """the forward that also writes its rows as e4m3 (the A operand of an fp8 proj): bf16 output and lse identical to the plain
    forward, the copy = sat(O * scale) of the same values, max|O| of the launch in the stripes (WHOLE and streaming forms, query range)"""
    from mofo_amd import ops
    F8 = torch.float8_e4m3fn
    D = H * 64
    qkv = _rand((B * N, 3 * D), dev, 1, 1.0)
    o0, o1 = (torch.empty(B * (N - qb), D, dtype=BF16, device=dev) for _ in range(2))
    l0, l1 = (torch.zeros(B * H * N, dtype=F32, device=dev) for _ in range(2))
    ops.attention_fwd(qkv, B, N, H, 0.125, o0, l0, q_begin=qb)
    amax = float(o0.float().abs().max())
    qs = torch.tensor([448.0 / (0.6 * amax)], dtype=F32, device=dev)        # saturates the largest values
    o8 = torch.zeros(B * (N - qb), D, dtype=F8, device=dev)
    am = torch.zeros(ops.FP8_AMAX_STRIPES, dtype=F32, device=dev)
    ops.attention_fwd(qkv, B, N, H, 0.125, o1, l1, q_begin=qb, out8=o8, q_scale=qs, q_amax=am)
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    lim = 0.6 * amax
    assert _rel(o8.float() / qs, o0.float().clamp(-lim, lim)) < 4e-2 and not torch.isnan(o8.float()).any()
    assert float(am.max()) == pytest.approx(amax, rel=1e-2)


def test_adamw_writes_the_e4m3_shadow(dev):
    """mofo_adamw_q8: masters, moments, bf16 shadow and the gradient norm are bit-identical to the plain update; the e4m3 shadow =
    sat(bf16 shadow * delayed scale) per weight matrix, chunks outside every matrix untouched, and the new maxima land in w_amax for
    mofo_fp8_roll_scales -- whose scales the next update uses.  A declined (gated) update changes neither shadow nor scales."""
    from mofo_amd import ops
    F8 = torch.float8_e4m3fn
    n = 40 * 1024
    seg_np = np.full(40, -1, dtype=np.int16)
    seg_np[2:11] = 0
    seg_np[11:12] = 1
    seg_np[20:39] = 2
    seg = torch.from_numpy(seg_np).to(dev)
    grp = torch.zeros(40, dtype=torch.uint8, device=dev)
    grp[30:] = 1
    p0, g = _rand((n,), dev, 1, 0.05, F32), _rand((n,), dev, 2, 0.01, F32)
    hyper = (1e-3, 0.05, 1e-3, 0.0, 0.9, 0.95, 1e-8)

    def fresh():
        return p0.clone(), torch.zeros(n, dtype=F32, device=dev), torch.zeros(n, dtype=F32, device=dev), torch.empty(n, dtype=BF16, device=dev)
    pa, ma, va, sa = fresh()
    pb, mb, vb, sb = fresh()
    part_a, part_b = (torch.empty(2048, dtype=F32, device=dev) for _ in range(2))
    na, nb = (torch.zeros(1, dtype=F32, device=dev) for _ in range(2))
    # state as runtime.FlatStore.refresh_shadow8 leaves it: exact maxima of the CURRENT weights in w_amax
    cur = p0.to(BF16)
    s8 = torch.zeros(n, dtype=F8, device=dev)
    s8.view(torch.uint8)[:2048] = 0x5A
    w_amax, w_si, w_sc = torch.empty(3, dtype=F32, device=dev), torch.ones(3, dtype=F32, device=dev), torch.ones(3, dtype=F32, device=dev)
    ops.fp8_quantize_segments(cur, seg, 3, w_amax, s8, w_si)
    amax0 = w_amax.clone()
    for step in (1, 2):
        ops.adamw(pa, g, ma, va, sa, grp, *hyper, step, norm_partial=part_a, norm_out=na)
        before = s8.clone()
        ops.fp8_roll_scales(w_amax, w_sc, w_si)
        used = w_sc.clone()
        ops.adamw(pb, g, mb, vb, sb, grp, *hyper, step, norm_partial=part_b, norm_out=nb, q8=(seg, w_sc, w_amax, s8))
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb) and torch.equal(sa, sb) and float(na) == float(nb)
        for i, (lo, hi) in enumerate(((2, 11), (11, 12), (20, 39))):
            x = sb[lo * 1024:hi * 1024].float()
            want = (x * used[i]).clamp(-448, 448).to(F8)
            assert torch.equal(s8[lo * 1024:hi * 1024].view(torch.uint8), want.view(torch.uint8))
            assert float(w_amax[i]) == float(x.abs().max())
            assert float(w_si[i]) == pytest.approx(1.0 / float(used[i]), rel=1e-6)
        if step == 1:
            assert torch.allclose(used, 448.0 / amax0, rtol=1e-6)
        for lo, hi in ((0, 2), (12, 20), (39, 40)):                        # not an fp8 operand: never written
            assert torch.equal(s8[lo * 1024:hi * 1024].view(torch.uint8), before[lo * 1024:hi * 1024].view(torch.uint8))
    # a declined update: nothing moves
    bad = torch.tensor([float("nan")], dtype=F32, device=dev)
    keep = (pb.clone(), s8.clone(), w_sc.clone(), w_si.clone(), w_amax.clone())
    ops.fp8_roll_scales(w_amax, w_sc, w_si, gate_finite=bad)
    ops.adamw(pb, g, mb, vb, sb, grp, *hyper, 3, q8=(seg, w_sc, w_amax, s8), gate_finite=bad)
    for a_, b_ in zip(keep, (pb, s8, w_sc, w_si, w_amax)):
        assert torch.equal(a_.view(torch.uint8) if a_.dtype == F8 else a_, b_.view(torch.uint8) if b_.dtype == F8 else b_)


@pytest.mark.parametrize("k2", ["0", "1"])
def test_gemm_nt_pos_rowmap(dev, monkeypatch, k2):
    """k2 = 1: the same through the one-tile-per-CU split-K kernel (the patch embed at 5 120 rows is routed to it)"""
    from mofo_amd import ops
    monkeypatch.setenv("MOFO_GEMM_K2", k2)
    Bc, nv, Ntok, K, N = 3, 20, 50, 128, 192
    M = Bc * nv
    A = _rand((M, K), dev, 1)
    W = _rand((N, K), dev, 2, 0.05)
    bias = _rand((N,), dev, 3, 1.0, F32)
    pos = _rand((Ntok, N), dev, 4, 1.0, F32)
    idx = torch.stack([torch.randperm(Ntok, generator=torch.Generator().manual_seed(b))[:nv].sort().values for b in range(Bc)]).int().to(dev)
    out = torch.full((Bc * Ntok, N), -7.0, dtype=F32, device=dev)
    ops.gemm(ops.GEMM_NT, ops.EPI_POS_F32, A, W, out, bias=bias, pos=pos, row_idx=idx.flatten(), rows_in=nv, rows_out=Ntok, row_off=0)
    ref = (A.float() @ W.float().t() + bias + pos[idx.flatten().long()]).view(Bc, nv, N)
    got = out.view(Bc, Ntok, N)
    assert _rel(got[:, :nv], ref) < 1e-5
    assert torch.all(got[:, nv:] == -7.0)      # rows outside the map untouched
    outb = torch.full((Bc * Ntok, N), -7.0, dtype=BF16, device=dev)   # the same into a bf16 stream
    ops.gemm(ops.GEMM_NT, ops.EPI_POS_BF16, A, W, outb, bias=bias, pos=pos, row_idx=idx.flatten(), rows_in=nv, rows_out=Ntok, row_off=0)
    gb = outb.view(Bc, Ntok, N)
    assert torch.equal(gb[:, :nv], ref.to(BF16)) or _rel(gb[:, :nv], ref) < 3e-3
    assert torch.all(gb[:, nv:] == -7.0)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (320, 768, 2304), (77, 128, 192), (640, 3072, 768), (50, 64, 256), (200, 136, 1600), (5000, 3064, 128)])
def test_gemm_nn_dgrad(dev, M, N, K):
    from mofo_amd import ops
    A = _rand((M, K), dev, 1)            # dY [M, Nf]
    W = _rand((K, N), dev, 2, 0.05)      # W  [Nf, Kf] read reduction-strided
    ref = A.float() @ W.float()
    Cb = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm(ops.GEMM_NN, ops.EPI_BF16, A, W, Cb)
    assert _rel(Cb, ref) < 6e-3
    idx = torch.randint(0, M * N, (64,), generator=torch.Generator().manual_seed(5))
    assert torch.allclose(Cb.flatten()[idx].float(), ref.flatten()[idx], rtol=2e-2, atol=2e-2)
    hpre = _rand((M, N), dev, 6)
    ops.gemm(ops.GEMM_NN, ops.EPI_DGELU_BF16, A, W, Cb, aux=hpre)
    x = hpre.float().requires_grad_(True)
    torch.nn.functional.gelu(x).backward(ref)
    assert _rel(Cb, x.grad) < 8e-3


@pytest.mark.parametrize("R,P,Q", [(64, 128, 128), (320, 2304, 768), (320, 768, 1536), (100, 192, 64), (1568, 384, 1536), (48, 64, 128)])
def test_gemm_tn_wgrad(dev, R, P, Q):
    from mofo_amd import ops
    dY = _rand((R, P), dev, 1)
    X = _rand((R, Q), dev, 2)
    ref = dY.float().t() @ X.float()
    C = torch.zeros(P, Q, dtype=F32, device=dev)
    ops.gemm(ops.GEMM_TN, ops.EPI_F32, dY, X, C)
    assert _rel(C, ref) < 1e-5
    idx = torch.randint(0, P * Q, (64,), generator=torch.Generator().manual_seed(5))
    assert torch.allclose(C.flatten()[idx], ref.flatten()[idx], rtol=1e-3, atol=1e-3)
    C.zero_()
    ops.gemm(ops.GEMM_TN, ops.EPI_F32, dY, X, C, splits=3, accumulate=True)
    assert _rel(C, ref) < 1e-5
    ops.gemm(ops.GEMM_TN, ops.EPI_F32, dY, X, C, splits=2, accumulate=True)
    assert _rel(C, 2 * ref) < 1e-5


def test_gemm_wgrad_fused_bias_grad(dev):
    """bias gradient (column sums of dY) riding on the wgrad GEMM via a ones-vector MFMA, with a skipped row range"""
    from mofo_amd import ops
    for R, P, Q, splits in [(320, 2304, 768, 1), (1568, 384, 1536, 3), (100, 192, 64, 1)]:
        dY, X = _rand((R, P), dev, 1), _rand((R, Q), dev, 2)
        G = torch.zeros(P, Q, dtype=F32, device=dev)
        bg = torch.full((P,), 0.5, dtype=F32, device=dev)
        lo, hi = P // 3, 2 * P // 3
        ops.gemm(ops.GEMM_TN, ops.EPI_F32, dY, X, G, splits=splits, accumulate=splits > 1, colsum=bg, colsum_skip=(lo, hi))
        assert _rel(G, dY.float().t() @ X.float()) < 1e-5
        ref = dY.float().sum(0) + 0.5
        ref[lo:hi] = 0.5
        assert _rel(bg, ref) < 1e-5
        assert torch.all(bg[lo:hi] == 0.5)


def test_gemm_grouped_wgrad(dev):
    """four weight-gradient GEMMs of different shapes in one launch == four separate launches"""
    from mofo_amd import ops
    R = 320
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    probs, refs = [], []
    for i, (P, Q) in enumerate(shapes):
        dY, X = _rand((R, P), dev, 10 + i), _rand((R, Q), dev, 20 + i)
        G = torch.zeros(P, Q, dtype=F32, device=dev)
        probs.append((dY, X, G, dict(splits=1, accumulate=False)))
        refs.append(dY.float().t() @ X.float())
    ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, probs)
    for (dY, X, G, _), ref in zip(probs, refs):
        assert _rel(G, ref) < 1e-5
    probs2 = [(dY, X, G, dict(splits=2, accumulate=True)) for dY, X, G, _ in probs[:3]]
    ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, probs2)
    for (dY, X, G, _), ref in zip(probs2, refs):
        assert _rel(G, 2 * ref) < 1e-5
    # thirteen problems in one launch (three blocks' four weight gradients + one more, as the runtime groups them), with
    # different reduction lengths inside one group
    big, bigrefs = [], []
    for rep in range(3):
        for i, (P, Q) in enumerate(shapes):
            Rr = R - 64 * rep
            dY, X = _rand((Rr, P), dev, 30 + 4 * rep + i), _rand((Rr, Q), dev, 50 + 4 * rep + i)
            big.append((dY, X, torch.full((P, Q), float("nan"), dtype=F32, device=dev), dict(splits=1, accumulate=False)))
            bigrefs.append(dY.float().t() @ X.float())
    dY, X = _rand((R, 384), dev, 70), _rand((R, 1536), dev, 71)
    big.append((dY, X, torch.full((384, 1536), float("nan"), dtype=F32, device=dev), dict(splits=1, accumulate=False)))
    bigrefs.append(dY.float().t() @ X.float())
    ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, big)
    for (dY, X, G, _), ref in zip(big, bigrefs):
        assert _rel(G, ref) < 1e-5
    with pytest.raises(RuntimeError, match="count"):
        ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, big + probs[:1])


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 384), (704, 1152, 256), (1000, 520, 768), (2048, 256, 1536)])
def test_gemm8_counted_vmcnt_family(dev, monkeypatch, M, N, K):
    """The 256 x 256 counted-vmcnt kernel (csrc/gemm8.h), forced with MOFO_GEMM8=1, for every (op, epilogue) it is built for, on
    whole and ragged tiles and several tiles per block, against fp32 torch on the same bf16 operands -- element-wise, so that a
    half-tile read before its DMA landed (a race that a norm would average away) shows."""
    from mofo_amd import ops
    monkeypatch.setenv("MOFO_GEMM8", "1")
    monkeypatch.setenv("MOFO_GEMM8_GRID", "3")     # few persistent blocks: every block walks several tiles (stream across tiles)

    def close(C, want, tol=2.5e-2):
        want = want.float()
        bad = ((C.float() - want).abs() > tol * want.abs().max()).sum().item()
        assert bad == 0, f"{bad} elements off"
        assert _rel(C, want) < 6e-3

    A = _rand((M, K), dev, 1)
    B = _rand((N, K), dev, 2, 0.05)
    bias = _rand((N,), dev, 3, 1.0, F32)
    ref = A.float() @ B.float().t()
    Cb = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A, B, Cb, bias=bias)
    close(Cb, ref + bias)
    C2 = torch.empty_like(Cb)
    ops.gemm(ops.GEMM_NT, ops.EPI_BIAS_GELU, A, B, Cb, C2=C2, bias=bias)
    close(Cb, ref + bias)
    close(C2, torch.nn.functional.gelu(ref + bias))
    R = _rand((M, N), dev, 4, 1.0, F32)
    Cf = torch.empty(M, N, dtype=F32, device=dev)
    ops.gemm(ops.GEMM_NT, ops.EPI_RESID_F32, A, B, Cf, bias=bias, resid=R)
    close(Cf, ref + bias + R)
    ops.gemm(ops.GEMM_NT, ops.EPI_RESID_BF16, A, B, Cb, bias=bias, aux=R.to(BF16))
    close(Cb, ref + bias + R.to(BF16).float())
    ops.gemm(ops.GEMM_NT, ops.EPI_F32, A, B, Cf)
    close(Cf, ref)
    ops.gemm(ops.GEMM_NT, ops.EPI_F32, A, B, Cf, accumulate=True)
    close(Cf, 2 * ref)
    # dgrad (NN): B is read reduction-strided
    Bn = B.t().contiguous()                      # [K, N]
    ops.gemm(ops.GEMM_NN, ops.EPI_BF16, A, Bn, Cb)
    close(Cb, ref)
    H = _rand((M, N), dev, 5, 1.0)
    h = H.float().requires_grad_(True)
    g, = torch.autograd.grad(torch.nn.functional.gelu(h).sum(), h)
    ops.gemm(ops.GEMM_NN, ops.EPI_DGELU_BF16, A, Bn, Cb, aux=H)
    close(Cb, ref * g)
    # wgrad (TN): both operands reduction-strided, split-K with atomics, any reduction length (here K + 8 rows)
    At = _rand((K + 8, M), dev, 6, 0.1)
    Bt = _rand((K + 8, N), dev, 7, 0.1)
    want = At.float().t() @ Bt.float()
    ops.gemm(ops.GEMM_TN, ops.EPI_F32, At, Bt, Cf)
    close(Cf, want)
    Cf.zero_()
    ops.gemm(ops.GEMM_TN, ops.EPI_F32, At, Bt, Cf, splits=2)
    close(Cf, want)
    # a grouped launch: three weight-gradient problems of different shapes in one grid (per-tile problem lookup, per-problem SRDs)
    R = K + 8
    dYs = [_rand((R, p_), dev, 20 + i, 0.1) for i, p_ in enumerate((M, 384, 264))]
    Xs = [_rand((R, q_), dev, 30 + i, 0.1) for i, q_ in enumerate((N, 136, 512))]
    Gs = [torch.empty(dy.shape[1], x.shape[1], dtype=F32, device=dev) for dy, x in zip(dYs, Xs)]
    ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, [(dy, x, g, dict(splits=1, accumulate=False)) for dy, x, g in zip(dYs, Xs, Gs)])
    for dy, x, g in zip(dYs, Xs, Gs):
        close(g, dy.float().t() @ x.float())


@pytest.mark.parametrize("stag", ["1", "0"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 768), (640, 768, 3072), (1000, 520, 704), (300, 136, 192), (5120, 768, 2304)])
def test_gemm_k2_one_tile_per_cu_family(dev, monkeypatch, M, N, K, stag):
    """The one-128x128-tile-per-CU kernel (csrc/gemm_k2.h: two wave groups, each half of the reduction through a two-stage LDS ring
    with a counted vmcnt), forced with MOFO_GEMM_K2=1, with and without the half-step stagger: every NT / NN epilogue, whole and
    ragged tiles, odd numbers of k-stages (K = 704 -> 11: the second group's last stage is empty; K = 64 / 192: one group has one
    stage or none in flight), against fp32 torch on the same bf16 operands -- element-wise, so that a stage read before its LDS-DMA
    landed (a race a norm would average away) shows."""
    from mofo_amd import ops
    monkeypatch.setenv("MOFO_GEMM_K2", "1")
    monkeypatch.setenv("MOFO_GEMM_K2_STAG", stag)

    def close(C, want, tol=2.5e-2):
        want = want.float()
        bad = ((C.float() - want).abs() > tol * want.abs().max()).sum().item()
        assert bad == 0, f"{bad} elements off"
        assert _rel(C, want) < 6e-3

    A = _rand((M, K), dev, 1)
    B = _rand((N, K), dev, 2, 0.05)
    bias = _rand((N,), dev, 3, 1.0, F32)
    ref = A.float() @ B.float().t()
    ops.gemm_route_counts(reset=True)
    Cb = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A, B, Cb, bias=bias)
    close(Cb, ref + bias)
    C2 = torch.empty_like(Cb)
    ops.gemm(ops.GEMM_NT, ops.EPI_BIAS_GELU, A, B, Cb, C2=C2, bias=bias)
    close(Cb, ref + bias)
    close(C2, torch.nn.functional.gelu(ref + bias))
    R = _rand((M, N), dev, 4, 1.0, F32)
    Cf = torch.empty(M, N, dtype=F32, device=dev)
    ops.gemm(ops.GEMM_NT, ops.EPI_RESID_F32, A, B, Cf, bias=bias, resid=R)
    close(Cf, ref + bias + R)
    ops.gemm(ops.GEMM_NT, ops.EPI_RESID_BF16, A, B, Cb, bias=bias, aux=R.to(BF16))
    close(Cb, ref + bias + R.to(BF16).float())
    ops.gemm(ops.GEMM_NT, ops.EPI_F32, A, B, Cf)
    close(Cf, ref)
    Bn = B.t().contiguous()                      # [K, N]: dgrad (NN), B read reduction-strided
    ops.gemm(ops.GEMM_NN, ops.EPI_BF16, A, Bn, Cb)
    close(Cb, ref)
    H = _rand((M, N), dev, 5, 1.0)
    h = H.float().requires_grad_(True)
    g, = torch.autograd.grad(torch.nn.functional.gelu(h).sum(), h)
    ops.gemm(ops.GEMM_NN, ops.EPI_DGELU_BF16, A, Bn, Cb, aux=H)
    close(Cb, ref * g)
    ops.gemm(ops.GEMM_NN, ops.EPI_F32, A, Bn, Cf)
    close(Cf, ref)
    assert ops.gemm_route_counts()["k2"] == 8
    # repeated launches of the same problem are bit-identical (a stage consumed before it landed would differ run to run)
    first = torch.empty_like(Cb)
    ops.gemm(ops.GEMM_NN, ops.EPI_BF16, A, Bn, first)
    for _ in range(20):
        ops.gemm(ops.GEMM_NN, ops.EPI_BF16, A, Bn, Cb)
        assert torch.equal(Cb, first)


@pytest.mark.parametrize("grid,tail", [("16", "1"), ("256", "1"), ("24", "0")])
@pytest.mark.parametrize("R,P,Q", [(192, 256, 128), (320, 768, 384), (1000, 520, 264), (2888, 512, 256), (4096, 2304, 768), (40, 64, 128)])
def test_gemm_r3_ring_family(dev, monkeypatch, R, P, Q, grid, tail):
    """The 256 x 128 three-stage ring kernel (csrc/gemm_r3.h: counted vmcnt, fragments read one k-substep ahead, one barrier per
    k-step), forced with MOFO_GEMM_R3=1 on a weight-gradient (TN, f32) problem: whole and ragged tiles, reductions that are not a
    multiple of 64, split-K, the fused bias-gradient column sums, a grouped launch with different reduction lengths.  Grid 16 makes
    every block walk several units AND share the last one of its run (rounds + tail chunks); grid 256 deals everything as tail
    chunks (more blocks than units); tail = 0 is the plain round-by-round form.  Element-wise against fp32 torch on the same
    bf16 operands, so that a stage read before its LDS-DMA landed (a race a norm would average away) shows; destinations the
    plan does not flag as shared are poisoned with NaN first (every element must be stored), flagged ones are zeroed (atomics)."""
    from mofo_amd import ops
    monkeypatch.setenv("MOFO_GEMM_R3", "1")
    monkeypatch.setenv("MOFO_GEMM_R4", "0")          # (outputs of whole 384 x 128 tiles would go to gemm_r4 by shape: test_gemm_r4_ring_family)
    monkeypatch.setenv("MOFO_GEMM_R3_GRID", grid)
    monkeypatch.setenv("MOFO_GEMM_R3_TAIL", tail)

    def close(C, want, tol=2e-3):
        bad = ((C.float() - want).abs() > tol * want.abs().max()).sum().item()
        assert bad == 0, f"{bad} elements off"
        assert _rel(C, want) < 1e-5

    def run(problems):
        used, shared = ops.gemm_grouped_plan(ops.GEMM_TN, ops.EPI_F32, problems)
        assert used
        for (_, _, G, kw), sh in zip(problems, shared):
            if kw.get("accumulate"):
                continue
            G.zero_() if sh else G.fill_(float("nan"))
        ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, problems)
        return shared

    dY, X = _rand((R, P), dev, 1, 0.1), _rand((R, Q), dev, 2, 0.1)
    want = dY.float().t() @ X.float()
    C = torch.empty(P, Q, dtype=F32, device=dev)
    ops.gemm_route_counts(reset=True)
    shared = run([(dY, X, C, dict(splits=1, accumulate=False))])
    close(C, want)
    if not shared[0]:       # one writer per element: repeated launches are bit-identical
        first = C.clone()
        for _ in range(10):
            run([(dY, X, C, dict(splits=1, accumulate=False))])
            assert torch.equal(C, first)
    if R >= 256:
        run([(dY, X, C, dict(splits=3, accumulate=False))])
        close(C, want)
        ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, [(dY, X, C, dict(splits=2, accumulate=True))])
        close(C, 2 * want)
    # bias gradient riding on the weight gradient, with a skipped row range
    bg = torch.full((P,), 0.5, dtype=F32, device=dev)
    lo, hi = P // 3, 2 * P // 3
    run([(dY, X, C, dict(splits=1, accumulate=False, colsum=bg, colsum_skip=(lo, hi)))])
    close(C, want)
    ref = dY.float().sum(0) + 0.5
    ref[lo:hi] = 0.5
    assert _rel(bg, ref) < 1e-5 and torch.all(bg[lo:hi] == 0.5)
    # a grouped launch: five problems, three reduction lengths, whole and ragged shapes
    shapes = [(P, Q, R), (384, 136, R), (256, 512, max(64, R - 72)), (264, 128, R), (768, 256, max(8, R // 2))]
    probs, wants = [], []
    for i, (p_, q_, r_) in enumerate(shapes):
        a, b = _rand((r_, p_), dev, 20 + i, 0.1), _rand((r_, q_), dev, 30 + i, 0.1)
        probs.append((a, b, torch.empty(p_, q_, dtype=F32, device=dev), dict(splits=1, accumulate=False)))
        wants.append(a.float().t() @ b.float())
    run(probs)
    for (_, _, G, _), w in zip(probs, wants):
        close(G, w)
    assert ops.gemm_route_counts()["r3"] >= 3 and ops.gemm_route_counts()["tile"] == 0


@pytest.mark.parametrize("grid,tail", [("16", "1"), ("256", "1"), ("24", "0")])
@pytest.mark.parametrize("R,P,Q", [(96, 384, 128), (320, 768, 384), (1000, 384, 256), (2888, 1152, 384), (4100, 1536, 384), (40, 384, 1536)])
def test_gemm_r4_ring_family(dev, monkeypatch, R, P, Q, grid, tail):
    """The 384 x 128 four-stage ring kernel (csrc/gemm_r4.h: 32-deep stages, counted vmcnt, single-buffered A fragments, one barrier
    per stage), forced with MOFO_GEMM_R3=1 + MOFO_GEMM_R4=1 on weight-gradient (TN, f32) problems whose outputs are whole 384 x 128
    tiles: reductions of 1, 2, 3 (mod 4) stages and not a multiple of 32, split-K, accumulate, the fused bias-gradient column sums
    with a skipped range, a grouped launch with three reduction lengths; grids as in test_gemm_r3_ring_family (rounds + tail chunks,
    all tail, plain rounds).  Element-wise against fp32 torch; destinations the plan does not flag are poisoned with NaN first."""
    from mofo_amd import ops
    monkeypatch.setenv("MOFO_GEMM_R3", "1")
    monkeypatch.setenv("MOFO_GEMM_R4", "1")
    monkeypatch.setenv("MOFO_GEMM_R3_GRID", grid)
    monkeypatch.setenv("MOFO_GEMM_R3_TAIL", tail)

    def close(C, want, tol=2e-3):
        bad = ((C.float() - want).abs() > tol * want.abs().max()).sum().item()
        assert bad == 0, f"{bad} elements off"
        assert _rel(C, want) < 1e-5

    def run(problems):
        used, shared = ops.gemm_grouped_plan(ops.GEMM_TN, ops.EPI_F32, problems)
        assert used
        for (_, _, G, kw), sh in zip(problems, shared):
            if kw.get("accumulate"):
                continue
            G.zero_() if sh else G.fill_(float("nan"))
        ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, problems)
        return shared

    dY, X = _rand((R, P), dev, 1, 0.1), _rand((R, Q), dev, 2, 0.1)
    want = dY.float().t() @ X.float()
    C = torch.empty(P, Q, dtype=F32, device=dev)
    ops.gemm_route_counts(reset=True)
    shared = run([(dY, X, C, dict(splits=1, accumulate=False))])
    close(C, want)
    if not shared[0]:
        first = C.clone()
        for _ in range(10):
            run([(dY, X, C, dict(splits=1, accumulate=False))])
            assert torch.equal(C, first)
    if R >= 256:
        run([(dY, X, C, dict(splits=3, accumulate=False))])
        close(C, want)
        ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, [(dY, X, C, dict(splits=2, accumulate=True))])
        close(C, 2 * want)
    bg = torch.full((P,), 0.5, dtype=F32, device=dev)
    lo, hi = P // 3, 2 * P // 3
    run([(dY, X, C, dict(splits=1, accumulate=False, colsum=bg, colsum_skip=(lo, hi)))])
    close(C, want)
    ref = dY.float().sum(0) + 0.5
    ref[lo:hi] = 0.5
    assert _rel(bg, ref) < 1e-5 and torch.all(bg[lo:hi] == 0.5)
    shapes = [(P, Q, R), (384, 128, R), (768, 512, max(64, R - 72)), (1152, 128, R), (384, 256, max(8, R // 2))]
    probs, wants = [], []
    for i, (p_, q_, r_) in enumerate(shapes):
        a, b = _rand((r_, p_), dev, 20 + i, 0.1), _rand((r_, q_), dev, 30 + i, 0.1)
        probs.append((a, b, torch.empty(p_, q_, dtype=F32, device=dev), dict(splits=1, accumulate=False)))
        wants.append(a.float().t() @ b.float())
    run(probs)
    for (_, _, G, _), w in zip(probs, wants):
        close(G, w)
    counts = ops.gemm_route_counts()
    assert counts["r4"] >= 3 and counts["tile"] == 0 and counts["r3"] == 0


@pytest.mark.parametrize("grid", ["16", "256"])
@pytest.mark.parametrize("slices", [8, 3, 1])
@pytest.mark.parametrize("whole", [True, False])
def test_gemm_wgrad_sliced(dev, monkeypatch, whole, slices, grid):
    """mofo_gemm_wgrad_sliced: a weight-gradient group with the reduction cut into row ranges (one per XCD label), partial sums in a
    workspace, summed by a second kernel -- no atomics on the gradients.  whole: every output is whole 384 x 128 tiles (gemm_r4);
    else ragged 256-row tiles (gemm_r3).  Problems with different reduction lengths (one whose last slice is short), fused
    bias-gradient column sums with a skipped range, destinations AND workspace poisoned with NaN (every element must be written),
    accumulate, bit-identical repeats; grid 16 makes every block walk several units of its slice."""
    from mofo_amd import ops
    monkeypatch.setenv("MOFO_GEMM_R3_GRID", grid)
    shapes = ([(384, 128, 4100), (1152, 384, 2888), (384, 1536, 4100), (768, 256, 1090)] if whole
              else [(520, 264, 4100), (384, 136, 2888), (256, 512, 4100), (264, 128, 1090)])
    probs, wants, bgs = [], [], []
    for i, (p_, q_, r_) in enumerate(shapes):
        a, b = _rand((r_, p_), dev, 40 + i, 0.1), _rand((r_, q_), dev, 50 + i, 0.1)
        bg = torch.full((p_,), 0.25, dtype=F32, device=dev)
        kw = dict(accumulate=False, colsum=bg, colsum_skip=(p_ // 3, 2 * p_ // 3)) if i % 2 == 0 else dict(accumulate=False)
        probs.append((a, b, torch.full((p_, q_), float("nan"), dtype=F32, device=dev), kw))
        wants.append(a.float().t() @ b.float())
        bgs.append((bg, a.float().sum(0), p_) if i % 2 == 0 else None)
    n = ops.gemm_wgrad_sliced_ws(probs, slices)
    assert n == slices * sum(p_ * q_ for p_, q_, _ in shapes)
    ws = torch.full((n,), float("nan"), dtype=F32, device=dev)
    ops.gemm_route_counts(reset=True)
    ops.gemm_wgrad_sliced(probs, ws, slices)
    for (_, _, G, _), w in zip(probs, wants):
        assert ((G - w).abs() > 2e-3 * w.abs().max()).sum().item() == 0 and _rel(G, w) < 1e-5
    for item in bgs:
        if item is not None:
            bg, cs, p_ = item
            ref = cs + 0.25
            ref[p_ // 3:2 * p_ // 3] = 0.25
            assert _rel(bg, ref) < 1e-5
    counts = ops.gemm_route_counts()
    assert counts["r4" if whole else "r3"] == 1 and counts["tile"] == 0
    first = [G.clone() for _, _, G, _ in probs]
    for _ in range(5):
        ops.gemm_wgrad_sliced(probs, ws, slices)
        assert all(torch.equal(G, f) for (_, _, G, _), f in zip(probs, first))
    acc = [(a, b, G, dict(accumulate=True)) for a, b, G, _ in probs]
    ops.gemm_wgrad_sliced(acc, ws, slices)
    for (_, _, G, _), w in zip(probs, wants):
        assert _rel(G, 2 * w) < 1e-5
    # a reduction shorter than the slices (most of them empty: they store zeros)
    a, b = _rand((40, 384), dev, 1, 0.1), _rand((40, 128), dev, 2, 0.1)
    G = torch.full((384, 128), float("nan"), dtype=F32, device=dev)
    ops.gemm_wgrad_sliced([(a, b, G, dict(accumulate=False))], ws, slices)
    assert _rel(G, a.float().t() @ b.float()) < 1e-5
    with pytest.raises(RuntimeError):      # a workspace that is too small is refused
        ops.gemm_wgrad_sliced(probs, ws[:n - 8], slices)


@pytest.mark.parametrize("Bc,n_all,skip,N,K", [(3, 40, 16, 192, 128), (2, 1568, 160, 384, 384), (4, 300, 44, 768, 1536), (32, 160, 24, 128, 64)])
def test_gemm_residual_row_map(dev, Bc, n_all, skip, N, K):
    """RESID_F32 / RESID_BF16 with a residual ROW MAP (rows_in > 0): the output is dense over the n_all - skip kept rows of each of
    Bc groups, the residual operand is the whole-sequence tensor (row (m / rows_in) * rows_out + row_off + m % rows_in) -- the last
    decoder block's proj GEMM.  Small groups (per-row division), the ViT-B decoder's 1408-of-1568 (one division per wave tile,
    tiles that straddle a group boundary), and a shape the one-tile-per-CU kernel takes (K = 1536, 1024 x 768)."""
    from mofo_amd import ops
    keep = n_all - skip
    M = Bc * keep
    A = _rand((M, K), dev, 1)
    W = _rand((N, K), dev, 2, 0.05)
    bias = _rand((N,), dev, 3, 1.0, F32)
    R = _rand((Bc * n_all, N), dev, 4, 1.0, F32)
    want = A.float() @ W.float().t() + bias + R.view(Bc, n_all, N)[:, skip:].reshape(M, N)
    Cf = torch.empty(M, N, dtype=F32, device=dev)
    ops.gemm(ops.GEMM_NT, ops.EPI_RESID_F32, A, W, Cf, bias=bias, resid=R, rows_in=keep, rows_out=n_all, row_off=skip)
    assert _rel(Cf, want) < 2e-3 and ((Cf - want).abs() > 0.05 * want.abs().max()).sum().item() == 0
    Rb = R.to(BF16)
    wantb = A.float() @ W.float().t() + bias + Rb.float().view(Bc, n_all, N)[:, skip:].reshape(M, N)
    Cb = torch.empty(M, N, dtype=BF16, device=dev)
    ops.gemm(ops.GEMM_NT, ops.EPI_RESID_BF16, A, W, Cb, bias=bias, aux=Rb, rows_in=keep, rows_out=n_all, row_off=skip)
    assert _rel(Cb, wantb) < 6e-3 and ((Cb.float() - wantb).abs() > 0.05 * wantb.abs().max()).sum().item() == 0
    with pytest.raises(ValueError):
        ops.gemm(ops.GEMM_NT, ops.EPI_RESID_F32, A, W, Cf, bias=bias, resid=R, rows_in=keep, rows_out=n_all, row_off=skip + 1)


@pytest.mark.parametrize("Bc,n_all,skip,D,xb", [(3, 40, 16, 384, True), (2, 1568, 160, 384, True), (5, 33, 1, 768, False)])
def test_layernorm_bwd_partial_residual(dev, Bc, n_all, skip, D, xb):
    """LayerNorm backward whose residual gradient exists for the rows t >= skip of every group of n_all rows only, stored compactly
    (the block below the last decoder block: only its masked tokens were passed on): equals the dense call with zeros on the other rows"""
    from mofo_amd import ops
    M = Bc * n_all
    x = _rand((M, D), dev, 1, 2.0, F32) + 0.5
    if xb:
        x = x.to(BF16)
    w = _rand((D,), dev, 2, 0.3, F32) + 1.0
    b = _rand((D,), dev, 3, 0.3, F32)
    y = torch.empty(M, D, dtype=BF16, device=dev)
    mean, rstd = torch.empty(M, dtype=F32, device=dev), torch.empty(M, dtype=F32, device=dev)
    ops.layernorm_fwd(x, w, b, 1e-6, y, mean, rstd)
    dy = _rand((M, D), dev, 4)
    dres_c = _rand((Bc * (n_all - skip), D), dev, 5)
    dense = torch.zeros(M, D, dtype=BF16, device=dev)
    dense.view(Bc, n_all, D)[:, skip:] = dres_c.view(Bc, n_all - skip, D)
    ws = torch.empty(2 * 1024 * D, dtype=F32, device=dev)
    outs = []
    for kw, dr in ((dict(), dense), (dict(dres_rows=(n_all, skip)), dres_c)):
        dxb = torch.empty(M, D, dtype=BF16, device=dev)
        dw, db = torch.zeros(D, dtype=F32, device=dev), torch.zeros(D, dtype=F32, device=dev)
        ops.layernorm_bwd(dy, x, w, mean, rstd, dr, None, dxb, dw, db, partial_ws=ws, **kw)
        outs.append((dxb, dw, db))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-5) and torch.allclose(outs[0][2], outs[1][2], rtol=1e-5, atol=1e-5)   # (finalize adds atomically)
    with pytest.raises(ValueError):
        ops.layernorm_bwd(dy, x, w, mean, rstd, dres_c, None, dxb, dw, db, partial_ws=ws, dres_rows=(n_all + 1, skip))


@pytest.mark.parametrize("B,N,n_vis,W", [(32, 1568, 160, 1152), (3, 288, 72, 384), (5, 64, 17, 128), (1, 40, 8, 8)])
def test_dec0_gather_reduce_inverse(dev, B, N, n_vis, W):
    """decoder block 0 row sharing (include/mofo_hip.h): gather = index_select of the cat rows, reduce = its adjoint with an f32
    sum per position (clip order), inverse = the slot table; plus the adjoint identity <gather(c), f> == <c, reduce(f)> in f64"""
    from mofo_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + N)
    n_msk = N - n_vis
    msk = torch.stack([torch.randperm(N, generator=g)[:n_msk].sort().values for _ in range(B)]).to(torch.int32).to(dev)
    cat = _rand((B * n_vis + N, W), dev, 3)
    full = torch.empty(B * N, W, dtype=BF16, device=dev)
    ops.dec0_gather(cat, msk, N, full)
    src = torch.cat([torch.arange(B * n_vis, device=dev).view(B, n_vis), B * n_vis + msk.long()], dim=1).reshape(-1)
    assert torch.equal(full, cat[src])
    inv = torch.empty(B, N, dtype=torch.int32, device=dev)
    ops.dec0_inverse(msk, N, inv)
    ref_inv = torch.full((B, N), -1, dtype=torch.int32, device=dev)
    ref_inv.scatter_(1, msk.long(), torch.arange(n_msk, dtype=torch.int32, device=dev).expand(B, n_msk).contiguous())
    assert torch.equal(inv, ref_inv)
    f = _rand((B * N, W), dev, 4)
    out = torch.full((B * n_vis + N, W), float("nan"), dtype=BF16, device=dev)
    ops.dec0_reduce(f, inv, n_vis, out)
    f3 = f.view(B, N, W)
    assert torch.equal(out[:B * n_vis], f3[:, :n_vis].reshape(B * n_vis, W))
    acc = torch.zeros(N, W, dtype=torch.float32, device=dev)
    for b in range(B):                                  # clip order, f32, one rounding at the end: what the kernel does
        acc.index_add_(0, msk[b].long(), f3[b, n_vis:].float())
    assert torch.equal(out[B * n_vis:], acc.to(BF16))
    lhs = (full.double() * f.double()).sum()
    rhs = (cat.double() * torch.cat([out[:B * n_vis].double(), acc.double()])).sum()
    assert float(lhs) == pytest.approx(float(rhs), rel=1e-9, abs=1e-6)
    with pytest.raises(ValueError):
        ops.dec0_gather(cat[:-1], msk, N, full)
    with pytest.raises(ValueError):
        ops.dec0_reduce(f, inv, n_vis + 1, out)


def test_gemm_rejects_bad_shapes(dev):
    from mofo_amd import ops
    A = _rand((64, 96), dev, 1)
    B = _rand((64, 96), dev, 2)
    C = torch.empty(64, 64, dtype=BF16, device=dev)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A, B, C)
    with pytest.raises(TypeError):
        ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A.float(), B, C)
    with pytest.raises(ValueError):
        ops.gemm(ops.GEMM_NT, ops.EPI_BF16, A.cpu(), B, C)
    # operands are addressed with UNSIGNED 32-bit buffer offsets: an operand image of 4 GiB or more is refused, not wrapped
    big = torch.empty((1 << 21) + 128, 1024, dtype=BF16, device=dev)          # 4 GiB + 256 KiB
    Wt = _rand((128, 1024), dev, 3)
    out = torch.empty(big.shape[0], 128, dtype=BF16, device=dev)
    with pytest.raises(RuntimeError, match="4 GiB"):
        ops.gemm(ops.GEMM_NT, ops.EPI_BF16, big, Wt, out)
    with pytest.raises(RuntimeError, match="4 GiB"):                          # ... as the reduction-strided operand of a wgrad
        ops.gemm(ops.GEMM_TN, ops.EPI_F32, big, big[:, :128], torch.empty(1024, 128, dtype=torch.float32, device=dev))


def test_gemm_operands_between_2_and_4_gib(dev):
    """Operand images beyond 2 GiB (the decoder's qkv / fc1 activations of ViT-L, 32 frames, at 256 clips per GPU: 2.5 / 3.3 GB): every
    buffer offset is unsigned, so up to 4 GiB minus a tile of slack is addressed.  NT (bf16 and e4m3) with a ragged last tile: rows
    below 2 GiB, across it and at the very end against torch; TN (the weight gradient: the big operand is reduction-strided, split-K
    with atomics and the unsplit ring route) against a chunked fp32 torch reduction."""
    from mofo_amd import ops
    F8 = torch.float8_e4m3fn
    g = torch.Generator(device=dev).manual_seed(11)
    M = 1_800_000 - 40                                                         # 3.69 GB of bf16 at K = 1024
    big = torch.empty(M, 1024, dtype=BF16, device=dev)
    for lo in range(0, M, 200_000):                                            # filled in pieces: no 7-GB f32 temporary
        hi = min(M, lo + 200_000)
        big[lo:hi] = torch.randn(hi - lo, 1024, generator=g, device=dev).to(BF16)
    Wt = _rand((128, 1024), dev, 3, 0.05)
    out = torch.empty(M, 128, dtype=BF16, device=dev)
    ops.gemm(ops.GEMM_NT, ops.EPI_BF16, big, Wt, out)
    two_gib_row = (1 << 31) // 2048
    for lo in (0, two_gib_row - 300, M - 700):
        ref = big[lo:lo + 700].float() @ Wt.float().t()
        assert _rel(out[lo:lo + 700], ref) < 6e-3, lo
    # e4m3: 3.0 GB of one-byte rows
    M8 = 2_900_000 + 24
    a8 = torch.empty(M8, 1024, dtype=F8, device=dev)
    for lo in range(0, M8, 400_000):
        hi = min(M8, lo + 400_000)
        a8[lo:hi] = (torch.randn(hi - lo, 1024, generator=g, device=dev) * 100.0).clamp(-448, 448).to(F8)
    w8 = (torch.randn(128, 1024, generator=g, device=dev) * 100.0).clamp(-448, 448).to(F8)
    one = torch.tensor([1e-2], dtype=F32, device=dev)
    out8 = torch.empty(M8, 128, dtype=BF16, device=dev)
    ops.gemm(ops.GEMM_NT_FP8, ops.EPI_BF16, a8, w8, out8, a_scale_inv=one, b_scale_inv=one)
    for lo in (0, (1 << 31) // 1024 - 300, M8 - 700):
        ref = (a8[lo:lo + 700].float() @ w8.float().t()) * 1e-4
        assert _rel(out8[lo:lo + 700], ref) < 6e-3, lo
    del a8, out8
    # TN: dW[m, n] = sum_k big[k, m] * big[k, n0 + n] over 1.8 M rows (every row beyond 2 GiB takes part in every output)
    Bv = big[:, 256:384]
    ref = torch.zeros(1024, 128, dtype=torch.float64, device=dev)
    for lo in range(0, M, 100_000):
        hi = min(M, lo + 100_000)
        ref += (big[lo:hi].float().t() @ Bv[lo:hi].float()).double()
    Cw = torch.zeros(1024, 128, dtype=F32, device=dev)
    ops.gemm(ops.GEMM_TN, ops.EPI_F32, big, Bv, Cw, splits=64)
    assert _rel(Cw, ref.float()) < 1e-4
    Cw.fill_(float("nan"))
    ops.gemm(ops.GEMM_TN, ops.EPI_F32, big, Bv, Cw)                            # unsplit, overwrite
    assert _rel(Cw, ref.float()) < 1e-4


def test_colsum(dev):
    from mofo_amd import ops
    for M, N in [(320, 768), (3136, 1536), (37, 64)]:
        X = _rand((M, N), dev, 1)
        out = torch.zeros(N, dtype=F32, device=dev)
        ops.colsum_bf16(X, out)
        assert _rel(out, X.float().sum(0)) < 1e-5
    # strided view (the q / v thirds of a dqkv buffer)
    X = _rand((200, 384), dev, 2)
    out = torch.zeros(128, dtype=F32, device=dev)
    ops.colsum_bf16(X[:, 256:], out)
    assert _rel(out, X[:, 256:].float().sum(0)) < 1e-5


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("M,D", [(320, 768), (3136, 384), (16, 128), (64, 64), (7, 1024), (33, 512)])
def test_layernorm_fwd_bwd(dev, M, D):
    from mofo_amd import ops
    x = _rand((M, D), dev, 1, 2.0, F32) + 0.5
    w = _rand((D,), dev, 2, 0.3, F32) + 1.0
    b = _rand((D,), dev, 3, 0.3, F32)
    y = torch.empty(M, D, dtype=BF16, device=dev)
    mean = torch.empty(M, dtype=F32, device=dev)
    rstd = torch.empty(M, dtype=F32, device=dev)
    ops.layernorm_fwd(x, w, b, 1e-6, y, mean, rstd)
    xr = x.clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), wr, br, 1e-6)
    assert _rel(y, yr) < 4e-3
    assert torch.allclose(mean, x.mean(1), atol=1e-5, rtol=1e-5)
    assert torch.allclose(rstd, 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-6), rtol=1e-4)
    dy = _rand((M, D), dev, 4)
    dres = _rand((M, D), dev, 5, 1.0, F32)
    yr.backward(dy.float())
    dx = torch.empty(M, D, dtype=F32, device=dev)
    dxb = torch.empty(M, D, dtype=BF16, device=dev)
    dw = torch.zeros(D, dtype=F32, device=dev)
    db = torch.zeros(D, dtype=F32, device=dev)
    ops.layernorm_bwd(dy, x, w, mean, rstd, dres, dx, dxb, dw, db)
    assert _rel(dx, xr.grad + dres) < 1e-5
    assert _rel(dxb, xr.grad + dres) < 4e-3
    assert _rel(dw, wr.grad) < 1e-4
    assert _rel(db, br.grad) < 1e-4
    # no residual, no bf16 copy
    dw.zero_(), db.zero_()
    ops.layernorm_bwd(dy, x, w, mean, rstd, None, dx, None, dw, db)
    assert _rel(dx, xr.grad) < 1e-5
    # residual gradient given in bf16, bf16-only output, block partials through a workspace (the form the step uses);
    # dw/db ACCUMULATE: pre-fill to check the += semantics
    dw.fill_(0.25), db.fill_(-0.5)
    dres_b = dres.to(BF16)
    ws = torch.empty(2 * 1024 * D, dtype=F32, device=dev)
    ops.layernorm_bwd(dy, x, w, mean, rstd, dres_b, None, dxb, dw, db, partial_ws=ws)
    assert _rel(dxb, xr.grad + dres_b.float()) < 4e-3
    assert _rel(dw - 0.25, wr.grad) < 1e-4 and _rel(db + 0.5, br.grad) < 1e-4
    # bf16 residual stream (the decoder): x itself is bf16; same arithmetic on the rounded input
    xb = x.to(BF16)
    xq = xb.float().requires_grad_(True)
    yq = torch.nn.functional.layer_norm(xq, (D,), wr, br, 1e-6)
    ops.layernorm_fwd(xb, w, b, 1e-6, y, mean, rstd)
    assert _rel(y, yq) < 4e-3
    assert torch.allclose(mean, xb.float().mean(1), atol=1e-5, rtol=1e-5)
    wr.grad = None
    br.grad = None
    yq.backward(dy.float())
    dw.zero_(), db.zero_()
    ops.layernorm_bwd(dy, xb, w, mean, rstd, dres_b, None, dxb, dw, db, partial_ws=ws)
    assert _rel(dxb, xq.grad + dres_b.float()) < 4e-3
    assert _rel(dw, wr.grad) < 1e-4 and _rel(db, br.grad) < 1e-4


def test_layernorm_bwd_deferred_grouped_finalize(dev):
    """dw = db = None: only block partials are written; ONE grouped launch reduces several LayerNorms (what the step does)"""
    from mofo_amd import ops
    items, want = [], []
    for k, (M, D) in enumerate([(320, 768), (3136, 384), (33, 512)]):
        x = _rand((M, D), dev, 10 + k, 2.0, F32)
        w = _rand((D,), dev, 20 + k, 0.3, F32) + 1.0
        y = torch.empty(M, D, dtype=BF16, device=dev)
        mean, rstd = torch.empty(M, dtype=F32, device=dev), torch.empty(M, dtype=F32, device=dev)
        ops.layernorm_fwd(x, w, torch.zeros_like(w), 1e-6, y, mean, rstd)
        dy = _rand((M, D), dev, 30 + k)
        dxb = torch.empty(M, D, dtype=BF16, device=dev)
        ref_dw, ref_db = torch.full((D,), 0.5, dtype=F32, device=dev), torch.full((D,), -1.0, dtype=F32, device=dev)
        ws0 = torch.empty(2 * 1024 * D, dtype=F32, device=dev)
        ops.layernorm_bwd(dy, x, w, mean, rstd, None, None, dxb, ref_dw, ref_db, partial_ws=ws0)        # immediate form
        dw, db = torch.full((D,), 0.5, dtype=F32, device=dev), torch.full((D,), -1.0, dtype=F32, device=dev)
        ws = torch.empty(2 * 1024 * D, dtype=F32, device=dev)
        dxb2 = torch.empty_like(dxb)
        nb = ops.layernorm_bwd(dy, x, w, mean, rstd, None, None, dxb2, None, None, partial_ws=ws)      # deferred
        assert nb == min(1024, -(-M // 8)) and torch.equal(dxb2, dxb)
        assert torch.all(dw == 0.5) and torch.all(db == -1.0)                                             # nothing added yet
        items.append((ws, nb, D, dw, db))
        want.append((ref_dw, ref_db))
    ops.layernorm_bwd_finalize(items)
    for (_, _, _, dw, db), (rw, rb) in zip(items, want):
        assert torch.allclose(dw, rw, rtol=1e-5, atol=1e-5) and torch.allclose(db, rb, rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        ops.layernorm_bwd_finalize(items * 14)     # more than 40 items


def test_layernorm_rowmap(dev):
    """decoder final norm: only the last n_msk rows of every clip (modeling_pretrain.py:157)."""
    from mofo_amd import ops
    Bc, N, nv, D = 3, 32, 8, 128
    nm = N - nv
    x = _rand((Bc * N, D), dev, 1, 1.0, F32)
    w = _rand((D,), dev, 2, 0.3, F32) + 1.0
    b = _rand((D,), dev, 3, 0.3, F32)
    y = torch.empty(Bc * nm, D, dtype=BF16, device=dev)
    mean = torch.empty(Bc * nm, dtype=F32, device=dev)
    rstd = torch.empty_like(mean)
    ops.layernorm_fwd(x, w, b, 1e-6, y, mean, rstd, rows_in=nm, rows_out=N, row_off=nv)
    xs = x.view(Bc, N, D)[:, nv:].reshape(-1, D).clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xs, (D,), w, b, 1e-6)
    assert _rel(y, yr) < 4e-3
    dy = _rand((Bc * nm, D), dev, 4)
    yr.backward(dy.float())
    dx = torch.zeros(Bc * N, D, dtype=F32, device=dev)
    dxb = torch.zeros(Bc * N, D, dtype=BF16, device=dev)
    dw = torch.zeros(D, dtype=F32, device=dev)
    db = torch.zeros(D, dtype=F32, device=dev)
    ops.layernorm_bwd(dy, x, w, mean, rstd, None, dx, dxb, dw, db, rows_in=nm, rows_out=N, row_off=nv)
    got = dx.view(Bc, N, D)
    assert torch.all(got[:, :nv] == 0)
    assert _rel(got[:, nv:].reshape(-1, D), xs.grad) < 1e-5


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(qkv, B, N, H, scale):
    q, k, v = qkv.float().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q * scale) @ k.transpose(-2, -1)
    p = s.softmax(-1)
    return (p @ v).transpose(1, 2).reshape(B * N, H * 64), s


@pytest.mark.parametrize("B,N,H", [(2, 160, 12), (1, 1568, 6), (3, 8, 2), (2, 32, 1), (2, 50, 3), (1, 224, 2), (1, 320, 4)])
def test_attention_fwd_bwd(dev, B, N, H):
    from mofo_amd import ops
    D = H * 64
    scale = 64 ** -0.5
    qkv = _rand((B * N, 3 * D), dev, 1, 1.5)
    out = torch.empty(B * N, D, dtype=BF16, device=dev)
    lse2 = torch.empty(B * H * N, dtype=F32, device=dev)
    ops.attention_fwd(qkv, B, N, H, scale, out, lse2)
    x = qkv.float().requires_grad_(True)
    ref, s = _attn_ref(x, B, N, H, scale)
    assert _rel(out, ref) < 8e-3
    lse_ref = torch.logsumexp(s, -1) * 1.4426950408889634     # [B,H,N] in log2 units
    assert torch.allclose(lse2.view(B, H, N), lse_ref.detach(), atol=2e-2, rtol=1e-3)
    dout = _rand((B * N, D), dev, 2)
    ref.backward(dout.float())
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B * H * N, dtype=F32, device=dev)
    ops.attention_bwd(qkv, out, dout, lse2, B, N, H, scale, dqkv, delta)
    g = x.grad
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert _rel(dqkv[:, sl], g[:, sl]) < 2e-2, name
    # delta = rowsum(dO * O) is an output of the combined entry too (N <= 160: written by the fused kernel)
    want = (dout.float() * out.float()).view(B, N, H, 64).sum(-1).permute(0, 2, 1).reshape(-1)
    assert torch.allclose(delta, want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,N,H", [(32, 160, 12), (50, 100, 6), (100, 160, 7), (9, 33, 30)])
def test_attention_bwd_short_sequences_persistent(dev, B, N, H):
    """the one-kernel backward of short sequences is persistent over the (clip, head) items (grid capped at the CU count, the next
    item's tiles prefetched into a second LDS operand set): 1.5 items per block (the encoder at B = 32), ragged tiles with
    1.2 items per block, 2.7 items per block (three iterations: both operand sets reused), 33-token sequences -- every item against
    the three-kernel form of the same library (MOFO_ATTN_NO_FUSED_BWD=1), which has its own test against fp32 torch"""
    from mofo_amd import ops
    D = H * 64
    scale = 64 ** -0.5
    qkv = _rand((B * N, 3 * D), dev, 21, 1.5)
    out = torch.empty(B * N, D, dtype=BF16, device=dev)
    lse2 = torch.empty(B * H * N, dtype=F32, device=dev)
    ops.attention_fwd(qkv, B, N, H, scale, out, lse2)
    dout = _rand((B * N, D), dev, 22)
    got = torch.full_like(qkv, float("nan"))
    delta = torch.full((B * H * N,), float("nan"), dtype=F32, device=dev)
    ops.attention_bwd(qkv, out, dout, lse2, B, N, H, scale, got, delta)
    ref = torch.empty_like(qkv)
    d2 = torch.empty_like(delta)
    ops.attention_delta(out, dout, B, N, H, d2)
    ops.attention_bwd_dkv(qkv, dout, lse2, d2, B, N, H, scale, ref)
    ops.attention_bwd_dq(qkv, dout, lse2, d2, B, N, H, scale, ref)
    assert torch.isfinite(got).all() and torch.isfinite(delta).all()
    assert torch.allclose(delta, d2, rtol=1e-5, atol=1e-5)
    g3, r3 = got.view(B, N, 3, H, 64).float(), ref.view(B, N, 3, H, 64).float()
    for part, name in enumerate(("dq", "dk", "dv")):
        # per (clip, head) item: a block that mixed up its items or read a half-landed operand set is off by O(1) in that item
        num = (g3[:, :, part] - r3[:, :, part]).pow(2).sum(dim=(1, 3)).sqrt()
        den = r3[:, :, part].pow(2).sum(dim=(1, 3)).sqrt() + 1e-20
        assert float((num / den).max()) < 1.5e-2, name
    # no atomics, no race between an item's steps and the next item's LDS-DMA: bit-identical on every further launch, also with a
    # copy stream keeping HBM busy beside it
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for rep in range(12):
        again = torch.full_like(qkv, float("nan"))
        if rep % 2:
            with torch.cuda.stream(side):
                big.copy_(big.flip(0))
        ops.attention_bwd(qkv, out, dout, lse2, B, N, H, scale, again, delta)
        assert torch.equal(again, got), rep
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,N,H,qb", [(2, 1568, 2, 0), (2, 1568, 2, 160), (1, 1561, 3, 168), (3, 100, 2, 0), (2, 160, 1, 16)])
def test_attention_dq_pass_computes_delta(dev, B, N, H, qb):
    """mofo_attention_bwd_dq_delta_range: the dQ pass computes delta = rowsum(dO * O) of its query rows itself and leaves it for the
    dK/dV pass (no delta kernel).  Same delta as mofo_attention_delta up to the order of f32 additions, same dq / dk / dv as the
    three-call sequence to that precision, rows of delta below q_begin untouched, long / short / ragged sequences."""
    from mofo_amd import ops
    D = H * 64
    scale = 64 ** -0.5
    qkv = _rand((B * N, 3 * D), dev, 31, 1.5)
    out_full = torch.empty(B * N, D, dtype=BF16, device=dev)
    lse2 = torch.empty(B * H * N, dtype=F32, device=dev)
    ops.attention_fwd(qkv, B, N, H, scale, out_full, lse2)
    nq = N - qb
    out = out_full.view(B, N, D)[:, qb:].reshape(B * nq, D).contiguous()
    dout = _rand((B * nq, D), dev, 32)
    d_ref = torch.full((B * H * N,), 7.0, dtype=F32, device=dev)
    ref = torch.zeros_like(qkv)
    ops.attention_delta(out, dout, B, N, H, d_ref, q_begin=qb)
    ops.attention_bwd_dkv(qkv, dout, lse2, d_ref, B, N, H, scale, ref, q_begin=qb)
    ops.attention_bwd_dq(qkv, dout, lse2, d_ref, B, N, H, scale, ref, q_begin=qb)
    d_got = torch.full((B * H * N,), 7.0, dtype=F32, device=dev)
    got = torch.zeros_like(qkv)
    ops.attention_bwd_dq_delta(qkv, out, dout, lse2, d_got, B, N, H, scale, got, q_begin=qb)
    ops.attention_bwd_dkv(qkv, dout, lse2, d_got, B, N, H, scale, got, q_begin=qb)
    assert torch.allclose(d_got, d_ref, rtol=1e-5, atol=1e-5)
    if qb:
        assert torch.all(d_got.view(B, H, N)[:, :, :qb] == 7.0)           # delta of the skipped queries: not written
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert _rel(got[:, sl], ref[:, sl]) < 2e-3, name


@pytest.mark.parametrize("B,N,H,qb", [(2, 1568, 2, 160), (1, 1561, 3, 168), (2, 224, 2, 40), (3, 100, 2, 32), (2, 160, 1, 16)])
def test_attention_query_range(dev, B, N, H, qb):
    """The *_range entries (queries q_begin .. N - 1 of every clip only, out / dout compact): what the last decoder block runs, whose
    visible-token outputs feed nothing.  Forward rows and dq rows are those of the whole-sequence call bit for bit (the same key
    loop per query, whatever tile the query lands in); dk / dv equal the whole-sequence backward with dO zeroed on the skipped
    rows (dP = 0 and delta = 0 there, hence dS = 0); rows below q_begin of dq are CLEARED by the dK/dV pass; long (streaming), short (all
    tiles staged at once) and ragged sequences, aligned and unaligned q_begin."""
    from mofo_amd import ops
    D = H * 64
    scale = 64 ** -0.5
    nq = N - qb
    qkv = _rand((B * N, 3 * D), dev, 1, 1.5)
    out = torch.empty(B * N, D, dtype=BF16, device=dev)
    lse2 = torch.empty(B * H * N, dtype=F32, device=dev)
    ops.attention_fwd(qkv, B, N, H, scale, out, lse2)
    out_c = torch.full((B * nq, D), 7.0, dtype=BF16, device=dev)
    lse_c = torch.full((B * H * N,), -5.0, dtype=F32, device=dev)
    ops.attention_fwd(qkv, B, N, H, scale, out_c, lse_c, q_begin=qb)
    assert torch.equal(out_c.view(B, nq, D), out.view(B, N, D)[:, qb:])
    assert torch.equal(lse_c.view(B, H, N)[:, :, qb:], lse2.view(B, H, N)[:, :, qb:])
    assert torch.all(lse_c.view(B, H, N)[:, :, :qb] == -5.0)                      # nothing written for the skipped queries
    dout = _rand((B * N, D), dev, 2)
    dout.view(B, N, D)[:, :qb] = 0
    dout_c = dout.view(B, N, D)[:, qb:].reshape(B * nq, D).contiguous()
    # whole-sequence three-pass backward with dO = 0 on the skipped rows
    dqkv = torch.zeros_like(qkv)
    delta = torch.empty(B * H * N, dtype=F32, device=dev)
    ops.attention_delta(out, dout, B, N, H, delta)
    ops.attention_bwd_dq(qkv, dout, lse2, delta, B, N, H, scale, dqkv)
    ops.attention_bwd_dkv(qkv, dout, lse2, delta, B, N, H, scale, dqkv)
    # range backward
    dq2 = torch.full_like(qkv, 3.0)
    delta2 = torch.full((B * H * N,), 9.0, dtype=F32, device=dev)
    ops.attention_delta(out_c, dout_c, B, N, H, delta2, q_begin=qb)
    assert torch.equal(delta2.view(B, H, N)[:, :, qb:], delta.view(B, H, N)[:, :, qb:]) and torch.all(delta2.view(B, H, N)[:, :, :qb] == 9.0)
    ops.attention_bwd_dq(qkv, dout_c, lse2, delta2, B, N, H, scale, dq2, q_begin=qb)
    ops.attention_bwd_dkv(qkv, dout_c, lse2, delta2, B, N, H, scale, dq2, q_begin=qb)
    a, b = dq2.view(B, N, 3 * D), dqkv.view(B, N, 3 * D)
    assert torch.equal(a[:, qb:, :D], b[:, qb:, :D])                               # dq of the range
    assert torch.all(a[:, :qb, :D] == 0.0)                                         # dq rows below q_begin: cleared by the dK/dV pass (round 6)
    for name, sl in (("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        # same products; with q_begin a multiple of 32 the query tiles coincide (only all-zero tiles are skipped: bit-identical),
        # otherwise the queries group differently inside the MFMA reduction (f32 summation order: last-bit differences in bf16)
        assert (torch.equal(a[:, :, sl], b[:, :, sl]) if qb % 32 == 0 else _rel(a[:, :, sl], b[:, :, sl]) < 3e-3), name
    # and against fp32 torch
    x = qkv.float().requires_grad_(True)
    ref, _ = _attn_ref(x, B, N, H, scale)
    ref.backward(dout.float())
    for name, sl in (("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert _rel(a[:, :, sl], x.grad.view(B, N, 3 * D)[:, :, sl]) < 2e-2, name
    assert _rel(a[:, qb:, :D], x.grad.view(B, N, 3 * D)[:, qb:, :D]) < 2e-2


def test_attention_fwd_lazy_rescale_branch(dev, monkeypatch):
    """The forward kernel moves a row's reference maximum only when a tile's partial row sum exceeds 2^MOFO_ATTN_RESCALE_THR against
    the reference the row has (default 20; the tile maximum is not computed on the common path).  Random data never takes that
    branch after the first tile, so the input FORCES it: chosen keys in late tiles (5, 23, 40 and the ragged last one) line up with
    chosen queries, scores jump by 20 to 60 log2 units there.  Checked against fp32 torch on the full tensors for thresholds 0
    (rescale on every new maximum), 6 and the shipped 20; rows below the threshold keep p <= 2^thr and must agree as well."""
    from mofo_amd import ops
    B, N, H = 2, 1568 - 7, 2
    D = H * 64
    scale = 0.125
    qkv = _rand((B * N, 3 * D), dev, 11, 1.0).float()
    q = qkv[:, :D].view(B, N, H, 64)
    k = qkv[:, D:2 * D].view(B, N, H, 64)
    for j, (row, key, mul) in enumerate(((3, 5 * 32 + 7, 3.0), (40, 23 * 32 + 1, 5.0), (700, 40 * 32 + 31, 4.0), (1500, N - 1, 6.0), (N - 1, 48 * 32, 2.5))):
        k[j % B, key, j % H] = q[j % B, row, j % H] * mul          # score = mul * |q|^2 * scale: far above every other key of that row
    # mild growth below the threshold on other rows: a later key scores ~3 log2 units above the first tiles' maximum
    k[0, 30 * 32 + 5, 0] = q[0, 100, 0] * 0.35
    qkv = qkv.to(BF16)
    x = qkv.float().requires_grad_(True)
    ref, s = _attn_ref(x, B, N, H, scale)
    lse_ref = (torch.logsumexp(s, -1) * 1.4426950408889634).detach()
    outs = {}
    for thr in ("0", "6", "20"):
        monkeypatch.setenv("MOFO_ATTN_RESCALE_THR", thr)
        out = torch.empty(B * N, D, dtype=BF16, device=dev)
        lse2 = torch.empty(B * H * N, dtype=F32, device=dev)
        ops.attention_fwd(qkv, B, N, H, scale, out, lse2)
        outs[thr] = (out.float(), lse2.view(B, H, N).clone())
    for thr in ("0", "6", "20"):
        out, lse2 = outs[thr]
        assert torch.isfinite(out).all() and torch.isfinite(lse2).all(), thr
        assert _rel(out, ref) < 8e-3, thr
        assert (out - ref.detach()).abs().max() < 0.06, thr          # every row, not just on average: a mis-scaled row is off by O(1)
        assert torch.allclose(lse2, lse_ref, atol=2e-2, rtol=1e-3), thr
    for thr in ("6", "20"):
        assert (outs["0"][0] - outs[thr][0]).abs().max() < 0.04
        assert torch.allclose(outs["0"][1], outs[thr][1], atol=1e-3, rtol=1e-5)


def test_attention_bwd_split_entries(dev):
    """delta / dQ / dK,dV as separate C-ABI calls (what the runtime issues) == the combined call, bit for bit"""
    from mofo_amd import ops
    B, N, H = 2, 224, 3
    D = H * 64
    qkv = _rand((B * N, 3 * D), dev, 1, 1.2)
    out = torch.empty(B * N, D, dtype=BF16, device=dev)
    lse2 = torch.empty(B * H * N, dtype=F32, device=dev)
    ops.attention_fwd(qkv, B, N, H, 0.125, out, lse2)
    dout = _rand((B * N, D), dev, 2)
    ref = torch.empty_like(qkv)
    delta = torch.empty(B * H * N, dtype=F32, device=dev)
    ops.attention_bwd(qkv, out, dout, lse2, B, N, H, 0.125, ref, delta)
    got = torch.zeros_like(qkv)
    d2 = torch.empty_like(delta)
    ops.attention_delta(out, dout, B, N, H, d2)
    assert torch.equal(d2, delta)
    want = (dout.float() * out.float()).view(B, N, H, 64).sum(-1).permute(0, 2, 1).reshape(-1)
    assert torch.allclose(d2, want, rtol=1e-4, atol=1e-4)
    side = torch.cuda.Stream()
    ev = torch.cuda.Event()
    ev.record()
    side.wait_event(ev)
    ops.use_stream(side)
    ops.attention_bwd_dkv(qkv, dout, lse2, d2, B, N, H, 0.125, got)
    ops.use_stream(None)
    ops.attention_bwd_dq(qkv, dout, lse2, d2, B, N, H, 0.125, got)
    torch.cuda.synchronize()
    D = H * 64
    assert torch.equal(got, ref)


@pytest.mark.parametrize("switch,val", [("MOFO_ATTN_DKV_FOLD", "0"), ("MOFO_ATTN_DKV_FOLD", "1")])
@pytest.mark.parametrize("B,N,H", [(1, 1568, 2), (2, 500, 1), (1, 3136, 1), (3, 190, 2)])
def test_attention_dkv_variants(dev, monkeypatch, switch, val, B, N, H):
    """the dK/dV pass with its accumulator-start folds off / delta only (the default folds both) against fp32 torch, ragged last
    tiles and blocks included"""
    from mofo_amd import ops
    monkeypatch.setenv(switch, val)
    D = H * 64
    scale = 64 ** -0.5
    qkv = _rand((B * N, 3 * D), dev, 11, 1.5)
    out = torch.empty(B * N, D, dtype=BF16, device=dev)
    lse2 = torch.empty(B * H * N, dtype=F32, device=dev)
    ops.attention_fwd(qkv, B, N, H, scale, out, lse2)
    x = qkv.float().requires_grad_(True)
    ref, _ = _attn_ref(x, B, N, H, scale)
    dout = _rand((B * N, D), dev, 12)
    ref.backward(dout.float())
    dqkv = torch.full_like(qkv, 3.0)
    delta = torch.empty(B * H * N, dtype=F32, device=dev)
    ops.attention_delta(out, dout, B, N, H, delta)
    ops.attention_bwd_dkv(qkv, dout, lse2, delta, B, N, H, scale, dqkv)
    assert torch.all(dqkv[:, :D] == 3.0)                       # the q third belongs to the dQ pass
    for name, sl in (("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert _rel(dqkv[:, sl], x.grad[:, sl]) < 2e-2, name


def test_attention_spiky_softmax(dev):
    """one key dominates late in the sequence -> the online-softmax rescale branch must fire and be right."""
    from mofo_amd import ops
    B, N, H = 1, 160, 1
    qkv = _rand((B * N, 192), dev, 3, 0.5)
    qkv[100, 64:128] = qkv[7, 0:64] * 40.0   # key 100 aligned with query 7
    out = torch.empty(B * N, 64, dtype=BF16, device=dev)
    lse2 = torch.empty(N, dtype=F32, device=dev)
    ops.attention_fwd(qkv, B, N, H, 0.125, out, lse2)
    ref, _ = _attn_ref(qkv, B, N, H, 0.125)
    assert _rel(out, ref) < 8e-3
    assert torch.allclose(out[7].float(), ref[7], atol=3e-2, rtol=3e-2)


# ------------------------------------------------------------------------------------------------ token plumbing
@pytest.mark.parametrize("grid,ratio", [((8, 14, 14), 0.9), ((16, 14, 14), 0.9), ((8, 2, 2), 0.75), ((1, 32, 32), 0.5)])
def test_device_tube_masks(dev, grid, ratio):
    """device-side tube masks against the oracle's numpy restatement of the same generator, bit for bit,
    plus the reference's structure: exactly int(ratio * patches) masked per frame, one pattern per clip in every temporal slot,
    a running clip counter (two calls of 3 and 4 clips == one call of 7), and the masks drive a training step like host masks"""
    from mofo_amd import ops
    from mofo_amd.masking_generator import DeviceTubeMaskingGenerator, TubeMaskingGenerator
    from oracle import pretrain_oracle as O
    F, Hh, Ww = grid
    P = Hh * Ww
    gen = DeviceTubeMaskingGenerator(grid, ratio, seed=123)
    host = TubeMaskingGenerator(grid, ratio)
    assert gen.num_masks_per_frame == host.num_masks_per_frame and repr(gen) == repr(host)
    a = gen(3, device=dev)
    b = gen(4, device=dev)
    got = torch.cat([a, b]).cpu().numpy()
    want = O.device_tube_masks(123, 0, 7, F, P, gen.num_masks_per_frame)
    assert got.dtype == np.uint8 and np.array_equal(got, want)
    per_frame = got.reshape(7, F, P)
    assert (per_frame.sum(-1) == gen.num_masks_per_frame).all()
    assert (per_frame == per_frame[:, :1]).all()
    if P >= 100:
        assert len({r.tobytes() for r in got}) == 7                               # clips differ (4 patches only have 4 patterns)
    again = DeviceTubeMaskingGenerator(grid, ratio, seed=123)(7, device=dev).cpu().numpy()
    assert np.array_equal(again, got)                                             # deterministic in (seed, clip counter)
    other = DeviceTubeMaskingGenerator(grid, ratio, seed=124)(7, device=dev).cpu().numpy()
    assert P < 100 or not np.array_equal(other, got)
    # data parallelism: ranks that share a seed draw DIFFERENT clips, and the job's masks do not depend on how the global batch is
    # split -- rank r of w takes clips [r*b, (r+1)*b) of every step's global batch of w*b clips
    g0, g1 = DeviceTubeMaskingGenerator(grid, ratio, seed=123, rank=0, world_size=2), DeviceTubeMaskingGenerator(grid, ratio, seed=123, rank=1, world_size=2)
    steps = [np.concatenate([g0(3, device=dev).cpu().numpy(), g1(3, device=dev).cpu().numpy()]) for _ in range(2)]
    one = DeviceTubeMaskingGenerator(grid, ratio, seed=123)
    assert all(np.array_equal(st_, one(6, device=dev).cpu().numpy()) for st_ in steps)
    assert P < 100 or not np.array_equal(steps[0][:3], steps[0][3:])
    # the index lists of such a mask
    nv = F * (P - gen.num_masks_per_frame)
    vis = torch.empty(7, nv, dtype=torch.int32, device=dev)
    msk = torch.empty(7, F * P - nv, dtype=torch.int32, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.mask_to_indices(torch.from_numpy(got).to(dev), nv, vis, msk, st)
    assert int(st.item()) == 0 and vis[0].cpu().tolist() == np.nonzero(got[0] == 0)[0].tolist()
    with pytest.raises(RuntimeError, match="bad sizes"):
        ops.tube_masks(0, 0, 1, 2048, 2049, torch.empty(1, 2048, dtype=torch.uint8, device=dev))


def test_mask_to_indices(dev):
    from mofo_amd import ops
    from oracle import pretrain_oracle as O
    np.random.seed(3)
    masks = np.stack([O.tube_mask((8, 14, 14), 0.9) for _ in range(5)]).astype(np.uint8)
    m = torch.from_numpy(masks).to(dev)
    vis = torch.empty(5, 160, dtype=torch.int32, device=dev)
    msk = torch.empty(5, 1408, dtype=torch.int32, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.mask_to_indices(m, 160, vis, msk, st)
    for b in range(5):
        assert vis[b].cpu().tolist() == np.nonzero(masks[b] == 0)[0].tolist()
        assert msk[b].cpu().tolist() == np.nonzero(masks[b] == 1)[0].tolist()
    assert int(st.item()) == 0
    bad = m.clone()
    bad[2, 0] = 1 - bad[2, 0]
    ops.mask_to_indices(bad, 160, vis, msk, st)
    assert int(st.item()) == 1
    # tiny geometry (N=32 < 256 threads)
    t = torch.tensor([[1, 0, 1, 1] * 8, [0, 1, 1, 1] * 8], dtype=torch.uint8, device=dev)
    vis = torch.empty(2, 8, dtype=torch.int32, device=dev)
    msk = torch.empty(2, 24, dtype=torch.int32, device=dev)
    st.zero_()
    ops.mask_to_indices(t, 8, vis, msk, st)
    assert vis[0].cpu().tolist() == list(range(1, 32, 4)) and vis[1].cpu().tolist() == list(range(0, 32, 4))
    assert int(st.item()) == 0


def test_ingest_uint8_bit_exact(dev):
    """stacked uint8 frames -> normalised f32 clip, bit-exact with the reference's CPU transform arithmetic (oracle)"""
    from mofo_amd import ops
    from oracle import pretrain_oracle as O
    for B, T, H, W in [(2, 16, 224, 224), (1, 32, 32, 48), (3, 16, 17, 33), (2, 8, 64, 64), (1, 4, 20, 36), (2, 3, 16, 16)]:   # 16- / 8- / 4- / 1-byte loads
        frames = torch.randint(0, 256, (B, H, W, T * 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(B))
        clips = torch.empty(B, 3, T, H, W, dtype=F32, device=dev)
        ops.ingest_u8(frames.to(dev), clips)
        assert torch.equal(clips.cpu(), O.ingest_uint8(frames))


@pytest.mark.parametrize("B,T,HW,n_tok", [(2, 16, 224, 160), (3, 16, 32, 4), (1, 32, 64, 100)])
def test_uint8_fused_reads_are_bit_identical(dev, B, T, HW, n_tok):
    """tubelet gather and target+MSE reading the uint8 frame stack directly == ingest_u8 followed by the
    f32 kernels, bit for bit (tokens, row losses, loss, d(loss)/d(pred)); random (non-tube) token subsets"""
    from mofo_amd import ops
    g = torch.Generator().manual_seed(B * 100 + T)
    frames = torch.randint(0, 256, (B, HW, HW, T * 3), dtype=torch.uint8, generator=g).to(dev)
    N = (T // 2) * (HW // 16) ** 2
    assert n_tok < N
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])
    vis = perm[:, :n_tok].sort(1).values.int().to(dev).contiguous()
    msk = perm[:, n_tok:].sort(1).values.int().to(dev).contiguous()
    n_msk = N - n_tok
    clips = torch.empty(B, 3, T, HW, HW, dtype=F32, device=dev)
    ops.ingest_u8(frames, clips)
    xa = torch.empty(B * n_tok, 1536, dtype=BF16, device=dev)
    xb = torch.zeros_like(xa)
    ops.patch_gather(clips, 2, 16, vis, xa)
    ops.patch_gather_u8(frames, 2, 16, vis, xb)
    assert torch.equal(xa.view(torch.int16), xb.view(torch.int16))
    pred = (torch.randn(B * n_msk, 1536, generator=g) * 0.7).to(dev).to(BF16)
    for normalize in (True, False):
        res = []
        for u8 in (False, True):
            row, loss = torch.zeros(B * n_msk, dtype=F32, device=dev), torch.zeros(1, dtype=F32, device=dev)
            dp = torch.zeros_like(pred)
            if u8:
                ops.target_mse_u8(frames, 2, 16, msk, pred, normalize, 0.5, row, loss, dp)
            else:
                ops.target_mse(clips, 2, 16, msk, pred, normalize, 0.5, row, loss, dp)
            res.append((row, loss, dp))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        assert torch.equal(res[0][2].view(torch.int16), res[1][2].view(torch.int16))
    # argument checks of the new entry points
    with pytest.raises(ValueError):
        ops.patch_gather_u8(frames[..., :-1].contiguous(), 2, 16, vis, xb)
    with pytest.raises(TypeError):
        ops.target_mse_u8(clips, 2, 16, msk, pred, True, 1.0, row, loss)


@pytest.mark.parametrize("cfgname", ["TINY", "VIT_B"])
def test_patch_gather_and_embed(dev, cfgname):
    from mofo_amd import ops
    from oracle import pretrain_oracle as O
    cfg = getattr(O, cfgname)
    Bc = 2
    x = O.keyed_clips(Bc, cfg)
    P = O.keyed_params(cfg, "xavier")
    np.random.seed(1)
    ratio = 0.75 if cfgname == "TINY" else 0.9
    mask = torch.from_numpy(np.stack([O.tube_mask(cfg.grid, ratio) for _ in range(Bc)])).bool()
    nv = int((~mask[0]).sum())
    vis = torch.stack([torch.nonzero(~mask[b]).flatten() for b in range(Bc)]).int().to(dev)
    out = torch.empty(Bc * nv, cfg.patch_dim, dtype=BF16, device=dev)
    ops.patch_gather(x.to(dev), cfg.tubelet, cfg.patch_size, vis, out)
    ref_rows = O.patchify_tubelets(x, cfg)[~mask].reshape(Bc * nv, -1)
    assert torch.equal(out.cpu(), ref_rows.to(BF16))          # pure data movement + one rounding: bit exact
    # patch-embed GEMM + bias + pos over visible tokens == reference PatchEmbed + pos + x[~mask]
    W = P["encoder.patch_embed.proj.weight"].reshape(cfg.enc_dim, -1).to(BF16).to(dev)
    bias = P["encoder.patch_embed.proj.bias"].to(dev)
    pos = O.sincos_table(cfg.num_patches, cfg.enc_dim)[0].to(dev)
    x0 = torch.empty(Bc * nv, cfg.enc_dim, dtype=F32, device=dev)
    ops.gemm(ops.GEMM_NT, ops.EPI_POS_F32, out, W, x0, bias=bias, pos=pos, row_idx=vis.flatten(), rows_in=Bc * nv, rows_out=Bc * nv)
    ref = (O.patch_embed(x, P, cfg) + O.sincos_table(cfg.num_patches, cfg.enc_dim))[~mask].reshape(Bc * nv, -1)
    assert _rel(x0.cpu(), ref) < 5e-3


def test_assemble_fwd_bwd(dev):
    from mofo_amd import ops
    Bc, N, nv, D = 3, 32, 8, 128
    tok = _rand((D,), dev, 1, 0.02, F32)
    pos = _rand((N, D), dev, 2, 1.0, F32)
    msk = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b))[: N - nv].sort().values for b in range(Bc)]).int().to(dev)
    xf = torch.full((Bc, N, D), 3.0, dtype=F32, device=dev)
    ops.fill_mask_tokens(tok, pos, msk, nv, xf)
    assert torch.all(xf[:, :nv] == 3.0)
    assert torch.equal(xf[:, nv:], (tok + pos[msk.long()]))
    xfb = torch.full((Bc, N, D), 3.0, dtype=BF16, device=dev)           # bf16 decoder stream
    ops.fill_mask_tokens(tok, pos, msk, nv, xfb)
    assert torch.all(xfb[:, :nv] == 3.0) and torch.equal(xfb[:, nv:], (tok + pos[msk.long()]).to(BF16))
    dx = _rand((Bc, N, D), dev, 3, 1.0, F32)
    de = torch.empty(Bc * nv, D, dtype=BF16, device=dev)
    dt = torch.zeros(D, dtype=F32, device=dev)
    ops.assemble_bwd(dx, nv, de, dt)
    assert torch.equal(de.view(Bc, nv, D), dx[:, :nv].to(BF16))
    assert _rel(dt, dx[:, nv:].sum((0, 1))) < 1e-5
    dxh = dx.to(BF16)
    dt.zero_()
    ops.assemble_bwd(dxh, nv, de, dt)
    assert torch.equal(de.view(Bc, nv, D), dxh[:, :nv])
    assert _rel(dt, dxh[:, nv:].float().sum((0, 1))) < 1e-5
    # the two-launch form: block partials in a workspace instead of same-address atomics (what the runtime uses)
    ws = torch.full((ops.assemble_bwd_blocks(Bc, N) * D,), float("nan"), dtype=F32, device=dev)
    for src in (dx, dxh):
        de.zero_(), dt.fill_(1.0)
        ops.assemble_bwd(src, nv, de, dt, partial_ws=ws)
        assert torch.equal(de.view(Bc, nv, D), src[:, :nv].to(BF16))
        assert _rel(dt - 1.0, src[:, nv:].float().sum((0, 1))) < 1e-5
    with pytest.raises(ValueError):
        ops.assemble_bwd(dx, nv, de, dt, partial_ws=ws[:-1])
    # the deferred form (what the runtime uses: the reduce of d(mask_token) runs later, on the side stream)
    if D % 8 == 0 and D <= 512:
        for src in (dx, dxh):
            de.zero_(), dt.fill_(2.0), ws.fill_(float("nan"))
            ops.assemble_bwd(src, nv, de, None, partial_ws=ws)
            assert torch.equal(de.view(Bc, nv, D), src[:, :nv].to(BF16)) and torch.all(dt == 2.0)
            ops.assemble_bwd_finalize(ws, Bc, N, dt)
            assert _rel(dt - 2.0, src[:, nv:].float().sum((0, 1))) < 1e-5


@pytest.mark.parametrize("cfgname,normalize", [("TINY", True), ("VIT_B", True), ("VIT_B", False)])
def test_target_mse(dev, cfgname, normalize):
    from mofo_amd import ops
    from oracle import pretrain_oracle as O
    cfg = getattr(O, cfgname)
    Bc = 2
    x = O.keyed_clips(Bc, cfg)
    np.random.seed(2)
    ratio = 0.75 if cfgname == "TINY" else 0.9
    mask = torch.from_numpy(np.stack([O.tube_mask(cfg.grid, ratio) for _ in range(Bc)])).bool()
    nm = int(mask[0].sum())
    msk = torch.stack([torch.nonzero(mask[b]).flatten() for b in range(Bc)]).int().to(dev)
    labels = O.build_targets(x, mask, cfg, normalize)                       # oracle, engine_for_pretraining.py:43-63
    pred = _rand((Bc * nm, 1536), dev, 5)
    row_loss = torch.empty(Bc * nm, dtype=F32, device=dev)
    loss = torch.empty(1, dtype=F32, device=dev)
    dpred = torch.empty_like(pred)
    tgt = torch.empty(Bc * nm, 1536, dtype=F32, device=dev)
    ops.target_mse(x.to(dev), 2, 16, msk, pred, normalize, 1.0, row_loss, loss, dpred, tgt)
    assert torch.allclose(tgt.cpu(), labels.reshape(Bc * nm, 1536), rtol=2e-4, atol=2e-5)
    pr = pred.float().cpu().requires_grad_(True)
    ref = O.mse_loss(pr, labels.reshape(Bc * nm, 1536))
    ref.backward()
    assert float(loss.item()) == pytest.approx(float(ref), rel=1e-5)
    assert _rel(dpred.cpu(), pr.grad) < 4e-3


# ------------------------------------------------------------------------------------------------ optimizer
def test_sumsq_adamw_cast(dev):
    from mofo_amd import ops
    n = 1024 * 37
    g = _rand((n,), dev, 1, 0.3, F32)
    partial = torch.empty(1024, dtype=F32, device=dev)
    norm = torch.empty(1, dtype=F32, device=dev)
    ops.sumsq_norm(g, partial, norm)
    assert float(norm.item()) == pytest.approx(float(g.double().norm()), rel=1e-6)
    p = _rand((n,), dev, 2, 1.0, F32)
    grp = (torch.arange(37) % 3 == 0).to(torch.uint8).to(dev)       # group 1 = no decay here
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    pb = torch.empty(n, dtype=BF16, device=dev)
    pr, mr, vr = p.clone().double(), m.clone().double(), v.clone().double()
    lr0, wd0, lr1, wd1 = 1.5e-4, 0.05, 3e-4, 0.0
    for step in (1, 2, 3):
        ops.adamw(p, g, m, v, pb, grp, lr0, wd0, lr1, wd1, 0.9, 0.95, 1e-8, step)
        lr = torch.where(grp.repeat_interleave(1024).bool(), lr1, lr0).double()
        wd = torch.where(grp.repeat_interleave(1024).bool(), wd1, wd0).double()
        gd = g.double()
        pr = pr * (1 - lr * wd)
        mr = 0.9 * mr + 0.1 * gd
        vr = 0.95 * vr + 0.05 * gd * gd
        pr = pr - lr / (1 - 0.9 ** step) * mr / (vr.sqrt() / math.sqrt(1 - 0.95 ** step) + 1e-8)
    assert _rel(p, pr) < 1e-6
    assert _rel(m, mr) < 1e-6 and _rel(v, vr) < 1e-6
    assert torch.equal(pb, p.to(BF16))
    # clipping: max_norm below the norm scales the gradient by max_norm / (norm + 1e-6)
    p2, m2, v2 = p.clone(), torch.zeros_like(p), torch.zeros_like(p)
    ops.adamw(p2, g, m2, v2, None, grp, lr0, 0.0, lr0, 0.0, 0.9, 0.95, 1e-8, 1, grad_norm=norm, max_norm=0.5)
    coef = 0.5 / (float(norm.item()) + 1e-6)
    assert _rel(m2, 0.1 * g * coef) < 1e-5
    # the un-clipped step can leave the gradient norm as a by-product (no separate pass over g)
    p3, m3, v3 = p.clone(), torch.zeros_like(p), torch.zeros_like(p)
    np_, no_ = torch.empty(2048, dtype=F32, device=dev), torch.zeros(1, dtype=F32, device=dev)
    ops.adamw(p3, g, m3, v3, None, grp, lr0, 0.0, lr0, 0.0, 0.9, 0.95, 1e-8, 1, norm_partial=np_, norm_out=no_)
    assert float(no_.item()) == pytest.approx(float(g.double().norm()), rel=1e-6)
    assert _rel(m3, 0.1 * g) < 1e-6
    dst = torch.empty(n, dtype=BF16, device=dev)
    ops.cast_bf16(p, dst)
    assert torch.equal(dst, p.to(BF16))

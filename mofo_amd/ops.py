"""Thin, checked wrappers: torch tensors in, C-ABI calls on torch's current HIP stream out.

PyTorch is plumbing here (device memory + streams); every computation is a kernel in libmofo_hip.so.
These wrappers validate device / dtype / layout so that a wrong shape raises here instead of faulting on the GPU.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (EPI_BF16, EPI_BIAS_GELU, EPI_DGELU_BF16, EPI_F32, EPI_POS_BF16, EPI_POS_F32, EPI_RESID_BF16, EPI_RESID_F32,
                   GEMM_NN, GEMM_NT, GEMM_NT_FP8, GEMM_TN, GemmArgs)

BF16, F32, I32, U8 = torch.bfloat16, torch.float32, torch.int32, torch.uint8
F8 = torch.float8_e4m3fn     # OCP e4m3 (gfx950's fp8), the storage type of the fp8 forward GEMM operands


_STREAM_OVERRIDE = None   # a torch.cuda.Stream while a launch sequence runs part of its work on a side stream


def _stream():
    s = _STREAM_OVERRIDE
    return (s if s is not None else torch.cuda.current_stream()).cuda_stream


def launch_stream():
    """the torch stream object the C-ABI launches currently go to (the override of use_stream, else torch's current stream)"""
    s = _STREAM_OVERRIDE
    return s if s is not None else torch.cuda.current_stream()


def use_stream(stream):
    """route the following C-ABI launches to ``stream`` (a torch.cuda.Stream) or, with None, back to torch's current
    stream; recorded into launch lists so that a replay switches at the same place"""
    global _STREAM_OVERRIDE
    rec = _lib.RECORDER
    if rec is not None:
        rec.append(("stream", stream, None, None, 0.0))
    _STREAM_OVERRIDE = None if _serialize() else stream


def _serialize():
    """while bench.py's FULL per-kernel profiler is on, everything runs on one stream: event pairs then time each kernel
    alone (comparable with rocprofv3's per-kernel durations) instead of two streams' kernels interleaved"""
    prof = _lib.PROFILER
    return prof is not None and prof.only is None


def _chk(t, dtype, name, dims=None):
    if t is None:
        raise ValueError(f"{name} is None")
    if not t.is_cuda:
        raise ValueError(f"{name} must be on the GPU (mofo_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if dims is not None and t.dim() != dims:
        raise ValueError(f"{name} must have {dims} dims, got {tuple(t.shape)}")
    if t.dim() >= 1 and t.stride(-1) != 1 and t.numel() > 0:
        raise ValueError(f"{name} must be contiguous in its last dim")
    return t


def _ld(t):
    """leading dimension (elements) of a 2-D row-major view"""
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


def _p(t):
    return 0 if t is None else t.data_ptr()


def _run(name, key, work, *args):
    """one C-ABI call on the current stream; optionally recorded into a launch list (runtime replays it without the
    Python-side checks) and optionally bracketed by HIP events (bench.py's per-kernel timing)"""
    fn = getattr(_lib.load(), name)
    rec = _lib.RECORDER
    if rec is not None:
        rec.append((fn, args, name, key, work))
    prof = _lib.PROFILER
    if prof is None:
        _lib.check(fn(*args, _stream()), name)
    else:
        prof.begin(key, work, _STREAM_OVERRIDE)
        _lib.check(fn(*args, _stream()), name)
        prof.end(_STREAM_OVERRIDE)


def host_op(f):
    """a torch-side op (memset, hook) that must keep its place in the launch order of a recorded list"""
    rec = _lib.RECORDER
    if rec is not None:
        rec.append((None, f, None, None, 0.0))
    f()


def replay(launches):
    """re-issue a recorded launch list on the current stream (pointers and sizes were validated when it was recorded)"""
    base = torch.cuda.current_stream().cuda_stream
    s, sobj = base, None
    prof = _lib.PROFILER
    for fn, args, name, key, work in launches:
        if fn is None:
            args()
        elif fn == "stream":
            sobj = None if _serialize() else args
            s = base if sobj is None else sobj.cuda_stream
        elif prof is None:
            rc = fn(*args, s)
            if rc:
                _lib.check(rc, name)
        else:
            prof.begin(key, work, sobj)
            rc = fn(*args, s)
            prof.end(sobj)
            if rc:
                _lib.check(rc, name)


def gemm(op, epi, A, B, C_, **kw):
    """C = epilogue(A (op) B).  A, B bf16 2-D; see include/mofo_hip.h for op / epilogue semantics."""
    a, flops, nbytes = _gemm_args(op, epi, A, B, C_, **kw)
    _run("mofo_gemm", ("gemm", op, epi), (flops, nbytes), C.byref(a))
    return C_


def gemm_grouped(op, epi, problems):
    """several GEMMs of one (op, epilogue) kind in ONE launch; ``problems`` = [(A, B, C, kwargs), ...] (at most 13; 32 for
    weight-gradient groups, TN + F32)"""
    built = [_gemm_args(op, epi, A, B, C_, **kw) for A, B, C_, kw in problems]
    arr = (GemmArgs * len(built))(*[b[0] for b in built])
    _run("mofo_gemm_grouped", ("gemm", op, epi), (sum(b[1] for b in built), sum(b[2] for b in built)), arr, len(built))


def gemm_grouped_plan(op, epi, problems):
    """host-only: (routed to the 256 x 128 ring kernel?, per problem: may its C receive f32 atomic adds from several workgroups?) --
    the caller zeroes those destinations before the launch (include/mofo_hip.h: mofo_gemm_grouped_plan)"""
    built = [_gemm_args(op, epi, A, B, C_, **kw) for A, B, C_, kw in problems]
    arr = (GemmArgs * len(built))(*[b[0] for b in built])
    shared = (C.c_int * len(built))()
    rc = _lib.load().mofo_gemm_grouped_plan(arr, len(built), shared)
    if rc < 0:
        _lib.check(rc, "mofo_gemm_grouped_plan")
    return rc == 1, [bool(x) for x in shared]


def _gemm_args(op, epi, A, B, C_, *, C2=None, bias=None, resid=None, aux=None, pos=None, row_idx=None, rows_in=0, rows_out=0,
               row_off=0, splits=1, accumulate=False, colsum=None, colsum_skip=(0, 0), a_scale_inv=None, b_scale_inv=None,
               C8=None, q_scale=None, q_amax=None):
    if op == GEMM_NT_FP8:
        _chk(A, F8, "A", 2), _chk(B, F8, "B", 2), _chk(a_scale_inv, F32, "a_scale_inv"), _chk(b_scale_inv, F32, "b_scale_inv")
    else:
        _chk(A, BF16, "A", 2), _chk(B, BF16, "B", 2)
    if op in (GEMM_NT, GEMM_NT_FP8):
        M, K = A.shape
        N, K2 = B.shape
    elif op == GEMM_NN:
        M, K = A.shape
        K2, N = B.shape
    elif op == GEMM_TN:
        K, M = A.shape
        K2, N = B.shape
    else:
        raise ValueError("bad op")
    if K != K2:
        raise ValueError(f"reduction mismatch: {tuple(A.shape)} vs {tuple(B.shape)} for op {op}")
    out_dtype = F32 if epi in (EPI_RESID_F32, EPI_POS_F32, EPI_F32) else BF16
    _chk(C_, out_dtype, "C", 2)
    if epi not in (EPI_POS_F32, EPI_POS_BF16) and tuple(C_.shape) != (M, N):
        raise ValueError(f"C must be {(M, N)}, got {tuple(C_.shape)}")
    if epi in (EPI_POS_F32, EPI_POS_BF16):
        _chk(pos, F32, "pos", 2), _chk(row_idx, I32, "row_idx")
        if row_idx.numel() != M or pos.shape[1] != N or C_.shape[1] != N or rows_in <= 0:
            raise ValueError("POS_F32: row_idx must have M entries, pos/C must have N columns")
        if (M // rows_in) * rows_out + row_off > C_.shape[0] or M % rows_in:
            raise ValueError("POS_F32: row map exceeds C")
    if bias is not None:
        _chk(bias, F32, "bias", 1)
        if bias.numel() != N:
            raise ValueError("bias must have N entries")
    if epi == EPI_BIAS_GELU:
        _chk(C2, BF16, "C2", 2)
        if tuple(C2.shape) != (M, N):
            raise ValueError("C2 shape")
    if C8 is not None:       # e4m3 copy of the activation (the A operand of an fp8 fc2), delayed scale, amax stripes
        if op != GEMM_NT_FP8 or epi != EPI_BIAS_GELU:
            raise ValueError("C8 rides on NT_FP8 + BIAS_GELU")
        _chk(C8, F8, "C8", 2), _chk(q_scale, F32, "q_scale"), _chk(q_amax, F32, "q_amax")
        if tuple(C8.shape) != (M, N) or q_amax.numel() != FP8_AMAX_STRIPES or not q_amax.is_contiguous():
            raise ValueError("C8 must be [M, N]; q_amax the %d stripes of one site" % FP8_AMAX_STRIPES)
    rmapped = epi in (EPI_RESID_F32, EPI_RESID_BF16) and rows_in > 0     # residual rows (m / rows_in) * rows_out + row_off + m % rows_in
    rrows = ((M // rows_in - 1) * rows_out + row_off + rows_in) if rmapped else M
    if rmapped and (M % rows_in or row_off < 0 or row_off + rows_in > rows_out):
        raise ValueError("residual row map: M % rows_in == 0, row_off + rows_in <= rows_out")
    if epi == EPI_RESID_F32:
        _chk(resid, F32, "resid", 2)
        if resid.shape[1] != N or (resid.shape[0] != M if not rmapped else resid.shape[0] < rrows):
            raise ValueError("resid shape")
    if epi in (EPI_DGELU_BF16, EPI_RESID_BF16):
        _chk(aux, BF16, "aux", 2)
        if aux.shape[1] != N or (aux.shape[0] != M if not (rmapped and epi == EPI_RESID_BF16) else aux.shape[0] < rrows):
            raise ValueError("aux shape")
    if colsum is not None:
        _chk(colsum, F32, "colsum", 1)
        if colsum.numel() != M or op != GEMM_TN or epi != EPI_F32:
            raise ValueError("colsum: f32 [M], TN + F32 epilogue only")
    a = GemmArgs(op=op, epilogue=epi, M=M, N=N, K=K, A=_p(A), lda=_ld(A), B=_p(B), ldb=_ld(B), C=_p(C_), ldc=_ld(C_),
                 C2=_p(C2), ldc2=_ld(C2) if C2 is not None else 0, bias=_p(bias), resid=_p(resid),
                 ldr=_ld(resid) if resid is not None else 0, aux=_p(aux), ldaux=_ld(aux) if aux is not None else 0,
                 pos=_p(pos), ldpos=_ld(pos) if pos is not None else 0, row_idx=_p(row_idx), rows_in=rows_in,
                 rows_out=rows_out, row_off=row_off, splits=splits, accumulate=1 if accumulate else 0, colsum=_p(colsum),
                 colsum_skip_lo=colsum_skip[0], colsum_skip_hi=colsum_skip[1], a_scale_inv=_p(a_scale_inv), b_scale_inv=_p(b_scale_inv),
                 C8=_p(C8), ldc8=_ld(C8) if C8 is not None else 0, q_scale=_p(q_scale), q_amax=_p(q_amax))
    # algorithmic HBM bytes of the launch: every operand read once, every output written once (bench.py's roofline block)
    esz = 4.0 if out_dtype == F32 else 2.0
    nbytes = A.element_size() * (M * K + N * K) + esz * M * N
    if epi == EPI_BIAS_GELU:
        nbytes += (3.0 if C8 is not None else 2.0) * M * N
    elif epi == EPI_RESID_F32:
        nbytes += resid.element_size() * M * N
    elif epi in (EPI_DGELU_BF16, EPI_RESID_BF16):
        nbytes += 2.0 * M * N
    elif epi in (EPI_POS_F32, EPI_POS_BF16):
        nbytes += 4.0 * M * N
    return a, 2.0 * M * N * K, nbytes


def gemm_wgrad_sliced_ws(problems, slices=8):
    """f32 elements of workspace ``gemm_wgrad_sliced`` needs for these weight-gradient problems"""
    built = [_gemm_args(GEMM_TN, EPI_F32, A, B, C_, **kw) for A, B, C_, kw in problems]
    arr = (GemmArgs * len(built))(*[b[0] for b in built])
    n = _lib.load().mofo_gemm_wgrad_sliced_ws(arr, len(built), slices)
    if n < 0:
        _lib.check(int(n), "mofo_gemm_wgrad_sliced_ws")
    return int(n)


def gemm_wgrad_sliced(problems, ws, slices=8):
    """a group of weight gradients C (+)= A^T B (TN, f32; fused bias-gradient column sums) with the token reduction cut into
    ``slices`` row ranges, one per XCD: partial sums go to ``ws`` with plain stores (no destination needs zeroing, no atomics),
    a second kernel sums the slices into C (include/mofo_hip.h: mofo_gemm_wgrad_sliced)"""
    _chk(ws, F32, "ws", 1)
    built = [_gemm_args(GEMM_TN, EPI_F32, A, B, C_, **kw) for A, B, C_, kw in problems]
    arr = (GemmArgs * len(built))(*[b[0] for b in built])
    _run("mofo_gemm_wgrad_sliced", ("gemm", GEMM_TN, EPI_F32), (sum(b[1] for b in built), sum(b[2] for b in built)),
         arr, len(built), slices, ws.data_ptr(), ws.numel())


GEMM_ROUTES = ("tile", "persistent", "persistent256", "ksplit", "gemm8", "fp8", "k2", "r3", "r4")


def gemm_route_counts(reset=False):
    """launches per GEMM main-loop family since the last reset (include/mofo_hip.h: mofo_gemm_route_counts)"""
    buf = (C.c_longlong * len(GEMM_ROUTES))()
    _lib.check(_lib.load().mofo_gemm_route_counts(buf, len(GEMM_ROUTES), 1 if reset else 0), "mofo_gemm_route_counts")
    return {name: int(buf[i]) for i, name in enumerate(GEMM_ROUTES)}


def colsum_bf16(X, out):
    _chk(X, BF16, "X", 2), _chk(out, F32, "out", 1)
    if out.numel() != X.shape[1]:
        raise ValueError("out must have N entries")
    _run("mofo_colsum_bf16", ("colsum",), 2.0 * X.shape[0] * X.shape[1], _p(X), _ld(X), X.shape[0], X.shape[1], _p(out))
    return out


def layernorm_fwd(x, w, b, eps, y, mean, rstd, M=None, rows_in=None, rows_out=None, row_off=0):
    """``x``: the residual stream, f32 or bf16 (the decoder's)"""
    if x is None or x.dtype not in (F32, BF16):
        raise TypeError("x must be f32 or bf16")
    _chk(x, x.dtype, "x", 2), _chk(w, F32, "w", 1), _chk(b, F32, "b", 1), _chk(y, BF16, "y", 2), _chk(mean, F32, "mean", 1), _chk(rstd, F32, "rstd", 1)
    D = x.shape[1]
    M = y.shape[0] if M is None else M
    rows_in = M if rows_in is None else rows_in
    rows_out = rows_in if rows_out is None else rows_out
    if w.numel() != D or b.numel() != D or y.shape[1] != D or mean.numel() < M or rstd.numel() < M or y.shape[0] < M:
        raise ValueError("layernorm_fwd: shape mismatch")
    if M % rows_in or (M // rows_in - 1) * rows_out + row_off + rows_in > x.shape[0]:
        raise ValueError("layernorm_fwd: row map exceeds x")
    xb = 1 if x.dtype == BF16 else 0
    _run("mofo_layernorm_fwd", ("ln_fwd",), (4.0 if xb else 6.0) * M * D, _p(x), xb, _ld(x), _p(w), _p(b), eps, M, D, rows_in, rows_out, row_off,
         _p(y), _ld(y), _p(mean), _p(rstd))
    return y


def layernorm_fwd_q(x, w, b, eps, y, mean, rstd, y8, qscale, amax_out):
    """layernorm_fwd that also writes the rows as e4m3 (y8 = sat(y * qscale[0])) and reports max|y| into amax_out's stripes"""
    if x is None or x.dtype not in (F32, BF16):
        raise TypeError("x must be f32 or bf16")
    _chk(x, x.dtype, "x", 2), _chk(w, F32, "w", 1), _chk(b, F32, "b", 1), _chk(y, BF16, "y", 2), _chk(mean, F32, "mean", 1), _chk(rstd, F32, "rstd", 1)
    _chk(y8, F8, "y8", 2), _chk(qscale, F32, "qscale"), _chk(amax_out, F32, "amax_out")
    if amax_out.numel() != FP8_AMAX_STRIPES or not amax_out.is_contiguous():
        raise ValueError("layernorm_fwd_q: amax_out must be the %d stripes of one site" % FP8_AMAX_STRIPES)
    M, D = y.shape
    if x.shape != (M, D) or y8.shape != (M, D) or w.numel() != D or b.numel() != D or mean.numel() < M or rstd.numel() < M:
        raise ValueError("layernorm_fwd_q: shape mismatch")
    xb = 1 if x.dtype == BF16 else 0
    _run("mofo_layernorm_fwd_q", ("ln_fwd",), (5.0 if xb else 7.0) * M * D, _p(x), xb, _ld(x), _p(w), _p(b), eps, M, D, M, M, 0, _p(y), _ld(y),
         _p(mean), _p(rstd), _p(y8), _ld(y8), _p(qscale), _p(amax_out))
    return y8


def fp8_quantize_segments(x_bf16, chunk_seg, nseg, amax_ws, out8, scale_inv):
    """flat bf16 buffer -> flat e4m3 buffer, one scale per segment (chunk_seg: int16 per 1024-element chunk, -1 = skip)"""
    _chk(x_bf16, BF16, "x", 1), _chk(out8, F8, "out8", 1), _chk(amax_ws, F32, "amax_ws", 1), _chk(scale_inv, F32, "scale_inv", 1)
    if chunk_seg is None or chunk_seg.dtype != torch.int16 or not chunk_seg.is_cuda:
        raise TypeError("chunk_seg must be an int16 GPU tensor")
    n = x_bf16.numel()
    if out8.numel() != n or chunk_seg.numel() * 1024 != n or amax_ws.numel() < nseg or scale_inv.numel() < nseg:
        raise ValueError("fp8_quantize_segments: size mismatch")
    _run("mofo_fp8_quantize_segments", ("fp8_quant_w",), 5.0 * n, _p(x_bf16), n, _p(chunk_seg), int(nseg), _p(amax_ws), _p(out8), _p(scale_inv))


def fp8_quantize_bf16(x, scale, out8, amax_out=None):
    _chk(x, BF16, "x"), _chk(scale, F32, "scale"), _chk(out8, F8, "out8")
    if not x.is_contiguous() or not out8.is_contiguous() or out8.numel() != x.numel():
        raise ValueError("fp8_quantize_bf16: contiguous tensors of one size")
    if amax_out is not None:
        _chk(amax_out, F32, "amax_out")
    _run("mofo_fp8_quantize_bf16", ("fp8_quant",), 3.0 * x.numel(), _p(x), x.numel(), _p(scale), _p(out8), _p(amax_out))
    return out8


FP8_AMAX_STRIPES = 1024   # include/mofo_hip.h MOFO_FP8_AMAX_STRIPES


def fp8_update_scales(amax, scales, margin=1.5):
    _chk(amax, F32, "amax", 2), _chk(scales, F32, "scales", 2)
    if amax.shape[1] != FP8_AMAX_STRIPES or not amax.is_contiguous():
        raise ValueError("fp8_update_scales: amax must be a contiguous [sites, %d] tensor" % FP8_AMAX_STRIPES)
    n = amax.shape[0]
    if tuple(scales.shape) != (n, 2) or not scales.is_contiguous():
        raise ValueError("fp8_update_scales: scales must be [n, 2]")
    _run("mofo_fp8_update_scales", ("fp8_scales",), 12.0 * n, _p(amax), _p(scales), n, float(margin))


def fp8_roll_scales(amax, scale, scale_inv, gate_finite=None, gate_zero=None, gate_one=None):
    """weights' delayed scaling, once per step before the first adamw(q8=...) launch: scale = 448 / amax, scale_inv = amax / 448, amax = 0;
    the gate words of that update (a declined update changes neither the e4m3 shadow nor its scales)"""
    _chk(amax, F32, "amax", 1), _chk(scale, F32, "scale", 1), _chk(scale_inv, F32, "scale_inv", 1)
    n = amax.numel()
    if scale.numel() != n or scale_inv.numel() != n:
        raise ValueError("fp8_roll_scales: three arrays of one length")
    _run("mofo_fp8_roll_scales", ("fp8_scales",), 16.0 * n, _p(amax), _p(scale), _p(scale_inv), n, _p(gate_finite), _p(gate_zero), _p(gate_one))


def layernorm_bwd_blocks(M):
    """block rows a LayerNorm backward over M rows leaves in its partial_ws"""
    return _lib.load().mofo_layernorm_bwd_blocks(int(M))


def layernorm_bwd(dy, x, w, mean, rstd, dres, dx, dxb, dw, db, M=None, rows_in=None, rows_out=None, row_off=0, partial_ws=None, dres_rows=None):
    """dx = dres + LN'(dy).  ``dres`` may be None, an f32 tensor or a bf16 tensor (shape of x); ``dx`` (f32) and ``dxb``
    (bf16) are the outputs, either may be None but not both.  ``dres_rows`` = (period, skip): the residual gradient covers only the
    rows t >= skip of every group of period rows and is stored compactly ([M / period * (period - skip), D])."""
    if x is None or x.dtype not in (F32, BF16):
        raise TypeError("x must be f32 or bf16")
    _chk(dy, BF16, "dy", 2), _chk(x, x.dtype, "x", 2), _chk(w, F32, "w", 1)
    defer = dw is None and db is None
    if defer:
        if partial_ws is None:
            raise ValueError("layernorm_bwd: deferred reduction (dw = db = None) needs its own partial_ws")
    else:
        _chk(dw, F32, "dw", 1), _chk(db, F32, "db", 1)
    D = x.shape[1]
    M = dy.shape[0] if M is None else M
    rows_in = M if rows_in is None else rows_in
    rows_out = rows_in if rows_out is None else rows_out
    dres_f = dres_b = None
    if dres is not None:
        if dres.dtype == BF16:
            dres_b = _chk(dres, BF16, "dres", 2)
        else:
            dres_f = _chk(dres, F32, "dres", 2)
        if dres_rows is None and dres.shape != x.shape:
            raise ValueError("dres shape")
    if dx is None and dxb is None:
        raise ValueError("layernorm_bwd: need dx and/or dxb")
    if dx is not None:
        _chk(dx, F32, "dx", 2)
        if dx.shape != x.shape:
            raise ValueError("dx shape")
    if dxb is not None:
        _chk(dxb, BF16, "dxb", 2)
        if dxb.shape != x.shape:
            raise ValueError("dxb shape")
    if dy.shape[1] != D or (not defer and (dw.numel() != D or db.numel() != D)) or dy.shape[0] < M:
        raise ValueError("layernorm_bwd: shape mismatch")
    if M % rows_in or (M // rows_in - 1) * rows_out + row_off + rows_in > x.shape[0]:
        raise ValueError("layernorm_bwd: row map exceeds x")
    if partial_ws is not None:
        _chk(partial_ws, F32, "partial_ws", 1)
        if partial_ws.numel() < 2 * layernorm_bwd_blocks(M) * D:      # [blocks][2][D]; at most 1024 blocks
            raise ValueError("partial_ws must hold 2 * layernorm_bwd_blocks(M) * D floats (2*1024*D always suffices)")
    period, skip = (0, 0) if dres_rows is None else (int(dres_rows[0]), int(dres_rows[1]))
    if dres_rows is not None:
        if dres is None or not 0 <= skip < period or M % period or rows_in != M or rows_out != M or row_off or tuple(dres.shape) != (M // period * (period - skip), D):
            raise ValueError("layernorm_bwd: dres_rows = (period, skip) needs dres [M / period * (period - skip), D] and no row map")
    xb = 1 if x.dtype == BF16 else 0
    bytes_ = ((4.0 if xb else 6.0) + (4.0 if dres_f is not None else 0.0) + (2.0 if dres_b is not None else 0.0) + (4.0 if dx is not None else 0.0)
              + (2.0 if dxb is not None else 0.0)) * M * D
    _run("mofo_layernorm_bwd_partial_res", ("ln_bwd",), bytes_,
         _p(dy), _ld(dy), _p(x), xb, _ld(x), _p(w), _p(mean), _p(rstd), _p(dres_f), _ld(dres_f) if dres_f is not None else 0, M, D,
         rows_in, rows_out, row_off, _p(dx), _ld(dx) if dx is not None else 0, _p(dxb), _ld(dxb) if dxb is not None else 0, _p(dw), _p(db),
         _p(dres_b), _ld(dres_b) if dres_b is not None else 0, _p(partial_ws), period, skip)
    return _lib.load().mofo_layernorm_bwd_blocks(M) if defer else None


def layernorm_bwd_finalize(items):
    """items = [(partial_ws, nblocks, D, dw, db), ...] (at most 40) left behind by deferred layernorm_bwd calls: one launch
    adds every LayerNorm's block partials to its dw / db"""
    n = len(items)
    if not 1 <= n <= 40:
        raise ValueError("layernorm_bwd_finalize: 1..40 items")
    for ws, nb, D, dw, db in items:
        _chk(ws, F32, "partial_ws", 1), _chk(dw, F32, "dw", 1), _chk(db, F32, "db", 1)
        if dw.numel() != D or db.numel() != D or ws.numel() < 2 * nb * D or not 1 <= nb <= 1024:
            raise ValueError("layernorm_bwd_finalize: item shape mismatch")
    vp, ci = C.c_void_p * n, C.c_int * n
    args = (vp(*[t[0].data_ptr() for t in items]), ci(*[t[1] for t in items]), ci(*[t[2] for t in items]),
            vp(*[t[3].data_ptr() for t in items]), vp(*[t[4].data_ptr() for t in items]))
    _run("mofo_layernorm_bwd_finalize", ("ln_bwd_fin",), sum(8.0 * t[1] * t[2] for t in items), *args, n)


def attention_fwd(qkv, B, N, H, scale, out, lse2, q_begin=0, out8=None, q_scale=None, q_amax=None):
    """``q_begin`` > 0: only the query rows q_begin .. N - 1 of every clip; ``out`` is then the compact [B * (N - q_begin), H * 64].
    ``out8`` (e4m3, shape of ``out``) + ``q_scale`` (f32 [1]) + ``q_amax`` (the stripes of one site): the rows also go out as
    sat(O * q_scale[0]) for an fp8 proj GEMM (mofo_attention_fwd_q8)"""
    _chk(qkv, BF16, "qkv", 2), _chk(out, BF16, "out", 2), _chk(lse2, F32, "lse2")
    if not 0 <= q_begin < N:
        raise ValueError("attention_fwd: q_begin out of range")
    if qkv.shape != (B * N, 3 * H * 64) or out.shape != (B * (N - q_begin), H * 64) or lse2.numel() != B * H * N:
        raise ValueError("attention_fwd: shape mismatch")
    if out8 is not None:
        _chk(out8, F8, "out8", 2), _chk(q_scale, F32, "q_scale"), _chk(q_amax, F32, "q_amax")
        if out8.shape != out.shape or q_amax.numel() != FP8_AMAX_STRIPES or not q_amax.is_contiguous():
            raise ValueError("attention_fwd: out8 must have out's shape, q_amax the %d stripes of one site" % FP8_AMAX_STRIPES)
        if _ld(out8) % 8 or out8.data_ptr() % 8:
            raise ValueError("attention_fwd: out8 rows must be 8-byte aligned (row pitch a multiple of 8 bytes)")
        _run("mofo_attention_fwd_q8", ("attn_fwd",), 4.0 * B * H * (N - q_begin) * N * 64, _p(qkv), _ld(qkv), B, N, H, scale, q_begin, _p(out), _ld(out),
             _p(lse2), _p(out8), _ld(out8), _p(q_scale), _p(q_amax))
        return out
    _run("mofo_attention_fwd_range", ("attn_fwd",), 4.0 * B * H * (N - q_begin) * N * 64, _p(qkv), _ld(qkv), B, N, H, scale, q_begin, _p(out), _ld(out),
         _p(lse2))
    return out


def attention_bwd(qkv, out, dout, lse2, B, N, H, scale, dqkv, delta):
    _chk(qkv, BF16, "qkv", 2), _chk(out, BF16, "out", 2), _chk(dout, BF16, "dout", 2), _chk(dqkv, BF16, "dqkv", 2)
    _chk(lse2, F32, "lse2"), _chk(delta, F32, "delta")
    if (qkv.shape != (B * N, 3 * H * 64) or dqkv.shape != qkv.shape or out.shape != (B * N, H * 64) or dout.shape != out.shape
            or lse2.numel() != B * H * N or delta.numel() != B * H * N):
        raise ValueError("attention_bwd: shape mismatch")
    _run("mofo_attention_bwd", ("attn_bwd",), 8.0 * B * H * N * N * 64, _p(qkv), _ld(qkv), _p(out), _ld(out), _p(dout), _ld(dout),
         _p(lse2), B, N, H, scale, _p(dqkv), _ld(dqkv), _p(delta))
    return dqkv


def ingest_u8(frames, clips):
    """frames uint8 [B,H,W,T*3] (reference Stack() layout) -> clips f32 [B,3,T,H,W], normalised as the reference does"""
    _chk(frames, U8, "frames", 4), _chk(clips, F32, "clips", 5)
    B, H, W, TC = frames.shape
    if not frames.is_contiguous() or not clips.is_contiguous() or TC % 3 or clips.shape != (B, 3, TC // 3, H, W):
        raise ValueError("ingest_u8: frames [B,H,W,T*3] contiguous, clips [B,3,T,H,W] contiguous")
    _run("mofo_ingest_u8", ("ingest_u8",), 5.0 * frames.numel(), _p(frames), B, TC // 3, H, W, _p(clips))
    return clips


def _attn_bwd_chk(qkv, out, dout, lse2, B, N, H, dqkv, delta, q_begin=0):
    _chk(qkv, BF16, "qkv", 2), _chk(dout, BF16, "dout", 2), _chk(dqkv, BF16, "dqkv", 2), _chk(lse2, F32, "lse2"), _chk(delta, F32, "delta")
    if not 0 <= q_begin < N:
        raise ValueError("attention backward: q_begin out of range")
    if out is not None:
        _chk(out, BF16, "out", 2)
        if out.shape != (B * (N - q_begin), H * 64):
            raise ValueError("out shape")
    if (qkv.shape != (B * N, 3 * H * 64) or dqkv.shape != qkv.shape or dout.shape != (B * (N - q_begin), H * 64)
            or lse2.numel() != B * H * N or delta.numel() != B * H * N):
        raise ValueError("attention backward: shape mismatch")


def attention_delta(out, dout, B, N, H, delta, q_begin=0):
    """``q_begin`` > 0 (here and in the two passes below): out / dout hold the query rows q_begin .. N - 1 of every clip compactly"""
    _chk(out, BF16, "out", 2), _chk(dout, BF16, "dout", 2), _chk(delta, F32, "delta")
    if not 0 <= q_begin < N or out.shape != (B * (N - q_begin), H * 64) or dout.shape != out.shape or delta.numel() != B * H * N:
        raise ValueError("attention_delta: shape mismatch")
    _run("mofo_attention_delta_range", ("attn_delta",), 4.0 * out.numel(), _p(out), _ld(out), _p(dout), _ld(dout), B, N, H, q_begin, _p(delta))


def attention_bwd_dq(qkv, dout, lse2, delta, B, N, H, scale, dqkv, q_begin=0):
    """dq rows below q_begin are NOT written (the caller clears them)"""
    _attn_bwd_chk(qkv, None, dout, lse2, B, N, H, dqkv, delta, q_begin)
    _run("mofo_attention_bwd_dq_range", ("attn_bwd_dq",), 4.0 * B * H * (N - q_begin) * N * 64, _p(qkv), _ld(qkv), _p(dout), _ld(dout), _p(lse2), _p(delta),
         B, N, H, scale, q_begin, _p(dqkv), _ld(dqkv))


def attention_bwd_dq_delta(qkv, out, dout, lse2, delta, B, N, H, scale, dqkv, q_begin=0):
    """the dQ pass that computes delta = rowsum(dO * O) itself and WRITES it (rows q_begin .. N - 1) for attention_bwd_dkv, which must
    follow on the same stream; attention_delta is then not needed"""
    _attn_bwd_chk(qkv, out, dout, lse2, B, N, H, dqkv, delta, q_begin)
    _run("mofo_attention_bwd_dq_delta_range", ("attn_bwd_dq",), 4.0 * B * H * (N - q_begin) * N * 64, _p(qkv), _ld(qkv), _p(out), _ld(out), _p(dout), _ld(dout),
         _p(lse2), _p(delta), B, N, H, scale, q_begin, _p(dqkv), _ld(dqkv))


def attention_bwd_dkv(qkv, dout, lse2, delta, B, N, H, scale, dqkv, q_begin=0):
    _attn_bwd_chk(qkv, None, dout, lse2, B, N, H, dqkv, delta, q_begin)
    _run("mofo_attention_bwd_dkv_range", ("attn_bwd_dkv",), 4.0 * B * H * (N - q_begin) * N * 64, _p(qkv), _ld(qkv), _p(dout), _ld(dout), _p(lse2), _p(delta),
         B, N, H, scale, q_begin, _p(dqkv), _ld(dqkv))


def mask_to_indices(mask_u8, n_vis, vis_idx, msk_idx, status):
    _chk(mask_u8, U8, "mask", 2), _chk(vis_idx, I32, "vis_idx", 2), _chk(msk_idx, I32, "msk_idx", 2), _chk(status, I32, "status")
    B, N = mask_u8.shape
    if not mask_u8.is_contiguous() or vis_idx.shape != (B, n_vis) or msk_idx.shape != (B, N - n_vis):
        raise ValueError("mask_to_indices: shape mismatch")
    _run("mofo_mask_to_indices", ("mask_idx",), 5.0 * B * N, _p(mask_u8), B, N, n_vis, _p(vis_idx), _p(msk_idx), _p(status))


def tube_masks(seed, counter, frames, patches_per_frame, n_mask, mask_u8):
    """device-side tube masks (include/mofo_hip.h: mofo_tube_masks): fills mask_u8 [B, frames * patches_per_frame]"""
    _chk(mask_u8, U8, "mask", 2)
    B, N = mask_u8.shape
    if not mask_u8.is_contiguous() or N != frames * patches_per_frame:
        raise ValueError("tube_masks: mask must be contiguous [B, frames * patches_per_frame]")
    _run("mofo_tube_masks", ("tube_masks",), 1.0 * B * N, int(seed) & 0xFFFFFFFF, int(counter) & 0xFFFFFFFF, B, frames, patches_per_frame, n_mask,
         _p(mask_u8))


def patch_gather(clips, pt, p, tok_idx, out):
    _chk(clips, F32, "clips", 5), _chk(tok_idx, I32, "tok_idx", 2), _chk(out, BF16, "out", 2)
    B, Cc, T, H, W = clips.shape
    if not clips.is_contiguous() or not tok_idx.is_contiguous() or tok_idx.shape[0] != B:
        raise ValueError("patch_gather: clips / tok_idx must be contiguous with matching batch")
    n_tok = tok_idx.shape[1]
    if out.shape != (B * n_tok, Cc * pt * p * p):
        raise ValueError("patch_gather: out shape")
    _run("mofo_patch_gather", ("patch_gather",), 6.0 * B * n_tok * Cc * pt * p * p, _p(clips), B, Cc, T, H, W, pt, p, _p(tok_idx), n_tok,
         _p(out), _ld(out))
    return out


def patch_gather_u8(frames, pt, p, tok_idx, out):
    """patch_gather straight from the uint8 frame stack [B,H,W,T*3] (normalised on the fly, bit-identical to ingest_u8 + patch_gather)"""
    _chk(frames, U8, "frames", 4), _chk(tok_idx, I32, "tok_idx", 2), _chk(out, BF16, "out", 2)
    B, H, W, TC = frames.shape
    if not frames.is_contiguous() or not tok_idx.is_contiguous() or tok_idx.shape[0] != B or TC % 3:
        raise ValueError("patch_gather_u8: frames [B,H,W,T*3] / tok_idx must be contiguous with matching batch")
    n_tok = tok_idx.shape[1]
    if out.shape != (B * n_tok, 3 * pt * p * p):
        raise ValueError("patch_gather_u8: out shape")
    _run("mofo_patch_gather_u8", ("patch_gather",), 3.0 * B * n_tok * 3 * pt * p * p, _p(frames), B, TC // 3, H, W, pt, p, _p(tok_idx), n_tok,
         _p(out), _ld(out))
    return out


def fill_mask_tokens(mask_token, pos, msk_idx, n_vis, x_full):
    if x_full is None or x_full.dtype not in (F32, BF16):
        raise TypeError("x_full must be f32 or bf16")
    _chk(mask_token, F32, "mask_token"), _chk(pos, F32, "pos", 2), _chk(msk_idx, I32, "msk_idx", 2), _chk(x_full, x_full.dtype, "x_full", 3)
    B, N, D = x_full.shape
    if not x_full.is_contiguous() or mask_token.numel() != D or pos.shape[1] != D or msk_idx.shape != (B, N - n_vis) or pos.shape[0] < N:
        raise ValueError("fill_mask_tokens: shape mismatch")
    xb = 1 if x_full.dtype == BF16 else 0
    _run("mofo_fill_mask_tokens", ("fill_mask",), (6.0 if xb else 8.0) * B * (N - n_vis) * D, _p(mask_token), _p(pos), _ld(pos), _p(msk_idx), B, N, n_vis, D,
         _p(x_full), xb)


def dec0_inverse(msk_idx, N, inv):
    """inv[b, j] = slot of position j in clip b's ascending masked list (msk_idx [B, n_msk]) or -1"""
    _chk(msk_idx, I32, "msk_idx", 2), _chk(inv, I32, "inv", 2)
    B, n_msk = msk_idx.shape
    if not msk_idx.is_contiguous() or not inv.is_contiguous() or inv.shape != (B, N) or not 0 < n_msk < N:
        raise ValueError("dec0_inverse: msk_idx [B, n_msk], inv [B, N]")
    _run("mofo_dec0_inverse", ("dec0_idx",), 8.0 * B * N, _p(msk_idx), B, N, N - n_msk, _p(inv))


def dec0_gather(cat, msk_idx, N, full):
    """full[b, r] = r < n_vis ? cat[b * n_vis + r] : cat[B * n_vis + msk_idx[b, r - n_vis]] (bf16 rows; include/mofo_hip.h)"""
    _chk(cat, BF16, "cat", 2), _chk(full, BF16, "full", 2), _chk(msk_idx, I32, "msk_idx", 2)
    B, n_msk = msk_idx.shape
    n_vis = N - n_msk
    if cat.shape[0] != B * n_vis + N or full.shape != (B * N, cat.shape[1]) or not msk_idx.is_contiguous():
        raise ValueError("dec0_gather: cat [B * n_vis + N, W], full [B * N, W]")
    _run("mofo_dec0_gather", ("dec0_gather",), 4.0 * full.numel(), _p(cat), _ld(cat), _p(msk_idx), B, N, n_vis, cat.shape[1], _p(full), _ld(full))
    return full


def dec0_reduce(full, inv, n_vis, cat):
    """adjoint of dec0_gather: visible rows copied, position rows summed over the clips that mask the position"""
    _chk(cat, BF16, "cat", 2), _chk(full, BF16, "full", 2), _chk(inv, I32, "inv", 2)
    B, N = inv.shape
    if cat.shape[0] != B * n_vis + N or full.shape != (B * N, cat.shape[1]) or not inv.is_contiguous():
        raise ValueError("dec0_reduce: cat [B * n_vis + N, W], full [B * N, W]")
    _run("mofo_dec0_reduce", ("dec0_reduce",), 2.0 * full.numel() + 2.0 * cat.numel(), _p(full), _ld(full), _p(inv), B, N, n_vis, cat.shape[1], _p(cat), _ld(cat))
    return cat


def assemble_bwd_blocks(B, N):
    return _lib.load().mofo_assemble_bwd_blocks(B, N)


def assemble_bwd_finalize(partial_ws, B, N, d_mask_token):
    """add the block partials a deferred assemble_bwd (d_mask_token=None) left in partial_ws to d_mask_token"""
    _chk(partial_ws, F32, "partial_ws"), _chk(d_mask_token, F32, "d_mask_token")
    D = d_mask_token.numel()
    if partial_ws.numel() < assemble_bwd_blocks(B, N) * D:
        raise ValueError("assemble_bwd_finalize: partial_ws needs assemble_bwd_blocks(B, N) * D floats")
    _run("mofo_assemble_bwd_finalize", ("assemble_bwd_fin",), 4.0 * assemble_bwd_blocks(B, N) * D, _p(partial_ws), B, N, D, _p(d_mask_token))


def assemble_bwd(dx_full, n_vis, d_e2d, d_mask_token, partial_ws=None):
    """``d_mask_token`` None (with ``partial_ws``): deferred -- only the block partials are written, assemble_bwd_finalize adds them"""
    if dx_full is None or dx_full.dtype not in (F32, BF16):
        raise TypeError("dx_full must be f32 or bf16")
    _chk(dx_full, dx_full.dtype, "dx_full", 3), _chk(d_e2d, BF16, "d_e2d", 2)
    B, N, D = dx_full.shape
    if d_mask_token is not None:
        _chk(d_mask_token, F32, "d_mask_token")
    elif partial_ws is None or D % 8 or D > 512:
        raise ValueError("assemble_bwd: the deferred form needs partial_ws and D a multiple of 8, <= 512")
    if not dx_full.is_contiguous() or d_e2d.shape != (B * n_vis, D) or not d_e2d.is_contiguous() or (d_mask_token is not None and d_mask_token.numel() != D):
        raise ValueError("assemble_bwd: shape mismatch")
    isb = 1 if dx_full.dtype == BF16 else 0
    if partial_ws is not None:
        _chk(partial_ws, F32, "partial_ws")
        if partial_ws.numel() < assemble_bwd_blocks(B, N) * D:
            raise ValueError("assemble_bwd: partial_ws needs assemble_bwd_blocks(B, N) * D floats")
    _run("mofo_assemble_bwd", ("assemble_bwd",), (2.0 if isb else 4.0) * B * N * D, _p(dx_full), isb, B, N, n_vis, D, _p(d_e2d), _p(d_mask_token),
         _p(partial_ws))


def target_mse(clips, pt, p, msk_idx, pred, normalize, grad_scale, row_loss, loss, dpred=None, target_out=None):
    _chk(clips, F32, "clips", 5), _chk(msk_idx, I32, "msk_idx", 2), _chk(pred, BF16, "pred", 2), _chk(row_loss, F32, "row_loss"), _chk(loss, F32, "loss")
    B, Cc, T, H, W = clips.shape
    n_msk = msk_idx.shape[1]
    L = Cc * pt * p * p
    if not clips.is_contiguous() or not msk_idx.is_contiguous() or msk_idx.shape[0] != B or pred.shape != (B * n_msk, L) or row_loss.numel() < B * n_msk:
        raise ValueError("target_mse: shape mismatch")
    if dpred is not None:
        _chk(dpred, BF16, "dpred", 2)
        if dpred.shape != pred.shape:
            raise ValueError("dpred shape")
    if target_out is not None:
        _chk(target_out, F32, "target_out", 2)
        if target_out.shape != pred.shape or not target_out.is_contiguous():
            raise ValueError("target_out shape")
    _run("mofo_target_mse", ("target_mse",), (4.0 + 2.0 + (2.0 if dpred is not None else 0.0)) * B * n_msk * L, _p(clips), B, Cc, T, H, W, pt, p,
         _p(msk_idx), n_msk, _p(pred), _ld(pred), 1 if normalize else 0, grad_scale, _p(row_loss), _p(loss), _p(dpred),
         _ld(dpred) if dpred is not None else 0, _p(target_out))
    return loss


def target_mse_u8(frames, pt, p, msk_idx, pred, normalize, grad_scale, row_loss, loss, dpred=None):
    """target_mse straight from the uint8 frame stack [B,H,W,T*3] (bit-identical to ingest_u8 + target_mse)"""
    _chk(frames, U8, "frames", 4), _chk(msk_idx, I32, "msk_idx", 2), _chk(pred, BF16, "pred", 2), _chk(row_loss, F32, "row_loss"), _chk(loss, F32, "loss")
    B, H, W, TC = frames.shape
    n_msk = msk_idx.shape[1]
    L = 3 * pt * p * p
    if (not frames.is_contiguous() or TC % 3 or not msk_idx.is_contiguous() or msk_idx.shape[0] != B or pred.shape != (B * n_msk, L)
            or row_loss.numel() < B * n_msk):
        raise ValueError("target_mse_u8: shape mismatch")
    if dpred is not None:
        _chk(dpred, BF16, "dpred", 2)
        if dpred.shape != pred.shape:
            raise ValueError("dpred shape")
    _run("mofo_target_mse_u8", ("target_mse",), (1.0 + 2.0 + (2.0 if dpred is not None else 0.0)) * B * n_msk * L, _p(frames), B, TC // 3, H, W, pt, p,
         _p(msk_idx), n_msk, _p(pred), _ld(pred), 1 if normalize else 0, grad_scale, _p(row_loss), _p(loss), _p(dpred),
         _ld(dpred) if dpred is not None else 0)
    return loss


def reconstruct(clips, pt, p, msk_idx, pred, rec, masked=None, ori=None):
    """run_videomae_vis.py:150-180 (see include/mofo_hip.h): pred bf16 or f32 [B*n_msk, 1536]"""
    _chk(clips, F32, "clips", 5), _chk(msk_idx, I32, "msk_idx", 2), _chk(rec, F32, "rec", 5)
    B, Cc, T, H, W = clips.shape
    n_msk = msk_idx.shape[1]
    L = Cc * pt * p * p
    if pred.dtype not in (BF16, F32) or pred.dim() != 2 or pred.shape != (B * n_msk, L) or pred.device != clips.device:
        raise ValueError("reconstruct: pred must be bf16/f32 [B*n_msk, C*pt*p*p] on the clips' device")
    if not clips.is_contiguous() or not msk_idx.is_contiguous() or msk_idx.shape[0] != B:
        raise ValueError("reconstruct: shape mismatch")
    for t, nm in ((rec, "rec"), (masked, "masked"), (ori, "ori")):
        if t is not None and (t.dtype != F32 or t.shape != clips.shape or not t.is_contiguous() or t.device != clips.device):
            raise ValueError(f"reconstruct: {nm} must be a contiguous f32 tensor shaped like clips")
    nout = 1 + (masked is not None) + (ori is not None)
    _run("mofo_reconstruct", ("reconstruct",), 4.0 * clips.numel() * (1 + nout) + pred.numel() * pred.element_size(), _p(clips), B, Cc, T, H, W,
         pt, p, _p(msk_idx), n_msk, _p(pred), 1 if pred.dtype == BF16 else 0, _ld(pred), _p(rec), _p(masked), _p(ori))
    return rec


def token_mean_norm(x, B, N, w, b, eps, pooled_ws, out_f32, out_bf16=None):
    """modeling_finetune.py:403-405: fc_norm(x.mean(1)) over x f32 [B*N, D]"""
    _chk(x, F32, "x", 2), _chk(w, F32, "w", 1), _chk(b, F32, "b", 1), _chk(pooled_ws, F32, "pooled_ws"), _chk(out_f32, F32, "out_f32", 2)
    D = x.shape[1]
    if x.shape[0] != B * N or w.numel() != D or b.numel() != D or pooled_ws.numel() < B * D or out_f32.shape != (B, D) or not out_f32.is_contiguous():
        raise ValueError("token_mean_norm: shape mismatch")
    if out_bf16 is not None:
        _chk(out_bf16, BF16, "out_bf16", 2)
        if out_bf16.shape != (B, D) or not out_bf16.is_contiguous():
            raise ValueError("token_mean_norm: out_bf16 shape")
    _run("mofo_token_mean_norm", ("token_mean_norm",), 4.0 * x.numel(), _p(x), _ld(x), B, N, D, _p(w), _p(b), eps, _p(pooled_ws), _p(out_f32), _p(out_bf16))
    return out_f32


def sumsq_norm(g, partial, out_norm):
    _chk(g, F32, "g", 1), _chk(partial, F32, "partial", 1), _chk(out_norm, F32, "out_norm")
    if partial.numel() < 1024:
        raise ValueError("partial must hold 1024 floats")
    _run("mofo_sumsq", ("sumsq",), 4.0 * g.numel(), _p(g), g.numel(), _p(partial), _p(out_norm))
    return out_norm


def adamw_blocks(n):
    """block partials one adamw launch over n elements writes into norm_partial"""
    return int(_lib.load().mofo_adamw_blocks(int(n)))


def norm_finalize(partial, count, out_norm):
    """out_norm[0] = sqrt(sum(partial[:count])): closes a range-by-range adamw sequence"""
    _chk(partial, F32, "partial", 1), _chk(out_norm, F32, "out_norm")
    if count < 1 or partial.numel() < count:
        raise ValueError("norm_finalize: count")
    _run("mofo_norm_finalize", ("norm_fin",), 4.0 * count, _p(partial), int(count), _p(out_norm))


def adamw(p, g, m, v, p_bf16, chunk_group, lr0, wd0, lr1, wd1, beta1, beta2, eps, step, grad_norm=None, max_norm=0.0, grad_mult=1.0,
          norm_partial=None, norm_out=None, gate_finite=None, gate_zero=None, gate_one=None, q8=None):
    """``q8`` = (chunk_seg int16 [n / 1024], w_scale f32 [nseg], w_amax f32 [nseg], p_e4m3 [n]): the update also writes the e4m3
    shadow of the fp8 forward's weights with the delayed per-matrix scale (mofo_adamw_q8; ``fp8_roll_scales`` once per step first).
    ``norm_partial`` (f32 [>= 2048]) + ``norm_out`` (f32 [1]): also leave the global L2 norm of ``g`` in norm_out.
    ``gate_finite`` (f32 [1]) / ``gate_zero`` (i32 [1]) / ``gate_one`` (f32 [1]): device words the kernel checks before it touches
    anything -- the update is skipped unless the first is finite, the second 0 and the third 1.0 (mofo_adamw_gated)."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, F32, n, 1)
    _chk(chunk_group, U8, "chunk_group", 1)
    n = p.numel()
    if g.numel() != n or m.numel() != n or v.numel() != n or chunk_group.numel() * 1024 != n:
        raise ValueError("adamw: flat buffers must share one length = 1024 * len(chunk_group)")
    if p_bf16 is not None:
        _chk(p_bf16, BF16, "p_bf16", 1)
        if p_bf16.numel() != n:
            raise ValueError("p_bf16 length")
    if norm_out is not None and norm_partial is None:
        raise ValueError("adamw: norm_out needs the norm_partial scratch")
    if norm_partial is not None:     # with norm_out None the block partials are left for norm_finalize (range-by-range update)
        _chk(norm_partial, F32, "norm_partial", 1)
        if norm_partial.numel() < adamw_blocks(n):
            raise ValueError("norm_partial must hold adamw_blocks(n) floats (at most 2048)")
    if norm_out is not None:
        _chk(norm_out, F32, "norm_out")
    if gate_finite is not None:
        _chk(gate_finite, F32, "gate_finite")
    if gate_zero is not None:
        _chk(gate_zero, I32, "gate_zero")
    if gate_one is not None:
        _chk(gate_one, F32, "gate_one")
    if q8 is not None:
        chunk_seg, w_scale, w_amax, p8 = q8
        _chk(w_scale, F32, "w_scale", 1), _chk(w_amax, F32, "w_amax", 1), _chk(p8, F8, "p_e4m3", 1)
        if chunk_seg is None or chunk_seg.dtype != torch.int16 or not chunk_seg.is_cuda or chunk_seg.numel() * 1024 != n or p8.numel() != n or p_bf16 is None:
            raise ValueError("adamw q8: chunk_seg int16 [n / 1024] on the GPU, p_e4m3 [n], and the bf16 shadow")
        _run("mofo_adamw_q8", ("adamw",), 31.0 * n, _p(p), _p(g), _p(m), _p(v), _p(p_bf16), n, _p(chunk_group),
             lr0, wd0, lr1, wd1, beta1, beta2, eps, step, _p(grad_norm), max_norm, grad_mult, _p(norm_partial), _p(norm_out),
             _p(gate_finite), _p(gate_zero), _p(gate_one), _p(chunk_seg), _p(w_scale), _p(w_amax), _p(p8))
        return
    _run("mofo_adamw_gated", ("adamw",), (28.0 + (2.0 if p_bf16 is not None else 0.0)) * n, _p(p), _p(g), _p(m), _p(v), _p(p_bf16), n, _p(chunk_group),
         lr0, wd0, lr1, wd1, beta1, beta2, eps, step, _p(grad_norm), max_norm, grad_mult, _p(norm_partial), _p(norm_out),
         _p(gate_finite), _p(gate_zero), _p(gate_one))


def cast_bf16(src, dst):
    _chk(src, F32, "src", 1), _chk(dst, BF16, "dst", 1)
    if src.numel() != dst.numel():
        raise ValueError("cast_bf16: length mismatch")
    _run("mofo_cast_bf16", ("cast",), 6.0 * src.numel(), _p(src), _p(dst), src.numel())
    return dst


def cast_f32(src, dst):
    """bf16 -> f32 (exact): the receiving end of the bf16 gradient transport (dist.GradSync, MOFO_GRAD_BF16=1)"""
    _chk(src, BF16, "src", 1), _chk(dst, F32, "dst", 1)
    if src.numel() != dst.numel():
        raise ValueError("cast_f32: length mismatch")
    _run("mofo_cast_f32", ("cast",), 6.0 * src.numel(), _p(src), _p(dst), src.numel())
    return dst


def zero_chunks(buf, chunk_ids):
    """zero the listed 1024-element chunks of a flat f32 buffer (chunk_ids: int32 GPU tensor)"""
    _chk(buf, F32, "buf", 1), _chk(chunk_ids, I32, "chunk_ids", 1)
    if buf.numel() % 1024:
        raise ValueError("zero_chunks: buffer of whole 1024-element chunks")
    if chunk_ids.numel() == 0:
        return              # nothing accumulates: every gradient is overwritten by plain stores
    _run("mofo_zero_chunks", ("zero_grad",), 4096.0 * chunk_ids.numel(), _p(buf), _p(chunk_ids), chunk_ids.numel())

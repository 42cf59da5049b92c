"""ctypes binding of libmofo_hip.so (the C-ABI declared in include/mofo_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails, this raises.  Import torch first so
that the HIP runtime this library binds to (libamdhip64.so.7) is the one torch already loaded.
"""
import ctypes as C
import os

import torch  # noqa: F401  (loads libamdhip64 before our library)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmofo_hip.so")
# measurement only (tools/attn_ab.py, same-box A/B of two builds): MOFO_HIP_LIB points the loader at another build of the library
if os.environ.get("MOFO_HIP_LIB"):
    LIB_PATH = os.path.abspath(os.environ["MOFO_HIP_LIB"])

# enums (mirror include/mofo_hip.h)
GEMM_NT, GEMM_NN, GEMM_TN, GEMM_NT_FP8 = 0, 1, 2, 3
EPI_BF16, EPI_BIAS_GELU, EPI_RESID_F32, EPI_POS_F32, EPI_DGELU_BF16, EPI_F32, EPI_RESID_BF16, EPI_POS_BF16 = 0, 1, 2, 3, 4, 5, 6, 7

_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong


class GemmArgs(C.Structure):
    _fields_ = [("op", _i), ("epilogue", _i), ("M", _i), ("N", _i), ("K", _i),
                ("A", _vp), ("lda", _i), ("B", _vp), ("ldb", _i), ("C", _vp), ("ldc", _i), ("C2", _vp), ("ldc2", _i),
                ("bias", _vp), ("resid", _vp), ("ldr", _i), ("aux", _vp), ("ldaux", _i),
                ("pos", _vp), ("ldpos", _i), ("row_idx", _vp), ("rows_in", _i), ("rows_out", _i), ("row_off", _i),
                ("splits", _i), ("accumulate", _i), ("colsum", _vp), ("colsum_skip_lo", _i), ("colsum_skip_hi", _i),
                ("a_scale_inv", _vp), ("b_scale_inv", _vp), ("C8", _vp), ("ldc8", _i), ("q_scale", _vp), ("q_amax", _vp)]


_SIGS = {
    "mofo_version": (_i, []),
    "mofo_last_error": (C.c_char_p, []),
    "mofo_gemm": (_i, [C.POINTER(GemmArgs), _vp]),
    "mofo_gemm_grouped": (_i, [C.POINTER(GemmArgs), _i, _vp]),
    "mofo_gemm_grouped_plan": (_i, [C.POINTER(GemmArgs), _i, C.POINTER(_i)]),
    "mofo_gemm_route_counts": (_i, [C.POINTER(_ll), _i, _i]),
    "mofo_gemm_wgrad_sliced_ws": (_ll, [C.POINTER(GemmArgs), _i, _i]),
    "mofo_gemm_wgrad_sliced": (_i, [C.POINTER(GemmArgs), _i, _i, _vp, _ll, _vp]),
    "mofo_colsum_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mofo_layernorm_fwd": (_i, [_vp, _i, _i, _vp, _vp, _f, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "mofo_layernorm_fwd_q": (_i, [_vp, _i, _i, _vp, _vp, _f, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "mofo_fp8_quantize_segments": (_i, [_vp, _ll, _vp, _i, _vp, _vp, _vp, _vp]),
    "mofo_fp8_quantize_bf16": (_i, [_vp, _ll, _vp, _vp, _vp, _vp]),
    "mofo_fp8_update_scales": (_i, [_vp, _vp, _i, _f, _vp]),
    "mofo_layernorm_bwd": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "mofo_layernorm_bwd_partial_res": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "mofo_layernorm_bwd_blocks": (_i, [_i]),
    "mofo_layernorm_bwd_finalize": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mofo_attention_fwd": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp]),
    "mofo_attention_bwd": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _f, _vp, _i, _vp, _vp]),
    "mofo_ingest_u8": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "mofo_attention_delta": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mofo_attention_bwd_dq": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _f, _vp, _i, _vp]),
    "mofo_attention_bwd_dkv": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _f, _vp, _i, _vp]),
    "mofo_attention_fwd_range": (_i, [_vp, _i, _i, _i, _i, _f, _i, _vp, _i, _vp, _vp]),
    "mofo_attention_fwd_q8": (_i, [_vp, _i, _i, _i, _i, _f, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "mofo_attention_delta_range": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mofo_attention_bwd_dq_range": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _f, _i, _vp, _i, _vp]),
    "mofo_attention_bwd_dq_delta_range": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _f, _i, _vp, _i, _vp]),
    "mofo_attention_bwd_dkv_range": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _f, _i, _vp, _i, _vp]),
    "mofo_mask_to_indices": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mofo_tube_masks": (_i, [C.c_uint, C.c_uint, _i, _i, _i, _i, _vp, _vp]),
    "mofo_patch_gather": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "mofo_fill_mask_tokens": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "mofo_dec0_inverse": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mofo_dec0_gather": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "mofo_dec0_reduce": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "mofo_assemble_bwd_blocks": (_i, [_i, _i]),
    "mofo_assemble_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mofo_assemble_bwd_finalize": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mofo_patch_gather_u8": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "mofo_target_mse_u8": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _f, _vp, _vp, _vp, _i, _vp]),
    "mofo_target_mse": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _f, _vp, _vp, _vp, _i, _vp, _vp]),
    "mofo_reconstruct": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "mofo_token_mean_norm": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "mofo_sumsq": (_i, [_vp, _ll, _vp, _vp, _vp]),
    "mofo_adamw": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _vp, _f, _f, _f, _f, _f, _f, _f, _i, _vp, _f, _f, _vp, _vp, _vp]),
    "mofo_adamw_gated": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _vp, _f, _f, _f, _f, _f, _f, _f, _i, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mofo_adamw_q8": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _vp, _f, _f, _f, _f, _f, _f, _f, _i, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mofo_fp8_roll_scales": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "mofo_adamw_blocks": (_i, [_ll]),
    "mofo_norm_finalize": (_i, [_vp, _i, _vp, _vp]),
    "mofo_cast_bf16": (_i, [_vp, _vp, _ll, _vp]),
    "mofo_cast_f32": (_i, [_vp, _vp, _ll, _vp]),
    "mofo_zero_chunks": (_i, [_vp, _vp, _i, _vp]),
    "mofo_comm_unique_id": (_i, [_vp]),
    "mofo_comm_init": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "mofo_comm_allreduce_f32": (_i, [_vp, _vp, _ll, _vp]),
    "mofo_comm_destroy": (_i, [_vp]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """Return the loaded library; raise (never fall back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(mofo_amd has no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.mofo_version() != 4:
            raise RuntimeError("libmofo_hip.so ABI version mismatch")
        _lib = lib
    return _lib


class EventProfiler:
    """Per-kernel-class timing with HIP events recorded on the launch stream (torch's current stream), used by bench.py.
    ``begin(key, work)`` / ``end()`` bracket one C-ABI call; ``summary()`` joins after a device synchronize."""

    def __init__(self, only=None):
        self.records = []   # (key, work, start_event, end_event)
        self._cur = None
        self.only = only    # bracket just this kernel class (each event costs ~5 us of host time)
        self.enabled = True  # bench.py switches the bracketing off on some timed steps when the class has many launches per step

    def begin(self, key, work=0.0, stream=None):
        if (self.only is not None and key != self.only) or not self.enabled:
            self._cur = None
            return
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record(stream) if stream is not None else e0.record()
        # ``work`` is FLOPs (MFMA classes) or bytes (HBM classes); the GEMM wrappers pass (FLOPs, algorithmic bytes)
        nbytes = 0.0
        if isinstance(work, tuple):
            work, nbytes = work
        self._cur = (key, work, e0, nbytes)

    def end(self, stream=None):
        if self._cur is None:
            return
        key, work, e0, nbytes = self._cur
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(stream) if stream is not None else e1.record()
        self.records.append((key, work, e0, e1, nbytes))

    def summary(self):
        """per class: launches, total ms, total work, max ms.  A launch that took more than 10x its class's 90th percentile
        (a one-off stall inside the bracket; classes are multi-modal, e.g. encoder vs decoder shapes, hence not the median)
        is left out and counted in 'dropped'."""
        torch.cuda.synchronize()
        per = {}
        for key, work, e0, e1, nbytes in self.records:
            per.setdefault(key, []).append((e0.elapsed_time(e1), work, nbytes))
        out = {}
        for key, lst in per.items():
            ts = sorted(t for t, _, _ in lst)
            p90 = ts[min(len(ts) - 1, (9 * len(ts)) // 10)]
            keep = [(t, w, b) for t, w, b in lst if t <= 10.0 * p90 or len(lst) < 20]
            out[key] = {"launches": len(keep), "ms": sum(t for t, _, _ in keep), "work": sum(w for _, w, _ in keep),
                        "bytes": sum(b for _, _, b in keep), "max_ms": max(t for t, _, _ in keep), "dropped": len(lst) - len(keep)}
        return out


PROFILER = None   # set to an EventProfiler by bench.py
RECORDER = None   # a list while the runtime records a launch sequence (ops._run / ops.host_op append to it)


def check(rc, what=""):
    if rc != 0:
        msg = load().mofo_last_error().decode(errors="replace")
        raise RuntimeError(f"libmofo_hip {what} failed with code {rc}: {msg}")

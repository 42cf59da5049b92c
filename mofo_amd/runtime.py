"""Host-side runtime of the pretraining hot path: flat parameter storage, a per-batch-size workspace, and the fixed
sequence of C-ABI kernel launches that makes up forward and backward.

Design (MI355X-first, not a translation of the reference's autograd graph):
  * every parameter lives in ONE flat fp32 buffer (+ flat fp32 grads, + a flat bf16 shadow that the MFMA GEMMs read);
    torch Parameters are views into it, so the optimizer is one fused kernel over the buffer and the data-parallel
    all-reduce runs on contiguous ranges of the gradient buffer with no bucket copies;
  * all activations of a given batch size are preallocated once (288 GB of HBM: nothing is recomputed), so a step is a
    static list of launches -- replayable without Python-side allocation and capturable into a hipGraph;
  * the residual stream is fp32, GEMM operands are bf16, LayerNorm / softmax statistics / the loss are fp32.
Reference arithmetic: modeling_pretrain.py:83-101,152-161,253-266; modeling_finetune.py:44-51,78-98,216-223,242-248;
engine_for_pretraining.py:43-67.
"""
import math
import os
from dataclasses import dataclass
from types import SimpleNamespace as NS
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32
CHUNK = 1024


@dataclass(frozen=True)
class Dims:
    img_size: int = 224
    patch_size: int = 16
    tubelet: int = 2
    num_frames: int = 16
    in_chans: int = 3
    enc_dim: int = 768
    enc_depth: int = 12
    enc_heads: int = 12
    dec_dim: int = 384
    dec_depth: int = 4
    dec_heads: int = 6
    mlp_ratio: float = 4.0
    eps: float = 1e-6
    patch_out: int = 1536  # decoder_num_classes

    @property
    def grid(self):
        g = self.img_size // self.patch_size
        return (self.num_frames // self.tubelet, g, g)

    @property
    def num_patches(self):
        t, h, w = self.grid
        return t * h * w

    @property
    def patch_dim(self):
        return self.in_chans * self.tubelet * self.patch_size ** 2


def sincos_table(n_pos: int, dim: int) -> torch.Tensor:
    """modeling_finetune.py:252-262 (float64 angles, sin on even / cos on odd columns, cast to f32); [n_pos, dim]."""
    j = np.arange(dim)
    ang = np.arange(n_pos, dtype=np.float64)[:, None] / np.power(10000, 2 * (j // 2) / dim)[None, :]
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.from_numpy(ang.astype(np.float32))


def _is_no_decay(name: str, shape, skip) -> bool:
    """optim_factory.py:56-61"""
    return len(shape) == 1 or name.endswith(".bias") or name in skip


def _align(n: int, a: int = CHUNK) -> int:
    return (n + a - 1) // a * a


# ======================================================================================================= flat store
class FlatStore:
    """All parameters of a module tree in one fp32 buffer; Parameters become views (data AND grad)."""

    def __init__(self, named_params: List[Tuple[str, torch.nn.Parameter]], device, skip_decay=()):
        self.device = device
        self.names = [n for n, _ in named_params]
        self.offset: Dict[str, int] = {}
        self.shape: Dict[str, Tuple[int, ...]] = {}
        plist = dict(named_params)
        groups = []  # (offset, length, no_decay)
        off = 0
        done = set()
        for name, p in named_params:
            if name in done:
                continue
            if name.endswith("attn.q_bias"):
                # fused qkv bias [3D] = (q_bias | zeros | v_bias): the reference concatenates exactly this every forward
                # (modeling_finetune.py:82); the zero third is not a parameter and never moves (its gradient stays 0).
                vname = name[: -len("q_bias")] + "v_bias"
                d = p.numel()
                self.offset[name], self.offset[vname] = off, off + 2 * d
                self.shape[name], self.shape[vname] = tuple(p.shape), tuple(plist[vname].shape)
                groups.append((off, 3 * d, True))
                done.update((name, vname))
                off = _align(off + 3 * d)
                continue
            self.offset[name] = off
            self.shape[name] = tuple(p.shape)
            groups.append((off, p.numel(), _is_no_decay(name, p.shape, skip_decay)))
            done.add(name)
            off = _align(off + p.numel())
        self.total = off
        self.params = torch.zeros(self.total, dtype=F32, device=device)
        self.grads = torch.zeros(self.total, dtype=F32, device=device)
        self.shadow = torch.zeros(self.total, dtype=BF16, device=device)
        cg = np.ones(self.total // CHUNK, dtype=np.uint8)  # 1 = no-decay group, 0 = decay group
        for o, n, nd in groups:
            if not nd:
                cg[o // CHUNK: (o + n + CHUNK - 1) // CHUNK] = 0
        self.chunk_group = torch.from_numpy(cg).to(device)
        self._params = plist
        with torch.no_grad():
            for name, p in named_params:
                v = self.view(name)
                v.copy_(p.detach().to(device=device, dtype=F32))
                p.data = v
        self.attach_grads()
        self._skip_zero = np.zeros(self.total // CHUNK, dtype=bool)   # chunks the backward overwrites (mark_overwritten)
        self._must_zero = np.zeros(self.total // CHUNK, dtype=bool)   # chunks some recorded backward adds to (mark_accumulated)
        self._zero_list = None
        self._shadow_version = -1
        self.shadow_epoch = 0      # bumped whenever the bf16 shadow changes (the e4m3 shadow follows it)
        self.fresh = True
        self.numel = sum(p.numel() for _, p in named_params)

    def _slice(self, buf, name):
        o = self.offset[name]
        n = int(np.prod(self.shape[name]))
        return buf[o:o + n]

    def view(self, name):
        return self._slice(self.params, name).view(self.shape[name])

    def gview(self, name):
        return self._slice(self.grads, name).view(self.shape[name])

    def bview(self, name, rows=None):
        """bf16 shadow as a 2-D [out, in] matrix"""
        s = self.shape[name]
        return self._slice(self.shadow, name).view(s[0] if rows is None else rows, -1)

    def g2d(self, name):
        s = self.shape[name]
        return self._slice(self.grads, name).view(s[0], -1)

    def fused_bias(self, qname, buf=None):
        o = self.offset[qname]
        d = int(np.prod(self.shape[qname]))
        return (self.params if buf is None else buf)[o:o + 3 * d]

    def range_of(self, names) -> Tuple[int, int]:
        lo = min(self.offset[n] for n in names)
        hi = max(self.offset[n] + int(np.prod(self.shape[n])) for n in names)
        return lo, _align(hi)

    def attach_grads(self):
        for name, p in self._params.items():
            p.grad = self.gview(name)

    def grads_attached(self) -> bool:
        p = next(iter(self._params.values()))
        return p.grad is not None and p.grad.data_ptr() == self.gview(self.names[0]).data_ptr()

    def owns(self, full: bool = False) -> bool:
        """True while the Parameters still are views of the flat buffer (.to()/.half() would break that).  The cheap form
        looks at the first and last parameter only (a module-wide .to() moves all of them)."""
        names = self.names if full else (self.names[0], self.names[-1])
        base = self.params.data_ptr()
        for name in names:
            p = self._params[name]
            if p.data_ptr() != base + 4 * self.offset[name] or p.dtype != F32:
                return False
        return True

    def _version(self) -> int:
        # in-place writes through a Parameter bump that Parameter's counter, writes through store views bump the buffer's
        return self.params._version + sum(p._version for p in self._params.values())

    def refresh_shadow(self, force=False):
        """bf16 shadow follows the fp32 masters; torch's version counters tell when someone wrote them
        (load_state_dict, a foreign optimizer, manual init).  The fused AdamW writes the shadow itself."""
        v = self._version()
        if force or v != self._shadow_version:
            ops.cast_bf16(self.params, self.shadow)
            self._shadow_version = v
            self.shadow_epoch += 1

    def mark_shadow_fresh(self):
        self._shadow_version = self._version()
        self.shadow_epoch += 1

    # ---- OCP e4m3 copy of the GEMM weights named by ``enable_fp8`` (the fp8 forward GEMMs, BASELINE configs[4])
    def enable_fp8(self, names):
        """a flat e4m3 shadow beside the bf16 one, one per-tensor scale per matrix in ``names``; refreshed by refresh_shadow8()"""
        self.fp8_names = list(names)
        seg = np.full(self.total // CHUNK, -1, dtype=np.int16)
        for i, n in enumerate(self.fp8_names):
            o, k = self.offset[n], int(np.prod(self.shape[n]))
            seg[o // CHUNK: (o + k + CHUNK - 1) // CHUNK] = i
        self.chunk_seg = torch.from_numpy(seg).to(self.device)
        self.shadow8 = torch.zeros(self.total, dtype=torch.float8_e4m3fn, device=self.device)
        self.w_scale_inv = torch.ones(len(self.fp8_names), dtype=F32, device=self.device)
        self.w_scale = torch.ones(len(self.fp8_names), dtype=F32, device=self.device)   # what the fused AdamW quantises with (delayed)
        self._amax_ws = torch.zeros(len(self.fp8_names), dtype=F32, device=self.device)
        self._shadow8_epoch = -1

    def b8view(self, name):
        s = self.shape[name]
        return self._slice(self.shadow8, name).view(s[0], -1)

    def w_si(self, name):
        i = self.fp8_names.index(name)
        return self.w_scale_inv[i:i + 1]

    def shadow8_current(self) -> bool:
        """the e4m3 shadow (and the per-matrix maxima in _amax_ws) belong to the current bf16 shadow"""
        return self._shadow8_epoch == self.shadow_epoch

    def q8_args(self, lo=0, hi=None):
        """what ops.adamw(q8=...) takes for the flat range [lo, hi): the fused update then writes the e4m3 shadow itself"""
        hi = self.total if hi is None else hi
        return (self.chunk_seg[lo // CHUNK:hi // CHUNK], self.w_scale, self._amax_ws, self.shadow8[lo:hi])

    def mark_shadow8_fresh(self):
        self._shadow8_epoch = self.shadow_epoch

    def refresh_shadow8(self):
        """exact per-matrix scales from the bf16 shadow (two launches): the first forward, and whenever somebody other than the fused
        AdamW changed the weights (load_state_dict, a foreign optimizer); the fused AdamW keeps the e4m3 shadow current itself"""
        if self._shadow8_epoch != self.shadow_epoch:
            ops.fp8_quantize_segments(self.shadow, self.chunk_seg, len(self.fp8_names), self._amax_ws, self.shadow8, self.w_scale_inv)
            self._shadow8_epoch = self.shadow_epoch

    def mark_overwritten(self, g: torch.Tensor):
        """``g`` (a view into the flat gradient buffer) is WRITTEN, not accumulated, by a backward that follows zero_grads()
        (an un-split weight-gradient GEMM in overwrite mode): zero_grads() need not clear it"""
        o = (g.data_ptr() - self.grads.data_ptr()) // 4
        lo, hi = -(-o // CHUNK), (o + g.numel()) // CHUNK          # whole chunks inside the tensor only
        if hi > lo and not bool(self._skip_zero[lo:hi].all()):
            self._skip_zero[lo:hi] = True
            self._zero_list = None

    def mark_accumulated(self, g: torch.Tensor):
        """``g`` is ADDED to by some recorded backward (a split weight-gradient GEMM: f32 atomics): zero_grads() must clear it,
        whatever another workspace's backward marked (a small batch runs the same GEMM unsplit and overwrites)"""
        o = (g.data_ptr() - self.grads.data_ptr()) // 4
        lo, hi = o // CHUNK, -(-(o + g.numel()) // CHUNK)          # every chunk the tensor touches
        stale = bool((self._skip_zero[lo:hi] & ~self._must_zero[lo:hi]).any())   # the zero_grads() before this call skipped it
        if not bool(self._must_zero[lo:hi].all()):
            self._must_zero[lo:hi] = True
            self._zero_list = None
        return stale

    def zero_grads(self):
        """optimizer.zero_grad(): everything the next backward accumulates into is cleared; the weight gradients it overwrites
        with plain stores (the encoder blocks' at ViT-B: 340 of 377 MB) are left as they are until that backward rewrites them"""
        skip = self._skip_zero & ~self._must_zero
        if not skip.any():
            self.grads.zero_()
        else:
            if self._zero_list is None:
                self._zero_list = torch.from_numpy(np.nonzero(~skip)[0].astype(np.int32)).to(self.device)
            ops.zero_chunks(self.grads, self._zero_list)
        self.fresh = True    # the next backward may overwrite (plain stores) instead of accumulate


# ======================================================================================================= runtime
def _wsplits(P, Q, R):
    """split the token reduction of a wgrad GEMM only when its 128x128 output tiles cannot fill the 256 CUs: every split
    costs one more f32 atomic pass over the weight gradient (chip-wide atomic rate ~1.3 TB/s), an unsplit one plain stores"""
    tiles = ((P + 127) // 128) * ((Q + 127) // 128)
    if tiles >= 96:
        return 1
    return int(max(1, min(-(-256 // tiles), 16, R // 1024)))


def _pick_wgrad_blocks(D: int, hid: int, depth: int = 12, bucketed: bool = True) -> int:
    """encoder blocks per grouped weight-gradient launch.  Two kernels take such a group (csrc/gemm.hip: r3_wanted): the 128 x 128 form
    (768 resident tiles, at most three blocks = 13 problems per launch) and the 256 x 128 ring kernel (256 resident units, up to 7
    blocks).  While every slot is filled both run at the same rate, so the group size is chosen for whole ROUNDS.  With gradient
    buckets to hand over (data parallel) a group never spans a bucket end and the small late buckets decide: <= 3 as before.  In
    one process the groups run on across the bucket ends: ViT-B 216 units per block -> 7 blocks = 5.9 rounds (then 5 + the patch embed =
    4.4) against 3 blocks = 1.69 rounds of the 128 x 128 form."""
    t128 = lambda p, q: -(-p // 128) * -(-q // 128)
    t = t128(3 * D, D) + t128(D, D) + t128(hid, D) + t128(D, hid)      # 128 x 128 tiles of one block's four weight gradients
    best, best_eff = 1, 0.0
    for g in (1, 2, 3):
        eff = g * t / (-(-g * t // 768) * 768)
        if eff > best_eff + 1e-9:
            best, best_eff = g, eff
    # tile rows of the ring kernel these widths go to (csrc/gemm.hip: ring_tm): 384 when every width is a multiple of 384 (ViT-B), else 256
    tm = 384 if (D % 384 == 0 and hid % 384 == 0 and os.environ.get("MOFO_GEMM_R4", "") != "0") else 256
    if D % tm == 0 and hid % tm == 0 and D % 128 == 0 and hid % 128 == 0 and os.environ.get("MOFO_GEMM_R3", "") != "0":
        # ring-kernel groups: clearly fuller rounds win (ViT-B: 7 blocks, 0.98 against 0.84); at equal, (nearly) whole rounds the ring
        # kernel wins on its main loop (ViT-L, 10 240 token rows: 2 blocks = 768 units = 3 rounds exactly, 1 005 against 946 TFLOP/s in
        # the step, 49.66 -> 49.09 ms).  ViT-B on 384-row tiles: 144 units per block, 7 blocks = 3.94 rounds, 5 (+ patch embed) = 2.9
        u = (3 * D // tm) * (D // 128) + (D // tm) * (D // 128) + (hid // tm) * (D // 128) + (D // tm) * (hid // 128)
        ring_best, ring_eff = None, 0.0
        for g in range(2, min(3 if bucketed else 7, depth) + 1):
            eff = g * u / (-(-g * u // 256) * 256)
            if g * u >= 3 * 256 and eff > ring_eff + 1e-9:
                ring_best, ring_eff = g, eff
        if ring_best is not None and (ring_eff > best_eff + 0.03 or (ring_eff >= 0.95 and ring_eff >= best_eff - 1e-9)):
            best = ring_best
    return best


_GRAPHS = os.environ.get("MOFO_GRAPH", "0") == "1"
_RING_SWITCHES = ("MOFO_GEMM_R3", "MOFO_GEMM_R3_TAIL", "MOFO_GEMM_R3_GRID", "MOFO_GEMM_R4")   # read per call by csrc/gemm.hip


class PretrainRuntime:
    """Forward / backward of encoder, bridge (encoder_to_decoder + token assembly), decoder and the fused loss."""

    def __init__(self, dims: Dims, store: FlatStore, enc_prefix: Optional[str] = "encoder.", dec_prefix: Optional[str] = "decoder.",
                 top: bool = True, forward_only: bool = False):
        self.d = dims
        self.store = store
        self.dev = store.device
        self.enc_prefix, self.dec_prefix, self.top = enc_prefix, dec_prefix, top
        self._ws: Dict[int, NS] = {}
        if enc_prefix is not None:
            self.pos_enc = sincos_table(dims.num_patches, dims.enc_dim).to(self.dev)
            self.encW = [self._block_weights(f"{enc_prefix}blocks.{i}.") for i in range(dims.enc_depth)]
        if top:
            self.pos_dec = sincos_table(dims.num_patches, dims.dec_dim).to(self.dev)
        if dec_prefix is not None:
            self.decW = [self._block_weights(f"{dec_prefix}blocks.{i}.") for i in range(dims.dec_depth)]
        self.segment_hook: Optional[Callable[[int, int, int], None]] = None  # (segment id, lo, hi) as gradient ranges complete
        # (set before the gradient buckets are planned: their sizes follow the group size)
        # encoder blocks per grouped weight-gradient launch: MOFO_WGRAD_BLOCKS, else the smallest group that fills whole rounds of the
        # 768 resident 128 x 128 tiles best (ViT-B: 432 tiles per block -> 3 blocks = 1.69 rounds; ViT-L: 768 per block -> 1 block = one
        # round exactly: 53.9 / 53.1 / 51.4 ms per step with 3 / 2 / 1 blocks per launch)
        # (in one process -- no process group at construction -- nothing is handed over at bucket ends and the groups may be larger)
        import torch.distributed as _td
        # (MOFO_FORCE_DP=1: bench.py's one-rank rehearsal of the data-parallel path hands buckets over like a real N > 1 job, so its groups
        # are the bucketed ones too -- otherwise the rehearsal would time group sizes no data-parallel job runs)
        bucketed = (_td.is_available() and _td.is_initialized() and _td.get_world_size() > 1) or os.environ.get("MOFO_FORCE_DP") == "1"
        self.wgrad_blocks = max(1, min(7, int(os.environ["MOFO_WGRAD_BLOCKS"]))) if os.environ.get("MOFO_WGRAD_BLOCKS") else \
            _pick_wgrad_blocks(dims.enc_dim, int(dims.enc_dim * dims.mlp_ratio), dims.enc_depth, bucketed)
        # under data parallelism the encoder's bucket / group plan may be switched after construction (set_enc_plan: bench.py times the
        # plans under the live exchange): the scratch is then sized for the largest group a plan may ask for
        self._enc_group_cap = max(self.wgrad_blocks, min(7, dims.enc_depth) if bucketed else 1)
        # DECODER (round 6): the weight gradients of ALL its blocks and of the head go into ONE launch at the end of its backward pass,
        # the token reduction sliced over the 8 XCDs with partial sums in a workspace (ops.gemm_wgrad_sliced; no f32 atomics, no split-K
        # passes over the gradients): 961 -> 816 us at ViT-B (tools/wgrad_dec_ab.py).  MOFO_WGRAD_SLICED=0: a grouped launch per
        # MOFO_WGRAD_BLOCKS_DEC blocks with split reductions, as in rounds 1-5.
        # The switch is read whenever a backward launch list is RECORDED (like MOFO_WGRAD_STREAM): bench.py times both routes in one process
        # (config.route_ab); the scratch below is sized for the larger group either way.
        self._sliced_capable = dec_prefix is not None and 1 <= dims.dec_depth <= 7
        self._slab_ws, self._slab_retired = None, []
        self._blocks_dec_env = max(1, min(3, int(os.environ.get("MOFO_WGRAD_BLOCKS_DEC", "1"))))
        self.wgrad_blocks_dec = self._blocks_dec_env
        # experiment switch MOFO_ENC_BUCKETS="6,3,2,1": encoder blocks per gradient bucket, read ONCE here (plan_segments and
        # encoder_backward must agree on the bucket ends) and refused aloud when malformed
        self._bucket_override = None
        if os.environ.get("MOFO_ENC_BUCKETS") and enc_prefix is not None:
            try:
                sizes = [int(x) for x in os.environ["MOFO_ENC_BUCKETS"].split(",")]
            except ValueError:
                raise ValueError(f"MOFO_ENC_BUCKETS={os.environ['MOFO_ENC_BUCKETS']!r}: expected comma-separated block counts, e.g. 6,3,2,1") from None
            if sum(sizes) != dims.enc_depth or any(x <= 0 for x in sizes):
                raise ValueError(f"MOFO_ENC_BUCKETS={os.environ['MOFO_ENC_BUCKETS']!r}: positive block counts that sum to the encoder depth {dims.enc_depth}")
            self._bucket_override = sizes
        # forward_only: the fine-tune / feature-extraction forward (modeling_finetune.py) -- no gradient buckets to plan
        self.segments = [] if forward_only else self.plan_segments()
        # The DECODER's residual stream is kept in bf16 (x_full, x_mid, x_out): its GEMMs reduce over 384 / 1536 and are bound
        # by HBM, and the f32 stream cost 16 B per token element and block in residual epilogues and LayerNorm reads (1.2 GB
        # per ViT-B B=32 step).  The encoder's stream (5 120 token rows, latency-bound kernels) stays f32.  MOFO_DEC_RESID=f32
        # restores the f32 decoder stream (A/B, parity debugging).
        self.dec_resid = F32 if os.environ.get("MOFO_DEC_RESID", "bf16") == "f32" else BF16
        # The encoder's stream stays f32 as in the reference: bf16 passed the parity gates and did not move the step (11.90 / 11.92 vs
        # 11.96 / 11.87 ms, round 2: its kernels are latency-bound at 5 120 token rows); the switch was retired in round 5.
        self.enc_resid = F32
        # MOFO_FP8=1 (BASELINE configs[4] "fp8 MFMA attention/MLP"): the four forward Linears of every block (qkv, proj, fc1, fc2) run on
        # OCP e4m3 operands with the block-scaled MFMA (2x the bf16 MFMA rate).  Per-tensor scales.  Weights: an e4m3 shadow written by
        # the fused AdamW with a delayed per-matrix scale (exact scales from the bf16 shadow before the first step / after a load).
        # Activations, delayed scaling per site: the e4m3 copies are written by the producing kernels -- LayerNorm 1 / 2 (qkv, fc1),
        # the attention forward (proj), fc1's GELU epilogue (fc2) -- with the scale from the amax the site saw in the previous forward;
        # the first forward of a runtime is run once more to calibrate.  The bf16 copies stay (the backward reads them): backward is bf16.
        self.fp8 = os.environ.get("MOFO_FP8", "0") == "1" and not forward_only and top and enc_prefix is not None and dec_prefix is not None
        if self.fp8:
            blocks = (self.encW if enc_prefix is not None else []) + (self.decW if dec_prefix is not None else [])
            kinds = (("qkv8", "attn.qkv.weight"), ("proj8", "attn.proj.weight"), ("fc18", "mlp.fc1.weight"), ("fc28", "mlp.fc2.weight"))
            names = [W.prefix + n for W in blocks for _, n in kinds if store.shape[W.prefix + n][1] % 128 == 0]
            self.fp8 = bool(names)
            if self.fp8:
                store.enable_fp8(names)
                for W in blocks:
                    for attr, n in kinds:
                        n = W.prefix + n
                        setattr(W, attr, store.b8view(n) if n in names else None)
                        setattr(W, attr + "_si", store.w_si(n) if n in names else None)
                nsite = 4 * len(blocks)       # per block: LayerNorm 1 out, attention out, LayerNorm 2 out, GELU out
                self.act_scales = torch.tensor([[16.0, 1.0 / 16.0]] * nsite, dtype=F32, device=self.dev)
                self.act_amax = torch.zeros(nsite, ops.FP8_AMAX_STRIPES, dtype=F32, device=self.dev)
                for i, W in enumerate(blocks):
                    W.site = 4 * i
                self._fp8_calibrated = False
        self._accumulate = False   # True when backward must ADD to existing gradients (no zero_grad since the last backward)
        self.side = torch.cuda.Stream(device=self.dev) if self.dev.type == "cuda" else None
        self.side2 = torch.cuda.Stream(device=self.dev) if self.dev.type == "cuda" else None
        self._seg_events = [torch.cuda.Event() for _ in range(8)] if self.dev.type == "cuda" else []
        self._bwd_done = torch.cuda.Event() if self.dev.type == "cuda" else None     # end of a backward's side-stream work
        self._asm_ready = torch.cuda.Event() if self.dev.type == "cuda" else None
        self.step_gate = None            # (loss, status, upstream) device words the next optimizer step is gated on
        self._side_launched = False
        self.norm_partial = torch.empty(1024, dtype=F32, device=self.dev)
        dmax = max(dims.enc_dim if enc_prefix is not None else 0, dims.dec_dim if dec_prefix is not None else 0, 64)
        self.ln_ws = torch.empty(2 * 1024 * dmax, dtype=F32, device=self.dev)   # LN-backward dgamma/dbeta block partials
        self._ln_dmax, self._ln_pool, self._ln_pending, self._ln_retired = dmax, [], [], []
        self.norm_out = torch.zeros(1, dtype=F32, device=self.dev)

    # ------------------------------------------------------------------ LayerNorm backward with grouped dgamma / dbeta reduction
    def _ln_bwd(self, dy, x, w, mean, rstd, dres, dx, dxb, gw, gb, **kw):
        """ops.layernorm_bwd whose dgamma / dbeta block partials stay in a workspace of their own; ``_ln_flush`` reduces up
        to forty LayerNorms' partials in one launch (before a gradient bucket is handed to the all-reduce / optimizer)"""
        if len(self._ln_pending) == 40:
            self._ln_flush()
        D = x.shape[1]
        k = len(self._ln_pending)
        M = kw.get("M") or dy.shape[0]
        need = 2 * ops.layernorm_bwd_blocks(M) * D      # what THIS LayerNorm's block partials take (an encoder one: 640 of the 1024 block rows)
        if k >= len(self._ln_pool):
            self._ln_pool.append(torch.empty(need, dtype=F32, device=self.dev))
        elif self._ln_pool[k].numel() < need:
            # a recorded launch list of ANOTHER workspace may hold this buffer's raw pointer (ops.replay passes data_ptr ints,
            # incl. the mofo_layernorm_bwd_finalize pointer table): the smaller buffer is retired, never freed
            self._ln_retired.append(self._ln_pool[k])
            self._ln_pool[k] = torch.empty(need, dtype=F32, device=self.dev)
        ws = self._ln_pool[k]
        nb = ops.layernorm_bwd(dy, x, w, mean, rstd, dres, dx, dxb, None, None, partial_ws=ws, **kw)
        self._ln_pending.append((ws, nb, D, gw, gb))

    def _ln_flush(self):
        if self._ln_pending:
            ops.layernorm_bwd_finalize(self._ln_pending)
            self._ln_pending = []

    # ------------------------------------------------------------------ weights
    def _block_weights(self, p):
        s = self.store
        return NS(prefix=p,
                  ln1w=s.view(p + "norm1.weight"), ln1b=s.view(p + "norm1.bias"), g_ln1w=s.gview(p + "norm1.weight"), g_ln1b=s.gview(p + "norm1.bias"),
                  qkvb=s.fused_bias(p + "attn.q_bias"), g_qkvb=s.fused_bias(p + "attn.q_bias", s.grads),
                  qkv=s.bview(p + "attn.qkv.weight"), g_qkv=s.g2d(p + "attn.qkv.weight"),
                  proj=s.bview(p + "attn.proj.weight"), g_proj=s.g2d(p + "attn.proj.weight"),
                  projb=s.view(p + "attn.proj.bias"), g_projb=s.gview(p + "attn.proj.bias"),
                  ln2w=s.view(p + "norm2.weight"), ln2b=s.view(p + "norm2.bias"), g_ln2w=s.gview(p + "norm2.weight"), g_ln2b=s.gview(p + "norm2.bias"),
                  fc1=s.bview(p + "mlp.fc1.weight"), g_fc1=s.g2d(p + "mlp.fc1.weight"), fc1b=s.view(p + "mlp.fc1.bias"), g_fc1b=s.gview(p + "mlp.fc1.bias"),
                  fc2=s.bview(p + "mlp.fc2.weight"), g_fc2=s.g2d(p + "mlp.fc2.weight"), fc2b=s.view(p + "mlp.fc2.bias"), g_fc2b=s.gview(p + "mlp.fc2.bias"),
                  names=[p + n for n in ("norm1.weight", "norm1.bias", "attn.q_bias", "attn.v_bias", "attn.qkv.weight", "attn.proj.weight",
                                         "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")])

    # ------------------------------------------------------------------ workspace
    def _block_ws(self, M, D, H, B, n, resid=F32):
        dev = self.dev
        hid = int(D * self.d.mlp_ratio)
        e = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
        f8 = torch.float8_e4m3fn
        extra = (dict(xln1_8=e(M, D, dt=f8), xln2_8=e(M, D, dt=f8), ao8=e(M, D, dt=f8), g8=e(M, hid, dt=f8))
                 if getattr(self, "fp8", False) and D % 128 == 0 and hid % 128 == 0 else {})
        return NS(**extra, xln1=e(M, D), mean1=e(M, dt=F32), rstd1=e(M, dt=F32), qkv=e(M, 3 * D), ao=e(M, D), lse=e(B * H * n, dt=F32),
                  x_mid=e(M, D, dt=resid), xln2=e(M, D), mean2=e(M, dt=F32), rstd2=e(M, dt=F32), h1=e(M, hid), g=e(M, hid),
                  x_out=e(M, D, dt=resid))

    def _block_ws_last(self, M, Mc, D, H, B, n, resid=F32):
        """workspace of the LAST decoder block when only the last ``Mc / B`` tokens of every clip are passed on (decoder.norm / head read
        x[:, -return_token_num:], modeling_pretrain.py:157): LayerNorm 1, qkv, keys and values cover all M rows, everything behind the
        attention (its output, proj, LayerNorm 2, the MLP, the block output) only the Mc rows that are read"""
        dev = self.dev
        hid = int(D * self.d.mlp_ratio)
        e = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
        f8 = torch.float8_e4m3fn
        extra = (dict(xln1_8=e(M, D, dt=f8), xln2_8=e(Mc, D, dt=f8), ao8=e(Mc, D, dt=f8), g8=e(Mc, hid, dt=f8))
                 if getattr(self, "fp8", False) and D % 128 == 0 and hid % 128 == 0 else {})
        return NS(**extra, compact=True, xln1=e(M, D), mean1=e(M, dt=F32), rstd1=e(M, dt=F32), qkv=e(M, 3 * D), lse=e(B * H * n, dt=F32),
                  ao=e(Mc, D), x_mid=e(Mc, D, dt=resid), xln2=e(Mc, D), mean2=e(Mc, dt=F32), rstd2=e(Mc, dt=F32), h1=e(Mc, hid), g=e(Mc, hid),
                  x_out=e(Mc, D, dt=resid))

    def _scratch(self, M, D, H, B, n, group=1, one_group=0):
        dev = self.dev
        hid = int(D * self.d.mlp_ratio)
        e = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
        ev = lambda: torch.cuda.Event() if self.dev.type == "cuda" else None
        # the weight-gradient GEMMs of ``group`` consecutive blocks run as ONE launch on a side stream while the next blocks'
        # activation-gradient chain proceeds, so everything they read (dh1, dx_mid copy, dqkv, the blocks' dx_out in the ring)
        # lives in 2 * group scratch sets / 2 * group + 1 ring buffers: one group being read by its launch, one being written.
        # The residual-stream GRADIENT lives in bf16 only (ring / dxbB): each LayerNorm backward reads it as bf16 and writes
        # one bf16 tensor -- no f32 copy is written and re-read (the forward residual stream stays f32).
        # ``one_group`` > 0 (decoder): a recording may also put all ``one_group`` blocks of the pass into ONE launch at its end (the sliced
        # route); nothing of the next pass runs beside that launch, so one_group sets / one_group + 1 ring buffers hold it
        nsets, nring = max(2 * group, one_group), max(2 * group + 1, one_group + 1)
        return NS(group=group, ring=[e(M, D) for _ in range(nring)], dxln=e(M, D),
                  sets=[NS(dh1=e(M, hid), dxbB=e(M, D), dqkv=e(M, 3 * D)) for _ in range(nsets)],
                  dao=e(M, D), delta=e(B * H * n, dt=F32), pending=[], gidx=0, gcount=0,
                  ready=[ev(), ev()], done=[ev(), ev()], used=[False, False], att_ready=ev(), att_done=ev())

    def ws(self, B: int, n_vis: Optional[int] = None, N: Optional[int] = None) -> NS:
        """workspace for batch size B (and visible-token count n_vis); allocated once, reused every step"""
        d = self.d
        N = d.num_patches if N is None else N
        key = (B, n_vis, N)
        if key in self._ws:
            return self._ws[key]
        dev = self.dev
        e = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
        w = NS(B=B, N=N, n_vis=n_vis, n_msk=N - n_vis if n_vis is not None else None)
        if self.enc_prefix is not None:
            w.clips = e(B, d.in_chans, d.num_frames, d.img_size, d.img_size, dt=F32)
        w.mask_u8 = torch.zeros(B, N, dtype=torch.uint8, device=dev)
        w.status = torch.zeros(1, dtype=I32, device=dev)
        if n_vis is not None:
            w.vis_idx = torch.zeros(B, n_vis, dtype=I32, device=dev)
            w.msk_idx = torch.zeros(B, N - n_vis, dtype=I32, device=dev)
        if self.enc_prefix is not None and n_vis is not None:
            Me = B * n_vis
            w.Me = Me
            w.xp = e(Me, d.patch_dim)
            w.enc_x0 = e(Me, d.enc_dim, dt=self.enc_resid)
            w.enc = [self._block_ws(Me, d.enc_dim, d.enc_heads, B, n_vis, self.enc_resid) for _ in range(d.enc_depth)]
            w.enc_out = e(Me, d.enc_dim)
            w.enc_mean, w.enc_rstd = e(Me, dt=F32), e(Me, dt=F32)
            # encoder: three blocks' weight gradients per launch (432 tiles of 128 x 128 per block = 1.69 per CU; 1296 = 5.06)
            w.enc_s = self._scratch(Me, d.enc_dim, d.enc_heads, B, n_vis, group=self._enc_group_cap)
            w.d_encout = e(Me, d.enc_dim)
        if self.dec_prefix is not None:
            Md = B * N
            w.Md = Md
            w.x_full = e(B, N, d.dec_dim, dt=self.dec_resid)
            # Dead work: the LAST decoder block's rows of the visible tokens feed nothing (only x[:, -return_token_num:] reaches
            # decoder.norm / head, modeling_pretrain.py:157; the visible tokens are the first n_vis rows of a clip, :259).  With the
            # full model (n_vis known here) that block works on the masked tokens only -- queries, proj, LayerNorm 2, MLP, their
            # backward and weight-gradient reductions: 10 % of its rows at mask 0.9.  MOFO_DEC_LAST_COMPACT=0 keeps all rows.
            w.dec_compact = (n_vis is not None and 0 < n_vis < N and d.dec_depth >= 1
                             and os.environ.get("MOFO_DEC_LAST_COMPACT", "1") == "1")
            w.dec = [self._block_ws(Md, d.dec_dim, d.dec_heads, B, N, self.dec_resid) for _ in range(d.dec_depth - (1 if w.dec_compact else 0))]
            if w.dec_compact:
                Mc = B * (N - n_vis)
                w.dec.append(self._block_ws_last(Md, Mc, d.dec_dim, d.dec_heads, B, N, self.dec_resid))
                hid = int(d.dec_dim * d.mlp_ratio)
                w.dec_c = NS(dx0=e(Mc, d.dec_dim), dxln=e(Mc, d.dec_dim), dh1=e(Mc, hid), dxbB=e(Mc, d.dec_dim), dao=e(Mc, d.dec_dim))
            w.dec_s = self._scratch(Md, d.dec_dim, d.dec_heads, B, N, group=self._blocks_dec_env,
                                    one_group=d.dec_depth if self._sliced_capable else 0)
            w.dec_s.is_dec = True
            # Shared work: in the FIRST decoder block the rows of the masked tokens are mask_token + pos[j] (modeling_pretrain.py:259-262)
            # -- a function of the position alone, and so are their LayerNorm 1 and qkv rows.  With the full model LayerNorm 1, the qkv
            # GEMM, its dgrad, the LayerNorm backward and the qkv weight-gradient reduction run on [B * n_vis visible rows | N position
            # rows] ("cat" rows: 6 688 instead of 50 176 at ViT-B, B = 32) and ops.dec0_gather / dec0_reduce move between the two
            # layouts.  MOFO_DEC0_SHARE=0 computes all rows.
            w.dec_share = (w.dec_compact and d.dec_depth >= 2 and self.top and self.enc_prefix is not None and self.dec_resid == BF16
                           and d.dec_dim % 8 == 0 and os.environ.get("MOFO_DEC0_SHARE", "1") == "1")
            if w.dec_share:
                Mc, Dd = B * n_vis + N, d.dec_dim
                w.dec0 = NS(xcat=e(Mc, Dd), xln1=e(Mc, Dd), mean1=e(Mc, dt=F32), rstd1=e(Mc, dt=F32), qkv=e(Mc, 3 * Dd),
                            inv=torch.zeros(B, N, dtype=I32, device=dev), pos_idx=torch.arange(N, dtype=I32, device=dev).view(1, N),
                            dqkv=e(Mc, 3 * Dd), dres=e(Mc, Dd), dxln=e(Mc, Dd), dxcat=e(Mc, Dd), msk_idx=w.msk_idx, n_vis=n_vis, N=N)
                w.dec[0].xln1 = w.dec[0].mean1 = w.dec[0].rstd1 = None      # block 0 keeps these per cat row (w.dec0)
            if n_vis is not None:
                Mm = B * (N - n_vis)
                w.Mm = Mm
                w.dec_ln = e(Mm, d.dec_dim)
                w.dec_mean, w.dec_rstd = e(Mm, dt=F32), e(Mm, dt=F32)
                w.pred = e(Mm, d.patch_out)
                w.dpred = e(Mm, d.patch_out)
                w.d_decln = e(Mm, d.dec_dim)
                w.row_loss = e(Mm, dt=F32)
                w.loss = torch.zeros(1, dtype=F32, device=dev)
                w.d_e2d = e(B * n_vis, d.dec_dim)
                w.asm_partial = e(ops.assemble_bwd_blocks(B, N) * d.dec_dim, dt=F32)   # column sums of d(mask_token), per block
        self._ws[key] = w
        return w

    # ------------------------------------------------------------------ inputs
    def set_inputs(self, w: NS, videos: torch.Tensor, mask: Optional[torch.Tensor]):
        """H2D / D2D copy of the batch into the persistent input buffers, then mask -> index lists on the device
        (no host sync: the reference's boolean indexing does a nonzero() round trip, modeling_pretrain.py:90)."""
        if videos.dtype == torch.uint8:
            # the reference's Stack() output [B,H,W,T*3]: kept as bytes, normalised inside the gather / target kernels
            d = self.d
            if tuple(videos.shape) != (w.B, d.img_size, d.img_size, d.num_frames * 3):
                raise ValueError(f"uint8 input must be frames [B,H,W,T*3] = {(w.B, d.img_size, d.img_size, d.num_frames * 3)}, got {tuple(videos.shape)}")
            if getattr(w, "frames_u8", None) is None:
                w.frames_u8 = torch.empty(videos.shape, dtype=torch.uint8, device=self.dev)
            if videos.data_ptr() != w.frames_u8.data_ptr():
                w.frames_u8.copy_(videos, non_blocking=True)
            w.src_u8 = True
        else:
            w.src_u8 = False
            if videos.data_ptr() != w.clips.data_ptr():
                w.clips.copy_(videos, non_blocking=True)
        if mask is not None:
            if mask.data_ptr() != w.mask_u8.data_ptr():       # a loader may write masks straight into the persistent buffer, like clips
                w.mask_u8.copy_(mask.reshape(w.B, -1), non_blocking=True)
            # persistent buffers on both sides: replayed from a one-entry list like every other launch of the step (the checked call was
            # 15 us of host time in front of the first kernel of the step, tools/step_start_gap.py)
            self.cached(w, ("mask_idx",), lambda: ops.mask_to_indices(w.mask_u8, w.n_vis, w.vis_idx, w.msk_idx, w.status))

    # ------------------------------------------------------------------ transformer block
    def _block_fwd(self, W, L, x_in, B, n, H, share=None, qb=0):
        """``share`` (the first decoder block of the full model, see ``ws``): LayerNorm 1 and the qkv GEMM run once per cat row.
        ``qb`` > 0 (the LAST decoder block on the tokens that are read, rows qb .. n - 1 of every clip): keys / values from all rows, queries
        and everything behind the attention from those rows only; the residual input is read through the GEMM's residual row map.
        MOFO_FP8=1: the four Linears on e4m3 operands (the shared qkv of ``share`` stays bf16: 13 % of a block's qkv rows)."""
        eps, scale = self.d.eps, 64 ** -0.5
        f8 = self.fp8 and getattr(W, "qkv8", None) is not None and hasattr(L, "ao8")
        sc, am, st = (self.act_scales, self.act_amax, W.site) if f8 else (None, None, 0)
        if share is not None:
            Z = share
            ops.layernorm_fwd(Z.xcat, W.ln1w, W.ln1b, eps, Z.xln1, Z.mean1, Z.rstd1)
            ops.gemm(ops.GEMM_NT, ops.EPI_BF16, Z.xln1, W.qkv, Z.qkv, bias=W.qkvb)
            ops.dec0_gather(Z.qkv, Z.msk_idx, Z.N, L.qkv)
        elif f8:
            ops.layernorm_fwd_q(x_in, W.ln1w, W.ln1b, eps, L.xln1, L.mean1, L.rstd1, L.xln1_8, sc[st, 0:1], am[st])
            ops.gemm(ops.GEMM_NT_FP8, ops.EPI_BF16, L.xln1_8, W.qkv8, L.qkv, bias=W.qkvb, a_scale_inv=sc[st, 1:2], b_scale_inv=W.qkv8_si)
        else:
            ops.layernorm_fwd(x_in, W.ln1w, W.ln1b, eps, L.xln1, L.mean1, L.rstd1)
            ops.gemm(ops.GEMM_NT, ops.EPI_BF16, L.xln1, W.qkv, L.qkv, bias=W.qkvb)
        rmap = dict(rows_in=n - qb, rows_out=n, row_off=qb) if qb else {}
        if f8:
            ops.attention_fwd(L.qkv, B, n, H, scale, L.ao, L.lse, q_begin=qb, out8=L.ao8, q_scale=sc[st + 1, 0:1], q_amax=am[st + 1])
            A, Wp, op, kw = L.ao8, W.proj8, ops.GEMM_NT_FP8, dict(a_scale_inv=sc[st + 1, 1:2], b_scale_inv=W.proj8_si)
        else:
            ops.attention_fwd(L.qkv, B, n, H, scale, L.ao, L.lse, q_begin=qb)
            A, Wp, op, kw = L.ao, W.proj, ops.GEMM_NT, {}
        if L.x_mid.dtype == BF16:      # bf16 residual stream (decoder): the residual rides in the GEMM's `aux` operand
            ops.gemm(op, ops.EPI_RESID_BF16, A, Wp, L.x_mid, bias=W.projb, aux=x_in, **rmap, **kw)
        else:
            ops.gemm(op, ops.EPI_RESID_F32, A, Wp, L.x_mid, bias=W.projb, resid=x_in, **rmap, **kw)
        if f8:
            ops.layernorm_fwd_q(L.x_mid, W.ln2w, W.ln2b, eps, L.xln2, L.mean2, L.rstd2, L.xln2_8, sc[st + 2, 0:1], am[st + 2])
            ops.gemm(ops.GEMM_NT_FP8, ops.EPI_BIAS_GELU, L.xln2_8, W.fc18, L.h1, C2=L.g, bias=W.fc1b, a_scale_inv=sc[st + 2, 1:2],
                     b_scale_inv=W.fc18_si, C8=L.g8, q_scale=sc[st + 3, 0:1], q_amax=am[st + 3])
            A, Wf, kw = L.g8, W.fc28, dict(a_scale_inv=sc[st + 3, 1:2], b_scale_inv=W.fc28_si)
        else:
            ops.layernorm_fwd(L.x_mid, W.ln2w, W.ln2b, eps, L.xln2, L.mean2, L.rstd2)
            ops.gemm(ops.GEMM_NT, ops.EPI_BIAS_GELU, L.xln2, W.fc1, L.h1, C2=L.g, bias=W.fc1b)
            A, Wf, kw = L.g, W.fc2, {}
        if L.x_out.dtype == BF16:
            ops.gemm(op, ops.EPI_RESID_BF16, A, Wf, L.x_out, bias=W.fc2b, aux=L.x_mid, **kw)
        else:
            ops.gemm(op, ops.EPI_RESID_F32, A, Wf, L.x_out, bias=W.fc2b, resid=L.x_mid, **kw)
        return L.x_out

    def _block_fwd_last(self, W, L, x_in, B, n, H, qb):
        """the last decoder block on the tokens that are read (rows qb .. n - 1 of every clip), see ``_block_fwd``"""
        return self._block_fwd(W, L, x_in, B, n, H, qb=qb)

    def _block_bwd_last(self, W, L, S, C, x_in, B, n, H, qb, flush=False):
        """backward of _block_fwd_last (always the first block of a decoder backward pass: scratch set 0, ring 0 -> 1).  The gradient
        wrt the block output arrives compact in ``C.dx0``; the gradient wrt the block INPUT covers all rows again (keys / values and
        LayerNorm 1 saw them) and lands in ``S.ring[1]``; the residual term exists for the rows qb .. n - 1 only (partial-residual
        LayerNorm backward).  The dq rows of the skipped queries are cleared: the qkv dgrad and weight gradient read all rows."""
        scale = 64 ** -0.5
        D = x_in.shape[1]
        R = len(S.ring)
        T = S.sets[0]
        dxb_in = S.ring[1 % R]
        slot = S.gidx % 2
        if S.gcount == 0 and S.used[slot]:
            ops.host_op(lambda ev=S.done[slot]: torch.cuda.current_stream().wait_event(ev))
            S.used[slot] = False
        ops.gemm(ops.GEMM_NN, ops.EPI_DGELU_BF16, C.dx0, W.fc2, C.dh1, aux=L.h1)
        ops.gemm(ops.GEMM_NN, ops.EPI_BF16, C.dh1, W.fc1, C.dxln)
        self._ln_bwd(C.dxln, L.x_mid, W.ln2w, L.mean2, L.rstd2, C.dx0, None, C.dxbB, W.g_ln2w, W.g_ln2b)
        ops.gemm(ops.GEMM_NN, ops.EPI_BF16, C.dxbB, W.proj, C.dao)
        # (the dq rows of the skipped queries are cleared by the dK/dV pass below: mofo_attention_bwd_dkv_range)
        ops.attention_bwd_dq_delta(L.qkv, L.ao, C.dao, L.lse, S.delta, B, n, H, scale, T.dqkv, q_begin=qb)
        ops.attention_bwd_dkv(L.qkv, C.dao, L.lse, S.delta, B, n, H, scale, T.dqkv, q_begin=qb)
        ops.gemm(ops.GEMM_NN, ops.EPI_BF16, T.dqkv, W.qkv, S.dxln)
        self._ln_bwd(S.dxln, x_in, W.ln1w, L.mean1, L.rstd1, C.dxbB, None, dxb_in, W.g_ln1w, W.g_ln1b, dres_rows=(n, qb))
        S.pending += [(C.dx0, L.g, W.g_fc2, W.g_fc2b, (0, 0)), (C.dh1, L.xln2, W.g_fc1, W.g_fc1b, (0, 0)),
                      (C.dxbB, L.ao, W.g_proj, W.g_projb, (0, 0)), (T.dqkv, L.xln1, W.g_qkv, W.g_qkvb, (D, 2 * D))]
        S.gcount += 1
        if S.gcount == S.group or flush:
            self._wgrad_flush(S, slot, n)
            S.gidx, S.gcount = S.gidx + 1, 0

    def _wgrad(self, dY, X, G, bias_grad=None):
        """dW (+)= dY^T X, and the bias gradient db += colsum(dY) fused into the same launch"""
        R, P = dY.shape
        Q = X.shape[1]
        ops.gemm(ops.GEMM_TN, ops.EPI_F32, dY, X, G, splits=_wsplits(P, Q, R), accumulate=self._accumulate, colsum=bias_grad)

    @property
    def wgrad_sliced(self) -> bool:
        return self._sliced_capable and os.environ.get("MOFO_WGRAD_SLICED", "1") == "1"

    def _wgrad_sliced(self, problems):
        """a whole pass's weight gradients in one launch, the reduction sliced over the XCDs (ops.gemm_wgrad_sliced): partial sums in
        ``self._slab_ws``, summed into the gradients by the call's second kernel -- every destination is plainly stored"""
        probs = [(dY, X, G, dict(accumulate=self._accumulate, colsum=bg, colsum_skip=skip)) for dY, X, G, bg, skip in problems]
        need = ops.gemm_wgrad_sliced_ws(probs, 8)
        if self._slab_ws is None or self._slab_ws.numel() < need:
            if self._slab_ws is not None:
                self._slab_retired.append(self._slab_ws)     # a recorded launch list may hold its pointer: retired, never freed
            self._slab_ws = torch.empty(need, dtype=F32, device=self.dev)
        if not self._accumulate and os.environ.get("MOFO_ZERO_ALL", "0") != "1":
            for pr in problems:
                self.store.mark_overwritten(pr[2])
        ops.gemm_wgrad_sliced(probs, self._slab_ws, 8)

    def _wgrad_group(self, problems, sliced=False):
        """the weight gradients of one transformer block as ONE grouped launch: their 128x128 tiles together fill the
        chip (ViT-B encoder: 108+36+144+144), so no split-K -> plain stores instead of f32 atomics"""
        if sliced and len(problems) <= 32:
            return self._wgrad_sliced(problems)
        R = problems[0][0].shape[0]
        tiles = sum(((pr[0].shape[1] + 127) // 128) * ((pr[1].shape[1] + 127) // 128) for pr in problems)
        # A group that leaves the 256 CUs with fewer than ~2.3 tiles each is split along the token reduction until it has
        # ~`target` blocks (756 = 3 resident blocks on 252 CUs) -- but only while every split keeps >= 4096 token rows: each
        # split adds a pass of f32 atomics over the gradients, which 64+ k-steps per tile amortise (decoder, 50 176 rows:
        # 108 tiles x 7: 242 -> 228 us) and 40 do not (encoder, 5 120 rows: 2.28 -> 2.68 ms per step when split).
        thr, target = int(os.environ.get("MOFO_WGRAD_THR", "600")), int(os.environ.get("MOFO_WGRAD_TARGET", "756"))
        splits = 1 if tiles >= thr else int(max(1, min(-(-target // tiles), 16, R // 4096)))
        probs = [(dY, X, G, dict(splits=splits, accumulate=self._accumulate, colsum=bg, colsum_skip=skip)) for dY, X, G, bg, skip in problems]
        # which kernel takes the group, and which destinations receive f32 atomics from several workgroups (split reductions, or --
        # ring kernel -- the units of a last, partial round dealt in chunks): those must hold zeros, the others are plainly stored
        # The C side reads the ring switches from the environment on EVERY call, the bookkeeping below is fixed when the list is
        # recorded: a switch changed between recording and a replay would send atomics onto gradients zero_grad skipped (or the reverse).
        # The recorded list therefore carries a check of the switches it was planned under; changing one needs invalidate_lists().
        snap = tuple(os.environ.get(k) for k in _RING_SWITCHES)

        def _same_switches(snap=snap):
            if tuple(os.environ.get(k) for k in _RING_SWITCHES) != snap:
                raise RuntimeError("a MOFO_GEMM_R3* / MOFO_GEMM_R4 switch changed after this backward was recorded: the recorded zero / overwrite "
                                   "plan no longer matches the kernel's route -- call runtime.invalidate_lists() after changing it")
        ops.host_op(_same_switches)
        ring, shared = ops.gemm_grouped_plan(ops.GEMM_TN, ops.EPI_F32, probs)
        if not ring and len(probs) > 13:
            # the 128 x 128 kernel takes 13 problems (three blocks + one) per launch
            chunks = [problems[i:i + 12] for i in range(0, len(problems), 12)]
            if len(chunks) > 1 and len(chunks[-1]) == 1:
                chunks[-2] += chunks.pop()
            for c in chunks:
                self._wgrad_group(c)
            return
        for pr, sh in zip(problems, shared):
            if not sh and not self._accumulate and os.environ.get("MOFO_ZERO_ALL", "0") != "1":
                self.store.mark_overwritten(pr[2])     # plain stores: zero_grad may skip this tensor from now on
            elif sh:
                # f32 atomics onto what zero_grad left there
                if self.store.mark_accumulated(pr[2]) and not self._accumulate:
                    # (recording run only) another workspace's backward had made zero_grads skip this tensor.  The clear runs
                    # on the stream the launch below is issued on (the side stream inside _wgrad_flush), so that it is
                    # ordered before that launch's atomics; a torch op on the CURRENT stream would race with them.
                    st = ops.launch_stream()
                    with torch.cuda.stream(st):
                        pr[2].zero_()
        ops.gemm_grouped(ops.GEMM_TN, ops.EPI_F32, probs)

    def _block_bwd(self, W, L, S, j, x_in, B, n, H, flush=False, hold=False, share=None):
        """Backward of the j-th block of a backward pass (j = 0 for the top block).  Reads the gradient wrt the block output
        from ``S.ring[j % R]`` (bf16), writes the gradient wrt its input to ``S.ring[(j + 1) % R]``.  The block's weight
        gradients are DEFERRED: they join the pending group, which is launched on the side stream once it holds ``S.group``
        blocks (or when ``flush`` says a gradient bucket ends here).  Scratch set j % (2 G) and ring buffer j % (2 G + 1) keep
        what that launch reads while the following blocks already run: groups alternate between two event slots, a group
        starts by waiting for the launch two groups back (launches are in order on the one side stream, and two consecutive
        groups span at most 2 G blocks, so every buffer block j rewrites -- last used by block j - 2 G -- is free by then)."""
        scale = 64 ** -0.5
        D = x_in.shape[1]
        G, R = S.group, len(S.ring)
        T = S.sets[j % len(S.sets)]
        dxb_out, dxb_in = S.ring[j % R], S.ring[(j + 1) % R]
        slot = S.gidx % 2
        if S.gcount == 0 and S.used[slot]:
            ops.host_op(lambda ev=S.done[slot]: torch.cuda.current_stream().wait_event(ev))
            S.used[slot] = False
        # MLP: x_out = x_mid + fc2(gelu(fc1(LN2(x_mid))))
        ops.gemm(ops.GEMM_NN, ops.EPI_DGELU_BF16, dxb_out, W.fc2, T.dh1, aux=L.h1)
        ops.gemm(ops.GEMM_NN, ops.EPI_BF16, T.dh1, W.fc1, S.dxln)
        self._ln_bwd(S.dxln, L.x_mid, W.ln2w, L.mean2, L.rstd2, dxb_out, None, T.dxbB, W.g_ln2w, W.g_ln2b)
        # attention: x_mid = x_in + proj(attn(LN1(x_in)))
        ops.gemm(ops.GEMM_NN, ops.EPI_BF16, T.dxbB, W.proj, S.dao)
        # the passes of mofo_attention_bwd as their own C-ABI calls (same stream, same kernels): each shows up under its own name in
        # the per-class timing.  (dQ and dK/dV on two streams measured neutral, round 1; retired in round 5.)
        if n <= 160:
            # short sequences (the encoder's visible tokens): one fused kernel per (clip, head) behind the combined entry
            ops.attention_bwd(L.qkv, L.ao, S.dao, L.lse, B, n, H, scale, T.dqkv, S.delta)
        elif os.environ.get("MOFO_ATTN_DELTA_KERNEL", "0") == "1":
            ops.attention_delta(L.ao, S.dao, B, n, H, S.delta)
            ops.attention_bwd_dkv(L.qkv, S.dao, L.lse, S.delta, B, n, H, scale, T.dqkv)
            ops.attention_bwd_dq(L.qkv, S.dao, L.lse, S.delta, B, n, H, scale, T.dqkv)
        else:
            # the dQ pass computes delta = rowsum(dO * O) on the way and leaves it for the dK/dV pass (no delta kernel: 19 us per layer)
            ops.attention_bwd_dq_delta(L.qkv, L.ao, S.dao, L.lse, S.delta, B, n, H, scale, T.dqkv)
            ops.attention_bwd_dkv(L.qkv, S.dao, L.lse, S.delta, B, n, H, scale, T.dqkv)
        if share is not None:
            # the adjoint of the forward's row sharing: qkv gradient and residual gradient summed per cat row (visible rows copied,
            # position rows added over the clips that mask the position, f32), then dgrad / LayerNorm backward / weight gradient on
            # those rows; the result is the gradient wrt the CAT rows of the decoder input (bridge_backward reads it as such)
            Z = share
            ops.dec0_reduce(T.dqkv, Z.inv, Z.n_vis, Z.dqkv)
            ops.dec0_reduce(T.dxbB, Z.inv, Z.n_vis, Z.dres)
            ops.gemm(ops.GEMM_NN, ops.EPI_BF16, Z.dqkv, W.qkv, Z.dxln)
            self._ln_bwd(Z.dxln, Z.xcat, W.ln1w, Z.mean1, Z.rstd1, Z.dres, None, Z.dxcat, W.g_ln1w, W.g_ln1b)
            qkv_grad = (Z.dqkv, Z.xln1, W.g_qkv, W.g_qkvb, (D, 2 * D))
        else:
            ops.gemm(ops.GEMM_NN, ops.EPI_BF16, T.dqkv, W.qkv, S.dxln)
            self._ln_bwd(S.dxln, x_in, W.ln1w, L.mean1, L.rstd1, T.dxbB, None, dxb_in, W.g_ln1w, W.g_ln1b)
            qkv_grad = (T.dqkv, L.xln1, W.g_qkv, W.g_qkvb, (D, 2 * D))
        # parameter gradients of the whole block (weights + biases; the bias gradients are column sums of the same dY
        # operands, fused into the GEMMs): one grouped launch on the SIDE stream, off the activation-gradient chain
        S.pending += [(dxb_out, L.g, W.g_fc2, W.g_fc2b, (0, 0)), (T.dh1, L.xln2, W.g_fc1, W.g_fc1b, (0, 0)),
                      (T.dxbB, L.ao, W.g_proj, W.g_projb, (0, 0)), qkv_grad]
        S.gcount += 1
        if (S.gcount == G or flush) and not hold:      # hold: the caller adds one more problem to this group and flushes it
            self._wgrad_flush(S, slot, n)
            S.gidx, S.gcount = S.gidx + 1, 0

    def _wgrad_flush(self, S, slot, n):
        """launch the pending blocks' weight gradients (one grouped launch) -- on the side stream unless MOFO_WGRAD_STREAM says
        otherwise"""
        if not S.pending:
            return
        group, S.pending = S.pending, []
        # Default since round 4: the SAME stream as the activation-gradient chain.  With round 4's kernels two same-box pairs measured
        # 11.67 / 11.67 ms (main) against 11.76 / 11.71 (side) for the step and 4.54 against 4.69 ms for the encoder-only step
        # (profiles/r04_wgrad_stream_k2.txt): a CU-filling grouped launch beside the chain slows the chain's kernels by what it hides,
        # and the one-tile-per-CU GEMMs (gemm_k2.h: 128 KiB of LDS) cannot start on a CU that still holds weight-gradient blocks.
        # MOFO_WGRAD_STREAM=side restores the side stream (main_enc / main_dec: per pass).
        # Round 6: the decoder's one sliced launch on the side stream (main_enc) beside the bridge and the first encoder blocks was tried as the
        # default: 11.13 / 11.12 / 11.15 / 11.09 against 11.21 / 11.20 / 11.17 / 11.12 ms on one box, but 10.615 / 10.576 / 10.571 against
        # 10.553 / 10.561 / 10.550 on another, and every main-stream kernel it overlaps reads 1.37 x longer in the event brackets and the
        # kernel trace (dgrad class 45 us per launch instead of 33): everything stays on the main stream.
        mode = os.environ.get("MOFO_WGRAD_STREAM", "main")
        sliced = self.wgrad_sliced and S is not None and getattr(S, "is_dec", False)
        if mode == "main" or (mode == "main_enc" and n <= 512) or (mode == "main_dec" and n > 512):
            self._wgrad_group(group, sliced)  # same stream: no fork / join events (each costs ~10 us of queue bubble)
            return
        side = self.side
        self._side_launched = True        # _backward joins the side stream once more at its end (see there)
        ops.host_op(lambda ev=S.ready[slot]: ev.record(torch.cuda.current_stream()))
        ops.use_stream(side)
        ops.host_op(lambda ev=S.ready[slot]: side.wait_event(ev))
        self._wgrad_group(group, sliced)
        ops.host_op(lambda ev=S.done[slot]: ev.record(side))
        ops.use_stream(None)
        S.used[slot] = True

    def _join_side(self, S):
        """main stream waits for every outstanding side-stream weight-gradient launch of this scratch"""
        for k in range(2):
            if S.used[k]:
                ops.host_op(lambda ev=S.done[k]: torch.cuda.current_stream().wait_event(ev))
                S.used[k] = False

    def set_enc_plan(self, buckets: Optional[List[int]], blocks: Optional[int] = None):
        """switch the encoder's gradient buckets (blocks per all-reduce range, from the top block down; None: the default plan for the
        group size) and the encoder blocks per grouped weight-gradient launch, after construction: the segments are planned again and
        every recorded launch list is dropped.  For bench.py's A/B of {6, 3, 2, 1 buckets with 3-block groups} against {7, 5 buckets
        with ring-kernel groups of 7 / 5} under the live exchange (round-5 review, item 6); every rank must make the same call."""
        if blocks is not None:
            if not 1 <= blocks <= self._enc_group_cap:
                raise ValueError(f"set_enc_plan: {blocks} blocks per launch, this runtime's scratch holds groups of up to {self._enc_group_cap}")
            self.wgrad_blocks = int(blocks)
        if buckets is not None and (sum(buckets) != self.d.enc_depth or any(x <= 0 for x in buckets)):
            raise ValueError(f"set_enc_plan: buckets {buckets} must be positive block counts that sum to the encoder depth {self.d.enc_depth}")
        self._bucket_override = list(buckets) if buckets is not None else None
        self.segments = self.plan_segments()
        self.invalidate_lists()

    def enc_buckets(self) -> List[int]:
        """the encoder's gradient buckets of THIS runtime (the override of MOFO_ENC_BUCKETS, else by depth and group size; groups of
        more than three blocks exist in one process only, where no bucket is handed over: the plan then follows the three-block rule)"""
        if self._bucket_override is not None:
            return list(self._bucket_override)
        return self._enc_buckets(self.d.enc_depth, min(self.wgrad_blocks, 3))

    @staticmethod
    def _enc_buckets(depth: int, group: int = 1) -> List[int]:
        """encoder blocks per gradient bucket, from the top block down: shrinking buckets (12 -> 5, 3, 2, 1, 1) so that the
        all-reduce left exposed after the last weight gradient is the smallest one (one block + patch embed: 33 MB at
        ViT-B instead of 85 MB with equal buckets of three), while the early, fully overlapped ones stay large.  With
        ``group`` > 1 encoder blocks per grouped weight-gradient launch the large early buckets are whole groups (12, group 3 ->
        6, 3, 2, 1): a bucket end cuts a group short (x1 / x2 groups run at 812 / 834 TFLOP/s against 963 for x3) and every
        bucket costs ~0.08 ms of hand-over (one rank under RCCL, ms per step: 5,3,2,1,1 11.80-11.83 | 6,3,2,1 11.71 | 3,3,3,3 11.68 |
        6,6 11.57 | 12 11.49 | no exchange 11.30 -- the last three expose 85 / 170 / 340 MB of all-reduce after the backward)"""
        sizes, rem = [], depth
        while rem > 0:
            take = max(1, min(rem, -(-rem * 2 // 5)))
            if group > 1 and rem > 2 * group and take % group:
                take = min(rem, -(-take // group) * group)
            sizes.append(take)
            rem -= take
        return sizes

    def plan_segments(self) -> List[Tuple[int, int]]:
        """Contiguous ranges of the flat gradient buffer in the order backward completes them (the data-parallel
        all-reduce buckets): [decoder + head + e2d + mask_token], then encoder blocks from the top in shrinking groups
        (encoder.norm rides with the first group, patch_embed with the last).  Together they tile the buffer."""
        d, s = self.d, self.store
        segs: List[List[str]] = []
        if self.dec_prefix is not None:
            p = self.dec_prefix
            names = [p + "head.weight", p + "head.bias", p + "norm.weight", p + "norm.bias"]
            for W in self.decW:
                names += W.names
            if self.top:
                names += ["encoder_to_decoder.weight", "mask_token"]
            segs.append(names)
        if self.enc_prefix is not None:
            p = self.enc_prefix
            cur = [p + "norm.weight", p + "norm.bias"]
            i = d.enc_depth - 1
            buckets = self.enc_buckets()
            for bi, nb in enumerate(buckets):
                for _ in range(nb):
                    cur += self.encW[i].names
                    i -= 1
                if bi + 1 < len(buckets):
                    segs.append(cur)
                    cur = []
            cur += [p + "patch_embed.proj.weight", p + "patch_embed.proj.bias"]
            segs.append(cur)
        return [s.range_of(n) for n in segs]

    def _seg(self, idx: int):
        ops.host_op(lambda: self._seg_now(idx))

    def _seg_now(self, idx: int):
        if self.segment_hook is None:
            return
        lo, hi = self.segments[idx]
        if self.side is None:
            self.segment_hook(idx, lo, hi)
            return
        # The range's kernels were issued on TWO streams (activation-gradient chain + LayerNorm reduces on the caller's, grouped
        # weight gradients on the side stream) and the consumer (torch.distributed's all-reduce) orders itself behind torch's
        # CURRENT stream: it is handed the range on the side stream, which first waits for an event recorded here on the main
        # stream.  The exchange starts behind both, and the main stream never joins a weight-gradient launch at a bucket end
        # (that join cost the data-parallel step 0.27 ms, round 2).
        ev = self._seg_events[idx % len(self._seg_events)]
        ev.record(torch.cuda.current_stream())
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            self.segment_hook(idx, lo, hi)

    def invalidate_lists(self):
        """forget every recorded launch list (and graph): the next call of each sequence runs its Python wrappers again and records
        anew.  For switches that are read while a list is RECORDED (MOFO_WGRAD_STREAM, ...): bench.py's data-parallel route A/B."""
        for w in self._ws.values():
            w.__dict__.pop("_lists", None)

    def cached(self, w: NS, tag, fn):
        """run ``fn`` (a fixed launch sequence over workspace ``w``) -- checked and recorded the first time, replayed as a
        flat launch list afterwards (no tensor checks, no Python-side argument marshalling beyond ctypes)"""
        cache = w.__dict__.setdefault("_lists", {})
        lst = cache.get(tag)
        if lst is None:
            from . import _lib
            rec = []
            _lib.RECORDER = rec
            try:
                out = fn()
            finally:
                _lib.RECORDER = None
            cache[tag] = (rec, out)
            return out
        if _GRAPHS and self.segment_hook is None:
            return self._graph_replay(cache, tag, lst)
        ops.replay(lst[0])
        return lst[1]

    def _graph_replay(self, cache, tag, lst):
        """MOFO_GRAPH=1: the recorded launch list as a hipGraph (captured on its second use, launched as ONE graph from then
        on).  The lists are capture-clean by construction -- fixed pointers, no allocation, no host sync; the side-stream
        sections fork and join through events inside the list, which is the capture-legal pattern.  Not used under data
        parallelism (the bucket all-reduces are issued from inside the list).  Off by default: the host already runs a full
        step ahead of the device, so graph launch changes the step time only by the start-of-step gap."""
        graphs = cache.setdefault("_graphs", {})
        g = graphs.get(tag)
        if g is None:
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                ops.replay(lst[0])
            graphs[tag] = g
        g.replay()
        return lst[1]

    # ------------------------------------------------------------------ encoder
    def encoder_forward(self, w: NS):
        """modeling_pretrain.py:83-101 over the VISIBLE tokens only; returns bf16 [B*n_vis, enc_dim] (after encoder.norm)."""
        d, s, p = self.d, self.store, self.enc_prefix
        if getattr(w, "src_u8", False):
            ops.patch_gather_u8(w.frames_u8, d.tubelet, d.patch_size, w.vis_idx, w.xp)
        else:
            ops.patch_gather(w.clips, d.tubelet, d.patch_size, w.vis_idx, w.xp)
        ops.gemm(ops.GEMM_NT, ops.EPI_POS_BF16 if w.enc_x0.dtype == BF16 else ops.EPI_POS_F32, w.xp, s.bview(p + "patch_embed.proj.weight"), w.enc_x0,
                 bias=s.view(p + "patch_embed.proj.bias"), pos=self.pos_enc, row_idx=w.vis_idx.view(-1), rows_in=w.Me, rows_out=w.Me)
        x = w.enc_x0
        for W, L in zip(self.encW, w.enc):
            x = self._block_fwd(W, L, x, w.B, w.n_vis, d.enc_heads)
        ops.layernorm_fwd(x, s.view(p + "norm.weight"), s.view(p + "norm.bias"), d.eps, w.enc_out, w.enc_mean, w.enc_rstd)
        return w.enc_out

    def encoder_backward(self, w: NS, d_out_bf16: torch.Tensor):
        d, s, p, S = self.d, self.store, self.enc_prefix, w.enc_s
        x_last = w.enc[-1].x_out if w.enc else w.enc_x0
        S.used, S.gidx, S.gcount = [False, False], 0, 0
        S.group = self.wgrad_blocks          # (the scratch holds groups of up to _enc_group_cap blocks; set_enc_plan may have changed the size)
        self._ln_bwd(d_out_bf16, x_last, s.view(p + "norm.weight"), w.enc_mean, w.enc_rstd, None, None, S.ring[0],
                     s.gview(p + "norm.weight"), s.gview(p + "norm.bias"))
        seg = 1 if self.dec_prefix is not None else 0
        j = 0
        # bucket boundaries (block index after which a gradient range is complete), as plan_segments laid them out
        ends, i_end = set(), d.enc_depth
        for nb in self.enc_buckets()[:-1]:
            i_end -= nb
            ends.add(i_end)
        R = len(S.ring)
        for i in range(d.enc_depth - 1, -1, -1):
            x_in = w.enc[i - 1].x_out if i > 0 else w.enc_x0
            # a gradient bucket's weight gradients must be complete when its range is handed to the all-reduce; without a
            # bucket consumer (one process) the groups of three run on across the bucket boundaries
            self._block_bwd(self.encW[i], w.enc[i], S, j, x_in, w.B, w.n_vis, d.enc_heads,
                            flush=(i in ends and self.segment_hook is not None), hold=(i == 0))
            j += 1
            if i in ends:
                if self.segment_hook is not None:      # a bucket consumer needs the range complete here; otherwise the LayerNorm
                    self._ln_flush()                   # partials wait for the ONE reduce launch at the end of the backward
                self._seg(seg)
                seg += 1
        # the patch-embed weight gradient (72 tiles alone) rides in the last blocks' grouped launch
        S.pending.append((S.ring[j % R], w.xp, s.g2d(p + "patch_embed.proj.weight"), s.gview(p + "patch_embed.proj.bias"), (0, 0)))
        self._wgrad_flush(S, S.gidx % 2, w.n_vis)
        S.gidx, S.gcount = S.gidx + 1, 0
        self._ln_flush()
        self._join_side(S)
        self._seg(seg)

    # ------------------------------------------------------------------ bridge
    def bridge_forward(self, w: NS, enc_out_bf16: torch.Tensor):
        """modeling_pretrain.py:256-263: encoder_to_decoder (no bias) + pos for the visible half, mask_token + pos for the rest."""
        d, s = self.d, self.store
        if getattr(w, "dec_share", False):
            # cat rows: the visible tokens' rows clip by clip, then ONE row per position (mask_token + pos[j]); the whole-sequence
            # input the decoder's residual path reads is gathered from them (bit-identical to writing every row on its own)
            Z = w.dec0
            ops.gemm(ops.GEMM_NT, ops.EPI_POS_BF16, enc_out_bf16, s.bview("encoder_to_decoder.weight"), Z.xcat[:w.Me],
                     pos=self.pos_dec, row_idx=w.vis_idx.view(-1), rows_in=w.Me, rows_out=w.Me)
            ops.fill_mask_tokens(s.view("mask_token").view(-1), self.pos_dec, Z.pos_idx, 0, Z.xcat[w.Me:].view(1, w.N, d.dec_dim))
            ops.dec0_gather(Z.xcat, w.msk_idx, w.N, w.x_full.view(w.Md, d.dec_dim))
            ops.dec0_inverse(w.msk_idx, w.N, Z.inv)
            return w.x_full
        ops.gemm(ops.GEMM_NT, ops.EPI_POS_BF16 if w.x_full.dtype == BF16 else ops.EPI_POS_F32, enc_out_bf16,
                 s.bview("encoder_to_decoder.weight"), w.x_full.view(w.Md, d.dec_dim),
                 pos=self.pos_dec, row_idx=w.vis_idx.view(-1), rows_in=w.n_vis, rows_out=w.N, row_off=0)
        ops.fill_mask_tokens(s.view("mask_token").view(-1), self.pos_dec, w.msk_idx, w.n_vis, w.x_full)
        return w.x_full

    def bridge_backward(self, w: NS, dx_full: torch.Tensor, enc_out_bf16: torch.Tensor):
        d, s = self.d, self.store
        gmask = s.gview("mask_token").view(-1)
        if getattr(w, "dec_share", False):
            # dx_full holds the gradient per CAT row: the visible rows are d(encoder_to_decoder output), the position rows' column
            # sum is d(mask_token) (each already summed over the clips that mask the position)
            d_e2d = dx_full[:w.Me]
            ops.colsum_bf16(dx_full[w.Me:], gmask)
            ops.gemm(ops.GEMM_NN, ops.EPI_BF16, d_e2d, s.bview("encoder_to_decoder.weight"), w.d_encout)
            self._wgrad(d_e2d, enc_out_bf16, s.g2d("encoder_to_decoder.weight"))
            return w.d_encout
        if self.side is not None and d.dec_dim % 8 == 0 and d.dec_dim <= 512:
            # d(mask_token) is needed by nothing before the optimizer (or the bucket hand-over, which is issued on the side stream
            # too): its 24-block reduce goes to the SIDE stream.  On the main stream it sat in the dependent chain behind a CU-filling
            # weight-gradient launch and took ~180 us for 27 us of work (profiles/r02_rocprof_kernel_stats.csv).
            ops.assemble_bwd(dx_full.view(w.B, w.N, d.dec_dim), w.n_vis, w.d_e2d, None, partial_ws=w.asm_partial)
            side, ev = self.side, self._asm_ready
            self._side_launched = True
            ops.host_op(lambda: ev.record(torch.cuda.current_stream()))
            ops.use_stream(side)
            ops.host_op(lambda: side.wait_event(ev))
            ops.assemble_bwd_finalize(w.asm_partial, w.B, w.N, gmask)
            ops.use_stream(None)
        else:
            ops.assemble_bwd(dx_full.view(w.B, w.N, d.dec_dim), w.n_vis, w.d_e2d, gmask, partial_ws=w.asm_partial)
        ops.gemm(ops.GEMM_NN, ops.EPI_BF16, w.d_e2d, s.bview("encoder_to_decoder.weight"), w.d_encout)
        self._wgrad(w.d_e2d, enc_out_bf16, s.g2d("encoder_to_decoder.weight"))
        return w.d_encout

    # ------------------------------------------------------------------ decoder
    def decoder_forward(self, w: NS, x_full: torch.Tensor, n_ret: int):
        """modeling_pretrain.py:152-161; x_full [B, N, dec_dim] (bf16, or f32 with MOFO_DEC_RESID=f32); returns bf16
        predictions [B*n_ret, patch_out]."""
        d, s, p = self.d, self.store, self.dec_prefix
        x = x_full.view(w.Md, d.dec_dim)
        compact = getattr(w, "dec_compact", False) and n_ret == w.n_msk
        for i, (W, L) in enumerate(zip(self.decW, w.dec)):
            if getattr(L, "compact", False):
                if not compact:
                    raise ValueError("this workspace's last decoder block keeps the masked tokens only: return_token_num must be their count")
                x = self._block_fwd_last(W, L, x, w.B, w.N, d.dec_heads, w.N - n_ret)
            else:
                x = self._block_fwd(W, L, x, w.B, w.N, d.dec_heads, share=w.dec0 if i == 0 and getattr(w, "dec_share", False) else None)
        if compact:      # the last block's output holds exactly the rows decoder.norm / head read
            ops.layernorm_fwd(x, s.view(p + "norm.weight"), s.view(p + "norm.bias"), d.eps, w.dec_ln, w.dec_mean, w.dec_rstd)
        else:
            ops.layernorm_fwd(x, s.view(p + "norm.weight"), s.view(p + "norm.bias"), d.eps, w.dec_ln, w.dec_mean, w.dec_rstd,
                              rows_in=n_ret, rows_out=w.N, row_off=w.N - n_ret)
        ops.gemm(ops.GEMM_NT, ops.EPI_BF16, w.dec_ln, s.bview(p + "head.weight"), w.pred, bias=s.view(p + "head.bias"))
        return w.pred

    def decoder_backward(self, w: NS, dpred_bf16: torch.Tensor, x_full: torch.Tensor, n_ret: int, defer_ln: bool = False):
        d, s, p, S = self.d, self.store, self.dec_prefix, w.dec_s
        ops.gemm(ops.GEMM_NN, ops.EPI_BF16, dpred_bf16, s.bview(p + "head.weight"), w.d_decln)
        x_last = w.dec[-1].x_out if w.dec else x_full.view(w.Md, d.dec_dim)
        S.used, S.gidx, S.gcount = [False, False], 0, 0
        # blocks per weight-gradient launch of THIS recording: all of them (one sliced launch at the end of the pass), or MOFO_WGRAD_BLOCKS_DEC
        S.group = max(1, d.dec_depth) if self.wgrad_sliced else self._blocks_dec_env
        # the head's weight gradient (36 tiles) joins the first decoder block's grouped launch on the side stream
        S.pending.append((dpred_bf16, w.dec_ln, s.g2d(p + "head.weight"), s.gview(p + "head.bias"), (0, 0)))
        compact = bool(w.dec) and getattr(w.dec[-1], "compact", False)
        j = 0
        top = d.dec_depth - 1
        if compact:
            # the last block's rows of the visible tokens do not exist: everything down to its attention stays on the n_ret rows
            self._ln_bwd(w.d_decln, x_last, s.view(p + "norm.weight"), w.dec_mean, w.dec_rstd, None, None, w.dec_c.dx0,
                         s.gview(p + "norm.weight"), s.gview(p + "norm.bias"))
            x_in = w.dec[top - 1].x_out if top > 0 else x_full.view(w.Md, d.dec_dim)
            self._block_bwd_last(self.decW[top], w.dec[top], S, w.dec_c, x_in, w.B, w.N, d.dec_heads, w.N - n_ret, flush=(top == 0))
            j, top = 1, top - 1
        else:
            # rows of the visible tokens get no gradient from the head (x[:, -n_ret:], modeling_pretrain.py:157)
            # (only those rows: the LayerNorm backward below writes the other n_ret rows of every clip)
            head_zero = S.ring[0].view(w.B, w.N, d.dec_dim)[:, :w.N - n_ret]
            ops.host_op(lambda: head_zero.zero_())
            self._ln_bwd(w.d_decln, x_last, s.view(p + "norm.weight"), w.dec_mean, w.dec_rstd, None, None, S.ring[0],
                         s.gview(p + "norm.weight"), s.gview(p + "norm.bias"), rows_in=n_ret, rows_out=w.N, row_off=w.N - n_ret)
        for i in range(top, -1, -1):
            x_in = w.dec[i - 1].x_out if i > 0 else x_full.view(w.Md, d.dec_dim)
            share = w.dec0 if i == 0 and getattr(w, "dec_share", False) else None
            self._block_bwd(self.decW[i], w.dec[i], S, j, x_in, w.B, w.N, d.dec_heads, flush=(i == 0), share=share)
            j += 1
        if getattr(w, "dec_share", False):
            if not defer_ln:
                self._ln_flush()
                if self.segment_hook is None:
                    self._join_side(S)
            return w.dec0.dxcat                  # gradient wrt the CAT rows of the decoder input, bf16 [B*n_vis + N, D]
        if S.pending:                      # a decoder without blocks: the head's weight gradient alone
            self._wgrad_flush(S, S.gidx % 2, w.N)
        if not defer_ln:                   # defer_ln: the caller runs the encoder backward next; its final LayerNorm reduce and its
            self._ln_flush()               # join of the (one, in-order) side stream cover this pass's launches as well
            if self.segment_hook is None:
                self._join_side(S)
        return S.ring[j % len(S.ring)]       # gradient wrt the decoder input, bf16 [B*N, D]

    # ------------------------------------------------------------------ whole model
    def _forward(self, w: NS, frozen: bool = False):
        enc_out = self.encoder_forward(w)
        x_full = self.bridge_forward(w, enc_out)
        out = self.decoder_forward(w, x_full, w.n_msk)
        if self.fp8 and not frozen:
            ops.fp8_update_scales(self.act_amax, self.act_scales)      # this forward's amax -> the next forward's scales
        return out

    def forward(self, w: NS):
        if self.fp8:
            self.store.refresh_shadow8()
            if not self._fp8_calibrated:       # delayed scaling needs one look at the activations before the first real step
                self._fp8_calibrated = True
                self._forward(w)
        # fp8_freeze (dist.GradSync.value_check): forwards that must quantise with the SAME delayed activation scales leave them alone
        frozen = bool(self.fp8 and getattr(self, "fp8_freeze", False))
        return self.cached(w, ("fwd", getattr(w, "src_u8", False), frozen), lambda: self._forward(w, frozen))

    def fp8_state_dict(self):
        """the delayed activation scales of the e4m3 forward (one (scale, 1 / scale) pair per site), or None without MOFO_FP8: what a
        checkpoint has to carry for a resumed run to quantise its first forward like the run that wrote it (utils.save_model).  The
        weights' e4m3 shadow is not state: the first forward after load_state_dict re-quantises it from the bf16 shadow with exact scales."""
        if not self.fp8 or not self._fp8_calibrated:
            return None
        return {"act_scales": self.act_scales.detach().cpu().clone()}

    def load_fp8_state_dict(self, sd):
        if not self.fp8 or not sd:
            return
        a = sd["act_scales"]
        if tuple(a.shape) != tuple(self.act_scales.shape):
            raise ValueError(f"fp8 activation scales: checkpoint has {tuple(a.shape)}, this model {tuple(self.act_scales.shape)}")
        self.act_scales.copy_(a.to(self.act_scales.dtype))
        self.act_amax.zero_()
        self._fp8_calibrated = True      # no calibration forward: the scales are the ones the interrupted run would have used next

    def _loss_forward(self, w, normalize_target, grad_scale):
        d = self.d
        if getattr(w, "src_u8", False):
            ops.target_mse_u8(w.frames_u8, d.tubelet, d.patch_size, w.msk_idx, w.pred, normalize_target, grad_scale, w.row_loss, w.loss, w.dpred)
        else:
            ops.target_mse(w.clips, d.tubelet, d.patch_size, w.msk_idx, w.pred, normalize_target, grad_scale, w.row_loss, w.loss, w.dpred)
        return w.loss

    def loss_forward(self, w: NS, normalize_target: bool = True, grad_scale: float = 1.0):
        """engine_for_pretraining.py:43-67 fused: target build + MSE + d(loss)/d(pred) in one pass."""
        return self.cached(w, ("loss", bool(normalize_target), float(grad_scale), getattr(w, "src_u8", False)),
                           lambda: self._loss_forward(w, normalize_target, grad_scale))

    def _backward(self, w: NS):
        self._side_launched = False
        dx_full = self.decoder_backward(w, w.dpred, w.x_full, w.n_msk, defer_ln=self.segment_hook is None)
        d_encout = self.bridge_backward(w, dx_full, w.enc_out)
        self._seg(0)
        self.encoder_backward(w, d_encout)
        if self._side_launched and self.side is not None:
            # The decoder pass leaves its side-stream launches to the encoder's final join (defer_ln).  That join only exists when
            # the encoder's own groups ran on the side stream: with MOFO_WGRAD_STREAM=main_enc nothing waited for the decoder's
            # weight-gradient GEMMs, and grad-norm / AdamW / the next step's scratch rewrite raced with them.  One event at the
            # very end of the (in-order) side stream covers every launch of this backward, whatever the stream modes were.
            side, ev = self.side, self._bwd_done
            ops.host_op(lambda: (ev.record(side), torch.cuda.current_stream().wait_event(ev)))

    def begin_backward(self):
        """decide overwrite vs accumulate for this backward's weight gradients (zero_grad since the last backward?)"""
        st = self.store
        self._accumulate = not st.fresh
        st.fresh = False

    def backward(self, w: NS):
        self.begin_backward()
        self.cached(w, ("bwd", self._accumulate, self.segment_hook is not None), lambda: self._backward(w))

    # ------------------------------------------------------------------ optimizer side
    def grad_norm(self) -> torch.Tensor:
        """utils.py:376-388: global L2 norm of all gradients; stays on the device."""
        ops.sumsq_norm(self.store.grads, self.norm_partial, self.norm_out)
        return self.norm_out

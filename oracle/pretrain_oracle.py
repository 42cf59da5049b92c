"""fp32 CPU restatement of the reference pretraining step (the ORACLE).

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  The product path (``mofo_amd``) never
imports this file; it exists so that the HIP path can be checked against the
reference's arithmetic on a box where ``/root/reference`` does not exist.

Every function cites the reference lines (relative to ``/root/reference``) whose
arithmetic it restates.  It is a *functional* restatement: parameters live in a
plain ``dict`` keyed by the reference's ``state_dict`` names, the
forward is written with elementary torch CPU ops (matmul / exp / sum), and
gradients come from torch autograd over that forward.

Pinning: the reference ships no tests or golden vectors, so this
oracle is pinned by fixtures generated from the reference itself, imported in
the build container by ``tools/make_goldens.py`` -> ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` holds the oracle to them.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)  # timm.data.constants, used at engine_for_pretraining.py:45
IMAGENET_STD = (0.229, 0.224, 0.225)   # engine_for_pretraining.py:46


# --------------------------------------------------------------------------- config
@dataclass
class OracleConfig:
    """Shape of one PretrainVisionTransformer (modeling_pretrain.py:166-190)."""
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
    ln_eps: float = 1e-6

    @property
    def grid(self) -> Tuple[int, int, int]:
        g = self.img_size // self.patch_size
        return (self.num_frames // self.tubelet, g, g)

    @property
    def num_patches(self) -> int:
        t, h, w = self.grid
        return t * h * w

    @property
    def patch_dim(self) -> int:  # decoder_num_classes, modeling_pretrain.py:112
        return self.in_chans * self.tubelet * self.patch_size ** 2


VIT_B = OracleConfig()
VIT_L32 = OracleConfig(num_frames=32, enc_dim=1024, enc_depth=24, enc_heads=16, dec_dim=512, dec_heads=8)
TINY = OracleConfig(img_size=32, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=64, dec_depth=1, dec_heads=1)


# --------------------------------------------------------------------------- masks
def tube_mask(grid: Tuple[int, int, int], mask_ratio: float, rng=np.random) -> np.ndarray:
    """masking_generator.py:3-24.  One shuffled per-frame 0/1 pattern repeated over
    the temporal slots.  ``rng`` defaults to numpy's GLOBAL generator, as the reference."""
    frames, h, w = grid
    per_frame = h * w
    n_mask = int(mask_ratio * per_frame)
    pattern = np.concatenate([np.zeros(per_frame - n_mask), np.ones(n_mask)])
    rng.shuffle(pattern)
    return np.tile(pattern, (frames, 1)).reshape(-1)


def bb_mask(grid: Tuple[int, int, int], mask_ratio: float, mask_ratio_bb: float, bb, rng=np.random) -> np.ndarray:
    """masking_generator.py:27-85 with its quirks kept: only ``bb[0]`` is looked at (:46,55),
    x is compared against the ROW index j and y against the column k (:50-55), the predicate is
    ``not((x1>16j+16 or x2<16j) and (y1>16k+16 or y2<16k))`` (a cross-shaped set), and the random
    fill draws from ``arange(num_masks_per_frame)`` rather than all patches (:72)."""
    frames, h, w = grid
    per_frame = h * w
    n_mask = int(mask_ratio * per_frame)
    x1, y1, x2, y2 = (bb[0][i] for i in range(4))
    inbox: List[int] = []
    for j in range(h):
        for k in range(w):
            x_miss = (x1 > 16 * j + 16) or (x2 < 16 * j)
            y_miss = (y1 > 16 * k + 16) or (y2 < 16 * k)
            if not (x_miss and y_miss):
                inbox.append(j * w + k)
    rng.shuffle(inbox)
    forced = inbox[: min(n_mask, int(len(inbox) * mask_ratio_bb))]
    pattern = np.zeros(per_frame)
    pattern[forced] = 1
    rest = np.setdiff1d(np.arange(n_mask), forced)
    rng.shuffle(rest)
    pattern[rest[: n_mask - len(forced)]] = 1
    return np.tile(pattern, (frames, 1)).reshape(-1)


# --------------------------------------------------------------------------- tables / schedules
def sincos_table(n_pos: int, dim: int) -> torch.Tensor:
    """modeling_finetune.py:252-262: angle[p,j] = p / 10000^(2*(j//2)/dim) in float64,
    sin on even j, cos on odd j, cast to float32, shape [1, n_pos, dim]."""
    j = np.arange(dim)
    denom = np.power(10000, 2 * (j // 2) / dim)
    ang = np.arange(n_pos, dtype=np.float64)[:, None] / denom[None, :]
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.from_numpy(ang.astype(np.float32)).unsqueeze(0)


def cosine_schedule(base: float, final: float, epochs: int, niter: int, warmup_epochs: int = 0,
                    start_warmup: float = 0.0, warmup_steps: int = -1) -> np.ndarray:
    """utils.py:391-408."""
    warm = warmup_epochs * niter
    if warmup_steps > 0:
        warm = warmup_steps
    head = np.linspace(start_warmup, base, warm) if warmup_epochs > 0 else np.array([])
    n = epochs * niter - warm
    tail = np.array([final + 0.5 * (base - final) * (1 + math.cos(math.pi * i / n)) for i in range(n)])
    out = np.concatenate((head, tail))
    assert len(out) == epochs * niter
    return out


# --------------------------------------------------------------------------- deterministic fixtures
def param_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict schema of PretrainVisionTransformer (modeling_pretrain.py:192-234,
    modeling_finetune.py:66-75,39-41,200-208,238-240).  Insertion order = named_parameters() order."""
    s: Dict[str, Tuple[int, ...]] = {}
    s["mask_token"] = (1, 1, cfg.dec_dim)
    s["encoder.patch_embed.proj.weight"] = (cfg.enc_dim, cfg.in_chans, cfg.tubelet, cfg.patch_size, cfg.patch_size)
    s["encoder.patch_embed.proj.bias"] = (cfg.enc_dim,)

    def blocks(prefix: str, depth: int, d: int):
        hid = int(d * cfg.mlp_ratio)
        for i in range(depth):
            p = f"{prefix}.blocks.{i}."
            s[p + "norm1.weight"] = (d,)
            s[p + "norm1.bias"] = (d,)
            s[p + "attn.q_bias"] = (d,)
            s[p + "attn.v_bias"] = (d,)
            s[p + "attn.qkv.weight"] = (3 * d, d)
            s[p + "attn.proj.weight"] = (d, d)
            s[p + "attn.proj.bias"] = (d,)
            s[p + "norm2.weight"] = (d,)
            s[p + "norm2.bias"] = (d,)
            s[p + "mlp.fc1.weight"] = (hid, d)
            s[p + "mlp.fc1.bias"] = (hid,)
            s[p + "mlp.fc2.weight"] = (d, hid)
            s[p + "mlp.fc2.bias"] = (d,)

    blocks("encoder", cfg.enc_depth, cfg.enc_dim)
    s["encoder.norm.weight"] = (cfg.enc_dim,)
    s["encoder.norm.bias"] = (cfg.enc_dim,)
    blocks("decoder", cfg.dec_depth, cfg.dec_dim)
    s["decoder.norm.weight"] = (cfg.dec_dim,)
    s["decoder.norm.bias"] = (cfg.dec_dim,)
    s["decoder.head.weight"] = (cfg.patch_dim, cfg.dec_dim)
    s["decoder.head.bias"] = (cfg.patch_dim,)
    s["encoder_to_decoder.weight"] = (cfg.dec_dim, cfg.enc_dim)
    return s


def keyed_normal(key: str, shape: Sequence[int]) -> np.ndarray:
    """Name-keyed deterministic N(0,1) draw: independent of construction order
    and of torch's RNG; numpy's legacy RandomState stream is frozen."""
    return np.random.RandomState(zlib.crc32(key.encode())).standard_normal(tuple(shape)).astype(np.float32)


def keyed_params(cfg: OracleConfig, mode: str = "small") -> Dict[str, torch.Tensor]:
    """Deterministic weights.  mode 'small': 0.02*n everywhere, LN weights 1+0.02*n (the fill the
    survey validated against the reference).  mode 'xavier': matrices at xavier/kaiming scale so that
    attention logits and activations have realistic magnitude."""
    out: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        n = keyed_normal(name, shape)
        is_ln_w = name.endswith(("norm1.weight", "norm2.weight", "norm.weight"))
        if mode == "small":
            v = 1.0 + 0.02 * n if is_ln_w else 0.02 * n
        elif mode == "xavier":
            if is_ln_w:
                v = 1.0 + 0.1 * n
            elif len(shape) == 2:
                v = n * math.sqrt(2.0 / (shape[0] + shape[1]))
            elif len(shape) == 5:
                v = n / math.sqrt(shape[1] * shape[2] * shape[3] * shape[4])
            else:
                v = 0.02 * n
        else:
            raise ValueError(mode)
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out


def keyed_clips(n: int, cfg: OracleConfig, base_seed: int = 1000) -> torch.Tensor:
    """clips[i] = RandomState(base_seed+i).standard_normal([3,T,H,W])."""
    shape = (cfg.in_chans, cfg.num_frames, cfg.img_size, cfg.img_size)
    return torch.from_numpy(np.stack(
        [np.random.RandomState(base_seed + i).standard_normal(shape).astype(np.float32) for i in range(n)]))


def ingest_uint8(frames: torch.Tensor) -> torch.Tensor:
    """uint8 [B,H,W,T*3] (Stack(), transforms.py:346-360) -> f32 [B,3,T,H,W]: ToTorchFormatTensor(div=True)
    (transforms.py:363-382: permute to [T*3,H,W], .float().div(255.)), GroupNormalize (datasets.py:12-14: sub mean, div std
    per channel, mean/std repeated per frame) and view(T,3,H,W).transpose(0,1) (kinetics.py:492-493)."""
    B, H, W, TC = frames.shape
    T = TC // 3
    x = frames.permute(0, 3, 1, 2).contiguous().float().div(255.)                 # [B, T*3, H, W]
    mean = torch.tensor(IMAGENET_MEAN * T, dtype=torch.float32)[None, :, None, None]
    std = torch.tensor(IMAGENET_STD * T, dtype=torch.float32)[None, :, None, None]
    x = x.sub(mean).div(std)
    return x.view(B, T, 3, H, W).transpose(1, 2).contiguous()


# --------------------------------------------------------------------------- forward pieces
# LIBRARY_OPS (tests/eager_gpu_baseline.py only): the same forward through the torch library ops the reference's nn.Modules call
# (F.linear / F.layer_norm / F.gelu / softmax) instead of the elementary ops below -- what "the reference under PyTorch" costs on a
# device, not a different arithmetic (tests/test_oracle_golden.py holds the two forms together on CPU).  Works on any device.
LIBRARY_OPS = False


def _linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    if LIBRARY_OPS:
        return torch.nn.functional.linear(x, w, b)
    y = x @ w.t()
    return y if b is None else y + b


def _layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    """nn.LayerNorm: biased variance over the last dim, eps inside the sqrt."""
    if LIBRARY_OPS:
        return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, eps)
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w + b


def _gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """nn.GELU() default = exact erf form (modeling_finetune.py:35,40)."""
    if LIBRARY_OPS:
        return torch.nn.functional.gelu(x)
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def patchify_tubelets(x: torch.Tensor, cfg: OracleConfig) -> torch.Tensor:
    """[B,C,T,H,W] -> [B, N, C*pt*ph*pw]: row n = t*Hg*Wg + h*Wg + w (flatten(2) order,
    modeling_finetune.py:247), columns in Conv3d weight order (c, p0, p1, p2)."""
    B, C, T, H, W = x.shape
    pt, p = cfg.tubelet, cfg.patch_size
    x = x.reshape(B, C, T // pt, pt, H // p, p, W // p, p)
    x = x.permute(0, 2, 4, 6, 1, 3, 5, 7)
    return x.reshape(B, (T // pt) * (H // p) * (W // p), C * pt * p * p)


def patch_embed(x: torch.Tensor, P: Dict[str, torch.Tensor], cfg: OracleConfig) -> torch.Tensor:
    """modeling_finetune.py:238-248: Conv3d with kernel == stride is one dot product per tubelet."""
    w = P["encoder.patch_embed.proj.weight"].reshape(cfg.enc_dim, -1)
    return _linear(patchify_tubelets(x, cfg), w, P["encoder.patch_embed.proj.bias"])


def attention(x: torch.Tensor, P: Dict[str, torch.Tensor], pre: str, heads: int) -> torch.Tensor:
    """modeling_finetune.py:78-98: fused QKV with bias cat(q_bias, 0, v_bias); q scaled by
    head_dim**-0.5 BEFORE q@k^T; softmax over keys; proj with bias."""
    B, N, D = x.shape
    hd = D // heads
    qb, vb = P[pre + "attn.q_bias"], P[pre + "attn.v_bias"]
    bias = torch.cat((qb, torch.zeros_like(vb), vb))
    qkv = _linear(x, P[pre + "attn.qkv.weight"], bias).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    s = q @ k.transpose(-2, -1)
    if LIBRARY_OPS:
        p = s.softmax(dim=-1)
    else:
        s = s - s.max(dim=-1, keepdim=True).values
        e = torch.exp(s)
        p = e / e.sum(dim=-1, keepdim=True)
    o = (p @ v).transpose(1, 2).reshape(B, N, D)
    return _linear(o, P[pre + "attn.proj.weight"], P[pre + "attn.proj.bias"])


def block(x: torch.Tensor, P: Dict[str, torch.Tensor], pre: str, heads: int, eps: float) -> torch.Tensor:
    """modeling_finetune.py:216-219 (gamma_1 is None because init_values=0., modeling_pretrain.py:185;
    DropPath is Identity at rate 0)."""
    x = x + attention(_layernorm(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"], eps), P, pre, heads)
    h = _linear(_layernorm(x, P[pre + "norm2.weight"], P[pre + "norm2.bias"], eps),
                P[pre + "mlp.fc1.weight"], P[pre + "mlp.fc1.bias"])
    return x + _linear(_gelu_erf(h), P[pre + "mlp.fc2.weight"], P[pre + "mlp.fc2.bias"])


def encoder_forward(x: torch.Tensor, mask: torch.Tensor, P: Dict[str, torch.Tensor], cfg: OracleConfig,
                    taps: Optional[dict] = None) -> torch.Tensor:
    """modeling_pretrain.py:83-101."""
    tok = patch_embed(x, P, cfg) + sincos_table(cfg.num_patches, cfg.enc_dim).to(x.device)
    B, _, C = tok.shape
    xv = tok[~mask].reshape(B, -1, C)
    if taps is not None:
        taps["x_vis0"] = xv
    for i in range(cfg.enc_depth):
        xv = block(xv, P, f"encoder.blocks.{i}.", cfg.enc_heads, cfg.ln_eps)
        if taps is not None:
            taps[f"enc_block{i}"] = xv
    return _layernorm(xv, P["encoder.norm.weight"], P["encoder.norm.bias"], cfg.ln_eps)


def decoder_forward(x: torch.Tensor, n_ret: int, P: Dict[str, torch.Tensor], cfg: OracleConfig,
                    taps: Optional[dict] = None) -> torch.Tensor:
    """modeling_pretrain.py:152-161."""
    for i in range(cfg.dec_depth):
        x = block(x, P, f"decoder.blocks.{i}.", cfg.dec_heads, cfg.ln_eps)
        if taps is not None:
            taps[f"dec_block{i}"] = x
    if n_ret > 0:
        x = x[:, -n_ret:]
    return _linear(_layernorm(x, P["decoder.norm.weight"], P["decoder.norm.bias"], cfg.ln_eps),
                   P["decoder.head.weight"], P["decoder.head.bias"])


def model_forward(x: torch.Tensor, mask: torch.Tensor, P: Dict[str, torch.Tensor], cfg: OracleConfig,
                  taps: Optional[dict] = None) -> torch.Tensor:
    """modeling_pretrain.py:253-266.  ``mask`` bool [B,N], True = masked; every clip must keep the
    same number of visible tokens (the reference's reshape at :90 requires it)."""
    xv = encoder_forward(x, mask, P, cfg, taps)
    if taps is not None:
        taps["enc_out"] = xv
    xv = _linear(xv, P["encoder_to_decoder.weight"])
    B, _, C = xv.shape
    pos = sincos_table(cfg.num_patches, cfg.dec_dim).to(xv.device).expand(B, -1, -1)
    pos_vis = pos[~mask].reshape(B, -1, C)
    pos_msk = pos[mask].reshape(B, -1, C)
    full = torch.cat([xv + pos_vis, P["mask_token"] + pos_msk], dim=1)
    if taps is not None:
        taps["x_full"] = full
    return decoder_forward(full, pos_msk.shape[1], P, cfg, taps)


def build_targets(x: torch.Tensor, mask: torch.Tensor, cfg: OracleConfig, normalize: bool = True) -> torch.Tensor:
    """engine_for_pretraining.py:43-63: un-normalise with the ImageNet constants, cut into tubelets with
    feature order (p0 p1 p2) c, per (token, channel) standardise over the 512 pixels with the UNBIASED
    variance and 1e-6 added AFTER the sqrt, flatten to (p c) and keep the masked tokens."""
    mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype, device=x.device)[None, :, None, None, None]
    std = torch.tensor(IMAGENET_STD, dtype=x.dtype, device=x.device)[None, :, None, None, None]
    u = x * std + mean
    B, C, T, H, W = u.shape
    pt, p = cfg.tubelet, cfg.patch_size
    u = u.reshape(B, C, T // pt, pt, H // p, p, W // p, p).permute(0, 2, 4, 6, 3, 5, 7, 1)
    u = u.reshape(B, cfg.num_patches, pt * p * p, C)
    if normalize:
        mu = u.mean(dim=-2, keepdim=True)
        var = ((u - mu) ** 2).sum(dim=-2, keepdim=True) / (u.shape[-2] - 1)
        u = (u - mu) / (var.sqrt() + 1e-6)
    u = u.reshape(B, cfg.num_patches, -1)
    return u[mask].reshape(B, -1, u.shape[-1])


def reconstruct_video(x: torch.Tensor, mask: torch.Tensor, outputs: torch.Tensor, cfg: OracleConfig):
    """run_videomae_vis.py:150-176: (ori_img, rec_img, img_mask), each [B,3,T,H,W] in [0,1] pixel units.
    ori = x*std+mean (:152); every token standardised per channel over its 512 pixels (unbiased variance, 1e-6 after the
    sqrt, :158-159); masked tokens replaced by the model's predictions (:161); multiplied back by that token's own
    (std+1e-6) and mean (:172) -- so visible tokens come back as the original pixels up to rounding and masked ones as
    de-standardised predictions; img_mask = rec * (1 on visible tokens, 0 on masked) (:163-167,180).  The .clamp(0, 0.996)
    of :174 belongs to the JPEG writer, not to this arithmetic."""
    mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype)[None, :, None, None, None]
    std = torch.tensor(IMAGENET_STD, dtype=x.dtype)[None, :, None, None, None]
    ori = x * std + mean
    B, C, T, H, W = ori.shape
    pt, p = cfg.tubelet, cfg.patch_size
    gt, gh, gw = T // pt, H // p, W // p
    sq = ori.reshape(B, C, gt, pt, gh, p, gw, p).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(B, gt * gh * gw, pt * p * p, C)
    mu = sq.mean(dim=-2, keepdim=True)
    sd = (((sq - mu) ** 2).sum(dim=-2, keepdim=True) / (sq.shape[-2] - 1)).sqrt() + 1e-6
    patch = ((sq - mu) / sd).reshape(B, gt * gh * gw, -1).clone()
    patch[mask] = outputs.reshape(-1, patch.shape[-1]).to(patch.dtype)
    keep = torch.ones_like(patch)
    keep[mask] = 0

    def unpatch(t):   # 'b (t h w) (p0 p1 p2) c -> b c (t p0) (h p1) (w p2)'
        t = t.reshape(B, gt, gh, gw, pt, p, p, C).permute(0, 7, 1, 4, 2, 5, 3, 6)
        return t.reshape(B, C, T, H, W)
    rec = unpatch(patch.reshape(B, -1, pt * p * p, C) * sd + mu)
    return ori, rec, rec * unpatch(keep.reshape(B, -1, pt * p * p, C))


# --------------------------------------------------------------------------- fine-tune model forward (next row)
def finetune_param_shapes(cfg: OracleConfig, num_classes: int) -> Dict[str, Tuple[int, ...]]:
    """state_dict schema of modeling_finetune.VisionTransformer with use_mean_pooling=True, init_values=0
    (modeling_finetune.py:328-352): the pretraining encoder's keys without the ``encoder.`` prefix, ``norm`` is
    Identity (no parameters), plus fc_norm and head."""
    s: Dict[str, Tuple[int, ...]] = {}
    for k, v in param_shapes(cfg).items():
        if k.startswith("encoder.") and not k.startswith("encoder.norm."):
            s[k[len("encoder."):]] = v
    s["fc_norm.weight"] = (cfg.enc_dim,)
    s["fc_norm.bias"] = (cfg.enc_dim,)
    s["head.weight"] = (num_classes, cfg.enc_dim)
    s["head.bias"] = (num_classes,)
    return s


def finetune_keyed_params(cfg: OracleConfig, num_classes: int) -> Dict[str, torch.Tensor]:
    """xavier-scale name-keyed weights (as keyed_params mode 'xavier') for the fine-tune model"""
    out: Dict[str, torch.Tensor] = {}
    for name, shape in finetune_param_shapes(cfg, num_classes).items():
        n = keyed_normal("ft." + name, shape)
        if name.endswith(("norm1.weight", "norm2.weight", "norm.weight")):
            v = 1.0 + 0.1 * n
        elif len(shape) == 2:
            v = n * math.sqrt(2.0 / (shape[0] + shape[1]))
        elif len(shape) == 5:
            v = n / math.sqrt(shape[1] * shape[2] * shape[3] * shape[4])
        else:
            v = 0.02 * n
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return out


def finetune_forward(x: torch.Tensor, P: Dict[str, torch.Tensor], cfg: OracleConfig, features_only: bool = False) -> torch.Tensor:
    """modeling_finetune.py:389-409: patch embed over ALL tokens + sincos pos -> blocks -> (norm = Identity) ->
    fc_norm(mean over tokens) -> head."""
    w = P["patch_embed.proj.weight"].reshape(cfg.enc_dim, -1)
    t = _linear(patchify_tubelets(x, cfg), w, P["patch_embed.proj.bias"]) + sincos_table(cfg.num_patches, cfg.enc_dim)
    for i in range(cfg.enc_depth):
        t = block(t, P, f"blocks.{i}.", cfg.enc_heads, cfg.ln_eps)
    f = _layernorm(t.mean(dim=1), P["fc_norm.weight"], P["fc_norm.bias"], cfg.ln_eps)
    return f if features_only else _linear(f, P["head.weight"], P["head.bias"])


def mse_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """nn.MSELoss() (engine_for_pretraining.py:27,67): mean over every element."""
    d = pred - target
    return (d * d).mean()


# --------------------------------------------------------------------------- optimizer side
NO_DECAY_NAMES = {"pos_embed", "cls_token", "mask_token"}  # modeling_pretrain.py:249-251


def is_no_decay(name: str, shape: Sequence[int]) -> bool:
    """optim_factory.py:56-61: 1-D tensors, '.bias' names and the skip list get weight_decay 0."""
    return len(shape) == 1 or name.endswith(".bias") or name in NO_DECAY_NAMES


def grad_norm(grads: Dict[str, torch.Tensor]) -> torch.Tensor:
    """utils.py:376-388 (norm_type 2): norm of the per-tensor norms."""
    return torch.norm(torch.stack([torch.norm(g, 2.0) for g in grads.values()]), 2.0)


@dataclass
class AdamWState:
    step: int = 0
    m: Dict[str, torch.Tensor] = field(default_factory=dict)
    v: Dict[str, torch.Tensor] = field(default_factory=dict)


def adamw_step(P: Dict[str, torch.Tensor], G: Dict[str, torch.Tensor], st: AdamWState, lr: float,
               weight_decay: float, betas=(0.9, 0.95), eps: float = 1e-8) -> None:
    """torch.optim.AdamW as configured by optim_factory.py:91-127 (decoupled decay, bias correction,
    eps added to sqrt(v_hat)); in place on P."""
    st.step += 1
    b1, b2 = betas
    c1 = 1.0 - b1 ** st.step
    c2 = 1.0 - b2 ** st.step
    for name, p in P.items():
        g = G[name]
        if name not in st.m:
            st.m[name] = torch.zeros_like(p)
            st.v[name] = torch.zeros_like(p)
        wd = 0.0 if is_no_decay(name, p.shape) else weight_decay
        p.mul_(1.0 - lr * wd)
        st.m[name].mul_(b1).add_(g, alpha=1.0 - b1)
        st.v[name].mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (st.v[name].sqrt() / math.sqrt(c2)).add_(eps)
        p.addcdiv_(st.m[name], denom, value=-lr / c1)


def clip_grads(grads: Dict[str, torch.Tensor], total_norm: torch.Tensor, max_norm: float) -> None:
    """torch.nn.utils.clip_grad_norm_ as the reference's scaler applies it (utils.py:359): every gradient times
    min(1, max_norm / (total_norm + 1e-6)), in place"""
    coef = torch.clamp(max_norm / (total_norm + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)


def train_step(x: torch.Tensor, mask: torch.Tensor, P: Dict[str, torch.Tensor], cfg: OracleConfig,
               st: Optional[AdamWState] = None, lr: float = 1.5e-4, weight_decay: float = 0.05,
               normalize_target: bool = True, clip_grad: Optional[float] = None):
    """One step of engine_for_pretraining.py:39-69,172-176 on CPU fp32 (the PNG dump at :74-166 and the
    two CUDA-only lines :177,:179 are not part of the arithmetic).  Returns (loss, grad_norm, grads)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    with torch.no_grad():
        labels = build_targets(x, mask, cfg, normalize_target)
    out = model_forward(x, mask, leaves, cfg)
    loss = mse_loss(out, labels)
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items()}
    gn = grad_norm(grads)
    if clip_grad:
        with torch.no_grad():
            clip_grads(grads, gn, clip_grad)       # the returned norm stays the one BEFORE clipping, as the reference reports it
    if st is not None:
        with torch.no_grad():
            adamw_step(P, grads, st, lr, weight_decay)
    return float(loss.detach()), float(gn), grads


# ---------------------------------------------------------------------------------------------- device-side tube masks
def _mix32(x):
    """the "lowbias32" integer finaliser of mofo_amd/csrc/tokens.hip (uint32 arithmetic)"""
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7feb352d) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846ca68b) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def device_tube_masks(seed, counter, B, frames, patches_per_frame, n_mask):
    """CPU restatement of mofo_tube_masks (the build's OWN generator for SURVEY.md 8f rank 3; the reference's tube mask,
    masking_generator.py:3-24, fixes the DISTRIBUTION it must have: n_mask of the patches of a frame, one pattern per clip
    repeated over the frames).  Returns uint8 [B, frames * patches_per_frame], 1 = masked."""
    out = np.zeros((B, frames * patches_per_frame), dtype=np.uint8)
    i = np.arange(patches_per_frame, dtype=np.uint64)
    for c in range(B):
        base = _mix32((seed & 0xFFFFFFFF) ^ int(_mix32((counter + c + 0x9e3779b9) & 0xFFFFFFFF)))
        key = _mix32((int(base) + 0x85ebca6b * (i + 1)) & 0xFFFFFFFF)
        order = np.lexsort((i, key))                     # ascending key, ties by index
        pat = np.zeros(patches_per_frame, dtype=np.uint8)
        pat[order[:n_mask]] = 1
        out[c] = np.tile(pat, frames)
    return out

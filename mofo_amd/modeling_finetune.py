"""Forward (inference / feature extraction) of the reference's fine-tune model on the HIP kernels -- SURVEY.md 8f rank 4,
the first consumer of the pretraining checkpoints.

    VisionTransformer              modeling_finetune.py:305-409   (use_mean_pooling=True, init_values=0: the VideoMAE recipe)
    VisionTransformer_feat_ext     modeling_finetune.py:411-420   (forward = forward_features)
    vit_{small,base,large}_patch16_224, vit_base_patch16_224_feature_ext     modeling_finetune.py:637-688

Same constructor arguments, attribute names and state_dict schema as the reference (the pretraining encoder's keys without
the ``encoder.`` prefix, ``fc_norm.*``, ``head.*``; ``norm`` is Identity), so a pretraining checkpoint loads the way
run_class_finetuning.py:350-411 loads it (``load_pretrained_encoder``).  The torch Modules only hold the parameters; the
forward runs through libmofo_hip.so: tubelet gather over ALL tokens -> patch-embed GEMM (+ sincos pos) -> the same
LayerNorm / GEMM / attention kernels as the pretraining blocks at N = 1568 -> token-mean + fc_norm kernel -> head GEMM.
It is forward-only: outputs carry no autograd graph (training the classifier is outside the pretraining path).
"""
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn

from . import ops
from .modeling_pretrain import Block, PatchEmbed, _block_names, _check_heads, _FlatModule, _ln_eps, _unsupported
from .runtime import BF16, F32, I32, NS, Dims, PretrainRuntime

__all__ = ["VisionTransformer", "VisionTransformer_feat_ext", "vit_small_patch16_224", "vit_base_patch16_224",
           "vit_base_patch16_224_feature_ext", "vit_large_patch16_224"]


def _cfg(url='', **kwargs):
    """modeling_finetune.py:10-17"""
    return {'url': url, 'num_classes': 400, 'input_size': (3, 224, 224), 'pool_size': None, 'crop_pct': .9,
            'interpolation': 'bicubic', 'mean': (0.5, 0.5, 0.5), 'std': (0.5, 0.5, 0.5), **kwargs}


class VisionTransformer(_FlatModule):
    """modeling_finetune.py:305-409"""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.,
                 norm_layer=nn.LayerNorm, init_values=0., use_learnable_pos_emb=False, init_scale=0., all_frames=16,
                 tubelet_size=2, use_mean_pooling=True):
        super().__init__()
        # dropout / drop-path are identities in eval mode, which is the only mode this forward serves
        _unsupported(qk_scale=qk_scale, init_values=init_values, use_learnable_pos_emb=use_learnable_pos_emb,
                     no_qkv_bias=not qkv_bias, no_mean_pooling=not use_mean_pooling, no_head=num_classes <= 0)
        _check_heads(embed_dim, num_heads)
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.tubelet_size = tubelet_size
        self.depth, self.num_heads, self.mlp_ratio, self.in_chans = depth, num_heads, mlp_ratio, in_chans
        self.drop_rate, self.attn_drop_rate, self.drop_path_rate = drop_rate, attn_drop_rate, drop_path_rate
        self.eps = _ln_eps(norm_layer, embed_dim)
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      num_frames=all_frames, tubelet_size=tubelet_size)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, self.eps) for _ in range(depth)])
        self.norm = nn.Identity()
        self.fc_norm = nn.LayerNorm(embed_dim, eps=self.eps)
        self.head = nn.Linear(embed_dim, num_classes)
        nn.init.trunc_normal_(self.head.weight, std=.02)
        self.apply(self._init_weights)
        self.head.weight.data.mul_(init_scale)
        self.head.bias.data.mul_(init_scale)

    @staticmethod
    def _init_weights(m):
        """modeling_finetune.py:365-373"""
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_num_layers(self):
        return len(self.blocks)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def get_classifier(self):
        return self.head

    # ------------------------------------------------------------------ checkpoint interchange
    def load_pretrained_encoder(self, checkpoint_model, strict=False):
        """run_class_finetuning.py:362-411: take a pretraining state_dict (``encoder.*`` / ``backbone.*`` prefixes stripped,
        decoder / mask_token / encoder.norm keys dropped, mismatching head dropped) and load it non-strictly."""
        own = self.state_dict()
        new = OrderedDict()
        for k, v in checkpoint_model.items():
            if k.startswith('backbone.'):
                k = k[9:]
            elif k.startswith('encoder.'):
                k = k[8:]
            if k in ('head.weight', 'head.bias') and k in own and v.shape != own[k].shape:
                continue
            if k in own:
                new[k] = v
        return self.load_state_dict(new, strict=strict)

    # ------------------------------------------------------------------ runtime plumbing
    def _dims(self):
        pe = self.patch_embed
        return Dims(img_size=pe.img_size[0], patch_size=pe.patch_size[0], tubelet=pe.tubelet_size, num_frames=pe.num_frames,
                    in_chans=self.in_chans, enc_dim=self.embed_dim, enc_depth=self.depth, enc_heads=self.num_heads,
                    mlp_ratio=self.mlp_ratio, eps=self.eps, dec_depth=0)

    def _flat_order(self):
        return (["patch_embed.proj.weight", "patch_embed.proj.bias"] + _block_names("", self.depth)
                + ["fc_norm.weight", "fc_norm.bias", "head.weight", "head.bias"])

    def _make_runtime(self, store):
        return PretrainRuntime(self._dims(), store, enc_prefix="", dec_prefix=None, top=False, forward_only=True)

    def _ws(self, rt, B):
        """inference workspace: ONE block's activations, reused by every layer"""
        cache = rt.__dict__.setdefault("_ft_ws", {})
        if B in cache:
            return cache[B]
        d, dev = rt.d, rt.dev
        N, D = d.num_patches, d.enc_dim
        M = B * N
        e = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
        w = NS(B=B, N=N, M=M,
               clips=e(B, d.in_chans, d.num_frames, d.img_size, d.img_size, dt=F32),
               idx=torch.arange(N, dtype=I32, device=dev).repeat(B, 1).contiguous(),
               xp=e(M, d.patch_dim), x0=e(M, D, dt=F32), L=rt._block_ws(M, D, d.enc_heads, B, N),
               pooled=e(B, D, dt=F32), feat=e(B, D, dt=F32), feat_bf16=e(B, D))
        # the GEMM writes 4 classes per lane: the head is padded to a multiple of 8 classes (zero rows), e.g. 174 -> 176
        ncp = -(-self.num_classes // 8) * 8
        w.head_w = torch.zeros(ncp, D, dtype=BF16, device=dev)
        w.head_b = torch.zeros(ncp, dtype=F32, device=dev)
        w.logits = e(B, ncp, dt=F32)
        cache[B] = w
        return w

    def _launch(self, rt, w):
        d, s = rt.d, rt.store
        ops.patch_gather(w.clips, d.tubelet, d.patch_size, w.idx, w.xp)
        ops.gemm(ops.GEMM_NT, ops.EPI_POS_F32, w.xp, s.bview("patch_embed.proj.weight"), w.x0, bias=s.view("patch_embed.proj.bias"),
                 pos=rt.pos_enc, row_idx=w.idx.view(-1), rows_in=w.M, rows_out=w.M)
        x = w.x0
        for W in rt.encW:
            x = rt._block_fwd(W, w.L, x, w.B, w.N, d.enc_heads)
        ops.token_mean_norm(x, w.B, w.N, s.view("fc_norm.weight"), s.view("fc_norm.bias"), d.eps, w.pooled, w.feat, w.feat_bf16)
        ops.gemm(ops.GEMM_NT, ops.EPI_F32, w.feat_bf16, w.head_w, w.logits, bias=w.head_b)

    def _run(self, x):
        if x.dim() != 5:
            raise ValueError("expected clips [B, C, T, H, W]")
        rt = self.runtime()
        w = self._ws(rt, x.shape[0])
        if tuple(x.shape) != tuple(w.clips.shape):
            raise ValueError(f"clip shape {tuple(x.shape)} does not match the model's {tuple(w.clips.shape)} (modeling_finetune.py:245)")
        w.clips.copy_(x, non_blocking=True)
        rt.store.refresh_shadow()
        w.head_w[:self.num_classes].copy_(rt.store.bview("head.weight"))
        w.head_b[:self.num_classes].copy_(rt.store.view("head.bias"))
        rt.cached(w, "ft_fwd", lambda: self._launch(rt, w))
        return w

    @torch.no_grad()
    def forward_features(self, x):
        """modeling_finetune.py:389-404: [B, embed_dim] = fc_norm(mean over the 1568 tokens)"""
        return self._run(x).feat.clone()

    @torch.no_grad()
    def forward(self, x):
        """modeling_finetune.py:406-409: logits [B, num_classes]"""
        return self._run(x).logits[:, :self.num_classes].clone()


class VisionTransformer_feat_ext(VisionTransformer):
    """modeling_finetune.py:411-420"""

    @torch.no_grad()
    def forward(self, x):
        return self.forward_features(x)


def vit_small_patch16_224(pretrained=False, **kwargs):
    model = VisionTransformer(patch_size=16, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True,
                              norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = _cfg()
    return model


def vit_base_patch16_224(pretrained=False, **kwargs):
    model = VisionTransformer(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                              norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = _cfg()
    return model


def vit_base_patch16_224_feature_ext(pretrained=False, **kwargs):
    model = VisionTransformer_feat_ext(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                                       norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = _cfg()
    return model


def vit_large_patch16_224(pretrained=False, **kwargs):
    model = VisionTransformer(patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, qkv_bias=True,
                              norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = _cfg()
    return model

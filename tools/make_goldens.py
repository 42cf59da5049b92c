#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference (Moohnai/MOFO) in the build container.

Runs only where /root/reference exists (the build container).  The reference's Python never travels:
only the small .npz fixtures written here are committed.  Inputs and weights are regenerated on both
sides from name-keyed deterministic draws (oracle.pretrain_oracle.keyed_*), so fixtures hold outputs only.

Missing third-party modules are replaced by in-memory stand-ins that carry NO arithmetic of the
pretraining path: timm (registry decorator, to_2tuple, trunc_normal_ used only for the
initial value of mask_token which load_state_dict overwrites, drop_path which is dead at rate 0, the
ImageNet constants), tensorboardX / wandb / cv2 (never called on the path we drive).

What is driven through the reference's own code:
  * masking_generator.TubeMaskingGenerator / TubeMaskingGenerator_BB          -> masks.npz
  * modeling_finetune.get_sinusoid_encoding_table                             -> sincos.npz
  * modeling_pretrain.PretrainVisionTransformer (+ hooks on its blocks)       -> tiny_*.npz, vitb_*.npz
  * optim_factory.create_optimizer + utils.NativeScalerWithGradNormCount     -> post-step losses
  * utils.cosine_scheduler                                                    -> sched.npz
  * engine_for_pretraining.train_one_epoch itself, one step, B=2 (BASELINE config[0]) -> engine_vitb.npz
    (its matplotlib PNG dump is pointed at no-op objects; torch.cuda.synchronize, which the reference
    calls unconditionally, is a no-op on this CPU-only box; the loss scaler is the reference's class with
    state_dict() answering {'scale': 1.0} because a disabled GradScaler answers {} -> KeyError at :177).
"""
import argparse
import os
import sys
import tempfile
import types
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def install_standins():
    def mk(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    mk("timm")
    tm = mk("timm.models")
    tl = mk("timm.models.layers")
    tr = mk("timm.models.registry")
    tu = mk("timm.utils")
    td = mk("timm.data")
    tdc = mk("timm.data.constants")
    registry = {}

    def register_model(fn):
        registry[fn.__name__] = fn
        return fn

    def create_model(name, pretrained=False, **kw):
        return registry[name](pretrained=pretrained, **{k: v for k, v in kw.items() if v is not None})

    def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):
        return torch.nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    def drop_path(x, drop_prob=0., training=False):
        assert drop_prob == 0. or not training, "drop_path>0 is not on the pretraining path"
        return x

    tl.trunc_normal_ = trunc_normal_
    tl.drop_path = drop_path
    tl.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    tr.register_model = register_model
    tm.create_model = create_model
    tu.get_state_dict = lambda m, *a, **k: m.state_dict()
    tdc.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
    tdc.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
    td.constants = tdc
    mk("timm.optim")
    for sub, cls in [("adafactor", "Adafactor"), ("adahessian", "Adahessian"), ("adamp", "AdamP"),
                     ("lookahead", "Lookahead"), ("nadam", "Nadam"), ("novograd", "NovoGrad"),
                     ("nvnovograd", "NvNovoGrad"), ("radam", "RAdam"), ("rmsprop_tf", "RMSpropTF"),
                     ("sgdp", "SGDP")]:
        setattr(mk("timm.optim." + sub), cls, type(cls, (), {}))
    mk("tensorboardX").SummaryWriter = type("SummaryWriter", (), {})
    mk("wandb")
    if "cv2" not in sys.modules:
        try:
            import cv2  # noqa: F401
        except Exception:
            mk("cv2")


class _Noop:
    def __getattr__(self, _):
        return lambda *a, **k: _Noop()

    def __call__(self, *a, **k):
        return _Noop()


def tap_once(taps, key, value):
    """record the FIRST forward only; returns None so a pre-hook does not replace the module input"""
    if key not in taps:
        taps[key] = value


def head(t, n=16):
    return t.detach().reshape(-1)[:n].double().numpy().copy()


def tensor_stats(d):
    """per-tensor [l2, sum, abs-max] + first 16 values, keyed by name order."""
    names = list(d.keys())
    st = np.array([[float(torch.norm(d[k].double())), float(d[k].double().sum()), float(d[k].abs().max())]
                   for k in names])
    hd = np.stack([np.pad(head(d[k]), (0, 16 - min(16, d[k].numel()))) for k in names])
    return names, st, hd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--skip-vitb", action="store_true")
    ap.add_argument("--only-next", action="store_true", help="only the fixtures of the SURVEY 8f 'next' rows (vis.npz, finetune_*.npz)")
    ap.add_argument("--only-bb-engine", action="store_true", help="only engine_vitb_bb.npz: one step of the reference's own train_one_epoch_BB")
    ap.add_argument("--only-l32", action="store_true", help="only vitl32.npz: ViT-L widths at 32 frames (BASELINE config 4 shapes) through the reference classes")
    ap.add_argument("--full", action="store_true", help="with --only-l32: the FULL ViT-L depth (24 + 4 blocks) -> vitl32_full.npz (SURVEY 8c fixture F6)")
    ap.add_argument("--batch", type=int, default=1, help="with --only-l32 --full: clips in the batch (4 -> vitl32_full_b4.npz: the routes a batch takes, pinned beyond clip 0)")
    ap.add_argument("--only-clip", action="store_true", help="only tiny_clip.npz: steps through the reference scaler with clip_grad")
    ap.add_argument("--only-ckpt", action="store_true", help="only ckpt_tiny.npz: a checkpoint WRITTEN by the reference's utils.save_model after two steps")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(os.cpu_count())

    install_standins()
    sys.path.insert(0, REF)
    import masking_generator as ref_mg
    import modeling_finetune as ref_mf
    import modeling_pretrain as ref_mp
    import optim_factory as ref_of
    import utils as ref_utils
    import engine_for_pretraining as ref_eng
    from oracle import pretrain_oracle as O

    if args.only_next:
        make_next(args, ref_mg, ref_mf, O)
        return
    if args.only_clip:
        make_clip(args, ref_mp, ref_of, ref_utils, O)
        return
    if args.only_ckpt:
        make_ckpt(args, ref_mp, ref_of, ref_utils, O)
        return
    if args.only_l32:
        make_l32(args, ref_mp, ref_mf, O)
        return
    if args.only_bb_engine:
        make_bb_engine(args, ref_mp, ref_of, ref_utils, ref_eng, O)
        return

    # ------------------------------------------------------------------ F1 masks
    out = {}
    for seed in (10, 0, 1, 2, 3):
        np.random.seed(seed)
        out[f"tube_s{seed}"] = ref_mg.TubeMaskingGenerator((8, 14, 14), 0.9)().astype(np.uint8)
    np.random.seed(10)
    out["tube_tiny_s10"] = np.stack([ref_mg.TubeMaskingGenerator((8, 2, 2), 0.75)() for _ in range(2)]).astype(np.uint8)
    np.random.seed(7)
    out["tube_l32_s7"] = ref_mg.TubeMaskingGenerator((16, 14, 14), 0.9)().astype(np.uint8)
    boxes = np.array([[60, 40, 160, 180], [0, 0, 1, 1], [0, 0, 224, 224], [100, 100, 120, 120],
                      [10, 150, 90, 223], [200, 3, 223, 40]], dtype=np.int64)
    out["bb_boxes"] = boxes
    for seed in (10, 0):
        ms = []
        for b in boxes:
            np.random.seed(seed)
            ms.append(ref_mg.TubeMaskingGenerator_BB((8, 14, 14), 0.9, 0.75)(np.tile(b, (16, 1))))
        out[f"bb_s{seed}"] = np.stack(ms).astype(np.uint8)
    np.random.seed(5)   # sequential draws from one stream (as a DataLoader worker would)
    out["bb_stream_s5"] = np.stack([ref_mg.TubeMaskingGenerator_BB((8, 14, 14), 0.9, 0.75)(np.tile(b, (16, 1)))
                                    for b in boxes]).astype(np.uint8)
    np.savez_compressed(os.path.join(args.out, "masks.npz"), **out)
    print("masks.npz", {k: v.shape for k, v in out.items()})

    # ------------------------------------------------------------------ F2 sincos + schedules
    out = {}
    for n, d in ((1568, 768), (1568, 384), (32, 128), (32, 64), (3136, 1024)):
        t = ref_mf.get_sinusoid_encoding_table(n, d)
        out[f"t{n}x{d}_head"] = t[0, :4, :8].numpy()
        out[f"t{n}x{d}_tail"] = t[0, -4:, -8:].numpy()
        out[f"t{n}x{d}_sum"] = np.array(t.double().sum().item())
        out[f"t{n}x{d}_row777"] = t[0, min(777, n - 1), ::max(1, d // 16)].numpy()
    np.savez_compressed(os.path.join(args.out, "sincos.npz"), **out)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        s1 = ref_utils.cosine_scheduler(1.5e-4, 1e-5, 10, 7, warmup_epochs=3)
        s2 = ref_utils.cosine_scheduler(0.05, 0.05, 4, 5)
        s3 = ref_utils.cosine_scheduler(1.2e-3, 1e-5, 6, 11, warmup_epochs=2, warmup_steps=9)
    np.savez_compressed(os.path.join(args.out, "sched.npz"), s1=s1, s2=s2, s3=s3)

    # ------------------------------------------------------------------ helpers around the reference model
    def build_ref(cfg, params):
        m = ref_mp.PretrainVisionTransformer(
            img_size=cfg.img_size, patch_size=cfg.patch_size, encoder_embed_dim=cfg.enc_dim,
            encoder_depth=cfg.enc_depth, encoder_num_heads=cfg.enc_heads, encoder_num_classes=0,
            decoder_num_classes=cfg.patch_dim, decoder_embed_dim=cfg.dec_dim, decoder_depth=cfg.dec_depth,
            decoder_num_heads=cfg.dec_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=True,
            norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
        if cfg.num_frames != 16:
            # documented deviation: the reference hard-wires 16 frames in the pretrain
            # PatchEmbed, so its tables are rebuilt with the reference's own table function.
            m.encoder.pos_embed = ref_mf.get_sinusoid_encoding_table(cfg.num_patches, cfg.enc_dim)
            m.pos_embed = ref_mf.get_sinusoid_encoding_table(cfg.num_patches, cfg.dec_dim)
        missing = m.load_state_dict(params, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        assert [k for k, _ in m.named_parameters()] == list(params.keys()), "param order differs from oracle schema"
        return m

    def ref_labels_via_engine_arith(videos, mask, cfg):
        # the label builder is inline in train_one_epoch (engine_for_pretraining.py:43-63); for configs the
        # engine cannot run (its PNG block hard-codes 1568 tokens) labels come from the capture in the
        # engine run below for ViT-B and are otherwise not a golden (the oracle's builder is pinned there).
        raise NotImplementedError

    class OptArgs:
        opt = "adamw"
        lr = 1.5e-4
        weight_decay = 0.05
        opt_eps = 1e-8
        opt_betas = (0.9, 0.95)
        momentum = 0.9

    def ref_train_steps(model, videos, mask, labels, nsteps):
        """reference model + reference create_optimizer + reference scaler, engine arithmetic lines
        :65-69,:172-176 re-stated (loss = MSELoss(model(videos,mask), labels))."""
        with contextlib.redirect_stdout(io.StringIO()):
            opt = ref_of.create_optimizer(OptArgs, model)
        scaler = ref_utils.NativeScalerWithGradNormCount()
        losses, norms, grads0 = [], [], None
        for s in range(nsteps):
            outp = model(videos, mask)
            loss = torch.nn.MSELoss()(outp, labels)
            opt.zero_grad()
            gn = scaler(loss, opt, clip_grad=None, parameters=model.parameters())
            if s == 0:
                grads0 = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
                out0 = outp.detach().clone()
            losses.append(loss.item())
            norms.append(float(gn))
        return losses, norms, grads0, out0, [len(g["params"]) for g in opt.param_groups]

    # ------------------------------------------------------------------ F3 tiny config, full tensors
    cfg = O.TINY
    for mode in ("small", "xavier"):
        P = O.keyed_params(cfg, mode)
        model = build_ref(cfg, P)
        videos = O.keyed_clips(2, cfg)
        mask = torch.from_numpy(out_tiny_mask := np.load(os.path.join(args.out, "masks.npz"))["tube_tiny_s10"]).bool()
        taps = {}
        hooks = []
        hooks.append(model.encoder.patch_embed.register_forward_hook(lambda m, i, o: tap_once(taps, "patch_embed", o.detach().clone())))
        for i, blk in enumerate(model.encoder.blocks):
            hooks.append(blk.register_forward_hook(lambda m, i_, o, i=i: tap_once(taps, f"enc_block{i}", o.detach().clone())))
            if i == 0:
                hooks.append(blk.register_forward_pre_hook(lambda m, i_: tap_once(taps, "x_vis0", i_[0].detach().clone())))
        hooks.append(model.encoder.register_forward_hook(lambda m, i, o: tap_once(taps, "enc_out", o.detach().clone())))
        hooks.append(model.decoder.register_forward_pre_hook(lambda m, i_: tap_once(taps, "x_full", i_[0].detach().clone())))
        for i, blk in enumerate(model.decoder.blocks):
            hooks.append(blk.register_forward_hook(lambda m, i_, o, i=i: tap_once(taps, f"dec_block{i}", o.detach().clone())))
        # labels for the tiny config: oracle builder (pinned on ViT-B by the engine capture below)
        labels = O.build_targets(videos, mask, cfg)
        losses, norms, grads0, out0, group_sizes = ref_train_steps(model, videos, mask, labels, 3)
        for h in hooks:
            h.remove()
        names, gstat, ghead = tensor_stats(grads0)
        pnames, pstat, phead = tensor_stats(dict(model.named_parameters()))
        fx = dict(output=out0.numpy(), out_sum=np.array(out0.double().sum().item()), labels=labels.numpy(),
                  losses=np.array(losses), grad_norms=np.array(norms), names=np.array(names),
                  grad_stats=gstat, grad_head=ghead, param_stats_after3=pstat, param_head_after3=phead,
                  group_sizes=np.array(group_sizes))
        for k, v in taps.items():
            fx["tap_" + k] = v.numpy()
        for k, g in grads0.items():
            if g.numel() <= 4096:
                fx["grad_" + k] = g.numpy()
        np.savez_compressed(os.path.join(args.out, f"tiny_{mode}.npz"), **fx)
        print(f"tiny_{mode}: out_sum={out0.double().sum().item():.15f} out[0,0,:3]={out0[0,0,:3].tolist()} losses={losses} gn={norms}")

    # ------------------------------------------------------------------ F7 ingest: Stack -> ToTorchFormatTensor -> GroupNormalize
    # (transforms.py imports torchvision / albumentations at module level but these three classes use only numpy/torch/PIL)
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "albumentations"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    import transforms as ref_tf
    from PIL import Image
    rs = np.random.RandomState(77)
    T, H, W = 16, 24, 40
    frames = rs.randint(0, 256, size=(T, H, W, 3)).astype(np.uint8)
    imgs = [Image.fromarray(frames[t], mode="RGB") for t in range(T)]
    stacked, _ = ref_tf.Stack(roll=False)((imgs, None))                       # [H, W, T*3] uint8
    ten, _ = ref_tf.ToTorchFormatTensor(div=True)((stacked, None))
    ten, _ = ref_tf.GroupNormalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])((ten, None))
    clip = ten.view((T, 3) + ten.size()[-2:]).transpose(0, 1).contiguous()    # kinetics.py:492-493 -> [3, T, H, W]
    np.savez_compressed(os.path.join(args.out, "ingest.npz"), stacked=np.ascontiguousarray(stacked), clip=clip.numpy())
    print("ingest.npz", stacked.shape, clip.shape)

    if args.skip_vitb:
        return

    # ------------------------------------------------------------------ engine run: BASELINE config[0]
    cfg = O.VIT_B
    masks = np.load(os.path.join(args.out, "masks.npz"))
    P = O.keyed_params(cfg, "xavier")
    videos = O.keyed_clips(2, cfg)
    tube = torch.from_numpy(np.stack([masks["tube_s10"], masks["tube_s0"]]).astype(np.float64))   # collate gives f64 [B,1568]
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref_eng_model = ref_mp.pretrain_videomae_base_patch16_224(decoder_depth=4)
    model.load_state_dict(P, strict=True)
    with contextlib.redirect_stdout(io.StringIO()):
        opt = ref_of.create_optimizer(OptArgs, model)
        lr_sched = ref_utils.cosine_scheduler(1.5e-4, 1e-5, 2, 1, warmup_epochs=0)
        wd_sched = ref_utils.cosine_scheduler(0.05, 0.05, 2, 1)
    captured = {}

    class CapMSE(torch.nn.MSELoss):
        def forward(self, input, target):
            captured["outputs"] = input.detach().clone()
            captured["labels"] = target.detach().clone()
            return super().forward(input, target)

    class Scaler(ref_utils.NativeScalerWithGradNormCount):
        def state_dict(self):
            return {"scale": 1.0}

    ref_eng.nn.MSELoss = CapMSE
    ref_eng.plt = _Noop()
    ref_eng.cv2 = _Noop()
    real_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                stats = ref_eng.train_one_epoch(model, [(videos, tube)], opt, torch.device("cpu"), 0, Scaler(),
                                                max_norm=None, patch_size=16, normlize_target=True,
                                                start_steps=0, lr_schedule_values=lr_sched,
                                                wd_schedule_values=wd_sched)
        finally:
            os.chdir(cwd)
            torch.cuda.synchronize = real_sync
            ref_eng.nn.MSELoss = torch.nn.MSELoss
    grads0 = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    names, gstat, ghead = tensor_stats(grads0)
    pnames, pstat, phead = tensor_stats(dict(model.named_parameters()))
    lab, outp = captured["labels"], captured["outputs"]
    fx = dict(loss=np.array(stats["loss"]), grad_norm=np.array(stats["grad_norm"]), lr=np.array(stats["lr"]),
              weight_decay=np.array(stats["weight_decay"]), names=np.array(names), grad_stats=gstat, grad_head=ghead,
              param_stats_after1=pstat, param_head_after1=phead,
              labels_slice=lab[:, :6, :48].numpy(), labels_sum=np.array(lab.double().sum().item()),
              labels_sqsum=np.array((lab.double() ** 2).sum().item()), labels_tail=lab[:, -3:, -24:].numpy(),
              out_slice=outp[:, :6, :48].numpy(), out_sum=np.array(outp.double().sum().item()),
              out_tail=outp[:, -3:, -24:].numpy())
    # two more steps with the restated driver (same reference model/optimizer objects continue)
    scaler = Scaler()
    more = []
    for s in range(2):
        o2 = model(videos, tube.bool())
        l2 = torch.nn.MSELoss()(o2, lab)
        opt.zero_grad()
        scaler(l2, opt, clip_grad=None, parameters=model.parameters())
        more.append(l2.item())
    fx["losses_after"] = np.array(more)
    np.savez_compressed(os.path.join(args.out, "engine_vitb.npz"), **fx)
    print("engine_vitb: loss", stats["loss"], "gn", stats["grad_norm"], "then", more)

    # ------------------------------------------------------------------ F5: BB masks, ViT-B, B=2 (config 3 shape)
    bbm = torch.from_numpy(masks["bb_s10"][[0, 3]]).bool()
    P = O.keyed_params(cfg, "xavier")
    model.load_state_dict(P, strict=True)
    labels = O.build_targets(videos, bbm, cfg)
    losses, norms, grads0, out0, _ = ref_train_steps(model, videos, bbm, labels, 1)
    names, gstat, ghead = tensor_stats(grads0)
    np.savez_compressed(os.path.join(args.out, "vitb_bb.npz"), loss=np.array(losses[0]), grad_norm=np.array(norms[0]),
                        names=np.array(names), grad_stats=gstat, grad_head=ghead,
                        out_slice=out0[:, :6, :48].numpy(), out_sum=np.array(out0.double().sum().item()))
    print("vitb_bb: loss", losses, "gn", norms)


def make_bb_engine(args, ref_mp, ref_of, ref_utils, ref_eng, O):
    """engine_vitb_bb.npz: ONE step of the reference's own motion-box epoch loop (engine_for_pretraining.py:215-468,
    train_one_epoch_BB) on ViT-B, B=2, batch = (videos, boxes int [16,4], BB masks) -- loss, grad norm, lr, weight decay and
    the per-tensor gradient norms.  Stand-ins: a scaler whose state_dict has 'scale' (the disabled GradScaler's is empty)
    and a no-op torch.cuda.synchronize; this loop's PNG dump is commented out in the reference."""
    import contextlib
    import io

    class OptArgs:
        opt, lr, weight_decay, opt_eps, opt_betas, momentum = "adamw", 1.5e-4, 0.05, 1e-8, (0.9, 0.95), 0.9

    class Scaler(ref_utils.NativeScalerWithGradNormCount):
        def state_dict(self):
            return {"scale": 1.0}

    cfg = O.VIT_B
    masks = np.load(os.path.join(args.out, "masks.npz"))
    P = O.keyed_params(cfg, "xavier")
    videos = O.keyed_clips(2, cfg)
    pick = [0, 3]
    bbm = torch.from_numpy(masks["bb_s10"][pick].astype(np.float64))                         # collate: f64 [B,1568]
    boxes = torch.from_numpy(np.stack([np.tile(masks["bb_boxes"][i], (16, 1)) for i in pick]))   # int64 [B,16,4]
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref_mp.pretrain_videomae_base_patch16_224(decoder_depth=4)
    model.load_state_dict(P, strict=True)
    with contextlib.redirect_stdout(io.StringIO()):
        opt = ref_of.create_optimizer(OptArgs, model)
        lr_sched = ref_utils.cosine_scheduler(1.5e-4, 1e-5, 2, 1, warmup_epochs=0)
        wd_sched = ref_utils.cosine_scheduler(0.05, 0.05, 2, 1)
    real_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            stats = ref_eng.train_one_epoch_BB(model, [(videos, boxes, bbm)], opt, torch.device("cpu"), 0, Scaler(), max_norm=None,
                                               patch_size=16, normlize_target=True, start_steps=0, lr_schedule_values=lr_sched,
                                               wd_schedule_values=wd_sched)
    finally:
        torch.cuda.synchronize = real_sync
    grads = {k: p.grad.detach() for k, p in model.named_parameters()}
    names, gstat, ghead = tensor_stats(grads)
    np.savez_compressed(os.path.join(args.out, "engine_vitb_bb.npz"), pick=np.array(pick), loss=np.array(stats["loss"]),
                        grad_norm=np.array(float(stats["grad_norm"])), lr=np.array(stats["lr"]), weight_decay=np.array(stats["weight_decay"]),
                        names=np.array(names), grad_stats=gstat)
    print("engine_vitb_bb:", {k: float(v) for k, v in stats.items()})


def make_l32(args, ref_mp, ref_mf, O):
    """vitl32.npz: BASELINE config 4's shapes -- ViT-L widths (1024 x 16 heads / 512 x 8 heads), 32 x 224 x 224 clips ->
    3136 tokens, 320 visible -- through the REFERENCE classes, 3 encoder + 1 decoder blocks (the full depth adds nothing
    new per block and this keeps the fixture a few-second job).  The reference hard-wires 16 frames in its position tables
    (modeling_pretrain.py:52-55,199-200: built for patch_embed.num_patches of a 16-frame clip); the two tables are rebuilt
    with the reference's own get_sinusoid_encoding_table for 3136 positions -- the documented deviation of SURVEY.md 5 / 8c F6."""
    import contextlib
    import io
    from functools import partial as _partial
    full = getattr(args, "full", False)       # --full: the whole pretrain_videomae_large_patch16_224 depth (modeling_pretrain.py:316-338), dec 4
    cfg = O.OracleConfig(num_frames=32, enc_dim=1024, enc_depth=24 if full else 3, enc_heads=16, dec_dim=512, dec_depth=4 if full else 1, dec_heads=8)
    P = O.keyed_params(cfg, "xavier")
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref_mp.PretrainVisionTransformer(
            img_size=cfg.img_size, patch_size=cfg.patch_size, encoder_embed_dim=cfg.enc_dim, encoder_depth=cfg.enc_depth,
            encoder_num_heads=cfg.enc_heads, encoder_num_classes=0, decoder_num_classes=cfg.patch_dim, decoder_embed_dim=cfg.dec_dim,
            decoder_depth=cfg.dec_depth, decoder_num_heads=cfg.dec_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=True,
            norm_layer=_partial(torch.nn.LayerNorm, eps=1e-6))
    model.load_state_dict(P, strict=True)
    model.encoder.pos_embed = ref_mf.get_sinusoid_encoding_table(cfg.num_patches, cfg.enc_dim)
    model.pos_embed = ref_mf.get_sinusoid_encoding_table(cfg.num_patches, cfg.dec_dim)
    nb = max(1, int(getattr(args, "batch", 1))) if full else 1
    videos = O.keyed_clips(nb, cfg)
    np.random.seed(7)
    import masking_generator as ref_mg
    mgen = ref_mg.TubeMaskingGenerator(cfg.grid, 0.9)
    mask = torch.from_numpy(np.stack([mgen() for _ in range(nb)])).bool()      # clip 0's mask is the one-clip fixture's
    assert mask.shape[1] == 3136 and all(int((~m).sum()) == 320 for m in mask)
    labels = O.build_targets(videos, mask, cfg)          # the target builder is pinned by engine_vitb.npz (reference engine capture)
    out = model(videos, mask)
    loss = torch.nn.MSELoss()(out, labels)
    loss.backward()
    grads = {k: p.grad.detach() for k, p in model.named_parameters()}
    names, gstat, ghead = tensor_stats(grads)
    gn = float(torch.sqrt(sum(g.double().pow(2).sum() for g in grads.values())))
    fname = ("vitl32_full.npz" if nb == 1 else f"vitl32_full_b{nb}.npz") if full else "vitl32.npz"
    np.savez_compressed(os.path.join(args.out, fname), mask=mask.numpy().astype(np.uint8), loss=np.array(loss.item()), grad_norm=np.array(gn),
                        names=np.array(names), grad_stats=gstat, grad_head=ghead, out_slice=out[:, :6, :48].detach().numpy(),
                        out_sum=np.array(out.detach().double().sum().item()))
    print("vitl32: loss", loss.item(), "grad norm", gn, "out_sum", out.detach().double().sum().item())


def make_clip(args, ref_mp, ref_of, ref_utils, O):
    """tiny_clip.npz: three steps of the tiny config through the reference's NativeScalerWithGradNormCount with
    clip_grad=0.1 (utils.py:353-367: unscale -> torch.nn.utils.clip_grad_norm_ -> optimizer.step; the value returned is the
    norm BEFORE clipping) and the reference's create_optimizer: losses, returned norms, parameter statistics afterwards."""
    import contextlib
    import io
    from functools import partial as _partial

    class OptArgs:
        opt, lr, weight_decay, opt_eps, opt_betas, momentum = "adamw", 1.5e-4, 0.05, 1e-8, (0.9, 0.95), 0.9

    cfg = O.TINY
    P = O.keyed_params(cfg, "xavier")
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref_mp.PretrainVisionTransformer(
            img_size=cfg.img_size, patch_size=cfg.patch_size, encoder_embed_dim=cfg.enc_dim, encoder_depth=cfg.enc_depth,
            encoder_num_heads=cfg.enc_heads, encoder_num_classes=0, decoder_num_classes=cfg.patch_dim, decoder_embed_dim=cfg.dec_dim,
            decoder_depth=cfg.dec_depth, decoder_num_heads=cfg.dec_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=True,
            norm_layer=_partial(torch.nn.LayerNorm, eps=1e-6))
    model.load_state_dict(P, strict=True)
    videos = O.keyed_clips(2, cfg)
    mask = torch.from_numpy(np.load(os.path.join(args.out, "masks.npz"))["tube_tiny_s10"]).bool()
    labels = O.build_targets(videos, mask, cfg)          # pinned on ViT-B by engine_vitb.npz
    with contextlib.redirect_stdout(io.StringIO()):
        opt = ref_of.create_optimizer(OptArgs, model)
    scaler = ref_utils.NativeScalerWithGradNormCount()
    losses, norms = [], []
    for _ in range(3):
        loss = torch.nn.MSELoss()(model(videos, mask), labels)
        opt.zero_grad()
        norm = scaler(loss, opt, clip_grad=0.1, parameters=model.parameters())
        losses.append(loss.item())
        norms.append(float(norm))
    names, pstat, phead = tensor_stats(dict(model.named_parameters()))
    np.savez_compressed(os.path.join(args.out, "tiny_clip.npz"), clip_grad=np.array(0.1), losses=np.array(losses), norms=np.array(norms),
                        names=np.array(names), param_stats_after3=pstat, param_head_after3=phead)
    print("tiny_clip: losses", losses, "norms (before clipping)", norms)


def make_ckpt(args, ref_mp, ref_of, ref_utils, O):
    """ckpt_tiny.npz: the tiny config trained for two steps with the reference's optimizer + scaler, then WRITTEN by the
    reference's own utils.save_model (utils.py:411-433) and read back with torch.load; the file's tensors are stored as plain
    arrays (model/<key>, opt/<index>/{exp_avg,exp_avg_sq,step}, the param_groups as JSON) next to the loss and gradient norm
    of the reference's THIRD step, which a resumed run must reproduce."""
    import contextlib
    import io
    import json
    import tempfile
    from functools import partial as _partial

    class OptArgs:
        opt, lr, weight_decay, opt_eps, opt_betas, momentum = "adamw", 1.5e-3, 0.05, 1e-8, (0.9, 0.95), 0.9

    cfg = O.TINY
    P = O.keyed_params(cfg, "xavier")
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref_mp.PretrainVisionTransformer(
            img_size=cfg.img_size, patch_size=cfg.patch_size, encoder_embed_dim=cfg.enc_dim, encoder_depth=cfg.enc_depth,
            encoder_num_heads=cfg.enc_heads, encoder_num_classes=0, decoder_num_classes=cfg.patch_dim, decoder_embed_dim=cfg.dec_dim,
            decoder_depth=cfg.dec_depth, decoder_num_heads=cfg.dec_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=True,
            norm_layer=_partial(torch.nn.LayerNorm, eps=1e-6))
    model.load_state_dict(P, strict=True)
    videos = O.keyed_clips(2, cfg)
    mask = torch.from_numpy(np.load(os.path.join(args.out, "masks.npz"))["tube_tiny_s10"]).bool()
    labels = O.build_targets(videos, mask, cfg)
    with contextlib.redirect_stdout(io.StringIO()):
        opt = ref_of.create_optimizer(OptArgs, model)
    scaler = ref_utils.NativeScalerWithGradNormCount()

    def step():
        loss = torch.nn.MSELoss()(model(videos, mask), labels)
        opt.zero_grad()
        norm = scaler(loss, opt, clip_grad=None, parameters=model.parameters())
        return loss.item(), float(norm)

    first = [step() for _ in range(2)]
    with tempfile.TemporaryDirectory() as td:
        a = argparse.Namespace(output_dir=td)
        ref_utils.save_model(a, 1, model, model, opt, scaler)                       # the reference's own writer
        ck = torch.load(os.path.join(td, "checkpoint-1.pth"), map_location="cpu", weights_only=False)
    third = step()
    # The file itself would be 8 MB (parameters + two moments); the fixture keeps its STRUCTURE exactly (key lists in file
    # order, shapes, dtypes, the optimizer's index -> tensor mapping, param_groups) and every tensor's L2 norm + first values.
    def stat(t):
        t = t.detach().double().reshape(-1)
        return np.array([float(t.norm()), float(t.sum())]), t[:8].numpy()

    out = {"top_keys": np.array(sorted(ck.keys())), "epoch": np.array(ck["epoch"]), "scaler_keys": np.array(sorted(ck["scaler"].keys()), dtype="U16"),
           "model_keys": np.array(list(ck["model"].keys())), "losses12": np.array([f[0] for f in first]), "norms12": np.array([f[1] for f in first]),
           "loss3": np.array(third[0]), "norm3": np.array(third[1]), "lr": np.array(OptArgs.lr),
           "model_dtypes": np.array([str(v.dtype) for v in ck["model"].values()])}
    mstat, mhead, mshape = [], [], []
    for k, v in ck["model"].items():
        a, b = stat(v)
        mstat.append(a), mhead.append(np.pad(b, (0, 8 - len(b)))), mshape.append(json.dumps(list(v.shape)))
    out["model_stats"], out["model_head"], out["model_shapes"] = np.array(mstat), np.array(mhead), np.array(mshape)
    so = ck["optimizer"]
    groups = []
    for g in so["param_groups"]:
        groups.append({k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in g.items()})
    out["param_groups_json"] = np.array(json.dumps(groups))
    out["opt_state_keys"] = np.array(sorted(next(iter(so["state"].values())).keys()))
    out["opt_indices"] = np.array(sorted(so["state"].keys()))
    ostat, oshape = [], []
    for i in sorted(so["state"].keys()):
        st = so["state"][i]
        ostat.append(np.concatenate([stat(st["exp_avg"])[0], stat(st["exp_avg_sq"])[0], [float(st["step"])]]))
        oshape.append(json.dumps(list(st["exp_avg"].shape)))
    out["opt_stats"], out["opt_shapes"] = np.array(ostat), np.array(oshape)
    out["opt_step_is_tensor"] = np.array(isinstance(next(iter(so["state"].values()))["step"], torch.Tensor))
    np.savez_compressed(os.path.join(args.out, "ckpt_tiny.npz"), **out)
    print("ckpt_tiny: top keys", sorted(ck.keys()), "epoch", ck["epoch"], "losses", first, "third step", third, "groups", [len(g["params"]) for g in so["param_groups"]])


def make_next(args, ref_mg, ref_mf, O):
    """Fixtures of the 'next' rows:
      * vis.npz: the reconstruction arithmetic of run_videomae_vis.py:150-180.  It is inline in that script's main()
        (between video decoding and JPEG writing), so the assignment statements of those lines are taken from the file AT
        GENERATION TIME and executed on seeded inputs; the PIL / file-writing statements are left out.
      * finetune_tiny.npz / finetune_vitb.npz: modeling_finetune.VisionTransformer forward (features + logits)."""
    import re
    import textwrap
    import einops
    cfg = O.VIT_B
    img = O.keyed_clips(1, cfg, base_seed=2000)
    np.random.seed(10)
    m = ref_mg.TubeMaskingGenerator(cfg.grid, 0.9)()
    mask = torch.from_numpy(m)[None].flatten(1).to(torch.bool)
    outputs = torch.from_numpy(np.random.RandomState(77).standard_normal((1, int(mask.sum()), cfg.patch_dim)).astype(np.float32))
    lines = open(os.path.join(REF, "run_videomae_vis.py")).read().split("\n")[146:181]
    keep = [l for l in lines if re.match(r"\s*(mean|std|ori_img|img_squeeze|img_norm|img_patch|mask|rec_img|img_mask)(\[bool_masked_pos\])?\s*=", l)]
    code = textwrap.dedent("\n".join(keep))
    print("---- executed reference statements (run_videomae_vis.py:150-180)\n" + code + "\n----")
    ns = dict(torch=torch, rearrange=einops.rearrange, IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406),
              IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225), device="cpu", img=img, outputs=outputs, bool_masked_pos=mask,
              patch_size=(16, 16))
    exec(code, ns)
    fx = {"mask": m}
    for k in ("ori_img", "rec_img", "img_mask"):
        t = ns[k]
        fx[k + "_head"] = t[0, :, :2, :48, :48].numpy()
        fx[k + "_tail"] = t[0, :, -2:, -48:, -48:].numpy()
        fx[k + "_sum"] = np.array(t.double().sum().item())
        fx[k + "_sqsum"] = np.array((t.double() ** 2).sum().item())
        fx[k + "_framesum"] = t.double().sum(dim=(0, 1, 3, 4)).numpy()
    np.savez_compressed(os.path.join(args.out, "vis.npz"), **fx)
    print("vis: rec sum", fx["rec_img_sum"], "masked sum", fx["img_mask_sum"])

    for tag, c, ncls, nb in (("tiny", O.TINY, 10, 2), ("vitb", O.VIT_B, 400, 1)):
        model = ref_mf.VisionTransformer(img_size=c.img_size, patch_size=c.patch_size, num_classes=ncls, embed_dim=c.enc_dim,
                                         depth=c.enc_depth, num_heads=c.enc_heads, mlp_ratio=c.mlp_ratio, qkv_bias=True,
                                         norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), all_frames=c.num_frames,
                                         tubelet_size=c.tubelet, use_mean_pooling=True, init_scale=1.0)
        P = O.finetune_keyed_params(c, ncls)
        r = model.load_state_dict(P, strict=True)
        assert not r.missing_keys and not r.unexpected_keys
        assert [k for k, _ in model.named_parameters()] == list(P.keys()), "param order differs from oracle schema"
        model.eval()
        x = O.keyed_clips(nb, c, base_seed=3000)
        with torch.no_grad():
            feat = model.forward_features(x)
            logits = model(x)
        np.savez_compressed(os.path.join(args.out, f"finetune_{tag}.npz"), features=feat.numpy(), logits=logits.numpy())
        print(f"finetune_{tag}: logits[0,:4]", logits[0, :4].tolist())


if __name__ == "__main__":
    main()

"""Launcher of the pretraining path -- the caller on top of ``train_one_epoch`` (reference: run_mae_pretraining.py:22-311 and
its motion-box twin run_mae_pretraining_BB.py), SURVEY.md 8f "callers either side of the path".

    python -m mofo_amd.run_mae_pretraining --batch_size 32 --epochs 2 --synthetic_clips 256 --output_dir /tmp/run
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m mofo_amd.run_mae_pretraining ...

Same flag names and the same arithmetic around the epoch loop as the reference: seed + rank (:164-167), window size from
the model's patch size (:173-175), steps per epoch = len(dataset) // batch // world (:184), lr / min_lr / warmup_lr scaled by
global_batch / 256 (:216-219), per-step cosine tables for lr and weight decay (:233-241), auto-resume (:243-244),
checkpoint every ``save_ckpt_freq`` epochs and at the end (:279-283), one JSON line per epoch in ``output_dir/log.txt``
(:285-292).  What is NOT here is the reference's storage side: decord video decoding, the Epic-Kitchens / SSV2 annotation
readers, augmentation, wandb / tensorboard.  The dataset is a plug: ``--synthetic_clips N`` builds ``SyntheticClips`` --
items with exactly the layout ``VideoMAE.__getitem__`` returns (kinetics.py:492-495: ``(clip f32 [3,T,H,W], mask f64 [N])``;
with ``--mask_ratio_BB`` the box variant's ``(clip, boxes int [T,4], mask)``) -- and ``Pretrainer(args, dataset=...)`` takes
any dataset with that item layout.  ``--uint8_frames`` makes the items the loader's pre-normalisation ``Stack()`` output
(uint8 [H,W,T*3]); the model then normalises inside its pixel-reading kernels (a quarter of the host-to-device bytes).
"""
import argparse
import datetime
import json
import os
import random
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL's peer mappings (multi-process runs)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before the HIP runtime starts (see mofo_amd/__init__.py)

import numpy as np
import torch

from . import utils
from .dist import DataParallel
from .engine_for_pretraining import train_one_epoch, train_one_epoch_BB
from .masking_generator import TubeMaskingGenerator, TubeMaskingGenerator_BB
from .modeling_pretrain import create_model
from .optim_factory import create_optimizer

# (flag, default, type or action, help) -- names and defaults of run_mae_pretraining.py:22-131 for everything the path reads
_FLAGS = [
    ("batch_size", 12, int, "clips per GPU"), ("epochs", 800, int, ""), ("save_ckpt_freq", 50, int, ""),
    ("model", "pretrain_videomae_base_patch16_224", str, "factory name in mofo_amd.modeling_pretrain"),
    ("decoder_depth", 4, int, ""), ("mask_type", "tube", str, "only 'tube' exists in the reference (datasets.py:22)"),
    ("mask_ratio", 0.9, float, ""), ("mask_ratio_BB", None, float, "motion-box masking (run_mae_pretraining_BB.py:40): share of in-box patches forced masked"),
    ("input_size", 224, int, ""), ("drop_path", 0.0, float, "must stay 0 on this path"),
    ("normlize_target", True, "bool", "per-patch standardised pixel targets (the reference's spelling)"),
    ("opt", "adamw", str, ""), ("opt_eps", 1e-8, float, ""), ("opt_betas", (0.9, 0.95), "floats", ""),
    ("clip_grad", None, float, ""), ("weight_decay", 0.05, float, ""), ("weight_decay_end", None, float, ""),
    ("lr", 1.5e-4, float, "per 256 clips of global batch"), ("warmup_lr", 1e-6, float, ""), ("min_lr", 1e-5, float, ""),
    ("warmup_epochs", 40, int, ""), ("warmup_steps", -1, int, ""),
    ("num_frames", 16, int, ""), ("sampling_rate", 2, int, "dataset side; unused by synthetic clips"),
    ("output_dir", "", str, "checkpoints + log.txt; empty = do not save"), ("device", "cuda", str, ""), ("seed", 0, int, ""),
    ("resume", "", str, ""), ("auto_resume", True, "flag", ""), ("start_epoch", 0, int, ""),
    ("num_workers", 0, int, "DataLoader workers"), ("pin_mem", True, "flag", ""),
    ("world_size", 1, int, ""), ("local_rank", -1, int, ""), ("dist_url", "env://", str, ""),
    # not in the reference: the dataset plug
    ("synthetic_clips", 0, int, "length of the synthetic dataset (drifting sinusoid textures, deterministic per index)"),
    ("uint8_frames", False, "store_true", "dataset yields uint8 [H,W,T*3] frame stacks; normalisation runs on the GPU"),
    ("prefetch", True, "flag", "copy batch i+1 to the GPU on a side stream while step i runs (utils.DevicePrefetcher)"),
    ("device_masks", False, "store_true", "draw the tube masks on the GPU (masking_generator.DeviceTubeMaskingGenerator: the reference generator's "
                                          "distribution from a counter-based stream, no mask crosses PCIe) instead of taking the dataset's; the "
                                          "stream's position is saved in / restored from the checkpoints"),
]


def get_args(argv=None):
    ap = argparse.ArgumentParser("MOFO / VideoMAE pre-training on MI355X (mofo_amd)")
    for name, default, kind, text in _FLAGS:
        flag = "--" + name
        if kind == "flag":
            ap.add_argument(flag, action="store_true", default=default, help=text)
            ap.add_argument("--no_" + name, action="store_false", dest=name)
        elif kind == "store_true":
            ap.add_argument(flag, action="store_true", help=text)
        elif kind == "bool":
            ap.add_argument(flag, default=default, type=lambda s: str(s).lower() not in ("0", "false", "no", ""), help=text)
        elif kind == "floats":
            ap.add_argument(flag, default=default, type=float, nargs="+", help=text)
        else:
            ap.add_argument(flag, default=default, type=kind, help=text)
    return ap.parse_args(argv)


class DeviceMaskLoader:
    """a loader whose batches carry masks drawn on the device: the last element of every tuple (the dataset's host mask) is replaced
    by ``generator(batch_size)`` -- uint8 [B, N] on the GPU, 1 = masked -- which the engine takes as it is (no host round trip)"""

    def __init__(self, loader, generator, device):
        self.loader, self.generator, self.device = loader, generator, device
        self.sampler = getattr(loader, "sampler", None)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for batch in self.loader:
            yield tuple(batch[:-1]) + (self.generator(batch[0].shape[0], device=self.device),)


class SyntheticClips(torch.utils.data.Dataset):
    """Stand-in for the reference's ``VideoMAE`` dataset object (kinetics.py:402-495) with the same item layout.  Content:
    a few drifting 2-D sinusoids per clip (so neighbouring patches and frames are predictable from the visible ones and
    the reconstruction loss can actually fall), quantised to uint8 frames, then -- unless ``uint8_frames`` -- taken through
    ToTorchFormatTensor(div=True) + GroupNormalize (transforms.py:363-382, datasets.py:12-21) exactly like a decoded video.
    The mask comes from the reference's generator objects, called once per item as DataAugmentationForVideoMAE does
    (datasets.py:27-29,56-58), i.e. from numpy's global RNG of the calling (worker) process."""

    MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
    STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)

    def __init__(self, length, num_frames, input_size, window_size, mask_ratio, mask_ratio_BB=None, uint8_frames=False, seed=0):
        self.length, self.T, self.S, self.seed = int(length), int(num_frames), int(input_size), int(seed)
        self.uint8_frames = bool(uint8_frames)
        self.with_boxes = mask_ratio_BB is not None
        self.masked_position_generator = (TubeMaskingGenerator_BB(window_size, mask_ratio, mask_ratio_BB) if self.with_boxes
                                          else TubeMaskingGenerator(window_size, mask_ratio))

    def __len__(self):
        return self.length

    def frames(self, index):
        """uint8 [H, W, T*3]: the layout of Stack(roll=False) over T RGB frames (transforms.py:346-360)"""
        rng = np.random.RandomState((self.seed * 1000003 + index) % (2 ** 31))
        T, S = self.T, self.S
        yy, xx = np.meshgrid(np.arange(S, dtype=np.float32), np.arange(S, dtype=np.float32), indexing="ij")
        tt = np.arange(T, dtype=np.float32)[:, None, None]
        img = np.zeros((T, S, S, 3), dtype=np.float32)
        for _ in range(4):
            fx, fy = rng.uniform(-0.12, 0.12, 2)
            vx, vy = rng.uniform(-0.4, 0.4, 2)
            phase, colour = rng.uniform(0, 2 * np.pi), rng.uniform(0.1, 0.35, 3).astype(np.float32)
            wave = np.sin(fx * (xx[None] - vx * tt * 8) + fy * (yy[None] - vy * tt * 8) + phase)
            img += wave[..., None] * colour
        img = np.clip(0.5 + img * 0.5, 0.0, 1.0)
        u8 = np.round(img * 255.0).astype(np.uint8)                       # [T, H, W, 3]
        return np.ascontiguousarray(u8.transpose(1, 2, 0, 3).reshape(S, S, T * 3))

    def __getitem__(self, index):
        u8 = self.frames(index)
        if self.uint8_frames:
            clip = torch.from_numpy(u8)
        else:
            x = torch.from_numpy(u8).permute(2, 0, 1).float().div(255)    # ToTorchFormatTensor(div=True): [T*3, H, W]
            x = x.view(self.T, 3, self.S, self.S)
            x = (x - torch.from_numpy(self.MEAN).view(1, 3, 1, 1)) / torch.from_numpy(self.STD).view(1, 3, 1, 1)
            clip = x.transpose(0, 1).contiguous()                          # kinetics.py:492-493: (T,C,H,W) -> (C,T,H,W)
        if self.with_boxes:
            rng = np.random.RandomState((self.seed * 7919 + index) % (2 ** 31))
            x1, y1 = rng.randint(0, max(1, self.S - 63), 2)
            w, h = rng.randint(32, max(33, self.S // 2 + 49), 2)
            box = np.array([x1, y1, min(self.S, x1 + w), min(self.S, y1 + h)])
            boxes = np.tile(box, (self.T, 1))
            return clip, torch.from_numpy(boxes), self.masked_position_generator(boxes)
        return clip, self.masked_position_generator()


class Pretrainer:
    """everything run_mae_pretraining.py's ``main`` sets up, as an object: ``Pretrainer(args).fit()``"""

    def __init__(self, args, dataset=None):
        self.args = args
        random.seed(args.seed), np.random.seed(args.seed), torch.manual_seed(args.seed)      # seed_everything, :147-153
        utils.init_distributed_mode(args)
        if not hasattr(args, "distributed"):
            args.distributed = False
        self.device = torch.device(args.device)
        seed = args.seed + utils.get_rank()
        torch.manual_seed(seed), np.random.seed(seed)
        if args.mask_type != "tube":
            raise NotImplementedError("mask_type: the reference builds a generator for 'tube' only (datasets.py:22)")

        model = create_model(args.model, pretrained=False, drop_path_rate=args.drop_path, drop_block_rate=None,
                             decoder_depth=args.decoder_depth, **({"num_frames": args.num_frames} if args.num_frames != 16 else {}),
                             **({"img_size": args.input_size} if args.input_size != 224 else {}))
        patch = model.encoder.patch_embed.patch_size
        args.window_size = (args.num_frames // 2, args.input_size // patch[0], args.input_size // patch[1])
        args.patch_size = patch
        self.with_boxes = args.mask_ratio_BB is not None

        if dataset is None:
            if args.synthetic_clips <= 0:
                raise SystemExit("no dataset: pass --synthetic_clips N, or construct Pretrainer(args, dataset=...) with items "
                                 "(clip, mask) / (clip, boxes, mask) -- video decoding is outside this package")
            dataset = SyntheticClips(args.synthetic_clips, args.num_frames, args.input_size, args.window_size, args.mask_ratio,
                                     args.mask_ratio_BB, args.uint8_frames, args.seed)
        world, rank = utils.get_world_size(), utils.get_rank()
        self.steps_per_epoch = len(dataset) // args.batch_size // world
        if self.steps_per_epoch < 1:
            raise SystemExit(f"dataset of {len(dataset)} clips gives no full batch of {args.batch_size} x {world}")
        self.sampler = torch.utils.data.DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True)
        self.loader = torch.utils.data.DataLoader(dataset, sampler=self.sampler, batch_size=args.batch_size, num_workers=args.num_workers,
                                                  pin_memory=args.pin_mem and self.device.type == "cuda", drop_last=True,
                                                  worker_init_fn=utils.seed_worker)

        if args.prefetch and self.device.type == "cuda":
            self.loader = utils.DevicePrefetcher(self.loader, self.device)
        if getattr(args, "device_masks", False):
            if self.with_boxes or self.device.type != "cuda":
                raise SystemExit("--device_masks: tube masks on a GPU only (the motion-box generator reads per-clip boxes on the host)")
            from .masking_generator import DeviceTubeMaskingGenerator
            args.mask_generator = DeviceTubeMaskingGenerator(args.window_size, args.mask_ratio, seed=args.seed)   # utils.save_model / auto_load_model carry its state
            self.loader = DeviceMaskLoader(self.loader, args.mask_generator, self.device)
        model.to(self.device)
        n_vis = args.window_size[0] * (args.window_size[1] * args.window_size[2] - int(args.mask_ratio * args.window_size[1] * args.window_size[2]))
        model.set_visible_tokens(n_vis)                                   # known on the host: no device sync on the first batch
        self.model_without_ddp = model
        self.n_parameters = sum(p.numel() for p in model.parameters() if p.requires_grad)
        total_batch = args.batch_size * world
        args.lr, args.min_lr, args.warmup_lr = (v * total_batch / 256 for v in (args.lr, args.min_lr, args.warmup_lr))
        self.say(f"model {args.model}: {self.n_parameters / 1e6:.2f} M parameters; window {args.window_size}, {n_vis} visible tokens per clip")
        self.say(f"LR = {args.lr:.8f}; global batch {total_batch}; {self.steps_per_epoch} steps per epoch")
        self.model = DataParallel(model, device_ids=[getattr(args, "gpu", 0)]) if args.distributed else model
        self.optimizer = create_optimizer(args, self.model_without_ddp)
        self.loss_scaler = utils.NativeScalerWithGradNormCount()
        self.lr_values = _quiet(utils.cosine_scheduler, args.lr, args.min_lr, args.epochs, self.steps_per_epoch,
                                warmup_epochs=args.warmup_epochs, warmup_steps=args.warmup_steps)
        if args.weight_decay_end is None:
            args.weight_decay_end = args.weight_decay
        self.wd_values = _quiet(utils.cosine_scheduler, args.weight_decay, args.weight_decay_end, args.epochs, self.steps_per_epoch)
        if args.output_dir:
            os.makedirs(args.output_dir, exist_ok=True)
            utils.auto_load_model(args=args, model=self.model, model_without_ddp=self.model_without_ddp, optimizer=self.optimizer,
                                  loss_scaler=self.loss_scaler)

    @staticmethod
    def say(text):
        if utils.is_main_process():
            print(text, flush=True)

    def fit(self):
        a = self.args
        epoch_fn = train_one_epoch_BB if self.with_boxes else train_one_epoch
        history, t0 = [], time.time()
        for epoch in range(a.start_epoch, a.epochs):
            self.sampler.set_epoch(epoch)
            stats = epoch_fn(self.model, self.loader, self.optimizer, self.device, epoch, self.loss_scaler, a.clip_grad,
                             log_writer=None, start_steps=epoch * self.steps_per_epoch, lr_schedule_values=self.lr_values,
                             wd_schedule_values=self.wd_values, patch_size=a.patch_size[0], normlize_target=a.normlize_target)
            last = epoch + 1 == a.epochs
            if a.output_dir and ((epoch + 1) % a.save_ckpt_freq == 0 or last):
                utils.save_model(args=a, model=self.model, model_without_ddp=self.model_without_ddp, optimizer=self.optimizer,
                                 loss_scaler=self.loss_scaler, epoch=epoch)
            record = {**{f"train_{k}": v for k, v in stats.items()}, "epoch": epoch, "n_parameters": self.n_parameters}
            history.append(record)
            if a.output_dir and utils.is_main_process():
                with open(os.path.join(a.output_dir, "log.txt"), mode="a", encoding="utf-8") as f:
                    f.write(json.dumps(record) + "\n")
        self.say(f"Training time {datetime.timedelta(seconds=int(time.time() - t0))}")
        return history


def _quiet(fn, *a, **k):
    """the schedule builder prints its warm-up length on every rank; keep rank 0's line only"""
    if utils.is_main_process():
        return fn(*a, **k)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main(argv=None):
    trainer = Pretrainer(get_args(argv))
    history = trainer.fit()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return history


if __name__ == "__main__":
    main()

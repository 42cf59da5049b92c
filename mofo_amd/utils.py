"""Drop-in for the parts of the reference's ``utils.py`` that the pretraining path touches: the loss-scaler object
(utils.py:347-373), the per-iteration cosine schedule (:391-408), meters (:27-170), distributed helpers (:197-296) and
checkpoint save / auto-resume (:411-472).  Host-side bookkeeping; the numeric parts (gradient norm, clipping, the
optimizer update) run as HIP kernels on the flat buffers."""
import datetime
import glob
import math
import os
import time
from collections import defaultdict, deque

import numpy as np
import torch
import torch.distributed as dist


# ----------------------------------------------------------------------------------------------- meters
class SmoothedValue:
    """Windowed + global statistics of one scalar series; public surface of the reference's meter (utils.py:27-86):
    ``update``, ``median`` / ``avg`` (window), ``global_avg``, ``max``, ``value``, ``count`` / ``total``, ``fmt``, str()."""

    def __init__(self, window_size=20, fmt=None):
        self._win = deque(maxlen=window_size)
        self.count, self.total = 0, 0.0
        self.fmt = "{median:.4f} ({global_avg:.4f})" if fmt is None else fmt

    def update(self, value, n=1):
        self._win.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """sum (count, total) over the ranks; the window stays local (utils.py:45-56)"""
        if not is_dist_avail_and_initialized():
            return
        where = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        pair = torch.tensor([float(self.count), self.total], dtype=torch.float64, device=where)
        dist.barrier()
        dist.all_reduce(pair)
        self.count, self.total = int(pair[0].item()), float(pair[1].item())

    # window statistics; torch.median semantics (lower of the two middle values) as the reference prints them
    median = property(lambda self: float(np.sort(np.asarray(self._win, dtype=np.float64))[(len(self._win) - 1) // 2]))
    avg = property(lambda self: float(np.mean(np.asarray(self._win, dtype=np.float32))))
    global_avg = property(lambda self: self.total / self.count)
    max = property(lambda self: max(self._win))
    value = property(lambda self: self._win[-1])

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger:
    """Named meters + the progress printer of the epoch loop (utils.py:89-170): ``update(**scalars)``, attribute access to
    meters, ``add_meter``, ``synchronize_between_processes``, ``log_every(iterable, print_freq, header)``."""

    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **scalars):
        for name, val in scalars.items():
            if val is None:
                continue
            val = val.item() if isinstance(val, torch.Tensor) else val
            if not isinstance(val, (float, int)):
                raise TypeError(f"meter {name}: {type(val).__name__} is not a scalar")
            self.meters[name].update(val)

    def __getattr__(self, name):
        meters = self.__dict__.get("meters", {})
        if name in meters:
            return meters[name]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __str__(self):
        return self.delimiter.join(f"{k}: {m}" for k, m in self.meters.items())

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def log_every(self, iterable, print_freq, header=None):
        header = header or ''
        total = len(iterable)
        width = len(str(total))
        step_time, wait_time = SmoothedValue(fmt='{avg:.4f}'), SmoothedValue(fmt='{avg:.4f}')
        t_start = t_mark = time.time()
        for i, item in enumerate(iterable):
            wait_time.update(time.time() - t_mark)
            yield item
            step_time.update(time.time() - t_mark)
            if i % print_freq == 0 or i == total - 1:
                eta = datetime.timedelta(seconds=int(step_time.global_avg * (total - i)))
                fields = [header, f"[{i:{width}d}/{total}]", f"eta: {eta}", str(self), f"time: {step_time}", f"data: {wait_time}"]
                print(self.delimiter.join(fields))
            t_mark = time.time()
        spent = time.time() - t_start
        print(f"{header} Total time: {datetime.timedelta(seconds=int(spent))} ({spent / max(total, 1):.4f} s / it)")


# ----------------------------------------------------------------------------------------------- distributed
def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def init_distributed_mode(args):
    """utils.py:255-296, env:// launch (torch.distributed.run / launch): one process per GPU, backend 'nccl' = RCCL."""
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ['WORLD_SIZE'])
        args.gpu = int(os.environ.get('LOCAL_RANK', 0))
    else:
        args.distributed = False
        return
    args.distributed = True
    backend = getattr(args, "dist_backend", None) or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        torch.cuda.set_device(args.gpu)
    args.dist_backend = backend
    dist.init_process_group(backend=backend, init_method=getattr(args, "dist_url", "env://"), world_size=args.world_size, rank=args.rank)
    dist.barrier()


def seed_worker(worker_id):
    """utils.py:196-199: numpy / random of a DataLoader worker follow torch's per-worker seed (the mask generators draw
    from numpy's global RNG inside the workers)"""
    import random
    s = torch.initial_seed() % 2 ** 32
    np.random.seed(s)
    random.seed(s)


# ----------------------------------------------------------------------------------------------- input pipeline
class DevicePrefetcher:
    """Wrap a DataLoader so that batch i+1's host->device copy of the CLIP tensor runs on its own HIP stream while step i
    computes (the reference copies on the compute stream at the top of every step, engine_for_pretraining.py:39-40).
    Measured at ViT-B, 32 clips per step from pinned host memory (tools/pcie_rate.py): f32 clips 19.0 ms/step unhidden
    (308 MB over PCIe) vs 12.8 resident; uint8 frame stacks 14.4 ms unhidden (77 MB).  Masks / boxes stay on the host (a few
    KB; the engine reads the visible-token count from them without a device sync).  Yields the loader's tuples unchanged
    except that element 0 lives on ``device``."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.sampler = getattr(loader, "sampler", None)

    def __len__(self):
        return len(self.loader)

    def _stage(self, it):
        try:
            batch = next(it)
        except StopIteration:
            return None
        with torch.cuda.stream(self.stream):
            on_dev = batch[0].to(self.device, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return (on_dev,) + tuple(batch[1:]), ready, batch[0]      # the host source stays referenced until the copy is consumed

    def __iter__(self):
        it = iter(self.loader)
        staged = self._stage(it)
        while staged is not None:
            batch, ready, _src = staged
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ready)
            batch[0].record_stream(cur)
            staged = self._stage(it)        # enqueue the NEXT copy before the consumer enqueues this step's kernels
            yield batch


# ----------------------------------------------------------------------------------------------- scaler / norms
class NativeScalerWithGradNormCount:
    """utils.py:347-373.  The reference wraps torch.cuda.amp.GradScaler for fp16; this path computes in bf16 with fp32
    accumulation, so the scale is the constant 1.0 -- the call contract (backward -> norm or clip -> optimizer step,
    returns the norm; ``state_dict()['scale']``) is what the engine relies on (engine_for_pretraining.py:175-177)."""
    state_dict_key = "amp_scaler"

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        if create_graph:
            raise NotImplementedError("second-order optimizers are not on the pretraining path")
        loss.backward()
        if not update_grad:
            return None
        model = getattr(optimizer, "model", None)
        if model is None:
            raise TypeError("use mofo_amd.optim_factory.create_optimizer: the fused step works on the model's flat buffers")
        sync = getattr(model, "_grad_sync", None)
        rt = model.runtime()
        if not clip_grad:
            # utils.py:376-388: the norm is only REPORTED here, so the AdamW pass computes it from the gradients it reads anyway;
            # under data parallelism each gradient range is updated as soon as ITS all-reduce has landed
            ranges = sync.drain() if sync is not None else None
            if sync is not None and ranges is None:
                sync.finish()
            optimizer.step(norm_out=rt.norm_out, ranges=ranges)
            return rt.norm_out
        if sync is not None:
            sync.finish()              # clipping needs the global norm of the fully exchanged gradients first
        norm = rt.grad_norm()                  # utils.py:376-388, device scalar; needed before the update for clip_grad_norm_
        optimizer.step(grad_norm=norm, max_norm=clip_grad)
        return norm

    def state_dict(self):
        return {"scale": 1.0}

    def load_state_dict(self, state_dict):
        pass


def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    """utils.py:376-388 on arbitrary parameter lists (torch ops; the engine uses the fused flat-buffer norm instead)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad.detach() for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.)
    if norm_type == math.inf:
        return max(g.abs().max() for g in grads)
    return torch.norm(torch.stack([torch.norm(g, norm_type) for g in grads]), norm_type)


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    """utils.py:391-408: one value per optimizer step -- linear warm-up from ``start_warmup_value`` to ``base_value`` (over
    ``warmup_steps`` if given, else ``warmup_epochs`` epochs; only when ``warmup_epochs > 0``, as in the reference), then half
    a cosine from ``base_value`` to ``final_value``.  float64 numpy, the reference's own rounding."""
    total = epochs * niter_per_ep
    n_warm = warmup_steps if warmup_steps > 0 else warmup_epochs * niter_per_ep
    print("Set warmup steps = %d" % n_warm)
    ramp = np.linspace(start_warmup_value, base_value, n_warm) if warmup_epochs > 0 else np.array([])
    n_cos = total - n_warm
    tail = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * k / n_cos)) for k in range(n_cos)])
    table = np.concatenate((ramp, tail))
    if len(table) != total:
        raise AssertionError(f"schedule has {len(table)} entries for {total} steps (warmup_steps without warmup_epochs?)")
    return table


# ----------------------------------------------------------------------------------------------- checkpoints
def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler, model_ema=None):
    """utils.py:411-433: output_dir/checkpoint-{epoch}.pth = {'model','optimizer','epoch','scaler','args'} from rank 0.
    'model' uses the reference's state_dict names/shapes, so the reference's fine-tuning loader reads it."""
    if loss_scaler is None:
        raise NotImplementedError("deepspeed checkpoints belong to fine-tuning (out of scope)")
    path = os.path.join(args.output_dir, 'checkpoint-%s.pth' % str(epoch))
    # beyond the reference's five keys (its loaders ignore unknown ones): where the device-side mask generator stands, so that a
    # resumed run continues the mask stream instead of replaying it from step 0 (args.mask_generator: masking_generator.DeviceTubeMaskingGenerator).
    # The generator OBJECT never enters the file: 'args' is pickled, and an instance of a mofo_amd class inside it would make the
    # checkpoint unreadable for the reference's loaders (any environment without this package) -- only its plain state dict travels.
    gen = getattr(args, "mask_generator", None)
    saved_args = args
    if gen is not None:
        import copy
        saved_args = copy.copy(args)
        try:
            delattr(saved_args, "mask_generator")
        except AttributeError:
            pass
    state = {'model': {k: v.detach().cpu() for k, v in model_without_ddp.state_dict().items()},
             'optimizer': optimizer.state_dict(), 'epoch': epoch, 'scaler': loss_scaler.state_dict(), 'args': saved_args}
    if gen is not None and hasattr(gen, "state_dict"):
        state['mask_generator'] = gen.state_dict()
    rt = getattr(model_without_ddp, "_rt", None)
    f8 = rt.fp8_state_dict() if rt is not None and hasattr(rt, "fp8_state_dict") else None
    if f8 is not None:
        state['fp8'] = f8                      # MOFO_FP8=1: the delayed activation scales (plain tensors; ignored by the reference's loaders)
    save_on_master(state, path)


def auto_load_model(args, model, model_without_ddp, optimizer, loss_scaler, model_ema=None):
    """utils.py:436-472: pick the highest checkpoint-*.pth in output_dir when auto_resume, restore model/optimizer/epoch."""
    if getattr(args, "auto_resume", False) and not getattr(args, "resume", ""):
        best = -1
        for ckpt in glob.glob(os.path.join(args.output_dir, 'checkpoint-*.pth')):
            t = ckpt.split('-')[-1].split('.')[0]
            if t.isdigit():
                best = max(int(t), best)
        if best >= 0:
            args.resume = os.path.join(args.output_dir, 'checkpoint-%d.pth' % best)
        print("Auto resume checkpoint: %s" % getattr(args, "resume", ""))
    path = getattr(args, "resume", "")
    if not path:
        return
    state = torch.load(path, map_location='cpu', weights_only=False)
    model_without_ddp.load_state_dict(state['model'])
    print("Resume checkpoint %s" % path)
    if {'optimizer', 'epoch'} <= set(state):            # a full training checkpoint: continue after its epoch
        optimizer.load_state_dict(state['optimizer'])
        args.start_epoch = state['epoch'] + 1
        if 'scaler' in state:
            loss_scaler.load_state_dict(state['scaler'])
        gen = getattr(args, "mask_generator", None)
        if gen is not None and hasattr(gen, "load_state_dict") and 'mask_generator' in state:
            gen.load_state_dict(state['mask_generator'])
        if 'fp8' in state and hasattr(model_without_ddp, "runtime") and next(model_without_ddp.parameters()).is_cuda:
            model_without_ddp.runtime().load_fp8_state_dict(state['fp8'])
        print("With optim & sched!")

"""Drop-in for the reference's ``engine_for_pretraining.py``: ``train_one_epoch`` / ``train_one_epoch_BB`` with the
reference's signature and returned meter dict.  Per step (engine_for_pretraining.py:29-69,168-208):
  schedule writes -> batch to device -> [target build + forward + MSE] -> zero_grad ->
  loss_scaler(backward, grad norm / clip, optimizer step) -> loss read-back + finite check (while the backward runs) -> meters.
Not reproduced on purpose: the debug block at :74-166 that writes B x 16 x 3 PNGs every step, and in the BB
variant the bbox rasterisation at :243-249 whose result is never used (the loss weighting is commented out at :294-303).
"""
import math
import sys
from typing import Iterable

import torch

from . import utils


def _step_common(model, videos, bool_masked_pos, optimizer, loss_scaler, max_norm, normlize_target, it, lr_schedule_values,
                 wd_schedule_values):
    # per-step schedule tables (engine_for_pretraining.py:31-37): lr scaled per group, weight decay only where it is on
    for group in optimizer.param_groups:
        if lr_schedule_values is not None:
            group["lr"] = lr_schedule_values[it] * group["lr_scale"]
        if wd_schedule_values is not None and group["weight_decay"] > 0:
            group["weight_decay"] = wd_schedule_values[it]
    raw = getattr(model, "module", model)
    # mask arrives from the loader as float64 [B, N] (1 = masked); the visible count is known on the host -> no device sync
    if not bool_masked_pos.is_cuda and raw._n_vis_cache is None:
        raw.set_visible_tokens(int((bool_masked_pos[0].reshape(-1) == 0).sum()))
    mask = bool_masked_pos.flatten(1).to(torch.uint8 if not bool_masked_pos.is_cuda else bool_masked_pos.dtype)
    loss = model.forward_loss(videos, mask, normlize_target) if hasattr(model, "forward_loss") else None
    if loss is None:
        raise TypeError("model must be a mofo_amd PretrainVisionTransformer (optionally wrapped in mofo_amd.dist.DataParallel)")
    # The reference reads the loss (a device sync) BEFORE it launches the backward (:69 then :172-176), which leaves the
    # GPU idle for the round trip.  Here backward + grad-norm + AdamW are enqueued first and the loss is read while they
    # run; a non-finite loss still ends the process at the same place, and the update itself is gated ON THE DEVICE by the
    # loss / status words (mofo_adamw_gated): a step the reference would have stopped before backward (:168-176) leaves
    # parameters, moments and the bf16 shadow untouched.
    optimizer.zero_grad()
    grad_norm = loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=None, create_graph=False)
    loss_value = loss.item()
    raw.check_status()
    if not math.isfinite(loss_value):
        print("Loss is {}, stopping training".format(loss_value))
        sys.exit(1)
    loss_scale_value = loss_scaler.state_dict()["scale"]
    torch.cuda.synchronize()
    return loss_value, grad_norm, loss_scale_value


def _group_summary(optimizer):
    """what the reference logs about the optimizer each step (engine_for_pretraining.py:184-196): the largest and smallest
    group lr, and the weight decay of the (last) decayed group"""
    rates = [g["lr"] for g in optimizer.param_groups]
    decays = [g["weight_decay"] for g in optimizer.param_groups if g["weight_decay"] > 0]
    return min([10.] + rates), max([0.] + rates), (decays[-1] if decays else None)


def _train(model, data_loader, optimizer, device, epoch, loss_scaler, max_norm, patch_size, normlize_target, log_writer,
           lr_scheduler, start_steps, lr_schedule_values, wd_schedule_values, has_bbox):
    model.train()
    if patch_size != 16:
        raise NotImplementedError("the fused target/loss kernel is built for patch_size 16")
    meters = utils.MetricLogger(delimiter="  ")
    for name in ("lr", "min_lr"):                      # shown as the current value, not a windowed median
        meters.add_meter(name, utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    for step, batch in enumerate(meters.log_every(data_loader, 10, f"Epoch: [{epoch}]")):
        videos, bool_masked_pos = batch[0], batch[-1]   # (videos, mask) or, for the motion-box loader, (videos, boxes, mask)
        loss_value, grad_norm, scale = _step_common(model, videos, bool_masked_pos, optimizer, loss_scaler, max_norm, normlize_target,
                                                    start_steps + step, lr_schedule_values, wd_schedule_values)
        lo, hi, decay = _group_summary(optimizer)
        # meter order = the reference's print order: loss, loss_scale, lr, min_lr, weight_decay, grad_norm
        record = {"loss": loss_value, "loss_scale": scale, "lr": hi, "min_lr": lo, "weight_decay": decay, "grad_norm": grad_norm}
        for name, value in record.items():
            meters.update(**{name: value})
        if log_writer is not None:
            for name, value in record.items():
                log_writer.update(head="loss" if name == "loss" else "opt", **{name: value})
            log_writer.set_step()
        if lr_scheduler is not None:
            lr_scheduler.step_update(start_steps + step)
    meters.synchronize_between_processes()
    print("Averaged stats:", meters)
    return {name: m.global_avg for name, m in meters.meters.items()}


def train_one_epoch(model: torch.nn.Module, data_loader: Iterable, optimizer: torch.optim.Optimizer,
                    device: torch.device, epoch: int, loss_scaler, max_norm: float = 0, patch_size: int = 16,
                    normlize_target: bool = True, log_writer=None, lr_scheduler=None, start_steps=None,
                    lr_schedule_values=None, wd_schedule_values=None):
    """engine_for_pretraining.py:16-212"""
    return _train(model, data_loader, optimizer, device, epoch, loss_scaler, max_norm, patch_size, normlize_target, log_writer,
                  lr_scheduler, start_steps, lr_schedule_values, wd_schedule_values, has_bbox=False)


def train_one_epoch_BB(model: torch.nn.Module, data_loader: Iterable, optimizer: torch.optim.Optimizer,
                       device: torch.device, epoch: int, loss_scaler, max_norm: float = 0, patch_size: int = 16,
                       normlize_target: bool = True, log_writer=None, lr_scheduler=None, start_steps=None,
                       lr_schedule_values=None, wd_schedule_values=None, loss_weight=None):
    """engine_for_pretraining.py:215-468: batches are (videos, bbox, mask); same arithmetic as train_one_epoch
    (MSELoss(reduction='none').mean() == MSELoss()); only the mask generator upstream differs."""
    return _train(model, data_loader, optimizer, device, epoch, loss_scaler, max_norm, patch_size, normlize_target, log_writer,
                  lr_scheduler, start_steps, lr_schedule_values, wd_schedule_values, has_bbox=True)

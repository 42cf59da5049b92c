"""Worker of test_ranks_share_one_gpu_like_the_multi_gpu_job (launched by torch.distributed.run, one process per rank).
By default all ranks use cuda:0 and exchange through gloo (RCCL refuses two ranks on one device; MOFO_DP_TEST_BACKEND=nccl
puts one rank on each GPU over RCCL where the box has them); everything else -- DataParallel's
flat broadcast, the per-bucket asynchronous all-reduce issued from inside the replayed backward, the join before the fused
AdamW -- is the code the multi-GPU job runs.  Each rank trains on its shard of a fixed global batch and reports its loss and
parameters; rank 0 writes the comparison next to the single-process run over the whole batch."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mofo_amd import modeling_pretrain as mp, optim_factory, utils   # noqa: E402
from mofo_amd.dist import DataParallel                                  # noqa: E402
from oracle import pretrain_oracle as O                                 # noqa: E402  (test infrastructure: seeded inputs only)


class Args:
    opt, lr, weight_decay, opt_eps, opt_betas = "adamw", 1.5e-3, 0.05, 1e-8, (0.9, 0.95)


def build(cfg, dev):
    from functools import partial
    model = mp.PretrainVisionTransformer(
        img_size=cfg.img_size, patch_size=cfg.patch_size, encoder_embed_dim=cfg.enc_dim, encoder_depth=cfg.enc_depth,
        encoder_num_heads=cfg.enc_heads, encoder_num_classes=0, decoder_num_classes=cfg.patch_dim, decoder_embed_dim=cfg.dec_dim,
        decoder_depth=cfg.dec_depth, decoder_num_heads=cfg.dec_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=True,
        norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_frames=cfg.num_frames)
    return model.to(dev)


def train(model, wrapped, x, mask, steps, labels):
    opt = optim_factory.create_optimizer(Args, model)
    scaler = utils.NativeScalerWithGradNormCount()
    losses = []
    # step 1 by hand, to look at the (all-reduced) gradient before the optimizer consumes it
    loss = wrapped.forward_loss(x, mask)
    opt.zero_grad()
    loss.backward()
    sync = getattr(model, "_grad_sync", None)
    if sync is not None:
        sync.finish()
    grads = model.runtime().store.grads.detach().clone()
    opt.step()
    losses.append(float(loss.detach()))
    norms = []
    for _ in range(steps - 1):
        loss = wrapped.forward_loss(x, mask)
        opt.zero_grad()
        norms.append(float(scaler(loss, opt, clip_grad=None)))    # under DP: range-by-range AdamW, norm from its block partials
        losses.append(float(loss.detach()))
    model.check_status()
    torch.cuda.synchronize()
    params = model.runtime().store.params.detach().clone()
    # the reference call pattern -- model(videos, mask) + nn.MSELoss on the outputs (engine_for_pretraining.py:66-67) --
    # must leave the same MEAN gradient under data parallelism as the fused path (the 1/world scale sits in the backward)
    out = wrapped(x, mask)
    opt.zero_grad()
    torch.nn.MSELoss()(out, labels.to(out.device)).backward()
    if sync is not None:
        sync.finish()
    generic = model.runtime().store.grads.detach().clone()
    torch.cuda.synchronize()
    return losses, params, grads, norms, generic


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # MOFO_DP_TEST_BACKEND=nccl: one GPU per rank over RCCL (a box with >= world GPUs); default: every rank on cuda:0 over gloo
    backend = os.environ.get("MOFO_DP_TEST_BACKEND", "gloo")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", init_method="env://", world_size=world, rank=rank, device_id=dev)
    else:
        dist.init_process_group("gloo", init_method="env://", world_size=world, rank=rank)
    cfg = O.OracleConfig(img_size=64, enc_dim=192, enc_depth=6, enc_heads=3, dec_dim=128, dec_depth=2, dec_heads=2)
    per_rank, steps = 2, 3
    x_all = O.keyed_clips(per_rank * world, cfg)
    np.random.seed(5)
    from mofo_amd.masking_generator import TubeMaskingGenerator
    gen = TubeMaskingGenerator(cfg.grid, 0.75)
    mask_all = torch.from_numpy(np.stack([gen() for _ in range(per_rank * world)])).bool()
    sl = slice(rank * per_rank, (rank + 1) * per_rank)

    torch.manual_seed(100 + rank)              # DIFFERENT initial weights per rank: the wrapper must broadcast rank 0's
    model = build(cfg, dev)
    wrapped = DataParallel(model)
    assert wrapped.sync.enabled and wrapped.world_size == world
    n_seg = len(model.runtime().segments)
    labels_all = O.build_targets(x_all, mask_all, cfg)
    losses, params, grads, norms, generic = train(model, wrapped, x_all[sl].to(dev), mask_all[sl].to(dev), steps, labels_all[sl])
    assert not wrapped.sync.handles
    params = params.cpu()
    if backend == "nccl":
        gathered = [torch.empty_like(params, device=dev) for _ in range(world)]
        dist.all_gather(gathered, params.to(dev))
        gathered = [g.cpu() for g in gathered]
    else:
        gathered = [torch.empty_like(params) for _ in range(world)] if rank == 0 else None
        dist.gather(params, gathered, dst=0)
    all_losses = [None] * world
    dist.all_gather_object(all_losses, losses)
    if rank == 0:
        torch.manual_seed(100)                 # the single-process run over the WHOLE batch, from rank 0's initial weights
        ref_model = build(cfg, dev)
        ref_losses, ref_params, ref_grads, ref_norms, ref_generic = train(ref_model, ref_model, x_all.to(dev), mask_all.to(dev), steps, labels_all)
        rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
        # ... and against the CPU ORACLE on the whole global batch from rank 0's initial weights: first-step loss, global gradient
        # norm and the flat gradient tensor by tensor (the data-parallel MEAN gradient = the whole-batch gradient)
        torch.manual_seed(100)
        P0 = {k: v.detach().cpu().float().clone() for k, v in build(cfg, dev).state_dict().items()}
        o_loss, o_norm, o_grads = O.train_step(x_all, mask_all, P0, cfg)
        st = model.runtime().store
        worst = 0.0
        total = float(torch.sqrt(sum(g.double().pow(2).sum() for g in o_grads.values())))
        for n, g in o_grads.items():
            if float(g.double().norm()) < 2e-3 * total:
                continue
            o, k = st.offset[n], g.numel()
            worst = max(worst, rel(grads[o:o + k], g.reshape(-1)))
        json.dump({"world": world, "segments": n_seg, "losses": all_losses, "ref_losses": ref_losses,
                   "rank_param_diff": [rel(g, gathered[0]) for g in gathered], "params_vs_single_process": rel(gathered[0], ref_params),
                   "grads_vs_single_process": rel(grads, ref_grads), "norms": norms, "ref_norms": ref_norms,
                   "generic_grads_vs_single_process": rel(generic, ref_generic), "backend": backend,
                   "oracle_loss": float(o_loss), "oracle_grad_norm": float(o_norm), "grads_vs_oracle_worst_tensor": worst,
                   "grad_norm_step1": float(grads.double().norm())}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

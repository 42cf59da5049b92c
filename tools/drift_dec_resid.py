#!/usr/bin/env python3
"""Drift evidence for the decoder's bf16 residual stream (runtime.py: MOFO_DEC_RESID; the reference keeps the stream in f32 even
under autocast): the SAME training run -- ViT-B, 32 clips per step, 8 DIFFERENT synthetic batches taken in turn, AdamW at a fixed
learning rate -- twice from identical weights, once with the bf16 decoder stream (the default) and once with MOFO_DEC_RESID=f32.
Reports the two loss curves side by side and, at the end, the relative difference of all 218 per-tensor parameter norms.
GPU box only.  usage: drift_dec_resid.py [steps] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
NB = 8
dev = torch.device("cuda:0")


class _Args:
    opt = "adamw"
    opt_eps = 1e-8
    opt_betas = (0.9, 0.95)
    weight_decay = 0.05
    lr = 1.5e-4 * B / 256 * 8          # a learning rate at which 300 steps move the loss (the recipe's peak is reached after 40 epochs of warm-up)


def run(resid, jitter=0.0):
    """``jitter`` > 0: every initial weight is multiplied by (1 + jitter * N(0,1)) -- a perturbation BELOW the bf16 rounding of the GEMM
    operands (2^-9): how far two runs that differ by rounding-level noise alone drift apart = the floor the bf16 stream is read against"""
    os.environ["MOFO_DEC_RESID"] = resid                 # read when the runtime is built
    from mofo_amd import modeling_pretrain as mp, optim_factory, utils
    from mofo_amd.masking_generator import TubeMaskingGenerator
    torch.manual_seed(0)
    model = mp.pretrain_videomae_base_patch16_224(decoder_depth=4).to(dev)
    if jitter:
        gj = torch.Generator(device=dev).manual_seed(99)
        with torch.no_grad():
            for p_ in model.parameters():
                p_.mul_(1.0 + jitter * torch.randn(p_.shape, device=dev, generator=gj))
    init = {n: p.detach().clone() for n, p in model.named_parameters()}
    clips, mask_u8 = model.input_buffers(B, 160)
    g = torch.Generator(device=dev).manual_seed(11)
    # drifting low-frequency content + noise, so that the masked patches are partly predictable and the loss falls
    base = torch.randn(NB, B, 3, 1, 14, 14, device=dev, generator=g)
    batches = [(torch.nn.functional.interpolate(base[i].expand(B, 3, 16, 14, 14).reshape(B, 48, 14, 14), size=(224, 224), mode="bilinear")
                .reshape(B, 3, 16, 224, 224) + 0.3 * torch.randn(B, 3, 16, 224, 224, device=dev, generator=g)).contiguous() for i in range(NB)]
    np.random.seed(5)
    mgen = TubeMaskingGenerator((8, 14, 14), 0.9)
    masks = [torch.from_numpy(np.stack([mgen() for _ in range(B)]).astype(np.uint8)).to(dev) for _ in range(NB)]
    opt = optim_factory.create_optimizer(_Args, model)
    scaler = utils.NativeScalerWithGradNormCount()
    assert str(model.runtime().dec_resid).endswith("bfloat16" if resid == "bf16" else "float32")
    losses = []
    for it in range(steps):
        clips.copy_(batches[it % NB])
        mask_u8.copy_(masks[it % NB])
        loss = model.forward_loss(clips, mask_u8, True)
        opt.zero_grad()
        scaler(loss, opt, clip_grad=None)
        losses.append(float(loss))
    model.check_status()
    norms = {n: float(p.detach().double().norm()) for n, p in model.named_parameters()}
    return losses, norms, {n: p.detach().clone() for n, p in model.named_parameters()}, init


la, na, pa, p0 = run("bf16")
lb, nb, pb, _ = run("f32")
lc, nc, pc, p0c = run("f32", jitter=2.0 ** -11)
print(f"# tools/drift_dec_resid.py: ViT-B, {B} clips per step, {NB} different synthetic batches in turn, {steps} AdamW steps (lr {_Args.lr:.2e}), same initial weights")
print("# step   loss (decoder stream bf16)   loss (decoder stream f32)   relative difference")
worst = 0.0
for it in range(steps):
    rel = abs(la[it] - lb[it]) / abs(lb[it])
    worst = max(worst, rel)
    if it < 4 or it % 25 == 24 or it == steps - 1:
        print(f"{it + 1:6d}   {la[it]:.6f}                    {lb[it]:.6f}                   {rel:.2e}")
rels = sorted(((abs(na[n] - nb[n]) / max(nb[n], 1e-12), n) for n in na), reverse=True)
print(f"# worst relative loss difference over all {steps} steps: {worst:.2e}; mean over the last 25 steps: "
      f"{np.mean(la[-25:]):.6f} (bf16 stream) vs {np.mean(lb[-25:]):.6f} (f32 stream)")
print(f"# per-tensor parameter norms after {steps} steps, {len(rels)} tensors: worst relative difference {rels[0][0]:.2e} ({rels[0][1]}), "
      f"median {rels[len(rels) // 2][0]:.2e}; tensors beyond 1e-3: {sum(r > 1e-3 for r, _ in rels)}")
for r, n in rels[:5]:
    print(f"#   {r:.2e}  {n}")
# the tensors beyond 1e-3 are the ones that START at (or near) zero -- biases, LayerNorm biases, q / v biases: their norm after 300 steps is the
# accumulated update itself, and Adam's update m / sqrt(v) does not shrink with the gradient.  Measured against the distance each tensor has
# TRAVELLED from its initial value (the update both runs made), element by element:
mat = [n for n in pa if pa[n].dim() >= 2]
vec = [n for n in pa if pa[n].dim() < 2]
for kind, names in (("matrices (>= 2-D)", mat), ("vectors (biases, LayerNorm, mask_token)", vec)):
    dn = sorted(float((pa[n] - pb[n]).double().norm() / pb[n].double().norm().clamp_min(1e-30)) for n in names)
    dt = sorted(float((pa[n] - pb[n]).double().norm() / (pb[n] - p0[n]).double().norm().clamp_min(1e-30)) for n in names)
    print(f"# {kind}: {len(names)} tensors; |p_bf16 - p_f32| / |p_f32|: median {dn[len(dn) // 2]:.2e}, worst {dn[-1]:.2e}; "
          f"|p_bf16 - p_f32| / |p_f32 - p_init| (share of the distance travelled): median {dt[len(dt) // 2]:.2e}, worst {dt[-1]:.2e}")
wn = sorted(abs(na[n] - nb[n]) / nb[n] for n in mat)
print(f"# norms of the {len(mat)} matrices: worst relative difference {wn[-1]:.2e}")
# the floor: the f32-stream run against ITSELF started from weights jittered by 2^-11 relative (a quarter of a bf16 ulp)
wl = max(abs(lc[it] - lb[it]) / abs(lb[it]) for it in range(steps))
for kind, names in (("matrices (>= 2-D)", mat), ("vectors", vec)):
    dt = sorted(float(((pc[n] - p0c[n]) - (pb[n] - p0[n])).double().norm() / (pb[n] - p0[n]).double().norm().clamp_min(1e-30)) for n in names)
    print(f"# floor, {kind}: |update_f32' - update_f32| / |update_f32|: median {dt[len(dt) // 2]:.2e}, worst {dt[-1]:.2e}")
print(f"# floor, loss: worst relative difference between the two f32-stream runs {wl:.2e}")

"""Reconstruction inference of the reference's visualisation script (run_videomae_vis.py:137-181) on the HIP path --
SURVEY.md 8f rank 4.  The script's video decoding (decord) and argument parsing are outside the path; what it computes
between the model call and the JPEG writer is here:

    out = reconstruct(model, img, bool_masked_pos)      # img [B,3,T,H,W] ImageNet-normalised (or uint8 frames [B,H,W,T*3]), mask [B,N] (1/True = masked)
    out["ori_img"], out["rec_img"], out["mask_img"]     # f32 [B,3,T,H,W] in [0,1] pixel units, on the GPU
    save_frames(out, "/some/dir")                       # ori_img{t}.jpg / rec_img{t}.jpg / mask_img{t}.jpg like :154-181

The model forward (tube-masked encoder + decoder) is the pretraining forward; the de-standardisation of the predicted
patches with each patch's own mean / std and the re-assembly into a video is one HIP kernel (``mofo_reconstruct``) that
reads the clip once -- the reference materialises four [B,1568,512,3] intermediates (:158-172).
"""
import os

import torch

from . import ops
from .runtime import F32


@torch.no_grad()
def reconstruct(model, img, bool_masked_pos):
    """run_videomae_vis.py:137-180.  ``model``: mofo_amd PretrainVisionTransformer on the GPU.  Returns a dict of f32
    [B,3,T,H,W] tensors: ``ori_img`` (:152), ``rec_img`` (:169-173, before the writer's clamp) and ``mask_img`` (:180)."""
    raw = getattr(model, "module", model)
    if img.dim() == 4:          # the script's single clip [3,T,H,W] (:141)
        img, bool_masked_pos = img.unsqueeze(0), bool_masked_pos.reshape(1, -1)
    rt, w = raw._prepare(img, bool_masked_pos.flatten(1))
    rt.store.refresh_shadow()
    rt.forward(w)                                       # predictions bf16 [B*n_msk, 1536], rows in msk_idx order
    raw.check_status(w)
    d = rt.d
    if w.src_u8:                                        # uint8 frame stack given: the video writer needs the f32 clip once
        ops.ingest_u8(w.frames_u8, w.clips)
    out = {k: torch.empty_like(w.clips, dtype=F32) for k in ("ori_img", "rec_img", "mask_img")}
    ops.reconstruct(w.clips, d.tubelet, d.patch_size, w.msk_idx, w.pred, out["rec_img"], masked=out["mask_img"], ori=out["ori_img"])
    return out


def save_frames(out, save_path, clip=0):
    """the JPEG writer of run_videomae_vis.py:154-156,174-181 (ToPILImage: x*255 -> uint8; rec is clamped to [0, 0.996])."""
    from PIL import Image
    os.makedirs(save_path, exist_ok=True)

    def frames(t):   # [3,T,H,W] f32 -> list of HxWx3 uint8 (torchvision ToPILImage: mul(255).byte())
        return [Image.fromarray(f) for f in t.mul(255).byte().permute(1, 2, 3, 0).cpu().numpy()]
    for name, t in (("ori_img", out["ori_img"][clip]), ("rec_img", out["rec_img"][clip].clamp(0, 0.996)), ("mask_img", out["mask_img"][clip])):
        for i, im in enumerate(frames(t)):
            im.save(os.path.join(save_path, f"{name}{i}.jpg"))

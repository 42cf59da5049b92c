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
import argparse
import os

import numpy as np
import torch

from . import ops
from .runtime import F32


@torch.no_grad()
def reconstruct(model, img, bool_masked_pos):
    """run_videomae_vis.py:137-180.  ``model``: mofo_amd PretrainVisionTransformer on the GPU.  Returns a dict of f32
    [B,3,T,H,W] tensors: ``ori_img`` (:152), ``rec_img`` (:169-173, before the writer's clamp) and ``mask_img`` (:180)."""
    raw = getattr(model, "module", model)
    if img.dim() == (3 if img.dtype == torch.uint8 else 4):   # a single clip: [3,T,H,W] (:141) or one uint8 frame stack [H,W,T*3]
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


def get_args(argv=None):
    """the reference script's arguments (run_videomae_vis.py:48-72); ``img_path`` is a ``.npy`` of decoded frames here"""
    ap = argparse.ArgumentParser("VideoMAE / MOFO reconstruction on MI355X (mofo_amd)")
    ap.add_argument("img_path", type=str, help="frames as .npy: uint8 [T, H, W, 3] already cropped to input_size (video decoding is outside this package)")
    ap.add_argument("save_path", type=str, help="directory for ori_img*.jpg / rec_img*.jpg / mask_img*.jpg")
    ap.add_argument("model_path", type=str, help="checkpoint written by utils.save_model (or by the reference)")
    ap.add_argument("--mask_type", default="tube", choices=["tube"], type=str)
    ap.add_argument("--num_frames", type=int, default=16)
    ap.add_argument("--decoder_depth", default=4, type=int)
    ap.add_argument("--input_size", default=224, type=int)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--mask_ratio", default=0.75, type=float)
    ap.add_argument("--model", default="pretrain_videomae_base_patch16_224", type=str)
    ap.add_argument("--seed", default=None, type=int, help="numpy seed for the mask draw")
    return ap.parse_args(argv)


def main(argv=None):
    """run_videomae_vis.py:75-181 without the decord reader: frames -> (clip, tube mask) -> model -> three JPEG series"""
    from .masking_generator import TubeMaskingGenerator
    from .modeling_pretrain import create_model
    args = get_args(argv)
    dev = torch.device(args.device)
    model = create_model(args.model, pretrained=False, drop_path_rate=0.0, drop_block_rate=None, decoder_depth=args.decoder_depth,
                         **({"num_frames": args.num_frames} if args.num_frames != 16 else {}),
                         **({"img_size": args.input_size} if args.input_size != 224 else {}))
    patch = model.encoder.patch_embed.patch_size
    window = (args.num_frames // 2, args.input_size // patch[0], args.input_size // patch[1])
    state = torch.load(args.model_path, map_location="cpu", weights_only=False)
    model.load_state_dict(state["model"] if "model" in state else state)
    model.to(dev).eval()
    frames = np.load(args.img_path)
    want = (args.num_frames, args.input_size, args.input_size, 3)
    if frames.dtype != np.uint8 or frames.shape != want:
        raise SystemExit(f"{args.img_path}: need uint8 frames {want}, got {frames.dtype} {frames.shape}")
    # the loader's Stack() layout [H, W, T*3]; ToTorchFormatTensor + GroupNormalize run inside the kernels
    stack = torch.from_numpy(np.ascontiguousarray(frames.transpose(1, 2, 0, 3).reshape(args.input_size, args.input_size, -1)))
    if args.seed is not None:
        np.random.seed(args.seed)
    mask = torch.from_numpy(TubeMaskingGenerator(window, args.mask_ratio)()).bool()
    out = reconstruct(model, stack.unsqueeze(0).to(dev), mask.unsqueeze(0).to(dev))
    save_frames(out, args.save_path)
    print(f"wrote {3 * args.num_frames} frames to {args.save_path}")
    return out


if __name__ == "__main__":
    main()

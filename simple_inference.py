#!/usr/bin/env python
"""Inference entry point (drop-in for the reference's simple_inference.py: same flags, `input:output` colon syntax,
`<name>_seg.<ext>` / `<name>_dep.png` outputs, nms / threshold overrides written into cfg.solov2).

The tensor path is the reference's: read BGR image -> resize to calc_size_preserve_ar(W, H, cfg.max_size) (cv2.INTER_LINEAR
arithmetic, no antialiasing) -> zero-pad to a multiple of 32 -> FastBaseTransform -> PlaneRecNet (eval) -> list[dict]; the
three staging steps run as one HIP launch on the uploaded uint8 frame.  The device comes from `cfg.device`.
Image file I/O and the overlay drawing use Pillow + numpy (OpenCV is not a dependency of this build); the iBims-1 `.mat`
exporters of the reference are visual / evaluation tooling outside the hot path and are not provided.
"""
import argparse
import os
from pathlib import Path

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")       # before the HIP runtime initialises: see planerecnet_amd/__init__.py
import torch  # noqa: E402

from planerecnet_amd.config import COLORS, cfg, set_cfg
from planerecnet_amd.funcs import calc_size_preserve_ar, frame_to_input


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="PlaneRecNet Inference (MI355X)")
    p.add_argument("--trained_model", default=None, type=str)
    p.add_argument("--config", default="PlaneRecNet_50_config")
    p.add_argument("--image", default=None, type=str, help="path or input:output")
    p.add_argument("--images", default=None, type=str, help="input_folder:output_folder")
    p.add_argument("--max_img", default=0, type=int)
    p.add_argument("--ibims1", default=None, type=str, help="not provided by this build")
    p.add_argument("--ibims1_pd", default=None, type=str, help="not provided by this build")
    p.add_argument("--no_mask", action="store_true")
    p.add_argument("--no_box", action="store_true")
    p.add_argument("--no_text", action="store_true")
    p.add_argument("--top_k", default=100, type=int)
    p.add_argument("--nms_mode", default="matrix", type=str, choices=["matrix", "mask"])
    p.add_argument("--score_threshold", default=0.3, type=float)
    p.add_argument("--depth_mode", default="colored", type=str, choices=["colored", "gray"])
    p.add_argument("--depth_shift", default=512, type=float)
    global args
    args = p.parse_args(argv)
    return args


def _imread_bgr(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1].copy()


def _imwrite_bgr(path, arr):
    from PIL import Image
    if arr.ndim == 3:
        Image.fromarray(arr[:, :, ::-1].astype(np.uint8)).save(path)
    elif arr.dtype == np.uint16:
        Image.fromarray(arr.astype(np.uint16)).save(path)
    else:
        Image.fromarray(arr.astype(np.uint8)).save(path)


def display_on_frame(result, frame, mask_alpha=0.5, no_mask=False, no_box=False, no_text=False):
    """Blend instance masks / boxes / scores over the (padded) BGR frame; returns (uint8 HxWx3 BGR, depth HxW float)."""
    from PIL import Image, ImageDraw
    depth = result["pred_depth"].squeeze().float().cpu().numpy()
    img = frame.float().cpu().numpy()
    if result["pred_scores"] is None:
        return img.astype(np.uint8), depth
    masks = result["pred_masks"].cpu().numpy()
    for i in range(masks.shape[0] - 1, -1, -1):
        if not no_mask:
            color = np.asarray(COLORS[(i * 5) % len(COLORS)][::-1], np.float32)      # stored RGB, frame is BGR
            m = masks[i][..., None]
            img = np.where(m, img * (1 - mask_alpha) + color * mask_alpha, img)
    pil = Image.fromarray(img[:, :, ::-1].astype(np.uint8))
    draw = ImageDraw.Draw(pil)
    boxes, scores = result["pred_boxes"].cpu().numpy(), result["pred_scores"].cpu().numpy()
    for i in range(masks.shape[0]):
        x0, y0, x1, y1 = [int(v) for v in boxes[i]]
        if not no_box:
            draw.rectangle([x0, y0, x1, y1], outline=COLORS[(i * 5) % len(COLORS)], width=1)
        if not no_text:
            draw.text((x0 + 2, y0 + 2), "plane: %.2f" % scores[i], fill=(255, 255, 255))
    return np.asarray(pil)[:, :, ::-1], depth


@torch.no_grad()
def inference_image(net, path, save_path=None, depth_mode="colored"):
    frame_np = _imread_bgr(path)
    H, W, _ = frame_np.shape
    # the decoded frame goes to the device as BYTES (page-locked, asynchronous); resize (cv2.INTER_LINEAR arithmetic), zero
    # padding to a multiple of 32 and FastBaseTransform are one HIP launch there (planerecnet_amd.funcs.frame_to_input)
    staged = torch.from_numpy(np.ascontiguousarray(frame_np))
    if torch.cuda.is_available():
        staged = staged.pin_memory()
    batch, frame = frame_to_input(staged.to(cfg.device, non_blocking=True), calc_size_preserve_ar(W, H, cfg.max_size))
    results = net(batch)
    blended, depth = display_on_frame(results[0], frame, no_mask=args.no_mask, no_box=args.no_box, no_text=args.no_text)
    name, ext = os.path.splitext(path if save_path is None else save_path)
    save_path = name + "_seg" + ext if save_path is None else save_path
    depth_path = name + "_dep.png"
    _imwrite_bgr(save_path, blended)
    if depth_mode == "colored":
        vmin, vmax = np.percentile(depth, 1), np.percentile(depth, 99)
        d = depth.clip(min=vmin, max=vmax)
        d = ((d - d.min()) / max(d.max() - d.min(), 1e-12) * 255).astype(np.uint8)
        viridis = np.stack([np.interp(d, [0, 64, 128, 192, 255], c) for c in ([68, 59, 33, 94, 253], [1, 82, 145, 201, 231], [84, 139, 140, 98, 37])], -1)
        _imwrite_bgr(depth_path, viridis[:, :, ::-1])
    else:
        _imwrite_bgr(depth_path, (depth * args.depth_shift).astype(np.uint16))
    return results


def inference_images(net, in_folder, out_folder, max_img=0, depth_mode="colored"):
    os.makedirs(out_folder, exist_ok=True)
    files = sorted(p for p in Path(in_folder).glob("*") if p.suffix in (".png", ".jpg"))
    for i, p in enumerate(files[: max_img if max_img > 0 else len(files)]):
        inference_image(net, str(p), os.path.join(out_folder, p.name), depth_mode=depth_mode)
        print("Inference images: " + p.name, end="\r")
    print("\nDone.")


def main(argv=None):
    parse_args(argv)
    torch.set_num_threads(4)                                   # host-side tensor ops are small: a wide OpenMP team only adds fork / join latency (as train.py)
    from planerecnet_amd import timer
    from planerecnet_amd.planerecnet import PlaneRecNet
    timer.disable_all()
    set_cfg(args.config)
    # the reference feeds --score_threshold into BOTH mask_thr and update_thr (quirk Q12)
    cfg.solov2.replace({"nms_type": args.nms_mode, "mask_thr": args.score_threshold, "update_thr": args.score_threshold, "top_k": args.top_k})
    if not torch.cuda.is_available():
        raise SystemExit("No GPU detected: the HIP path has no CPU fallback.")
    cfg.device = "cuda:0" if cfg.device == "cuda" else cfg.device
    net = PlaneRecNet(cfg)
    if args.trained_model is not None:
        net.load_weights(args.trained_model)
    else:
        backbone = "weights/" + cfg.backbone.path
        if os.path.exists(backbone):
            net.init_weights(backbone_path=backbone)
        else:
            net.init_head_weights()
        print(cfg.backbone.name)
    net.train(mode=False)
    net = net.to(cfg.device)
    if args.ibims1 is not None or args.ibims1_pd is not None:
        raise SystemExit("iBims-1 exporters are not part of this build (evaluation tooling outside the hot path).")
    if args.image is not None:
        inp, out = args.image.split(":") if ":" in args.image else (args.image, None)
        print("Inference image: {}".format(inp))
        inference_image(net, inp, out, depth_mode=args.depth_mode)
    if args.images is not None:
        inp, out = args.images.split(":")
        inference_images(net, inp, out, max_img=args.max_img, depth_mode=args.depth_mode)


if __name__ == "__main__":
    main()

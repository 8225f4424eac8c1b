"""Geometry / init helpers on the hot path (reference: models/functions/funcs.py:195-224,329-332 and the
`imrescale` call of losses.py:243-247).  cv2 is not a dependency: the only cv2 call on the path is a uint8
bilinear resize at exactly 1/4 scale, restated in closed form below."""
import math

import numpy as np
import torch


def bias_init_with_prob(prior_prob):
    return float(-math.log((1 - prior_prob) / prior_prob))


def calc_size_preserve_ar(img_w, img_h, max_size):
    if img_w > img_h:
        return int(max_size), int(img_h / img_w * max_size)
    return int(img_w / img_h * max_size), int(max_size)


def pad_even_divided(img, divisor=32):
    """Zero-pad an HxWxC array (numpy) on the bottom/right so both sides divide `divisor`."""
    h, w, c = img.shape
    out = np.zeros((h + (-h) % divisor, w + (-w) % divisor, c))
    out[:h, :w] = img
    return out


def center_of_mass(bitmasks):
    _, h, w = bitmasks.size()
    ys = torch.arange(0, h, dtype=torch.float32, device=bitmasks.device)
    xs = torch.arange(0, w, dtype=torch.float32, device=bitmasks.device)
    m00 = bitmasks.sum(dim=-1).sum(dim=-1).clamp(min=1e-6)
    return (bitmasks * xs).sum(dim=-1).sum(dim=-1) / m00, (bitmasks * ys[:, None]).sum(dim=-1).sum(dim=-1) / m00


def quarter_mask_u8(masks):
    """uint8 [N,H,W] -> uint8 [N,H/4,W/4]: OpenCV INTER_LINEAR at exact 1/4 samples the centre of each 4x4 block,
    i.e. the mean of its 2x2 middle pixels, rounded half up in fixed point."""
    if masks.shape[-2] % 4 or masks.shape[-1] % 4:
        # (the reference's imrescale handles any size through cv2's general sampling, output int(h * 0.25 + 0.5); the hot path
        # only ever sees 480x640 / sizes padded to a multiple of 32)
        raise ValueError("quarter_mask_u8: mask size %dx%d is not a multiple of 4 (the closed form of the exact-1/4 resize needs it)"
                         % (masks.shape[-2], masks.shape[-1]))
    a = masks.to(torch.int32)
    s = a[:, 1::4, 1::4] + a[:, 1::4, 2::4] + a[:, 2::4, 1::4] + a[:, 2::4, 2::4]
    return ((s + 2) >> 2).to(torch.uint8)


class FastBaseTransform(torch.nn.Module):
    """[n,h,w,3] BGR 0..255 -> [n,3,h,w] RGB normalised (reference data/augmentations.py:496-530)."""

    def __init__(self):
        super().__init__()
        from .config import MEANS, STD, cfg
        self.register_buffer("mean", torch.tensor(MEANS, dtype=torch.float32)[None, :, None, None], persistent=False)
        self.register_buffer("std", torch.tensor(STD, dtype=torch.float32)[None, :, None, None], persistent=False)
        self.transform = cfg.backbone.transform

    def forward(self, img):
        img = img.permute(0, 3, 1, 2).contiguous()
        mean, std = self.mean.to(img.device), self.std.to(img.device)
        if self.transform.normalize:
            img = (img - mean) / std
        elif self.transform.subtract_means:
            img = img - mean
        elif self.transform.to_float:
            img = img / 255.0
        if self.transform.channel_order != "RGB":
            raise NotImplementedError
        return img[:, (2, 1, 0), :, :].contiguous()


def frame_to_input(frame_u8, size_wh, divisor=32, want_frame=True):
    """Input staging of simple_inference.py:143-152 as ONE HIP launch (include/prn.h: prn_frame_to_input): frame_u8 = the decoded
    uint8 BGR frame [H,W,3] ON THE DEVICE (upload the bytes, not floats); cv2-style INTER_LINEAR resize to size_wh = (w, h),
    zero padding to a multiple of `divisor`, FastBaseTransform.  -> (batch [1,3,Hp,Wp] float RGB, frame [Hp,Wp,3] float BGR | None)."""
    import ctypes
    from . import ops
    from .config import MEANS, STD, cfg
    if not frame_u8.is_cuda or frame_u8.dtype != torch.uint8 or frame_u8.dim() != 3 or frame_u8.shape[2] != 3:
        raise RuntimeError("frame_to_input needs a uint8 [H,W,3] device tensor (got %s %s %s)" % (frame_u8.device, frame_u8.dtype, tuple(frame_u8.shape)))
    frame_u8 = frame_u8.contiguous()
    Hs, Ws = frame_u8.shape[:2]
    Wr, Hr = int(size_wh[0]), int(size_wh[1])
    Hp, Wp = Hr + (-Hr) % divisor, Wr + (-Wr) % divisor
    tr = cfg.backbone.transform
    # FastBaseTransform (augmentations.py:518-530): normalize, else subtract_means, else to_float (x / 255), else the values as they are
    mode = 0 if tr.normalize else (1 if tr.subtract_means else (2 if tr.to_float else 3))
    if tr.channel_order != "RGB":
        raise NotImplementedError
    out = torch.empty(1, 3, Hp, Wp, device=frame_u8.device, dtype=torch.float32)
    frame = torch.empty(Hp, Wp, 3, device=frame_u8.device, dtype=torch.float32) if want_frame else None
    mean = (ctypes.c_float * 3)(*[float(v) for v in MEANS])
    std = (ctypes.c_float * 3)(*[float(v) for v in STD])
    ops.check(ops.lib.prn_frame_to_input(ops._p(frame_u8), Hs, Ws, Hr, Wr, Hp, Wp, mean, std, mode, ops._p(out), ops._p(frame), ops._stream()),
              "prn_frame_to_input")
    return out, frame

"""TEST INFRASTRUCTURE -- numpy restatement of the inference entry point's input staging:
cv2.resize(INTER_LINEAR) on a uint8 frame -> zero padding to a multiple of 32 -> FastBaseTransform
(reference simple_inference.py:143-152, models/functions/funcs.py:195-210, data/augmentations.py:496-530).

PARITY UNPINNED for the resize: OpenCV is a third-party dependency absent from this image and from /root/reference.
The restatement follows OpenCV's published resize.cpp for 8-bit INTER_LINEAR: coefficients (1 - f, f) x 2048 rounded to
short (INTER_RESIZE_COEF_BITS = 11), source coordinate f = (d + 0.5) * scale - 0.5 in float with the border clamps of
`resizeGeneric_`, horizontal pass in int, vertical pass `(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`.
It is pinned by its own consistency with the closed form the reference path relies on elsewhere (exact 1/4 scale = rounded
mean of the 2x2 centre pixels, funcs.quarter_mask_u8 / ref_shim's cv2 stub) and by the identity case (tests/test_frame.py).
Only tests/ may import this module.
"""
import numpy as np


def _coef(n_dst, n_src):
    scale = n_src / n_dst                                   # double, like cv2's inv_scale
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo, hi = s < 0, s >= n_src - 1
    f = np.where(lo | hi, np.float32(0), f)
    s = np.where(lo, 0, np.where(hi, n_src - 1, s))
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, np.minimum(s + 1, n_src - 1), a0, a1


def resize_linear_u8(img, size_wh):
    """uint8 [H,W,C] -> uint8 [h,w,C], cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR)."""
    w, h = size_wh
    H, W = img.shape[:2]
    sx, sx1, ax0, ax1 = _coef(w, W)
    sy, sy1, by0, by1 = _coef(h, H)
    a = img.astype(np.int64)
    rows0 = a[sy][:, sx] * ax0[None, :, None] + a[sy][:, sx1] * ax1[None, :, None]
    rows1 = a[sy1][:, sx] * ax0[None, :, None] + a[sy1][:, sx1] * ax1[None, :, None]
    v = (((by0[:, None, None] * (rows0 >> 4)) >> 16) + ((by1[:, None, None] * (rows1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def frame_to_input(img_bgr_u8, size_wh, means_bgr, std_bgr, divisor=32):
    """-> (input float32 [1,3,Hp,Wp] RGB normalised, padded frame float64 [Hp,Wp,3] BGR)."""
    r = resize_linear_u8(img_bgr_u8, size_wh)
    h, w, c = r.shape
    pad = np.zeros((h + (-h) % divisor, w + (-w) % divisor, c))
    pad[:h, :w] = r
    x = (pad.astype(np.float32) - np.asarray(means_bgr, np.float32)) / np.asarray(std_bgr, np.float32)
    return np.ascontiguousarray(x[:, :, ::-1].transpose(2, 0, 1))[None], pad

"""TEST INFRASTRUCTURE -- container-only importer for the real reference.

Imports /root/reference (pure Python, PyTorch) on CPU under four in-process shims so that
tests/golden/make_golden.py can (a) check the oracle restatement (oracle/*.py) against the real
reference and (b) emit golden vectors.  Nothing here is importable on the GPU box (the reference
does not travel); only tests/golden/make_golden.py uses it.

Shims:
  1. `cv2` stub: interpolation constants + `resize` restricted to the one call the hot path makes
     (losses.py:243-247 -> funcs.py:173-193: uint8 HxWxC, exact 1/4 scale, INTER_LINEAR). At exact
     1/4 the sample point of dst pixel i is 4i+1.5, i.e. the mean of the 2x2 centre pixels of each
     4x4 block; OpenCV's fixed-point path rounds half up.
  2. `torchvision.ops.deform_conv2d` stub -> oracle.dcn_ref.deform_conv2d_ref (torchvision is not
     installed here; "parity unpinned" at this boundary, see oracle/dcn_ref.py).
  3. torch.cuda.current_device -> 0   (planerecnet.py:18)
  4. torch.Tensor.cuda -> identity    (vnl.py:12-31, losses.py:313,319)
"""
import sys
import types

import numpy as np
import torch

REF_ROOT = "/root/reference"


def _cv2_resize(img, size, dst=None, interpolation=1):
    w_new, h_new = size
    h, w = img.shape[:2]
    if (h_new, w_new) == (h, w):
        return img.copy()
    assert h == 4 * h_new and w == 4 * w_new and img.dtype == np.uint8, \
        "cv2 stub only restates the exact-1/4 uint8 bilinear case"
    a = img.astype(np.int32)
    s = a[1::4, 1::4] + a[1::4, 2::4] + a[2::4, 1::4] + a[2::4, 2::4]
    return ((s + 2) >> 2).astype(np.uint8)


def install():
    """Install the shims and put the reference on sys.path. Idempotent."""
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.INTER_CUBIC, cv2.INTER_AREA, cv2.INTER_LANCZOS4 = 0, 1, 2, 3, 4
        cv2.resize = _cv2_resize
        sys.modules["cv2"] = cv2
    if "torchvision" not in sys.modules:
        from oracle.dcn_ref import deform_conv2d_ref
        tv = types.ModuleType("torchvision")
        ops = types.ModuleType("torchvision.ops")

        def deform_conv2d(input, offset, weight, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), mask=None):
            s = stride if isinstance(stride, int) else stride[0]
            p = padding if isinstance(padding, int) else padding[0]
            return deform_conv2d_ref(input, offset, mask, weight, bias, s, p)

        ops.deform_conv2d = deform_conv2d
        tv.ops = ops
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.ops"] = ops
    torch.cuda.current_device = lambda: 0
    torch.Tensor.cuda = lambda self, *a, **k: self
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def load_reference(config_name="PlaneRecNet_50_config"):
    """Returns (ref_modules dict) with cfg set to `config_name` on CPU."""
    install()
    import data.config as rcfg
    from utils import timer
    rcfg.set_cfg(config_name)
    rcfg.cfg.device = "cpu"
    timer.disable_all()
    import planerecnet as rprn
    import models.functions.losses as rloss
    return {"config": rcfg, "planerecnet": rprn, "losses": rloss, "timer": timer}

"""Evaluation metrics on the device (reference: eval.py:164-207 `compute_depth_metrics`).

Same signature and return tuple as the reference function: (abs_rel, sq_rel, rmse, log10, a1, a2, a3, ratio) as CPU
tensors.  The seven error terms come from ONE fused HIP reduction (include/prn.h: prn_depth_metrics); the median ratio is
computed as in the reference (median of ALL ground-truth depths over the median of the valid predictions, eval.py:184 --
reported only, the prediction is not rescaled).
"""
import ctypes

import torch

from . import _lib
from ._lib import check, lib
from .config import cfg


def compute_depth_metrics(pred_depth, gt_depth, median_scaling=True):
    """pred_depth, gt_depth: [1, H, W] dense depth maps in the same unit (device tensors)."""
    if not (pred_depth.is_cuda and gt_depth.is_cuda):
        raise RuntimeError("compute_depth_metrics needs device tensors; there is no CPU path in the product")
    pred = pred_depth.detach().float().contiguous()
    gt = gt_depth.detach().float().contiguous()
    if pred.numel() != gt.numel():
        raise RuntimeError("compute_depth_metrics: prediction %s and ground truth %s differ in size" % (tuple(pred.shape), tuple(gt.shape)))
    out = torch.empty(8, device=pred.device, dtype=torch.float64)
    ws = torch.empty(lib.prn_depth_metrics_ws_doubles(), device=pred.device, dtype=torch.float64)
    stream = ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(pred.device.index))
    check(lib.prn_depth_metrics(ctypes.c_void_p(pred.data_ptr()), ctypes.c_void_p(gt.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                ctypes.c_void_p(ws.data_ptr()), pred.numel(), float(cfg.dataset.min_depth), float(cfg.dataset.max_depth), stream),
          "prn_depth_metrics")
    if median_scaling:
        valid = (gt.flatten() > 0.5).logical_and(pred.flatten() > 0.5)
        ratio = (torch.median(gt) / torch.median(pred.flatten()[valid])).cpu()
    else:
        ratio = torch.tensor(0)
    o = out.cpu()
    return o[0], o[1], o[2], o[3], o[4], o[5], o[6], ratio

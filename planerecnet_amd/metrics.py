"""Evaluation metrics (reference eval.py:164-354): `compute_depth_metrics` and the detection metrics
(`compute_segmentation_metrics`, `APDataObject`, `calc_map`), the latter further down.

Depth errors:

Same signature and return tuple as the reference function: (abs_rel, sq_rel, rmse, log10, a1, a2, a3, ratio) as CPU
tensors.  The seven error terms come from ONE fused HIP reduction (include/prn.h: prn_depth_metrics); the median ratio is
computed as in the reference (median of ALL ground-truth depths over the median of the valid predictions, eval.py:184 --
reported only, the prediction is not rescaled).
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch

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


# ---------------------------------------------------------------------------------------------------------------------------
# Detection metrics (reference eval.py:210-354).  The pairwise IoU matrices of a frame come from the device
# (include/prn.h: prn_pairwise_iou -- bit-packed masks, AND + popcount); the per-threshold matching and the AP integral are a
# few hundred scalar operations per frame and stay on the host, vectorised, with the reference's exact outcomes (including
# its quirks, see compute_segmentation_metrics).
iou_thresholds = [x / 100 for x in range(50, 100, 5)]                                   # eval.py:61
depth_metrics = ["abs_rel", "sq_rel", "rmse", "log10", "a1", "a2", "a3", "ratio"]      # eval.py:60


def pairwise_iou(masks_a=None, masks_b=None, boxes_a=None, boxes_b=None):
    """-> (mask_iou [A,B] | None, box_iou [A,B] | None), fp32 device tensors.  masks: [A,H,W] / [B,H,W], any dtype, non-zero = set
    (the reference multiplies their 0/1 float copies); boxes: [A,4] / [B,4] (x1,y1,x2,y2), converted to fp32 like eval.py:215."""
    ref = masks_a if masks_a is not None else boxes_a
    if ref is None or not ref.is_cuda:
        raise RuntimeError("pairwise_iou needs device tensors; there is no CPU path in the product")
    dev = ref.device
    A = ref.shape[0]
    B = (masks_b if masks_a is not None else boxes_b).shape[0]
    want_m, want_b = masks_a is not None, boxes_a is not None
    if A == 0 or B == 0:
        e = torch.empty(A, B, device=dev)
        return (e if want_m else None), (e.clone() if want_b else None)
    HW, ma, mb, miou, ws = 1, None, None, None, None
    if want_m:
        if masks_a.shape[1:] != masks_b.shape[1:]:
            raise RuntimeError("pairwise_iou: mask sets of different size %s / %s" % (tuple(masks_a.shape), tuple(masks_b.shape)))
        HW = int(masks_a[0].numel())
        ma = (masks_a if masks_a.dtype in (torch.uint8, torch.bool) else masks_a != 0).contiguous().view(torch.uint8)
        mb = (masks_b if masks_b.dtype in (torch.uint8, torch.bool) else masks_b != 0).to(dev).contiguous().view(torch.uint8)
        miou = torch.empty(A, B, device=dev)
        ws = torch.empty(lib.prn_pairwise_iou_ws_bytes(A, B, HW), device=dev, dtype=torch.uint8)
    ba = bb = biou = None
    if want_b:
        ba, bb = boxes_a.to(dev).float().contiguous(), boxes_b.to(dev).float().contiguous()
        biou = torch.empty(A, B, device=dev)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else None)      # noqa: E731
    stream = ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(dev.index))
    check(lib.prn_pairwise_iou(ptr(ma), ptr(mb), ptr(ba), ptr(bb), A, B, HW, ptr(miou), ptr(biou), ptr(ws), stream), "prn_pairwise_iou")
    return miou, biou


def category_scores(cate_logits):
    """list of [B, C, S, S] category logits (one per grid level) -> [B, cells, C] scores after sigmoid + 2x2 point NMS, levels concatenated
    (reference planerecnet.py:113 + models/functions/nms.py:8-12 + the level concatenation of `inference`), one launch per level."""
    B, C = cate_logits[0].shape[:2]
    cells = sum(int(c.shape[2]) * int(c.shape[3]) for c in cate_logits)
    out = torch.empty(B, cells, C, device=cate_logits[0].device, dtype=torch.float32)
    stream = ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(out.device.index))
    off = 0
    for c in cate_logits:
        S = int(c.shape[2])
        if not c.is_cuda or c.dtype != torch.float32 or c.shape[3] != S or c.shape[:2] != (B, C):
            raise RuntimeError("category_scores needs square fp32 device maps of one batch / class count")
        c = c.detach().contiguous()
        check(lib.prn_sigmoid_point_nms(ctypes.c_void_p(c.data_ptr()), ctypes.c_void_p(out.data_ptr() + 4 * off * C), B, C, S, cells * C, stream), "prn_sigmoid_point_nms")
        off += S * S
    return out


def mask_stats(seg, thr):
    """[n,h,w] soft masks -> (count [n], msum [n]) of the values above thr, fp32 (include/prn.h: prn_mask_stats)."""
    if not seg.is_cuda or seg.dtype != torch.float32:
        raise RuntimeError("mask_stats needs fp32 device tensors")
    n = seg.shape[0]
    out = torch.empty(2, n, device=seg.device, dtype=torch.float32)
    if n:
        seg = seg.contiguous()
        stream = ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(seg.device.index))
        check(lib.prn_mask_stats(ctypes.c_void_p(seg.data_ptr()), n, int(seg[0].numel()), float(thr), ctypes.c_void_p(out[0].data_ptr()),
                                 ctypes.c_void_p(out[1].data_ptr()), stream), "prn_mask_stats")
    return out[0], out[1]


def mask_boxes(masks):
    """[n,H,W] binary masks (bool / uint8, on the device) -> [n,4] float (x0, y0, x1, y1) of their set pixels, one launch
    (reference planerecnet.py:282-287; an empty mask gives (H+W, H+W, -1, -1) like the vectorised torch form)."""
    if not masks.is_cuda or masks.dtype not in (torch.bool, torch.uint8):
        raise RuntimeError("mask_boxes needs bool / uint8 device masks")
    n, H, W = masks.shape
    out = torch.empty(n, 4, device=masks.device, dtype=torch.float32)
    if n:
        m = masks.contiguous().view(torch.uint8)
        stream = ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(masks.device.index))
        check(lib.prn_mask_boxes(ctypes.c_void_p(m.data_ptr()), n, H, W, ctypes.c_void_p(out.data_ptr()), stream), "prn_mask_boxes")
    return out


def matrix_nms_scores(iou, labels, scores, sigma, gaussian):
    """Matrix-NMS decayed scores from the [n, n] mask-IoU matrix of detections in descending score order (include/prn.h: prn_matrix_nms)."""
    n = scores.shape[0]
    iou, labels, scores = iou.contiguous(), labels.contiguous().long(), scores.contiguous().float()
    out = torch.empty(2, n, device=scores.device, dtype=torch.float32)          # decayed scores | workspace (per-column denominators)
    stream = ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(scores.device.index))
    check(lib.prn_matrix_nms(ctypes.c_void_p(iou.data_ptr()), ctypes.c_void_p(labels.data_ptr()), ctypes.c_void_p(scores.data_ptr()), n, float(sigma),
                             int(bool(gaussian)), ctypes.c_void_p(out[0].data_ptr()), ctypes.c_void_p(out[1].data_ptr()), stream), "prn_matrix_nms")
    return out[0]


def mask_iou(masks_a, masks_b, iscrowd=False):
    """[a,h,w] x [b,h,w] -> [a,b]  (reference models/functions/funcs.py:58-71)."""
    if iscrowd:
        raise NotImplementedError("mask_iou(iscrowd=True) has no call site in the reference and is not built")
    return pairwise_iou(masks_a=masks_a, masks_b=masks_b)[0]


def bbox_iou(box_a, box_b):
    """[A,4] x [B,4] -> [A,B]  (reference models/functions/funcs.py:30-56, the un-batched form eval.py uses)."""
    if box_a.dim() != 2:
        raise NotImplementedError("bbox_iou: the batched [n,A,4] form has no call site in the reference and is not built")
    return pairwise_iou(boxes_a=box_a, boxes_b=box_b)[1]


class APDataObject:
    """Scores and hit flags of the detections seen for one IoU threshold and one class (reference eval.py:254-325)."""

    def __init__(self):
        self.scores, self.hits = [], []
        self.num_gt_positives = 0

    def push(self, score, is_true):
        self.scores.append(float(score))
        self.hits.append(bool(is_true))

    def extend(self, scores, hits):
        self.scores.extend(scores)
        self.hits.extend(hits)

    def add_gt_positives(self, num_positives):
        self.num_gt_positives += num_positives

    def is_empty(self):
        return len(self.scores) == 0 and self.num_gt_positives == 0

    def get_ap(self):
        """101-point interpolated AP: precision envelope sampled at recall 0, 0.01, ..., 1 (eval.py:272-325)."""
        if self.num_gt_positives == 0:
            return 0
        n = len(self.scores)
        if n == 0:
            return 0.0
        order = np.argsort(-np.asarray(self.scores, np.float64), kind="stable")       # equal scores keep their push order
        hit = np.asarray(self.hits, bool)[order]
        tp, fp = np.cumsum(hit), np.cumsum(~hit)
        prec = tp / (tp + fp)
        rec = tp / self.num_gt_positives
        prec = np.maximum.accumulate(prec[::-1])[::-1]
        idx = np.searchsorted(rec, np.array([x / 100 for x in range(101)]), side="left")
        bars = np.where(idx < n, prec[np.minimum(idx, n - 1)], 0.0)
        return sum(bars.tolist()) / 101                                               # left-to-right like the reference's sum()


def new_ap_data():
    return {"box": [APDataObject() for _ in iou_thresholds], "mask": [APDataObject() for _ in iou_thresholds]}     # eval.py:77-80


def match_frame(ap_data, mask_iou_np, box_iou_np, scores, num_gt_for_class):
    """Host half of compute_segmentation_metrics: push one frame's detections into `ap_data` given its [A,B] IoU matrices."""
    scores = np.asarray(scores, np.float64)
    order = np.argsort(-scores, kind="stable")                                        # eval.py:217
    s = scores[order]
    for t, thr in enumerate(iou_thresholds):
        for kind, iou in (("box", box_iou_np), ("mask", mask_iou_np)):
            obj = ap_data[kind][t]
            obj.add_gt_positives(num_gt_for_class)
            matched = (iou[order] > thr).any(axis=1) if iou.shape[1] else np.zeros(len(order), bool)    # NaN (empty masks) never matches
            # every detection is pushed as a miss, a matched one as a hit before it (eval.py:248-252)
            reps = 1 + matched.astype(np.int64)
            flags = np.zeros(int(reps.sum()), bool)
            flags[(np.cumsum(reps) - reps)[matched]] = True
            obj.extend(np.repeat(s, reps).tolist(), flags.tolist())


def compute_segmentation_metrics(ap_data, gt_masks, gt_boxes, gt_classes, pred_masks, pred_boxes, pred_classes, pred_scores):
    """One frame's detections against its ground truth, for every IoU threshold, boxes and masks (reference eval.py:210-252;
    same argument list).  Two reference behaviours are kept on purpose: a matched detection counts as a hit AND as a miss, and
    one ground-truth instance may be matched by several detections."""
    miou, biou = pairwise_iou(pred_masks, gt_masks.to(pred_masks.device), pred_boxes, gt_boxes)
    num_gt_for_class = int((torch.as_tensor(gt_classes) == 0).sum())                  # eval.py:234
    match_frame(ap_data, miou.cpu().numpy(), biou.cpu().numpy(), torch.as_tensor(pred_scores).detach().cpu().numpy(), num_gt_for_class)


def calc_map(ap_data, quiet=False):
    """-> {'box': {'all', 50, 55, ..., 95}, 'mask': {...}} in percent, rounded to two decimals (reference eval.py:327-354)."""
    if not quiet:
        print("Calculating mAP...")
    all_maps = {}
    for kind in ("box", "mask"):
        row = OrderedDict([("all", 0)])
        for t, thr in enumerate(iou_thresholds):
            obj = ap_data[kind][t]
            row[int(thr * 100)] = obj.get_ap() * 100 if not obj.is_empty() else 0
        row["all"] = sum(row.values()) / (len(row) - 1)
        all_maps[kind] = row
    if not quiet:
        print_maps(all_maps)
    return {k: {j: round(u, 2) for j, u in v.items()} for k, v in all_maps.items()}


def print_maps(all_maps):
    """The reference's table layout (eval.py:356-370)."""
    keys = list(all_maps["box"].keys())
    cell = " %5s |"
    rule = "-------+" * (len(keys) + 1)
    print()
    print(cell * (len(keys) + 1) % tuple([""] + [(".%d " % k) if isinstance(k, int) else k + " " for k in keys]))
    print(rule)
    for kind in ("box", "mask"):
        print(cell * (len(keys) + 1) % tuple([kind] + [("%.2f" % v) if v < 100 else ("%.1f" % v) for v in all_maps[kind].values()]))
    print(rule)
    print()

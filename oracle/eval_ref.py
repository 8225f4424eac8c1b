"""TEST INFRASTRUCTURE -- CPU restatement of the reference's detection metrics (eval.py:210-354 with the two IoU helpers of
models/functions/funcs.py:9-71): pairwise IoU, the per-threshold matching of one frame, the AP integral and the mAP table.

Pinned against the reference's own functions by tests/golden/make_golden_eval.py (fixture tests/golden/eval_metrics.npz,
checked on any machine by tests/test_eval_metrics.py).  Only tests/ may import this module.

Scalar loops throughout -- it is the checker, sized for a few dozen detections.
"""
import numpy as np
import torch

IOU_THRESHOLDS = [x / 100 for x in range(50, 100, 5)]       # eval.py:61


def mask_iou_ref(masks_a, masks_b):
    """[a,h,w] x [b,h,w] -> [a,b]  (funcs.py:58-71, iscrowd=False): float matmul of the flattened masks over the union."""
    fa = masks_a.reshape(masks_a.shape[0], -1).float()
    fb = masks_b.reshape(masks_b.shape[0], -1).float()
    inter = fa @ fb.t()
    return inter / (fa.sum(1)[:, None] + fb.sum(1)[None, :] - inter)


def bbox_iou_ref(box_a, box_b):
    """[A,4] x [B,4] (x1,y1,x2,y2) -> [A,B]  (funcs.py:9-55)."""
    A, B = box_a.shape[0], box_b.shape[0]
    out = torch.empty(A, B, dtype=box_a.dtype)
    for i in range(A):
        for j in range(B):
            w = torch.clamp(torch.min(box_a[i, 2], box_b[j, 2]) - torch.max(box_a[i, 0], box_b[j, 0]), min=0)
            h = torch.clamp(torch.min(box_a[i, 3], box_b[j, 3]) - torch.max(box_a[i, 1], box_b[j, 1]), min=0)
            inter = w * h
            area_a = (box_a[i, 2] - box_a[i, 0]) * (box_a[i, 3] - box_a[i, 1])
            area_b = (box_b[j, 2] - box_b[j, 0]) * (box_b[j, 3] - box_b[j, 1])
            out[i, j] = inter / (area_a + area_b - inter)
    return out


class APDataRef:
    """Scores + hit flags of every detection pushed for one IoU threshold (eval.py:254-325)."""

    def __init__(self):
        self.points = []
        self.gt_total = 0

    def empty(self):
        return not self.points and self.gt_total == 0

    def ap(self):
        if self.gt_total == 0:
            return 0
        pts = sorted(self.points, key=lambda sp: -sp[0])        # stable: equal scores keep their push order (eval.py:281)
        prec, rec, hit, miss = [], [], 0, 0
        for _, good in pts:
            hit, miss = hit + bool(good), miss + (not good)
            prec.append(hit / (hit + miss))
            rec.append(hit / self.gt_total)
        for i in range(len(prec) - 1, 0, -1):                   # monotone envelope from the right (eval.py:303-305)
            prec[i - 1] = max(prec[i - 1], prec[i])
        total = 0.0
        for k in range(101):                                    # 101-bar Riemann sum over recall (eval.py:309-324)
            j = int(np.searchsorted(np.array(rec), k / 100, side="left"))
            total += prec[j] if j < len(prec) else 0
        return total / 101


def new_ap_data():
    return {"box": [APDataRef() for _ in IOU_THRESHOLDS], "mask": [APDataRef() for _ in IOU_THRESHOLDS]}      # eval.py:77-80


def segmentation_metrics_ref(ap_data, gt_masks, gt_boxes, gt_classes, pred_masks, pred_boxes, pred_classes, pred_scores):
    """One frame (eval.py:210-252).  Quirks kept: a matched detection is pushed as a hit AND, like every detection, as a
    miss (:249-252: no `else`); a ground truth may be matched by several detections (`gt_used` is written, never read)."""
    miou = mask_iou_ref(pred_masks, gt_masks)
    biou = bbox_iou_ref(pred_boxes.float(), gt_boxes.float())
    order = sorted(range(len(pred_classes)), key=lambda i: -float(pred_scores[i]))
    n_gt = sum(1 for c in gt_classes if c == 0)                 # :234
    for t, thr in enumerate(IOU_THRESHOLDS):
        for kind, iou in (("box", biou), ("mask", miou)):
            obj = ap_data[kind][t]
            obj.gt_total += n_gt
            for i in order:
                best, match = thr, -1
                for j in range(len(gt_classes)):
                    v = iou[i, j].item()
                    if v > best:                                 # strict: first ground truth reaching the maximum wins
                        best, match = v, j
                if match >= 0:
                    obj.points.append((float(pred_scores[i]), True))
                obj.points.append((float(pred_scores[i]), False))


def calc_map_ref(ap_data):
    """-> {'box': {'all': .., 50: .., ..., 95: ..}, 'mask': {...}} in percent, unrounded (eval.py:327-349)."""
    table = {}
    for kind in ("box", "mask"):
        row = {}
        for t, thr in enumerate(IOU_THRESHOLDS):
            obj = ap_data[kind][t]
            row[int(thr * 100)] = 0 if obj.empty() else obj.ap() * 100
        table[kind] = {"all": sum(row.values()) / len(row), **row}
    return table

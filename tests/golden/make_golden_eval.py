"""Golden vectors for the detection metrics -- runs ONLY in the build container (needs /root/reference).

    python tests/golden/make_golden_eval.py

Feeds seeded frames (`make_frame`, duplicated in tests/test_eval_metrics.py) to the reference's own
`compute_segmentation_metrics` / `APDataObject` / `calc_map` (eval.py:210-354) and `mask_iou` / `bbox_iou`
(models/functions/funcs.py), asserts that the oracle restatement (oracle/eval_ref.py) reproduces the IoU matrices bit for
bit and the mAP table to 1e-12, and writes IoU matrices + table to tests/golden/eval_metrics.npz."""
import ast
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
from oracle import eval_ref, ref_shim  # noqa: E402

SEEDS = (0, 1, 2, 3, 4)


def make_frame(seed, H=60, W=80):
    """Ground-truth rectangles, detections that are jittered / eroded copies of some of them plus false positives, scores
    rounded to one decimal so that ties occur.  Frame 3 has an empty detection mask (0/0 -> NaN row in the reference)."""
    rng = np.random.RandomState(1000 + seed)
    n_gt = int(rng.randint(3, 7))
    gt_masks, gt_boxes = np.zeros((n_gt, H, W), np.uint8), np.zeros((n_gt, 4), np.float64)
    for i in range(n_gt):
        w, h = int(rng.randint(8, W // 2)), int(rng.randint(8, H // 2))
        x0, y0 = int(rng.randint(0, W - w)), int(rng.randint(0, H - h))
        gt_masks[i, y0:y0 + h, x0:x0 + w] = 1
        gt_boxes[i] = (x0, y0, x0 + w, y0 + h)
    pm, pb = [], []
    for i in range(n_gt):
        for _ in range(int(rng.randint(0, 3))):
            x0, y0, x1, y1 = gt_boxes[i] + rng.randint(-4, 5, size=4)
            x0, y0, x1, y1 = int(max(x0, 0)), int(max(y0, 0)), int(min(max(x1, x0 + 2), W)), int(min(max(y1, y0 + 2), H))
            m = np.zeros((H, W), bool)
            m[y0:y1, x0:x1] = True
            m &= rng.rand(H, W) > 0.05
            pm.append(m)
            pb.append((x0 + rng.rand(), y0 + rng.rand(), x1 - rng.rand(), y1 - rng.rand()))
    for _ in range(int(rng.randint(1, 4))):                      # false positives
        w, h = int(rng.randint(4, 20)), int(rng.randint(4, 20))
        x0, y0 = int(rng.randint(0, W - w)), int(rng.randint(0, H - h))
        m = np.zeros((H, W), bool)
        m[y0:y0 + h, x0:x0 + w] = True
        pm.append(m)
        pb.append((x0, y0, x0 + w, y0 + h))
    if seed == 3:
        pm[0][:] = False
    n = len(pm)
    scores = np.round(0.15 + 0.85 * rng.rand(n), 1).astype(np.float32)
    return {"gt_masks": torch.from_numpy(gt_masks), "gt_boxes": torch.from_numpy(gt_boxes), "gt_classes": torch.zeros(n_gt, dtype=torch.int64),
            "pred_masks": torch.from_numpy(np.stack(pm)), "pred_boxes": torch.tensor(pb, dtype=torch.float32),
            "pred_classes": torch.zeros(n, dtype=torch.int64), "pred_scores": torch.from_numpy(scores)}


def reference_namespace():
    """The reference's metric code compiled from eval.py where it lies (the module itself cannot be imported here: cv2 /
    tensorboardX / pycocotools at import time); nothing of it is stored in this repository."""
    ref_shim.install()
    sys.path.insert(0, ref_shim.REF_ROOT)
    from models.functions.funcs import bbox_iou, mask_iou
    src = open(os.path.join(ref_shim.REF_ROOT, "eval.py")).read()
    want = {"compute_segmentation_metrics", "APDataObject", "calc_map", "print_maps"}
    body = [n for n in ast.parse(src).body
            if (isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in want)
            or (isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "iou_thresholds")]
    ns = {"torch": torch, "np": np, "OrderedDict": OrderedDict, "mask_iou": mask_iou, "bbox_iou": bbox_iou}
    exec(compile(ast.Module(body=body, type_ignores=[]), os.path.join(ref_shim.REF_ROOT, "eval.py"), "exec"), ns)
    return ns


def main():
    ns = reference_namespace()
    assert ns["iou_thresholds"] == eval_ref.IOU_THRESHOLDS
    ref_data = {k: [ns["APDataObject"]() for _ in ns["iou_thresholds"]] for k in ("box", "mask")}
    ora_data = eval_ref.new_ap_data()
    fix = {}
    for seed in SEEDS:
        f = make_frame(seed)
        pm, gm = f["pred_masks"].float(), f["gt_masks"].float()                      # eval.py:99-100
        ns["compute_segmentation_metrics"](ref_data, gm, f["gt_boxes"], f["gt_classes"], pm, f["pred_boxes"], f["pred_classes"], f["pred_scores"])
        eval_ref.segmentation_metrics_ref(ora_data, gm, f["gt_boxes"], f["gt_classes"], pm, f["pred_boxes"], f["pred_classes"], f["pred_scores"])
        miou = ns["mask_iou"](pm, gm)
        biou = ns["bbox_iou"](f["pred_boxes"].float(), f["gt_boxes"].float())
        assert np.array_equal(miou.numpy(), eval_ref.mask_iou_ref(pm, gm).numpy(), equal_nan=True)
        assert np.array_equal(biou.numpy(), eval_ref.bbox_iou_ref(f["pred_boxes"].float(), f["gt_boxes"].float()).numpy(), equal_nan=True)
        fix["mask_iou_%d" % seed], fix["box_iou_%d" % seed] = miou.numpy(), biou.numpy()
    for kind in ("box", "mask"):
        for t in range(len(eval_ref.IOU_THRESHOLDS)):
            r, o = ref_data[kind][t], ora_data[kind][t]
            assert r.num_gt_positives == o.gt_total and r.data_points == o.points
    aps = {k: np.array([ref_data[k][t].get_ap() for t in range(10)]) for k in ("box", "mask")}
    table = ns["calc_map"](ref_data)                                                  # rounded to 2 decimals (eval.py:352-353)
    ora = eval_ref.calc_map_ref(ora_data)
    for kind in ("box", "mask"):
        assert np.allclose(aps[kind] * 100, [ora[kind][int(t * 100)] for t in eval_ref.IOU_THRESHOLDS], rtol=0, atol=1e-12)
        assert all(abs(table[kind][k] - round(ora[kind][k], 2)) < 1e-9 for k in table[kind]), (table[kind], ora[kind])
        fix["ap_" + kind] = aps[kind]
        fix["map_rounded_" + kind] = np.array([table[kind][k] for k in ["all"] + [int(t * 100) for t in eval_ref.IOU_THRESHOLDS]])
        fix["points_" + kind] = np.array([len(ref_data[kind][t].data_points) for t in range(10)])
    fix["gt_positives"] = np.array(ref_data["box"][0].num_gt_positives)
    np.savez(os.path.join(HERE, "eval_metrics.npz"), **fix)
    print("box ", fix["map_rounded_box"])
    print("mask", fix["map_rounded_mask"])
    print("written", os.path.join(HERE, "eval_metrics.npz"))


if __name__ == "__main__":
    main()

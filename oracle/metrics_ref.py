"""TEST INFRASTRUCTURE -- CPU restatement of the reference's depth-error metrics (eval.py:164-207).

Pinned against the shim-imported reference function by tests/golden/make_golden_metrics.py (fixture
tests/golden/depth_metrics.npz, checked on any machine by tests/test_oracle_golden.py).
Only tests/ may import this module.
"""
import torch


def compute_depth_metrics_ref(pred_depth, gt_depth, min_depth, max_depth, median_scaling=True):
    """-> (abs_rel, sq_rel, rmse, log10, a1, a2, a3, ratio), fp32 arithmetic like the reference, a1..a3 in fp64 (eval.py:193-195)."""
    _, H, W = gt_depth.shape
    p = pred_depth.squeeze().reshape(-1, H * W)              # eval.py:176-177
    g = gt_depth.squeeze().reshape(-1, H * W)
    valid = (g > 0.5).logical_and(p > 0.5)                   # :178
    p, g = p[valid].clone(), g[valid]
    ratio = torch.median(gt_depth) / torch.median(p) if median_scaling else torch.tensor(0)   # :184 (median of ALL gt values)
    p[p < min_depth] = min_depth                             # :189-190
    p[p > max_depth] = max_depth
    th = torch.max(g / p, p / g)                             # :192
    a1, a2, a3 = [(th < 1.25 ** k).double().mean() for k in (1, 2, 3)]
    rmse = torch.sqrt(((g - p) ** 2).mean())                 # :197-198
    log10 = torch.mean(torch.abs(torch.log10(g) - torch.log10(p)))   # :203
    abs_rel = torch.mean(torch.abs(g - p) / g)               # :205
    sq_rel = torch.mean(((g - p) ** 2) / g)                  # :206
    return abs_rel, sq_rel, rmse, log10, a1, a2, a3, ratio

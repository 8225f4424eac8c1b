"""SOLOv2 NMS family on device tensors (reference: models/functions/nms.py). Small tensors; plain device ops."""
import torch
import torch.nn.functional as F


def point_nms(heat, kernel=2):
    """Keep local maxima of a 2x2 window (nms.py:8-12)."""
    assert kernel == 2
    hmax = F.max_pool2d(heat, (2, 2), stride=1, padding=1)
    return heat * (hmax[:, :, :-1, :-1] == heat).float()


def matrix_nms(cate_labels, seg_masks, sum_masks, cate_scores, sigma=2.0, kernel="gaussian"):
    """Parallel soft-NMS by pairwise mask IoU decay (nms.py:15-50)."""
    n = len(cate_labels)
    if n == 0:
        return []
    if seg_masks.is_cuda:
        # pairwise IoU of the bit-packed masks by AND + popcount (include/prn.h: prn_pairwise_iou) instead of an [n, h*w] x [h*w, n]
        # float GEMM: the intersections are the same integers and the quotient inter / ((area_i + area_j) - inter) is formed in the same
        # order, so the matrix is bit-identical (tests/test_eval_metrics.py pins the kernel on the reference's float matmul)
        from .metrics import mask_iou, matrix_nms_scores
        iou = mask_iou(seg_masks, seg_masks)
        if kernel in ("gaussian", "linear"):
            # the decay of the scores from the IoU matrix in two launches (prn_matrix_nms): the same operations per element as the dense form below
            return matrix_nms_scores(iou, cate_labels, cate_scores, float(sigma), kernel == "gaussian")
        iou = iou.triu(diagonal=1)
    else:
        flat = seg_masks.reshape(n, -1).float()
        inter = flat @ flat.t()
        area = sum_masks.expand(n, n)
        iou = (inter / (area + area.t() - inter)).triu(diagonal=1)
    lab = cate_labels.expand(n, n)
    decay = iou * (lab == lab.t()).float().triu(diagonal=1)
    comp = decay.max(0)[0].expand(n, n).t()
    if kernel == "linear":
        coef = ((1 - decay) / (1 - comp)).min(0)[0]
    else:
        coef = (torch.exp(-sigma * decay ** 2) / torch.exp(-sigma * comp ** 2)).min(0)[0]
    return cate_scores * coef


def mask_nms(cate_labels, seg_masks, sum_masks, cate_scores, nms_thr=0.5):
    """Greedy mask NMS (nms.py:53-81), with the pairwise IoU matrix computed once instead of per pair."""
    n = len(cate_scores)
    if n == 0:
        return []
    flat = seg_masks.reshape(n, -1).float()
    inter = flat @ flat.t()
    union = sum_masks[:, None] + sum_masks[None, :] - inter
    same = cate_labels[:, None] == cate_labels[None, :]
    suppress = (same & ((union <= 0) | (inter / union.clamp(min=1e-12) > nms_thr))).cpu()
    keep = [True] * n
    for i in range(n - 1):
        if keep[i]:
            for j in range(i + 1, n):
                if keep[j] and suppress[i, j]:
                    keep[j] = False
    return torch.tensor(keep, device=seg_masks.device).to(seg_masks.dtype)

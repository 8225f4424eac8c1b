"""Joint five-term loss (reference: models/functions/losses.py + vnl.py), restructured for a GPU that must not wait
for the host.

Interface and semantics follow the reference: `PlaneRecNetLoss().forward(net, mask_preds, cate_preds, kernel_preds,
depth_preds, gt_instances, gt_depths) -> {'ins','cat','dpt','pln','lav'}` with the quirks of SURVEY.md A.4.  The
structure is different:

  * everything that depends only on the ground truth -- SOLOv2 target assignment (losses.py:200-286), the random
    triplet indices of the virtual-normal loss (vnl.py:43-55, drawn from numpy's global RNG in the reference's call
    order), the depth-gradient map of the lava term (losses.py:169-185) -- is computed by `prepare()` on HOST copies of
    the GT and uploaded asynchronously.  A training loop calls `prepare()` before it enqueues the forward, so this host
    work overlaps with the previous step's backward on the GPU and the step contains no device->host synchronisation.
    (`forward()` without `targets=` calls it itself and then has to synchronise, like the reference does.)
  * per-(level, image) Python loops become one batched launch each: one dynamic 1x1 conv per image over all its
    positive cells (losses.py:86-93), one vectorised Dice / focal / lava evaluation for the batch, and ONE pass over all
    virtual-normal triplets of all planes of all images (the reference loops planes and images, vnl.py:119-165) with a
    single segmented sort for the "drop the best 25 %" rule.
  * lava: sum(up4(s) * g) is evaluated as sum(s * up4^T(g)); the adjoint-resized gradient map depends only on the GT.
"""
import collections
import os
import threading

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .config import cfg
from .funcs import center_of_mass, quarter_mask_u8


def _pin(x):
    return x.pin_memory()


class Targets:
    """Device-resident, GT-only inputs of one loss evaluation (built by PlaneRecNetLoss.prepare)."""
    __slots__ = ("B", "cell_ids", "cell_gidx", "cell_inv", "cells_unique", "n_pos", "n_pos_dev", "pos_img", "ins_labels", "cate_labels", "num_ins", "vnl", "lava_adj", "lava_gsum",
                 "ready")         # ready: event after which the tensors may be read (uploads issued on another stream), or None


LOSS_STREAMS = bool(int(os.environ.get("PRN_LOSS_STREAMS", "0")))        # off by default: see ops.BRANCH_STREAMS


class PlaneRecNetLoss(nn.Module):
    def __init__(self):
        super().__init__()
        s = cfg.solov2
        self.num_classes = cfg.num_classes
        self.num_grids, self.scale_ranges, self.strides, self.sigma = s.num_grids, s.fpn_scale_ranges, s.fpn_instance_strides, s.sigma
        self.focal_loss_alpha, self.focal_loss_gamma = cfg.focal_alpha, cfg.focal_gamma
        self.ins_loss_weight, self.conf_loss_weight = cfg.dice_weight, cfg.focal_weight
        self.depth_loss_weight, self.lava_loss_weight, self.pln_loss_weight = cfg.depth_weight, cfg.lava_weight, cfg.pln_weight
        self.depth_resolution, self.dataset_name = cfg.dataset.depth_resolution, cfg.dataset.name
        self.vnl = VNL_Loss((480, 640))                                     # hard-wired size: quirk Q5

    # ------------------------------------------------------------------ SOLOv2 targets (host)
    @torch.no_grad()
    def prepare_ground_truth(self, inst, mask_feat_size):
        """losses.py:200-286 for one image, on host tensors. Returns per-level lists
        (ins_label uint8 [n,h,w], cate_label int64 [S,S], ins_ind bool [S*S], grid_order list)."""
        boxes, labels, masks = inst["boxes"].cpu(), inst["classes"].cpu(), inst["masks"].cpu()
        small_all = quarter_mask_u8(masks.to(torch.uint8))
        cx_all, cy_all = center_of_mass(masks)
        nonempty = masks.flatten(1).sum(1) > 0
        which_l, cate_l, ind_l, order_l = self.assign_cells(boxes, labels, cx_all, cy_all, nonempty, mask_feat_size)
        fh, fw = int(mask_feat_size[0]), int(mask_feat_size[1])
        ins_l = []
        for which in which_l:
            if which:
                lab = torch.zeros(len(which), fh, fw, dtype=torch.uint8)
                sm = small_all[which]
                lab[:, :sm.shape[1], :sm.shape[2]] = sm
            else:
                lab = torch.zeros(0, fh, fw, dtype=torch.uint8)
            ins_l.append(lab)
        return ins_l, cate_l, ind_l, order_l

    def cell_regions(self, boxes, cx_all, cy_all, mask_feat_size):
        """The arithmetic of losses.py:213-262 for MANY instances at once (any number of images' instances back to back): per level
        (hit, top, down, left, right) -- lists over the instances.  The same IEEE operations in the same precisions as the reference's 0-d tensor code
        (float64 boxes, float32 centres; float32 op float64 -> float64, a python float next to a float32 value stays float32), as numpy ARRAY operations:
        ~100 of them per batch instead of ~25 scalar ones per instance and level (1.2 ms per batch of 8 before).
        The python-float operands (g, up_w, up_h) are cast to the OTHER operand's dtype explicitly, so the result does not depend on
        numpy's scalar promotion rules (NumPy 1.x would promote float32 // python float to float64; NEP 50 does not)."""
        fh, fw = int(mask_feat_size[0]), int(mask_feat_size[1])
        up_h, up_w = fh * 4, fw * 4
        bx = np.asarray(boxes.numpy() if torch.is_tensor(boxes) else boxes, dtype=np.float64).reshape(-1, 4)
        cx = np.asarray(cx_all.numpy() if torch.is_tensor(cx_all) else cx_all, dtype=np.float32)
        cy = np.asarray(cy_all.numpy() if torch.is_tensor(cy_all) else cy_all, dtype=np.float32)
        f32, f64 = np.float32, np.float64
        areas = np.sqrt((bx[:, 2] - bx[:, 0]) * (bx[:, 3] - bx[:, 1]))
        hw = 0.5 * (bx[:, 2] - bx[:, 0]) * self.sigma
        hh = 0.5 * (bx[:, 3] - bx[:, 1]) * self.sigma
        out = []
        with np.errstate(all="ignore"):
            for (lo, hi), S in zip(self.scale_ranges, self.num_grids):
                g = 1.0 / S
                hit = (areas >= lo) & (areas <= hi)
                coord_w = ((cx / f32(up_w)) // f32(g)).astype(np.int64)                                     # float32 centre: all float32
                coord_h = ((cy / f32(up_h)) // f32(g)).astype(np.int64)
                top = np.maximum(np.maximum(0, (((cy - hh) / f64(up_h)) // f64(g)).astype(np.int64)), coord_h - 1)      # float32 - float64 half extent: float64
                down = np.minimum(np.minimum(S - 1, (((cy + hh) / f64(up_h)) // f64(g)).astype(np.int64)), coord_h + 1)
                left = np.maximum(coord_w - 1, np.maximum(0, (((cx - hw) / f64(up_w)) // f64(g)).astype(np.int64)))
                right = np.minimum(np.minimum(S - 1, (((cx + hw) / f64(up_w)) // f64(g)).astype(np.int64)), coord_w + 1)
                out.append((hit.tolist(), top.tolist(), down.tolist(), left.tolist(), right.tolist()))
        return out

    def assign_cells(self, boxes, labels, cx_all, cy_all, nonempty, mask_feat_size, regions=None, base=0):
        """The centre-region assignment of losses.py:213-279 given the per-instance mask statistics (centre of mass, empty flag):
        per level -> (instance index of every positive cell, category map [S,S], positive flags [S*S], cell index of every positive
        cell).  Shared by the host path (prepare_ground_truth) and the device path (targets.DeviceTargetBuilder, which gets the
        statistics from prn_gt_mask_stats and passes `regions` = cell_regions() of the whole batch, this image's instances starting at `base`)."""
        lab = labels.numpy() if torch.is_tensor(labels) else np.asarray(labels)
        ne = (nonempty.numpy() if torch.is_tensor(nonempty) else np.asarray(nonempty)).tolist()
        n = len(ne)
        if regions is None:
            regions, base = self.cell_regions(boxes, cx_all, cy_all, mask_feat_size), 0
        which_l, cate_l, ind_l, order_l = [], [], [], []
        for (hit, top, down, left, right), S in zip(regions, self.num_grids):
            cate = np.full((S, S), self.num_classes, dtype=np.int64)
            ind = np.zeros(S * S, dtype=np.bool_)
            which, order = [], []
            for i in range(n):
                k = base + i
                if not hit[k] or not ne[i]:
                    continue
                t, d_, l, r_ = top[k], down[k], left[k], right[k]
                cate[t:d_ + 1, l:r_ + 1] = lab[i]
                for r in range(t, d_ + 1):
                    for c in range(l, r_ + 1):
                        which.append(i)
                        order.append(r * S + c)
                        ind[r * S + c] = True
            which_l.append(which)
            cate_l.append(torch.from_numpy(cate))
            ind_l.append(torch.from_numpy(ind))
            order_l.append(order)
        return which_l, cate_l, ind_l, order_l

    # ------------------------------------------------------------------ GT-only work, before the forward
    @torch.no_grad()
    def prepare_host(self, gt_instances, hw, mask_feat_size=None, with_vnl=True, pin=True):
        """Pure host part (numpy / CPU torch; safe to run on a worker thread -- see TargetPrefetcher): SOLOv2 targets and
        the virtual-normal triplet indices for a list of per-image GT dicts."""
        H, W = hw
        fh, fw = mask_feat_size if mask_feat_size is not None else (H // 4, W // 4)
        host = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in g.items()} for g in gt_instances]
        B, L = len(host), len(self.num_grids)
        level_start = np.concatenate([[0], np.cumsum([g * g for g in self.num_grids])])
        cell_ids, ins_labels, cate_rows, n_pos, num_ins = [], [], [[] for _ in range(L)], [], 0
        for b in range(B):
            ins_l, cate_l, ind_l, order_l = self.prepare_ground_truth(host[b], (fh, fw))
            cell_ids.append(np.concatenate([level_start[lv] + np.asarray(order_l[lv], dtype=np.int64) for lv in range(L)]))
            ins_labels.append(torch.cat(ins_l, 0))
            n_pos.append(int(cell_ids[-1].shape[0]))
            num_ins += sum(int(i.sum()) for i in ind_l)
            for lv in range(L):
                cate_rows[lv].append(cate_l[lv].flatten())
        pin = _pin if (pin and torch.cuda.is_available()) else (lambda x: x)       # page-locked staging: the later H2D copies are truly async
        n_cells = int(level_start[-1])
        cell_gidx = np.concatenate([b * n_cells + cell_ids[b] for b in range(B)]) if B else np.zeros(0, np.int64)
        cell_u, cell_inv, cell_cnt = np.unique(cell_gidx, return_inverse=True, return_counts=True)
        cell_mult = int(cell_cnt.max()) if cell_cnt.size else 1
        return {"B": B, "hw": (H, W), "feat": (fh, fw), "n_pos": n_pos, "num_ins": num_ins,
                # (as a tensor too: built in upload() it would be pageable memory, and a copy from pageable memory blocks the trainer
                # until the stream it is issued on has caught up -- the GPU then idles ~1 ms per step behind the blocked trainer)
                "n_pos_f": pin(torch.as_tensor(n_pos, dtype=torch.float32)),
                # rows of the positive cells in the [B * n_cells, E] matrix of predicted kernels: the distinct ones, and where each
                # listed cell is among them (a cell claimed by two instances is listed twice); see instance_terms
                "cell_gidx": pin(torch.from_numpy(cell_u if cell_mult > 1 else cell_gidx)),          # (np.unique sorts: listed order when nothing repeats)
                "cell_inv": pin(torch.from_numpy(cell_inv)) if cell_mult > 1 else None,
                "cells_unique": cell_mult <= 2,
                "cell_ids": pin(torch.from_numpy(np.concatenate(cell_ids))), "pos_img": pin(torch.from_numpy(np.repeat(np.arange(B), n_pos))),
                "ins_labels": pin(torch.cat(ins_labels, 0)),
                # level-major, image-minor flattening == the reference's cat order (losses.py:121-131)
                "cate_labels": pin(torch.cat([r for lv in range(L) for r in cate_rows[lv]])),
                "vnl": (self.vnl.prepare_host(host, (H, W)) if cfg.use_plane_loss else None) if with_vnl else "deferred"}

    @torch.no_grad()
    def upload(self, h, gt_depths, device):
        """Asynchronous uploads of prepare_host()'s result + the GT-only device work of the lava term."""
        t = Targets()
        t.ready = None
        t.B, t.n_pos, t.num_ins = h["B"], h["n_pos"], h["num_ins"]
        up = lambda x: x.to(device, non_blocking=True)
        t.cell_ids = list(up(h["cell_ids"]).split(h["n_pos"]))
        t.cell_gidx, t.cells_unique = up(h["cell_gidx"]), h["cells_unique"]
        t.cell_inv = up(h["cell_inv"]) if h["cell_inv"] is not None else None
        t.n_pos_dev = up(h["n_pos_f"])
        t.pos_img, t.ins_labels, t.cate_labels = up(h["pos_img"]), up(h["ins_labels"]), up(h["cate_labels"])
        t.vnl = self.vnl.upload(h["vnl"], device) if h["vnl"] is not None else None
        t.lava_gsum = t.lava_adj = None
        if cfg.use_lava_loss:
            B, (H, W), (fh, fw) = h["B"], h["hw"], h["feat"]
            # Q3: dataset_name never equals 'ScanNet' / 'Stanford 2D3DS' (it is 'ScanNetDataset'), so valid_mask is None
            if gt_depths.is_cuda and gt_depths.dtype == torch.float32 and LAVA_GT_KERNEL:
                # sobel^2 / depth^2, both clamps and the threshold in one launch (include/prn.h: prn_lava_gt_weights), the per-image sums in the library's
                # fixed-order channel sum (the map viewed as [1, B, H*W])
                gd = gt_depths.contiguous()
                grad = torch.empty_like(gd)
                ops.check(ops.lib.prn_lava_gt_weights(ops._p(gd), ops._p(grad), B, H, W, float(self.depth_resolution), ops._stream()), "prn_lava_gt_weights")
                t.lava_gsum = ops.channel_sum(grad.view(1, B, H, W))
            else:
                grad = sobel_sq(gt_depths) / gt_depths.clamp(min=self.depth_resolution) ** 2
                grad = grad.clamp(max=1e-2)
                grad = torch.where(grad < 1e-4, torch.zeros_like(grad), grad)
                t.lava_gsum = grad.flatten(1).sum(1)
            adj = torch.empty(B, 1, fh, fw, device=device, dtype=torch.float32)
            ops.check(ops.lib.prn_resize_bilinear_bwd(ops._p(grad.contiguous()), ops._p(adj), B, fh, fw, H, W, ops._stream()), "prn_resize_bilinear_bwd")
            t.lava_adj = adj
        return t

    def prepare(self, gt_instances, gt_depths, device, mask_feat_size=None):
        """gt_instances: list of dicts of HOST tensors (device tensors are accepted but force a synchronising copy).
        gt_depths: [B,1,H,W] on `device`.  Returns Targets."""
        return self.upload(self.prepare_host(gt_instances, tuple(gt_depths.shape[-2:]), mask_feat_size), gt_depths, device)

    # ------------------------------------------------------------------ forward
    def forward(self, net, mask_preds, cate_preds, kernel_preds, depth_preds, gt_instances, gt_depths, targets=None):
        dev = mask_preds.device
        B = mask_preds.shape[0]
        fh, fw = mask_preds.shape[-2:]
        t = targets if targets is not None else self.prepare(gt_instances, gt_depths, dev, (fh, fw))
        if getattr(t, "ready", None) is not None:            # uploads were issued on the side stream (TargetPrefetcher.get(overlap=True))
            torch.cuda.current_stream().wait_event(t.ready)
            t.ready = None
        E = kernel_preds[0].shape[1]

        def instance_terms():
            out = {}
            # ---- ins (Dice) -- losses.py:69-118 : one dynamic conv per image over all of its positive cells
            flat_k = torch.cat([k.reshape(B, E, -1) for k in kernel_preds], 2)                          # [B, E, 3728]
            preds = []
            per_image = torch.split(mask_preds, 1)          # one backward node (cat) instead of B zero-filled slice gradients + adds
            if t.cells_unique and sum(t.n_pos) > 0:
                # all images' positive cells in ONE gather of the distinct rows (index_select: its gradient is one index_add_,
                # collision-free) plus, when a cell is listed more than once, one expansion whose gradient adds at most two rows
                # per cell (order-free: a + b == b + a) -- instead of B advanced-indexing ops with B sort-based index_put gradients.
                # (A cell claimed by three or more instances takes the per-image path below: deterministic there too.)
                rows = flat_k.transpose(1, 2).reshape(B * flat_k.shape[2], E).index_select(0, t.cell_gidx)
                if t.cell_inv is not None:
                    rows = rows.index_select(0, t.cell_inv)
                w_img = [r.reshape(-1, E, 1, 1) for r in torch.split(rows, t.n_pos)]
            else:
                flat_kb = flat_k.unbind(0)
                w_img = [flat_kb[b][:, t.cell_ids[b]].t().reshape(t.n_pos[b], E, 1, 1).contiguous() if t.n_pos[b] else None for b in range(B)]
            for b in range(B):
                if t.n_pos[b] == 0:
                    continue
                preds.append(ops.conv2d(per_image[b], w_img[b]).view(t.n_pos[b], fh, fw))
            logits = torch.cat(preds, 0)                                                                # [sum n_pos, fh, fw]
            if FUSED_LOSS and logits.is_cuda and logits.shape[0] > 0 and (fh * fw) % 4 == 0 and B <= 64:
                # Dice + lava as ONE pass over the logits each way (include/prn.h: prn_mask_loss_fwd / _bwd)
                lava = cfg.use_lava_loss
                ins, lav = _MaskLoss.apply(logits, t.ins_labels, t.pos_img, t.lava_adj if lava else None, t.lava_gsum if lava else None,
                                           t.n_pos_dev if lava else None, float(self.ins_loss_weight), float(self.lava_loss_weight))
                out["ins"] = ins
                if lava:
                    out["lav"] = lav
                return out
            ins_sig = torch.sigmoid(logits)
            out["ins"] = dice_loss(ins_sig, t.ins_labels).mean() * self.ins_loss_weight
            # ---- lav -- losses.py:169-197 : sum(up(s) * g) / (sum(g) * n)  ==  sum(s * up^T(g)) / (sum(g) * n)
            if cfg.use_lava_loss:
                seg, npos = t.pos_img, t.n_pos_dev
                num = torch.zeros(B, device=dev, dtype=ins_sig.dtype).index_add_(0, seg, (ins_sig * t.lava_adj[seg, 0]).flatten(1).sum(1))
                ok = (t.lava_gsum > 0) & (npos > 0)
                per_img = torch.where(ok, num / (t.lava_gsum * npos).clamp(min=1e-30), torch.zeros_like(num))
                # mean over qualifying images; 0 when none qualifies (the reference then returns a [1]-shaped zero, quirk Q4)
                out["lav"] = per_img.sum() / ok.sum().clamp(min=1) * self.lava_loss_weight
            return out

        def category_term():
            # ---- cat (sigmoid focal, sum / (num_pos + 1)) -- losses.py:121-138
            flat_pred = torch.cat([c.permute(0, 2, 3, 1).reshape(-1, self.num_classes) for c in cate_preds])
            if FUSED_LOSS and FUSED_CAT_DPT and flat_pred.is_cuda:
                return {"cat": self.conf_loss_weight * _FocalSum.apply(flat_pred, t.cate_labels, self.focal_loss_alpha, self.focal_loss_gamma) / (t.num_ins + 1)}
            onehot = F.one_hot(t.cate_labels, self.num_classes + 1)[:, : self.num_classes].to(flat_pred.dtype)
            return {"cat": self.conf_loss_weight * sigmoid_focal_sum(flat_pred, onehot, self.focal_loss_alpha, self.focal_loss_gamma) / (t.num_ins + 1)}

        def depth_terms():
            # ---- dpt (RMSE-log at full resolution) -- losses.py:142-147 (the clamp there is discarded: quirk Q2)
            dp = ops.resize_bilinear(depth_preds, (2 * depth_preds.shape[2], 2 * depth_preds.shape[3]))
            if FUSED_LOSS and FUSED_CAT_DPT and dp.is_cuda:
                out = {"dpt": self.depth_loss_weight * _RmseLog.apply(dp, gt_depths, cfg.dataset.min_depth, 1e-9)}
            else:
                out = {"dpt": self.depth_loss_weight * rmse_log(dp, gt_depths, gt_depths > cfg.dataset.min_depth)}
            # ---- pln (virtual normals) -- losses.py:151-165
            if cfg.use_plane_loss:
                out["pln"] = self.vnl.batched(dp, gt_depths, t.vnl).mean() * self.pln_loss_weight
            return out

        # three independent chains of small launches (the backward replays on the same streams): ops.run_branches
        parts = ops.run_branches([instance_terms, category_term, depth_terms]) if (LOSS_STREAMS and mask_preds.is_cuda) else \
            [instance_terms(), category_term(), depth_terms()]
        losses = {}
        for part in parts:
            losses.update(part)
        if all(k in losses for k in ("ins", "cat", "dpt", "pln", "lav")):
            losses = {k: losses[k] for k in ("ins", "cat", "dpt", "pln", "lav")}      # reference order (logging)
        return losses


FUSED_CAT_DPT = bool(int(os.environ.get("PRN_FUSED_CAT_DPT", "1")))  # 0: focal / RMSE-log terms operator by operator (A/B)
FUSED_LOSS = bool(int(os.environ.get("PRN_FUSED_LOSS", "1")))       # 0: the operator-by-operator evaluation below (A/B, cross-check in the tests)


class _MaskLoss(torch.autograd.Function):
    """(ins, lav) from the mask logits of all positive cells: sigmoid, Dice sums and the lava numerator in one pass forward,
    the logit gradient of both terms in one pass backward."""

    @staticmethod
    def forward(ctx, logits, labels, pos_img, adj, gsum, npos, w_ins, w_lav):
        logits, labels = logits.contiguous(), labels.contiguous()
        P, HW = logits.shape[0], logits[0].numel()
        B = adj.shape[0] if adj is not None else 1
        dev = logits.device
        out = torch.empty(2, device=dev, dtype=torch.float32)
        coef = torch.empty(P, 3, device=dev, dtype=torch.float32)
        ws = torch.empty(ops.lib.prn_mask_loss_ws_floats(P), device=dev, dtype=torch.float32)
        adj_c = adj.contiguous() if adj is not None else None
        ops.check(ops.lib.prn_mask_loss_fwd(ops._p(logits), ops._p(labels), ops._p(adj_c), ops._p(pos_img), ops._p(gsum), ops._p(npos), ops._p(out),
                                            ops._p(coef), ops._p(ws), P, HW, B, w_ins, w_lav, ops._stream()), "prn_mask_loss_fwd")
        ctx.save_for_backward(logits, labels, pos_img, adj_c, coef)
        ins, lav = out.unbind(0)
        return ins, lav

    @staticmethod
    def backward(ctx, g_ins, g_lav):
        logits, labels, pos_img, adj, coef = ctx.saved_tensors
        dz = torch.empty_like(logits)
        g_ins = None if g_ins is None else g_ins.contiguous().float()
        g_lav = None if g_lav is None else g_lav.contiguous().float()
        ops.check(ops.lib.prn_mask_loss_bwd(ops._p(logits), ops._p(labels), ops._p(adj), ops._p(pos_img), ops._p(coef), ops._p(g_ins), ops._p(g_lav),
                                            ops._p(dz), logits.shape[0], logits[0].numel(), ops._stream()), "prn_mask_loss_bwd")
        return dz, None, None, None, None, None, None, None


def dice_loss(p, t):
    p = p.reshape(p.shape[0], -1)
    t = t.reshape(t.shape[0], -1).float()
    return 1 - 2 * (p * t).sum(1) / ((p * p).sum(1) + 0.001 + (t * t).sum(1) + 0.001)


def sigmoid_focal_sum(x, t, alpha, gamma):
    p = torch.sigmoid(x)
    ce = F.binary_cross_entropy_with_logits(x, t, reduction="none")
    p_t = p * t + (1 - p) * (1 - t)
    loss = ce * (1 - p_t) ** gamma
    if alpha >= 0:
        loss = (alpha * t + (1 - alpha) * (1 - t)) * loss
    return loss.sum()


def rmse_log(pred, gt, valid, clamp=1e-9):
    n = pred.shape[0]
    l1 = (torch.log(pred.reshape(n, -1).clamp(min=clamp)) - torch.log(gt.reshape(n, -1).clamp(min=clamp))).abs().mul(valid.reshape(n, -1))
    return torch.sqrt((l1 ** 2).sum(1) / valid.reshape(n, -1).sum(1)).mean()


LAVA_GT_KERNEL = os.environ.get("PRN_LAVA_GT_KERNEL", "1") == "1"      # 0: the tensor formulation (cross-check)


@torch.no_grad()
def sobel_sq(d):
    """gx^2 + gy^2 of the reflect-padded 3x3 Sobel / 8 (losses.py:304-329)."""
    p = F.pad(d, (1, 1, 1, 1), mode="reflect")
    tl, tc, tr = p[..., :-2, :-2], p[..., :-2, 1:-1], p[..., :-2, 2:]
    ml, mr = p[..., 1:-1, :-2], p[..., 1:-1, 2:]
    bl, bc, br = p[..., 2:, :-2], p[..., 2:, 1:-1], p[..., 2:, 2:]
    gx = (tl - tr + 2 * ml - 2 * mr + bl - br) / 8.0
    gy = (tl + 2 * tc + tr - bl - 2 * bc - br) / 8.0
    return gx ** 2 + gy ** 2


class _GatherRows(torch.autograd.Function):
    """rows = src[idx]; backward = deterministic segmented sum through the precomputed sort of idx: `order` = argsort(idx),
    `counts[r]` = number of times row r of src is gathered (length src.shape[0])."""

    @staticmethod
    def forward(ctx, src, idx, order, counts):
        ctx.save_for_backward(order, counts)
        return src[idx]

    @staticmethod
    def backward(ctx, g):
        order, counts = ctx.saved_tensors
        return torch.segment_reduce(g[order].contiguous(), "sum", lengths=counts, axis=0, unsafe=True), None, None, None


class _FocalSum(torch.autograd.Function):
    """sigmoid_focal_sum(x, one_hot(labels)) in one pass each way (include/prn.h: prn_focal_sum_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, labels, alpha, gamma):
        x = x.contiguous()
        rows, C = x.shape
        out = torch.empty(1, device=x.device, dtype=torch.float32)
        ws = torch.empty(ops.lib.prn_loss_ws_doubles(1), device=x.device, dtype=torch.float64)
        ops.check(ops.lib.prn_focal_sum_fwd(ops._p(x), ops._p(labels), ops._p(out), ops._p(ws), rows, C, float(alpha), float(gamma), ops._stream()), "prn_focal_sum_fwd")
        ctx.save_for_backward(x, labels)
        ctx.cfg = (float(alpha), float(gamma))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        x, labels = ctx.saved_tensors
        dx = torch.empty_like(x)
        ops.check(ops.lib.prn_focal_sum_bwd(ops._p(x), ops._p(labels), ops._p(g.float().reshape(1).contiguous()), ops._p(dx), x.shape[0], x.shape[1],
                                            ctx.cfg[0], ctx.cfg[1], ops._stream()), "prn_focal_sum_bwd")
        return dx, None, None, None


class _RmseLog(torch.autograd.Function):
    """rmse_log(pred, gt, gt > min_depth) in one pass each way (include/prn.h: prn_rmse_log_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, pred, gt, min_depth, clamp):
        pred, gt = pred.contiguous(), gt.contiguous()
        B = pred.shape[0]
        HW = pred.numel() // B
        out = torch.empty(1, device=pred.device, dtype=torch.float32)
        coef = torch.empty(B, device=pred.device, dtype=torch.float32)
        ws = torch.empty(ops.lib.prn_loss_ws_doubles(B), device=pred.device, dtype=torch.float64)
        ops.check(ops.lib.prn_rmse_log_fwd(ops._p(pred), ops._p(gt), ops._p(out), ops._p(coef), ops._p(ws), B, HW, float(min_depth), float(clamp), ops._stream()),
                  "prn_rmse_log_fwd")
        ctx.save_for_backward(pred, gt, coef)
        ctx.cfg = (float(min_depth), float(clamp))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        pred, gt, coef = ctx.saved_tensors
        B = pred.shape[0]
        d = torch.empty_like(pred)
        ops.check(ops.lib.prn_rmse_log_bwd(ops._p(pred), ops._p(gt), ops._p(coef), ops._p(g.float().reshape(1).contiguous()), ops._p(d), B, pred.numel() // B,
                                           ctx.cfg[0], ctx.cfg[1], ops._stream()), "prn_rmse_log_bwd")
        return d, None, None, None


class _TrimmedMeans(torch.autograd.Function):
    """VNL_Loss._trimmed_means with its gradient written out (see there).  On the device: the sort key, ATen's sort, one workgroup per
    region and one combining thread (include/prn.h: prn_vnl_trim_*) instead of the ~65-launch operator chain around the sort."""

    @staticmethod
    def forward(ctx, loss, valid, t):
        ctx.n = loss.shape[0]
        if FUSED_LOSS and loss.is_cuda and loss.dtype == torch.float64 and t.n_seg > 0:
            dev, n, nseg = loss.device, loss.shape[0], int(t.seg_start.shape[0])
            loss_c = loss.detach().contiguous()
            v8 = valid.contiguous().view(torch.uint8) if valid.dtype == torch.bool else valid.contiguous()
            key = torch.empty(n, device=dev, dtype=torch.float64)
            ops.check(ops.lib.prn_vnl_trim_key(ops._p(loss_c), ops._p(v8), ops._p(t.seg), ops._p(key), n, ops._stream()), "prn_vnl_trim_key")
            order = torch.argsort(key)
            out = torch.empty(t.B, device=dev, dtype=torch.float64)
            seg_sum, seg_coef = torch.empty(nseg, device=dev, dtype=torch.float64), torch.empty(nseg, device=dev, dtype=torch.float64)
            seg_m = torch.empty(nseg, device=dev, dtype=torch.int32)
            plane8 = t.seg_is_plane.view(torch.uint8) if t.seg_is_plane.dtype == torch.bool else t.seg_is_plane
            ws = torch.empty(ops.lib.prn_vnl_trim_ws_bytes(nseg) // 8, device=dev, dtype=torch.float64)
            ops.check(ops.lib.prn_vnl_trim_fwd(ops._p(loss_c), ops._p(v8), ops._p(order), ops._p(t.seg_start), ops._p(plane8), ops._p(t.seg_img), ops._p(t.N),
                                               nseg, n, t.B, ops._p(out), ops._p(seg_sum), ops._p(seg_m), ops._p(seg_coef), ops._p(ws), ops._stream()), "prn_vnl_trim_fwd")
            ctx.save_for_backward(loss_c, order, seg_m, seg_coef)
            ctx.t = t
            ctx.fused = True
            return out
        out, order, coef, img_s = VNL_Loss._trimmed_means_autograd(loss.detach(), valid, t, loss.device, want_coef=True)
        ctx.save_for_backward(order, coef, img_s)
        ctx.fused = False
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.fused:
            loss_c, order, seg_m, seg_coef = ctx.saved_tensors
            t = ctx.t
            grad = torch.empty(ctx.n, device=g.device, dtype=torch.float64)
            ops.check(ops.lib.prn_vnl_trim_bwd(ops._p(loss_c), ops._p(order), ops._p(t.seg_start), ops._p(seg_m), ops._p(seg_coef), ops._p(t.seg_img),
                                               ops._p(g.double().contiguous()), int(t.seg_start.shape[0]), ctx.n, ops._p(grad), ops._stream()), "prn_vnl_trim_bwd")
            return grad, None, None
        order, coef, img_s = ctx.saved_tensors
        grad = torch.empty(ctx.n, device=g.device, dtype=coef.dtype).index_copy_(0, order, coef * g.to(coef.dtype)[img_s])   # order is a permutation
        return grad, None, None


class _VnlTriplets(torch.autograd.Function):
    """(loss [n] float64, valid [n] uint8) of all sampled triplets from the predicted / GT depth maps (include/prn.h:
    prn_vnl_triplets); the kernel also leaves d loss / d depth of each triplet's three points, so the backward pass is one
    fixed-order scatter (prn_vnl_scatter) through the GT-only inverse of the sampling pattern."""

    @staticmethod
    def forward(ctx, pred, gt, t, hw, delta_z):
        n, dev = t.n_tot, pred.device
        H, W = hw
        loss = torch.empty(n, device=dev, dtype=torch.float64)
        valid = torch.empty(n, device=dev, dtype=torch.uint8)
        g3 = torch.empty(n, 3, device=dev, dtype=torch.float32)
        ops.check(ops.lib.prn_vnl_triplets(ops._p(pred), ops._p(gt), ops._p(t.gid32), ops._p(t.seg), ops._p(t.seg_is_plane), ops._p(t.seg_normal),
                                           ops._p(t.seg_img), ops._p(t.fx), ops._p(t.fy), ops._p(loss), ops._p(valid), ops._p(g3), n, H, W, delta_z,
                                           ops._stream()), "prn_vnl_triplets")
        ctx.save_for_backward(g3)
        ctx.t, ctx.shape = t, pred.shape
        ctx.mark_non_differentiable(valid)
        return loss, valid

    @staticmethod
    def backward(ctx, g_loss, _g_valid):
        (g3,), t = ctx.saved_tensors, ctx.t
        npts = 1
        for d in ctx.shape:
            npts *= d
        if t.gid_start is None:                              # GT only: per-point counts -> exclusive prefix sum (once per step)
            counts = torch.zeros(npts, dtype=torch.int64, device=g3.device).scatter_add_(0, t.gid_flat, torch.ones_like(t.gid_flat))
            t.gid_start = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)])
        dd = torch.empty(npts, device=g3.device, dtype=torch.float32)
        ops.check(ops.lib.prn_vnl_scatter(ops._p(g_loss.contiguous().double()), ops._p(g3), ops._p(t.gid_order), ops._p(t.gid_start), ops._p(dd), npts,
                                          t.n_tot, ops._stream()), "prn_vnl_scatter")
        return dd.view(ctx.shape), None, None, None, None


class VNLTargets:
    __slots__ = ("B", "N", "fx", "fy", "gid", "seg", "seg_start", "seg_img", "seg_is_plane", "seg_normal", "n_seg", "n_tot",
                 "gid_flat", "gid_order", "gid_counts", "gid32", "gid_start")


class VNL_Loss(nn.Module):
    """Virtual-normal plane loss (vnl.py:6-165), all planes of all images in one pass."""

    def __init__(self, input_size, delta_cos=0.867, delta_z=0.0001, sample_ratio=0.3):
        super().__init__()
        H, W = input_size
        self.input_size = input_size
        self.register_buffer("u_u0", torch.arange(W, dtype=torch.float32).view(1, 1, W).expand(1, H, W) - float(W // 2), persistent=False)
        self.register_buffer("v_v0", torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(1, H, W) - float(H // 2), persistent=False)
        self.delta_cos, self.delta_z, self.sample_ratio = delta_cos, delta_z, sample_ratio

    def _draw(self, num):
        """vnl.py:43-55: three (choice, shuffle) pairs from numpy's global RNG."""
        H, W = self.input_size
        if not num <= W * H:
            raise AssertionError()
        n = int(num * self.sample_ratio)
        out = []
        for _ in range(3):
            p = np.random.choice(num, n, replace=True)
            np.random.shuffle(p)
            out.append(p)
        return out

    @torch.no_grad()
    def prepare_host(self, host_instances, hw, pin=True):
        """Host: pixel ids of every sampled triplet (global over the batch) + segment bookkeeping, as CPU tensors."""
        H, W = hw
        gids, seg_len, seg_img, seg_plane, normals, N_per, fx, fy = [], [], [], [], [], [], [], []
        for b, g in enumerate(host_instances):
            masks = g["masks"].numpy().astype(bool)
            K = g["k_matrix"].numpy()
            fx.append(K[0, 0]); fy.append(K[1, 1])
            N = masks.shape[0]
            N_per.append(N)
            planes = g["plane_paras"].numpy()[:, :3]
            regions = [masks[i] for i in range(N)]
            nonplanar = ~masks.any(0) if N > 0 else np.ones((H, W), bool)
            if int(nonplanar.sum()) > 0:
                regions.append(nonplanar)
            for r, m in enumerate(regions):
                px = np.flatnonzero(m)
                p123 = self._draw(px.shape[0])
                gids.append(np.stack([px[p] for p in p123], 0) + b * H * W)           # [3, n]
                seg_len.append(p123[0].shape[0])
                seg_img.append(b)
                seg_plane.append(r < N)
                normals.append(planes[r] if r < N else np.zeros(3))
        seg_len = np.asarray(seg_len, dtype=np.int64)
        n_seg = len(seg_len)
        pin = _pin if (pin and torch.cuda.is_available()) else (lambda x: x)
        return {"B": len(host_instances), "n_seg": n_seg, "n_tot": int(seg_len.sum()), "npts": len(host_instances) * H * W,
                "N": pin(torch.as_tensor(N_per, dtype=torch.float64)), "fx": pin(torch.as_tensor(np.asarray(fx), dtype=torch.float64)),
                "fy": pin(torch.as_tensor(np.asarray(fy), dtype=torch.float64)),
                # the two big index arrays travel as int32 over PCIe (12 B per triplet less) and are widened on the device
                "gid": pin(torch.from_numpy((np.concatenate(gids, 1) if gids else np.zeros((3, 0), np.int64)).astype(np.int32))),
                "seg": pin(torch.from_numpy(np.repeat(np.arange(n_seg, dtype=np.int32), seg_len))),
                "seg_start": pin(torch.from_numpy(np.concatenate([[0], np.cumsum(seg_len)[:-1]]) if n_seg else np.zeros(0, np.int64))),
                "seg_img": pin(torch.as_tensor(seg_img, dtype=torch.int64)), "seg_is_plane": pin(torch.as_tensor(seg_plane, dtype=torch.bool)),
                "seg_normal": pin(torch.from_numpy(np.asarray(normals, dtype=np.float64).reshape(-1, 3)))}

    @staticmethod
    def upload(h, device):
        t = VNLTargets()
        t.B, t.n_seg, t.n_tot = h["B"], h["n_seg"], h["n_tot"]
        for k in ("N", "fx", "fy", "gid", "seg", "seg_start", "seg_img", "seg_is_plane", "seg_normal"):
            setattr(t, k, h[k].to(device, non_blocking=True))
        gid32 = t.gid                                        # int32 as uploaded: the radix sort below has half the key bits to do
        t.gid32 = gid32 if gid32.dtype == torch.int32 else gid32.int()
        t.gid_start = None                                   # filled by the fused path (prefix sum of the per-point counts)
        t.gid, t.seg = t.gid.long(), t.seg.long()
        # inverse of the triplet gather, built once per step (GT only): which gathered rows land on which cloud point
        # (sort + per-point counts over ALL B*H*W cloud points: fixed-size outputs, so no device->host sync)
        t.gid_flat = t.gid.reshape(-1)
        srt = torch.sort(gid32.reshape(-1)) if gid32.dtype == torch.int32 else torch.sort(t.gid_flat)
        t.gid_order = srt.indices
        if h.get("npts"):
            # first sorted position of every cloud point (= exclusive prefix sum of the per-point counts) by binary search in the
            # sorted ids: GT only, so it is built here, with the uploads, instead of in the backward pass on the main stream
            t.gid_start = torch.searchsorted(srt.values, torch.arange(h["npts"] + 1, device=device, dtype=srt.values.dtype))
        t.gid_counts = None                                  # filled by _triplets_t, which knows the size of the cloud
        return t

    def prepare(self, host_instances, hw, device):
        return self.upload(self.prepare_host(host_instances, hw), device)

    def _cloud(self, depth, t):
        """vnl.py:34-41 for the batch: [B,1,H,W] -> [B*H*W, 3]."""
        u, v = self.u_u0.to(depth.device), self.v_v0.to(depth.device)
        fx, fy = t.fx.to(depth.dtype).view(-1, 1, 1, 1), t.fy.to(depth.dtype).view(-1, 1, 1, 1)
        ad = depth.abs()
        return torch.stack([u * ad / fx, v * ad / fy, depth], -1).reshape(-1, 3)

    @staticmethod
    def _triplets(cloud, gid):
        return torch.stack([cloud[gid[0]], cloud[gid[1]], cloud[gid[2]]], 2)                     # [n, xyz, p]

    @staticmethod
    def _triplets_t(cloud, t):
        """Same gather with a backward that does not sort: autograd's index_put backward sorts the 0.84 M indices of each of
        the three gathers and accumulates serially (1.6 ms/step); the inverse map is known from the GT-only index set."""
        n = t.gid.shape[1]
        if n == 0 or not cloud.requires_grad:
            return VNL_Loss._triplets(cloud, t.gid)
        if t.gid_counts is None:
            t.gid_counts = torch.zeros(cloud.shape[0], dtype=torch.int64, device=cloud.device).scatter_add_(0, t.gid_flat, torch.ones_like(t.gid_flat))
        return _GatherRows.apply(cloud, t.gid_flat, t.gid_order, t.gid_counts).view(3, n, 3).permute(1, 2, 0)

    def _filter(self, pw, delta_diff, delta_cos=0.985):
        """vnl.py:71-104 with the 3x3 Gram matrix written out (no batched GEMM): delta_diff is per-triplet."""
        d = torch.stack([pw[:, :, 1] - pw[:, :, 0], pw[:, :, 2] - pw[:, :, 0], pw[:, :, 2] - pw[:, :, 1]], 2)   # [n, xyz, pair]
        qn = d.norm(2, dim=1)                                                                    # [n, pair]
        energy = (d.unsqueeze(3) * d.unsqueeze(2)).sum(1)                                        # [n, pair, pair]
        ne = (energy / (qn.unsqueeze(2) * qn.unsqueeze(1) + 1e-8)).reshape(-1, 9)
        m_cos = ((ne > delta_cos) | (ne < -delta_cos)).sum(1) > 3
        m_pad = (pw[:, 2, :] > self.delta_z).sum(1) == 3
        dd = delta_diff.view(-1, 1)
        near = ((d[:, 0, :].abs() < dd).sum(1) > 0) & ((d[:, 1, :].abs() < dd).sum(1) > 0) & ((d[:, 2, :].abs() < dd).sum(1) > 0)
        return m_pad & ~(near | m_cos)

    @staticmethod
    def _normals(tri):
        n = torch.cross(tri[:, :, 1] - tri[:, :, 0], tri[:, :, 2] - tri[:, :, 0], dim=1)
        nn_ = n.norm(2, dim=1, keepdim=True)
        return n / (nn_ + (nn_ == 0.0).float() * 0.01)

    def batched(self, pred_depth, gt_depth, t):
        """-> per-image loss [B] (float64), equal to vnl.py:119-165 evaluated image by image."""
        dev = pred_depth.device
        if FUSED_LOSS and pred_depth.is_cuda and t.n_tot > 0:
            # per-triplet part (cloud points, filter, normals, cosine term AND its three depth derivatives) in ONE kernel
            loss, valid = _VnlTriplets.apply(pred_depth.contiguous(), gt_depth.contiguous(), t, self.input_size, float(self.delta_z))
            valid = valid.bool()
            return self._trimmed_means(loss, valid, t, dev)
        pc_pred, pc_gt = self._cloud(pred_depth, t), self._cloud(gt_depth, t)
        is_plane = t.seg_is_plane[t.seg]                                                          # per triplet
        tri_pred = self._triplets_t(pc_pred, t)
        tri_gt = self._triplets(pc_gt, t.gid)
        # planes filter on the predicted cloud (delta_diff 0.005); the non-planar region on the GT cloud (0.1)
        tri_f = torch.where(is_plane.view(-1, 1, 1), tri_pred.detach(), tri_gt)
        delta = torch.where(is_plane, torch.full((), 0.005, dtype=tri_f.dtype, device=dev), torch.full((), 0.1, dtype=tri_f.dtype, device=dev))
        with torch.no_grad():
            valid = self._filter(tri_f, delta)
        # predicted normals; the non-planar branch zero-fixes its predicted triplets first (vnl.py:151: rows whose
        # z-coordinates of a point are exactly 0 are overwritten with 1e-4 -- indexed the reference's quirky way)
        zero_row = (tri_pred[:, 2, :] == 0) & (~is_plane).view(-1, 1)                             # [n, 3] -> rows of dim 1
        tri_p = torch.where(zero_row.unsqueeze(2), torch.full_like(tri_pred, 0.0001), tri_pred)
        dn = self._normals(tri_p)
        gn_np = self._normals(tri_gt)
        cos_plane = F.cosine_similarity(dn.double(), t.seg_normal[t.seg], dim=1).abs()
        cos_np = F.cosine_similarity(dn, gn_np, dim=1).abs().double()
        loss = 1 - torch.where(is_plane, cos_plane, cos_np)                                       # [n_tot] float64
        return self._trimmed_means(loss, valid, t, dev)

    @staticmethod
    def _trimmed_means(loss, valid, t, dev):
        """Per-triplet losses -> per-image loss [B] (vnl.py:106-117,133-165).  The value is computed by
        `_trimmed_means_autograd`'s operator chain without a graph; the result is linear in the kept losses, so its gradient is
        one coefficient per triplet (0 for dropped / invalid / NaN ones) scattered back through the sort permutation -- three
        launches instead of the ~30 (incl. a sort-based index_put and a reversed cumsum) autograd replays for the chain."""
        if not loss.requires_grad:
            return VNL_Loss._trimmed_means_autograd(loss, valid, t, dev)
        return _TrimmedMeans.apply(loss, valid, t)

    @staticmethod
    def _trimmed_means_autograd(loss, valid, t, dev, want_coef=False):
        """Segmented "sort ascending, drop the first 25 % of the valid ones, nansum / remaining"."""
        # segments are contiguous runs in triplet order: segment sums are differences of one prefix sum (no atomics)
        seg_end = torch.cat([t.seg_start[1:], t.seg_start.new_full((1,), t.n_tot)])

        def seg_total(v):
            cs = torch.cat([v.new_zeros(1), torch.cumsum(v, 0)])
            return cs[seg_end] - cs[t.seg_start]

        m = seg_total(valid.long())
        key = t.seg.double() * 4.0 + torch.where(valid, torch.nan_to_num(loss.detach(), nan=1.5), torch.full_like(loss, 2.0))
        order = torch.argsort(key)
        seg_s = t.seg[order]
        rank = torch.arange(t.n_tot, device=dev) - t.seg_start[seg_s]
        drop = m // 4
        keep = valid[order] & (rank >= drop[seg_s])
        contrib = torch.where(keep, torch.nan_to_num(loss[order], nan=0.0), torch.zeros_like(loss))
        seg_sum = seg_total(contrib)                                                               # sorted order keeps segments contiguous
        # per image: sum over planes (+ non-planar term unless it sampled nothing valid) / (N or N+1)
        np_ok = (~t.seg_is_plane) & (m > 0)
        use = t.seg_is_plane | np_ok
        den = torch.where(use, (m - drop).double(), torch.ones_like(seg_sum))                     # plane with no valid triplet: 0/0 -> NaN like the reference
        seg_loss = seg_sum / den
        img_sum = torch.zeros(t.B, device=dev, dtype=torch.float64).index_add_(0, t.seg_img, torch.where(use, seg_loss, torch.zeros_like(seg_loss)))
        extra = torch.zeros(t.B, device=dev, dtype=torch.float64).index_add_(0, t.seg_img, np_ok.double())
        out = img_sum / (t.N + extra)
        if want_coef:
            # d out[img] / d loss[order[j]]: kept, not NaN, in a segment that counts -> 1 / (segment denominator * image denominator)
            img_s = t.seg_img[seg_s]
            coef = torch.where(keep & ~torch.isnan(loss[order]) & use[seg_s], 1.0 / (den[seg_s] * (t.N + extra)[img_s]), torch.zeros_like(contrib))
            return out, order, coef, img_s
        return out

    def forward(self, pred_depth, gt_masks, gt_planes, gt_depth, k_matrix, select=True):
        """Single-image entry with the reference's signature (vnl.py:119); routes through the batched path."""
        assert select, "select=False is unused by the reference's loss"
        inst = [{"masks": gt_masks.cpu(), "plane_paras": gt_planes.cpu(), "k_matrix": k_matrix.cpu()}]
        t = self.prepare(inst, tuple(pred_depth.shape[-2:]), pred_depth.device)
        return self.batched(pred_depth.unsqueeze(0), gt_depth.unsqueeze(0), t)[0]


_WORKER = {}


def _worker_init(cfg_blob, rng_state):
    """Runs once in each prefetch worker process: the parent's active config and numpy RNG state (the virtual-normal draws
    use numpy's global generator, vnl.py:43-55, so the worker continues the parent's sequence)."""
    import pickle
    torch.set_num_threads(2)
    cfg.replace(pickle.loads(cfg_blob))
    np.random.set_state(rng_state)
    _WORKER["crit"] = PlaneRecNetLoss()


def _worker_targets(gt_instances, hw, mask_feat_size):
    return _WORKER["crit"].prepare_host(gt_instances, hw, mask_feat_size, False, pin=False)


def _worker_vnl(host, hw):
    return _WORKER["crit"].vnl.prepare_host(host, hw, pin=False)


def _worker_loop(conn, cfg_blob, rng_state, role, sharing):
    """Main of a prefetch worker process: one request at a time over the pipe, results go back in submission order."""
    import traceback
    import torch.multiprocessing as mp
    try:
        mp.set_sharing_strategy(sharing)
        _worker_init(cfg_blob, rng_state)
        fn = _worker_targets if role == "targets" else _worker_vnl
        conn.send("ready")
    except Exception:                                         # noqa: BLE001
        conn.send(("error", traceback.format_exc()))
        return
    while True:
        try:
            msg = conn.recv()
        except EOFError:
            return
        if msg is None:
            return
        try:
            conn.send(("ok", _pack_tree(fn(*msg))))
        except Exception:                                     # noqa: BLE001
            conn.send(("error", traceback.format_exc()))


def _pack_tree(o):
    """All tensors of a (nested dict / list) result as ONE byte tensor + a skeleton with (offset, dtype, shape) leaves: a
    tensor costs ~0.5 ms to hand over between processes whatever its size, a batch's targets are ~15 of them."""
    chunks, off = [], [0]

    def walk(v):
        if torch.is_tensor(v):
            b = v.contiguous().reshape(-1).view(torch.uint8)
            spec = ("__t__", off[0], v.dtype, tuple(v.shape))
            chunks.append(b)
            pad = (-b.numel()) % 16
            if pad:
                chunks.append(torch.zeros(pad, dtype=torch.uint8))
            off[0] += b.numel() + pad
            return spec
        if isinstance(v, dict):
            return {k: walk(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return type(v)(walk(x) for x in v)
        return v
    skel = walk(o)
    return skel, (torch.cat(chunks) if chunks else torch.zeros(0, dtype=torch.uint8))


def _unpack_tree(skel, blob):
    def walk(v):
        if isinstance(v, tuple) and len(v) == 4 and v[0] == "__t__":
            _, o, dt, shp = v
            n = 1
            for d in shp:
                n *= d
            return blob[o:o + n * torch.empty(0, dtype=dt).element_size()].view(dt).view(shp)
        if isinstance(v, dict):
            return {k: walk(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return type(v)(walk(x) for x in v)
        return v
    return walk(skel)


class _PipeWorker:
    """One worker process behind a pipe.  Unlike concurrent.futures.ProcessPoolExecutor there is NO helper thread in the
    trainer process: requests are sent and results received by the calling thread itself, at submit() / result() time.
    (The executor's manager thread unpickles results whenever they arrive -- in the middle of the forward pass -- and every
    such wake-up takes the GIL from the trainer: 22 -> 10 ms for the forward's enqueue work, tools/host_ops_profile.py.)"""

    def __init__(self, ctx, role, init, sharing):
        self.conn, child = ctx.Pipe()
        self.proc = ctx.Process(target=_worker_loop, args=(child,) + init + (role, sharing), daemon=True)
        self.proc.start()
        child.close()
        self.sent = self.received = 0
        self.results = {}
        self.lock = threading.Lock()                          # results are taken off the pipe by one thread at a time (trainer or receiver thread)
        self._check(self.conn.recv())

    @staticmethod
    def _check(r):
        if isinstance(r, tuple) and r and r[0] == "error":
            raise RuntimeError("prefetch worker failed:\n" + r[1])
        return r

    def submit(self, fn_unused, *args):
        self.conn.send(args)
        self.sent += 1
        return _PipeFuture(self, self.sent - 1)

    def result(self, index):
        with self.lock:
            while index not in self.results:
                r = self._check(self.conn.recv())
                skel, blob = r[1]
                if torch.cuda.is_available():
                    blob = _pin(blob)                         # page-locked staging (cannot cross the process boundary): one copy
                self.results[self.received] = _unpack_tree(skel, blob)
                self.received += 1
            return self.results.pop(index)

    def shutdown(self, wait=True, cancel_futures=True):
        try:
            self.conn.send(None)
        except (OSError, ValueError):
            pass
        self.proc.join(5 if wait else 0)
        if self.proc.is_alive():
            self.proc.terminate()
        self.conn.close()


class _PipeFuture:
    def __init__(self, worker, index):
        self.worker, self.index, self.value = worker, index, None
        self.lock = threading.Lock()

    def result(self):
        with self.lock:                                       # fetched once, whoever asks first
            if self.worker is not None:
                self.value, self.worker = self.worker.result(self.index), None
        return self.value


class TargetPrefetcher:
    """Runs the GT-only host work for the NEXT batch on workers while the GPU executes the current step (the role the
    reference gives to its DataLoader workers + the host part of its loss).  Two jobs per batch -- SOLOv2 targets (CPU torch
    ops) and virtual-normal triplet draws (numpy) -- run side by side; each kind is processed one batch at a time in
    submission order, so the numpy RNG stream of the sampling stays deterministic.

    workers="process" (default on a GPU box): two spawned worker PROCESSES.  The per-instance Python loops of the target
    assignment hold the GIL; on threads they slowed the trainer's own enqueue work from ~30 to ~65 ms per step (B=8), which
    made the step host-bound (tools/host_vs_gpu.py).  GT tensors should live in shared memory (DataLoader workers put
    them there; bench.py calls share_memory_()) so that submitting a batch sends handles, not pixels.
    workers="thread": in-process threads (no start-up cost; tests and short runs)."""

    def __init__(self, criterion, workers=None):
        from concurrent.futures import ThreadPoolExecutor
        self.criterion = criterion
        self.workers = workers or os.environ.get("PRN_PREFETCH_WORKERS") or ("process" if torch.cuda.is_available() else "thread")
        if self.workers == "process":
            try:
                self._start_processes()
            except Exception as e:                            # noqa: BLE001  (no /dev/shm, no semaphores, ...): same results from threads, slower trainer
                import sys
                print("TargetPrefetcher: worker processes unavailable (%s: %s); using threads" % (type(e).__name__, e), file=sys.stderr)
                self.workers = "thread"
        if self.workers != "process":
            self.pool_t = ThreadPoolExecutor(max_workers=1, thread_name_prefix="prn-targets")
            self.pool_v = ThreadPoolExecutor(max_workers=1, thread_name_prefix="prn-vnl")
        self.queue = collections.deque()                      # FIFO: submit() batches ahead of time, get() returns the oldest
        # Results are taken off the pipes (and copied to page-locked staging, ~30 MB per batch of 8) by a receiver thread as soon
        # as the workers deliver them -- both steps release the GIL -- instead of by the trainer inside get(): the trainer's enqueue
        # work per step drops from 42-43 to 35-37 ms against a 51-52 ms GPU step (tools/host_vs_gpu.py, H).  The step itself is
        # GPU-bound either way on a quiet host (bench.py 51.66 vs 51.68 ms); the margin is what a busy host eats into.
        # PRN_PREFETCH_EARLY=0: receive in line.
        self._early = None
        if self.workers == "process" and os.environ.get("PRN_PREFETCH_EARLY", "1") != "0":
            # (the current device is per-thread state: without the initializer the thread's page-locked allocations would go through
            # device 0 on every rank of a multi-GPU job)
            dev = torch.cuda.current_device() if torch.cuda.is_available() else None
            self._early = ThreadPoolExecutor(max_workers=1, thread_name_prefix="prn-recv",
                                             initializer=(lambda: torch.cuda.set_device(dev)) if dev is not None else None)

    @staticmethod
    def _receive(ft, fv):
        ft.result()
        if fv is not None:
            fv.result()

    def _start_processes(self):
        import pickle
        import sys
        import torch.multiprocessing as mp
        try:                                                   # batches and results travel through /dev/shm (~60 MB in flight at B=8)
            st = os.statvfs("/dev/shm")
            if st.f_bavail * st.f_frsize < (1 << 30):
                raise RuntimeError("/dev/shm has less than 1 GiB free")
        except FileNotFoundError:
            raise RuntimeError("no /dev/shm")
        ctx = mp.get_context("spawn")
        # "file_system" sharing: a shared-memory tensor travels as a file name.  (The default strategy passes file
        # descriptors, which the sender serves from a background thread -- one more GIL customer in the trainer.)
        sharing = os.environ.get("PRN_PREFETCH_SHARING", "file_system")
        if sharing in mp.get_all_sharing_strategies():
            mp.set_sharing_strategy(sharing)
        sharing = mp.get_sharing_strategy()
        init = (pickle.dumps(cfg), np.random.get_state())
        # The workers need this module only.  A spawned child normally re-imports the parent's __main__ script first
        # (a training script without an `if __name__ == "__main__"` guard would run again inside every worker): hide
        # the script from multiprocessing while the two workers start.
        main = sys.modules.get("__main__")
        saved = {a: getattr(main, a) for a in ("__file__", "__spec__") if hasattr(main, a)}
        try:
            if "__file__" in saved:
                del main.__file__
            if main is not None:
                main.__spec__ = None
            self.pool_t = _PipeWorker(ctx, "targets", init, sharing)
            self.pool_v = _PipeWorker(ctx, "vnl", init, sharing)
        finally:
            for a, v in saved.items():
                setattr(main, a, v)

    @property
    def pending(self):
        return self.queue[0] if self.queue else None

    def submit(self, gt_instances, hw, mask_feat_size=None):
        proc = self.workers == "process"
        if proc:
            gt_instances = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in g.items()} for g in gt_instances]
            ft = self.pool_t.submit(_worker_targets, gt_instances, hw, mask_feat_size)
        else:
            ft = self.pool_t.submit(self.criterion.prepare_host, gt_instances, hw, mask_feat_size, False)
        fv = None
        if cfg.use_plane_loss:
            host = [{k: g[k].cpu() for k in ("masks", "plane_paras", "k_matrix")} for g in gt_instances]
            fv = self.pool_v.submit(_worker_vnl, host, hw) if proc else self.pool_v.submit(self.criterion.vnl.prepare_host, host, hw)
        self.queue.append((ft, fv, self._early.submit(self._receive, ft, fv) if self._early is not None else None))

    def get(self, gt_depths, device, overlap=False):
        """Targets of the OLDEST submitted batch (blocks only if a worker has not finished yet).  Keeping two batches in
        flight hides the workers' latency (~40 ms per batch of 8 next to a ~60 ms step) completely."""
        ft, fv, early = self.queue.popleft()
        if early is not None:
            early.result()                                    # (re-raises a worker failure)
        h = ft.result()
        h["vnl"] = fv.result() if fv is not None else None
        if not overlap or not torch.cuda.is_available():
            return self.criterion.upload(h, gt_depths, device)
        # The uploads (~20 MB of indices / labels), the device-side sort of the triplet indices and the lava preparation go to
        # the weight-gradient side stream, which is idle during the forward pass; the loss waits for `ready`.  On the compute
        # stream they delayed the forward pass by 2.6 ms per step (tools/host_vs_gpu.py, FIXED_TARGETS=1: 58.0 vs 60.6 ms).
        main = torch.cuda.current_stream()
        side = ops._side_stream(torch.device(device), main)
        side.wait_stream(main)                               # gt_depths (and the previous step's readers of recycled memory)
        with torch.cuda.stream(side):
            t = self.criterion.upload(h, gt_depths, device)
            t.ready = torch.cuda.Event()
            t.ready.record()
        for obj in (t, t.vnl) if ops.GRAD_RECORD_STREAM else ():    # (not needed, and one marker packet on the main stream per tensor
            if obj is None:                                          # when the targets are released: see ops.GRAD_RECORD_STREAM)
                continue
            for name in obj.__slots__:
                v = getattr(obj, name, None)
                for x in (v if isinstance(v, (list, tuple)) else (v,)):
                    if torch.is_tensor(x) and x.is_cuda:
                        x.record_stream(main)               # allocated under the side stream, read (and released) under main
        return t

    def discard(self):
        """Drop every batch that was submitted but not fetched (end of an epoch, early exit from a loop)."""
        while self.queue:
            ft, fv, early = self.queue.popleft()
            if early is not None:
                early.result()
            ft.result()
            if fv is not None:
                fv.result()

    def close(self):
        wait = self.workers == "process"                     # (worker processes are joined: nothing left behind at exit)
        if self._early is not None:
            self._early.shutdown(wait=True, cancel_futures=True)
        self.pool_t.shutdown(wait=wait, cancel_futures=True)
        self.pool_v.shutdown(wait=wait, cancel_futures=True)

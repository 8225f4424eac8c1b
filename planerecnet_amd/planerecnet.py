"""PlaneRecNet model assembly on the HIP operators (reference: planerecnet.py).

Same constructor, module tree / state-dict keys, train-mode return `(mask_pred, cate_pred[4], kernel_pred[4],
depth_pred)`, eval-mode return `list[dict]`, and the weight-I/O helpers (`init_weights`, `load_weights`,
`save_weights`, `freeze_bn`).  nn.Conv2d / nn.BatchNorm2d / nn.GroupNorm objects are parameter containers;
arithmetic goes through planerecnet_amd.ops (HIP).  Differences in *how* (never in what):

  * ReflectionPad2d and nearest-x2 Upsample are folded into the conv operand gather (ops.IN_REFLECT / IN_UP2_REFLECT);
  * BatchNorm launches apply ReLU; GroupNorm launches apply ReLU;
  * the plane-prior branch (planerecnet.py:589-594) is evaluated only at the pixels the x0.25 bilinear resize reads:
    interpolate(conv1x1(sigmoid(K.M))) == conv1x1(avgpool2x2(sigmoid(K.M)[centre pixels])) because a 1x1 conv is
    linear per pixel, the resize weights sum to one and all inputs are detached.  The [B,3728,120,160] tensor
    (286 MB/image) is never formed; see DESIGN.md for the accounting of executed vs reference FLOPs.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, timer
from .backbone import can_fold, construct_backbone, conv_bn, folded_bn
from .config import cfg
from .fpn import FPN
from .funcs import bias_init_with_prob
from .nms import mask_nms, matrix_nms


_COORD = {}


def _coord_channels(feat):
    """(x, y) in [-1,1], channel order x then y (planerecnet.py:370-376,484-490).  Constant per (batch, size, device): built
    once (the reference rebuilds it with linspace / meshgrid / expand / cat in every forward)."""
    B, _, h, w = feat.shape
    key = (B, h, w, feat.device)
    c = _COORD.get(key)
    if c is None:
        xr = torch.linspace(-1, 1, w, device=feat.device)
        yr = torch.linspace(-1, 1, h, device=feat.device)
        y, x = torch.meshgrid(yr, xr, indexing="ij")
        c = _COORD[key] = torch.cat([x.expand(B, 1, h, w), y.expand(B, 1, h, w)], 1).contiguous()
    return c


def _coord_channels_resized(feat, size):
    """resize_bilinear(_coord_channels(feat), size), cached: the resize works channel by channel, so
    resize(cat([feat, coords])) == cat([resize(feat), resize(coords)]) bit for bit -- and the second part is a constant."""
    B, _, h, w = feat.shape
    key = (B, h, w, feat.device, int(size[0]), int(size[1]))
    c = _COORD.get(key)
    if c is None:
        with torch.no_grad():
            c = _COORD[key] = ops.resize_bilinear(_coord_channels(feat), size).contiguous()
    return c


def _conv_gn_relu(x, conv, gn):
    return ops.group_norm_relu(ops.conv2d(x, conv.weight, conv.bias, pad=conv.padding[0]), gn.weight, gn.bias, gn.num_groups, gn.eps)


class PlaneRecNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.device = torch.device(cfg.device)
        self.depth_decoder_indices = cfg.depth.selected_layers
        self.fpn_indices = cfg.fpn.selected_layers
        s = cfg.solov2
        self.num_classes, self.num_kernels, self.num_grids = cfg.num_classes, s.num_kernels, s.num_grids
        self.instance_in_features, self.instance_strides = s.instance_in_features, s.fpn_instance_strides
        self.instance_in_channels, self.instance_channels = cfg.fpn.num_features, s.instance_channels
        self.mask_in_features, self.mask_in_channels = s.masks_in_features, cfg.fpn.num_features
        self.mask_channels, self.num_masks = s.masks_channels, s.num_masks
        self.max_before_nms, self.score_threshold, self.update_threshold = s.nms_pre, s.score_thr, s.update_thr
        self.mask_threshold, self.max_per_img = s.mask_thr, s.top_k
        self.nms_kernel, self.nms_sigma, self.nms_type = s.nms_kernel, s.nms_sigma, s.nms_type

        self.backbone = construct_backbone(cfg.backbone)
        if cfg.freeze_bn:
            self.freeze_bn()
        src = self.backbone.channels
        self.fpn = FPN([src[i] for i in self.fpn_indices], start_level=cfg.fpn.start_level)
        self.depth_decoder = DepthDecoder_FPN()
        self.inst_head = SOLOv2InsHead(cfg, [cfg.fpn.num_features] * len(s.instance_in_features))
        self.mask_head = SOLOv2MaskHead(cfg, [cfg.fpn.num_features] * len(s.masks_in_features))

    def _refresh_dgrad_weights(self):
        """One launch that lays every conv weight out for its input-gradient GEMM (ops.FlippedWeights)."""
        fw = self.__dict__.get("_flipped")
        if fw is not None and any(m._merged()[0] is not d for m, d in self.__dict__.get("_flip_dcn", ())):
            for d in fw.data:                                  # a DCN block re-linked its merged storage (module.to(), .data assignment)
                ops._FLIPPED.pop(d.data_ptr(), None)
            fw = None
        if fw is None:
            from .dcn import DeformableConv2d
            dcns = [m for m in self.modules() if isinstance(m, DeformableConv2d)]
            dcn_w = {id(m.regular_conv.weight) for m in dcns}
            dcn_om = {id(c.weight) for m in dcns for c in (m.offset_conv, m.modulator_conv)}
            items = []
            for m in self.modules():
                if isinstance(m, nn.Conv2d) and m.weight.requires_grad and m.weight.is_cuda and id(m.weight) not in dcn_om:
                    M, C, KH, KW = m.weight.shape
                    items.append((m.weight, (M, C * KH * KW, 1, 1) if id(m.weight) in dcn_w else (M, C, KH, KW)))
            for m in dcns:
                # the offset + modulator conv is ONE 27-channel conv over the merged weight (dcn.DeformableConv2d._merged): its
                # input gradient reads the merged tensor's flipped layout; the offset parameter (its leading rows) carries the version
                if m.offset_conv.weight.requires_grad and m.offset_conv.weight.is_cuda:
                    w27 = m._merged()[0]
                    items.append((m.offset_conv.weight, tuple(w27.shape), w27, (m.modulator_conv.weight,)))
            self.__dict__["_flip_dcn"] = [(m, e[2]) for m, e in zip([m for m in dcns if m.offset_conv.weight.requires_grad and m.offset_conv.weight.is_cuda],
                                                                     [e for e in items if len(e) > 2])]
            fw = self.__dict__["_flipped"] = ops.FlippedWeights(items)
            self.__dict__["_flip_steps"] = 0
        # after two full steps: keep only the weights whose flipped layout was actually requested (the 3x3 layers on the
        # Winograd path use the transform-domain operand instead -- 60 % of the bytes); anything requested later is flipped
        # on demand by ops.flip_transpose
        self.__dict__["_flip_steps"] += 1
        if self.__dict__["_flip_steps"] == 3:
            used = [(w, shp, d, gd[1:]) for (w, shp), d, gd in zip(fw.weights, fw.data, fw.guards) if d.data_ptr() in ops._FLIP_USED]
            if used and len(used) < len(fw.weights):
                for d in fw.data:
                    ops._FLIPPED.pop(d.data_ptr(), None)
                fw = self.__dict__["_flipped"] = ops.FlippedWeights(used)
        fw.refresh()
        # transform-domain operands of the 3x3 weights the Winograd path asked for in earlier steps (ops.WinogradWeights)
        mine = self.__dict__.get("_param_ids")
        if mine is None:
            mine = self.__dict__["_param_ids"] = {id(p) for p in self.parameters()}
        seen = [w for w in ops.winograd_seen() if id(w) in mine]
        ww = self.__dict__.get("_wino")
        if seen and (ww is None or len(ww.weights) != len(seen)):
            ww = self.__dict__["_wino"] = ops.WinogradWeights(seen)
        if ww is not None:
            ww.refresh()
        ops.split_refresh_all()          # weight images of the bf16-split GEMM launches (parameters, flipped and transform-domain layouts): one launch
        # every derived layout these three keep is now current for the weights' versions: the block calls re-use their parameter tables (blocks.py)
        ops.vouch_refreshed([t for gd in fw.guards for t in gd])
        if ww is not None:
            ops.vouch_refreshed(ww.weights)

    def forward(self, x):
        if x.is_cuda:
            # which plain GEMMs take the bf16-split kernel is a board-level trade that differs between training and inference
            # (ops.split_gemm_policy, csrc/prn_gemm_split.hip)
            ops.split_gemm_policy("train" if self.training else "eval")
        if self.training and torch.is_grad_enabled() and x.is_cuda:
            self._refresh_dgrad_weights()
        with timer.env("backbone"):
            enc = self.backbone(x)
        with timer.env("fpn"):
            # the decoder reads the same backbone features: it gets them back from the FPN's forked lateral convs
            feats, fenc = self.fpn([enc[i] for i in self.fpn_indices], return_inputs=True)
            enc = list(enc)
            for i, f in zip(self.fpn_indices, fenc):
                enc[i] = f
        with timer.env("instance head"):
            # (the finest map also feeds the mask head: it gets it back from the forked resize, whose backward kernel sums the two gradients)
            ins_feats, feats[0] = self.split_feats([feats[f] for f in range(len(self.instance_in_features))], fork=True)
            cate_pred, kernel_pred = self.inst_head(ins_feats)            # five levels on five streams (ops.run_branches)
        with timer.env("mask head"):
            mask_pred = self.mask_head([feats[f] for f in range(len(self.mask_in_features))])
        with timer.env("depth_decoder"):
            depth_pred = self.depth_decoder([enc[i] for i in self.depth_decoder_indices], mask_pred, kernel_pred)
        with timer.env("Inferencing"):
            if self.training:
                return mask_pred, cate_pred, kernel_pred, depth_pred
            from .metrics import category_scores
            # sigmoid + point NMS (planerecnet.py:113) + level concatenation, one launch per level (include/prn.h: prn_sigmoid_point_nms)
            return self.inference(mask_pred, category_scores(cate_pred), kernel_pred, depth_pred, x)

    @staticmethod
    def split_feats(feats, fork=False):
        h, w = feats[0].shape[2:]
        if fork:                                            # -> (features, the first input handed back: ops.resize_bilinear_fork)
            r, f0 = ops.resize_bilinear_fork(feats[0], (int(h * 0.5), int(w * 0.5)))
            return (r, feats[1], feats[2], feats[3]), f0
        return (ops.resize_bilinear(feats[0], (int(h * 0.5), int(w * 0.5))), feats[1], feats[2], feats[3])

    # ---- weight I/O (planerecnet.py:121-153)
    def _flush_bn_counters(self):
        """Write the host-side `num_batches_tracked` counts (ops.batch_norm_module) into the buffers."""
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                ops.flush_batch_count(m)

    def state_dict(self, *args, **kwargs):
        self._flush_bn_counters()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        for m in self.modules():
            m.__dict__.pop("_prn_nbt_pending", None)
        return super().load_state_dict(*args, **kwargs)

    def save_weights(self, path):
        torch.save(self.state_dict(), path)

    def load_weights(self, path):
        self.load_state_dict(torch.load(path, map_location="cpu"))

    def init_weights(self, backbone_path):
        self.backbone.init_backbone(backbone_path)
        self.init_head_weights()

    def init_head_weights(self):
        """Xavier-uniform for every conv outside the pretrained backbone; focal prior on inst_head.cate_pred bias."""
        for name, module in self.named_modules():
            if isinstance(module, nn.Conv2d) and module not in self.backbone.backbone_modules:
                nn.init.xavier_uniform_(module.weight.data)
                if module.bias is not None:
                    if "inst_head" in name and "cate_pred" in name:
                        module.bias.data.fill_(bias_init_with_prob(cfg.solov2.focal_loss_init_pi))
                    else:
                        module.bias.data.fill_(0)

    def freeze_bn(self, enable=False):
        for module in self.modules():
            if isinstance(module, nn.BatchNorm2d):
                module.train() if enable else module.eval()
                module.weight.requires_grad = enable
                module.bias.requires_grad = enable

    # ---- post-process (planerecnet.py:155-289)
    def inference(self, pred_masks, pred_cates, pred_kernels, pred_depths, batched_images, ori_size=None):
        """Reference planerecnet.py:155-289 (`inference` + `inference_single_image`) for the whole batch.  The steps that do not mix
        images -- candidate selection, dynamic mask convolution, mask statistics, the small-mask filter -- run once over the candidates of
        ALL images (two device->host synchronisations per step for the batch instead of one or more per image); sorting, matrix NMS and
        the final selection run per image on slices.  Every per-candidate value is computed by an operation that works row by row with a
        fixed summation order, so an image's result does not depend on what else is in the batch."""
        from .metrics import mask_boxes, mask_stats
        assert torch.is_tensor(pred_cates) or len(pred_cates) == len(pred_kernels)
        B = pred_masks.shape[0]
        ori_size = tuple(batched_images[0].shape[1:]) if ori_size is None else tuple(ori_size)
        depth = ops.resize_bilinear(pred_depths.detach(), ori_size).detach()
        results = [{"pred_masks": None, "pred_boxes": None, "pred_classes": None, "pred_scores": None, "pred_depth": depth[b:b + 1]} for b in range(B)]
        cate = pred_cates if torch.is_tensor(pred_cates) else torch.cat([c.detach().reshape(B, -1, self.num_classes) for c in pred_cates], 1)   # [B, cells, classes]
        kern = torch.cat([k.detach().permute(0, 2, 3, 1).reshape(B, -1, self.num_kernels) for k in pred_kernels], 1)   # [B, cells, E]
        pred_masks = pred_masks.detach()
        # candidates: (image, cell, class) with a category score above the threshold, image-major like the per-image nonzero
        nz = (cate > self.score_threshold).nonzero(as_tuple=False)
        if nz.shape[0] == 0:
            return results
        img, cell, labels = nz[:, 0], nz[:, 1], nz[:, 2]
        n_img = self._per_image_counts(img, B)
        scores = cate[img, cell, labels]
        kernels = kern[img, cell]
        strides = self._cell_strides(kernels).index_select(0, cell)
        # dynamic conv: per image one 1x1 implicit GEMM over its mask features (candidates of an image are contiguous)
        segs, o = [], 0
        for b in range(B):
            if n_img[b]:
                k = kernels[o:o + n_img[b]]
                segs.append(ops.conv2d(pred_masks[b:b + 1], k.reshape(n_img[b], -1, 1, 1), epilogue=ops.EPI_SIGMOID).squeeze(0))
                o += n_img[b]
        seg = segs[0] if len(segs) == 1 else torch.cat(segs, 0)                                                         # [N, h, w]
        sum_masks, msum = mask_stats(seg, self.mask_threshold)          # pixels above the mask threshold and the sum of their values
        kept = (sum_masks > strides).nonzero(as_tuple=False).flatten()
        if kept.shape[0] == 0:
            return results
        img = img.index_select(0, kept)
        n_img = self._per_image_counts(img, B)
        seg, sum_masks = seg.index_select(0, kept), sum_masks.index_select(0, kept)
        labels = labels.index_select(0, kept)
        scores = scores.index_select(0, kept) * (msum.index_select(0, kept) / sum_masks)       # mask-quality ("maskness") weighting
        seg_masks = seg > self.mask_threshold
        boxes, o = [], 0
        for b in range(B):
            if n_img[b]:
                sl = slice(o, o + n_img[b])
                o += n_img[b]
                boxes.append(self._finish_image(results[b], seg[sl], seg_masks[sl], sum_masks[sl], scores[sl], labels[sl], ori_size))
        # boxes of all images in one transfer; returned on the CPU like the reference's default-device torch.zeros (quirk Q11)
        boxes = [(r, bx) for r, bx in boxes if bx is not None]
        if boxes:
            host = torch.cat([bx for _, bx in boxes], 0).cpu().split([bx.shape[0] for _, bx in boxes])
            for (r, _), h in zip(boxes, host):
                r["pred_boxes"] = h
        return results

    def _finish_image(self, result, seg_preds, seg_masks, sum_masks, cate_scores, cate_labels, ori_size):
        """Sort, NMS, final selection, full-size masks and (device) boxes of one image's candidates.  -> (result, boxes | None)"""
        from .metrics import mask_boxes
        order = torch.argsort(cate_scores, descending=True)[: self.max_before_nms]
        seg_masks, seg_preds, sum_masks = seg_masks.index_select(0, order), seg_preds.index_select(0, order), sum_masks.index_select(0, order)
        cate_scores, cate_labels = cate_scores.index_select(0, order), cate_labels.index_select(0, order)
        if self.nms_type == "matrix":
            cate_scores = matrix_nms(cate_labels, seg_masks, sum_masks, cate_scores, sigma=self.nms_sigma, kernel=self.nms_kernel)
            keep = cate_scores >= self.update_threshold
        elif self.nms_type == "mask":
            keep = mask_nms(cate_labels, seg_masks, sum_masks, cate_scores, nms_thr=self.mask_threshold)
        else:
            raise NotImplementedError
        kept = keep.nonzero(as_tuple=False).flatten()
        if kept.shape[0] == 0:
            return result, None
        # (selection and the final ordering in one gather: positions of the kept detections, by descending score)
        order = kept.index_select(0, torch.argsort(cate_scores.index_select(0, kept), descending=True)[: self.max_per_img])
        seg_preds, cate_scores, cate_labels = seg_preds.index_select(0, order), cate_scores.index_select(0, order), cate_labels.index_select(0, order)
        seg_masks = ops.resize_bilinear(seg_preds.unsqueeze(0), ori_size).squeeze(0) > self.mask_threshold
        result["pred_scores"], result["pred_classes"], result["pred_masks"] = cate_scores, cate_labels, seg_masks
        # tight boxes of the masks in one launch (the reference loops over instances with torch.where, planerecnet.py:282-287)
        return result, mask_boxes(seg_masks)

    def _per_image_counts(self, img, B):
        """How many entries of the sorted image-index vector belong to each image (one synchronisation; torch.bincount adds one of its own
        to find the largest index)."""
        key = (B, str(img.device))
        cache = self.__dict__.setdefault("_image_ids", {})
        if key not in cache:
            cache[key] = torch.arange(B, device=img.device).unsqueeze(1)
        return (img.unsqueeze(0) == cache[key]).sum(1).tolist()

    def _cell_strides(self, like):
        """Instance stride of every grid cell (levels concatenated), built once per device."""
        key = str(like.device)
        cache = self.__dict__.setdefault("_cell_stride_cache", {})
        if key not in cache:
            cache[key] = torch.cat([torch.full((g * g,), float(s), dtype=like.dtype, device=like.device) for g, s in zip(self.num_grids, self.instance_strides)])
        return cache[key]

    def inference_single_image(self, seg_preds, cate_preds, kernel_preds, depth_pred, ori_size):
        """One image through the post-process (the reference's entry point of the same name, planerecnet.py:199-289):
        seg_preds [1,E,h,w], cate_preds [cells, classes], kernel_preds [cells, E], depth_pred [1,1,H',W']."""
        levels = [g * g for g in self.num_grids]
        cates = [c.reshape(1, g, g, self.num_classes) for c, g in zip(cate_preds.split(levels, 0), self.num_grids)]
        kerns = [k.reshape(1, g, g, self.num_kernels).permute(0, 3, 1, 2) for k, g in zip(kernel_preds.split(levels, 0), self.num_grids)]
        return self.inference(seg_preds, cates, kerns, depth_pred, None, ori_size)[0]


RAGGED_HEADS = bool(int(os.environ.get("PRN_RAGGED_HEADS", "1")))
BN_CAT = bool(int(os.environ.get("PRN_BN_CAT", "1")))      # 0: BatchNorm outputs concatenated by torch.cat (cross-check)
PLANE_PRIOR_BLOCK = bool(int(os.environ.get("PRN_PLANE_PRIOR_BLOCK", "1")))      # 0: the operator-by-operator plane prior (cross-check)


class SOLOv2InsHead(nn.Module):
    """Category + kernel towers shared by all levels (planerecnet.py:292-391)."""

    def __init__(self, cfg, in_channels):
        super().__init__()
        s = cfg.solov2
        self.num_classes, self.num_kernels, self.num_grids = cfg.num_classes, s.num_kernels, s.num_grids
        self.instance_in_features, self.instance_strides = s.instance_in_features, s.fpn_instance_strides
        self.instance_in_channels, self.instance_channels = cfg.fpn.num_features, s.instance_channels
        self.num_levels = len(self.instance_in_features)
        assert self.num_levels == len(self.instance_strides) and len(set(in_channels)) == 1
        norm = None if s.norm == "none" else s.norm
        if norm != "GN" or s.use_dcn_in_instance:
            raise NotImplementedError("hot path: GroupNorm towers without DCN")
        for head, use_coord in (("cate", False), ("kernel", s.use_coord_conv)):
            tower = []
            for i in range(s.num_instance_convs):
                cin = (self.instance_in_channels + (2 if use_coord else 0)) if i == 0 else self.instance_channels
                tower += [nn.Conv2d(cin, self.instance_channels, 3, padding=1, bias=False), nn.GroupNorm(32, self.instance_channels),
                          nn.ReLU(inplace=True)]
            self.add_module(head + "_tower", nn.Sequential(*tower))
        self.cate_pred = nn.Conv2d(self.instance_channels, self.num_classes, 3, padding=1)
        self.kernel_pred = nn.Conv2d(self.instance_channels, self.num_kernels, 3, padding=1)

    @staticmethod
    def _tower(tower, x):
        mods = list(tower)
        for i in range(0, len(mods), 3):
            x = _conv_gn_relu(x, mods[i], mods[i + 1])
        return x

    def _level(self, idx, feat):
        g = self.num_grids[idx]
        cf = ops.resize_bilinear(feat, (g, g))               # (the coordinate channels are resized once: _coord_channels_resized)
        kf = torch.cat([cf, _coord_channels_resized(feat, (g, g))], 1)
        kf = self._tower(self.kernel_tower, kf)
        kp = ops.conv2d(kf, self.kernel_pred.weight, self.kernel_pred.bias, pad=1)
        cf = self._tower(self.cate_tower, cf)
        return ops.conv2d(cf, self.cate_pred.weight, self.cate_pred.bias, pad=1), kp

    def branches(self, features):
        """One closure per level (independent chains; see ops.run_branches)."""
        return [lambda i=i, f=f: self._level(i, f) for i, f in enumerate(features)]

    @staticmethod
    def gather(outs):
        return [o[0] for o in outs], [o[1] for o in outs]

    def _ragged(self, features):
        """All levels through the shared towers as ONE ragged batch per layer (ops.ragged_conv2d): the five grids
        (40^2 .. 12^2 cells) give one GEMM over 3872 cells per image instead of five small ones, and the weight gradients
        come out of one launch instead of five launches plus four accumulation kernels per parameter."""
        B = features[0].shape[0]
        kfs, cfs = [], []
        for idx, feat in enumerate(features):
            g = self.num_grids[idx]
            # resize(cat([feat, coords])) == cat([resize(feat), resize(coords)]): the 256 feature channels are not copied at the
            # level's own resolution just to append two constant channels (23 M elements per step), and the gradient of the
            # resize arrives as a whole tensor instead of a channel slice that needs a copy
            cfs.append(ops.resize_bilinear(feat, (g, g)))
            kfs.append(torch.cat([cfs[-1], _coord_channels_resized(feat, (g, g))], 1))
        rs = self.__dict__.setdefault("_rs", {}).get(B)
        if rs is None:
            rs = self._rs[B] = ops.RaggedShape(B, [(g, g) for g in self.num_grids])
        if not rs.supported():
            return None
        kp, cp = rs.pack(kfs), rs.pack(cfs)

        def tower(t, x):
            mods = list(t)
            for i in range(0, len(mods), 3):
                x = ops.ragged_group_norm_relu(ops.ragged_conv2d(x, mods[i].weight, mods[i].bias, rs), mods[i + 1].weight, mods[i + 1].bias,
                                               mods[i + 1].num_groups, mods[i + 1].eps, rs)
            return x
        kp = ops.ragged_conv2d(tower(self.kernel_tower, kp), self.kernel_pred.weight, self.kernel_pred.bias, rs)
        cp = ops.ragged_conv2d(tower(self.cate_tower, cp), self.cate_pred.weight, self.cate_pred.bias, rs)
        return rs.unpack(cp, self.num_classes), rs.unpack(kp, self.num_kernels)

    def forward(self, features):
        if RAGGED_HEADS and features[0].is_cuda:
            out = self._ragged(features)
            if out is not None:
                return out
        return self.gather(ops.run_branches(self.branches(features)))


class SOLOv2MaskHead(nn.Module):
    """Unified mask feature: per-level conv-GN-ReLU(-x2 up) chains summed at 1/4 scale, then 1x1-GN-ReLU
    (planerecnet.py:394-496)."""

    def __init__(self, cfg, input_shape):
        super().__init__()
        s = cfg.solov2
        self.num_masks, self.mask_in_features = s.num_masks, s.masks_in_features
        self.mask_in_channels, self.mask_channels = cfg.fpn.num_features, s.masks_channels
        self.num_levels = len(input_shape)
        assert self.num_levels == len(self.mask_in_features)
        if (None if s.norm == "none" else s.norm) != "GN":
            raise NotImplementedError("hot path: GroupNorm mask head")
        self.convs_all_levels = nn.ModuleList()
        for i in range(self.num_levels):
            level = nn.Sequential()
            for j in range(max(i, 1)):
                cin = (self.mask_in_channels + (2 if i == 3 else 0)) if j == 0 else self.mask_channels
                level.add_module("conv" + str(j), nn.Sequential(nn.Conv2d(cin, self.mask_channels, 3, padding=1, bias=False),
                                                                nn.GroupNorm(32, self.mask_channels), nn.ReLU(inplace=False)))
                if i > 0:
                    level.add_module("upsample" + str(j), nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False))
            self.convs_all_levels.append(level)
        self.conv_pred = nn.Sequential(nn.Conv2d(self.mask_channels, self.num_masks, 1, bias=False), nn.GroupNorm(32, self.num_masks),
                                       nn.ReLU(inplace=True))

    def _level(self, i, x, acc=None):
        """Level i's chain; `acc` (the sum of the levels before it) is added by the chain's last upsample launch."""
        n = max(i, 1)
        for j in range(n):
            blk = getattr(self.convs_all_levels[i], "conv" + str(j))
            x = _conv_gn_relu(x, blk[0], blk[1])
            if i > 0:
                x = ops.resize_bilinear(x, (2 * x.shape[2], 2 * x.shape[3]), acc if j == n - 1 else None)
        return x

    def _branch(self, i, f, acc=None):
        if i == 3:
            f = torch.cat([f, _coord_channels(f)], 1)
        return self._level(i, f, acc)

    def branches(self, features):
        assert len(features) == self.num_levels
        return [lambda i=i, f=f: self._branch(i, f) for i, f in enumerate(features)]

    def gather(self, levels):
        acc = levels[0]
        for l in levels[1:]:
            acc = acc + l
        return _conv_gn_relu(acc, self.conv_pred[0], self.conv_pred[1])

    def forward(self, features):
        # (sequential on purpose: putting these four chains, or the decoder's lateral branches, on streams of their own
        # next to the instance head's measured no gain -- 85.7 off / 84.9 instance head only / 85.7 with these too)
        # feature_add_all_level (planerecnet.py:431-441) as a running sum carried through the levels: level i > 0 ends in an
        # upsample, whose launch adds the sum so far (three full-map add passes fewer each way)
        acc = None
        for i, f in enumerate(features):
            acc = self._branch(i, f, acc)
        return _conv_gn_relu(acc, self.conv_pred[0], self.conv_pred[1])


class DepthDecoder_FPN(nn.Module):
    """Plane-prior gated depth decoder (planerecnet.py:499-607)."""

    def __init__(self):
        super().__init__()
        self.num_output_channels = 1
        self.num_kernels = cfg.solov2.num_kernels
        self.channels_kernels_flatten = sum(g * g for g in cfg.solov2.num_grids)
        for i, c in enumerate((2048, 1024, 512, 256)):
            setattr(self, "latlayer%d" % (i + 1), nn.Conv2d(c, 256, 1))

        def block(cin, cout, up):
            mods = ([nn.Upsample(scale_factor=2, mode="nearest")] if up else []) + [
                nn.ReflectionPad2d(1), nn.Conv2d(cin, cout, 3, padding=0), nn.BatchNorm2d(cout, eps=0.001, momentum=0.01),
                nn.ReLU(inplace=True)]
            return nn.Sequential(*mods)

        for i, co in enumerate((256, 128, 128, 128)):
            setattr(self, "conv%d" % (i + 1), block(256, co, False))
        for i, co in enumerate((256, 128, 128, 64)):
            setattr(self, "deconv%d" % (i + 1), block(256, co, True))
        self.depth_pred = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(64, 1, 3, padding=0), nn.Softplus())
        self.conv1x1 = nn.Sequential(nn.Conv2d(self.channels_kernels_flatten, 256, 1))
        self.refine_conv = block(512, 128, False)

    @staticmethod
    def _cbr(seq, x, defer_bn=False):
        """[Upsample] -> ReflectionPad -> Conv -> BN -> ReLU as one gathered conv + one BN launch.
        defer_bn: return (conv output, BatchNorm module) -- the caller normalises two such outputs into one buffer (_bn_cat)."""
        mods = list(seq)
        up = isinstance(mods[0], nn.Upsample)
        conv, bn = mods[2 if up else 1], mods[3 if up else 2]
        if can_fold(bn):                                  # inference: BatchNorm folded into the conv, ReLU in its epilogue
            if not up:
                return conv_bn(x, conv, bn, 1, 1, relu=True, in_mode=ops.IN_REFLECT)
            w, b = folded_bn(conv.weight, conv.bias, bn)
            return ops.conv_up2_inference(x, w, b, relu=True)
        y = ops.conv2d(x, conv.weight, conv.bias, pad=1, in_mode=ops.IN_UP2_REFLECT if up else ops.IN_REFLECT)
        if defer_bn:
            return y, bn
        return ops.batch_norm_module(bn, y, None, True)

    def _defer_bn(self, x):
        """Training-mode BatchNorm layers whose outputs are concatenated write straight into the concatenated buffer (ops.batch_norm_relu_cat)."""
        return BN_CAT and x.is_cuda and all(seq[-2].training for seq in (self.conv2, self.conv3, self.conv4, self.refine_conv, self.deconv2, self.deconv3))

    @staticmethod
    def _bn_cat(a, b):
        return ops.batch_norm_relu_cat(a[1], a[0], b[1], b[0]) if isinstance(a, tuple) else torch.cat([a, b], 1)

    def _centre_index(self, n, device):
        """Indices {4k+1, 4k+2}: the two centre samples of every 4-block (what the x0.25 bilinear resize reads). Cached per
        (size, device) so that the forward stays free of host->device copies (HIP-graph capturable)."""
        key = (n, str(device))
        cache = self.__dict__.setdefault("_centre_cache", {})
        if key not in cache:
            cache[key] = (torch.arange(n // 4)[:, None] * 4 + torch.tensor([1, 2])).flatten().to(device)
        return cache[key]

    def plane_prior(self, seg_preds, kernel_preds):
        B = seg_preds.shape[0]
        h, w = seg_preds.shape[2:]
        if h % 4 or w % 4:
            raise NotImplementedError("plane prior expects a mask feature size divisible by 4")
        with torch.no_grad():
            flat = torch.cat([k.permute(0, 2, 3, 1).reshape(B, -1, self.num_kernels) for k in kernel_preds], 1)   # [B, 3728, E]
        c = self.conv1x1[0]
        if PLANE_PRIOR_BLOCK and seg_preds.is_cuda:
            # centre samples -> per-image dynamic conv + sigmoid -> 2x2 mean -> conv1x1, one C call (include/prn.h: prn_plane_prior_fwd)
            return ops.plane_prior(seg_preds.detach(), flat, c.weight, c.bias)
        with torch.no_grad():
            hi, wi = self._centre_index(h, seg_preds.device), self._centre_index(w, seg_preds.device)
            centre = seg_preds.detach()[:, :, hi][:, :, :, wi].contiguous()                                            # [B,E,h/2,w/2]
            sig = torch.cat([ops.conv2d(centre[b:b + 1], flat[b].reshape(-1, self.num_kernels, 1, 1).contiguous(),
                                        epilogue=ops.EPI_SIGMOID) for b in range(B)], 0)                                # [B,3728,h/2,w/2]
            pooled = ops.resize_bilinear(sig, (h // 4, w // 4))                                                         # exact 2x2 mean
        return ops.conv2d(pooled, c.weight, c.bias)

    def branches(self, feature_maps):
        """The parts of the decoder that only read backbone features: the deepest chain up to deconv1 and the three
        lateral -> conv branches."""
        c2, c3, c4, c5 = feature_maps
        lat = lambda m, f: ops.conv2d(f, m.weight, m.bias)
        d = self._defer_bn(c2)
        return [lambda: self._cbr(self.deconv1, self._cbr(self.conv1, lat(self.latlayer1, c5))),
                lambda: self._cbr(self.conv2, lat(self.latlayer2, c4), d),
                lambda: self._cbr(self.conv3, lat(self.latlayer3, c3), d),
                lambda: self._cbr(self.conv4, lat(self.latlayer4, c2), d)]

    def forward(self, feature_maps, seg_preds, kernel_preds):
        prior = self.plane_prior(seg_preds, kernel_preds)
        x, l2, l3, l4 = [b() for b in self.branches(feature_maps)]
        d = isinstance(l2, tuple)                            # training: (conv output, BatchNorm) pairs, normalised into the concatenated buffer
        x = self._cbr(self.refine_conv, torch.cat([x, x * prior], 1), d)
        x = self._cbr(self.deconv2, self._bn_cat(l2, x), d)
        x = self._cbr(self.deconv3, self._bn_cat(l3, x), d)
        x = self._cbr(self.deconv4, self._bn_cat(l4, x))
        c = self.depth_pred[1]
        return F.softplus(ops.conv2d(x, c.weight, c.bias, pad=1, in_mode=ops.IN_REFLECT))

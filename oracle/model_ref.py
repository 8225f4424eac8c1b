"""TEST INFRASTRUCTURE -- CPU restatement (plain PyTorch fp32/fp64) of the PlaneRecNet forward.

A *functional* restatement: `forward(sd, x, arch, training)` evaluates the network from a plain
state dict `sd` whose keys are the reference's, so weights can be shared in memory
with either the real reference (tests/golden/make_golden.py, this container only) or the HIP
product model (tests on the GPU box).  Each function cites the reference lines it follows.
It is pinned against the shim-imported reference by tests/golden/make_golden.py, which also emits
the golden fixtures checked in tests/test_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from collections import namedtuple

import torch
import torch.nn.functional as F

from oracle.dcn_ref import deform_conv2d_ref

Arch = namedtuple("Arch", "layers num_grids num_kernels num_classes")
ARCH = {
    # data/config.py:232,248,505-509 ; num_classes config.py:413
    "PlaneRecNet_101_config": Arch((3, 4, 23, 3), (40, 36, 24, 16), 128, 2),
    "PlaneRecNet_50_config": Arch((3, 4, 6, 3), (40, 36, 24, 16), 128, 2),
}


def _bn(sd, p, x, training, eps, momentum, update_stats):
    # nn.BatchNorm2d semantics (backbone.py:24,44,48,102: eps 1e-5 mom 0.1; planerecnet.py:518..582: 1e-3 / 0.01)
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    if training and not update_stats:
        rm, rv = rm.clone(), rv.clone()
    return F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training, momentum, eps)


# How the 3x3 / stride-1 / pad-1 convolutions are evaluated.  None (always, except inside the golden generators): F.conv2d, the reference's arithmetic.
# "winograd": F(4x4, 3x3) in the tensors' dtype for exactly the layers the HIP product runs on its Winograd path (planerecnet_amd.ops.winograd_ok:
# W % 4 == 0, H >= 8, >= 64 channels in and out, >= 128 tiles in the batch; the instance head's towers as one ragged batch) -- autograd then differentiates
# THROUGH the transforms, as the product's input- and weight-gradient launches do.  tests/golden/make_golden_r101.py runs the fp32 oracle once this way to
# measure, per parameter, how far an fp32 implementation of that ALGORITHM lands from the fp64 gradient (transform coefficients up to 8 cost ~2 digits of
# the forward result; GroupNorm's backward turns that into 1e-3 .. 1e-2 on a few ill-conditioned tower parameters): the yardstick the GPU test holds the
# Winograd build to, instead of a hand-kept allowance list.
CONV3X3 = None
WINOGRAD_MIN_TILES = 128

_BT = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]]
_G = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]
_AT = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]


def winograd_conv3x3(xp, w):
    """y = conv3x3(x) for an input xp [B,C,H+2,W+2] that already carries its one-pixel border (zeros or reflection): per 4x4 output tile
    Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A  (Lavin & Gray, F(4x4, 3x3)), every step in xp's dtype."""
    B, C, Hp, Wp = xp.shape
    H, W = Hp - 2, Wp - 2
    th, tw = -(-H // 4), -(-W // 4)
    xp = F.pad(xp, (0, 4 * tw - W, 0, 4 * th - H))
    bt, g, at = (torch.tensor(m, dtype=xp.dtype) for m in (_BT, _G, _AT))
    d = xp.unfold(2, 6, 4).unfold(3, 6, 4)                                  # [B, C, th, tw, 6, 6]
    V = torch.einsum("ij,bcthjk,lk->bcthil", bt, d, bt)
    U = torch.einsum("ij,mcjk,lk->mcil", g, w, g)
    Mt = torch.einsum("mcil,bcthil->bmthil", U, V)
    Y = torch.einsum("ij,bmthjk,lk->bmthil", at, Mt, at)                    # [B, M, th, tw, 4, 4]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], 4 * th, 4 * tw)[:, :, :H, :W]


def _on_winograd_path(B, C, H, W, M, ragged):
    return W % 4 == 0 and H >= 8 and C >= 64 and M >= 64 and (ragged or B * ((H + 3) // 4) * (W // 4) >= WINOGRAD_MIN_TILES)


def _conv(sd, p, x, stride=1, padding=0, prepadded=False, ragged=False):
    """prepadded: x carries a reflected one-pixel border (planerecnet.py's ReflectionPad2d + Conv2d(padding=0)); ragged: one of the instance head's
    grid levels (the product runs all levels of a tower layer as one batch).  Both only matter under CONV3X3 = "winograd"."""
    w = sd[p + ".weight"]
    if CONV3X3 == "winograd" and tuple(w.shape[2:]) == (3, 3) and stride == 1 and (padding == 1 or prepadded):
        H, W = (x.shape[2] - 2, x.shape[3] - 2) if prepadded else x.shape[2:]
        if _on_winograd_path(x.shape[0], x.shape[1], H, W, w.shape[0], ragged):
            y = winograd_conv3x3(x if prepadded else F.pad(x, (1, 1, 1, 1)), w)
            b = sd.get(p + ".bias")
            return y if b is None else y + b.view(1, -1, 1, 1)
    return F.conv2d(x, w, sd.get(p + ".bias"), stride=stride, padding=padding)


def dcn_block(sd, p, x, stride):
    """models/dcn.py:52-67"""
    h, w = x.shape[2:]
    max_offset = max(h, w) / 4.0
    offset = _conv(sd, p + ".offset_conv", x, stride, 1).clamp(-max_offset, max_offset)
    modulator = 2.0 * torch.sigmoid(_conv(sd, p + ".modulator_conv", x, stride, 1))
    return deform_conv2d_ref(x, offset, modulator, sd[p + ".regular_conv.weight"],
                             sd.get(p + ".regular_conv.bias"), stride, 1)


def bottleneck(sd, p, x, stride, training, upd):
    """models/backbone.py:53-73"""
    bn = lambda q, t: _bn(sd, q, t, training, 1e-5, 0.1, upd)
    out = F.relu(bn(p + ".bn1", _conv(sd, p + ".conv1", x)))
    if (p + ".conv2.offset_conv.weight") in sd:
        out = dcn_block(sd, p + ".conv2", out, stride)
    else:
        out = _conv(sd, p + ".conv2", out, stride, 1)
    out = F.relu(bn(p + ".bn2", out))
    out = bn(p + ".bn3", _conv(sd, p + ".conv3", out))
    if (p + ".downsample.0.weight") in sd:
        x = bn(p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride))
    return F.relu(out + x)


def backbone(sd, x, layers, training, upd, p="backbone"):
    """models/backbone.py:197-209"""
    x = _conv(sd, p + ".conv1", x, 2, 3)
    x = F.relu(_bn(sd, p + ".bn1", x, training, 1e-5, 0.1, upd))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = []
    for s, nblocks in enumerate(layers):
        for b in range(nblocks):
            stride = 2 if (b == 0 and s > 0) else 1      # backbone.py:114-136,170-178
            x = bottleneck(sd, f"{p}.layers.{s}.{b}", x, stride, training, upd)
        outs.append(x)
    return outs


def fpn(sd, feats, p="fpn"):
    """models/fpn.py:45-63 (bottom-up bilinear accumulate, quirk Q1)"""
    lats = []
    x = torch.zeros(1)
    for i, f in enumerate(feats):
        if i > 0:
            x = F.interpolate(x, size=f.shape[2:], mode="bilinear", align_corners=False)
        x = _conv(sd, f"{p}.lateral_convs.{i}", f) + x
        lats.append(x)
    return [F.relu(_conv(sd, f"{p}.fpn_convs.{i}", l, 1, 1)) for i, l in enumerate(lats)]


def _coord(feat):
    # planerecnet.py:370-376 / :484-490 -- channel order (x, y), meshgrid 'ij'
    B, _, h, w = feat.shape
    xr = torch.linspace(-1, 1, w)          # default dtype (fp32) as in the reference, even for fp64 features
    yr = torch.linspace(-1, 1, h)
    y, x = torch.meshgrid(yr, xr, indexing="ij")
    return torch.cat([x.expand(B, 1, h, w), y.expand(B, 1, h, w)], 1)


def _gn_relu(sd, p, x):
    return F.relu(F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-5))


def ins_head(sd, feats, num_grids, p="inst_head"):
    """planerecnet.py:355-391"""
    cate, kern = [], []
    for idx, f in enumerate(feats):
        kf = torch.cat([f, _coord(f)], 1)
        kf = F.interpolate(kf, size=num_grids[idx], mode="bilinear", align_corners=False)
        cf = kf[:, :-2]
        for i in (0, 3, 6):
            kf = _gn_relu(sd, f"{p}.kernel_tower.{i + 1}", _conv(sd, f"{p}.kernel_tower.{i}", kf, 1, 1, ragged=True))
            cf = _gn_relu(sd, f"{p}.cate_tower.{i + 1}", _conv(sd, f"{p}.cate_tower.{i}", cf, 1, 1, ragged=True))
        kern.append(_conv(sd, p + ".kernel_pred", kf, 1, 1, ragged=True))
        cate.append(_conv(sd, p + ".cate_pred", cf, 1, 1, ragged=True))
    return cate, kern


def mask_head(sd, feats, p="mask_head"):
    """planerecnet.py:467-496"""
    def level(i, x):
        for j in range(max(i, 1)):
            q = f"{p}.convs_all_levels.{i}.conv{j}"
            x = _gn_relu(sd, q + ".1", _conv(sd, q + ".0", x, 1, 1))
            if i > 0:
                x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        return x
    acc = level(0, feats[0])
    for i in range(1, len(feats)):
        f = feats[i]
        if i == 3:
            f = torch.cat([f, _coord(f)], 1)
        acc = acc + level(i, f)
    return _gn_relu(sd, p + ".conv_pred.1", _conv(sd, p + ".conv_pred.0", acc))


def plane_prior(sd, mask_pred, kernel_preds, num_kernels, p="depth_decoder"):
    """planerecnet.py:587-594: sigmoid(K.M) -> 1x1 (3728->256) -> x0.25 bilinear; inputs detached."""
    B = mask_pred.shape[0]
    flat = torch.cat([k.permute(0, 2, 3, 1).reshape(B, -1, num_kernels) for k in kernel_preds], 1).detach()
    mp = torch.cat([F.conv2d(mask_pred[b:b + 1].detach(), flat[b].view(-1, num_kernels, 1, 1)) for b in range(B)], 0)
    mp = mp.sigmoid().detach()
    mp = _conv(sd, p + ".conv1x1.0", mp)
    return F.interpolate(mp, scale_factor=0.25, mode="bilinear", align_corners=False, recompute_scale_factor=False)


def depth_decoder(sd, feats, mask_pred, kernel_preds, num_kernels, training, upd, p="depth_decoder"):
    """planerecnet.py:586-607"""
    def cbr(q, x, ci, bi, up):
        if up:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = _conv(sd, f"{q}.{ci}", F.pad(x, (1, 1, 1, 1), mode="reflect"), prepadded=not up)       # (the product's upsample-convolutions run in sub-pixel form, not on the Winograd path)
        return F.relu(_bn(sd, f"{q}.{bi}", x, training, 1e-3, 0.01, upd))
    prior = plane_prior(sd, mask_pred, kernel_preds, num_kernels, p)
    c2, c3, c4, c5 = feats
    x = cbr(p + ".deconv1", cbr(p + ".conv1", _conv(sd, p + ".latlayer1", c5), 1, 2, False), 2, 3, True)
    x = cbr(p + ".refine_conv", torch.cat([x, x * prior], 1), 1, 2, False)
    x = cbr(p + ".deconv2", torch.cat([cbr(p + ".conv2", _conv(sd, p + ".latlayer2", c4), 1, 2, False), x], 1), 2, 3, True)
    x = cbr(p + ".deconv3", torch.cat([cbr(p + ".conv3", _conv(sd, p + ".latlayer3", c3), 1, 2, False), x], 1), 2, 3, True)
    x = cbr(p + ".deconv4", torch.cat([cbr(p + ".conv4", _conv(sd, p + ".latlayer4", c2), 1, 2, False), x], 1), 2, 3, True)
    x = _conv(sd, p + ".depth_pred.1", F.pad(x, (1, 1, 1, 1), mode="reflect"))
    return F.softplus(x)


def forward(sd, x, arch, training=True, update_stats=False, return_stages=False):
    """planerecnet.py:73-103 train-mode return: (mask_pred, cate_pred[4], kernel_pred[4], depth_pred).
    `training` selects BatchNorm batch statistics (True) or running statistics (False)."""
    cs = backbone(sd, x, arch.layers, training, update_stats)
    ps = fpn(sd, cs)
    split = [F.interpolate(ps[0], scale_factor=0.5, mode="bilinear", align_corners=False,
                           recompute_scale_factor=False)] + ps[1:]          # planerecnet.py:113-118
    cate, kern = ins_head(sd, split, arch.num_grids)
    mask = mask_head(sd, ps)
    depth = depth_decoder(sd, cs, mask, kern, arch.num_kernels, training, update_stats)
    if return_stages:
        return {"c": cs, "p": ps, "cate": cate, "kernel": kern, "mask": mask, "depth": depth}
    return mask, cate, kern, depth


# ----------------------------------------------------------------------------------------------
# inference post-process (planerecnet.py:104-111,155-289 ; models/functions/nms.py)
# ----------------------------------------------------------------------------------------------
def point_nms(heat):
    hmax = F.max_pool2d(heat, (2, 2), stride=1, padding=1)
    return heat * (hmax[:, :, :-1, :-1] == heat).to(heat.dtype)


def matrix_nms(labels, masks, sum_masks, scores, sigma=2.0, kernel="gaussian"):
    n = len(labels)
    if n == 0:
        return []
    m = masks.reshape(n, -1).to(scores.dtype)
    inter = m @ m.t()
    sx = sum_masks.expand(n, n)
    iou = (inter / (sx + sx.t() - inter)).triu(diagonal=1)
    lx = labels.expand(n, n)
    same = (lx == lx.t()).to(scores.dtype).triu(diagonal=1)
    decay = iou * same
    comp = decay.max(0)[0].expand(n, n).t()
    if kernel == "linear":
        coef = ((1 - decay) / (1 - comp)).min(0)[0]
    else:
        coef = (torch.exp(-sigma * decay ** 2) / torch.exp(-sigma * comp ** 2)).min(0)[0]
    return scores * coef


def postprocess_image(seg, cate, kernels, depth, ori_size, arch, strides=(8, 8, 16, 32), score_thr=0.1,
                      mask_thr=0.1, update_thr=0.15, nms_pre=500, top_k=100, sigma=2.0, nms_kernel="gaussian"):
    """planerecnet.py:182-289 for nms_type == 'matrix'. seg [1,E,h,w]; cate [sumS2, C]; kernels [sumS2, E]."""
    res = {"pred_masks": None, "pred_boxes": None, "pred_classes": None, "pred_scores": None,
           "pred_depth": F.interpolate(depth, size=ori_size, mode="bilinear", align_corners=False)}
    inds = cate > score_thr
    scores = cate[inds]
    if len(scores) == 0:
        return res
    nz = inds.nonzero()
    labels = nz[:, 1]
    kp = kernels[nz[:, 0]]
    stride_vec = torch.cat([torch.full((g * g,), float(s)) for g, s in zip(arch.num_grids, strides)])[nz[:, 0]]
    sp = F.conv2d(seg, kp.view(kp.shape[0], -1, 1, 1)).squeeze(0).sigmoid()
    sm = sp > mask_thr
    area = sm.sum((1, 2)).float()
    keep = area > stride_vec
    if keep.sum() == 0:
        return res
    sm, sp, area, scores, labels = sm[keep], sp[keep], area[keep], scores[keep], labels[keep]
    scores = scores * ((sp * sm.float()).sum((1, 2)) / area)
    order = torch.argsort(scores, descending=True)[:nms_pre]
    sm, sp, area, scores, labels = sm[order], sp[order], area[order], scores[order], labels[order]
    scores = matrix_nms(labels, sm, area, scores, sigma=sigma, kernel=nms_kernel)
    keep = scores >= update_thr
    if keep.sum() == 0:
        return res
    sp, scores, labels = sp[keep], scores[keep], labels[keep]
    order = torch.argsort(scores, descending=True)[:top_k]
    sp, scores, labels = sp[order], scores[order], labels[order]
    full = F.interpolate(sp.unsqueeze(0), size=ori_size, mode="bilinear", align_corners=False).squeeze(0) > mask_thr
    boxes = torch.zeros(full.size(0), 4)
    for i in range(full.size(0)):
        ys, xs = torch.where(full[i])
        boxes[i] = torch.tensor([xs.min(), ys.min(), xs.max(), ys.max()]).float()
    res.update(pred_scores=scores, pred_classes=labels, pred_masks=full, pred_boxes=boxes)
    return res


@torch.no_grad()
def inference(sd, x, arch, **kw):
    """Eval-mode forward + post-process: planerecnet.py:104-111,155-180."""
    mask, cate, kern, depth = forward(sd, x, arch, training=False)
    cate = [point_nms(c.sigmoid()).permute(0, 2, 3, 1) for c in cate]
    out = []
    for b in range(x.shape[0]):
        c = torch.cat([ci[b].reshape(-1, arch.num_classes) for ci in cate], 0)
        k = torch.cat([ki[b].permute(1, 2, 0).reshape(-1, arch.num_kernels) for ki in kern], 0)
        out.append(postprocess_image(mask[b:b + 1], c, k, depth[b:b + 1], tuple(x.shape[2:]), arch, **kw))
    return out

"""TEST INFRASTRUCTURE -- CPU restatement of the PlaneRecNet joint loss (five terms).

Follows /root/reference/models/functions/losses.py:53-392 and vnl.py:6-165, quirks included
(Q2 discarded clamp, Q3 lava valid_mask always None, Q5 hard-wired 480x640 VNL
principal point, Q6 float64 plane term).  Pinned against the shim-imported reference by
tests/golden/make_golden.py.  The virtual-normal term draws its triplets from `np.random` in the
same call order as vnl.py:43-55 so that a shared seed reproduces the reference stream.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F

SCALE_RANGES = ((1, 128), (64, 256), (128, 512), (256, 2048))     # config.py:371
SIGMA = 0.2                                                        # config.py:377
W_DICE, W_FOCAL, W_DEPTH, W_LAVA, W_PLANE = 3.0, 1.0, 5.0, 1.0, 1.0  # config.py:459-466,511-514
MIN_DEPTH = DEPTH_RES = 1 / 1000                                   # config.py:131-133


def quarter_mask(m):
    """losses.py:243-247 -> cv2 INTER_LINEAR at exact 1/4: mean of the 2x2 centre pixels, round-half-up.
    m: uint8 [N,H,W] -> uint8 [N,H/4,W/4]."""
    a = m.to(torch.int32)
    s = a[:, 1::4, 1::4] + a[:, 1::4, 2::4] + a[:, 2::4, 1::4] + a[:, 2::4, 2::4]
    return ((s + 2) >> 2).to(torch.uint8)


def center_of_mass(m):
    """funcs.py:213-224"""
    _, h, w = m.shape
    ys = torch.arange(h, dtype=torch.float32)
    xs = torch.arange(w, dtype=torch.float32)
    m00 = m.sum(-1).sum(-1).clamp(min=1e-6)
    return (m * xs).sum(-1).sum(-1) / m00, (m * ys[:, None]).sum(-1).sum(-1) / m00


@torch.no_grad()
def assign_targets(inst, feat_hw, num_grids, num_classes=2):
    """losses.py:200-286 for one image. Returns per level: ins_label uint8 [n,h,w], cate_label int64
    [S,S], ins_ind bool [S*S], grid_order list[int]."""
    boxes, labels, masks = inst["boxes"], inst["classes"], inst["masks"]
    areas = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    fh, fw = feat_hw
    up_h, up_w = fh * 4, fw * 4
    out = []
    for (lo, hi), S in zip(SCALE_RANGES, num_grids):
        hit = ((areas >= lo) & (areas <= hi)).nonzero().flatten()
        cate = torch.full((S, S), num_classes, dtype=torch.int64)
        ind = torch.zeros(S * S, dtype=torch.bool)
        ins, order = [], []
        if len(hit) > 0:
            bx, lb, mk = boxes[hit], labels[hit], masks[hit]
            half_w = 0.5 * (bx[:, 2] - bx[:, 0]) * SIGMA
            half_h = 0.5 * (bx[:, 3] - bx[:, 1]) * SIGMA
            cw, ch = center_of_mass(mk)
            valid = mk.sum(-1).sum(-1) > 0
            small = quarter_mask(mk.to(torch.uint8))
            for seg, l, hh, hw, cy, cx, ok in zip(small, lb, half_h, half_w, ch, cw, valid):
                if not ok:
                    continue
                g = 1.0 / S
                coord_w = int((cx / up_w) // g)
                coord_h = int((cy / up_h) // g)
                top = max(max(0, int(((cy - hh) / up_h) // g)), coord_h - 1)
                down = min(min(S - 1, int(((cy + hh) / up_h) // g)), coord_h + 1)
                left = max(coord_w - 1, max(0, int(((cx - hw) / up_w) // g)))
                right = min(min(S - 1, int(((cx + hw) / up_w) // g)), coord_w + 1)
                cate[top:down + 1, left:right + 1] = l
                for i in range(top, down + 1):
                    for j in range(left, right + 1):
                        lab = torch.zeros(fh, fw, dtype=torch.uint8)
                        lab[:seg.shape[0], :seg.shape[1]] = seg
                        ins.append(lab)
                        ind[i * S + j] = True
                        order.append(i * S + j)
        ins = torch.stack(ins, 0) if ins else torch.zeros(0, fh, fw, dtype=torch.uint8)
        out.append((ins, cate, ind, order))
    return out


def dice_loss(p, t):
    """losses.py:355-368"""
    p = p.reshape(p.shape[0], -1)
    t = t.reshape(t.shape[0], -1).float()
    return 1 - 2 * (p * t).sum(1) / ((p * p).sum(1) + 0.001 + (t * t).sum(1) + 0.001)


def focal_loss_sum(x, t, alpha=0.25, gamma=2.0):
    """losses.py:331-352"""
    p = torch.sigmoid(x)
    ce = F.binary_cross_entropy_with_logits(x, t, reduction="none")
    p_t = p * t + (1 - p) * (1 - t)
    return ((alpha * t + (1 - alpha) * (1 - t)) * ce * (1 - p_t) ** gamma).sum()


def rmse_log(pred, gt, valid, clamp=1e-9):
    """losses.py:371-392 (reduction mean)"""
    N = pred.shape[0]
    l1 = (torch.log(pred.reshape(N, -1).clamp(min=clamp)) - torch.log(gt.reshape(N, -1).clamp(min=clamp))).abs()
    l1 = l1.mul(valid.reshape(N, -1))
    return torch.sqrt((l1 ** 2).sum(1) / valid.reshape(N, -1).sum(1)).mean()


@torch.no_grad()
def sobel_sq(d):
    """losses.py:304-329 (valid_mask None)"""
    kx = torch.tensor([[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]]).view(1, 1, 3, 3) / 8.0
    ky = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]]).view(1, 1, 3, 3) / 8.0
    dp = F.pad(d, (1, 1, 1, 1), mode="reflect")
    return F.conv2d(dp, kx) ** 2 + F.conv2d(dp, ky) ** 2


# ---------------------------------------------------------------------------- virtual normal loss
class VNL:
    """vnl.py:6-165"""

    def __init__(self, size=(480, 640), sample_ratio=0.3, delta_z=1e-4):
        H, W = size
        self.u_u0 = (torch.arange(W, dtype=torch.float32).view(1, 1, W).expand(1, H, W) - float(W // 2))
        self.v_v0 = (torch.arange(H, dtype=torch.float32).view(1, H, 1).expand(1, H, W) - float(H // 2))
        self.ratio, self.delta_z = sample_ratio, delta_z

    def xyz(self, depth, K):
        x = self.u_u0 * depth.abs() / K[0, 0]
        y = self.v_v0 * depth.abs() / K[1, 1]
        return torch.cat([x, y, depth], 0).permute(1, 2, 0)

    def draw(self, num):
        n = int(num * self.ratio)
        out = []
        for _ in range(3):
            p = np.random.choice(num, n, replace=True)
            np.random.shuffle(p)
            out.append(p)
        return out

    @staticmethod
    def groups(p123, pw):
        return torch.stack([pw[p123[0]], pw[p123[1]], pw[p123[2]]], 2)     # [n, xyz, p]

    def filt(self, p123, pc, delta_cos=0.985, delta_diff=0.005):
        pw = self.groups(p123, pc)
        d = torch.stack([pw[:, :, 1] - pw[:, :, 0], pw[:, :, 2] - pw[:, :, 0], pw[:, :, 2] - pw[:, :, 1]], 2)
        q = d.permute(0, 2, 1)
        qn = q.norm(2, dim=2)
        e = torch.bmm(q, d) / (torch.bmm(qn.unsqueeze(2), qn.unsqueeze(1)) + 1e-8)
        e = e.reshape(e.shape[0], -1)
        m_cos = ((e > delta_cos) + (e < -delta_cos)).sum(1) > 3
        m_pad = (pw[:, 2, :] > self.delta_z).sum(1) == 3
        near = [(d[:, a, :].abs() < delta_diff).sum(1) > 0 for a in range(3)]
        return m_pad & ~((near[0] & near[1] & near[2]) | m_cos), pw

    @staticmethod
    def normals(tri, m):
        t = tri[m]
        n = torch.cross(t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 0], dim=1)
        nn = n.norm(2, dim=1, keepdim=True)
        return n / (nn + (nn == 0.0).float() * 0.01)

    @staticmethod
    def trimmed(loss):
        loss = torch.sort(loss, dim=0)[0]
        loss = loss[int(loss.shape[0] * 0.25):]
        return torch.nansum(loss) / loss.shape[0]

    def __call__(self, pred, masks, normals, gt_depth, K):
        pc = self.xyz(pred, K)
        N = normals.shape[0]
        total = 0
        nonplanar = ~masks.sum(0).bool()
        for i in range(N):
            seg = pc[masks[i]]
            p123 = self.draw(seg.shape[0])
            m, pw = self.filt(p123, seg)
            dn = self.normals(pw, m)
            total = total + self.trimmed(1 - F.cosine_similarity(dn, normals[i].unsqueeze(0), dim=1).abs())
        if nonplanar.sum() > 0:
            gpc = self.xyz(gt_depth, K)
            pp, gp = pc[nonplanar], gpc[nonplanar]
            p123 = self.draw(gp.shape[0])
            m, pw_gt = self.filt(p123, gp, delta_diff=0.1)
            if m.sum() == 0:
                return total / N
            pw_pred = self.groups(p123, pp)
            pw_pred[pw_pred[:, 2, :] == 0] = 0.0001
            total = total + self.trimmed(1 - F.cosine_similarity(self.normals(pw_pred, m), self.normals(pw_gt, m), dim=1).abs())
            return total / (N + 1)
        return total / N


# ---------------------------------------------------------------------------------- joint loss
def joint_loss(mask_pred, cate_preds, kernel_preds, depth_pred, gt_instances, gt_depths, num_grids=(40, 36, 24, 16),
               num_classes=2):
    """losses.py:53-198 -> {'ins','cat','dpt','pln','lav'}"""
    B = mask_pred.shape[0]
    fh, fw = mask_pred.shape[-2:]
    tg = [assign_targets(g, (fh, fw), num_grids, num_classes) for g in gt_instances]
    L = len(num_grids)
    per_img = [[] for _ in range(B)]
    ins_terms = []
    for lv in range(L):
        preds = []
        for b in range(B):
            order = tg[b][lv][3]
            if len(order) == 0:
                continue
            k = kernel_preds[lv][b].reshape(kernel_preds[lv].shape[1], -1)[:, order]        # [E, n]
            p = F.conv2d(mask_pred[b:b + 1], k.t().reshape(len(order), -1, 1, 1)).view(-1, fh, fw)
            preds.append(p)
            per_img[b].append(p)
        if preds:
            tgt = torch.cat([tg[b][lv][0] for b in range(B)], 0)
            ins_terms.append(dice_loss(torch.sigmoid(torch.cat(preds, 0)), tgt))
    num_ins = torch.cat([tg[b][lv][2] for lv in range(L) for b in range(B)]).sum()
    out = {"ins": torch.cat(ins_terms).mean() * W_DICE}

    flat_lab = torch.cat([tg[b][lv][1].flatten() for lv in range(L) for b in range(B)])
    flat_pred = torch.cat([c.permute(0, 2, 3, 1).reshape(-1, num_classes) for c in cate_preds])
    pos = torch.nonzero(flat_lab != num_classes).squeeze(1)
    oh = torch.zeros_like(flat_pred)
    oh[pos, flat_lab[pos]] = 1
    out["cat"] = W_FOCAL * focal_loss_sum(flat_pred, oh) / (num_ins + 1)

    dp = F.interpolate(depth_pred, scale_factor=2, mode="bilinear", align_corners=False)
    out["dpt"] = W_DEPTH * rmse_log(dp, gt_depths, gt_depths > MIN_DEPTH)

    vnl = VNL((480, 640))
    pl = [vnl(dp[b], gt_instances[b]["masks"].bool(), gt_instances[b]["plane_paras"][:, :3], gt_depths[b],
              gt_instances[b]["k_matrix"]) for b in range(B)]
    out["pln"] = torch.stack(pl).mean() * W_PLANE

    g = sobel_sq(gt_depths) / gt_depths.clamp(min=DEPTH_RES) ** 2
    g = g.clamp(max=1e-2)
    g[g < 1e-4] = 0
    lav = []
    for b in range(B):
        if per_img[b] and g[b].sum() > 0:
            s = torch.cat(per_img[b], 0).sigmoid()
            s = F.interpolate(s.unsqueeze(0), size=g[b].shape[1:], mode="bilinear").squeeze(0)
            lav.append((s * g[b]).sum() / (g[b].sum() * s.shape[0]))
    out["lav"] = torch.stack(lav).mean() * W_LAVA if lav else torch.tensor([0.])
    return out

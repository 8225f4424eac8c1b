"""TEST INFRASTRUCTURE -- seeded synthetic weights and inputs shared by the oracle, the golden
generator, the parity tests and bench.py's cpu_baseline leg.

The reference's released checkpoints are not available offline (README.md:37-38,71-72), so parity
is established on seeded random weights.  `spec()` enumerates the reference's state-dict layout
 independently of the product model, which lets tests assert that the product's
`state_dict()` has exactly these keys/shapes (strict-load compatibility, planerecnet.py:125-128).
Synthetic inputs follow SURVEY.md 8(d).
"""
import math

import numpy as np
import torch

from oracle.model_ref import ARCH


def _dcn_blocks(layers, dcn_layers, interval):
    """models/backbone.py:170,184"""
    out = set()
    for s, (n, d) in enumerate(zip(layers, dcn_layers)):
        for i in range(n):
            if (i == 0 and d >= n) or (i > 0 and (i + d) >= n and i % interval == 0):
                out.add((s, i))
    return out


DCN_RULE = {"PlaneRecNet_101_config": ((0, 4, 23, 3), 3), "PlaneRecNet_50_config": ((0, 4, 6, 3), 1)}


def spec(config_name):
    """-> list of (key, shape, kind); kind in conv|bias|bn_w|bn_b|bn_rm|bn_rv|bn_nbt|gn_w|gn_b|dcn_off_w|dcn_off_b"""
    arch = ARCH[config_name]
    dcn = _dcn_blocks(arch.layers, *DCN_RULE[config_name])
    out = []

    def conv(k, co, ci, ks, bias):
        out.append((k + ".weight", (co, ci, ks, ks), "conv"))
        if bias:
            out.append((k + ".bias", (co,), "bias"))

    def bn(k, c):
        out.extend([(k + ".weight", (c,), "bn_w"), (k + ".bias", (c,), "bn_b"), (k + ".running_mean", (c,), "bn_rm"),
                    (k + ".running_var", (c,), "bn_rv"), (k + ".num_batches_tracked", (), "bn_nbt")])

    def gn(k, c):
        out.extend([(k + ".weight", (c,), "gn_w"), (k + ".bias", (c,), "gn_b")])

    conv("backbone.conv1", 64, 3, 7, False)
    bn("backbone.bn1", 64)
    inpl = 64
    for s, n in enumerate(arch.layers):
        pl = 64 << s
        for b in range(n):
            p = f"backbone.layers.{s}.{b}"
            conv(p + ".conv1", pl, inpl, 1, False)
            bn(p + ".bn1", pl)
            if (s, b) in dcn:
                out.append((p + ".conv2.offset_conv.weight", (18, pl, 3, 3), "dcn_off_w"))
                out.append((p + ".conv2.offset_conv.bias", (18,), "dcn_off_b"))
                out.append((p + ".conv2.modulator_conv.weight", (9, pl, 3, 3), "dcn_off_w"))
                out.append((p + ".conv2.modulator_conv.bias", (9,), "dcn_off_b"))
                conv(p + ".conv2.regular_conv", pl, pl, 3, True)
            else:
                conv(p + ".conv2", pl, pl, 3, False)
            bn(p + ".bn2", pl)
            conv(p + ".conv3", pl * 4, pl, 1, False)
            bn(p + ".bn3", pl * 4)
            if b == 0:
                conv(p + ".downsample.0", pl * 4, inpl, 1, False)
                bn(p + ".downsample.1", pl * 4)
            inpl = pl * 4
    for i, c in enumerate((256, 512, 1024, 2048)):
        conv(f"fpn.lateral_convs.{i}", 256, c, 1, True)
    for i in range(4):
        conv(f"fpn.fpn_convs.{i}", 256, 256, 3, True)
    for tower, cin0 in (("cate", 256), ("kernel", 258)):
        for j, i in enumerate((0, 3, 6)):
            conv(f"inst_head.{tower}_tower.{i}", 256, cin0 if j == 0 else 256, 3, False)
            gn(f"inst_head.{tower}_tower.{i + 1}", 256)
    conv("inst_head.cate_pred", arch.num_classes, 256, 3, True)
    conv("inst_head.kernel_pred", arch.num_kernels, 256, 3, True)
    for lv in range(4):
        for j in range(max(lv, 1)):
            cin = (258 if lv == 3 else 256) if j == 0 else 128
            conv(f"mask_head.convs_all_levels.{lv}.conv{j}.0", 128, cin, 3, False)
            gn(f"mask_head.convs_all_levels.{lv}.conv{j}.1", 128)
    conv("mask_head.conv_pred.0", 128, 128, 1, False)
    gn("mask_head.conv_pred.1", 128)
    for i, c in enumerate((2048, 1024, 512, 256)):
        conv(f"depth_decoder.latlayer{i + 1}", 256, c, 1, True)
    for i, co in enumerate((256, 128, 128, 128)):
        conv(f"depth_decoder.conv{i + 1}.1", co, 256, 3, True)
        bn(f"depth_decoder.conv{i + 1}.2", co)
    for i, co in enumerate((256, 128, 128, 64)):
        conv(f"depth_decoder.deconv{i + 1}.2", co, 256, 3, True)
        bn(f"depth_decoder.deconv{i + 1}.3", co)
    conv("depth_decoder.depth_pred.1", 1, 64, 3, True)
    conv("depth_decoder.conv1x1.0", 256, sum(g * g for g in arch.num_grids), 1, True)
    conv("depth_decoder.refine_conv.1", 128, 512, 3, True)
    bn("depth_decoder.refine_conv.2", 128)
    return out


def make_state_dict(config_name, seed=0, dcn_offset_px=0.6, dtype=torch.float32):
    """Seeded weights with O(1) activations through the whole net and NON-ZERO DCN offset /
    modulator convs (so bilinear sampling is exercised, unlike the reference's zero init dcn.py:32-43).
    Conditioning matters for a parity oracle: offsets of ~`dcn_offset_px` pixels r.m.s. and a damped residual
    branch (bn3 gamma x0.3) keep the fp32-vs-fp64 spread of the oracle itself at ~1e-5 (train-mode BN) / 1e-6 (eval);
    with large offsets it is 3e-3..1e-1 and no fp32 implementation could be told apart from a wrong one."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape, kind in spec(config_name):
        if kind == "conv":
            fan_in = shape[1] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * math.sqrt(1.6 / fan_in)
        elif kind == "dcn_off_w":
            t = torch.randn(shape, generator=g) * dcn_offset_px / math.sqrt(shape[1] * 9)
        elif kind in ("bias", "bn_b", "gn_b", "bn_rm", "dcn_off_b"):
            t = torch.randn(shape, generator=g) * 0.1
        elif kind in ("bn_w", "gn_w", "bn_rv"):
            t = 0.6 + 0.5 * torch.rand(shape, generator=g)
            if key.endswith("bn3.weight"):
                t = t * 0.3
        elif kind == "bn_nbt":
            t = torch.zeros((), dtype=torch.long)
        else:
            raise KeyError(kind)
        sd[key] = t if kind == "bn_nbt" else t.to(dtype)
    return sd


def make_batch(B, H=480, W=640, seed=0):
    """images N(0,1); GT depth U(0.5,4.5); 3-8 rectangular plane masks per image with
    float64 xyxy boxes, int64 zero classes, float64 plane params (unit normal + offset + 2 pad, datasets.py:247)
    and a float64 K (fx=fy=577)."""
    rng = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(B, 3, H, W, generator=g)
    depths = 0.5 + 4.0 * torch.rand(B, 1, H, W, generator=g)
    inst = []
    for _ in range(B):
        n = int(rng.randint(3, 9))
        masks = np.zeros((n, H, W), np.uint8)
        boxes = np.zeros((n, 4), np.float64)
        for i in range(n):
            bw = int(rng.randint(max(W // 16, 8), W // 2))
            bh = int(rng.randint(max(H // 16, 8), H // 2))
            x0 = int(rng.randint(0, W - bw))
            y0 = int(rng.randint(0, H - bh))
            masks[i, y0:y0 + bh, x0:x0 + bw] = 1
            boxes[i] = (x0, y0, x0 + bw, y0 + bh)
        nrm = rng.randn(n, 3)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        paras = np.concatenate([nrm, rng.rand(n, 1) * 3.0, np.zeros((n, 2))], 1)
        K = np.array([[577.0, 0, W / 2], [0, 577.0, H / 2], [0, 0, 1]], np.float64)
        inst.append({"masks": torch.from_numpy(masks), "boxes": torch.from_numpy(boxes),
                     "classes": torch.zeros(n, dtype=torch.int64), "plane_paras": torch.from_numpy(paras),
                     "k_matrix": torch.from_numpy(K)})
    return images, inst, depths

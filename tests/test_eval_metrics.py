"""Detection metrics (reference eval.py:210-354): the oracle and the host half of the product against the reference's golden
values (CPU); the device IoU kernel against the oracle, bit for bit, and the whole path against the golden mAP table (GPU)."""
import os

import numpy as np
import pytest
import torch

SEEDS = (0, 1, 2, 3, 4)


def make_frame(seed, H=60, W=80):          # the generator of tests/golden/make_golden_eval.py
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
    for _ in range(int(rng.randint(1, 4))):
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


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "eval_metrics.npz"))


def _table_row(table):
    return np.array([table[k] for k in ["all"] + [int(t * 100) for t in [x / 100 for x in range(50, 100, 5)]]])


def test_oracle_equals_reference_golden(fx):
    from oracle import eval_ref
    data = eval_ref.new_ap_data()
    for seed in SEEDS:
        f = make_frame(seed)
        pm, gm = f["pred_masks"].float(), f["gt_masks"].float()
        assert np.array_equal(eval_ref.mask_iou_ref(pm, gm).numpy(), fx["mask_iou_%d" % seed], equal_nan=True)
        assert np.array_equal(eval_ref.bbox_iou_ref(f["pred_boxes"].float(), f["gt_boxes"].float()).numpy(), fx["box_iou_%d" % seed], equal_nan=True)
        eval_ref.segmentation_metrics_ref(data, gm, f["gt_boxes"], f["gt_classes"], pm, f["pred_boxes"], f["pred_classes"], f["pred_scores"])
    assert (fx["mask_iou_3"][0] == 0).all()                      # the empty detection mask of frame 3
    table = eval_ref.calc_map_ref(data)
    for kind in ("box", "mask"):
        assert [len(o.points) for o in data[kind]] == fx["points_" + kind].tolist()
        assert np.allclose([o.ap() for o in data[kind]], fx["ap_" + kind], rtol=0, atol=1e-12)
        assert np.allclose(np.round(_table_row(table[kind]), 2), fx["map_rounded_" + kind], rtol=0, atol=1e-9)


def test_host_matching_and_ap_equal_reference_golden(fx, capsys):
    """The product's host half (matching, AP integral, table) fed with the reference's IoU matrices."""
    from planerecnet_amd import metrics
    data = metrics.new_ap_data()
    for seed in SEEDS:
        f = make_frame(seed)
        metrics.match_frame(data, fx["mask_iou_%d" % seed], fx["box_iou_%d" % seed], f["pred_scores"].numpy(), int((f["gt_classes"] == 0).sum()))
    for kind in ("box", "mask"):
        assert [len(o.scores) for o in data[kind]] == fx["points_" + kind].tolist()
        assert data[kind][0].num_gt_positives == int(fx["gt_positives"])
        assert np.allclose([o.get_ap() for o in data[kind]], fx["ap_" + kind], rtol=0, atol=1e-12)
    table = metrics.calc_map(data)
    out = capsys.readouterr().out
    assert "Calculating mAP..." in out and "  mask |" in out and " .50  |" in out
    for kind in ("box", "mask"):
        assert np.allclose(_table_row(table[kind]), fx["map_rounded_" + kind], rtol=0, atol=1e-9)


def test_ap_object_edge_cases():
    from oracle.eval_ref import APDataRef
    from planerecnet_amd.metrics import APDataObject, calc_map, new_ap_data
    o = APDataObject()
    assert o.is_empty() and o.get_ap() == 0
    o.add_gt_positives(3)
    assert not o.is_empty() and o.get_ap() == 0.0            # ground truth but no detections
    r = APDataRef()
    r.gt_total = 3
    rng = np.random.RandomState(7)
    for _ in range(40):                                         # random pushes with tied scores: same AP as the scalar restatement
        s, h = round(float(rng.rand()), 1), bool(rng.rand() < 0.4)
        o.push(s, h)
        r.points.append((s, h))
    assert abs(o.get_ap() - r.ap()) < 1e-15
    table = calc_map(new_ap_data(), quiet=True)                 # nothing seen at all: a table of zeros
    assert all(v == 0 for v in table["box"].values()) and list(table["mask"].keys())[0] == "all"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_device_iou_equals_oracle_and_golden_bitwise(fx, seed):
    from oracle import eval_ref
    from planerecnet_amd import metrics
    f = make_frame(seed)
    for pm in (f["pred_masks"], f["pred_masks"].float(), f["pred_masks"].to(torch.uint8)):          # bool (model output), float (eval.py:99), bytes
        miou, biou = metrics.pairwise_iou(pm.cuda(), f["gt_masks"].cuda(), f["pred_boxes"].cuda(), f["gt_boxes"].cuda())
        assert np.array_equal(miou.cpu().numpy(), fx["mask_iou_%d" % seed], equal_nan=True)
        assert np.array_equal(biou.cpu().numpy(), fx["box_iou_%d" % seed], equal_nan=True)
    assert np.array_equal(metrics.mask_iou(f["pred_masks"].cuda(), f["gt_masks"].cuda()).cpu().numpy(),
                          eval_ref.mask_iou_ref(f["pred_masks"], f["gt_masks"]).numpy(), equal_nan=True)
    assert np.array_equal(metrics.bbox_iou(f["pred_boxes"].cuda(), f["gt_boxes"].float().cuda()).cpu().numpy(), fx["box_iou_%d" % seed])


@pytest.mark.gpu
@pytest.mark.parametrize("shape,A,B", [((480, 640), 100, 24), ((736, 960), 37, 9), ((17, 23), 5, 3), ((1, 33), 2, 2), ((5, 13), 1, 1)])
def test_device_iou_full_size_and_ragged_word_tails(shape, A, B):
    """Full-size frames (the top_k = 100 detections of eval.py), plane sizes that are not multiples of 32 pixels (tail word) or of
    16 bytes (unaligned rows), against the oracle's float matmul -- integer counts, so equality is exact."""
    from oracle import eval_ref
    from planerecnet_amd import metrics
    g = torch.Generator().manual_seed(A * 1000 + B)
    pm = torch.rand(A, *shape, generator=g) < 0.3
    gm = (torch.rand(B, *shape, generator=g) < 0.4).to(torch.uint8)
    if A > 1:
        pm[0], gm[0] = False, 0                                    # two empty masks: 0/0 = NaN on both sides
    pb = torch.rand(A, 4, generator=g) * 100
    pb[:, 2:] += pb[:, :2]
    gb = (torch.rand(B, 4, generator=g) * 100).double()
    gb[:, 2:] += gb[:, :2]
    miou, biou = metrics.pairwise_iou(pm.cuda(), gm.cuda(), pb.cuda(), gb.cuda())
    assert np.array_equal(miou.cpu().numpy(), eval_ref.mask_iou_ref(pm, gm).numpy(), equal_nan=True)
    assert np.array_equal(biou.cpu().numpy(), eval_ref.bbox_iou_ref(pb, gb.float()).numpy(), equal_nan=True)
    # size-independent properties: IoU of a set with itself is 1, the matrix of (a, b) is the transpose of (b, a)
    self_iou = metrics.mask_iou(gm.cuda(), gm.cuda()).cpu()
    assert torch.equal(torch.diagonal(self_iou)[1:], torch.ones(B - 1))
    assert np.array_equal(metrics.mask_iou(gm.cuda(), pm.cuda()).cpu().numpy(), miou.cpu().t().numpy(), equal_nan=True)


@pytest.mark.gpu
def test_segmentation_metrics_end_to_end_equal_reference_golden(fx):
    from planerecnet_amd import metrics
    data = metrics.new_ap_data()
    for seed in SEEDS:
        f = make_frame(seed)
        # the argument types evaluate() hands over: device masks / scores, boxes on the host (planerecnet.py:282), float64 gt boxes
        metrics.compute_segmentation_metrics(data, f["gt_masks"].cuda().float(), f["gt_boxes"].cuda(), f["gt_classes"].cuda(),
                                             f["pred_masks"].cuda().float(), f["pred_boxes"], f["pred_classes"].cuda(), f["pred_scores"].cuda())
    table = metrics.calc_map(data, quiet=True)
    for kind in ("box", "mask"):
        assert np.allclose([o.get_ap() for o in data[kind]], fx["ap_" + kind], rtol=0, atol=1e-12)
        assert np.allclose(_table_row(table[kind]), fx["map_rounded_" + kind], rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_pairwise_iou_rejects_and_empty_sets():
    from planerecnet_amd import metrics
    with pytest.raises(RuntimeError):
        metrics.pairwise_iou(torch.ones(2, 4, 4), torch.ones(2, 4, 4))
    with pytest.raises(RuntimeError):
        metrics.pairwise_iou(torch.ones(2, 4, 4).cuda(), torch.ones(2, 4, 5).cuda())
    with pytest.raises(NotImplementedError):
        metrics.mask_iou(torch.ones(2, 4, 4).cuda(), torch.ones(2, 4, 4).cuda(), iscrowd=True)
    m, b = metrics.pairwise_iou(torch.ones(3, 4, 4).cuda(), torch.ones(0, 4, 4).cuda(), torch.ones(3, 4).cuda(), torch.ones(0, 4).cuda())
    assert m.shape == (3, 0) and b.shape == (3, 0)
    data = metrics.new_ap_data()                                   # a frame without ground truth: every detection is a miss
    metrics.match_frame(data, m.cpu().numpy(), b.cpu().numpy(), np.array([0.9, 0.5, 0.7], np.float32), 0)
    assert data["mask"][0].hits == [False] * 3 and data["mask"][0].scores == pytest.approx([0.9, 0.7, 0.5])


@pytest.mark.gpu
@pytest.mark.parametrize("n,H,W", [(7, 480, 640), (3, 37, 129), (1, 1, 1), (5, 736, 960), (4, 30, 48)])
def test_mask_boxes_equal_the_vectorised_torch_form(n, H, W):
    """prn_mask_boxes against the where / min / max form of the tight boxes (reference planerecnet.py:282-287), incl. an empty mask,
    single pixels in the corners and sizes that are not multiples of the wave width."""
    from planerecnet_amd import metrics
    g = torch.Generator().manual_seed(n * 1000 + W)
    m = torch.zeros(n, H, W, dtype=torch.bool)
    for i in range(n):
        if i == 1:
            continue                                                # stays empty
        if i == 2:
            m[i, H - 1, W - 1] = True                               # one pixel, last row / column
            continue
        y0, x0 = int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))
        y1, x1 = int(torch.randint(y0, H, (1,), generator=g)), int(torch.randint(x0, W, (1,), generator=g))
        m[i, y0:y1 + 1, x0:x1 + 1] = torch.rand(y1 - y0 + 1, x1 - x0 + 1, generator=g) < 0.3
        m[i, y0, x0] = True
    rows, cols = m.any(2), m.any(1)
    ar_h, ar_w, big = torch.arange(H), torch.arange(W), H + W
    want = torch.stack([torch.where(cols, ar_w, big).min(1)[0], torch.where(rows, ar_h, big).min(1)[0],
                        torch.where(cols, ar_w, -1).max(1)[0], torch.where(rows, ar_h, -1).max(1)[0]], 1).float()
    assert torch.equal(metrics.mask_boxes(m.cuda()).cpu(), want)
    assert torch.equal(metrics.mask_boxes(m.to(torch.uint8).cuda()).cpu(), want)
    assert metrics.mask_boxes(m[:0].cuda()).shape == (0, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("n,kernel", [(1, "gaussian"), (2, "gaussian"), (37, "gaussian"), (300, "gaussian"), (500, "linear"), (64, "linear"), (2500, "gaussian")])
def test_matrix_nms_kernel_equals_the_dense_torch_form_bitwise(n, kernel):
    """prn_matrix_nms against the dense [n, n] form of models/functions/nms.py:15-50 evaluated with torch on the device (same IoU matrix)."""
    from planerecnet_amd import metrics
    g = torch.Generator().manual_seed(n)
    H, W = 60, 80
    masks = torch.zeros(n, H, W, dtype=torch.bool)
    for i in range(n):
        y0, x0 = int(torch.randint(0, H - 8, (1,), generator=g)), int(torch.randint(0, W - 8, (1,), generator=g))
        masks[i, y0:y0 + int(torch.randint(4, 30, (1,), generator=g)), x0:x0 + int(torch.randint(4, 40, (1,), generator=g))] = True
    labels = torch.randint(0, 3, (n,), generator=g)
    scores = torch.sort(torch.rand(n, generator=g), descending=True)[0]
    md, ld, sd = masks.cuda(), labels.cuda(), scores.cuda()
    iou = metrics.mask_iou(md, md)
    got = metrics.matrix_nms_scores(iou, ld, sd, 2.0, kernel == "gaussian")
    it = iou.triu(diagonal=1)
    lab = ld.expand(n, n)
    decay = it * (lab == lab.t()).float().triu(diagonal=1)
    comp = decay.max(0)[0].expand(n, n).t()
    coef = ((1 - decay) / (1 - comp)).min(0)[0] if kernel == "linear" else (torch.exp(-2 * decay ** 2) / torch.exp(-2 * comp ** 2)).min(0)[0]
    want = sd * coef
    assert torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w", [(1, 60, 80), (9, 120, 160), (40, 17, 23)])
def test_mask_stats_counts_exact_and_rows_independent_of_the_batch(n, h, w):
    from planerecnet_amd import metrics
    g = torch.Generator().manual_seed(n)
    seg = torch.sigmoid(torch.randn(n, h, w, generator=g) * 2)
    seg[0, :2] = 0.05                                               # values below the threshold
    cnt, msum = metrics.mask_stats(seg.cuda(), 0.1)
    m = seg > 0.1
    assert torch.equal(cnt.cpu(), m.sum((1, 2)).float())
    want = (seg.double() * m).sum((1, 2))
    assert torch.allclose(msum.cpu().double(), want, rtol=2e-6, atol=0)
    c1, s1 = metrics.mask_stats(seg[n // 2:n // 2 + 1].cuda(), 0.1)   # the same row alone: bit-identical
    assert torch.equal(c1, cnt[n // 2:n // 2 + 1]) and torch.equal(s1, msum[n // 2:n // 2 + 1])


@pytest.mark.gpu
def test_category_scores_equal_sigmoid_point_nms_and_concatenation_bitwise():
    """prn_sigmoid_point_nms against sigmoid -> max_pool2d(2, stride 1, pad 1)[:-1, :-1] -> eq -> mul -> permute -> cat evaluated with
    torch on the device (reference planerecnet.py:113, models/functions/nms.py:8-12), incl. ties between neighbouring cells."""
    from planerecnet_amd import metrics
    from planerecnet_amd.nms import point_nms
    g = torch.Generator().manual_seed(3)
    B, C = 3, 2
    levels = []
    for S in (40, 36, 24, 16, 12):
        x = torch.randn(B, C, S, S, generator=g) * 3
        x[:, :, 2:4, 5:7] = 1.25                                    # a plateau: equal scores in one window
        levels.append(x.cuda())
    got = metrics.category_scores(levels)
    want = torch.cat([point_nms(c.sigmoid(), kernel=2).permute(0, 2, 3, 1).reshape(B, -1, C) for c in levels], 1)
    assert got.shape == want.shape
    assert torch.equal(got, want), float((got - want).abs().max())

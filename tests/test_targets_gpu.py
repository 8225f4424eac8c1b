"""GPU: the device path of the loss's GT-only preparation (planerecnet_amd/targets.py, csrc/prn_targets.hip) -- SOLOv2 target assignment
(reference models/functions/losses.py:200-286) and virtual-normal triplet sampling (models/functions/vnl.py:43-70,119-140).

* region statistics (pixel count, sum x, sum y, per-segment prefix) and the 1/4 masks: exact against numpy on irregular masks, incl. an
  empty mask, overlapping masks and an image without planes;
* with INJECTED ranks (the reference's numpy stream drawn by the caller) every tensor of the resulting Targets -- cell lists, instance
  labels, category labels, triplet pixel ids, segment bookkeeping -- is BIT-IDENTICAL to the host path (losses.PlaneRecNetLoss.prepare,
  itself pinned to the reference's golden targets in test_model_gpu.py), and so is the loss;
* with the device sampler (Philox) every triplet point lies inside its region, every region gets int(0.3 * pixels) triplets, the draws
  are uniform over the region and reproducible, and a training loss on them is finite and close to the host-sampled one.
"""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _irregular_batch(seed=0, H=480, W=640):
    """Three images: blobs of random rows / columns (not rectangles), one EMPTY mask, overlapping masks, an image without planes."""
    rng = np.random.RandomState(seed)
    inst = []
    for b, n in enumerate((5, 0, 3)):
        masks = np.zeros((n, H, W), np.uint8)
        boxes = np.zeros((n, 4), np.float64)
        for i in range(n):
            y0, x0 = rng.randint(0, H - 60), rng.randint(0, W - 80)
            h, w = rng.randint(20, min(300, H - y0)), rng.randint(20, min(400, W - x0))
            blob = (rng.rand(h, w) < 0.7).astype(np.uint8)
            blob[rng.rand(h) < 0.2] = 0                               # holes: whole rows / columns missing
            blob[:, rng.rand(w) < 0.2] = 0
            if not (b == 0 and i == 2):                               # image 0, mask 2 stays empty
                masks[i, y0:y0 + h, x0:x0 + w] = blob
            boxes[i] = (x0, y0, x0 + w, y0 + h)
        nrm = rng.randn(n, 3)
        nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-9)
        inst.append({"masks": torch.from_numpy(masks), "boxes": torch.from_numpy(boxes), "classes": torch.zeros(n, dtype=torch.int64),
                     "plane_paras": torch.from_numpy(np.concatenate([nrm, rng.rand(n, 1) * 3, np.zeros((n, 2))], 1)),
                     "k_matrix": torch.tensor([[577.0, 0, W / 2], [0, 577.0, H / 2], [0, 0, 1]], dtype=torch.float64)})
    return inst


def test_region_statistics_and_quarter_masks_exact():
    from planerecnet_amd import ops
    from planerecnet_amd.funcs import quarter_mask_u8
    lib, _p, _s = ops.lib, ops._p, ops._stream
    H, W = 480, 640
    inst = _irregular_batch(1)
    N_per = [int(g["masks"].shape[0]) for g in inst]
    Ntot, B = sum(N_per), len(inst)
    first = np.concatenate([[0], np.cumsum(N_per)]).astype(np.int32)
    packed = torch.cat([g["masks"].reshape(-1) for g in inst]).cuda()
    nseg = int(lib.prn_gt_segments(H, W))
    R = Ntot + B
    segcnt = torch.empty(R * nseg, dtype=torch.uint8, device="cuda")
    segstart = torch.empty(R * (nseg + 1), dtype=torch.int32, device="cuda")
    totals = torch.empty(R * 3, dtype=torch.int64, device="cuda")
    first_d = torch.from_numpy(first).cuda()
    ops.check(lib.prn_gt_mask_stats(_p(packed), _p(first_d), B, Ntot, H, W, _p(segcnt), _p(segstart), _p(totals), _s()), "stats")
    small = torch.empty(Ntot, H // 4, W // 4, dtype=torch.uint8, device="cuda")
    ops.check(lib.prn_gt_quarter_masks(_p(packed), _p(small), Ntot, H, W, _s()), "quarter")
    torch.cuda.synchronize()
    tot = totals.cpu().numpy().reshape(R, 3)
    ss = segstart.cpu().numpy().reshape(R, nseg + 1)
    ys, xs = np.mgrid[0:H, 0:W]
    allm = np.concatenate([g["masks"].numpy() for g in inst], 0).astype(bool)
    regions = [allm[i] for i in range(Ntot)] + [~allm[first[b]:first[b + 1]].any(0) if N_per[b] else np.ones((H, W), bool) for b in range(B)]
    for r, m in enumerate(regions):
        assert tuple(tot[r]) == (int(m.sum()), int(xs[m].sum()), int(ys[m].sum())), r
        want = np.concatenate([[0], np.cumsum(m.reshape(-1, 64).sum(1))])
        assert np.array_equal(ss[r], want), r
    assert torch.equal(small.cpu(), quarter_mask_u8(torch.from_numpy(allm.astype(np.uint8))))


@pytest.fixture(scope="module")
def crit():
    from planerecnet_amd.config import set_cfg
    from planerecnet_amd.losses import PlaneRecNetLoss
    set_cfg("PlaneRecNet_50_config")
    return PlaneRecNetLoss().cuda()


def _targets_equal(a, b):
    for k in ("B", "n_pos", "num_ins", "cells_unique"):
        assert getattr(a, k) == getattr(b, k), k
    for k in ("cell_gidx", "cell_inv", "n_pos_dev", "pos_img", "ins_labels", "cate_labels", "lava_adj", "lava_gsum"):
        x, y = getattr(a, k), getattr(b, k)
        assert (x is None) == (y is None), k
        if x is not None:
            assert x.dtype == y.dtype and torch.equal(x, y), k
    assert len(a.cell_ids) == len(b.cell_ids) and all(torch.equal(x, y) for x, y in zip(a.cell_ids, b.cell_ids))
    va, vb = a.vnl, b.vnl
    for k in ("B", "n_seg", "n_tot"):
        assert getattr(va, k) == getattr(vb, k), k
    for k in ("N", "fx", "fy", "gid", "seg", "seg_start", "seg_img", "seg_is_plane", "seg_normal", "gid_flat", "gid_order", "gid32", "gid_start"):
        x, y = getattr(va, k), getattr(vb, k)
        assert (x is None) == (y is None), k
        if x is not None:
            assert torch.equal(x, y), k


@pytest.mark.parametrize("batch", ["synthetic", "irregular"])
def test_injected_ranks_reproduce_the_host_targets_bit_for_bit(crit, batch):
    from oracle import synth
    from planerecnet_amd.targets import DeviceTargetBuilder
    if batch == "synthetic":
        _, inst, gtd = synth.make_batch(3, 480, 640, seed=5)
    else:
        inst = _irregular_batch(7)
        gtd = 0.5 + 4.0 * torch.rand(3, 1, 480, 640, generator=torch.Generator().manual_seed(1))
    gtd = gtd.cuda()
    np.random.seed(99)
    host = crit.prepare(inst, gtd, torch.device("cuda"))
    host2 = crit.prepare(inst, gtd, torch.device("cuda"))                 # second batch: the numpy stream continues
    tb = DeviceTargetBuilder(crit, sampler="numpy")
    np.random.seed(99)
    tb.submit(inst, (480, 640))
    tb.submit(inst, (480, 640))                                           # two batches in flight, draws happen at get() in FIFO order
    d1 = tb.get(gtd, torch.device("cuda"))
    d2 = tb.get(gtd, torch.device("cuda"), overlap=True)
    d2.ready.synchronize()
    torch.cuda.synchronize()
    _targets_equal(host, d1)
    _targets_equal(host2, d2)
    assert host.vnl.n_tot > 100000


def test_device_sampler_draws_uniformly_inside_every_region(crit):
    from planerecnet_amd.targets import DeviceTargetBuilder
    inst = _irregular_batch(3)
    gtd = (0.5 + 4.0 * torch.rand(3, 1, 480, 640, generator=torch.Generator().manual_seed(2))).cuda()
    H, W = 480, 640

    def run(seed):
        tb = DeviceTargetBuilder(crit, sampler="philox", seed=seed)
        tb.submit(inst, (H, W))
        t = tb.get(gtd, torch.device("cuda"))
        torch.cuda.synchronize()
        return t
    t = run(4)
    v = t.vnl
    allm = [g["masks"].numpy().astype(bool) for g in inst]
    gid = v.gid.cpu().numpy()                                             # [3, n_tot]
    seg = v.seg.cpu().numpy()
    seg_img, is_plane = v.seg_img.cpu().numpy(), v.seg_is_plane.cpu().numpy()
    g = 0
    for b in range(3):
        N = allm[b].shape[0]
        regions = [allm[b][i] for i in range(N)]
        nonplanar = ~allm[b].any(0) if N else np.ones((H, W), bool)
        if nonplanar.sum() > 0:
            regions.append(nonplanar)
        for r, m in enumerate(regions):
            sel = seg == g
            assert seg_img[g] == b and bool(is_plane[g]) == (r < N)
            assert int(sel.sum()) == int(int(m.sum()) * 0.3), (b, r)
            if sel.any():
                px = gid[:, sel] - b * H * W
                assert px.min() >= 0 and px.max() < H * W and m.reshape(-1)[px.reshape(-1)].all(), (b, r)       # every point lies in its region
                if sel.sum() > 2000:                                       # uniform over the region: mean rank ~ half, three draws uncorrelated
                    rank = np.cumsum(m.reshape(-1))[px] - 1
                    u = rank / float(m.sum())
                    assert abs(u.mean() - 0.5) < 0.03 and abs(np.corrcoef(u[0], u[1])[0, 1]) < 0.08, (b, r, u.mean())
            g += 1
    assert g == v.n_seg
    assert torch.equal(run(4).vnl.gid, v.gid)                              # reproducible for a given (seed, call number)
    assert not torch.equal(run(5).vnl.gid, v.gid)


def test_loss_on_device_targets(crit):
    """The joint loss on device-built targets: with injected ranks equal to the host path's value (same triplets), with the device sampler
    finite and within a few per cent of it (the plane term is a mean over ~10^5 random triplets)."""
    from oracle import synth
    from planerecnet_amd.targets import DeviceTargetBuilder
    _, inst, gtd = synth.make_batch(2, 480, 640, seed=8)
    gtd = gtd.cuda()
    g = torch.Generator().manual_seed(0)
    S = crit.num_grids
    mask = torch.randn(2, 128, 120, 160, generator=g).cuda() * 0.3
    cate = [torch.randn(2, crit.num_classes, s, s, generator=g).cuda() - 2.0 for s in S]
    kern = [torch.randn(2, 128, s, s, generator=g).cuda() * 0.2 for s in S]
    half = gtd[:, :, ::2, ::2]
    depth = (half * (1 + 0.1 * torch.randn(half.shape, generator=g).cuda())).abs().contiguous()

    def loss_with(t):
        out = crit(None, mask, cate, kern, depth, inst, gtd, targets=t)
        return {k: float(v.sum()) for k, v in out.items()}
    np.random.seed(3)
    ref = loss_with(crit.prepare(inst, gtd, torch.device("cuda")))
    tb = DeviceTargetBuilder(crit, sampler="numpy")
    np.random.seed(3)
    tb.submit(inst, (480, 640))
    inj = loss_with(tb.get(gtd, torch.device("cuda")))
    assert inj == ref, (inj, ref)
    tp = DeviceTargetBuilder(crit, sampler="philox", seed=1)
    tp.submit(inst, (480, 640))
    ph = loss_with(tp.get(gtd, torch.device("cuda")))
    for k in ref:
        assert np.isfinite(ph[k])
        if k != "pln":
            assert ph[k] == ref[k], k                                       # only the plane term depends on the draws
    assert abs(ph["pln"] - ref["pln"]) <= 0.05 * abs(ref["pln"]) + 1e-3, (ph["pln"], ref["pln"])


def test_batch_without_a_single_plane_and_resumed_sampler_stream(crit):
    """(1) A batch in which NO image has a plane (a real loader can produce one; the synthetic set never does): the device path must build
    targets -- no positive cell, only the non-planar regions sampled -- instead of failing on the empty mask tensor.  (2) The device sampler's
    stream is a function of (key, batch counter): a builder started at first_call = k reproduces the k-th batch of one started at 0 (a resumed
    run continues its stream), and two keys (ranks / run seeds) never share a batch."""
    from planerecnet_amd.targets import DeviceTargetBuilder
    H, W = 480, 640
    empty = [{"masks": torch.zeros(0, H, W, dtype=torch.uint8), "boxes": torch.zeros(0, 4, dtype=torch.float64), "classes": torch.zeros(0, dtype=torch.int64),
              "plane_paras": torch.zeros(0, 6, dtype=torch.float64), "k_matrix": torch.tensor([[577.0, 0, W / 2], [0, 577.0, H / 2], [0, 0, 1]], dtype=torch.float64)}
             for _ in range(2)]
    gtd = (0.5 + 4.0 * torch.rand(2, 1, H, W, generator=torch.Generator().manual_seed(3))).cuda()
    tb = DeviceTargetBuilder(crit, sampler="philox", seed=1)
    tb.submit(empty, (H, W))
    t = tb.get(gtd, torch.device("cuda"))
    torch.cuda.synchronize()
    assert t.vnl.n_seg == 2 and t.vnl.n_tot == 2 * int(H * W * 0.3) and not bool(t.vnl.seg_is_plane.any())

    inst = _irregular_batch(3)
    gtd3 = (0.5 + 4.0 * torch.rand(3, 1, H, W, generator=torch.Generator().manual_seed(2))).cuda()

    def batches(seed, first, n):
        tb = DeviceTargetBuilder(crit, sampler="philox", seed=seed, first_call=first)
        out = []
        for _ in range(n):
            tb.submit(inst, (H, W))
            out.append(tb.get(gtd3, torch.device("cuda")).vnl.gid.clone())
        torch.cuda.synchronize()
        return out
    a = batches((7 << 32) | 0, 0, 3)
    assert torch.equal(batches((7 << 32) | 0, 2, 1)[0], a[2])             # resumed at iteration 2
    assert not torch.equal(a[0], a[1]) and not torch.equal(a[1], a[2])
    b = batches((7 << 32) | 1, 0, 2)                                       # the next rank
    assert all(not torch.equal(x, y) for x in a for y in b)

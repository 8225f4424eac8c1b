"""GPU parity of the assembled model + loss (HIP path, through the C ABI) against the CPU oracle
(oracle/model_ref.py, oracle/loss_ref.py -- proven equal to the real reference by tests/golden/make_golden.py)
on the same seeded weights and inputs.

Tolerances (fp32, stated per north_star): dense outputs 5e-4 max-abs relative to the tensor's max (a 50-layer fp32
chain with train-mode BatchNorm; on these well-conditioned seeded weights the oracle's own fp32-vs-fp64 spread is
~1e-5), losses rtol 1e-3 / atol 1e-4.  End-to-end parameter gradients: relative L2 <= 3e-2 and max-abs <= 1e-1 of the
gradient's max -- calibrated on the oracle itself, whose fp32-vs-fp64 gradients on this very step differ by 1e-2 (L2) and up
to 5e-2 (max-abs) for backbone / decoder parameters (ReLU / max-pool / bilinear-kink masks flip under 1-ulp perturbations);
every backward KERNEL is checked tightly (2e-4..5e-4) in tests/test_ops_gpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CN = "PlaneRecNet_50_config"


def close(got, ref, rtol, what):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    den = ref.abs().max().item() + 1e-12
    assert err <= rtol * den, f"{what}: max-abs {err:.3e} vs scale {den:.3e} (rel {err / den:.2e})"


@pytest.fixture(scope="module")
def setup():
    from oracle import model_ref, synth
    from planerecnet_amd.config import cfg, set_cfg
    from planerecnet_amd.planerecnet import PlaneRecNet
    set_cfg(CN)
    sd = synth.make_state_dict(CN, seed=1)
    net = PlaneRecNet(cfg)
    net.load_state_dict(sd)                       # strict: identical keys/shapes to the reference layout
    net = net.cuda()
    return net, sd, model_ref.ARCH[CN]


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_forward_matches_oracle(setup, mode, gemm_arith):
    from oracle import model_ref, synth
    net, sd, arch = setup
    net.load_state_dict(sd)
    x, _, _ = synth.make_batch(2, 128, 160, seed=2)
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.train(mode == "train")
    with torch.no_grad():
        mask, cate, kern, depth = net(x.cuda())
        o_mask, o_cate, o_kern, o_depth = model_ref.forward(sd, x, arch, training=(mode == "train"))
    close(mask, o_mask, 5e-4, "mask_pred")
    close(depth, o_depth, 5e-4, "depth_pred")
    for i in range(4):
        close(cate[i], o_cate[i], 5e-4, f"cate{i}")
        close(kern[i], o_kern[i], 5e-4, f"kernel{i}")
    if mode == "train":                           # running statistics were updated like nn.BatchNorm2d would
        sd2 = {k: v.clone() for k, v in sd.items()}
        with torch.no_grad():
            model_ref.forward(sd2, x, arch, training=True, update_stats=True)
        got = net.state_dict()
        for k in ("backbone.bn1.running_mean", "backbone.layers.2.3.bn2.running_var", "depth_decoder.deconv4.3.running_var"):
            close(got[k], sd2[k], 2e-3, k)


def test_batchnorm_bookkeeping_and_fold_cache(setup):
    """Training-mode forwards advance num_batches_tracked like nn.BatchNorm2d and bump the running statistics' version
    counters, so the inference path's folded conv+BN weights (cached on those versions) are rebuilt after the statistics move."""
    from oracle import synth
    net, sd, _ = setup
    net.load_state_dict(sd)
    x, _, _ = synth.make_batch(2, 128, 160, seed=2)
    xd = x.cuda()
    net.eval()
    with torch.no_grad():
        before = [t.clone() for t in net.backbone(xd)]                   # folded path, caches the folded weights
    net.train()
    v0 = net.backbone.bn1.running_mean._version
    with torch.no_grad():
        net(xd)
        net(xd)
    assert net.backbone.bn1.running_mean._version > v0
    # a SUB-module's state dict is as current as the root's (the counters are flushed by a pre-hook on every BatchNorm module)
    assert int(net.backbone.state_dict()["layers.0.0.bn1.num_batches_tracked"]) == 2
    assert int(net.depth_decoder.deconv3[3].state_dict()["num_batches_tracked"]) == 2 if hasattr(net.depth_decoder.deconv3[3], "num_batches_tracked") else True
    got = net.state_dict()
    assert int(got["backbone.bn1.num_batches_tracked"]) == 2 and int(got["depth_decoder.deconv4.3.num_batches_tracked"]) == 2
    assert int(net.state_dict()["backbone.layers.1.0.bn2.num_batches_tracked"]) == 2          # (flushing twice does not double count)
    net.eval()
    with torch.no_grad():
        after = [t.clone() for t in net.backbone(xd)]                    # folded with the NEW statistics
    with torch.enable_grad():
        unfolded = net.backbone(xd)                                      # autograd on: separate conv + BatchNorm launches
    assert float((after[0] - before[0]).abs().max()) > 1e-4               # the statistics did move
    for a, u in zip(after, unfolded):
        close(a, u, 2e-4, "folded vs unfolded after a statistics update")
    net.load_state_dict(sd)


def test_weight_caches_follow_the_one_launch_adam(setup):
    """FusedAdam writes the parameters through raw pointers: it must advance their version counters, or the eval path would go on
    using conv+BN weights folded before the update (validation passes between training epochs).  Eval output after an update ==
    eval output of the same weights with every cache keyed from scratch (separate conv + BatchNorm launches)."""
    from oracle import synth
    from planerecnet_amd.optim import FusedAdam
    net, sd, _ = setup
    net.load_state_dict(sd)
    x, _, _ = synth.make_batch(1, 128, 160, seed=5)
    xd = x.cuda()
    net.eval()
    with torch.no_grad():
        before = [t.clone() for t in net.backbone(xd)]                   # caches the folded weights
    opt = FusedAdam(net.backbone.parameters(), lr=1e-2)
    for p in net.backbone.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    with torch.no_grad():
        after = [t.clone() for t in net.backbone(xd)]                    # folded again from the updated weights
    with torch.enable_grad():
        unfolded = net.backbone(xd)
    assert float((after[-1] - before[-1]).abs().max()) > 1e-3
    for a, u in zip(after, unfolded):
        close(a, u, 5e-4, "folded (cached) vs unfolded after an optimizer step")
    for p in net.backbone.parameters():
        p.grad = None
    net.load_state_dict(sd)


def test_backward_with_ragged_heads_and_deferred_wgrads(setup):
    """B=4 at 128x160 (the batch sizes for which the SOLO grid levels pack into whole GEMM tiles): the instance head runs as
    ragged batches, weight gradients are deferred to the side stream, backbone-feature gradients meet in forked epilogues.
    Outputs and parameter gradients of a scalar test loss against the oracle's autograd."""
    from oracle import model_ref, synth
    from planerecnet_amd import ops
    net, sd, arch = setup
    net.load_state_dict(sd)
    net.train()
    x, _, _ = synth.make_batch(4, 128, 160, seed=5)
    assert ops.RaggedShape(4, [(g, g) for g in net.inst_head.num_grids]).supported()

    def scalar(mask, cate, kern, depth):
        return mask.square().mean() + depth.mean() + sum(c.square().mean() for c in cate) + sum(k.square().mean() for k in kern)

    ops.set_wgrad_async(True)
    try:
        net.zero_grad(set_to_none=True)
        out = net(x.cuda())
        scalar(*out).backward()
        ops.wgrad_join()
    finally:
        ops.set_wgrad_async(False)
    torch.cuda.synchronize()
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    oo = model_ref.forward(sdg, x, arch, training=True)
    close(out[0], oo[0], 5e-4, "mask_pred")
    close(out[3], oo[3], 5e-4, "depth_pred")
    for i in range(len(oo[1])):
        close(out[1][i], oo[1][i], 5e-4, f"cate{i}")
        close(out[2][i], oo[2][i], 5e-4, f"kernel{i}")
    names = ["backbone.conv1.weight", "backbone.layers.1.0.conv1.weight", "backbone.layers.1.0.downsample.0.weight",
             "backbone.layers.2.5.conv3.weight", "fpn.lateral_convs.1.weight", "fpn.fpn_convs.0.weight",
             "inst_head.kernel_tower.0.weight", "inst_head.kernel_tower.4.weight", "inst_head.cate_tower.3.weight",
             "inst_head.cate_pred.weight", "inst_head.cate_pred.bias", "inst_head.kernel_pred.weight",
             "mask_head.conv_pred.0.weight", "depth_decoder.latlayer2.weight", "depth_decoder.deconv3.2.weight"]
    grads = torch.autograd.grad(scalar(*oo), [sdg[n] for n in names])
    params = dict(net.named_parameters())
    for n, g in zip(names, grads):
        got = params[n].grad.detach().double().cpu()
        l2 = ((got - g.double()).norm() / (g.double().norm() + 1e-30)).item()
        assert l2 <= 3e-2, f"grad {n}: relative L2 {l2:.2e}"


def test_inference_matches_oracle(setup):
    from oracle import model_ref, synth
    net, sd, arch = setup
    sd_inf = dict(sd)
    sd_inf["inst_head.cate_pred.bias"] = sd["inst_head.cate_pred.bias"] + 1.0
    net.load_state_dict(sd_inf)
    net.eval()
    x, _, _ = synth.make_batch(1, 128, 160, seed=3)
    with torch.no_grad():
        res = net(x.cuda())[0]
    ref = model_ref.inference(sd_inf, x, arch)[0]
    assert res["pred_scores"] is not None and len(res["pred_scores"]) == len(ref["pred_scores"])
    close(res["pred_scores"], ref["pred_scores"], 5e-3, "scores")
    assert torch.equal(res["pred_classes"].cpu(), ref["pred_classes"])
    assert (res["pred_boxes"] - ref["pred_boxes"]).abs().max() <= 2.0
    assert not res["pred_boxes"].is_cuda                                  # quirk Q11
    close(res["pred_depth"], ref["pred_depth"], 2e-3, "pred_depth")
    assert (res["pred_masks"].cpu() != ref["pred_masks"]).float().mean() < 2e-3
    net.load_state_dict(sd)


def test_full_size_inference_matches_oracle(setup, gemm_arith):
    """The inference workloads bench.py times (c2 / c5) at their real frame size: PlaneRecNet_50, B = 2, 480x640, with the category
    bias conditioned so that every image keeps >= 5 detections through matrix NMS -- the post-process is checked where it is timed, on
    non-empty candidate sets.  Detections are matched by mask IoU (near-equal scores may swap places between two fp32
    implementations): same count, a one-to-one matching at IoU >= 0.98, scores 5e-3, boxes +-2 px, depth 2e-3 of its range."""
    from oracle import model_ref, synth
    net, sd, arch = setup
    x, _, _ = synth.make_batch(2, 480, 640, seed=31)
    ref = sd_inf = None
    for shift in (1.0, 1.5, 2.0, 2.5, 3.0):                              # the oracle picks the conditioning, the device path follows
        sd_try = dict(sd)
        sd_try["inst_head.cate_pred.bias"] = sd["inst_head.cate_pred.bias"] + shift
        r = model_ref.inference(sd_try, x, arch)
        if all(q["pred_scores"] is not None and len(q["pred_scores"]) >= 5 for q in r):
            ref, sd_inf = r, sd_try
            break
    assert ref is not None, "no bias shift gives >= 5 detections per image"
    net.load_state_dict(sd_inf)
    net.eval()
    try:
        with torch.no_grad():
            res = net(x.cuda())
        for b in range(2):
            g, o = res[b], ref[b]
            n = len(o["pred_scores"])
            assert g["pred_scores"] is not None and len(g["pred_scores"]) == n, (b, None if g["pred_scores"] is None else len(g["pred_scores"]), n)
            gm, om = g["pred_masks"].cpu().flatten(1).float(), o["pred_masks"].flatten(1).float()
            inter = om @ gm.t()
            iou = inter / (om.sum(1)[:, None] + gm.sum(1)[None, :] - inter).clamp_min(1.0)
            best = iou.argmax(1)
            assert sorted(best.tolist()) == list(range(n)), (b, best.tolist())           # a permutation
            assert float(iou.gather(1, best[:, None]).min()) >= 0.98, (b, iou.gather(1, best[:, None]).flatten().tolist())
            assert (g["pred_scores"].cpu()[best] - o["pred_scores"]).abs().max() <= 5e-3
            assert torch.equal(g["pred_classes"].cpu()[best], o["pred_classes"])
            assert (g["pred_boxes"][best] - o["pred_boxes"]).abs().max() <= 2.0
            close(g["pred_depth"], o["pred_depth"], 2e-3, "pred_depth")
    finally:
        net.load_state_dict(sd)


def test_batched_post_process_equals_image_by_image(setup):
    """PlaneRecNet.inference over a batch (candidate selection, mask statistics and the small-mask filter run once over all images)
    against the same images post-processed one at a time through inference_single_image: every output bit-identical, including an
    image without candidates and one whose candidates are all filtered out."""
    net, _, arch = setup
    g = torch.Generator().manual_seed(21)
    B, E, h, w = 4, net.inst_head.num_kernels, 60, 80
    C = net.inst_head.num_classes
    masks = (torch.randn(B, E, h, w, generator=g) * 0.5).cuda()
    masks[2] = 1.0                                                  # (with its kernels at -3: every soft mask of image 2 is empty)
    depth = (torch.rand(B, 1, 2 * h, 2 * w, generator=g) * 4 + 0.3).cuda()
    cates, kerns = [], []
    for S in arch.num_grids:
        c = torch.where(torch.rand(B, S, S, C, generator=g) < 0.02, 0.3 + 0.6 * torch.rand(B, S, S, C, generator=g), torch.zeros(B, S, S, C))
        c[1] = 0.0                                                  # image 1: no candidate at all
        cates.append(c.cuda())
        k = torch.randn(B, E, S, S, generator=g) * 0.3
        k[2] = -3.0                                                 # image 2: masks of sigmoid(very negative) -> empty -> all filtered out
        kerns.append(k.cuda())
    imgs = [torch.empty(3, 4 * h, 4 * w) for _ in range(B)]
    with torch.no_grad():
        batch = net.inference(masks, cates, kerns, depth, imgs)
        single = []
        for b in range(B):
            cate_b = torch.cat([c[b].reshape(-1, C) for c in cates], 0)
            kern_b = torch.cat([k[b].permute(1, 2, 0).reshape(-1, E) for k in kerns], 0)
            single.append(net.inference_single_image(masks[b:b + 1], cate_b, kern_b, depth[b:b + 1], (4 * h, 4 * w)))
    assert batch[1]["pred_scores"] is None and batch[2]["pred_scores"] is None
    assert batch[0]["pred_scores"] is not None and batch[3]["pred_scores"] is not None and len(batch[0]["pred_scores"]) > 1
    for rb, rs in zip(batch, single):
        for key in ("pred_masks", "pred_boxes", "pred_classes", "pred_scores", "pred_depth"):
            assert (rb[key] is None) == (rs[key] is None), key
            if rb[key] is not None:
                assert torch.equal(rb[key], rs[key]), key
        if rb["pred_boxes"] is not None:
            assert not rb["pred_boxes"].is_cuda and rb["pred_masks"].shape[1:] == (4 * h, 4 * w)
            assert (rb["pred_scores"][:-1] >= rb["pred_scores"][1:]).all()


def _loss_inputs(arch):
    from oracle import synth
    g = torch.Generator().manual_seed(5)
    B = 2
    _, inst, gtd = synth.make_batch(B, 480, 640, seed=4)
    mask_pred = torch.randn(B, 128, 120, 160, generator=g).relu_()
    cate = [torch.randn(B, 2, s, s, generator=g) - 2.0 for s in arch.num_grids]
    kern = [torch.randn(B, 128, s, s, generator=g) * 0.1 for s in arch.num_grids]
    depth = torch.rand(B, 1, 240, 320, generator=g) * 4 + 0.3
    return mask_pred, cate, kern, depth, inst, gtd


def test_loss_matches_oracle_and_golden(setup, golden_dir):
    """Loss on synthetic predictions: device path vs oracle AND vs the committed golden values of the real reference."""
    import os
    from oracle import loss_ref
    from planerecnet_amd.losses import PlaneRecNetLoss
    _, _, arch = setup
    mask_pred, cate, kern, depth, inst, gtd = _loss_inputs(arch)
    cpu_leaves = [mask_pred] + cate + kern + [depth]
    for t in cpu_leaves:
        t.requires_grad_(True)
    np.random.seed(7)
    ol = loss_ref.joint_loss(mask_pred, cate, kern, depth, inst, gtd)
    og = torch.autograd.grad(sum(ol.values()).sum(), cpu_leaves, allow_unused=True)

    dl = [t.detach().cuda().requires_grad_(True) for t in cpu_leaves]
    crit = PlaneRecNetLoss().cuda()
    np.random.seed(7)
    inst_d = [{k: v.cuda() for k, v in g.items()} for g in inst]
    out = crit(None, dl[0], dl[1:5], dl[5:9], dl[9], inst_d, gtd.cuda())
    fx = np.load(os.path.join(golden_dir, "loss_synth.npz"))
    for k in ("ins", "cat", "dpt", "pln", "lav"):
        assert abs(float(out[k]) - float(ol[k])) <= 1e-3 * abs(float(ol[k])) + 1e-4, (k, float(out[k]), float(ol[k]))
        assert abs(float(out[k]) - float(fx[k])) <= 1e-3 * abs(float(fx[k])) + 1e-4, ("golden", k)
    assert out["pln"].dtype == torch.float64
    dg = torch.autograd.grad(sum(out.values()).sum(), dl, allow_unused=True)
    for i, (a, b) in enumerate(zip(dg, og)):
        if b is None:                       # a level without positive cells: the batched path yields an all-zero gradient
            assert a is None or float(a.abs().max()) == 0.0, i
        else:
            close(a, b, 2e-3, f"loss grad {i}")


def test_kernel_gather_branches_agree(setup):
    """The Dice term gathers the predicted kernels of all positive cells with ONE index_select when no cell is listed twice, and
    image by image (advanced indexing, sort-based gradient) otherwise: same loss, same gradients."""
    from planerecnet_amd.losses import PlaneRecNetLoss
    _, _, arch = setup
    mask_pred, cate, kern, depth, inst, gtd = _loss_inputs(arch)
    crit = PlaneRecNetLoss().cuda()
    inst_d = [{k: v.cuda() for k, v in g.items()} for g in inst]
    res = []
    for unique in (True, False):
        dl = [t.detach().cuda().requires_grad_(True) for t in [mask_pred] + cate + kern + [depth]]
        np.random.seed(7)
        t = crit.prepare(inst_d, gtd.cuda(), torch.device("cuda"))
        assert t.cells_unique                                 # (no cell claimed by three instances; doubly claimed ones are expanded)
        t.cells_unique = unique
        out = crit(None, dl[0], dl[1:5], dl[5:9], dl[9], inst_d, gtd.cuda(), targets=t)
        g = torch.autograd.grad(out["ins"].sum() + out["lav"].sum(), [dl[0]] + dl[5:9], allow_unused=True)
        res.append((out["ins"].detach(), out["lav"].detach(), g))
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-6) and torch.allclose(res[0][1], res[1][1], rtol=1e-6)
    for a, b in zip(res[0][2], res[1][2]):
        assert (a is None) == (b is None)
        if a is not None:
            close(a, b, 1e-5, "kernel gather branches")


def test_fused_focal_and_rmse_log_equal_operator_chains():
    """prn_focal_sum / prn_rmse_log (one pass each way) against the operator chains they replace (losses.sigmoid_focal_sum, rmse_log:
    the reference's formulas, pinned through the whole-loss tests): value and gradient, incl. background rows, pixels without ground
    truth and a non-positive prediction (below the clamp: zero gradient)."""
    import torch.nn.functional as F
    from planerecnet_amd import losses as L
    g = torch.Generator().manual_seed(0)
    rows, C = 29824, 2
    x = (torch.randn(rows, C, generator=g) * 3).cuda()
    labels = torch.full((rows,), C, dtype=torch.int64)
    pos = torch.randperm(rows, generator=g)[:600]
    labels[pos] = torch.randint(0, C, (600,), generator=g)
    labels = labels.cuda()
    for alpha, gamma in ((0.25, 2.0), (-1.0, 1.5)):
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        a = L._FocalSum.apply(xa, labels, alpha, gamma)
        onehot = F.one_hot(labels, C + 1)[:, :C].float()
        b = L.sigmoid_focal_sum(xb, onehot, alpha, gamma)
        assert abs(float(a) - float(b)) <= 1e-5 * abs(float(b))
        (ga,), (gb,) = torch.autograd.grad(a * 0.37, xa), torch.autograd.grad(b * 0.37, xb)
        close(ga, gb, 1e-5, "focal grad")
    B, H, W = 3, 96, 128
    gt = (torch.rand(B, 1, H, W, generator=g) * 5).cuda()
    gt[:, :, :10] = 0.0                                             # no ground truth there
    pred = (gt * (1 + 0.3 * torch.randn(B, 1, H, W, generator=g).cuda())).abs() + 0.05
    pred[0, 0, 50, 50] = 0.0                                        # below the clamp
    pa, pb = pred.clone().requires_grad_(True), pred.clone().requires_grad_(True)
    a = L._RmseLog.apply(pa, gt, 0.1, 1e-9)
    b = L.rmse_log(pb, gt, gt > 0.1)
    assert abs(float(a) - float(b)) <= 1e-5 * abs(float(b))
    (ga,), (gb,) = torch.autograd.grad(a * 5.0, pa), torch.autograd.grad(b * 5.0, pb)
    assert float(ga[0, 0, 50, 50]) == 0.0 and float(ga[:, :, :10].abs().max()) == 0.0
    close(ga, gb, 1e-5, "rmse-log grad")


def test_trimming_rule_kernels_equal_operator_chain():
    """prn_vnl_trim_* (sort key, region sums, per-image combination, scattered gradient) against the operator chain
    (VNL_Loss._trimmed_means_autograd, itself pinned through the oracle's per-plane loops): ties, NaN losses, invalid triplets,
    a region without any valid triplet (0 / 0 like the reference, with a finite gradient for the other image)."""
    from planerecnet_amd.losses import VNL_Loss, VNLTargets
    rng = np.random.RandomState(11)
    seg_len = np.array([4000, 7, 12000, 3, 640, 1600, 300])
    n = int(seg_len.sum())
    t = VNLTargets()
    t.B, t.n_seg, t.n_tot = 2, len(seg_len), n
    t.seg = torch.from_numpy(np.repeat(np.arange(len(seg_len)), seg_len)).cuda()
    t.seg_start = torch.from_numpy(np.concatenate([[0], np.cumsum(seg_len)[:-1]])).cuda()
    t.seg_img = torch.tensor([0, 0, 0, 1, 1, 1, 1]).cuda()
    t.seg_is_plane = torch.tensor([True, True, False, True, True, True, False]).cuda()
    t.N = torch.tensor([2.0, 3.0], dtype=torch.float64).cuda()
    loss = torch.from_numpy(np.round(rng.rand(n), 3)).double().cuda()
    loss[5], loss[4100] = float("nan"), float("nan")
    valid = torch.from_numpy(rng.rand(n) < 0.8).cuda()
    for degenerate in (False, True):
        if degenerate:
            valid[t.seg == 3] = False
        la, lb = loss.clone().requires_grad_(True), loss.clone().requires_grad_(True)
        ya = VNL_Loss._trimmed_means(la, valid, t, loss.device)
        yb = VNL_Loss._trimmed_means_autograd(lb, valid, t, loss.device)
        assert torch.allclose(ya, yb, rtol=1e-12, atol=0, equal_nan=True) and bool(torch.isnan(ya[1])) == degenerate
        (ga,) = torch.autograd.grad(ya[0] * 0.7, la, retain_graph=True)
        assert bool(torch.isfinite(ga).all()) and float(ga[5]) == 0.0
        if not degenerate:
            w = torch.tensor([0.7, 1.3], dtype=torch.float64, device=loss.device)
            (ga,), (gb,) = torch.autograd.grad((ya * w).sum(), la), torch.autograd.grad((yb * w).sum(), lb)
            assert torch.allclose(ga, gb, rtol=1e-12, atol=0) and float(ga.abs().sum()) > 0


def test_gt_assignment_bit_exact_vs_golden(setup, golden_dir):
    import os
    from oracle import synth
    from planerecnet_amd.losses import PlaneRecNetLoss
    fx = np.load(os.path.join(golden_dir, "loss_synth.npz"))
    _, inst, _ = synth.make_batch(2, 480, 640, seed=4)
    ins, cate, ind, order = PlaneRecNetLoss().prepare_ground_truth({k: v.cuda() for k, v in inst[0].items()}, (120, 160))
    for lv in range(4):
        assert np.array_equal(cate[lv].numpy(), fx[f"tg_cate{lv}"])
        assert np.array_equal(np.asarray(order[lv], dtype=np.int64), fx[f"tg_order{lv}"])
        assert np.array_equal(ins[lv].sum((1, 2)).numpy(), fx[f"tg_ins_area{lv}"])


@pytest.mark.parametrize("wgrad_async", [False, True])
def test_e2e_train_step_matches_oracle(setup, golden_dir, wgrad_async, gemm_arith):
    """R50, 480x640, B=1: forward + joint loss + backward. Losses vs oracle and vs the reference's golden values;
    parameter gradients vs oracle autograd.  Run with the weight gradients in line and deferred to the side stream
    (ops.set_wgrad_async, the mode train.py / bench.py use)."""
    import os
    from oracle import loss_ref, model_ref, synth
    from planerecnet_amd import ops
    from planerecnet_amd.losses import PlaneRecNetLoss
    net, sd, arch = setup
    ops.set_wgrad_async(wgrad_async)
    net.load_state_dict(sd)
    net.train()
    x, inst, gtd = synth.make_batch(1, 480, 640, seed=6)
    crit = PlaneRecNetLoss().cuda()
    np.random.seed(11)
    out = net(x.cuda())
    losses = crit(net, *out, [{k: v.cuda() for k, v in g.items()} for g in inst], gtd.cuda())
    total = sum(losses.values()).sum()
    net.zero_grad()
    try:
        total.backward()
        ops.wgrad_join()
    finally:
        ops.set_wgrad_async(False)
    torch.cuda.synchronize()
    fx = np.load(os.path.join(golden_dir, "e2e_r50_480x640.npz"))
    for k in ("ins", "cat", "dpt", "pln", "lav"):
        assert abs(float(losses[k]) - float(fx[k])) <= 1e-3 * abs(float(fx[k])) + 1e-4, (k, float(losses[k]), float(fx[k]))

    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    np.random.seed(11)
    oo = model_ref.forward(sdg, x, arch, training=True)
    ol = loss_ref.joint_loss(*oo, inst, gtd)
    for k in ol:
        assert abs(float(losses[k]) - float(ol[k])) <= 1e-3 * abs(float(ol[k])) + 1e-4, (k, float(losses[k]), float(ol[k]))
    names = ["backbone.conv1.weight", "backbone.layers.1.0.conv2.regular_conv.weight", "backbone.layers.1.0.conv2.offset_conv.weight",
             "backbone.layers.2.5.conv3.weight", "backbone.layers.3.2.bn3.weight", "fpn.lateral_convs.2.weight", "fpn.fpn_convs.0.bias",
             "inst_head.kernel_tower.0.weight", "inst_head.cate_pred.bias", "mask_head.convs_all_levels.3.conv0.0.weight",
             "mask_head.conv_pred.1.weight", "depth_decoder.conv1x1.0.weight", "depth_decoder.deconv4.2.weight",
             "depth_decoder.refine_conv.2.bias", "depth_decoder.depth_pred.1.weight"]
    grads = torch.autograd.grad(sum(ol.values()).sum(), [sdg[n] for n in names])
    params = dict(net.named_parameters())
    for n, g in zip(names, grads):
        got = params[n].grad.detach().double().cpu()
        l2 = ((got - g.double()).norm() / g.double().norm()).item()
        assert l2 <= 3e-2, f"grad {n}: relative L2 {l2:.2e}"
        close(params[n].grad, g, 1e-1, "grad " + n)


def test_r101_highres_inference_shapes_match_oracle():
    """BASELINE config 5 shape family: PlaneRecNet_101, max_size=960 -> 736x960 input (C5 = 23x30, odd sizes at every
    pyramid level, H*W not a multiple of 4 on the coarse levels), eval-mode BatchNorm, batch 1."""
    from oracle import model_ref, synth
    from planerecnet_amd.config import cfg, set_cfg
    from planerecnet_amd.planerecnet import PlaneRecNet
    name = "PlaneRecNet_101_config"
    set_cfg(name)
    try:
        sd = synth.make_state_dict(name, seed=3)
        net = PlaneRecNet(cfg)
        net.load_state_dict(sd)
        net = net.cuda().train()
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
        x, _, _ = synth.make_batch(1, 736, 960, seed=9)
        with torch.no_grad():
            mask, cate, kern, depth = net(x.cuda())
            o_mask, o_cate, o_kern, o_depth = model_ref.forward(sd, x, model_ref.ARCH[name], training=False)
        assert tuple(mask.shape) == (1, 128, 184, 240) and tuple(depth.shape) == (1, 1, 368, 480)
        close(mask, o_mask, 5e-4, "mask_pred 736x960")
        close(depth, o_depth, 5e-4, "depth_pred 736x960")
        for i in range(4):
            close(cate[i], o_cate[i], 5e-4, f"cate{i}")
            close(kern[i], o_kern[i], 5e-4, f"kernel{i}")
    finally:
        set_cfg(CN)


def test_prefetched_targets_equal_direct_preparation():
    """TargetPrefetcher (worker processes, packed pinned blob, uploads on the side stream with the `ready` event) hands the
    loss the same device tensors as PlaneRecNetLoss.prepare() computes in place -- and the loss values agree."""
    import numpy as np
    from planerecnet_amd.config import cfg, set_cfg
    from planerecnet_amd.losses import PlaneRecNetLoss, TargetPrefetcher
    from oracle import synth
    set_cfg("PlaneRecNet_50_config")
    dev = torch.device("cuda:0")
    crit = PlaneRecNetLoss().to(dev)
    _, inst, gtd = synth.make_batch(2, 480, 640, seed=11)
    gtd = gtd.to(dev)
    np.random.seed(3)
    direct = crit.prepare(inst, gtd, dev)
    np.random.seed(3)
    pf = TargetPrefetcher(crit)
    try:
        pf.submit(inst, (480, 640))
        t = pf.get(gtd, dev, overlap=True)
        assert t.ready is not None
        torch.cuda.current_stream().wait_event(t.ready)
        for name in ("pos_img", "ins_labels", "cate_labels", "n_pos_dev", "lava_adj", "lava_gsum"):
            assert torch.equal(getattr(t, name), getattr(direct, name)), name
        assert all(torch.equal(a, b) for a, b in zip(t.cell_ids, direct.cell_ids))
        for name in ("gid", "seg", "seg_start", "seg_img", "seg_is_plane", "seg_normal", "N", "fx", "fy"):
            assert torch.equal(getattr(t.vnl, name), getattr(direct.vnl, name)), name
    finally:
        pf.close()


def test_fused_virtual_normal_kernel_equals_operator_chain():
    """prn_vnl_triplets (+ forward-mode depth derivatives, prn_vnl_scatter) against the operator-by-operator evaluation of the
    same batched loss (itself equal to the oracle's per-plane loops: tests/test_host_cpu.py): per-image values and the gradient
    w.r.t. the predicted depth map, incl. exact zeros in the prediction (the non-planar zero fix of vnl.py:151)."""
    import numpy as np
    from oracle import synth
    from planerecnet_amd import losses
    from planerecnet_amd.config import set_cfg
    set_cfg("PlaneRecNet_50_config")
    dev = torch.device("cuda:0")
    crit = losses.PlaneRecNetLoss().to(dev)
    _, inst, gtd = synth.make_batch(3, 480, 640, seed=8)
    g = torch.Generator().manual_seed(3)
    pred = (gtd * (1 + 0.05 * torch.randn(gtd.shape, generator=g)) + 0.02).clamp(min=0.05)
    pred[0, 0, 100:110, 200:230] = 0.0                    # exact zeros: the zero-fix branch and |d| at its kink
    np.random.seed(21)
    t = crit.vnl.prepare(inst, (480, 640), dev)
    outs = {}
    for fused in (False, True):
        losses.FUSED_LOSS = fused
        try:
            p = pred.to(dev).requires_grad_(True)
            per_img = crit.vnl.batched(p, gtd.to(dev), t)
            w = torch.tensor([1.0, 0.7, 1.3], device=dev, dtype=per_img.dtype)
            gp, = torch.autograd.grad((per_img * w).sum(), p)
            outs[fused] = (per_img.detach().cpu(), gp.detach().cpu())
        finally:
            losses.FUSED_LOSS = True
    assert torch.allclose(outs[True][0], outs[False][0], rtol=1e-6, atol=1e-9), (outs[True][0], outs[False][0])
    a, b = outs[True][1].double(), outs[False][1].double()
    assert ((a - b).norm() / b.norm()).item() < 1e-4 and (a - b).abs().max().item() <= 1e-3 * b.abs().max().item()


def test_frozen_batchnorm_training_path_ignores_the_hand_overs(setup):
    """`freeze_bn` (models/planerecnet.py freeze_bn: BatchNorm modules in eval mode while the rest trains -- train.py switches it on for small batches): the
    backbone then runs operator by operator with eval-mode statistics.  The block entry points (training-mode statistics only, with their producer ->
    BatchNorm hand-overs) must not engage; the stage-output fork does.  Outputs and gradients must equal the path without the fork, and the running
    statistics must not move."""
    from planerecnet_amd import backbone as bb, blocks, ops
    from planerecnet_amd.config import cfg
    from planerecnet_amd.planerecnet import PlaneRecNet
    _, sd, _ = setup
    net = PlaneRecNet(cfg)                                         # (a fresh instance: the module-scoped one carries launch caches that do not deep-copy)
    net.load_state_dict(sd)
    net = net.cuda()
    net.train()
    net.freeze_bn()
    x = torch.randn(2, 3, 256, 320, generator=torch.Generator().manual_seed(3)).cuda()
    bns = [m for m in net.backbone.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    rm0 = [m.running_mean.clone() for m in bns]
    params = [p for p in net.backbone.parameters() if p.requires_grad]
    assert params and all(not m.training and not m.weight.requires_grad for m in bns)

    def run():
        before = dict(blocks.STATS)
        outs = net.backbone(x)
        loss = sum((o * o).mean() for o in outs)
        g = torch.autograd.grad(loss, params)
        ops.wgrad_join()
        took = {k: blocks.STATS[k] - before[k] for k in before}
        return [o.detach() for o in outs], [t.detach() for t in g], took

    saved = bb.STAGE_FORK
    try:
        bb.STAGE_FORK = False
        o0, g0, t0 = run()
        bb.STAGE_FORK = True
        o1, g1, t1 = run()
    finally:
        bb.STAGE_FORK = saved
    assert not any(t0.values()) and not any(t1.values()), (t0, t1)
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)                                    # the forward pass is the same sequence of launches
    for a, b in zip(g0, g1):
        close(b, a, 2e-4, "frozen-BN gradient with / without the stage fork")
    for m, r in zip(bns, rm0):
        assert torch.equal(m.running_mean, r)

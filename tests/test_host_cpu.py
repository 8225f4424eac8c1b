"""CPU: host-side logic that needs no GPU -- config surface, state-dict layout, checkpoint naming, target assignment."""
import os

import numpy as np
import torch


def test_config_surface_and_in_place_switch():
    from planerecnet_amd import config as C
    captured = C.cfg
    C.set_cfg("PlaneRecNet_101_config")
    assert captured is C.cfg and captured.name == "PlaneRecNet_101"
    assert captured.backbone.args == ([3, 4, 23, 3], [0, 4, 23, 3], 3) and captured.solov2.num_grids == [40, 36, 24, 16]
    assert captured.fpn.high_level_mode is None and captured.use_plane_loss and captured.lava_weight == 1.0
    C.set_cfg("PlaneRecNet_50_config")
    assert captured.name == "PlaneRecNet_50" and captured.backbone.args == ([3, 4, 6, 3], [0, 4, 6, 3])
    # shallow copy semantics (reference data/config.py:55-66): nested Config objects are shared
    a = C.PlaneRecNet_101_config.copy({"name": "x"})
    assert a.solov2 is C.PlaneRecNet_101_config.solov2 and a.name == "x" and C.PlaneRecNet_101_config.name == "PlaneRecNet_101"
    from planerecnet_amd.backbone import ResNetBackbone
    assert captured.backbone.type is ResNetBackbone


def test_state_dict_layout_matches_reference_spec():
    from oracle import synth
    from planerecnet_amd import config as C
    from planerecnet_amd.planerecnet import PlaneRecNet
    for cn in ("PlaneRecNet_50_config", "PlaneRecNet_101_config"):
        C.set_cfg(cn)
        net = PlaneRecNet(C.cfg)
        sd = net.state_dict()
        spec = synth.spec(cn)
        assert sorted(sd.keys()) == sorted(k for k, _, _ in spec)
        for k, shape, _ in spec:
            assert tuple(sd[k].shape) == tuple(shape), k
        net.load_state_dict(synth.make_state_dict(cn, seed=0))        # strict load of a reference-layout checkpoint
        # DCN offset / modulator convs are zero-initialised (models/dcn.py:32-43)
        fresh = PlaneRecNet(C.cfg).state_dict()
        assert all(float(v.abs().max()) == 0 for k, v in fresh.items() if "offset_conv" in k or "modulator_conv" in k)
    C.set_cfg("PlaneRecNet_50_config")


def test_dcn_placement_rule():
    from planerecnet_amd import config as C
    from planerecnet_amd.dcn import DeformableConv2d
    from planerecnet_amd.planerecnet import PlaneRecNet
    C.set_cfg("PlaneRecNet_101_config")
    net = PlaneRecNet(C.cfg)
    dcn = [n for n, m in net.named_modules() if isinstance(m, DeformableConv2d)]
    assert dcn == ["backbone.layers.1.0.conv2", "backbone.layers.1.3.conv2"] + [f"backbone.layers.2.{i}.conv2" for i in range(0, 23, 3)] + \
        ["backbone.layers.3.0.conv2"]
    C.set_cfg("PlaneRecNet_50_config")


def test_init_head_weights_focal_prior():
    from planerecnet_amd import config as C
    from planerecnet_amd.planerecnet import PlaneRecNet
    C.set_cfg("PlaneRecNet_50_config")
    net = PlaneRecNet(C.cfg)
    net.init_head_weights()
    assert abs(float(net.inst_head.cate_pred.bias[0]) + np.log(99.0)) < 1e-6
    assert float(net.fpn.lateral_convs[0].bias.abs().max()) == 0
    assert float(net.backbone.layers[1][0].conv2.offset_conv.weight.abs().max()) == 0      # backbone modules untouched


def test_savepath_roundtrip(tmp_path):
    from planerecnet_amd.utils import MovingAverage, SavePath
    p = SavePath("PlaneRecNet_101", 3, 12500).get_path(str(tmp_path))
    assert os.path.basename(p) == "PlaneRecNet_101_3_12500.pth"
    sp = SavePath.from_str(p)
    assert (sp.model_name, sp.epoch, sp.iteration) == ("PlaneRecNet_101", 3, 12500)
    sp = SavePath.from_str("/x/PlaneRecNet_50_7_99_interrupt.pth")
    assert (sp.model_name, sp.epoch, sp.iteration) == ("PlaneRecNet_50", 7, 99)
    open(p, "w").close()
    open(SavePath("PlaneRecNet_101", 4, 25000).get_path(str(tmp_path)), "w").close()
    assert SavePath.get_latest(str(tmp_path), "PlaneRecNet_101").endswith("_4_25000.pth")
    m = MovingAverage(2)
    for v in (1.0, float("nan"), 3.0, 5.0):
        m.add(v)
    assert m.get_avg() == 4.0


def test_target_assignment_host_matches_golden(golden_dir):
    from oracle import synth
    from planerecnet_amd import config as C
    from planerecnet_amd.losses import PlaneRecNetLoss
    C.set_cfg("PlaneRecNet_50_config")
    fx = np.load(os.path.join(golden_dir, "loss_synth.npz"))
    _, inst, _ = synth.make_batch(2, 480, 640, seed=4)
    ins, cate, ind, order = PlaneRecNetLoss().prepare_ground_truth(inst[0], (120, 160))
    for lv in range(4):
        assert np.array_equal(cate[lv].numpy(), fx[f"tg_cate{lv}"])
        assert np.array_equal(np.asarray(order[lv], dtype=np.int64), fx[f"tg_order{lv}"])
        assert np.array_equal(ins[lv].sum((1, 2)).numpy(), fx[f"tg_ins_area{lv}"])


def test_nms_and_helpers_cpu():
    from oracle import model_ref
    from planerecnet_amd import funcs, nms
    g = torch.Generator().manual_seed(0)
    heat = torch.rand(1, 2, 9, 9, generator=g)
    assert torch.equal(nms.point_nms(heat), model_ref.point_nms(heat))
    masks = torch.rand(6, 12, 12, generator=g) > 0.5
    scores = torch.rand(6, generator=g).sort(descending=True)[0]
    labels = torch.zeros(6, dtype=torch.long)
    a = nms.matrix_nms(labels, masks, masks.sum((1, 2)).float(), scores)
    b = model_ref.matrix_nms(labels, masks, masks.sum((1, 2)).float(), scores)
    assert torch.allclose(a, b)
    keep = nms.mask_nms(labels, masks, masks.sum((1, 2)).float(), scores, nms_thr=0.3)
    assert keep[0] and keep.shape == (6,)
    assert funcs.calc_size_preserve_ar(640, 480, 960) == (960, 720)
    assert funcs.pad_even_divided(np.ones((720, 960, 3))).shape == (736, 960, 3)


def test_batched_virtual_normal_loss_matches_oracle_per_image():
    """The one-pass, all-planes-all-images VNL (segmented sort) against the oracle's per-image, per-plane loops
    (which are pinned against the real reference), with the same numpy RNG stream; gradients included."""
    from oracle import loss_ref, synth
    from planerecnet_amd import config as C
    from planerecnet_amd.losses import VNL_Loss
    C.set_cfg("PlaneRecNet_50_config")
    B = 3
    _, inst, gtd = synth.make_batch(B, 480, 640, seed=21)
    inst[2]["masks"][:] = 1                       # no non-planar pixels in image 2
    g = torch.Generator().manual_seed(3)
    pred = (torch.rand(B, 1, 480, 640, generator=g) * 4 + 0.3)
    pred[0, 0, 100:110, 200:260] = 0.0            # exercises the zero-depth fix of the non-planar branch
    pred.requires_grad_(True)
    vnl = VNL_Loss((480, 640))
    np.random.seed(5)
    t = vnl.prepare(inst, (480, 640), torch.device("cpu"))
    got = vnl.batched(pred, gtd, t)
    ref_vnl = loss_ref.VNL((480, 640))
    np.random.seed(5)
    ref = torch.stack([ref_vnl(pred[b], inst[b]["masks"].bool(), inst[b]["plane_paras"][:, :3], gtd[b], inst[b]["k_matrix"]) for b in range(B)])
    assert got.dtype == torch.float64 and ref.dtype == torch.float64
    ok = torch.tensor([0, 1, 2])
    assert torch.allclose(got[ok], ref[ok], rtol=1e-6, atol=1e-9), (got, ref)
    (gg,) = torch.autograd.grad(got[ok].sum(), pred)
    (gr,) = torch.autograd.grad(ref[ok].sum(), pred)
    assert torch.allclose(gg[ok], gr[ok], rtol=1e-4, atol=1e-9)


def test_trimmed_means_hand_written_gradient_equals_autograd():
    """VNL_Loss._trimmed_means: value and gradient of the custom autograd node (one coefficient per triplet, scattered through the
    sort permutation) against autograd replaying the operator chain -- ties, NaN losses, invalid triplets, an empty-valid plane."""
    from planerecnet_amd.losses import VNL_Loss, VNLTargets
    rng = np.random.RandomState(11)
    seg_len = np.array([40, 7, 120, 3, 64, 16])
    n = int(seg_len.sum())
    t = VNLTargets()
    t.B, t.n_seg, t.n_tot = 2, len(seg_len), n
    t.seg = torch.from_numpy(np.repeat(np.arange(len(seg_len)), seg_len))
    t.seg_start = torch.from_numpy(np.concatenate([[0], np.cumsum(seg_len)[:-1]]))
    t.seg_img = torch.tensor([0, 0, 0, 1, 1, 1])
    t.seg_is_plane = torch.tensor([True, True, False, True, True, False])
    t.N = torch.tensor([2.0, 2.0], dtype=torch.float64)
    loss = torch.from_numpy(np.round(rng.rand(n), 2)).double()           # rounded: equal losses inside a segment
    loss[5] = float("nan")
    valid = torch.from_numpy(rng.rand(n) < 0.8)
    la, lb = loss.clone().requires_grad_(True), loss.clone().requires_grad_(True)
    ya = VNL_Loss._trimmed_means(la, valid, t, torch.device("cpu"))
    yb = VNL_Loss._trimmed_means_autograd(lb, valid, t, torch.device("cpu"))
    assert torch.equal(ya, yb) and bool(torch.isfinite(ya).all())
    w = torch.tensor([1.0, 0.37], dtype=torch.float64)
    (ga,) = torch.autograd.grad((ya * w).sum(), la)
    (gb,) = torch.autograd.grad((yb * w).sum(), lb)
    assert torch.allclose(ga, gb, rtol=1e-12, atol=0) and float(ga.abs().sum()) > 0 and float(ga[5]) == 0.0
    # a plane without a valid triplet: 0 / 0 = NaN for its image like the reference (the step is then skipped); the other image's
    # value is unaffected and the hand-written gradient stays finite (autograd's chain leaks the NaN through the prefix sums)
    valid[t.seg == 3] = False
    lc = loss.clone().requires_grad_(True)
    yc = VNL_Loss._trimmed_means(lc, valid, t, torch.device("cpu"))
    yd = VNL_Loss._trimmed_means_autograd(loss, valid, t, torch.device("cpu"))
    assert torch.allclose(yc, yd, equal_nan=True) and bool(torch.isnan(yc[1])) and float(yc[0]) == float(ya[0])
    (gc,) = torch.autograd.grad(yc[0], lc)
    assert bool(torch.isfinite(gc).all()) and torch.allclose(gc[t.seg < 3], ga[t.seg < 3], rtol=1e-12, atol=0)


def test_train_cli_surface_and_synthetic_dataset_contract():
    import train
    a = train.parser.parse_args(["--config", "PlaneRecNet_101_config", "--batch_size", "16", "--resume", "latest", "--keep_latest",
                                 "--no_autoscale", "--lr", "0.001", "--batch_alloc", "8,8"])
    assert a.config == "PlaneRecNet_101_config" and a.batch_size == 16 and a.keep_latest and not a.autoscale and a.lr == 0.001
    for flag in ("dataset", "save_folder", "log_folder", "backbone_folder", "start_iter", "validation_size", "validation_epoch", "no_tensorboard",
                 "reproductablity", "momentum", "decay", "gamma", "num_workers", "save_interval", "keep_latest_interval", "interrupt"):
        assert hasattr(a, flag), flag
    img, inst, depth = train.SyntheticPlaneDataset(4)[1]
    assert img.shape == (3, 480, 640) and depth.shape == (1, 480, 640)
    assert inst["masks"].dtype == torch.uint8 and inst["boxes"].dtype == torch.float64 and inst["classes"].dtype == torch.int64
    imgs, insts, depths = train.detection_collate([train.SyntheticPlaneDataset(4)[0], train.SyntheticPlaneDataset(4)[1]])
    assert len(imgs) == len(insts) == len(depths) == 2


def test_simple_inference_cli_surface():
    import simple_inference as si
    a = si.parse_args(["--image", "a.png:b.png", "--nms_mode", "mask", "--score_threshold", "0.2", "--top_k", "7", "--depth_mode", "gray"])
    assert a.image == "a.png:b.png" and a.nms_mode == "mask" and a.score_threshold == 0.2 and a.top_k == 7 and a.depth_mode == "gray"
    for flag in ("trained_model", "config", "images", "max_img", "ibims1", "ibims1_pd", "no_mask", "no_box", "no_text", "depth_shift"):
        assert hasattr(a, flag), flag


def test_block_ownership_calibration_survives_an_import_under_no_grad():
    """planerecnet_amd.blocks measures once, at import, what a solely owned incoming gradient looks like (TensorImpl use count, Python reference count) with a
    two-element autograd graph; an application that first imports the package inside a torch.no_grad() / inference_mode() region must get the same numbers (the
    probe re-enables grad mode for itself) instead of an exception at import."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import torch, sys; sys.path.insert(0, %r)\n"
            "ctx = {'plain': None, 'no_grad': torch.no_grad(), 'inference': torch.inference_mode()}[sys.argv[1]]\n"
            "if ctx is not None: ctx.__enter__()\n"
            "from planerecnet_amd import blocks\n"
            "print('OWNED', blocks._OWNED)\n") % root
    seen = []
    for mode in ("plain", "no_grad", "inference"):
        r = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        seen.append([ln for ln in r.stdout.splitlines() if ln.startswith("OWNED")][-1])
    assert seen[0] == seen[1] == seen[2] and seen[0] != "OWNED (0, 0)", seen

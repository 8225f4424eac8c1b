"""CPU: the oracle restatement against the committed golden vectors, which are outputs of the REAL
reference (shim-imported in the build container by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_ref, model_ref, synth

CN = "PlaneRecNet_50_config"
ARCH = model_ref.ARCH[CN]


def digest(t, n=512, seed=123):
    if t is None:
        return np.zeros(4 + n)
    t = t.detach().double().flatten()
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, t.numel(), (n,), generator=g)
    return np.concatenate([[t.mean().item(), t.std().item(), t.abs().sum().item(), float(t.numel())], t[idx].numpy()])


def close(a, b, rtol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b).max() / (np.abs(b).max() + 1e-12)
    assert err < rtol, f"{what}: max-rel {err:.2e} >= {rtol}"


@pytest.fixture(scope="module")
def sd():
    return synth.make_state_dict(CN, seed=1)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_model_small_matches_reference(sd, golden_dir, mode):
    fx = np.load(os.path.join(golden_dir, "model_r50_small.npz"))
    x, _, _ = synth.make_batch(2, 64, 96, seed=2)
    with torch.no_grad():
        mask, cate, kern, depth = model_ref.forward(sd, x, ARCH, training=(mode == "train"))
    # fp32 tolerance: 1-ulp kernel-selection noise between in-place/out-of-place ATen paths (see make_golden.py)
    close(mask, fx[f"{mode}_mask"], 5e-5, "mask")
    close(depth, fx[f"{mode}_depth"], 5e-5, "depth")
    for i in range(4):
        close(cate[i], fx[f"{mode}_cate{i}"], 5e-5, f"cate{i}")
        close(digest(kern[i]), fx[f"{mode}_kern{i}_digest"], 5e-5, f"kern{i}")


def test_inference_postprocess_matches_reference(sd, golden_dir):
    fx = np.load(os.path.join(golden_dir, "model_r50_small.npz"))
    x2, _, _ = synth.make_batch(1, 128, 160, seed=3)
    sd_inf = dict(sd)
    sd_inf["inst_head.cate_pred.bias"] = sd["inst_head.cate_pred.bias"] + 1.0
    res = model_ref.inference(sd_inf, x2, ARCH)[0]
    assert len(res["pred_scores"]) == len(fx["inf_scores"])
    close(res["pred_scores"], fx["inf_scores"], 1e-3, "scores")
    assert np.array_equal(res["pred_classes"].numpy(), fx["inf_classes"])
    assert np.abs(res["pred_boxes"].numpy() - fx["inf_boxes"]).max() <= 1.0
    close(digest(res["pred_depth"]), fx["inf_depth_digest"], 2e-4, "depth")
    area = res["pred_masks"].sum((1, 2)).numpy()
    assert np.abs(area - fx["inf_mask_area"]).max() <= 0.01 * fx["inf_mask_area"].max()


def _loss_inputs():
    g = torch.Generator().manual_seed(5)
    B = 2
    _, inst, gtd = synth.make_batch(B, 480, 640, seed=4)
    mask_pred = torch.randn(B, 128, 120, 160, generator=g).relu_()
    cate = [torch.randn(B, 2, s, s, generator=g) - 2.0 for s in ARCH.num_grids]
    kern = [torch.randn(B, 128, s, s, generator=g) * 0.1 for s in ARCH.num_grids]
    depth = torch.rand(B, 1, 240, 320, generator=g) * 4 + 0.3
    return mask_pred, cate, kern, depth, inst, gtd


def test_loss_terms_and_grads_match_reference(golden_dir):
    fx = np.load(os.path.join(golden_dir, "loss_synth.npz"))
    mask_pred, cate, kern, depth, inst, gtd = _loss_inputs()
    leaves = [mask_pred] + cate + kern + [depth]
    for t in leaves:
        t.requires_grad_(True)
    np.random.seed(7)
    out = loss_ref.joint_loss(mask_pred, cate, kern, depth, inst, gtd)
    for k in ("ins", "cat", "dpt", "pln", "lav"):
        assert abs(float(out[k]) - float(fx[k])) <= 2e-6 * max(1.0, abs(float(fx[k]))), k
    assert out["pln"].dtype == torch.float64          # quirk Q6
    grads = torch.autograd.grad(sum(out.values()).sum(), leaves, allow_unused=True)
    close(digest(grads[0]), fx["grad_mask_digest"], 1e-5, "d mask")
    close(digest(grads[-1]), fx["grad_depth_digest"], 1e-5, "d depth")
    for i in range(4):
        close(digest(grads[1 + i]), fx[f"grad_cate{i}_digest"], 1e-5, f"d cate{i}")
        close(digest(grads[5 + i]), fx[f"grad_kern{i}_digest"], 1e-5, f"d kern{i}")


def test_gt_assignment_bit_exact(golden_dir):
    fx = np.load(os.path.join(golden_dir, "loss_synth.npz"))
    _, inst, _ = synth.make_batch(2, 480, 640, seed=4)
    tg = loss_ref.assign_targets(inst[0], (120, 160), ARCH.num_grids)
    for lv in range(4):
        ins, cate, ind, order = tg[lv]
        assert np.array_equal(cate.numpy(), fx[f"tg_cate{lv}"])
        assert np.array_equal(np.asarray(order, dtype=np.int64), fx[f"tg_order{lv}"])
        assert np.array_equal(ins.sum((1, 2)).numpy(), fx[f"tg_ins_area{lv}"])
        assert int(ind.sum()) == len(set(order))


def test_e2e_480x640_losses_match_reference(sd, golden_dir):
    fx = np.load(os.path.join(golden_dir, "e2e_r50_480x640.npz"))
    x3, inst3, gtd3 = synth.make_batch(1, 480, 640, seed=6)
    np.random.seed(11)
    with torch.no_grad():
        out = model_ref.forward(sd, x3, ARCH, training=True)
        ls = loss_ref.joint_loss(*out, inst3, gtd3)
    for k in ("ins", "cat", "dpt", "pln", "lav"):
        assert abs(float(ls[k]) - float(fx[k])) <= 1e-4 * max(1.0, abs(float(fx[k]))), (k, float(ls[k]), float(fx[k]))
    close(digest(out[0]), fx["mask_digest"], 2e-4, "mask")
    close(digest(out[3]), fx["depth_digest"], 2e-4, "depth")


def test_state_dict_layout_counts():
    for cn, nkeys in (("PlaneRecNet_50_config", 520), ("PlaneRecNet_101_config", 816)):
        sp = synth.spec(cn)
        assert len(sp) == nkeys and len({k for k, _, _ in sp}) == nkeys

"""The residual block as ONE call each way (include/prn.h: prn_bottleneck_train_fwd / _bwd; planerecnet_amd/blocks.py) against
(a) the same module evaluated operator by operator (planerecnet_amd.ops: conv2d / batch_norm / deform_conv_block nodes) and
(b) a plain torch fp64 restatement of models/backbone.py:53-73 on the CPU (plain blocks; the deformable variant's reference is (a), whose
operators are pinned against oracle/dcn_ref.py in tests/test_ops_gpu.py).

A block call issues the launches of (a) with the arguments of (a) -- and, on small maps, leaves K-split sums / Winograd transforms to the
BatchNorm kernels (PRN_BLK_HANDOVER) -- so without hand-overs the two agree bit for bit, with them to fp32 rounding (statistics summed in
another order)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def rnd(*s, seed=0, scale=1.0):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed), dtype=torch.float64) * scale


def make_block(cin, planes, stride, downsample, dcn, seed):
    from planerecnet_amd.backbone import Bottleneck
    from torch import nn
    torch.manual_seed(seed)
    ds = None
    if downsample:
        ds = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
    blk = Bottleneck(cin, planes, stride, ds, use_dcn=dcn)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
        if dcn:                                              # non-zero offsets / modulation: the sampler is exercised (models/dcn.py initialises them to zero)
            blk.conv2.offset_conv.weight.copy_(0.05 * torch.randn(blk.conv2.offset_conv.weight.shape, generator=g))
            blk.conv2.modulator_conv.weight.copy_(0.05 * torch.randn(blk.conv2.modulator_conv.weight.shape, generator=g))
            blk.conv2.offset_conv.bias.copy_(0.3 * torch.randn(18, generator=g))
            blk.conv2.modulator_conv.bias.copy_(0.3 * torch.randn(9, generator=g))
    return blk.to(dev()).train()


def run(blk, x, go, hand_back, ext):
    """forward + backward -> (tensors to compare, named)"""
    from planerecnet_amd import ops
    blk.zero_grad(set_to_none=True)
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.reset_running_stats()
    xl = x.clone().requires_grad_(True)
    if hand_back:
        out, xid = blk(xl, hand_back=True)
        loss = (out * go).sum() + (xid * ext).sum()            # the handed-back input has another reader (the FPN / the decoder in the model)
    else:
        out = blk(xl)
        loss = (out * go).sum()
    loss.backward()
    ops.wgrad_join()
    torch.cuda.synchronize()
    res = {"out": out.detach().clone(), "dx": xl.grad.detach().clone()}
    for n, p in blk.named_parameters():
        res["d " + n] = p.grad.detach().clone()
    for n, b in blk.named_buffers():
        if "running" in n:
            res[n] = b.detach().clone()
    return res


CASES = [
    # B, Cin, planes, H, W, stride, downsample, dcn          what it covers
    (2, 64, 16, 12, 16, 1, False, False),                    # conv2 on the direct kernel (too few channels for Winograd), one-launch BatchNorm
    (2, 256, 64, 16, 24, 1, False, False),                   # Winograd conv2 + every hand-over (small map)
    (8, 1024, 256, 30, 40, 1, False, False),                 # the stage-3 block of the benchmark: K-split sums left to bn1 / bn2's backward
    (2, 256, 64, 120, 160, 1, False, False),                 # large map: two-launch BatchNorm, no hand-overs, V kept for the weight gradient
    (2, 64, 64, 24, 32, 1, True, False),                     # stride-1 downsample branch (layers.0.0)
    (2, 128, 64, 24, 32, 2, True, False),                    # stride-2 plain block with a downsample branch (dilated input gradient of conv2)
    (2, 256, 64, 16, 24, 1, False, True),                    # deformable conv2, stride 1
    (2, 128, 64, 24, 32, 2, True, True),                     # deformable conv2, stride 2, downsample branch (first block of stages 2-4)
    (2, 256, 128, 30, 40, 2, True, True),
    (3, 64, 16, 7, 9, 1, False, False),                      # H*W % 4 != 0: nothing vectorised, nothing handed over
    (8, 2048, 512, 15, 20, 1, False, False),                 # the stage-4 block of the benchmark
    (4, 256, 64, 18, 28, 1, False, False),                   # H % 4 != 0
    (5, 256, 64, 20, 28, 1, False, False),                   # 175 Winograd tiles: a padded tile axis
]


@pytest.mark.parametrize("B,Cin,P,H,W,stride,ds,dcn", CASES)
@pytest.mark.parametrize("deferred", [False, True])
def test_block_call_equals_the_operator_sequence(B, Cin, P, H, W, stride, ds, dcn, deferred):
    from planerecnet_amd import blocks, ops
    d = dev()
    blk = make_block(Cin, P, stride, ds, dcn, seed=3)
    x = (rnd(B, Cin, H, W, seed=1).relu() * 0.7).float().to(d)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    go = rnd(B, 4 * P, Ho, Wo, seed=6).float().to(d)
    ext = rnd(B, Cin, H, W, seed=7).float().to(d)
    hb = bool(ds and stride == 2)
    saved = (blocks.ENABLED, blocks.HANDOVER, ops.WGRAD_ASYNC)
    try:
        ops.set_wgrad_async(deferred)
        blocks.ENABLED = False                                # the operator sequence: ops.conv2d / batch_norm / deform_conv_block nodes
        ref = run(blk, x, go, hb, ext)
        blocks.ENABLED, blocks.HANDOVER = True, False
        n0 = dict(blocks.STATS)
        plain = run(blk, x, go, hb, ext)
        assert blocks.STATS["fwd"] == n0["fwd"] + 1 and blocks.STATS["bwd"] == n0["bwd"] + 1, "the block entry points were not used"
        blocks.HANDOVER = True
        n1 = blocks.STATS["scatter_acc"]
        handed = run(blk, x, go, hb, ext)
        assert blocks.STATS["scatter_acc"] == n1 + (1 if hb else 0), "stride-2 block: the input gradient was not added into the handed-back gradient in place"
    finally:
        blocks.ENABLED, blocks.HANDOVER = saved[0], saved[1]
        ops.set_wgrad_async(saved[2])
    assert set(ref) == set(plain) == set(handed)
    # (incl. the deformable blocks: the CSR bins of the sampler's input gradient are sorted before the gather -- csrc/prn_dcn.hip: csr_sort_bin -- so nothing
    # behind it differs run to run any more)
    for k in ref:
        assert torch.equal(plain[k], ref[k]), "%s: block call without hand-overs differs from the operator sequence (max %.3e)" % (k, (plain[k] - ref[k]).abs().max().item())
    for k in ref:
        a, b = handed[k].double(), ref[k].double()
        tol = 2e-3 if k == "dx" or k.startswith("d ") else 2e-5      # (a BatchNorm output within rounding of the ReLU's zero flips single gradient elements)
        assert (a - b).norm().item() <= tol * b.norm().item() + 1e-9, "%s: with hand-overs, rms difference %.3e of %.3e" % (k, (a - b).norm().item(), b.norm().item())


@pytest.mark.parametrize("B,Cin,P,H,W,stride,ds", [(2, 256, 64, 16, 24, 1, False), (2, 64, 64, 24, 32, 1, True), (2, 128, 64, 24, 32, 2, True), (4, 1024, 256, 30, 40, 1, False)])
def test_block_call_against_torch_fp64(B, Cin, P, H, W, stride, ds):
    """models/backbone.py:53-73 restated with torch.nn.functional in fp64 on the CPU."""
    from planerecnet_amd import ops
    d = dev()
    blk = make_block(Cin, P, stride, ds, False, seed=5)
    x = (rnd(B, Cin, H, W, seed=1).relu() * 0.7)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    go = rnd(B, 4 * P, Ho, Wo, seed=6)
    got = run(blk, x.float().to(d), go.float().to(d), False, None)
    ps = {n: p.detach().double().cpu().requires_grad_(True) for n, p in blk.named_parameters()}
    xr = x.clone().requires_grad_(True)

    def bn(t, pre):
        return F.batch_norm(t, None, None, ps[pre + ".weight"], ps[pre + ".bias"], True, 0.1, 1e-5)
    o = F.relu(bn(F.conv2d(xr, ps["conv1.weight"]), "bn1"))
    o = F.relu(bn(F.conv2d(o, ps["conv2.weight"], stride=stride, padding=1), "bn2"))
    o = bn(F.conv2d(o, ps["conv3.weight"]), "bn3")
    res = bn(F.conv2d(xr, ps["downsample.0.weight"], stride=stride), "downsample.1") if ds else xr
    out = F.relu(o + res)
    out.backward(go)
    ref = {"out": out.detach(), "dx": xr.grad}
    for n, p in ps.items():
        ref["d " + n] = p.grad
    for k, r in ref.items():
        a = got[k].double().cpu()
        tol = 1e-3 if k == "out" else 4e-3
        assert (a - r).norm().item() <= tol * r.norm().item() + 1e-9, "%s: rms error %.3e of %.3e" % (k, (a - r).norm().item(), r.norm().item())
    # running statistics: momentum 0.1 from (0, 1), unbiased variance (nn.BatchNorm2d)
    c1 = F.conv2d(x, ps["conv1.weight"].detach())
    n = c1.numel() / c1.shape[1]
    assert torch.allclose(got["bn1.running_mean"].double().cpu(), 0.1 * c1.mean((0, 2, 3)), rtol=1e-4, atol=1e-6)
    assert torch.allclose(got["bn1.running_var"].double().cpu(), 0.9 + 0.1 * c1.var((0, 2, 3), unbiased=False) * n / (n - 1), rtol=1e-4, atol=1e-6)
    assert blk.bn1.state_dict()["num_batches_tracked"].item() >= 1


def test_a_gradient_somebody_else_holds_is_not_added_into():
    """The stride-2 blocks add their input gradient INTO the gradient the stage output's other readers sent back -- only when this node is the
    sole owner of that tensor.  A reader that keeps its gradient (a hook, retain_grad) must find it unchanged."""
    from planerecnet_amd import blocks, ops
    d = dev()
    blk = make_block(128, 64, 2, True, False, seed=4)
    x = rnd(2, 128, 24, 32, seed=1).float().to(d).requires_grad_(True)
    ext = rnd(2, 128, 24, 32, seed=2).float().to(d)
    kept = []
    out, xid = blk(x, hand_back=True)
    xid.register_hook(lambda g: kept.append(g))              # somebody looks at (and keeps) the gradient of the handed-back input
    n0 = blocks.STATS["scatter_acc"]
    ((out * out).sum() + (xid * ext).sum()).backward()
    ops.wgrad_join()
    torch.cuda.synchronize()
    assert blocks.STATS["scatter_acc"] == n0, "the block added into a gradient tensor that a hook still holds"
    assert torch.equal(kept[0], ext)


def test_block_backward_notices_a_parameter_changed_after_forward():
    from planerecnet_amd import ops
    d = dev()
    blk = make_block(256, 64, 1, False, False, seed=4)
    x = rnd(2, 256, 16, 24, seed=1).float().to(d).requires_grad_(True)
    out = blk(x)
    with torch.no_grad():
        blk.conv3.weight.mul_(2.0)
    with pytest.raises(RuntimeError, match="modified"):
        out.sum().backward()
    ops.wgrad_flush()

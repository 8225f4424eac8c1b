"""GPU parity of every HIP operator (called through the C ABI via planerecnet_amd.ops) against the CPU
restatement of the same operator: torch CPU fp64 for conv / norm / resample (the ATen ops the reference
calls), oracle/dcn_ref.py for DCNv2.  fp32 tolerance: max-abs error <= 2e-4 * max|ref| (K up to 4608 fp32
accumulations, different summation order)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 2e-4


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def rnd(*s, seed=0, scale=1.0):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed), dtype=torch.float64) * scale


def close(got, ref, what, rtol=RTOL):
    got = got.detach().double().cpu()
    ref = ref.detach().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    den = ref.abs().max().item() + 1e-12
    assert err <= rtol * den, f"{what}: max-abs {err:.3e} vs scale {den:.3e} (rel {err / den:.2e})"


def ref_conv(x, w, b, stride, pad, mode, addend=None, epi=0):
    if mode == 2:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    if mode in (1, 2):
        x = F.pad(x, (1, 1, 1, 1), mode="reflect")
        pad = 0
    y = F.conv2d(x, w, b, stride=stride, padding=pad)
    if addend is not None:
        y = y + addend
    if epi == 1:
        y = F.relu(y)
    elif epi == 2:
        y = torch.sigmoid(y)
    return y


CONV_CASES = [
    # B, C, H, W, M, K, stride, pad, mode, bias, addend, epi
    (2, 16, 13, 17, 24, 1, 1, 0, 0, False, False, 0),
    (2, 64, 30, 40, 256, 1, 1, 0, 0, False, False, 0),
    (2, 70, 12, 20, 130, 3, 1, 1, 0, True, False, 1),
    (1, 3, 38, 50, 64, 7, 2, 3, 0, False, False, 0),
    (2, 32, 15, 21, 48, 3, 2, 1, 0, True, False, 0),
    (2, 40, 16, 20, 72, 1, 2, 0, 0, False, False, 0),
    (2, 24, 9, 12, 40, 3, 1, 1, 1, True, False, 0),
    (2, 24, 7, 10, 40, 3, 1, 1, 2, True, False, 0),
    (1, 258, 16, 16, 64, 3, 1, 1, 0, False, False, 0),
    (2, 20, 10, 14, 27, 3, 1, 1, 0, True, True, 0),
    (1, 128, 12, 16, 300, 1, 1, 0, 0, True, False, 2),
    (1, 64, 24, 32, 1, 3, 1, 1, 1, True, False, 0),
    (2, 24, 15, 21, 40, 1, 2, 0, 0, False, False, 0),        # stride-2 1x1 on odd sizes (scatter-form input gradient)
    (2, 130, 9, 13, 70, 3, 1, 1, 2, True, False, 0),         # nearest-x2 + reflect + 3x3, sub-pixel form, ragged tiles
    (1, 16, 2, 2, 8, 3, 1, 1, 2, False, False, 0),           # ... smallest map it accepts
    (2, 24, 160, 208, 1, 3, 1, 1, 1, True, False, 0),        # depth-head shape class: direct (non-GEMM) kernels, reflect padding
    (1, 40, 256, 260, 2, 3, 1, 1, 0, True, True, 1),         # ... two output channels, zero padding, addend + ReLU
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(case):
    from planerecnet_amd import ops
    B, C, H, W, M, K, stride, pad, mode, has_b, has_a, epi = case
    x = rnd(B, C, H, W, seed=1).requires_grad_(True)
    w = rnd(M, C, K, K, seed=2, scale=(C * K * K) ** -0.5).requires_grad_(True)
    b = rnd(M, seed=3).requires_grad_(True) if has_b else None
    yr = ref_conv(x, w, b, stride, pad, mode, None, 0)
    a = rnd(*yr.shape, seed=4).requires_grad_(True) if has_a else None
    yr = ref_conv(x, w, b, stride, pad, mode, a, epi)
    go = rnd(*yr.shape, seed=5)
    leaves = [t for t in (x, w, b, a) if t is not None]
    gr = torch.autograd.grad(yr, leaves, go)

    d = dev()
    xd, wd = x.detach().float().to(d).requires_grad_(True), w.detach().float().to(d).requires_grad_(True)
    bd = b.detach().float().to(d).requires_grad_(True) if has_b else None
    ad = a.detach().float().to(d).requires_grad_(True) if has_a else None
    yd = ops.conv2d(xd, wd, bd, stride, pad, mode, epi, ad)
    close(yd, yr, "conv fwd")
    gd = torch.autograd.grad(yd, [t for t in (xd, wd, bd, ad) if t is not None], go.float().to(d))
    names = ["dx", "dw"] + (["db"] if has_b else []) + (["dadd"] if has_a else [])
    for name, g1, g0 in zip(names, gd, gr):
        close(g1, g0, "conv " + name)


def test_up2_subpixel_form_matches_gather_form():
    """PRN_IN_UP2_PHASE (four 2x2 phase convolutions of the source) against PRN_IN_UP2_REFLECT (3x3 gather on the upsampled
    map): same operator, both in fp32 on the GPU -- forward and all three gradients."""
    from planerecnet_amd import ops
    d = dev()
    x = rnd(2, 48, 11, 14, seed=1).float().to(d)
    w = rnd(36, 48, 3, 3, seed=2, scale=(48 * 9) ** -0.5).float().to(d)
    b = rnd(36, seed=3).float().to(d)
    go = rnd(2, 36, 22, 28, seed=4).float().to(d)
    outs = []
    for sub in (True, False):
        ops.UP2_SUBPIXEL = sub
        try:
            ts = [t.clone().requires_grad_(True) for t in (x, w, b)]
            y = ops.conv2d(ts[0], ts[1], ts[2], 1, 1, ops.IN_UP2_REFLECT)
            outs.append([y] + list(torch.autograd.grad(y, ts, go)))
        finally:
            ops.UP2_SUBPIXEL = True
    for name, a, r in zip(["y", "dx", "dw", "db"], outs[0], outs[1]):
        close(a, r.double().cpu(), "up2 subpixel vs gather " + name, rtol=2e-5)


@pytest.mark.parametrize("C,M,H,W", [(64, 128, 13, 16), (32, 96, 9, 20)])
def test_up2_subpixel_phases_on_the_16bit_pipe(C, M, H, W):
    """Upsample(x2) -> ReflectionPad2d(1) -> Conv3x3 (+ bias) in its sub-pixel form with the four phases on the split kernel (csrc/prn_gemm_split.hip: TAPS = 2,
    phase = z axis, replicate-border tap gather, phase-interleaved store) and its 4x4 / stride-2 input gradient (TAPS = 1): forced onto the 16-bit pipe, output and
    input gradient must agree with the fp64 operator as closely as the fp32 kernels do."""
    from planerecnet_amd import ops
    d = dev()
    x = rnd(2, C, H, W, seed=1)
    w = rnd(M, C, 3, 3, seed=2, scale=(9 * C) ** -0.5)
    b = rnd(M, seed=3)
    go = rnd(2, M, 2 * H, 2 * W, seed=4)
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(F.pad(F.interpolate(xr, scale_factor=2, mode="nearest"), (1, 1, 1, 1), mode="reflect"), w, b)
    (gr,) = torch.autograd.grad(yr, [xr], go)
    res = {}
    old = ops.set_split_gemm(mode=0)
    try:
        for mode in (0, 2):
            ops.set_split_gemm(mode=mode)
            xd = x.float().to(d).requires_grad_(True)
            yd = ops.conv2d(xd, w.float().to(d), b.float().to(d), 1, 1, ops.IN_UP2_REFLECT)
            (gd,) = torch.autograd.grad(yd, [xd], go.float().to(d))
            res[mode] = (yd.detach(), gd)
    finally:
        ops.set_split_gemm(**old)
    assert not torch.equal(res[0][0], res[2][0]) and not torch.equal(res[0][1], res[2][1]), "mode 2 did not reach the 16-bit-pipe kernels"
    for k, (ref, what) in enumerate(((yr, "y"), (gr, "dx"))):
        e0 = (res[0][k].double().cpu() - ref.detach()).abs().max().item() / ref.abs().max().item()
        e2 = (res[2][k].double().cpu() - ref.detach()).abs().max().item() / ref.abs().max().item()
        assert e2 <= max(2.0 * e0, 2e-6), (what, e0, e2)


@pytest.mark.parametrize("epi", ["none", "relu", "sigmoid"])
def test_up2_phase_descriptor_applies_its_epilogue_on_either_pipe(epi):
    """A PRN_IN_UP2_PHASE descriptor with each of the three epilogues, fp32 kernels only (mode 0) and with the 16-bit pipe forced (mode 2): the
    activation is applied in both (advisor, round 5: the phase launch of the split kernel has bias and ReLU only -- a sigmoid descriptor used to come
    back without its activation there; it now stays on the fp32 kernel)."""
    from planerecnet_amd import ops
    from planerecnet_amd._lib import EPI_NONE, EPI_RELU, EPI_SIGMOID, IN_UP2_PHASE
    d = dev()
    C, M, H, W = 64, 128, 13, 16
    x = rnd(2, C, H, W, seed=1).float().to(d)
    w = rnd(M, C, 3, 3, seed=2, scale=(9 * C) ** -0.5).float().to(d)
    b = rnd(M, seed=3).float().to(d)
    wp = torch.empty(4, M, C, 2, 2, device=d, dtype=torch.float32)
    ops.check(ops.lib.prn_up2_phase_weights(ops._p(w), ops._p(wp), M, C, ops._stream()), "prn_up2_phase_weights")
    code = {"none": EPI_NONE, "relu": EPI_RELU, "sigmoid": EPI_SIGMOID}[epi]
    act = {"none": lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid}[epi]
    old = ops.set_split_gemm(mode=0)
    try:
        for mode in (0, 2):
            ops.set_split_gemm(mode=mode)
            plain = ops.conv_fwd_raw(x, wp, b, None, M, 2, 1, 0, 2 * H, 2 * W, IN_UP2_PHASE, 1, EPI_NONE)
            got = ops.conv_fwd_raw(x, wp, b, None, M, 2, 1, 0, 2 * H, 2 * W, IN_UP2_PHASE, 1, code)
            assert (plain < 0).any()
            close(got, act(plain.double()).cpu(), "up2 phase epilogue %s, mode %d" % (epi, mode), rtol=3e-6 if epi != "sigmoid" or mode == 0 else 2e-5)
    finally:
        ops.set_split_gemm(**old)


@pytest.mark.parametrize("C,M,H,W,pad,K", [(64, 160, 22, 30, 3, 4), (32, 96, 17, 12, 1, 4), (96, 130, 10, 16, 2, 4), (64, 160, 21, 30, 0, 1), (32, 72, 16, 16, 0, 1)])
def test_conv4x4_stride2_on_the_16bit_pipe_is_the_fp32_conv(C, M, H, W, pad, K):
    """The 4x4 / stride-2 zero-padded convolution (input gradient of the sub-pixel upsample-convolutions, ops._ConvUp2.backward) with the split kernel's
    tap gather (csrc/prn_gemm_split.hip: TAPS, tap-major weight images): forced onto the 16-bit pipe (mode 2) it must agree with the fp64 convolution as
    closely as the fp32 MFMA kernel does, tails in M, in the pixel tiles and at every image border included."""
    from planerecnet_amd import ops
    d = dev()
    x = rnd(2, C, H, W, seed=1)
    w = rnd(M, C, K, K, seed=2, scale=(K * K * C) ** -0.5)             # (K = 1: the stride-2 downsample convolutions, one tap)
    Ho, Wo = (H + 2 * pad - K) // 2 + 1, (W + 2 * pad - K) // 2 + 1
    ref = F.conv2d(x, w, None, stride=2, padding=pad)
    got = {}
    old = ops.set_split_gemm(mode=0)
    try:
        for mode in (0, 2):
            ops.set_split_gemm(mode=mode)
            got[mode] = ops.conv_fwd_raw(x.float().to(d), w.float().to(d), None, None, M, K, 2, pad, Ho, Wo)
    finally:
        ops.set_split_gemm(**old)
    e0 = (got[0].double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    e2 = (got[2].double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert not torch.equal(got[0], got[2]), "mode 2 did not reach the 16-bit-pipe kernel"
    assert e2 <= max(2.0 * e0, 2e-6), (e0, e2)
    close(got[2], ref, "conv 4x4 s2 on the 16-bit pipe", rtol=2e-5)


@pytest.mark.parametrize("relu", [False, True])
def test_conv_up2_inference_equals_upsample_pad_conv(relu):
    """The no-autograd sub-pixel form with bias + ReLU epilogue (eval-mode decoder blocks with folded BatchNorm) against
    Upsample(2, nearest) -> ReflectionPad2d(1) -> Conv2d(3) (-> ReLU); the phase kernels follow an in-place weight update."""
    from planerecnet_amd import ops
    d = dev()
    x = rnd(2, 48, 11, 14, seed=1)
    w = rnd(36, 48, 3, 3, seed=2, scale=(48 * 9) ** -0.5)
    b = rnd(36, seed=3)

    def ref(wt):
        y = F.conv2d(F.pad(F.interpolate(x, scale_factor=2, mode="nearest"), (1, 1, 1, 1), mode="reflect"), wt, b)
        return F.relu(y) if relu else y
    wd = w.float().to(d)
    with torch.no_grad():
        close(ops.conv_up2_inference(x.float().to(d), wd, b.float().to(d), relu), ref(w), "up2 inference")
        wd.mul_(0.5)                                               # version bump: cached phase kernels must be rebuilt
        close(ops.conv_up2_inference(x.float().to(d), wd, b.float().to(d), relu), ref(w * 0.5), "up2 inference after weight update")


@pytest.mark.parametrize("K,C,M,bias", [(3, 40, 72, False), (3, 258, 256, False), (3, 64, 2, True), (1, 48, 40, True)])
def test_ragged_conv_and_group_norm_match_per_segment_ops(K, C, M, bias):
    """One GEMM over the pixels of five differently sized maps (shared weights) == the five dense convolutions; same for the
    GroupNorm+ReLU that follows.  Forward, input gradient, weight / affine gradients."""
    from planerecnet_amd import ops
    d = dev()
    B, sizes = 8, [(12, 12), (8, 8), (6, 4), (4, 4), (4, 2)]
    rs = ops.RaggedShape(B, sizes)
    assert rs.supported()
    xs = [rnd(B, C, h, w, seed=10 + i).float().to(d) for i, (h, w) in enumerate(sizes)]
    w = rnd(M, C, K, K, seed=2, scale=(C * K * K) ** -0.5).float().to(d)
    b = rnd(M, seed=3).float().to(d) if bias else None
    G = 2 if M % 2 == 0 else 1
    gam, bet = (rnd(M, seed=4).float() + 1.5).to(d), rnd(M, seed=5).float().to(d)
    gos = [rnd(B, M, h, w_, seed=20 + i).float().to(d) for i, (h, w_) in enumerate(sizes)]

    def run(ragged):
        leaves = [t.clone().requires_grad_(True) for t in xs] + [t.clone().requires_grad_(True) for t in ([w, gam, bet] + ([b] if bias else []))]
        xl, (wl, gl, bl) = leaves[:5], leaves[5:8]
        bb = leaves[8] if bias else None
        # (a bias in front of a GroupNorm has a mathematically zero gradient: the biased cases test the conv alone)
        if ragged:
            y = ops.ragged_conv2d(rs.pack(xl), wl, bb, rs)
            if not bias:
                y = ops.ragged_group_norm_relu(y, gl, bl, G, 1e-5, rs)
            loss = (y * rs.pack(gos)).sum() + 0.0 * (gl.sum() + bl.sum())
            ys = rs.unpack(y, M)
        else:
            ys = [ops.conv2d(x, wl, bb, 1, (K - 1) // 2) for x in xl]
            if not bias:
                ys = [ops.group_norm_relu(y, gl, bl, G, 1e-5) for y in ys]
            loss = sum((y * g).sum() for y, g in zip(ys, gos)) + 0.0 * (gl.sum() + bl.sum())
        return ys, torch.autograd.grad(loss, leaves)

    ya, ga = run(True)
    yb, gb = run(False)
    for i, (a, r) in enumerate(zip(ya, yb)):
        close(a, r.double().cpu(), "ragged y[%d]" % i, rtol=2e-5)
    for i, (a, r) in enumerate(zip(ga, gb)):
        close(a, r.double().cpu(), "ragged grad[%d]" % i, rtol=1e-4)


def test_ragged_shape_rejects_unaligned_segments():
    from planerecnet_amd import ops
    assert not ops.RaggedShape(1, [(36, 36), (24, 24)]).supported()       # 1296 pixels: not a whole number of 64-pixel tiles
    assert ops.RaggedShape(8, [(40, 40), (36, 36), (24, 24), (16, 16), (12, 12)]).supported()


@pytest.mark.parametrize("shape", [
    # C, H, W, M, K, pad, mode (0 zero / 1 reflect), bias, addend, epilogue (0 none / 1 relu / 2 sigmoid)      (B = 8: the K-split layers of the network, and tail-split ones)
    (1024, 30, 40, 256, 1, 0, 0, False, False, 0),      # stage-3 1x1 (split 2..3)
    (256, 30, 40, 1024, 1, 0, 0, True, True, 1),
    (2048, 15, 20, 512, 1, 0, 0, True, False, 2),    # stage 4: 2400 pixels
    (512, 15, 20, 512, 3, 1, 0, False, True, 0),
    (256, 16, 16, 256, 3, 1, 1, True, False, 1),
    (3728, 30, 40, 256, 1, 0, 0, True, False, 0),       # plane prior: K = 3728
    (64, 120, 160, 64, 3, 1, 0, True, False, 0),        # no K split, tail split only (or none)
])
def test_conv_split_sum_inside_the_gemm_is_bit_identical(shape, monkeypatch):
    """prn_conv2d_fwd_counted (K-split partials summed by the last workgroup to arrive, agent-scope stores / loads across the
    eight XCDs) against the two-kernel path (counters = NULL): bit-identical output, counters back at zero, 30 launches in
    a row on the same workspace and counters (a stale or not-yet-visible partial would show up as a mismatch)."""
    from planerecnet_amd import ops
    monkeypatch.setenv("PRN_CONV_FUSED_REDUCE", "1")        # (opt-in: neutral on the training step, see fused_reduce_ok in prn_conv.hip)
    C, H, W, M, K, pad, mode, has_b, has_a, epi = shape
    B = 8
    g = torch.Generator().manual_seed(C + M)
    lib, _p, _stream, check = ops.lib, ops._p, ops._stream, ops.check
    x0 = torch.randn(B, C, H, W, generator=g).cuda()
    w = (torch.randn(M, C * K * K, generator=g) * (C * K * K) ** -0.5).cuda()
    bias = torch.randn(M, generator=g).cuda() if has_b else None
    add = torch.randn(B, M, H, W, generator=g).cuda() if has_a else None
    _, ref, nbytes, _, _ = ops._desc(B, C, H, W, M, K, 1, pad, H, W, mode, 1, epi)
    ws = torch.empty(max(nbytes, 16) // 4, device="cuda")
    cnt = torch.zeros(ops.TILE_COUNTERS, device="cuda", dtype=torch.int32)
    y_two, y_one = torch.empty(B, M, H, W, device="cuda"), torch.empty(B, M, H, W, device="cuda")
    for rep in range(30):
        x = x0 * (1.0 + rep)                                # different partials every round
        ws.fill_(float("nan"))
        check(lib.prn_conv2d_fwd_counted(ref, _p(x), _p(w), None, _p(bias), _p(add), _p(y_two), _p(ws), None, _stream(), 0), "two-kernel")
        ws.fill_(float("nan"))
        y_one.fill_(float("nan"))
        check(lib.prn_conv2d_fwd_counted(ref, _p(x), _p(w), None, _p(bias), _p(add), _p(y_one), _p(ws), _p(cnt), _stream(), 0), "folded")
        assert torch.equal(y_one, y_two), (rep, float((y_one - y_two).abs().max()))
        assert int(cnt.abs().sum()) == 0
    ref_y = F.conv2d(F.pad(x.double().cpu(), (pad,) * 4, mode="reflect") if mode == ops.IN_REFLECT else x.double().cpu(),
                     w.double().cpu().view(M, C, K, K), bias.double().cpu() if has_b else None, padding=0 if mode == ops.IN_REFLECT else pad)
    if has_a:
        ref_y = ref_y + add.double().cpu()
    ref_y = torch.relu(ref_y) if epi == ops.EPI_RELU else (torch.sigmoid(ref_y) if epi == ops.EPI_SIGMOID else ref_y)
    assert float((y_one.double().cpu() - ref_y).abs().max()) <= 2e-4 * max(float(ref_y.abs().max()), 1.0)


@pytest.mark.parametrize("G,C,H,W,M,K,stride,pad,mode", [
    (8, 1024, 30, 40, 256, 1, 1, 0, 0),       # the 1x1 layers of stage 3: a group fills the CUs with 4 pixel splits instead of ~30
    (5, 256, 30, 40, 1024, 1, 1, 0, 0),
    (16, 96, 15, 20, 40, 1, 1, 0, 0),         # tile tails in both GEMM dimensions, the largest group
    (1, 64, 24, 32, 64, 1, 1, 0, 0),
    (3, 32, 17, 23, 48, 3, 1, 1, 0),          # 3x3 zero padding, odd plane
    (2, 24, 16, 16, 32, 3, 1, 1, 1),          # reflect
    (2, 64, 32, 32, 128, 1, 2, 0, 0),         # stride-2 1x1 (downsample)
])
def test_conv_wgrad_grouped_equals_layer_by_layer(G, C, H, W, M, K, stride, pad, mode):
    """prn_conv2d_wgrad_grouped (G same-shape layers, one launch) against G single-layer launches and torch fp64."""
    from planerecnet_amd import ops
    B = 8 if C >= 256 else 2
    g = torch.Generator().manual_seed(G * 100 + C)
    Ho, Wo = ops._out_hw(H, W, K, stride, pad, mode)
    xs = [torch.randn(B, C, H, W, generator=g).cuda() for _ in range(G)]
    dys = [torch.randn(B, M, Ho, Wo, generator=g).cuda() for _ in range(G)]
    dw = ops.conv_wgrad_grouped_raw(xs, dys, M, K, stride, pad, mode)
    assert dw.shape == (G, M, C, K, K)
    for i in range(G):
        one = ops.conv_wgrad_raw(xs[i], dys[i], M, K, stride, pad, mode)
        scale = float(one.abs().max())
        assert float((dw[i] - one).abs().max()) <= 2e-5 * scale, (i, float((dw[i] - one).abs().max()), scale)
    xr = xs[-1].double().cpu().requires_grad_(False)
    xp = F.pad(xr, (pad,) * 4, mode="reflect") if mode == 1 else xr
    wr = torch.zeros(M, C, K, K, dtype=torch.float64, requires_grad=True)
    F.conv2d(xp, wr, stride=stride, padding=0 if mode == 1 else pad).backward(dys[-1].double().cpu())
    assert float((dw[-1].double().cpu() - wr.grad).abs().max()) <= 2e-4 * float(wr.grad.abs().max())


def test_conv_wgrad_grouped_rejects():
    from planerecnet_amd import ops
    x, dy = torch.randn(1, 8, 8, 8).cuda(), torch.randn(1, 8, 8, 8).cuda()
    with pytest.raises(RuntimeError):
        ops.conv_wgrad_grouped_raw([x] * 17, [dy] * 17, 8, 1, 1, 0, 0)          # more than PRN_WGRAD_GROUP_MAX layers


def test_conv2d_is_transpose_safe():
    """A = I style check with asymmetric data: 1x1 conv with a permutation weight must permute channels."""
    from planerecnet_amd import ops
    d = dev()
    C = 96
    perm = torch.randperm(C, generator=torch.Generator().manual_seed(0))
    w = torch.zeros(C, C, 1, 1)
    w[torch.arange(C), perm] = 1.0
    x = torch.arange(2 * C * 5 * 7, dtype=torch.float32).view(2, C, 5, 7)
    y = ops.conv2d(x.to(d), w.to(d))
    assert torch.equal(y.cpu(), x[:, perm])


# toy shapes (tails in every GEMM dimension, C % 4 != 0, pad != 1, no mask) + the five call shapes of the network
# (Cin = Cout, 3x3, pad 1) at B = 2
DCN_CASES = [(2, 16, 12, 15, 24, 1, 1, True), (2, 32, 13, 16, 32, 2, 1, True), (1, 6, 9, 11, 5, 1, 0, True), (2, 20, 10, 12, 70, 1, 2, False),
             (2, 128, 120, 160, 128, 2, 1, True), (2, 128, 60, 80, 128, 1, 1, True), (2, 256, 60, 80, 256, 2, 1, True),
             (2, 256, 30, 40, 256, 1, 1, True), (2, 512, 30, 40, 512, 2, 1, True)]


@pytest.mark.parametrize("B,C,H,W,M,stride,pad,with_mask", DCN_CASES)
def test_deform_conv2d_torchvision_signature_fwd_bwd(B, C, H, W, M, stride, pad, with_mask):
    """ops.deform_conv2d(input, offset, weight, bias, stride, padding, mask=...) -- the call of models/dcn.py:59-66 -- against
    the fp64 oracle: output and all five gradients (input, offset, weight, bias, mask)."""
    from planerecnet_amd import ops
    from oracle.dcn_ref import deform_conv2d_ref
    Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    x = rnd(B, C, H, W, seed=1).requires_grad_(True)
    off = rnd(B, 18, Ho, Wo, seed=2, scale=1.7) + 0.137       # keep sampling points off the integer kinks
    off[0, 3, 0, 0] = 50.0                                     # far outside the image: contributes zero
    off[0, 4, 1, 1] = -50.0
    off = off.requires_grad_(True)
    msk = (2 * torch.sigmoid(rnd(B, 9, Ho, Wo, seed=6))).requires_grad_(True) if with_mask else None
    w = rnd(M, C, 3, 3, seed=3, scale=(9 * C) ** -0.5).requires_grad_(True)
    b = rnd(M, seed=4).requires_grad_(True)
    yr = deform_conv2d_ref(x, off, msk, w, b, stride, pad)
    go = rnd(*yr.shape, seed=5)
    leaves = [x, off, w, b] + ([msk] if with_mask else [])
    gr = torch.autograd.grad(yr, leaves, go)
    d = dev()
    xs = [t.detach().float().to(d).requires_grad_(True) for t in leaves]
    yd = ops.deform_conv2d(xs[0], xs[1], xs[2], xs[3], stride=(stride, stride), padding=(pad, pad), mask=xs[4] if with_mask else None)
    close(yd, yr, "dcn fwd")                                   # 2e-4 of the tensor max (the tolerance of every conv test)
    gd = torch.autograd.grad(yd, xs, go.float().to(d))
    for n, g1, g0 in zip(["dx", "d_offset", "dw", "db", "d_mask"], gd, gr):
        close(g1, g0, "dcn " + n, rtol=5e-4)


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 30, 40), (1, 16, 120, 160)])
def test_deform_conv2d_input_gradient_is_bit_stable_run_to_run(B, C, H, W):
    """The input gradient of the deformable convolution is a gather over CSR bins whose slots the fill hands out with atomics (arrival order); the bins are
    sorted (source point, weight) before the gather, so ten repetitions give ONE result -- on the one-launch construction (30x40: the bins of a tap plane in
    LDS) and on the five-launch one (120x160 > 15360 bins per plane).  Offsets of ~2 pixels r.m.s.: ~4-10 entries per bin, orders that did differ between
    runs before the sort (advisor, round 5)."""
    from planerecnet_amd import ops
    d = dev()
    M = C
    x = rnd(B, C, H, W, seed=1).float().to(d).requires_grad_(True)
    off = (rnd(B, 18, H, W, seed=2, scale=2.0) + 0.137).float().to(d).requires_grad_(True)
    msk = (2 * torch.sigmoid(rnd(B, 9, H, W, seed=6))).float().to(d).requires_grad_(True)
    w = rnd(M, C, 3, 3, seed=3, scale=(9 * C) ** -0.5).float().to(d).requires_grad_(True)
    go = rnd(B, M, H, W, seed=5).float().to(d)
    first = None
    for rep in range(10):
        y = ops.deform_conv2d(x, off, w, None, stride=(1, 1), padding=(1, 1), mask=msk)
        gx, goff, gm = torch.autograd.grad(y, [x, off, msk], go)
        ops.wgrad_join()
        if first is None:
            first = (gx.clone(), goff.clone(), gm.clone())
            assert float(gx.abs().max()) > 0
        else:
            assert torch.equal(gx, first[0]), "dx differs in repetition %d: %g" % (rep, (gx - first[0]).abs().max().item())
            assert torch.equal(goff, first[1]) and torch.equal(gm, first[2])


@pytest.mark.parametrize("stride", [1, 2])
def test_dcnv2_window_and_fallback_paths_agree_bit_for_bit(stride):
    """The windowed forward (csrc/prn_dcnv2.hip: the input window of a 64-pixel patch staged in LDS) and its per-patch fallback (four guarded
    global loads per sample, taken when the offsets spread the window beyond WROWS x WPITCH) use the same weights in the same order.  ONE
    sampling point per image is sent ~34 rows away (still inside the image): its whole patch falls back, and every OTHER output pixel of
    that patch -- whose nine samples did not change -- must come out bit-identical to the all-windowed run; the moved pixel and everything
    else is checked against the fp64 oracle."""
    from planerecnet_amd import ops
    from oracle.dcn_ref import deform_conv2d_ref
    B, C, H, W, M = 2, 32, 44, 48, 96
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    x = rnd(B, C, H, W, seed=1)
    w = rnd(M, C, 3, 3, seed=3, scale=(9 * C) ** -0.5)
    b = rnd(M, seed=4)
    msk = 2 * torch.sigmoid(rnd(B, 9, Ho, Wo, seed=6))
    off_small = rnd(B, 18, Ho, Wo, seed=2, scale=0.7) + 0.137
    off_far = off_small.clone()
    moved = [(0, 2, 3), (1, 3, 9)]                                            # (image, output row, output column)
    for (bi, ho, wo) in moved:
        off_far[bi, 2 * 4, ho, wo] = 34.3                                     # dy of the centre tap: row ho * stride + 34.3 < H
        assert ho * stride + 34.3 < H - 1
    d = dev()
    to = lambda t: t.float().to(d)
    ys = ops.deform_conv2d(to(x), to(off_small), to(w), to(b), stride=(stride, stride), padding=(1, 1), mask=to(msk))
    yf = ops.deform_conv2d(to(x), to(off_far), to(w), to(b), stride=(stride, stride), padding=(1, 1), mask=to(msk))
    close(ys, deform_conv2d_ref(x, off_small, msk, w, b, stride, 1), "dcn fwd, windowed patches")
    close(yf, deform_conv2d_ref(x, off_far, msk, w, b, stride, 1), "dcn fwd, windowed + fallback patches")
    same = torch.ones(B, 1, Ho, Wo, dtype=torch.bool, device=d)
    for (bi, ho, wo) in moved:
        same[bi, 0, ho, wo] = False
        assert not torch.equal(ys[bi, :, ho, wo], yf[bi, :, ho, wo])
    assert torch.equal(torch.where(same, ys, torch.zeros_like(ys)), torch.where(same, yf, torch.zeros_like(yf)))


def test_deform_conv2d_rejects_what_the_path_does_not_cover():
    from planerecnet_amd import ops
    d = dev()
    x, w = torch.zeros(1, 4, 8, 8, device=d), torch.zeros(4, 4, 3, 3, device=d)
    with pytest.raises(NotImplementedError):
        ops.deform_conv2d(x, torch.zeros(1, 18, 8, 8, device=d), w, padding=1, dilation=2)
    with pytest.raises(NotImplementedError):
        ops.deform_conv2d(x, torch.zeros(1, 50, 8, 8, device=d), torch.zeros(4, 4, 5, 5, device=d), padding=2)
    with pytest.raises(RuntimeError):
        ops.deform_conv2d(x, torch.zeros(1, 36, 8, 8, device=d), w, padding=1)          # two offset groups


def test_dcn_module_reference_form_equals_fused_block():
    """DeformableConv2d.forward (27-channel conv + raw-map operator) == the reference's statement sequence on the drop-in
    operator (dcn.py:52-67: clamp, 2*sigmoid, deform_conv2d(input=, offset=, weight=, bias=, padding=, mask=, stride=))."""
    from planerecnet_amd.dcn import DeformableConv2d
    d = dev()
    torch.manual_seed(0)
    m = DeformableConv2d(32, 48, 3, stride=2, padding=1, bias=True)
    with torch.no_grad():
        m.offset_conv.weight.normal_(0, 0.05); m.offset_conv.bias.normal_(0, 0.3)
        m.modulator_conv.weight.normal_(0, 0.05); m.modulator_conv.bias.normal_(0, 0.3)
    m = m.to(d)
    x = rnd(2, 32, 19, 23, seed=1).float().to(d).requires_grad_(True)
    go = rnd(2, 48, 10, 12, seed=2).float().to(d)
    ya = m(x)
    ga = torch.autograd.grad(ya, [x] + list(m.parameters()), go)
    yb = m.forward_reference_form(x)
    gb = torch.autograd.grad(yb, [x] + list(m.parameters()), go)
    close(ya, yb.cpu(), "module forms fwd", rtol=2e-5)
    for a, b_ in zip(ga, gb):
        close(a, b_.cpu(), "module forms grad", rtol=2e-4)


@pytest.mark.parametrize("stride", [1, 2])
def test_deform_conv_block_matches_unfused(stride):
    """offset conv + DCNv2 as one autograd node (x's two gradients summed in the dgrad epilogue) vs the two separate ops
    vs the fp64 oracle."""
    from planerecnet_amd import ops
    from oracle.dcn_ref import deform_conv2d_ref
    B, C, H, W, M = 2, 32, 14, 18, 40
    maxoff = max(H, W) / 4.0
    x = rnd(B, C, H, W, seed=1)
    w27 = rnd(27, C, 3, 3, seed=2, scale=0.4 * (9 * C) ** -0.5)
    b27 = rnd(27, seed=3, scale=0.3) + 0.137
    w = rnd(M, C, 3, 3, seed=4, scale=(9 * C) ** -0.5)
    b = rnd(M, seed=5)
    ts = [t.clone().requires_grad_(True) for t in (x, w27, b27, w, b)]
    om = F.conv2d(ts[0], ts[1], ts[2], stride=stride, padding=1)
    yr = deform_conv2d_ref(ts[0], om[:, :18].clamp(-maxoff, maxoff), 2 * torch.sigmoid(om[:, 18:]), ts[3], ts[4], stride, 1)
    go = rnd(*yr.shape, seed=6)
    gr = torch.autograd.grad(yr, ts, go)
    d = dev()
    xd, w27d, b27d, wd, bd = [t.detach().float().to(d) for t in (x, w27, b27, w, b)]
    # the four offset / modulator parameters as views of the merged storage (what dcn.DeformableConv2d keeps)
    leaves = [xd.requires_grad_(True), w27d[:18].requires_grad_(True), w27d[18:].requires_grad_(True), b27d[:18].requires_grad_(True),
              b27d[18:].requires_grad_(True), wd.requires_grad_(True), bd.requires_grad_(True)]
    yd = ops.deform_conv_block(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], w27d, b27d, leaves[5], leaves[6], stride, maxoff)
    close(yd, yr, "dcn block fwd")
    gd = torch.autograd.grad(yd, leaves, go.float().to(d))
    got = [gd[0], torch.cat([gd[1], gd[2]]), torch.cat([gd[3], gd[4]]), gd[5], gd[6]]
    for n, g1, g0 in zip(["dx", "dw27", "db27", "dw", "db"], got, gr):
        close(g1, g0, "dcn block " + n, rtol=5e-4)


def test_conv_fork_sums_both_input_gradients():
    """y, x_id = conv2d_fork(x, w); loss uses both: dx = dgrad(dy) + d(x_id), summed inside the dgrad epilogue."""
    from planerecnet_amd import ops
    x = rnd(2, 48, 11, 14, seed=1)
    w = rnd(32, 48, 1, 1, seed=2, scale=48 ** -0.5)
    g1, g2 = rnd(2, 32, 11, 14, seed=3), rnd(2, 48, 11, 14, seed=4)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ((F.conv2d(xr, wr) * g1).sum() + (xr * g2).sum()).backward()
    d = dev()
    xd, wd = x.float().to(d).requires_grad_(True), w.float().to(d).requires_grad_(True)
    y, xid = ops.conv2d_fork(xd, wd)
    assert xid.data_ptr() == xd.data_ptr()
    ((y * g1.float().to(d)).sum() + (xid * g2.float().to(d)).sum()).backward()
    close(xd.grad, xr.grad, "fork dx")
    close(wd.grad, wr.grad, "fork dw")
    # identity branch only / conv branch only
    xd2 = x.float().to(d).requires_grad_(True)
    y, xid = ops.conv2d_fork(xd2, wd)
    (xid * g2.float().to(d)).sum().backward()
    close(xd2.grad, g2, "fork identity-only")
    xd3 = x.float().to(d).requires_grad_(True)
    y, xid = ops.conv2d_fork(xd3, wd)
    (y * g1.float().to(d)).sum().backward()
    close(xd3.grad, torch.autograd.grad((F.conv2d(xr, wr) * g1).sum(), xr)[0], "fork conv-only")


def test_deferred_weight_gradients_accumulate_like_autograd():
    """ops.set_wgrad_async: the weight gradient is written to .grad on the side stream; two backward passes accumulate, a weight
    used twice in one graph gets both contributions, and a non-leaf weight falls back to the autograd path.  (A parameter
    gets its gradients either all through deferred ops or all through autograd -- never mixed; the model obeys that.)"""
    from planerecnet_amd import ops
    d = dev()
    x = rnd(2, 24, 10, 12, seed=1).float().to(d)
    w0 = rnd(16, 24, 3, 3, seed=2, scale=0.1).float().to(d)
    g = rnd(2, 16, 10, 12, seed=3).float().to(d)

    def run(deferred):
        w = w0.clone().requires_grad_(True)
        w2 = (w0 * 0.5).requires_grad_(True)
        scale = torch.ones((), device=d, requires_grad=True)
        ops.set_wgrad_async(deferred)
        try:
            for _ in range(2):                                            # two backward passes -> accumulation
                y = ops.conv2d(x, w, None, 1, 1) + ops.conv2d(x.flip(3), w, None, 1, 1)      # the same leaf weight twice
                y = y + ops.conv2d(x, w2 * scale, None, 1, 1)           # a non-leaf weight (autograd path even when deferred)
                (y * g).sum().backward()
                ops.wgrad_join()
        finally:
            ops.set_wgrad_async(False)
        torch.cuda.synchronize()
        return w.grad.clone(), scale.grad.clone(), w2.grad.clone()

    (wa, sa, va), (wb, sb, vb) = run(True), run(False)
    close(wa, wb.double().cpu(), "deferred dw", rtol=1e-5)
    close(sa, sb.double().cpu(), "deferred d(scale)", rtol=1e-5)
    close(va, vb.double().cpu(), "autograd-path dw", rtol=1e-5)


def test_flipped_weights_batched_matches_single():
    from planerecnet_amd import ops
    d = dev()
    ws = [rnd(24, 16, 3, 3, seed=1).float().to(d), rnd(8, 40, 1, 1, seed=2).float().to(d), rnd(5, 3, 7, 7, seed=3).float().to(d),
          rnd(12, 4, 3, 3, seed=4).float().to(d)]
    fw = ops.FlippedWeights([(ws[0], ws[0].shape), (ws[1], ws[1].shape), (ws[2], ws[2].shape), (ws[3], (12, 36, 1, 1))])
    fw.refresh()
    for w, v in zip(ws[:3], fw.views[:3]):
        assert torch.equal(v, w.flip(2, 3).transpose(0, 1).contiguous())
    assert torch.equal(fw.views[3].view(36, 12), ws[3].view(12, 36).t())
    assert ops.flip_transpose(ws[0]) is fw.views[0]                  # cache hit while the version is unchanged
    ws[0].mul_(2.0)
    assert ops.flip_transpose(ws[0]) is not fw.views[0]             # stale after an in-place update -> per-call flip
    fw.refresh()
    assert torch.equal(ops.flip_transpose(ws[0]), ws[0].flip(2, 3).transpose(0, 1).contiguous())


def test_side_stream_gradients_survive_block_reuse_without_allocator_registration():
    """ops.GRAD_RECORD_STREAM = 0 (the default): gradients are allocated under the weight-gradient stream and read under the compute
    stream WITHOUT Tensor.record_stream.  Stress the case that registration would protect: the compute stream is kept busy so that its
    reads of the gradients execute late, the gradients are released on the host at once, and the next iteration's side-stream
    launches ask the allocator for blocks of the same sizes.  Every side-stream batch starts with side.wait_stream(main), so the
    results must equal, bit for bit, those of the run with the registrations on."""
    from planerecnet_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(0)
    ws = [(torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(d).requires_grad_(True) for _ in range(6)]
    xs = [torch.randn(4, 64, 40, 40, generator=g).to(d) for _ in range(8)]
    big = torch.ones(64, 1024, 1024, device=d)                  # 256 MB: each pass over it keeps the compute stream busy

    def run(registered):
        saved = ops.GRAD_RECORD_STREAM
        ops.GRAD_RECORD_STREAM = registered
        ops.set_wgrad_async(True)
        sums = []
        try:
            for it in range(8):
                h = xs[it]
                for w in ws:
                    h = ops.conv2d(h, w, pad=1)
                h.square().mean().backward()
                ops.wgrad_join()
                for _ in range(3):
                    big.mul_(1.0)
                sums.append(torch.stack([w.grad.double().sum() for w in ws] + [w.grad.double().abs().max() for w in ws]))
                for w in ws:
                    w.grad = None
        finally:
            ops.set_wgrad_async(False)
            ops.GRAD_RECORD_STREAM = saved
        torch.cuda.synchronize()
        return torch.stack(sums).cpu()

    a, b = run(False), run(True)
    assert torch.isfinite(a).all() and float(a.abs().max()) > 0
    assert torch.equal(a, b)


def test_deform_conv_zero_offsets_equals_conv():
    from planerecnet_amd import ops
    d = dev()
    x = rnd(2, 24, 11, 13, seed=1).float()
    w = rnd(20, 24, 3, 3, seed=2, scale=0.1).float()
    off = torch.zeros(2, 18, 11, 13)        # offset 0, no mask
    y = ops.deform_conv2d(x.to(d), off.to(d), w.to(d), None, stride=1, padding=1)
    close(y, F.conv2d(x.double(), w.double(), padding=1), "dcn(0) == conv")


@pytest.mark.parametrize("B,C,H,W,res,relu,training", [(4, 64, 15, 20, False, True, True), (2, 256, 8, 10, True, True, True),
                                                       (3, 33, 7, 9, False, False, True), (2, 64, 12, 12, True, True, False)])
def test_batch_norm_fwd_bwd(B, C, H, W, res, relu, training):
    from planerecnet_amd import ops
    x = (rnd(B, C, H, W, seed=1) * 1.5 + 0.3).requires_grad_(True)
    g = (rnd(C, seed=2) * 0.2 + 1).requires_grad_(True)
    bt = rnd(C, seed=3, scale=0.2).requires_grad_(True)
    rm, rv = rnd(C, seed=4, scale=0.1), rnd(C, seed=5).abs() + 0.5
    r = rnd(B, C, H, W, seed=6).requires_grad_(True) if res else None
    rm0, rv0 = rm.clone(), rv.clone()
    yr = F.batch_norm(x, rm0, rv0, g, bt, training, 0.1, 1e-5)
    if res:
        yr = yr + r
    if relu:
        yr = F.relu(yr)
    go = rnd(*yr.shape, seed=7)
    leaves = [x, g, bt] + ([r] if res else [])
    gr = torch.autograd.grad(yr, leaves, go)
    d = dev()
    xs = [t.detach().float().to(d).requires_grad_(True) for t in leaves]
    rmd, rvd = rm.float().to(d), rv.float().to(d)
    yd = ops.batch_norm(xs[0], xs[1], xs[2], rmd, rvd, training, 1e-5, 0.1, xs[3] if res else None, relu)
    close(yd, yr, "bn fwd")
    if training:
        close(rmd, rm0, "running_mean")
        close(rvd, rv0, "running_var")
    gd = torch.autograd.grad(yd, xs, go.float().to(d))
    for n, g1, g0 in zip(["dx", "dgamma", "dbeta", "dres"], gd, gr):
        close(g1, g0, "bn " + n, rtol=5e-4)


@pytest.mark.parametrize("B,C,H,W,M", [(2, 24, 15, 21, 40), (2, 64, 30, 40, 128), (1, 256, 120, 160, 512)])
def test_stride2_1x1_fork_sums_the_two_gradients_of_its_input(B, C, H, W, M):
    """The downsample convolution of a stage's first Bottleneck (1x1, stride 2, models/backbone.py:45) as a fork: the gradient of its input's other readers
    (FPN lateral, depth decoder) arrives as the forked identity's gradient; the convolution's input gradient -- non-zero at the even positions only -- is
    scattered by the GEMM's strided epilogue and the two are summed.  Against fp64, odd sizes included.  (The block entry points add INTO the incoming
    gradient in place when they own it: tests/test_blocks_gpu.py.)"""
    from planerecnet_amd import ops
    d = dev()
    x = rnd(B, C, H, W, seed=1)
    w = rnd(M, C, 1, 1, seed=2, scale=C ** -0.5)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    g1, g2 = rnd(B, M, Ho, Wo, seed=3), rnd(B, C, H, W, seed=4)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = torch.autograd.grad([F.conv2d(xr, wr, stride=2), xr * 1.0], [xr, wr], [g1, g2])
    xd, wd = x.float().to(d).requires_grad_(True), w.float().to(d).requires_grad_(True)
    y, xid = ops.conv2d_fork(xd, wd, stride=2)
    g2d = g2.float().to(d)
    keep = g2d.clone()
    got = torch.autograd.grad([y, xid * 1.0], [xd, wd], [g1.float().to(d), g2d])
    ops.wgrad_join()
    close(got[0], ref[0], "dx")
    close(got[1], ref[1], "dw", rtol=5e-4)
    assert torch.equal(g2d, keep), "a plain operator wrote into a gradient tensor autograd handed it"


@pytest.mark.parametrize("B,Ca,Cb,H,W", [(4, 64, 32, 15, 20), (2, 128, 128, 60, 80), (3, 5, 9, 7, 9), (8, 16, 24, 30, 40)])
def test_batch_norm_relu_cat_equals_two_layers_and_cat(B, Ca, Cb, H, W):
    """cat([relu(bn_a(xa)), relu(bn_b(xb))], 1) with both layers writing into / reading from channel slices of one buffer
    (prn_bn_train_fwd_into / prn_bn_bwd_from): one-pass kernels (<= 12288 values per channel) and the two-launch path, HW % 4 != 0."""
    from planerecnet_amd import ops
    d = dev()
    mods, xs, ref_in = [], [], []
    for k, C in enumerate((Ca, Cb)):
        m = torch.nn.BatchNorm2d(C, eps=0.001, momentum=0.01)
        with torch.no_grad():
            m.weight.copy_(rnd(C, seed=10 + k) * 0.2 + 1)
            m.bias.copy_(rnd(C, seed=20 + k, scale=0.2))
            m.running_mean.copy_(rnd(C, seed=30 + k, scale=0.1))
            m.running_var.copy_(rnd(C, seed=40 + k).abs() + 0.5)
        mods.append(m)
        xs.append(rnd(B, C, H, W, seed=50 + k) * 1.5 + 0.3)
    import copy
    ref = [copy.deepcopy(m).double().train() for m in mods]
    xr = [x.clone().requires_grad_(True) for x in xs]
    yr = torch.cat([F.relu(ref[0](xr[0])), F.relu(ref[1](xr[1]))], 1)
    go = rnd(*yr.shape, seed=7)
    gr = torch.autograd.grad(yr, xr + [ref[0].weight, ref[0].bias, ref[1].weight, ref[1].bias], go)
    dm = [m.float().to(d).train() for m in mods]
    xd = [x.float().to(d).requires_grad_(True) for x in xs]
    yd = ops.batch_norm_relu_cat(dm[0], xd[0], dm[1], xd[1])
    assert yd.shape == yr.shape and yd.is_contiguous()
    close(yd, yr, "bn-cat fwd")
    for k in range(2):
        close(dm[k].running_mean, ref[k].running_mean, "running_mean %d" % k)
        close(dm[k].running_var, ref[k].running_var, "running_var %d" % k)
    gd = torch.autograd.grad(yd, xd + [dm[0].weight, dm[0].bias, dm[1].weight, dm[1].bias], go.float().to(d))
    for n, g1, g0 in zip(["dxa", "dxb", "dgamma_a", "dbeta_a", "dgamma_b", "dbeta_b"], gd, gr):
        close(g1, g0, "bn-cat " + n, rtol=5e-4)
    # bit-identical to the two separate layers followed by torch.cat (same kernels, other addresses)
    dm2 = [copy.deepcopy(m).float().to(d).train() for m in mods]
    y2 = torch.cat([ops.batch_norm_module(dm2[0], xd[0].detach(), None, True), ops.batch_norm_module(dm2[1], xd[1].detach(), None, True)], 1)
    assert torch.equal(yd.detach(), y2)


@pytest.mark.parametrize("B,C,H,W", [(2, 256, 16, 16), (2, 128, 30, 40), (1, 128, 9, 7), (1, 128, 96, 88)])   # last: 1024-thread blocks
def test_group_norm_relu_fwd_bwd(B, C, H, W):
    from planerecnet_amd import ops
    x = (rnd(B, C, H, W, seed=1) * 2 + 0.5).requires_grad_(True)
    g = (rnd(C, seed=2) * 0.2 + 1).requires_grad_(True)
    bt = rnd(C, seed=3, scale=0.3).requires_grad_(True)
    yr = F.relu(F.group_norm(x, 32, g, bt, 1e-5))
    go = rnd(*yr.shape, seed=4)
    gr = torch.autograd.grad(yr, [x, g, bt], go)
    d = dev()
    xs = [t.detach().float().to(d).requires_grad_(True) for t in (x, g, bt)]
    yd = ops.group_norm_relu(xs[0], xs[1], xs[2])
    close(yd, yr, "gn fwd")
    gd = torch.autograd.grad(yd, xs, go.float().to(d))
    for n, g1, g0 in zip(["dx", "dgamma", "dbeta"], gd, gr):
        close(g1, g0, "gn " + n, rtol=5e-4)


@pytest.mark.parametrize("H,W,Ho,Wo", [(12, 16, 6, 8), (6, 8, 12, 16), (15, 20, 24, 24), (5, 7, 3, 4)])
def test_resize_bilinear_fork_sums_the_second_gradient_in_the_kernel(H, W, Ho, Wo):
    """y, x_id = resize_bilinear_fork(x): the gradient arriving at x_id is added inside the resize's backward launch
    (prn_resize_bilinear_bwd_add: x0.5, x2 and generic kernels); also with only one of the two outputs used."""
    from planerecnet_amd import ops
    d = dev()
    x = rnd(2, 5, H, W, seed=1)
    go, g2 = rnd(2, 5, Ho, Wo, seed=2), rnd(2, 5, H, W, seed=3)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=False)
    (gr,) = torch.autograd.grad((yr * go).sum() + (xr * xr * g2).sum(), xr)
    xd = x.float().to(d).requires_grad_(True)
    yd, xid = ops.resize_bilinear_fork(xd, (Ho, Wo))
    close(yd, yr, "resize fork fwd")
    assert xid.data_ptr() == xd.data_ptr()
    (gd,) = torch.autograd.grad((yd * go.float().to(d)).sum() + (xid * xid * g2.float().to(d)).sum(), xd)
    close(gd, gr, "resize fork: both gradients")
    xd2 = x.float().to(d).requires_grad_(True)
    yd2, xid2 = ops.resize_bilinear_fork(xd2, (Ho, Wo))
    (g_only_id,) = torch.autograd.grad((xid2 * g2.float().to(d)).sum(), xd2)
    close(g_only_id, g2, "resize fork: identity only")
    xd3 = x.float().to(d).requires_grad_(True)
    yd3, _ = ops.resize_bilinear_fork(xd3, (Ho, Wo))
    (g_only_y,) = torch.autograd.grad((yd3 * go.float().to(d)).sum(), xd3)
    (gr_y,) = torch.autograd.grad((F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=False) * go).sum(), xr)
    close(g_only_y, gr_y, "resize fork: resized output only")


@pytest.mark.parametrize("H,W,Ho,Wo", [(12, 16, 6, 8), (6, 8, 12, 16), (15, 20, 40, 40), (30, 40, 36, 36), (16, 24, 4, 6), (15, 20, 24, 24),
                                        (2, 2, 4, 4), (4, 6, 2, 3), (3, 5, 6, 10), (1, 1, 2, 2)])
def test_resize_bilinear_fwd_bwd(H, W, Ho, Wo):
    from planerecnet_amd import ops
    x = rnd(2, 5, H, W, seed=1).requires_grad_(True)
    yr = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False)
    go = rnd(*yr.shape, seed=2)
    (gr,) = torch.autograd.grad(yr, [x], go)
    d = dev()
    xd = x.detach().float().to(d).requires_grad_(True)
    yd = ops.resize_bilinear(xd, (Ho, Wo))
    close(yd, yr, "resize fwd", rtol=1e-5)
    (gd,) = torch.autograd.grad(yd, [xd], go.float().to(d))
    close(gd, gr, "resize bwd", rtol=1e-5)


def test_resize_up2_adjoint_confines_a_non_finite_gradient_to_the_pixels_that_use_it():
    """x2 upsample backward: an inf in dy reaches exactly the input pixels whose interpolation weight on that output is non-zero; every other
    pixel keeps the value it has with that entry removed.  Taps of the straight-line 4 x 4 footprint that lie outside the map or carry weight 0
    are selected away, never multiplied (0 * inf = NaN) -- advisor, round 4.  (ATen's own backward multiplies the clamped border taps by their
    zero weight and does leak NaN there, so the expectation is built from the interpolation matrix, not from ATen.)"""
    from planerecnet_amd import ops
    H, W = 6, 8
    go = rnd(1, 3, 2 * H, 2 * W, seed=4)
    Uh = F.interpolate(torch.eye(H, dtype=torch.float64)[None, None], size=(2 * H, H), mode="bilinear", align_corners=False)[0, 0]   # [2H, H]
    Uw = F.interpolate(torch.eye(W, dtype=torch.float64)[None, None], size=(W, 2 * W), mode="bilinear", align_corners=False)[0, 0].t()   # [2W, W]
    for (oh, ow) in ((0, 0), (0, 5), (2 * H - 1, 2 * W - 1), (5, 0), (6, 7)):
        g = go.clone()
        g[0, 1, oh, ow] = float("inf")
        g0 = go.clone()
        g0[0, 1, oh, ow] = 0.0
        want = torch.einsum("oh,bcop,pw->bchw", Uh, g0, Uw)                     # the adjoint without that entry
        hit = torch.zeros(1, 3, H, W, dtype=torch.bool)
        hit[0, 1] = (Uh[oh] != 0)[:, None] & (Uw[ow] != 0)[None, :]
        xd = torch.zeros(1, 3, H, W, device=dev(), requires_grad=True)
        (gd,) = torch.autograd.grad(ops.resize_bilinear(xd, (2 * H, 2 * W)), [xd], g.float().to(dev()))
        gd = gd.cpu()
        assert not torch.isnan(gd).any(), (oh, ow)
        assert torch.equal(torch.isinf(gd), hit), (oh, ow)
        assert (gd[~hit].double() - want[~hit]).abs().max().item() <= 1e-5 * want.abs().max().item()


@pytest.mark.parametrize("H,W,Ho,Wo", [(15, 20, 30, 40), (30, 40, 15, 20), (7, 9, 16, 11)])
def test_resize_bilinear_with_addend_fwd_bwd(H, W, Ho, Wo):
    """resize(x) + addend from one launch (the level sum of SOLOv2MaskHead): value, and both gradients (d addend = dy as is)."""
    from planerecnet_amd import ops
    x = rnd(2, 5, H, W, seed=1).requires_grad_(True)
    a = rnd(2, 5, Ho, Wo, seed=3).requires_grad_(True)
    yr = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False) + a
    go = rnd(*yr.shape, seed=2)
    gr = torch.autograd.grad(yr, [x, a], go)
    d = dev()
    xd, ad = x.detach().float().to(d).requires_grad_(True), a.detach().float().to(d).requires_grad_(True)
    yd = ops.resize_bilinear(xd, (Ho, Wo), ad)
    close(yd, yr, "resize+addend fwd", rtol=1e-5)
    gd = torch.autograd.grad(yd, [xd, ad], go.float().to(d))
    close(gd[0], gr[0], "resize+addend d x", rtol=1e-5)
    assert torch.equal(gd[1].cpu(), go.float())


def test_resize_matches_scale_factor_semantics():
    """x0.5 / x0.25 / x2 with scale_factor (recompute_scale_factor=False) == size-based resize at these exact ratios."""
    from planerecnet_amd import ops
    d = dev()
    x = rnd(1, 3, 16, 24, seed=1)
    for sf in (0.5, 0.25, 2.0):
        yr = F.interpolate(x, scale_factor=sf, mode="bilinear", align_corners=False, recompute_scale_factor=False)
        yd = ops.resize_bilinear(x.float().to(d), yr.shape[2:])
        close(yd, yr, f"resize x{sf}", rtol=1e-5)


@pytest.mark.parametrize("H,W", [(24, 32), (15, 21), (17, 20), (9, 4)])
def test_maxpool_fwd_bwd(H, W):
    from planerecnet_amd import ops
    x = rnd(2, 6, H, W, seed=1).requires_grad_(True)
    yr = F.max_pool2d(x, 3, 2, 1)
    go = rnd(*yr.shape, seed=2)
    (gr,) = torch.autograd.grad(yr, [x], go)
    d = dev()
    xd = x.detach().float().to(d).requires_grad_(True)
    yd = ops.max_pool_3x3_s2(xd)
    close(yd, yr, "maxpool fwd", rtol=1e-6)
    (gd,) = torch.autograd.grad(yd, [xd], go.float().to(d))
    close(gd, gr, "maxpool bwd", rtol=1e-5)


def test_ops_reject_cpu_tensors():
    from planerecnet_amd import ops
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 4, 4, 4), torch.zeros(4, 4, 1, 1))


WINO_CASES = [
    # B, C, H, W, M, mode, bias, addend, epi
    (2, 64, 30, 40, 96, 0, True, False, 0),
    (2, 64, 30, 40, 96, 1, True, True, 1),
    (1, 128, 64, 64, 64, 0, False, True, 0),
    (8, 256, 15, 20, 256, 0, False, False, 0),
    (3, 70, 17, 48, 130, 0, True, False, 0),
    (1, 64, 12, 20, 64, 0, False, False, 0),        # 15 tiles: the padded 16th column of the transform-domain operands must not count
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv3x3_winograd_fwd_bwd(case, monkeypatch):
    """The F(4x4, 3x3) path (include/prn.h: prn_conv3x3_winograd) against torch CPU fp64, forward / input gradient (the
    rotated-tap transform) / weight gradient (direct kernel), and against the direct kernel it replaces."""
    from planerecnet_amd import ops
    B, C, H, W, M, mode, has_b, has_a, epi = case
    monkeypatch.setattr(ops, "WINOGRAD_MIN_TILES", 1)
    assert ops.winograd_ok(B, C, H, W, M, 3, 1, 1, mode, epi)
    x, w = rnd(B, C, H, W, seed=1), rnd(M, C, 3, 3, seed=2, scale=(C * 9) ** -0.5)
    b = rnd(M, seed=3) if has_b else None
    a = rnd(B, M, H, W, seed=4) if has_a else None
    g = rnd(B, M, H, W, seed=5)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = ref_conv(xr, wr, b, 1, 1, mode, a, epi)
    yr.backward(g)

    def run():
        xd, wd = x.float().to(dev()).requires_grad_(), w.float().to(dev()).requires_grad_()
        bd = b.float().to(dev()) if has_b else None
        ad = a.float().to(dev()) if has_a else None
        y = ops.conv2d(xd, wd, bd, 1, 1, mode, epi, ad)
        y.backward(g.float().to(dev()))
        return y, xd.grad, wd.grad
    y, dx, dw = run()
    close(y, yr, "winograd fwd")
    close(dx, xr.grad, "winograd dgrad")
    close(dw, wr.grad, "winograd wgrad")
    ops.WINOGRAD = False
    try:
        y0, dx0, dw0 = run()
    finally:
        ops.WINOGRAD = True
    close(y, y0.double().cpu(), "winograd vs direct fwd", rtol=5e-5)
    close(dx, dx0.double().cpu(), "winograd vs direct dgrad", rtol=5e-5)
    close(dw, dw0.double().cpu(), "winograd vs direct wgrad", rtol=5e-5)


def test_winograd_weights_batched_refresh():
    """WinogradWeights.refresh(): one launch for several weights; cached operands are used until the weight changes."""
    from planerecnet_amd import ops
    ws = [torch.nn.Parameter(rnd(M, C, 3, 3, seed=M).float().to(dev())) for M, C in ((64, 64), (96, 70), (256, 128))]
    ww = ops.WinogradWeights(ws)
    ww.refresh()
    for w in ws:
        U, Ut = ops.winograd_weights(w)
        assert U.data_ptr() == ops._WINO[w.data_ptr()][2].data_ptr()
        with torch.no_grad():
            fresh = torch.nn.Parameter(w.detach().clone())
        U2, Ut2 = ops.winograd_weights(fresh)                 # not registered: computed on the spot
        assert torch.equal(U, U2) and torch.equal(Ut, Ut2)
        # U[z, m, c] of the rotated, transposed weight == Ut[z, c, m]
        U3, _ = ops.winograd_weights(torch.nn.Parameter(w.detach().flip(2, 3).transpose(0, 1).contiguous()))
        assert torch.equal(U3, Ut)
    old = ops._WINO[ws[0].data_ptr()][2]
    old_vals = old.clone()
    with torch.no_grad():
        ws[0].mul_(2.0)
    U = ops.winograd_weights(ws[0])[0]                        # stale entry is not used: recomputed for the new values
    assert U.data_ptr() != old.data_ptr() and torch.allclose(U, 2.0 * old_vals, rtol=1e-6, atol=0)


@pytest.mark.parametrize("C,M", [(66, 64), (128, 96)])
def test_ragged_winograd_matches_direct_ragged_kernel(C, M, monkeypatch):
    """Ragged batch (the instance head's five grids, shared weights) on the Winograd path == the direct ragged GEMM and the
    per-segment dense convs: forward, input gradient, weight gradient."""
    from planerecnet_amd import ops
    d = dev()
    B, sizes = 8, [(40, 40), (36, 36), (24, 24), (16, 16), (12, 12)]
    xs = [rnd(B, C, h, w, seed=10 + i).float().to(d) for i, (h, w) in enumerate(sizes)]
    w = rnd(M, C, 3, 3, seed=2, scale=(C * 9) ** -0.5).float().to(d)
    gos = [rnd(B, M, h, w_, seed=20 + i).float().to(d) for i, (h, w_) in enumerate(sizes)]

    def run(wino):
        monkeypatch.setattr(ops, "WINOGRAD", wino)
        rs = ops.RaggedShape(B, sizes)
        assert rs.supported() and bool(rs.winograd(C, M, 3)) == wino
        xl = [t.clone().requires_grad_(True) for t in xs]
        wl = w.clone().requires_grad_(True)
        y = ops.ragged_conv2d(rs.pack(xl), wl, None, rs)
        return rs.unpack(y, M), torch.autograd.grad((y * rs.pack(gos)).sum(), xl + [wl])
    ya, ga = run(True)
    yb, gb = run(False)
    for i, (a, r) in enumerate(zip(ya, yb)):
        close(a, r.double().cpu(), "ragged winograd y[%d]" % i, rtol=5e-5)
        close(a, F.conv2d(xs[i].double().cpu(), w.double().cpu(), padding=1), "ragged winograd y[%d] vs torch" % i)
    for i, (a, r) in enumerate(zip(ga, gb)):
        close(a, r.double().cpu(), "ragged winograd grad[%d]" % i, rtol=1e-4)


def test_fused_mask_loss_matches_operator_chain():
    """Dice + lava in one pass (prn_mask_loss_fwd / _bwd) against the reference formulas evaluated operator by operator in
    fp64 (models/functions/losses.py:69-118,169-197): values and logit gradients, incl. an image without positive cells, an
    image whose depth-gradient map is all zero, and non-unit upstream gradients."""
    from planerecnet_amd.losses import _MaskLoss
    d = dev()
    g = torch.Generator().manual_seed(0)
    B, fh, fw = 4, 12, 20
    rows = [9, 0, 17, 11]                                     # image 1 has no positive cell
    P = sum(rows)
    img = torch.repeat_interleave(torch.arange(B), torch.tensor(rows))
    z = (torch.randn(P, fh, fw, generator=g, dtype=torch.float64) * 2).requires_grad_(True)
    t = (torch.rand(P, fh, fw, generator=g) < 0.3).to(torch.uint8)
    adj = torch.rand(B, 1, fh, fw, generator=g, dtype=torch.float64)
    gsum = adj.flatten(1).sum(1) * 16.0
    adj[3] = 0.0
    gsum[3] = 0.0                                             # image 3 does not qualify
    npos = torch.tensor(rows, dtype=torch.float64)
    w_ins, w_lav = 3.0, 1.0
    p = torch.sigmoid(z)
    pf, tf = p.flatten(1), t.flatten(1).double()
    dice = 1 - 2 * (pf * tf).sum(1) / ((pf * pf).sum(1) + 0.001 + (tf * tf).sum(1) + 0.001)
    ins_ref = dice.mean() * w_ins
    num = torch.zeros(B, dtype=torch.float64).index_add_(0, img, (p * adj[img, 0]).flatten(1).sum(1))
    ok = (gsum > 0) & (npos > 0)
    lav_ref = torch.where(ok, num / (gsum * npos).clamp(min=1e-30), torch.zeros_like(num)).sum() / ok.sum().clamp(min=1) * w_lav
    gz_ref, = torch.autograd.grad(0.7 * ins_ref + 1.9 * lav_ref, z)
    zd = z.detach().float().to(d).requires_grad_(True)
    ins, lav = _MaskLoss.apply(zd, t.to(d), img.to(d), adj.float().to(d), gsum.float().to(d), npos.float().to(d), w_ins, w_lav)
    assert abs(float(ins) - float(ins_ref)) <= 1e-5 * abs(float(ins_ref)) and abs(float(lav) - float(lav_ref)) <= 1e-5 * abs(float(lav_ref))
    gz, = torch.autograd.grad(0.7 * ins + 1.9 * lav, zd)
    close(gz, gz_ref, "mask loss d logits", rtol=1e-5)
    # Dice only (no lava inputs)
    ins2, lav2 = _MaskLoss.apply(zd, t.to(d), img.to(d), None, None, None, w_ins, w_lav)
    assert abs(float(ins2) - float(ins_ref)) <= 1e-5 * abs(float(ins_ref)) and float(lav2) == 0.0


def test_plane_prior_block_equals_reference_form():
    """prn_plane_prior_fwd / _wgrad against the reference's statement sequence (planerecnet.py:586-594): sigmoid(K . M) for every
    kernel at full mask resolution, conv1x1, x0.25 bilinear resize -- in fp64 on the CPU; output and the conv1x1 gradients."""
    from planerecnet_amd import ops
    d = dev()
    B, E, h, w, NK, Fo = 2, 16, 24, 32, 150, 24
    seg = rnd(B, E, h, w, seed=1)
    kern = rnd(B, NK, E, seed=2, scale=0.4)
    w1 = rnd(Fo, NK, 1, 1, seed=3, scale=NK ** -0.5).requires_grad_(True)
    b1 = rnd(Fo, seed=4).requires_grad_(True)
    sig = torch.stack([torch.sigmoid(F.conv2d(seg[b:b + 1], kern[b].reshape(NK, E, 1, 1)))[0] for b in range(B)])
    ref = F.interpolate(F.conv2d(sig, w1, b1), scale_factor=0.25, mode="bilinear", align_corners=False)
    go = rnd(*ref.shape, seed=5)
    gr = torch.autograd.grad(ref, [w1, b1], go)
    xs = [w1.detach().float().to(d).requires_grad_(True), b1.detach().float().to(d).requires_grad_(True)]
    out = ops.plane_prior(seg.float().to(d), kern.float().to(d), xs[0], xs[1])
    close(out, ref, "plane prior fwd")
    gd = torch.autograd.grad(out, xs, go.float().to(d))
    close(gd[0], gr[0], "plane prior dw1", rtol=5e-4)
    close(gd[1], gr[1], "plane prior db1", rtol=5e-4)


@pytest.mark.parametrize("B,C,H,W,with_prev", [(2, 96, 16, 20, True), (1, 64, 9, 11, False), (2, 128, 32, 40, True)])
def test_fpn_level_block_equals_operator_sequence(B, C, H, W, with_prev):
    """prn_fpn_level_fwd (models/fpn.py:51-63) vs torch CPU fp64: lateral 1x1 + bottom-up resized addend, 3x3 + ReLU."""
    from planerecnet_amd import ops
    d = dev()
    Fo = 64
    x = rnd(B, C, H, W, seed=1)
    wl, bl = rnd(Fo, C, 1, 1, seed=2, scale=C ** -0.5), rnd(Fo, seed=3)
    wo, bo = rnd(Fo, Fo, 3, 3, seed=4, scale=(9 * Fo) ** -0.5), rnd(Fo, seed=5)
    prev = rnd(B, Fo, 2 * H, 2 * W - 1, seed=6) if with_prev else None
    lat_ref = F.conv2d(x, wl, bl)
    if prev is not None:
        lat_ref = lat_ref + F.interpolate(prev, size=(H, W), mode="bilinear", align_corners=False)
    p_ref = F.relu(F.conv2d(lat_ref, wo, bo, padding=1))
    with torch.no_grad():
        lat, p = ops.fpn_level(x.float().to(d), wl.float().to(d), bl.float().to(d), None if prev is None else prev.float().to(d), wo.float().to(d),
                               bo.float().to(d), True)
    close(lat, lat_ref, "fpn lateral")
    close(p, p_ref, "fpn output")


@pytest.fixture
def split_everywhere():
    """PRN_SPLIT_GEMM=2 for one test: every plain GEMM the split kernel can take runs on it (plans and workspace sizes are keyed by the options)."""
    from planerecnet_amd import ops
    old = ops.set_split_gemm(mode=2)
    yield ops
    ops.set_split_gemm(**old)


def _gemm_errors(y, w, x, bias, add, epi):
    """(max, rms) of (y - fp64 reference) / (sum_k |w||x| + |bias| + |addend|) per element."""
    wd, xd = w.double().cpu(), x.double().cpu()
    ref = torch.einsum("mk,bkp->bmp", wd, xd)
    mag = torch.einsum("mk,bkp->bmp", wd.abs(), xd.abs())
    if bias is not None:
        ref = ref + bias.double().cpu()[None, :, None]; mag = mag + bias.double().cpu().abs()[None, :, None]
    if add is not None:
        ref = ref + add.double().cpu(); mag = mag + add.double().cpu().abs()
    if epi == 1:
        ref = torch.relu(ref)
    e = (y.double().cpu() - ref) / (mag + 1e-30)
    return float(e.abs().max()), float(e.pow(2).mean().sqrt())


@pytest.fixture(params=["f16", "bf16"])
def split_kind(request):
    """Both piece formats of csrc/prn_gemm_split.hip: two fp16 pieces / three products (default) and three bf16 pieces / six products."""
    from planerecnet_amd import ops
    old = ops.set_split_gemm(kind=request.param)
    yield request.param
    ops.set_split_gemm(**old)


@pytest.mark.parametrize("M,K,B,HW,bias,add,epi", [
    (1024, 256, 2, 1200, False, False, 0),      # stage-3 expand
    (256, 1024, 2, 1200, True, True, 1),        # stage-3 reduce: K split (few tiles), bias + residual + ReLU through the reduce kernel
    (200, 72, 3, 1205, True, True, 1),          # tails in M, K (72 = 2 slices + 8) and pixels (odd plane)
    (128, 32, 1, 100, False, False, 0),         # one tile, one slice
    (512, 2048, 8, 300, False, False, 0),       # stage-4 reduce: 4 K splits
    (64, 256, 1, 19200, True, False, 0),        # half-empty row tile
])
def test_split_gemm_is_an_fp32_gemm(M, K, B, HW, bias, add, epi, split_everywhere, split_kind):
    """csrc/prn_gemm_split.hip (two fp16 pieces per operand scaled into range by exact powers of two, three fp16 MFMA products -- or three exact
    bf16 pieces, six bf16 MFMA products --, fp32 accumulate) against fp64, next to the
    fp32 MFMA kernel on the same operands: its error is fp32 ROUNDING error -- at every element below 4e-7 of sum|w||x| (or twice the
    fp32 kernel's maximum on that shape; that kernel's own maximum is 1e-7 .. 3.7e-7 depending on its K split), rms below 2.5e-8 (or
    1.5x the fp32 kernel's 1.2e-8 .. 2.2e-8) -- four orders of magnitude from a bf16 or TF32 product's 1e-3."""
    ops = split_everywhere
    lib, _p, _stream, check = ops.lib, ops._p, ops._stream, ops.check
    g = torch.Generator().manual_seed(M + K)
    x = (torch.rand(B, K, HW, generator=g) * 2 - 1).cuda()
    w = ((torch.rand(M, K, generator=g) * 2 - 1) * K ** -0.5).cuda()
    bv = torch.randn(M, generator=g).cuda() if bias else None
    av = torch.randn(B, M, HW, generator=g).cuda() if add else None
    assert ops.gemm_pipe(M, K, B, HW, 1) >= 1
    out = {}
    for mode in (2, 0):
        ops.set_split_gemm(mode=mode)
        _, ref, nbytes, _, _ = ops._desc(B, K, 1, HW, M, 1, 1, 0, 1, HW, 0, 1, epi)
        assert lib.prn_conv2d_kernel_kind(ref) == (0 if mode == 0 else (3 if ops.gemm_pipe(M, K, B, HW, 1) > 1 else 2))
        ws = torch.full((max(nbytes, 16) // 4,), float("nan"), device="cuda")
        y = torch.full((B, M, HW), float("nan"), device="cuda")
        check(lib.prn_conv2d_fwd(ref, _p(x), _p(w), _p(bv), _p(av), _p(y), _p(ws), _stream()), "conv")
        torch.cuda.synchronize()
        assert bool(torch.isfinite(y).all())
        out[mode] = _gemm_errors(y, w, x, bv, av, epi)
    (smax, srms), (fmax, frms) = out[2], out[0]
    assert smax <= max(2.0 * fmax, 4e-7) and srms <= max(1.5 * frms, 2.5e-8), (out,)


def test_split_gemm_batched_and_special_values(split_everywhere, split_kind):
    """prn_gemm_batched on the split kernel (the Winograd products: z = 36 independent GEMMs), and operand values the split must
    survive: exact zeros, powers of two, denormal-range and large magnitudes, negative numbers (pieces carry the operand's sign)."""
    ops = split_everywhere
    lib, _p, _stream, check = ops.lib, ops._p, ops._stream, ops.check
    M, C, P, nb = 256, 96, 640, 5
    assert ops.gemm_pipe(M, C, 1, P, nb) == 1
    g = torch.Generator().manual_seed(3)
    U = torch.randn(nb, M, C, generator=g)
    V = torch.randn(nb, C, P, generator=g)
    U[0, :, :8] = 0.0; U[1, :, 8:16] = 2.0 ** -20; U[2, 3] = 1.0; V[0, :4] = 0.0; V[3] *= 1e18; U[3] *= 1e-18; V[4, :, ::7] = -4096.0
    V[2, 5] = 1e-38                                                            # below bf16's / fp32's normal range after the second slice
    Y = torch.empty(nb, M, P, device="cuda")
    nws = lib.prn_gemm_batched_ws_bytes(M, C, P, nb, ops.opts_ref())
    assert nws == lib.prn_split_images_bytes(M, C, nb)
    ws = torch.empty(nws // 4, device="cuda")
    check(lib.prn_gemm_batched(M, C, P, nb, _p(U.cuda()), None, _p(V.cuda()), _p(Y), _p(ws), ops.opts_ref(), _stream()), "batched")
    torch.cuda.synchronize()
    ref = torch.bmm(U.double(), V.double())
    mag = torch.bmm(U.double().abs(), V.double().abs())
    e = ((Y.double().cpu() - ref) / (mag + 1e-300)).abs().max()
    assert bool(torch.isfinite(Y).all()) and float(e) <= 6e-7, float(e)


def test_split_gemm_on_operands_spanning_twelve_decades(split_everywhere, split_kind):
    """Where the two piece formats differ: log-normal operands whose rows / columns AND elements within them span ~12 decades.  The bf16
    pieces cover fp32's exponent range element by element and stay at the fp32 kernel's error; the fp16 pieces are scaled per weight row /
    activation column, so elements more than 2^17 below their row's / column's largest lose relative precision: the rms error is unchanged,
    the maximum (at outputs whose sum|a||b| is carried by such elements) grows to ~1e-5 of sum|a||b| -- still ten times below ONE bf16 product."""
    ops = split_everywhere
    lib, _p, _stream, check = ops.lib, ops._p, ops._stream, ops.check
    M, K, B, HW = 1024, 256, 2, 1200
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B, K, HW, generator=g) * torch.exp(torch.randn(B, K, HW, generator=g) * 3) * torch.exp(torch.randn(B, 1, HW, generator=g) * 6)).cuda()
    w = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, K, generator=g) * 3) * torch.exp(torch.randn(M, 1, generator=g) * 6)).cuda()
    out = {}
    for mode in (2, 0):
        ops.set_split_gemm(mode=mode)
        _, ref, nbytes, _, _ = ops._desc(B, K, 1, HW, M, 1, 1, 0, 1, HW, 0, 1, 0)
        ws = torch.empty(max(nbytes, 16) // 4, device="cuda")
        y = torch.empty(B, M, HW, device="cuda")
        check(lib.prn_conv2d_fwd(ref, _p(x), _p(w), None, None, _p(y), _p(ws), _stream()), "conv")
        torch.cuda.synchronize()
        assert bool(torch.isfinite(y).all())
        out[mode] = _gemm_errors(y, w, x, None, None, 0)
    (smax, srms), (fmax, frms) = out[2], out[0]
    assert srms <= 1.5 * frms, out
    assert smax <= (5e-5 if split_kind == "f16" else 3.0 * fmax), out


def test_split_gemm_weight_images_follow_the_weight(split_everywhere, split_kind, monkeypatch):
    """ops.split_images: a parameter's images are cut once, reused while its version counter stands still, re-cut after an in-place
    update (by the launch itself when nobody called split_refresh_all, by ONE batched launch when the model does), and dropped with
    the parameter; a temporary weight tensor is cut inside its launch and never cached."""
    ops = split_everywhere
    monkeypatch.setattr(ops, "SPLIT_CACHE", "1")
    monkeypatch.setattr(ops, "SPLIT_CACHE_MIN_TILES", 0)
    monkeypatch.setitem(ops._SPLIT_POLICY, "mode", "train")     # (what a model's training forward sets: split_refresh_all is a training-step pass)
    M, C, B, H, W = 256, 128, 2, 24, 32
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, C, H, W, generator=g).cuda()
    w = torch.nn.Parameter((torch.randn(M, C, 1, 1, generator=g) * C ** -0.5).cuda())

    def run(wt):
        return ops.conv_fwd_raw(x, wt.view(M, C), None, None, M, 1, 1, 0, H, W)

    def ref(wt):
        return F.conv2d(x.double().cpu(), wt.detach().double().cpu())
    n0 = len(ops._SPLIT_IMG)
    with torch.no_grad():
        close(run(w), ref(w), "first use")
        assert len(ops._SPLIT_IMG) == n0 + 1 and w.data_ptr() in ops._SPLIT_IMG
        img = ops._SPLIT_IMG[w.data_ptr()].images
        before = img.clone()
        close(run(w), ref(w), "cached")
        assert torch.equal(img, before)
        w.mul_(-3.0)                                            # version bump, no refresh: the launch site notices
        close(run(w), ref(w), "after an in-place update")
        assert not torch.equal(img, before)
        w.add_(0.25)
        ops.split_refresh_all()                                 # the per-step batch
        after = img.clone()
        close(run(w), ref(w), "after the batched refresh")
        assert torch.equal(img, after)
        tmp = (torch.randn(M, C, generator=g) * C ** -0.5).cuda()          # not a parameter: cut per launch, not cached
        close(run(tmp), ref(tmp.view(M, C, 1, 1)), "temporary weight")
        assert tmp.data_ptr() not in ops._SPLIT_IMG
    ptr = w.data_ptr()
    del w
    import gc
    gc.collect()
    assert ptr not in ops._SPLIT_IMG


def test_split_gemm_never_reads_a_stale_image_after_an_optimizer_step(split_everywhere, monkeypatch):
    """The C side keeps no table of images (include/prn.h): a launch reads the images its caller passes -- ops.split_images validates them
    against the parameter's version counter AND storage address -- or cuts the weight itself.  Sequence the advisor flagged in round 3:
    train-mode launches keep images of an FPN lateral weight, optimizer.step() changes it, then the INFERENCE block (ops.fpn_level -> C)
    runs on a split-planned shape: it must see the new weight.  Same after the parameter's storage is replaced (p.data = ...)."""
    ops = split_everywhere
    monkeypatch.setattr(ops, "SPLIT_CACHE", "1")
    monkeypatch.setattr(ops, "SPLIT_CACHE_MIN_TILES", 0)
    monkeypatch.setitem(ops._SPLIT_POLICY, "mode", "train")
    B, C, H, W, Fc = 2, 256, 24, 32, 256
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, C, H, W, generator=g).cuda()
    wl = torch.nn.Parameter((torch.randn(Fc, C, 1, 1, generator=g) * C ** -0.5).cuda())
    bl = torch.randn(Fc, generator=g).cuda()
    wo = torch.nn.Parameter((torch.randn(Fc, Fc, 3, 3, generator=g) * (9 * Fc) ** -0.5).cuda())
    bo = torch.randn(Fc, generator=g).cuda()

    def ref():
        lat = F.conv2d(x.double().cpu(), wl.detach().double().cpu(), bl.double().cpu())
        return lat, F.relu(F.conv2d(lat, wo.detach().double().cpu(), bo.double().cpu(), padding=1))
    with torch.no_grad():
        y = ops.conv_fwd_raw(x, wl.view(Fc, C), bl, None, Fc, 1, 1, 0, H, W)          # a training-mode launch: images of wl are cut and kept
        assert wl.data_ptr() in ops._SPLIT_IMG
        close(y, ref()[0], "lateral, training launch")
        wl.add_(torch.randn(wl.shape, generator=g).cuda() * 0.05)                    # optimizer.step()
        wo.mul_(1.5)
        monkeypatch.setitem(ops._SPLIT_POLICY, "mode", "eval")
        lat, p = ops.fpn_level(x, wl, bl, None, wo, bo, True)
        close(lat, ref()[0], "fpn lateral after the update")
        close(p, ref()[1], "fpn output after the update")
        monkeypatch.setitem(ops._SPLIT_POLICY, "mode", "train")
        old_ptr = wl.data_ptr()
        wl.data = (torch.randn(wl.shape, generator=g) * C ** -0.5).cuda()            # storage replaced, version counter unchanged
        y = ops.conv_fwd_raw(x, wl.view(Fc, C), bl, None, Fc, 1, 1, 0, H, W)
        close(y, ref()[0], "lateral after the storage moved")
        if old_ptr != wl.data_ptr():
            e = ops._SPLIT_IMG.get(old_ptr)
            assert e is None or ops._split_state(e) is None


def test_two_threads_with_different_options_do_not_interfere():
    """No process-wide mode in the library: thread A runs 1x1 convolutions with the split kernel on every launch, thread B with fp32 MFMAs
    only, concurrently on two streams through the raw C ABI; each must reproduce its own single-threaded result bit for bit."""
    import ctypes, threading
    from planerecnet_amd import ops, _lib
    lib, _p, check = ops.lib, ops._p, ops.check
    M, K, B, HW = 256, 128, 2, 1280
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, K, HW, generator=g).cuda()
    w = (torch.randn(M, K, generator=g) * K ** -0.5).cuda()

    def make(mode):
        o = _lib.GemmOpts(mode, 16, 3, 300, 4.0, 0, 0, 0)
        d = _lib.ConvDesc(B, K, 1, HW, M, 1, 1, 1, 0, 1, HW, 0, 1, 0, 0, 0, 0, 0, o)
        nb = lib.prn_conv2d_fwd_ws_bytes(ctypes.byref(d))
        return d, torch.empty(max(nb, 16) // 4, device="cuda")

    def run(d, ws, st, y):
        check(lib.prn_conv2d_fwd(ctypes.byref(d), _p(x), _p(w), None, None, _p(y), _p(ws), ctypes.c_void_p(st.cuda_stream)), "conv")
    torch.cuda.synchronize()
    want = {}
    for mode in (2, 0):
        d, ws = make(mode)
        y = torch.empty(B, M, HW, device="cuda")
        run(d, ws, torch.cuda.current_stream(), y)
        torch.cuda.synchronize()
        want[mode] = y.clone()
    assert not torch.equal(want[2], want[0])                     # (different summation orders: the two kernels are distinguishable)
    bad = []

    def worker(mode):
        d, ws = make(mode)
        st = torch.cuda.Stream()
        y = torch.empty(B, M, HW, device="cuda")
        for _ in range(50):
            y.fill_(float("nan"))
            st.wait_stream(torch.cuda.current_stream())
            run(d, ws, st, y)
            st.synchronize()
            if not torch.equal(y, want[mode]):
                bad.append(mode)
    ts = [threading.Thread(target=worker, args=(m,)) for m in (2, 0)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert bad == []


def _wgrad_errors(dw, dy, x):
    """(max, rms, mean) of (dw - fp64 reference) / sum_n |dy||x| per element; dy [B,M,HW], x [B,C,HW]."""
    dyd, xd = dy.double().cpu(), x.double().cpu()
    ref = torch.einsum("bmp,bcp->mc", dyd, xd)
    mag = torch.einsum("bmp,bcp->mc", dyd.abs(), xd.abs())
    e = (dw.double().cpu().reshape(ref.shape) - ref) / (mag + 1e-300)
    return float(e.abs().max()), float(e.pow(2).mean().sqrt()), float(e.mean())


@pytest.mark.parametrize("M,C,B,H,W", [
    (1024, 256, 2, 30, 40),      # stage-3 expand
    (256, 1024, 2, 30, 40),      # stage-3 reduce
    (200, 136, 3, 15, 20),       # tails in both tile dimensions; 300 pixels per image: chunks straddle images
    (128, 128, 1, 8, 8),         # one tile, four chunks
    (2048, 512, 8, 15, 20),      # stage 4
    (256, 256, 1, 120, 160),     # FPN-sized map: many chunks per workgroup
])
def test_wgrad16_is_an_fp32_weight_gradient(M, C, B, H, W):
    """csrc/prn_wgrad16.hip (both operands cut into two fp16 pieces inside the launch, per-row power-of-two scaling that follows the running
    maximum, three fp16 MFMA products, fp32 accumulate) against fp64, next to the fp32 MFMA kernel on the same operands: error at fp32
    rounding level -- max <= 2x the fp32 kernel's (or 4e-7 of sum|dy||x|), rms <= 1.5x (or 2.5e-8)."""
    from planerecnet_amd import ops
    g = torch.Generator().manual_seed(M + C + H)
    x = torch.relu(torch.randn(B, C, H, W, generator=g)).cuda()                 # post-ReLU activations
    dy = (torch.randn(B, M, H, W, generator=g) * torch.exp(torch.randn(1, M, 1, 1, generator=g))).cuda()      # per-channel gradient scales
    out = {}
    old = ops.set_split_gemm(wgrad=2)
    try:
        for mode in (2, 0):
            ops.set_split_gemm(wgrad=mode)
            dw = ops.conv_wgrad_raw(x, dy, M, 1, 1, 0, ops.IN_ZERO)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(dw).all())
            out[mode] = _wgrad_errors(dw, dy.flatten(2), x.flatten(2))
    finally:
        ops.set_split_gemm(**old)
    (smax, srms, smean), (fmax, frms, fmean) = out[2], out[0]
    print("wgrad16 %s: max %.2e rms %.2e mean %+.1e | fp32 max %.2e rms %.2e mean %+.1e" % ((M, C, B, H, W), smax, srms, smean, fmax, frms, fmean))
    assert smax <= max(2.0 * fmax, 4e-7) and srms <= max(1.5 * frms, 2.5e-8), out


def test_wgrad16_rows_that_grow_vanish_and_overflow_fp16():
    """The per-row running scale: rows whose magnitude jumps by 2^20 halfway through the pixels (accumulators rescaled), all-zero rows, rows of
    1e-20 / 1e+15 magnitudes (far outside fp16's range before scaling), a row that is zero except for its last pixel, negative rows."""
    from planerecnet_amd import ops
    M, C, B, H, W = 256, 128, 2, 20, 24
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, C, H, W, generator=g)
    dy = torch.randn(B, M, H, W, generator=g)
    dy[:, 3] = 0.0
    dy[:, 5] *= 1e-20; dy[:, 6] *= 1e15; x[:, 7] *= 1e-15; x[:, 8] *= 1e10      # (products stay inside fp32's NORMAL range: 1e-35 .. 1e25)
    dy[1, 9] *= 2.0 ** 20; x[1, 10, 10:] *= 2.0 ** 24                            # maxima that rise late
    dy[:, 11] = 0.0; dy[1, 11, -1, -1] = -3.0
    x[:, 12] = -x[:, 12].abs()
    dy[0, 13] *= 2.0 ** 20                                                      # and one that falls: early pixels dominate, late ones lose relative precision only
    x, dy = x.cuda(), dy.cuda()
    old = ops.set_split_gemm(wgrad=2)
    try:
        dw = ops.conv_wgrad_raw(x, dy, M, 1, 1, 0, ops.IN_ZERO)
    finally:
        ops.set_split_gemm(**old)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dw).all())
    smax, srms, _ = _wgrad_errors(dw, dy.flatten(2), x.flatten(2))
    assert smax <= 1e-6 and srms <= 3e-8, (smax, srms)
    assert float(dw[3].abs().max()) == 0.0


def test_wgrad16_grouped_and_batched_products():
    """The same kernel behind prn_conv2d_wgrad_grouped (blockIdx.z = layer, pointer tables) and prn_gemm_batched_nt (the 36 products of the
    Winograd weight gradient: partial sums in the fp32 kernel's layout)."""
    from planerecnet_amd import ops
    lib, _p, _stream, check = ops.lib, ops._p, ops._stream, ops.check
    g = torch.Generator().manual_seed(9)
    M, C, B, H, W, G = 256, 256, 2, 30, 40, 3
    xs = [torch.randn(B, C, H, W, generator=g).cuda() for _ in range(G)]
    dys = [torch.randn(B, M, H, W, generator=g).cuda() for _ in range(G)]
    old = ops.set_split_gemm(wgrad=2)
    try:
        dw = ops.conv_wgrad_grouped_raw(xs, dys, M, 1, 1, 0, ops.IN_ZERO)
        torch.cuda.synchronize()
        for i in range(G):
            smax, srms, _ = _wgrad_errors(dw[i], dys[i].flatten(2), xs[i].flatten(2))
            assert smax <= 4e-7 and srms <= 2.5e-8, (i, smax, srms)
        nb, P = 5, 644
        A = torch.randn(nb, M, P, generator=g).cuda()
        Bm = torch.randn(nb, 136, P, generator=g).cuda()
        oref = ops.opts_ref()
        S = lib.prn_gemm_batched_nt_splits(M, 136, P, nb, oref)
        part = torch.full((S, nb, M, 136), float("nan"), device="cuda")
        check(lib.prn_gemm_batched_nt(M, 136, P, nb, _p(A), _p(Bm), _p(part), oref, _stream()), "batched nt")
        torch.cuda.synchronize()
        got = part.sum(0)
        for zi in range(nb):
            smax, srms, _ = _wgrad_errors(got[zi], A[zi][None], Bm[zi][None])
            assert smax <= 4e-7 and srms <= 2.5e-8, (zi, smax, srms)
    finally:
        ops.set_split_gemm(**old)


@pytest.mark.parametrize("B,H,W", [(2, 480, 640), (3, 7, 9), (1, 2, 2)])
def test_lava_gt_weights_equal_the_tensor_formulation(B, H, W):
    """prn_lava_gt_weights (one launch) against the tensor formulation of models/functions/losses.py:288-329 as planerecnet_amd.losses keeps it
    (reflection pad, Sobel / 8, square, / clamp(depth)^2, clamp, threshold): the same operations in the same order -- bit for bit against the
    formulation evaluated on the CPU, within the device division's last bits against the formulation evaluated as device tensor operations -- incl. depths below
    the resolution clamp, flat regions (weight exactly 0) and steps (weight clamped to 1e-2)."""
    import ctypes
    from planerecnet_amd import losses, ops
    d = dev()
    g = torch.Generator().manual_seed(5)
    gt = (0.5 + 4.0 * torch.rand(B, 1, H, W, generator=g))
    gt[:, :, : H // 2, : W // 3] = 2.0                                  # a flat region
    gt[:, :, H // 2:, W // 2:] *= 0.001                                # below the depth resolution
    res = 0.02

    def tensor_formulation(t):
        r = losses.sobel_sq(t) / t.clamp(min=res) ** 2
        r = r.clamp(max=1e-2)
        return torch.where(r < 1e-4, torch.zeros_like(r), r)

    ref = tensor_formulation(gt)                                        # on the CPU: IEEE round-to-nearest in every operation, as the kernel's explicit _rn arithmetic
    gt = gt.to(d)
    out = torch.empty_like(gt)
    ops.check(ops.lib.prn_lava_gt_weights(ops._p(gt), ops._p(out), B, H, W, ctypes.c_float(res), ops._stream()), "prn_lava_gt_weights")
    assert torch.equal(out.cpu(), ref), (out.cpu() - ref).abs().max().item()
    assert (out == 0).any() and (out == 1e-2).any() if H > 2 else True
    # the same formulation as device tensor operations: its division is not correctly rounded (an ulp or two), which can also move a value across a threshold
    dev_ref = tensor_formulation(gt)
    off = (out - dev_ref).abs() > 4e-7 * dev_ref.abs()
    assert off.float().mean().item() <= 1e-5, off.float().mean().item()
    assert ((out - dev_ref).abs()[off] <= 1.01e-4).all()               # (a value that crossed the 1e-4 threshold by the last bit)
    ref = ref.to(d)
    s = ops.channel_sum(out.view(1, B, H, W))
    assert torch.allclose(s, ref.flatten(1).sum(1), rtol=1e-5)

"""CPU: property tests that pin oracle/dcn_ref.py (torchvision is absent: 'parity unpinned' at the
deform_conv2d boundary; these are the known-answer cases of SURVEY.md section 7 step 0)."""
import pytest
import torch
import torch.nn.functional as F

from oracle.dcn_ref import deform_conv2d_ref, deform_sample_cols


def rnd(*s, seed=0, dtype=torch.float64):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed), dtype=dtype)


@pytest.mark.parametrize("stride", [1, 2])
def test_zero_offset_unit_mask_is_conv2d(stride):
    x, w, b = rnd(2, 5, 9, 11), rnd(7, 5, 3, 3, seed=1), rnd(7, seed=2)
    Ho, Wo = (9 + 2 - 3) // stride + 1, (11 + 2 - 3) // stride + 1
    off = torch.zeros(2, 18, Ho, Wo, dtype=x.dtype)
    m = torch.ones(2, 9, Ho, Wo, dtype=x.dtype)
    y = deform_conv2d_ref(x, off, m, w, b, stride, 1)
    assert torch.allclose(y, F.conv2d(x, w, b, stride=stride, padding=1), atol=1e-12)


def test_integer_offsets_are_shifted_conv():
    x, w = rnd(1, 3, 8, 10), rnd(4, 3, 3, 3, seed=1)
    off = torch.zeros(1, 18, 8, 10, dtype=x.dtype)
    off[:, 0::2] = 1.0      # dy = +1 for every tap
    off[:, 1::2] = -2.0     # dx = -2
    y = deform_conv2d_ref(x, off, None, w, None, 1, 1)
    xs = torch.zeros_like(x)
    xs[:, :, :-1, 2:] = x[:, :, 1:, :-2]       # xs[h,w] = x[h+1, w-2], zero outside
    # border outputs differ by construction: the shifted taps of the zero-pad ring land inside the image
    assert torch.allclose(y[:, :, 1:-1, 1:-1], F.conv2d(xs, w, padding=1)[:, :, 1:-1, 1:-1], atol=1e-12)


def test_linear_ramp_is_reproduced_exactly_inside():
    H, W = 12, 14
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    x = (0.5 * yy - 0.25 * xx + 3.0).view(1, 1, H, W)
    off = rnd(1, 18, H, W, seed=3) * 0.9
    cols = deform_sample_cols(x, off, None, 3, 3, 1, 1)[0, 0]       # [9,H,W]
    for k in range(9):
        i, j = divmod(k, 3)
        sy = yy - 1 + i + off[0, 2 * k]
        sx = xx - 1 + j + off[0, 2 * k + 1]
        interior = (sy >= 0) & (sy <= H - 1) & (sx >= 0) & (sx <= W - 1)
        expect = 0.5 * sy - 0.25 * sx + 3.0
        assert torch.allclose(cols[k][interior], expect[interior], atol=1e-12)


def test_out_of_bounds_rule():
    x = torch.ones(1, 1, 4, 4, dtype=torch.float64)
    off = torch.zeros(1, 18, 4, 4, dtype=torch.float64)
    off[0, 8, 0, 0] = -1.0 - 0.0     # centre tap at (0,0): y = -1 exactly -> zero
    off[0, 8, 1, 1] = -1.5            # y = -0.5: only the lower corners (row 0) count -> 0.5
    off[0, 9, 2, 2] = 1.5             # x = 3.5: only left corners (col 3) -> 0.5
    off[0, 9, 3, 3] = 1.0             # x = 4 = W -> zero
    c = deform_sample_cols(x, off, None, 3, 3, 1, 1)[0, 0, 4]
    assert c[0, 0] == 0 and abs(c[1, 1] - 0.5) < 1e-12 and abs(c[2, 2] - 0.5) < 1e-12 and c[3, 3] == 0


def test_mask_scales_columns_and_offset_order_is_dy_dx():
    x = rnd(1, 2, 6, 6)
    off = torch.zeros(1, 18, 6, 6, dtype=x.dtype)
    off[0, 2 * 4] = 1.0    # centre tap dy=+1
    m = torch.full((1, 9, 6, 6), 0.5, dtype=x.dtype)
    c = deform_sample_cols(x, off, m, 3, 3, 1, 1)
    assert torch.allclose(c[0, :, 4, :-1, :], 0.5 * x[0, :, 1:, :], atol=1e-12)


def test_gradcheck_fp64():
    x = rnd(1, 2, 5, 6).requires_grad_(True)
    off = (rnd(1, 18, 5, 6, seed=1) * 0.7 + 0.13).requires_grad_(True)   # keep away from integer kinks
    m = torch.sigmoid(rnd(1, 9, 5, 6, seed=2)).requires_grad_(True)
    w = rnd(3, 2, 3, 3, seed=3).requires_grad_(True)
    b = rnd(3, seed=4).requires_grad_(True)
    frac = (off.detach() - off.detach().floor())
    assert ((frac > 1e-3) & (frac < 1 - 1e-3)).all()
    assert torch.autograd.gradcheck(lambda *a: deform_conv2d_ref(a[0], a[1], a[2], a[3], a[4], 1, 1), (x, off, m, w, b),
                                    eps=1e-6, atol=1e-6)


@pytest.mark.parametrize("stride,pad,with_mask", [(1, 1, True), (2, 1, True), (1, 0, False)])
def test_matches_atens_grid_sampler_for_fractional_offsets(stride, pad, with_mask):
    """A second, independent implementation of the sampling rule: torchvision's bilinear_interpolate (zero outside (-1, H) x (-1, W),
    corners outside the image contribute 0) is F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True) at the pixel
    coordinates (ho*stride - pad + ki + dy, wo*stride - pad + kj + dx) -- ATen's grid sampler shares no code with oracle/dcn_ref.py.
    Offsets up to +-3 px, so that samples cross the border and leave the image; offset channel order (dy, dx) per tap, taps row-major,
    weight columns (c, ki, kj): torchvision.ops.deform_conv2d's layout."""
    B, C, H, W, M = 2, 5, 11, 13, 7
    x, w, b = rnd(B, C, H, W), rnd(M, C, 3, 3, seed=1), rnd(M, seed=2)
    Ho, Wo = (H + 2 * pad - 3) // stride + 1, (W + 2 * pad - 3) // stride + 1
    off = rnd(B, 18, Ho, Wo, seed=3) * 1.5
    off[:, :, 0, 0] = 3.0                                             # far outside for the corner pixel
    m = torch.sigmoid(rnd(B, 9, Ho, Wo, seed=4)) * 2 if with_mask else None
    y = deform_conv2d_ref(x, off, m, w, b, stride, pad)
    ho = torch.arange(Ho, dtype=torch.float64).view(1, Ho, 1) * stride - pad
    wo = torch.arange(Wo, dtype=torch.float64).view(1, 1, Wo) * stride - pad
    ref = b.view(1, M, 1, 1).expand(B, M, Ho, Wo).clone()
    for t in range(9):
        ki, kj = divmod(t, 3)
        py = ho + ki + off[:, 2 * t]                                   # [B, Ho, Wo] pixel coordinates
        px = wo + kj + off[:, 2 * t + 1]
        grid = torch.stack([2 * px / (W - 1) - 1, 2 * py / (H - 1) - 1], dim=-1)      # (x, y) in [-1, 1], align_corners=True
        s = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=True)      # [B, C, Ho, Wo]
        if m is not None:
            s = s * m[:, t:t + 1]
        ref = ref + torch.einsum("mc,bchw->bmhw", w[:, :, ki, kj], s)
    assert torch.allclose(y, ref, atol=1e-10), float((y - ref).abs().max())

"""TEST INFRASTRUCTURE -- CPU restatement of modulated deformable convolution (DCNv2).

Restates `torchvision.ops.deform_conv2d` (torchvision 0.11.1, pinned by the reference's
environment.yml:149; sole call site /root/reference/models/dcn.py:59-66) for the only configuration
the reference uses: groups = offset_groups = 1, dilation 1, square kernel, symmetric stride/pad.

PARITY UNPINNED at this boundary: torchvision is a third-party dependency that is neither under
/root/reference nor installed in this image, and the reference holds no tests/golden vectors for
it.  The restatement follows the published algorithm:

  tap k = i*kw + j;  offset channel 2k = dy, 2k+1 = dx
  y = ho*stride - pad + i + dy ;  x = wo*stride - pad + j + dx
  sample = 0 if (y <= -1 or y >= H or x <= -1 or x >= W) else bilinear with each of the four
           corners contributing only when it lies inside [0,H-1]x[0,W-1] (zero padding, no clamp)
  col[c*kh*kw + k, b, ho, wo] = mask[b,k,ho,wo] * sample(input[b,c], y, x)
  out[b,co,ho,wo] = sum_{c,k} weight[co,c,k] * col[...] + bias[co]

and is pinned by tests/test_oracle_dcn.py (zero offsets == conv2d; integer offsets == shifted conv;
linear-ramp analytic value; out-of-bounds rule; fp64 gradcheck; for fractional offsets that cross and
leave the image: equal to 1e-10 to ATen's grid sampler evaluated at the same pixel coordinates -- an
implementation of the same published rule that shares no code with this file).  It is written with differentiable
torch ops so autograd provides d-input / d-offset / d-mask / d-weight / d-bias (floor() has zero
gradient, which reproduces torchvision's one-sided coordinate derivative).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch


def deform_sample_cols(inp, offset, mask, kh, kw, stride, pad):
    """Returns modulated columns [B, C, kh*kw, Ho, Wo]."""
    B, C, H, W = inp.shape
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    K = kh * kw
    assert offset.shape == (B, 2 * K, Ho, Wo), (offset.shape, (B, 2 * K, Ho, Wo))
    dt, dev = inp.dtype, inp.device
    ho = torch.arange(Ho, dtype=dt, device=dev).view(1, 1, Ho, 1)
    wo = torch.arange(Wo, dtype=dt, device=dev).view(1, 1, 1, Wo)
    ki = torch.arange(kh, dtype=dt, device=dev).repeat_interleave(kw).view(1, K, 1, 1)
    kj = torch.arange(kw, dtype=dt, device=dev).repeat(kh).view(1, K, 1, 1)
    off = offset.view(B, K, 2, Ho, Wo)
    y = ho * stride - pad + ki + off[:, :, 0]            # [B,K,Ho,Wo]
    x = wo * stride - pad + kj + off[:, :, 1]
    inside = (y > -1) & (y < H) & (x > -1) & (x < W)
    y0 = torch.floor(y)
    x0 = torch.floor(x)
    ly = y - y0
    lx = x - x0
    hy = 1 - ly
    hx = 1 - lx
    y0i = y0.long()
    x0i = x0.long()
    y1i = y0i + 1
    x1i = x0i + 1
    flat = inp.reshape(B, C, H * W)

    def corner(yi, xi, wgt):
        ok = inside & (yi >= 0) & (yi <= H - 1) & (xi >= 0) & (xi <= W - 1)
        idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).view(B, 1, -1).expand(B, C, -1)
        v = torch.gather(flat, 2, idx).view(B, C, K, Ho, Wo)
        return v * (wgt * ok.to(dt)).unsqueeze(1)

    val = corner(y0i, x0i, hy * hx) + corner(y0i, x1i, hy * lx) + corner(y1i, x0i, ly * hx) + corner(y1i, x1i, ly * lx)
    if mask is not None:
        val = val * mask.view(B, 1, K, Ho, Wo)
    return val


def deform_conv2d_ref(inp, offset, mask, weight, bias, stride, pad):
    """DCNv2 forward. inp [B,C,H,W], offset [B,2*K,Ho,Wo], mask [B,K,Ho,Wo] or None,
    weight [Co,C,kh,kw], bias [Co] or None."""
    Co, C, kh, kw = weight.shape
    cols = deform_sample_cols(inp, offset, mask, kh, kw, stride, pad)
    B, _, K, Ho, Wo = cols.shape
    out = torch.matmul(weight.reshape(Co, C * K), cols.reshape(B, C * K, Ho * Wo)).view(B, Co, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, Co, 1, 1)
    return out

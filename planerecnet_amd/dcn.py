"""Modulated deformable convolution module (reference: models/dcn.py:11-67).

Same parameters / state-dict keys as the reference module (`offset_conv`, `modulator_conv`, `regular_conv`;
zero-initialised offset and modulator convs, dcn.py:32-43).  Forward differs in *how* it is computed, not
in what: the 18-channel offset conv and the 9-channel modulator conv read the same input with the same
geometry, so they run as ONE 27-channel implicit-GEMM launch; the clamp(+-max(h,w)/4) and 2*sigmoid are
folded into the sampling kernel (prn_dcn_sample), whose output feeds the MFMA contraction.
"""
import torch
from torch import nn

from . import ops


class DeformableConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False):
        super().__init__()
        if kernel_size != 3 or padding != 1:
            raise NotImplementedError("the hot path only uses 3x3 / pad 1 deformable convolutions")
        self.stride, self.padding = stride, padding
        self.offset_conv = nn.Conv2d(in_channels, 2 * 9, 3, stride=stride, padding=1, bias=True)
        self.modulator_conv = nn.Conv2d(in_channels, 9, 3, stride=stride, padding=1, bias=True)
        for m in (self.offset_conv, self.modulator_conv):
            nn.init.zeros_(m.weight)
            nn.init.zeros_(m.bias)
        self.regular_conv = nn.Conv2d(in_channels, out_channels, 3, stride=stride, padding=1, bias=bias)

    def forward(self, x, fold=None):
        """fold: an eval-mode BatchNorm2d to fold into the contraction (inference; the output is then ReLU'd as well)."""
        h, w = x.shape[2:]
        w27 = torch.cat([self.offset_conv.weight, self.modulator_conv.weight], 0)
        b27 = torch.cat([self.offset_conv.bias, self.modulator_conv.bias], 0)
        if fold is not None:
            from .backbone import folded_bn
            wf, bf = folded_bn(self.regular_conv.weight, self.regular_conv.bias, fold)
            om = ops.conv2d(x, w27, b27, stride=self.stride, pad=1)
            return ops.deform_conv2d(x, om, wf, bf, self.stride, max(h, w) / 4.0, relu=True)
        return ops.deform_conv_block(x, w27, b27, self.regular_conv.weight, self.regular_conv.bias, self.stride, max(h, w) / 4.0)
